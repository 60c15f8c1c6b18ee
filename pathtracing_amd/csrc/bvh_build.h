// bvh_build.h — host-side BVH builder of libptrt (binned SAH BVH2, optional collapse to BVH4),
// emitting the docs/SPEC.md §4.1 blob. The reference has no acceleration structure.
#pragma once
#include <stdint.h>
#include <vector>

namespace ptrt {

struct BvhSlot { float lo[3]; int32_t ref; float hi[3]; uint32_t aux; };                              // 32 B
struct BvhTri { float v0[3]; uint32_t id; float e1[3]; uint32_t mat; float e2[3]; uint32_t pad; };    // 48 B

struct BvhBlob {
    uint32_t width = 0;               // 2, 4 or 8
    std::vector<BvhSlot> slots;       // n_nodes * width
    std::vector<BvhTri> tris;         // leaf order
    uint32_t n_nodes = 0, max_depth = 0, stack_need = 0;
    float sah_cost = 0.f;
    double build_ms = 0.0;
};

// verts9: 9 floats per triangle; mats may be null (all 0). width: 2, 4 or 8.
// octant_slots (width 8): children placed in slots by the octant of the node they sit in (layout BVH8O, docs/SPEC.md §4.1)
void build_bvh(const float *verts9, const uint32_t *mats, uint32_t n_tris, uint32_t width, BvhBlob &out, bool octant_slots = false);

// Binary LBVH as the device builder (lbvh.hip) returns it: n leaves of one triangle each in Morton order, n-1 internal nodes, node 0 = root.
struct BinaryBvh {
    std::vector<uint32_t> order;        // triangle index of leaf j
    std::vector<int32_t> left, right;   // children of internal node i: >= 0 internal node, < 0 leaf ~j
    std::vector<uint32_t> first, last;  // leaf range [first, last] covered by internal node i
    std::vector<float> box;             // 6 floats per internal node: union of the padded triangle boxes below it
    double device_ms = 0.0;             // upload + kernels + sort + read-back
};
// pack a device-built binary tree into a width-2/4 blob (subtrees of <= 4 triangles become leaves)
void build_bvh_from_binary(const BinaryBvh &bt, const float *verts9, const uint32_t *mats, uint32_t n_tris, uint32_t width, BvhBlob &out, bool octant_slots = false);

// Binned-SAH binary tree over n boxes (6 floats each), one box per leaf: children >= 0 internal node, < 0 leaf ~i (box i); node k's box in
// node_boxes6. Used for the top storey of the GPU builder (a few thousand LBVH clusters).
void build_sah_over_boxes(const float *boxes6, uint32_t n, std::vector<int32_t> &left, std::vector<int32_t> &right, std::vector<float> &node_boxes6, int32_t &root);

// What the GPU builder leaves on the device when it also packs the tree (lbvh.hip build_lbvh_blob4q_device): BVH4Q nodes and 64-byte
// triangle records in hipMalloc-ed arrays the caller takes over.
struct DeviceBlob4Q {
    void *nodes = nullptr; void *tris = nullptr; // n_nodes x 64 B, n_tris x 64 B
    uint32_t n_nodes = 0, max_depth = 0, stack_need = 0;
    float sah_cost = 0.f;
    double device_ms = 0.0;
};

// BVH4Q (layout id 68): repack a width-4 blob into 64-byte nodes with 8-bit child boxes:
//   +0 origin f32[3] | +12 exponent u8[3],0 | +16 ref i32[4] | +32 qlo_x,qlo_y,qlo_z u8[4] each | +44 qhi_x,qhi_y,qhi_z | +56 pad
void quantize_bvh4(const BvhBlob &in, std::vector<uint8_t> &out);
// BVH8Q (layout id 72): a width-8 blob in 128-byte nodes (96 used), the same scheme with 8 children:
//   +0 origin f32[3] | +12 exponent u8[3],0 | +16 ref i32[8] | +48 qlo_x,qlo_y,qlo_z u8[8] each | +72 qhi_x,qhi_y,qhi_z | +96 pad
void quantize_bvh8(const BvhBlob &in, std::vector<uint8_t> &out);

} // namespace ptrt
