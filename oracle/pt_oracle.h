/*
 * pt_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Scalar C restatement of the hot path, used as the correctness checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg. Nothing in the product
 * (pathtracing_amd/, include/, host/) may include, link or call this.
 *
 * What it restates:
 *   - pto_reference_sphere(): the reference's only ray kernel,
 *     RayTracing/Assets/Shaders/Source/Ray/Test.hlsl:1-40 in the op order of the committed
 *     Assets/Shaders/Compiled/Ray/Test.spirv, plus the R8G8B8A8Unorm store of
 *     RayTracing/Graphics/Renderer.cs:124.  PINNED by tests/golden/reference_sphere_kat.json
 *     (SURVEY.md §8c known-answer table).
 *   - pto_render(): the path tracer of docs/SPEC.md §2-§6. The reference has no BVH, triangles,
 *     BSDFs, RNG or accumulation, so for this part PARITY IS UNPINNED against the
 *     reference: the spec is this repo's own and this file is its executable form, cross-checked
 *     by analytic tests (furnace, brute force vs BVH, reciprocity).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { PTO_LAMBERT = 0, PTO_METAL = 1, PTO_DIELECTRIC = 2 };
#define PTO_MISS 0xFFFFFFFFu
#define PTO_BVH_EMPTY 0x7fffffff
#define PTO_BVH_LAYOUT_4Q 68u /* bvh_width value of the quantised 64-byte BVH4 node format (SPEC §4.1) */
#define PTO_BVH_LAYOUT_8Q 72u /* bvh_width value of the quantised 128-byte BVH8 node format (SPEC §4.1) */
#define PTO_BVH_LAYOUT_8O 73u /* the same node bytes, children in octant slots, visited in the order slot ^ ray octant (SPEC §4.1) */

typedef struct { uint32_t kind; float albedo[3]; float emission[3]; float roughness; float ior; uint32_t pad[3]; } pto_material; /* 48 B */
typedef struct { float origin[3], forward[3], right[3], up[3]; float scale, cx, cy; uint32_t jitter; } pto_camera;              /* 64 B */
typedef struct {
    uint32_t width, height, spp, max_depth, rr_start, seed, sample_offset, mode;
    float ray_eps; uint32_t rank, nranks, tile_size, flags;
    uint32_t streams; /* partial sums per pixel (SPEC §5); 0 = 1, at most 64 */
    uint32_t pad[2];
} pto_params; /* 64 B, same layout as pt_render_params */

typedef struct {
    uint32_t n_tris;    const float *tri_verts;  const uint32_t *tri_mat;   /* 9 floats per triangle */
    uint32_t n_spheres; const float *spheres;    const uint32_t *sph_mat;   /* cx,cy,cz,r */
    uint32_t n_mats;    const pto_material *mats;
    float sky[3];
    pto_camera cam;
    /* optional acceleration structure (SPEC §4.1). nodes==NULL: brute force over all triangles. */
    uint32_t bvh_width, n_nodes; const void *nodes; const void *tris48;
} pto_scene;

typedef struct {
    uint64_t rays, paths, node_visits, tri_tests, sphere_tests;
    uint64_t primary_misses; /* paths whose camera ray hits nothing (one ray, straight into the sky) */
} pto_stats;

/* SPEC §1. rgba (W*H*4 floats) and/or rgba8 (W*H*4 bytes) may be NULL. */
int pto_reference_sphere(uint32_t w, uint32_t h, float *rgba, uint8_t *rgba8);
uint8_t pto_unorm8(float c);

/* SPEC §5. rgba = H*W*4 floats, row-major. threads<=0: all cores (OpenMP). */
int pto_render(const pto_scene *s, const pto_params *p, int threads, float *rgba, pto_stats *st);

/* own median-split BVH2 builder emitting the SPEC §4.1 blob; caller frees with pto_free. */
int pto_bvh_build(uint32_t n_tris, const float *tri_verts, const uint32_t *tri_mat,
                  uint32_t *n_nodes, void **nodes, void **tris48);
void pto_free(void *p);
/* structural check of a blob: every triangle in exactly one leaf with matching bytes, boxes enclose
 * their subtrees, refs in range, depth bounded. returns 0 if ok, else a negative code. */
int pto_bvh_validate(uint32_t width, uint32_t n_nodes, const void *nodes, const void *tris48,
                     uint32_t n_tris, const float *tri_verts, const uint32_t *tri_mat, uint32_t *max_depth_out);

/* pieces exposed for unit tests */
uint32_t pto_pcg(uint32_t x);
uint32_t pto_path_key(uint32_t seed, uint32_t pixel, uint32_t sample);
float    pto_u01(uint32_t key, uint32_t dim);
void     pto_sincos2pi(float u, float *s, float *c);
/* closest hit of one ray; returns prim id or PTO_MISS; t,u,v out */
uint32_t pto_closest(const pto_scene *s, const float o[3], const float d[3], float *t, pto_stats *st);
/* one BSDF sample in world space (SPEC §5); returns alive flag */
int pto_bsdf_sample(const pto_material *m, const float d[3], const float n[3], int front,
                    float u1, float u2, float u3, float wi[3], float W[3], float *side);
void pto_camera_ray(const pto_camera *c, uint32_t x, uint32_t y, uint32_t key, float o[3], float d[3]);
int pto_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
