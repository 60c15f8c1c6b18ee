// P/Invoke layer over libptrt.so (include/ptrt.h). NOT compiled in this repository's image (no dotnet); kept in step with the
// header by tests/test_csharp_binding.py, which parses this file (struct field order and types, every [DllImport] signature) and
// compares it with include/ptrt.h. Written so a maintainer of chairclr/PathTracing can drop it into RayTracing/Graphics/:
// RayTracing.csproj already sets AllowUnsafeBlocks (RayTracing.csproj:8) and net8.0 (:5).
using System;
using System.Runtime.InteropServices;

namespace RayTracing.Graphics;

public enum PtStatus : int { Ok = 0, InvalidArgument, NoDevice, Hip, OutOfMemory, NotCommitted, Unsupported, Internal }
public enum PtMode : uint { ReferenceSphere = 0, PathTrace = 1 }
public enum PtMaterialKind : uint { Lambert = 0, Metal = 1, Dielectric = 2 }
public enum PtSceneKind : uint { Cornell = 0, CornellGlass = 1, TriangleSoup = 2, CornellTess = 3 }
[Flags] public enum PtFlags : uint { ProfileKernels = 1, CountVisits = 2, ExtendPacked = 4, ExtendSimple = 8, Accumulate = 16, BucketSpecular = 32, SplitKernels = 64, ExtendPool = 128 }
[Flags] public enum PtCommFlags : uint { ForceRccl = 1, CopyExchange = 2 }
public enum PtBvhWidth : uint { Default = 0, W2 = 2, W4 = 4, W4Q = 68, W8Q = 72, W8O = 73, BuildLbvh = 0x100 }

[StructLayout(LayoutKind.Sequential)] public unsafe struct PtDeviceDesc { public int device_ordinal; public void* stream; public uint flags; public uint reserved; }
[StructLayout(LayoutKind.Sequential)] public unsafe struct PtMaterial { public uint kind; public fixed float albedo[3]; public fixed float emission[3]; public float roughness; public float ior; public fixed uint pad[3]; }
[StructLayout(LayoutKind.Sequential)] public unsafe struct PtCamera { public fixed float origin[3]; public fixed float forward[3]; public fixed float right[3]; public fixed float up[3]; public float scale; public float cx; public float cy; public uint jitter; }
[StructLayout(LayoutKind.Sequential)] public unsafe struct PtRenderParams
{
    public uint width; public uint height; public uint spp; public uint max_depth; public uint rr_start; public uint seed; public uint sample_offset; public uint mode;
    public float ray_eps; public uint rank; public uint nranks; public uint tile_size; public uint flags; public uint streams; public fixed uint pad[2];
}
[StructLayout(LayoutKind.Sequential)] public unsafe struct PtStats
{
    public ulong rays; public ulong paths; public ulong node_visits; public ulong tri_tests; public ulong sphere_tests; public uint iterations; public uint extend_launches;
    public double gpu_ms; public double extend_ms; public double shade_ms; public double other_ms; public fixed ulong reserved[4];
}
[StructLayout(LayoutKind.Sequential)] public unsafe struct PtTuning { public uint bounces; public uint loops; public uint finish_below; public uint packed_chunk; public float compact_below; public float sparse_below; public uint sticky_samples; public uint lag; public uint extend_kernel; public uint readback; }
[StructLayout(LayoutKind.Sequential)] public struct PtBvhInfo
{
    public uint width; public uint n_nodes; public uint n_tris; public uint max_depth; public ulong node_bytes; public ulong tri_bytes; public double build_ms; public float sah_cost; public uint stack_need;
}
[StructLayout(LayoutKind.Sequential)] public struct PtTileLayout { public uint tile_size; public uint tiles_x; public uint tiles_y; public uint n_tiles; public uint tiles_mine; public uint tiles_per_rank; public ulong floats_per_tile; }
[StructLayout(LayoutKind.Sequential)] public struct PtSceneCounts { public ulong n_tris; public ulong n_spheres; public ulong n_mats; }

public static unsafe class Ptrt
{
    private const string Lib = "ptrt"; // libptrt.so next to the executable or on LD_LIBRARY_PATH
    public const uint AbiVersion = 2;  // PTRT_ABI_VERSION of the include/ptrt.h this file mirrors; HipRenderer.Init compares it with pt_abi_version()

    [DllImport(Lib)] public static extern uint pt_abi_version();
    [DllImport(Lib)] public static extern PtStatus pt_context_create(PtDeviceDesc* desc, void** ctx);
    [DllImport(Lib)] public static extern void pt_context_destroy(void* ctx);
    [DllImport(Lib)] public static extern sbyte* pt_last_error(void* ctx);
    [DllImport(Lib)] public static extern PtStatus pt_context_get_tuning(void* ctx, PtTuning* tuning);
    [DllImport(Lib)] public static extern PtStatus pt_context_set_tuning(void* ctx, PtTuning* tuning);
    [DllImport(Lib)] public static extern PtStatus pt_scene_create(void* ctx, void** scene);
    [DllImport(Lib)] public static extern void pt_scene_destroy(void* scene);
    [DllImport(Lib)] public static extern PtStatus pt_scene_set_triangles(void* s, float* verts9, uint* material_ids, ulong count);
    [DllImport(Lib)] public static extern PtStatus pt_scene_set_spheres(void* s, float* cxyzr, uint* material_ids, ulong count);
    [DllImport(Lib)] public static extern PtStatus pt_scene_set_materials(void* s, PtMaterial* mats, ulong count);
    [DllImport(Lib)] public static extern PtStatus pt_scene_set_camera(void* s, PtCamera* cam);
    [DllImport(Lib)] public static extern PtStatus pt_scene_set_sky(void* s, float* rgb);
    [DllImport(Lib)] public static extern PtStatus pt_scene_commit(void* s, uint bvh_width);
    [DllImport(Lib)] public static extern PtStatus pt_scene_bvh_info(void* s, PtBvhInfo* info);
    [DllImport(Lib)] public static extern PtStatus pt_scene_bvh_read(void* s, void* nodes, ulong node_bytes, void* tris48, ulong tri_bytes);
    [DllImport(Lib)] public static extern PtStatus pt_render(void* ctx, void* scene, PtRenderParams* p, PtStats* stats);
    [DllImport(Lib)] public static extern PtStatus pt_framebuffer_read(void* ctx, float* rgba, ulong n_floats);
    [DllImport(Lib)] public static extern PtStatus pt_framebuffer_read_rgba8(void* ctx, byte* rgba8, ulong n_bytes);
    [DllImport(Lib)] public static extern PtStatus pt_framebuffer_read_srgb8(void* ctx, byte* rgba8, ulong n_bytes);
    [DllImport(Lib)] public static extern PtStatus pt_framebuffer_device_ptr(void* ctx, void** dptr, ulong* n_floats);
    [DllImport(Lib)] public static extern PtStatus pt_tile_layout_query(PtRenderParams* p, PtTileLayout* layout);
    [DllImport(Lib)] public static extern PtStatus pt_tiles_device_ptr(void* ctx, void** dptr, ulong* n_floats);
    [DllImport(Lib)] public static extern PtStatus pt_assemble_tiles(void* ctx, PtRenderParams* p, void* gathered_dptr, ulong n_floats);
    [DllImport(Lib)] public static extern PtStatus pt_comm_create(void** ctxs, uint n_ranks, uint root, uint flags, void** comm);
    [DllImport(Lib)] public static extern void pt_comm_destroy(void* comm);
    [DllImport(Lib)] public static extern PtStatus pt_comm_render(void* comm, void** scenes, PtRenderParams* p, PtStats* stats);
    [DllImport(Lib)] public static extern PtStatus pt_comm_stage_tiles(void* comm, uint rank);
    [DllImport(Lib)] public static extern PtStatus pt_comm_assemble(void* comm, PtRenderParams* p);
    [DllImport(Lib)] public static extern PtStatus pt_scenegen(uint kind, uint detail, uint seed, uint width, uint height,
        PtSceneCounts* counts, float* verts9, uint* tri_mat, float* spheres, uint* sph_mat, PtMaterial* mats, PtCamera* cam, float* sky);

    // Same observable behaviour as the reference: every failure is a thrown Exception (e.g. Renderer.cs:1022-1025).
    public static void Check(PtStatus st, void* ctx = null)
    {
        if (st == PtStatus.Ok) return;
        string msg = Marshal.PtrToStringUTF8((IntPtr)pt_last_error(ctx)) ?? "";
        throw new Exception($"ptrt {st}: {msg}");
    }
}
