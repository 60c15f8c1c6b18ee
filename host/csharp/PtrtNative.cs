// P/Invoke layer over libptrt.so (include/ptrt.h). NOT compiled in this repository's image (no dotnet);
// written against the frozen header so a maintainer of chairclr/PathTracing can drop it into RayTracing/Graphics/.
// RayTracing.csproj already sets AllowUnsafeBlocks (RayTracing.csproj:8) and net8.0 (:5).
using System;
using System.Runtime.InteropServices;

namespace RayTracing.Graphics;

public enum PtStatus : int { Ok = 0, InvalidArgument, NoDevice, Hip, OutOfMemory, NotCommitted, Unsupported, Internal }
public enum PtMode : uint { ReferenceSphere = 0, PathTrace = 1 }
public enum PtMaterialKind : uint { Lambert = 0, Metal = 1, Dielectric = 2 }
public enum PtSceneKind : uint { Cornell = 0, CornellGlass = 1, TriangleSoup = 2, CornellTess = 3 }

[StructLayout(LayoutKind.Sequential)] public unsafe struct PtDeviceDesc { public int DeviceOrdinal; public void* Stream; public uint Flags, Reserved; }
[StructLayout(LayoutKind.Sequential)] public unsafe struct PtMaterial { public uint Kind; public fixed float Albedo[3]; public fixed float Emission[3]; public float Roughness, Ior; public fixed uint Pad[3]; }
[StructLayout(LayoutKind.Sequential)] public unsafe struct PtCamera { public fixed float Origin[3]; public fixed float Forward[3]; public fixed float Right[3]; public fixed float Up[3]; public float Scale, Cx, Cy; public uint Jitter; }
[StructLayout(LayoutKind.Sequential)] public unsafe struct PtRenderParams
{
    public uint Width, Height, Spp, MaxDepth, RrStart, Seed, SampleOffset, Mode;
    public float RayEps; public uint Rank, NRanks, TileSize, Flags, Streams; public fixed uint Pad[2];
}
[StructLayout(LayoutKind.Sequential)] public unsafe struct PtStats
{
    public ulong Rays, Paths, NodeVisits, TriTests, SphereTests; public uint Iterations, ExtendLaunches;
    public double GpuMs, ExtendMs, ShadeMs, OtherMs; public fixed ulong Reserved[4];
}
[StructLayout(LayoutKind.Sequential)] public struct PtSceneCounts { public ulong NTris, NSpheres, NMats; }

public static unsafe class Ptrt
{
    private const string Lib = "ptrt"; // libptrt.so next to the executable or on LD_LIBRARY_PATH

    [DllImport(Lib)] public static extern uint pt_abi_version();
    [DllImport(Lib)] public static extern PtStatus pt_context_create(PtDeviceDesc* desc, void** ctx);
    [DllImport(Lib)] public static extern void pt_context_destroy(void* ctx);
    [DllImport(Lib)] public static extern sbyte* pt_last_error(void* ctx);
    [DllImport(Lib)] public static extern PtStatus pt_scene_create(void* ctx, void** scene);
    [DllImport(Lib)] public static extern void pt_scene_destroy(void* scene);
    [DllImport(Lib)] public static extern PtStatus pt_scene_set_triangles(void* s, float* verts9, uint* materialIds, ulong count);
    [DllImport(Lib)] public static extern PtStatus pt_scene_set_spheres(void* s, float* cxyzr, uint* materialIds, ulong count);
    [DllImport(Lib)] public static extern PtStatus pt_scene_set_materials(void* s, PtMaterial* mats, ulong count);
    [DllImport(Lib)] public static extern PtStatus pt_scene_set_camera(void* s, PtCamera* cam);
    [DllImport(Lib)] public static extern PtStatus pt_scene_set_sky(void* s, float* rgb);
    [DllImport(Lib)] public static extern PtStatus pt_scene_commit(void* s, uint bvhWidth);
    [DllImport(Lib)] public static extern PtStatus pt_render(void* ctx, void* scene, PtRenderParams* p, PtStats* stats);
    [DllImport(Lib)] public static extern PtStatus pt_framebuffer_read(void* ctx, float* rgba, ulong nFloats);
    [DllImport(Lib)] public static extern PtStatus pt_framebuffer_read_rgba8(void* ctx, byte* rgba8, ulong nBytes);
    [DllImport(Lib)] public static extern PtStatus pt_scenegen(PtSceneKind kind, uint detail, uint seed, uint width, uint height,
        PtSceneCounts* counts, float* verts9, uint* triMat, float* spheres, uint* sphMat, PtMaterial* mats, PtCamera* cam, float* sky);

    // Same observable behaviour as the reference: every failure is a thrown Exception (e.g. Renderer.cs:1022-1025).
    public static void Check(PtStatus st, void* ctx = null)
    {
        if (st == PtStatus.Ok) return;
        string msg = Marshal.PtrToStringUTF8((IntPtr)pt_last_error(ctx)) ?? "";
        throw new Exception($"ptrt {st}: {msg}");
    }
}
