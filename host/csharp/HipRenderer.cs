// Compute-path replacement for RayTracing.Graphics.Renderer (Renderer.cs): same public shape
// (ctor / Init / Update / Render(delta) / Dispose), the Vulkan objects replaced by two opaque libptrt handles.
// NOT compiled here (no dotnet in the image) — see INTEGRATION.md.
using System;

namespace RayTracing.Graphics;

public unsafe class HipRenderer : IDisposable
{
    private bool _disposed;
    private void* _ctx, _scene;
    public PtRenderParams Params;
    public PtStats LastStats;
    public readonly uint Width, Height;

    public HipRenderer(uint width = 1920, uint height = 1080) // App.cs:27 window size
    {
        Width = width; Height = height;
        Params = new PtRenderParams { width = width, height = height, spp = 1, max_depth = 8, rr_start = 3, seed = 0x5EED0001,
                                      mode = (uint)PtMode.ReferenceSphere, ray_eps = 1e-4f, nranks = 1, streams = 8 };
    }

    // Renderer.Init (Renderer.cs:66-84): device + resources + compute pipeline
    public void Init(int device = 0)
    {
        PtDeviceDesc d = new() { device_ordinal = device };
        void* c; Ptrt.Check(Ptrt.pt_context_create(&d, &c)); _ctx = c;
    }

    public void LoadSyntheticScene(PtSceneKind kind, uint detail = 0, uint seed = 0x5EED0001, uint bvhWidth = 0)
    {
        PtSceneCounts n; PtCamera cam; float* sky = stackalloc float[3];
        Ptrt.Check(Ptrt.pt_scenegen((uint)kind, detail, seed, Width, Height, &n, null, null, null, null, null, null, null));
        float[] verts = new float[n.n_tris * 9]; uint[] tmat = new uint[n.n_tris];
        float[] sph = new float[Math.Max(1, n.n_spheres * 4)]; uint[] smat = new uint[Math.Max(1, n.n_spheres)];
        PtMaterial[] mats = new PtMaterial[n.n_mats];
        fixed (float* v = verts, s = sph) fixed (uint* tm = tmat, sm = smat) fixed (PtMaterial* m = mats)
        {
            Ptrt.Check(Ptrt.pt_scenegen((uint)kind, detail, seed, Width, Height, &n, v, tm, s, sm, m, &cam, sky));
            if (_scene != null) Ptrt.pt_scene_destroy(_scene);
            void* sc; Ptrt.Check(Ptrt.pt_scene_create(_ctx, &sc), _ctx); _scene = sc;
            Ptrt.Check(Ptrt.pt_scene_set_triangles(sc, v, tm, n.n_tris), _ctx);
            Ptrt.Check(Ptrt.pt_scene_set_spheres(sc, s, sm, n.n_spheres), _ctx);
            Ptrt.Check(Ptrt.pt_scene_set_materials(sc, m, n.n_mats), _ctx);
            Ptrt.Check(Ptrt.pt_scene_set_camera(sc, &cam), _ctx);
            Ptrt.Check(Ptrt.pt_scene_set_sky(sc, sky), _ctx);
            Ptrt.Check(Ptrt.pt_scene_commit(sc, bvhWidth), _ctx);
        }
        Params.mode = (uint)PtMode.PathTrace;
    }

    public void Update(float deltaTime) { } // empty in the reference too (Renderer.cs:86-89)

    // Renderer.Render (Renderer.cs:933-1004) minus acquire/draw/present: ComputeFrame + fence wait
    public void Render(float delta) => ComputeFrame(delta);

    // Renderer.ComputeFrame (Renderer.cs:1006-1040); pt_render returns after the stream is idle (= WaitForFences, :972)
    private void ComputeFrame(float delta)
    {
        fixed (PtRenderParams* p = &Params) fixed (PtStats* st = &LastStats)
            Ptrt.Check(Ptrt.pt_render(_ctx, Params.mode == (uint)PtMode.PathTrace ? _scene : null, p, st), _ctx);
    }

    public float[] ReadFramebuffer()
    {
        float[] rgba = new float[(ulong)Width * Height * 4];
        fixed (float* p = rgba) Ptrt.Check(Ptrt.pt_framebuffer_read(_ctx, p, (ulong)rgba.Length), _ctx);
        return rgba;
    }

    public byte[] ReadFramebufferRgba8() // the R8G8B8A8Unorm image of Renderer.cs:124
    {
        byte[] px = new byte[(ulong)Width * Height * 4];
        fixed (byte* p = px) Ptrt.Check(Ptrt.pt_framebuffer_read_rgba8(_ctx, p, (ulong)px.Length), _ctx);
        return px;
    }

    protected virtual void Dispose(bool disposing)
    {
        if (_disposed) return;
        if (_scene != null) Ptrt.pt_scene_destroy(_scene);
        if (_ctx != null) Ptrt.pt_context_destroy(_ctx);
        _scene = null; _ctx = null; _disposed = true;
    }
    public void Dispose() { Dispose(true); GC.SuppressFinalize(this); }
    ~HipRenderer() { Dispose(false); }
}
