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
        if (Ptrt.pt_abi_version() != Ptrt.AbiVersion) throw new Exception($"libptrt has ABI version {Ptrt.pt_abi_version()}, this binding expects {Ptrt.AbiVersion}");
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

    public byte[] ReadFramebufferSrgb8() // what the reference's B8G8R8A8Srgb swapchain shows of that image (SwapChain.cs:157-158)
    {
        byte[] px = new byte[(ulong)Width * Height * 4];
        fixed (byte* p = px) Ptrt.Check(Ptrt.pt_framebuffer_read_srgb8(_ctx, p, (ulong)px.Length), _ctx);
        return px;
    }

    // scheduling knobs (none changes a pixel): read, modify, write back
    public PtTuning Tuning
    {
        get { PtTuning t; Ptrt.Check(Ptrt.pt_context_get_tuning(_ctx, &t), _ctx); return t; }
        set { Ptrt.Check(Ptrt.pt_context_set_tuning(_ctx, &value), _ctx); }
    }

    internal void* Context => _ctx;
    internal void* Scene => _scene;

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

// One frame over several GPUs of the node (include/ptrt.h pt_comm): a HipRenderer per device, the same scene on each, tiles dealt
// round-robin, one ncclGather per frame inside libptrt. The reference is single-device (GraphicsDevice.cs:176-183); this is what stands
// above its Renderer when the node has 8 GPUs. NOT compiled here (no dotnet in the image).
public unsafe class HipMultiRenderer : IDisposable
{
    private readonly HipRenderer[] _r;
    private void* _comm;
    public PtRenderParams Params;
    public PtStats[] LastStats;

    private readonly uint _commFlags;

    // oneDevice: every rank gets its own context on device 0 and the tiles are exchanged by device copies (PtCommFlags.CopyExchange)
    public HipMultiRenderer(int gpus, uint width = 1920, uint height = 1080, bool oneDevice = false)
    {
        _commFlags = oneDevice ? (uint)PtCommFlags.CopyExchange : 0u;
        _r = new HipRenderer[gpus];
        for (int i = 0; i < gpus; i++) { _r[i] = new HipRenderer(width, height); _r[i].Init(oneDevice ? 0 : i); }
        Params = _r[0].Params;
        LastStats = new PtStats[gpus];
    }

    public void LoadSyntheticScene(PtSceneKind kind, uint detail = 0)
    {
        foreach (HipRenderer r in _r) r.LoadSyntheticScene(kind, detail); // replicated scene
        Params.mode = (uint)PtMode.PathTrace;
        void** ctxs = stackalloc void*[_r.Length];
        for (int i = 0; i < _r.Length; i++) ctxs[i] = _r[i].Context;
        if (_comm != null) Ptrt.pt_comm_destroy(_comm);
        void* c; Ptrt.Check(Ptrt.pt_comm_create(ctxs, (uint)_r.Length, 0, _commFlags, &c)); _comm = c;
    }

    public void Render(float delta)
    {
        void** scenes = stackalloc void*[_r.Length];
        for (int i = 0; i < _r.Length; i++) scenes[i] = _r[i].Scene;
        fixed (PtRenderParams* p = &Params) fixed (PtStats* st = LastStats)
            Ptrt.Check(Ptrt.pt_comm_render(_comm, scenes, p, st), _r[0].Context);
    }

    public HipRenderer Root => _r[0]; // holds the assembled frame

    public void Dispose()
    {
        if (_comm != null) Ptrt.pt_comm_destroy(_comm);
        _comm = null;
        foreach (HipRenderer r in _r) r.Dispose();
        GC.SuppressFinalize(this);
    }
}
