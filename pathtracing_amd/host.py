"""Host side above the C ABI, mirroring the reference's object shape
`Program -> App -> Renderer.{Init, Update, Render(dt)} -> ComputeFrame(dt)`
(RayTracing/Program.cs:1-9, App.cs:7-68, Graphics/Renderer.cs:59-89, 933-1004, 1006-1040).

The reference's host language is C# (no `dotnet` in this image — see INTEGRATION.md for the P/Invoke binding and
host/csharp for the C# sources); this Python mirror keeps the same names, argument meaning and error behaviour:
every failure raises (the reference does `throw new Exception(...)`, e.g. Renderer.cs:1022-1025), objects are
disposable, `Render(delta)` is synchronous like the compute-fence wait at Renderer.cs:970-972.
Everything here is plumbing around libptrt.so; no pixel is computed in Python.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _native as N

MATERIAL_DTYPE = np.dtype([("kind", "<u4"), ("albedo", "<f4", 3), ("emission", "<f4", 3), ("roughness", "<f4"),
                           ("ior", "<f4"), ("pad", "<u4", 3)])
assert MATERIAL_DTYPE.itemsize == 48


class PtException(Exception):
    """Raised for every non-zero pt_status (the reference throws System.Exception on every Vulkan failure)."""

    def __init__(self, status, message):
        super().__init__(f"ptrt status {status}: {message}")
        self.status = status


def _check(status, ctx=None):
    if status != N.PT_OK:
        msg = N.lib.pt_last_error(ctx)
        raise PtException(status, msg.decode("utf-8", "replace") if msg else "")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class SceneData:
    """Host-side scene arrays — what replaces the shader literals of Test.hlsl:6,8,12,13."""
    verts: np.ndarray = field(default_factory=lambda: np.zeros((0, 9), np.float32))
    tri_mat: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.uint32))
    spheres: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.float32))
    sph_mat: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.uint32))
    mats: np.ndarray = field(default_factory=lambda: np.zeros((0,), MATERIAL_DTYPE))
    cam: N.pt_camera = field(default_factory=N.pt_camera)
    sky: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))


def make_scene(kind, detail=0, seed=0x5EED0001, width=1920, height=1080):
    """Deterministic synthetic scenes C1..C5 from the library's host-only generator."""
    cnt = N.pt_scene_counts()
    cam = N.pt_camera()
    _check(N.lib.pt_scenegen(kind, detail, seed, width, height, C.byref(cnt), None, None, None, None, None, None, None))
    s = SceneData(
        verts=np.zeros((cnt.n_tris, 9), np.float32), tri_mat=np.zeros(cnt.n_tris, np.uint32),
        spheres=np.zeros((cnt.n_spheres, 4), np.float32), sph_mat=np.zeros(cnt.n_spheres, np.uint32),
        mats=np.zeros(cnt.n_mats, MATERIAL_DTYPE), cam=cam, sky=np.zeros(3, np.float32))
    _check(N.lib.pt_scenegen(kind, detail, seed, width, height, C.byref(cnt), _ptr(s.verts), _ptr(s.tri_mat),
                             _ptr(s.spheres), _ptr(s.sph_mat), _ptr(s.mats), C.byref(cam), _ptr(s.sky)))
    return s


def build_bvh_detached(scene, bvh_width=0):
    """Run the library's host-side BVH builder on a detached (context-less) scene; returns (pt_bvh_info, nodes, tris48).
    No device is touched — this is how the builder is checked on machines without a GPU."""
    s = C.c_void_p()
    _check(N.lib.pt_scene_create(None, C.byref(s)))
    try:
        verts = np.ascontiguousarray(scene.verts, np.float32)
        tri_mat = np.ascontiguousarray(scene.tri_mat, np.uint32)
        spheres = np.ascontiguousarray(scene.spheres, np.float32)
        sph_mat = np.ascontiguousarray(scene.sph_mat, np.uint32)
        mats = np.ascontiguousarray(scene.mats)
        _check(N.lib.pt_scene_set_triangles(s, _ptr(verts), _ptr(tri_mat), len(tri_mat)))
        _check(N.lib.pt_scene_set_spheres(s, _ptr(spheres), _ptr(sph_mat), len(sph_mat)))
        _check(N.lib.pt_scene_set_materials(s, _ptr(mats), len(mats)))
        _check(N.lib.pt_scene_set_camera(s, C.byref(scene.cam)))
        _check(N.lib.pt_scene_commit(s, bvh_width))
        info = N.pt_bvh_info()
        _check(N.lib.pt_scene_bvh_info(s, C.byref(info)))
        nodes = np.zeros(max(int(info.node_bytes), 1), np.uint8)
        tris = np.zeros(max(int(info.tri_bytes), 1), np.uint8)
        _check(N.lib.pt_scene_bvh_read(s, _ptr(nodes), info.node_bytes, _ptr(tris), info.tri_bytes))
        return info, nodes[: int(info.node_bytes)], tris[: int(info.tri_bytes)]
    finally:
        N.lib.pt_scene_destroy(s)


def make_params(width, height, spp=1, max_depth=8, rr_start=3, seed=0x5EED0001, mode=N.PT_PATH_TRACE, ray_eps=1e-4,
                rank=0, nranks=1, flags=0, sample_offset=0, streams=1):
    p = N.pt_render_params()
    p.width, p.height, p.spp, p.max_depth, p.rr_start, p.seed = width, height, spp, max_depth, rr_start, seed
    p.sample_offset, p.mode, p.ray_eps, p.rank, p.nranks, p.tile_size, p.flags = sample_offset, mode, ray_eps, rank, nranks, 0, flags
    p.streams = streams
    return p


def tile_layout(params):
    lay = N.pt_tile_layout()
    _check(N.lib.pt_tile_layout_query(C.byref(params), C.byref(lay)))
    return lay


class _DevicePtr:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can alias it (no copy)."""

    def __init__(self, ptr, n_floats):
        self.__cuda_array_interface__ = {"shape": (int(n_floats),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class Window:
    """Headless stand-in for Silk.NET's IWindow (App.cs:25-33): only the framebuffer size survives; MI355X has no display."""

    def __init__(self, width=1920, height=1080, title="ptrt"):
        self.FramebufferSize = (width, height)
        self.Title = title


class Renderer:
    """Mirror of RayTracing.Graphics.Renderer (Renderer.cs:18-1241), compute path only."""

    def __init__(self, window, device_ordinal=0, stream=None):
        self.Window = window
        self._device = device_ordinal
        self._stream = stream
        self._ctx = C.c_void_p()
        self._scene = C.c_void_p()
        self._disposed = False
        self.Params = make_params(*window.FramebufferSize, mode=N.PT_REFERENCE_SPHERE)
        self.LastStats = N.pt_stats()

    # Renderer.Init (Renderer.cs:66-84): GraphicsDevice.Init + CreateResources + CreateComputePipeline
    def Init(self):
        desc = N.pt_device_desc(self._device, self._stream, 0, 0)
        _check(N.lib.pt_context_create(C.byref(desc), C.byref(self._ctx)))

    def GetTuning(self):
        t = N.pt_tuning()
        _check(N.lib.pt_context_get_tuning(self._ctx, C.byref(t)), self._ctx)
        return t

    def SetTuning(self, **kw):
        """Scheduling knobs of the context (include/ptrt.h pt_tuning: bounces, loops, finish_below, packed_chunk, compact_below,
        sparse_below, sticky_samples, lag). None changes a pixel."""
        t = self.GetTuning()
        for k, v in kw.items():
            if not hasattr(t, k):
                raise AttributeError(f"pt_tuning has no field {k!r}")
            setattr(t, k, v)
        _check(N.lib.pt_context_set_tuning(self._ctx, C.byref(t)), self._ctx)

    def SetScene(self, scene, bvh_width=0):
        """Upload a SceneData and build its BVH (the reference has no scene API; Test.hlsl:6,8,12,13 are literals)."""
        ctx = self._ctx
        if self._scene:
            N.lib.pt_scene_destroy(self._scene)
            self._scene = C.c_void_p()
        _check(N.lib.pt_scene_create(ctx, C.byref(self._scene)), ctx)
        s = self._scene
        verts = np.ascontiguousarray(scene.verts, np.float32)
        tri_mat = np.ascontiguousarray(scene.tri_mat, np.uint32)
        spheres = np.ascontiguousarray(scene.spheres, np.float32)
        sph_mat = np.ascontiguousarray(scene.sph_mat, np.uint32)
        mats = np.ascontiguousarray(scene.mats)
        _check(N.lib.pt_scene_set_triangles(s, _ptr(verts), _ptr(tri_mat), len(tri_mat)), ctx)
        _check(N.lib.pt_scene_set_spheres(s, _ptr(spheres), _ptr(sph_mat), len(sph_mat)), ctx)
        _check(N.lib.pt_scene_set_materials(s, _ptr(mats), len(mats)), ctx)
        _check(N.lib.pt_scene_set_camera(s, C.byref(scene.cam)), ctx)
        sky = (C.c_float * 3)(*[float(v) for v in scene.sky])
        _check(N.lib.pt_scene_set_sky(s, C.byref(sky)), ctx)
        _check(N.lib.pt_scene_commit(s, bvh_width), ctx)

    def BvhInfo(self):
        info = N.pt_bvh_info()
        _check(N.lib.pt_scene_bvh_info(self._scene, C.byref(info)), self._ctx)
        return info

    def BvhRead(self):
        """(nodes bytes, tris48 bytes) of the SPEC §4.1 blob, for a checker that wants to traverse the same bytes."""
        info = self.BvhInfo()
        nodes = np.zeros(max(int(info.node_bytes), 1), np.uint8)
        tris = np.zeros(max(int(info.tri_bytes), 1), np.uint8)
        _check(N.lib.pt_scene_bvh_read(self._scene, _ptr(nodes), info.node_bytes, _ptr(tris), info.tri_bytes), self._ctx)
        return nodes[: int(info.node_bytes)], tris[: int(info.tri_bytes)]

    # Renderer.Update (Renderer.cs:86-89) is empty in the reference
    def Update(self, deltaTime):
        pass

    # Renderer.Render (Renderer.cs:933-1004): [acquire] -> ComputeFrame -> wait fence -> [draw, present]
    def Render(self, delta):
        self.ComputeFrame(delta)
        return self.LastStats

    # Renderer.ComputeFrame (Renderer.cs:1006-1040). `delta` is unused there too.
    def ComputeFrame(self, delta):
        scene = self._scene if self.Params.mode == N.PT_PATH_TRACE else None
        stats = N.pt_stats()  # a fresh object per frame: callers keep the stats of earlier frames
        _check(N.lib.pt_render(self._ctx, scene, C.byref(self.Params), C.byref(stats)), self._ctx)
        self.LastStats = stats

    def ReadFramebuffer(self):
        w, h = self.Params.width, self.Params.height
        out = np.empty((h, w, 4), np.float32)
        _check(N.lib.pt_framebuffer_read(self._ctx, _ptr(out), out.size), self._ctx)
        return out

    def ReadFramebufferRGBA8(self):
        w, h = self.Params.width, self.Params.height
        out = np.empty((h, w, 4), np.uint8)
        _check(N.lib.pt_framebuffer_read_rgba8(self._ctx, _ptr(out), out.size), self._ctx)
        return out

    def ReadFramebufferSRGB8(self):
        """The frame as the reference's window would show it (UNORM8 image -> B8G8R8A8Srgb swapchain, SwapChain.cs:157-158)."""
        w, h = self.Params.width, self.Params.height
        out = np.empty((h, w, 4), np.uint8)
        _check(N.lib.pt_framebuffer_read_srgb8(self._ctx, _ptr(out), out.size), self._ctx)
        return out

    def SaveImage(self, path, srgb=False):
        """Image output — what replaces the reference's window (display path Renderer.cs:1042-1121).
        `.ppm`: 8-bit, the clamp-and-round R8G8B8A8Unorm image of Renderer.cs:124 — or, with srgb=True, what the reference's
        sRGB swapchain shows of it (SwapChain.cs:157-158; no tone mapping, values above 1 clip); `.pfm`: linear float radiance,
        bottom-up rows."""
        w, h = self.Params.width, self.Params.height
        if path.lower().endswith(".pfm"):
            rgb = np.ascontiguousarray(self.ReadFramebuffer()[::-1, :, :3], "<f4")
            with open(path, "wb") as f:
                f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
                f.write(rgb.tobytes())
        else:
            rgb = np.ascontiguousarray((self.ReadFramebufferSRGB8() if srgb else self.ReadFramebufferRGBA8())[..., :3])
            with open(path, "wb") as f:
                f.write(b"P6\n%d %d\n255\n" % (w, h))
                f.write(rgb.tobytes())

    def TilesDevice(self):
        """Device view (for torch.as_tensor) of this rank's tile-major radiance sums after a frame."""
        ptr, n = C.c_void_p(), C.c_uint64()
        _check(N.lib.pt_tiles_device_ptr(self._ctx, C.byref(ptr), C.byref(n)), self._ctx)
        return _DevicePtr(ptr.value, n.value)

    def AssembleTiles(self, gathered_ptr, n_floats):
        _check(N.lib.pt_assemble_tiles(self._ctx, C.byref(self.Params), C.c_void_p(gathered_ptr), n_floats), self._ctx)

    # IDisposable pattern (Renderer.cs:1192-1216, 1235-1240)
    def Dispose(self):
        if self._disposed:
            return
        if self._scene:
            N.lib.pt_scene_destroy(self._scene)
            self._scene = C.c_void_p()
        if self._ctx:
            N.lib.pt_context_destroy(self._ctx)
            self._ctx = C.c_void_p()
        self._disposed = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.Dispose()

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass


class Comm:
    """Several ranks of one frame behind one call (include/ptrt.h pt_comm): one Renderer per rank on its own GPU (tiles exchanged by
    one ncclGather per frame inside libptrt), or the same Renderer for every rank (virtual ranks rendered one after the other: the
    partition rehearsed on a single GPU). Every renderer must hold the same scene. The frame lands in renderers[root]."""

    def __init__(self, renderers, root=0, flags=0):
        self.Renderers = list(renderers)
        self.Root = root
        self._comm = C.c_void_p()
        ctxs = (C.c_void_p * len(self.Renderers))(*[r._ctx for r in self.Renderers])
        _check(N.lib.pt_comm_create(ctxs, len(self.Renderers), root, flags, C.byref(self._comm)))

    def Render(self, params):
        """One frame of `params` over all ranks; returns the per-rank pt_stats."""
        n = len(self.Renderers)
        scenes = (C.c_void_p * n)(*[r._scene for r in self.Renderers])
        stats = (N.pt_stats * n)()
        _check(N.lib.pt_comm_render(self._comm, scenes, C.byref(params), stats), self.Renderers[self.Root]._ctx)
        self.Renderers[self.Root].Params = params
        return list(stats)

    def StageTiles(self, rank):
        _check(N.lib.pt_comm_stage_tiles(self._comm, rank), self.Renderers[self.Root]._ctx)

    def Assemble(self, params):
        _check(N.lib.pt_comm_assemble(self._comm, C.byref(params)), self.Renderers[self.Root]._ctx)
        self.Renderers[self.Root].Params = params

    def Dispose(self):
        if self._comm:
            N.lib.pt_comm_destroy(self._comm)
            self._comm = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.Dispose()


class App:
    """Mirror of RayTracing.App (App.cs:7-68): Run() = InitWindow -> InitRenderer -> render loop."""

    def __init__(self, frames=1, width=1920, height=1080, device_ordinal=0):
        self.Window = None
        self.Renderer = None
        self._frames = frames
        self._size = (width, height)
        self._device = device_ordinal
        self._disposed = False

    def Run(self):
        self.InitWindow()
        self.InitRenderer()
        for _ in range(self._frames):  # Window.Run() -> Render event (App.cs:20,39-42)
            self.Renderer.Render(0.0)

    def InitWindow(self):
        self.Window = Window(*self._size)  # App.cs:25-29: 1920 x 1080

    def InitRenderer(self):
        self.Renderer = Renderer(self.Window, self._device)
        self.Renderer.Init()

    def Dispose(self):
        if not self._disposed:
            if self.Renderer is not None:
                self.Renderer.Dispose()
            self._disposed = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.Dispose()
