"""ctypes binding of oracle/libpt_oracle.so — TEST INFRASTRUCTURE ONLY.

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the checker.
Never import this from pathtracing_amd/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not found: run `make -C oracle`")
lib = C.CDLL(LIB_PATH)

MATERIAL_DTYPE = np.dtype([("kind", "<u4"), ("albedo", "<f4", 3), ("emission", "<f4", 3), ("roughness", "<f4"),
                           ("ior", "<f4"), ("pad", "<u4", 3)])


class pto_camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("forward", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("scale", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("jitter", C.c_uint32)]


class pto_params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
                ("rr_start", C.c_uint32), ("seed", C.c_uint32), ("sample_offset", C.c_uint32), ("mode", C.c_uint32),
                ("ray_eps", C.c_float), ("rank", C.c_uint32), ("nranks", C.c_uint32), ("tile_size", C.c_uint32),
                ("flags", C.c_uint32), ("streams", C.c_uint32), ("pad", C.c_uint32 * 2)]


class pto_scene(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("tri_verts", C.c_void_p), ("tri_mat", C.c_void_p),
                ("n_spheres", C.c_uint32), ("spheres", C.c_void_p), ("sph_mat", C.c_void_p),
                ("n_mats", C.c_uint32), ("mats", C.c_void_p),
                ("sky", C.c_float * 3), ("cam", pto_camera),
                ("bvh_width", C.c_uint32), ("n_nodes", C.c_uint32), ("nodes", C.c_void_p), ("tris48", C.c_void_p)]


class pto_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("primary_misses", C.c_uint64)]


lib.pto_reference_sphere.restype = C.c_int
lib.pto_reference_sphere.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
lib.pto_unorm8.restype = C.c_uint8
lib.pto_unorm8.argtypes = [C.c_float]
lib.pto_render.restype = C.c_int
lib.pto_render.argtypes = [C.POINTER(pto_scene), C.POINTER(pto_params), C.c_int, C.c_void_p, C.POINTER(pto_stats)]
lib.pto_bvh_build.restype = C.c_int
lib.pto_bvh_build.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
lib.pto_free.argtypes = [C.c_void_p]
lib.pto_bvh_validate.restype = C.c_int
lib.pto_bvh_validate.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
lib.pto_pcg.restype = C.c_uint32
lib.pto_pcg.argtypes = [C.c_uint32]
lib.pto_path_key.restype = C.c_uint32
lib.pto_path_key.argtypes = [C.c_uint32] * 3
lib.pto_u01.restype = C.c_float
lib.pto_u01.argtypes = [C.c_uint32, C.c_uint32]
lib.pto_sincos2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
lib.pto_closest.restype = C.c_uint32
lib.pto_closest.argtypes = [C.POINTER(pto_scene), C.c_float * 3, C.c_float * 3, C.POINTER(C.c_float), C.POINTER(pto_stats)]
lib.pto_bsdf_sample.restype = C.c_int
lib.pto_bsdf_sample.argtypes = [C.c_void_p, C.c_float * 3, C.c_float * 3, C.c_int, C.c_float, C.c_float, C.c_float,
                                C.c_float * 3, C.c_float * 3, C.POINTER(C.c_float)]
lib.pto_camera_ray.argtypes = [C.POINTER(pto_camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_float * 3, C.c_float * 3]
lib.pto_num_threads.restype = C.c_int


def _p(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)


def reference_sphere(w, h):
    f = np.zeros((h, w, 4), np.float32)
    b = np.zeros((h, w, 4), np.uint8)
    assert lib.pto_reference_sphere(w, h, _p(f), _p(b)) == 0
    return f, b


class Scene:
    """Keeps numpy arrays alive next to the C struct. `sd` is any object with verts/tri_mat/spheres/sph_mat/mats/cam/sky."""

    def __init__(self, sd, bvh=None):
        self.keep = dict(
            verts=np.ascontiguousarray(sd.verts, np.float32), tri_mat=np.ascontiguousarray(sd.tri_mat, np.uint32),
            spheres=np.ascontiguousarray(sd.spheres, np.float32), sph_mat=np.ascontiguousarray(sd.sph_mat, np.uint32),
            mats=np.ascontiguousarray(sd.mats))
        k = self.keep
        s = pto_scene()
        s.n_tris, s.tri_verts, s.tri_mat = len(k["tri_mat"]), _p(k["verts"]), _p(k["tri_mat"])
        s.n_spheres, s.spheres, s.sph_mat = len(k["sph_mat"]), _p(k["spheres"]), _p(k["sph_mat"])
        s.n_mats, s.mats = len(k["mats"]), _p(k["mats"])
        for i in range(3):
            s.sky[i] = float(sd.sky[i])
        C.memmove(C.byref(s.cam), C.byref(sd.cam), C.sizeof(pto_camera))
        self.c = s
        self.own = None
        if bvh is not None:
            self.set_bvh(*bvh)

    def set_bvh(self, width, nodes_bytes, tris_bytes):
        self.keep["nodes"] = np.ascontiguousarray(nodes_bytes, np.uint8)
        self.keep["tris48"] = np.ascontiguousarray(tris_bytes, np.uint8)
        self.c.bvh_width = width
        self.c.n_nodes = self.keep["nodes"].size // (64 if width == 68 else 128 if width in (72, 73) else 32 * width)
        self.c.nodes, self.c.tris48 = _p(self.keep["nodes"]), _p(self.keep["tris48"])

    def build_own_bvh(self):
        n, nodes, tris = C.c_uint32(), C.c_void_p(), C.c_void_p()
        rc = lib.pto_bvh_build(self.c.n_tris, self.c.tri_verts, self.c.tri_mat, C.byref(n), C.byref(nodes), C.byref(tris))
        assert rc == 0, rc
        nb = np.ctypeslib.as_array(C.cast(nodes, C.POINTER(C.c_uint8)), (n.value * 64,)).copy()
        tb = np.ctypeslib.as_array(C.cast(tris, C.POINTER(C.c_uint8)), (self.c.n_tris * 48,)).copy()
        lib.pto_free(nodes)
        lib.pto_free(tris)
        self.set_bvh(2, nb, tb)
        return nb, tb

    def closest(self, o, d):
        """(primitive id or 0xFFFFFFFF, t) of the closest hit of one ray (SPEC §4)."""
        t = C.c_float()
        pid = lib.pto_closest(C.byref(self.c), (C.c_float * 3)(*[float(x) for x in o]), (C.c_float * 3)(*[float(x) for x in d]),
                              C.byref(t), None)
        return int(pid), float(t.value)

    def validate_bvh(self):
        d = C.c_uint32()
        rc = lib.pto_bvh_validate(self.c.bvh_width, self.c.n_nodes, self.c.nodes, self.c.tris48, self.c.n_tris,
                                  self.c.tri_verts, self.c.tri_mat, C.byref(d))
        return rc, d.value


def render(scene, params, threads=0):
    """params: any ctypes struct with the pt_render_params layout. Returns (rgba float32 HxWx4, pto_stats)."""
    p = pto_params()
    C.memmove(C.byref(p), C.byref(params), C.sizeof(pto_params))
    out = np.zeros((p.height, p.width, 4), np.float32)
    st = pto_stats()
    rc = lib.pto_render(C.byref(scene.c), C.byref(p), threads, _p(out), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"pto_render failed: {rc}")
    return out, st
