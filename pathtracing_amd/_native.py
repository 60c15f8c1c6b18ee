"""ctypes binding of libptrt.so (include/ptrt.h). Plumbing only: every symbol the header declares, nothing else.

There is no fallback: if the HIP library is missing this module raises at import, and without a gfx950 device
`pt_context_create` fails with PT_ERR_NO_DEVICE (the product path never routes through oracle/ or the CPU).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTRT_LIB") or os.path.join(_HERE, "libptrt.so")  # PTRT_LIB: developer aid for A/B-ing kernel builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C pathtracing_amd/csrc` (hipcc --offload-arch=gfx950) "
        "or `python -c 'import __graft_entry__ as g; g.build()'`. libptrt has no CPU fallback."
    )

# One HIP runtime per process. PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1; if
# libptrt.so pulled in /opt/rocm's copy first, a later `import torch` would find two runtimes fighting over the
# device ("No HIP GPUs are available"). Loading torch first makes the dynamic loader satisfy libptrt's
# NEEDED libamdhip64.so.7 with the copy that is already mapped (same SONAME), so torch tensors, RCCL and the
# path tracer share one runtime, one device context and interoperable streams. Hosts without torch (C#, C++)
# simply get the system runtime.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover - torch is optional plumbing
    torch = None

lib = C.CDLL(LIB_PATH)

# status codes / enums (ptrt.h)
PT_OK, PT_ERR_INVALID_ARGUMENT, PT_ERR_NO_DEVICE, PT_ERR_HIP, PT_ERR_OUT_OF_MEMORY, PT_ERR_NOT_COMMITTED, \
    PT_ERR_UNSUPPORTED, PT_ERR_INTERNAL = range(8)
PT_REFERENCE_SPHERE, PT_PATH_TRACE = 0, 1
PT_LAMBERT, PT_METAL, PT_DIELECTRIC = 0, 1, 2
PT_FLAG_PROFILE_KERNELS, PT_FLAG_COUNT_VISITS, PT_FLAG_EXTEND_PACKED, PT_FLAG_EXTEND_SIMPLE, PT_FLAG_ACCUMULATE, PT_FLAG_BUCKET_SPECULAR = 1, 2, 4, 8, 16, 32
PT_FLAG_SPLIT_KERNELS = 64
PT_FLAG_EXTEND_POOL = 128
PT_SCENE_CORNELL, PT_SCENE_CORNELL_GLASS, PT_SCENE_TRIANGLE_SOUP, PT_SCENE_CORNELL_TESS = 0, 1, 2, 3
PT_BVH_WIDTH_2, PT_BVH_WIDTH_4, PT_BVH_WIDTH_4Q, PT_BVH_WIDTH_8Q, PT_BVH_WIDTH_8O, PT_BVH_BUILD_LBVH = 2, 4, 68, 72, 73, 0x100
PT_COMM_FORCE_RCCL, PT_COMM_COPY_EXCHANGE = 1, 2


class pt_device_desc(C.Structure):
    _fields_ = [("device_ordinal", C.c_int32), ("stream", C.c_void_p), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class pt_material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("albedo", C.c_float * 3), ("emission", C.c_float * 3),
                ("roughness", C.c_float), ("ior", C.c_float), ("pad", C.c_uint32 * 3)]


class pt_camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("forward", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("scale", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("jitter", C.c_uint32)]


class pt_render_params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
                ("rr_start", C.c_uint32), ("seed", C.c_uint32), ("sample_offset", C.c_uint32), ("mode", C.c_uint32),
                ("ray_eps", C.c_float), ("rank", C.c_uint32), ("nranks", C.c_uint32), ("tile_size", C.c_uint32),
                ("flags", C.c_uint32), ("streams", C.c_uint32), ("pad", C.c_uint32 * 2)]


class pt_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("iterations", C.c_uint32), ("extend_launches", C.c_uint32),
                ("gpu_ms", C.c_double), ("extend_ms", C.c_double), ("shade_ms", C.c_double), ("other_ms", C.c_double),
                ("reserved", C.c_uint64 * 4)]


class pt_bvh_info(C.Structure):
    _fields_ = [("width", C.c_uint32), ("n_nodes", C.c_uint32), ("n_tris", C.c_uint32), ("max_depth", C.c_uint32),
                ("node_bytes", C.c_uint64), ("tri_bytes", C.c_uint64), ("build_ms", C.c_double),
                ("sah_cost", C.c_float), ("stack_need", C.c_uint32)]


class pt_tuning(C.Structure):
    _fields_ = [("bounces", C.c_uint32), ("loops", C.c_uint32), ("finish_below", C.c_uint32), ("packed_chunk", C.c_uint32),
                ("compact_below", C.c_float), ("sparse_below", C.c_float), ("sticky_samples", C.c_uint32), ("lag", C.c_uint32),
                ("extend_kernel", C.c_uint32), ("readback", C.c_uint32)]


class pt_tile_layout(C.Structure):
    _fields_ = [("tile_size", C.c_uint32), ("tiles_x", C.c_uint32), ("tiles_y", C.c_uint32), ("n_tiles", C.c_uint32),
                ("tiles_mine", C.c_uint32), ("tiles_per_rank", C.c_uint32), ("floats_per_tile", C.c_uint64)]


class pt_scene_counts(C.Structure):
    _fields_ = [("n_tris", C.c_uint64), ("n_spheres", C.c_uint64), ("n_mats", C.c_uint64)]


assert C.sizeof(pt_material) == 48 and C.sizeof(pt_camera) == 64 and C.sizeof(pt_render_params) == 64 and C.sizeof(pt_tuning) == 40

_vp, _u32, _u64, _st = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
_P = C.POINTER

# name -> (restype, argtypes): exactly the entry points of include/ptrt.h
SYMBOLS = {
    "pt_abi_version": (_u32, []),
    "pt_context_create": (_st, [_P(pt_device_desc), _P(_vp)]),
    "pt_context_destroy": (None, [_vp]),
    "pt_last_error": (C.c_char_p, [_vp]),
    "pt_context_get_tuning": (_st, [_vp, _P(pt_tuning)]),
    "pt_context_set_tuning": (_st, [_vp, _P(pt_tuning)]),
    "pt_scene_create": (_st, [_vp, _P(_vp)]),
    "pt_scene_destroy": (None, [_vp]),
    "pt_scene_set_triangles": (_st, [_vp, _vp, _vp, _u64]),
    "pt_scene_set_spheres": (_st, [_vp, _vp, _vp, _u64]),
    "pt_scene_set_materials": (_st, [_vp, _vp, _u64]),
    "pt_scene_set_camera": (_st, [_vp, _P(pt_camera)]),
    "pt_scene_set_sky": (_st, [_vp, _P(C.c_float * 3)]),
    "pt_scene_commit": (_st, [_vp, _u32]),
    "pt_scene_bvh_info": (_st, [_vp, _P(pt_bvh_info)]),
    "pt_scene_bvh_read": (_st, [_vp, _vp, _u64, _vp, _u64]),
    "pt_render": (_st, [_vp, _vp, _P(pt_render_params), _P(pt_stats)]),
    "pt_framebuffer_read": (_st, [_vp, _vp, _u64]),
    "pt_framebuffer_read_rgba8": (_st, [_vp, _vp, _u64]),
    "pt_framebuffer_read_srgb8": (_st, [_vp, _vp, _u64]),
    "pt_framebuffer_device_ptr": (_st, [_vp, _P(_vp), _P(_u64)]),
    "pt_tile_layout_query": (_st, [_P(pt_render_params), _P(pt_tile_layout)]),
    "pt_tiles_device_ptr": (_st, [_vp, _P(_vp), _P(_u64)]),
    "pt_assemble_tiles": (_st, [_vp, _P(pt_render_params), _vp, _u64]),
    "pt_comm_create": (_st, [_P(_vp), _u32, _u32, _u32, _P(_vp)]),
    "pt_comm_destroy": (None, [_vp]),
    "pt_comm_render": (_st, [_vp, _P(_vp), _P(pt_render_params), _P(pt_stats)]),
    "pt_comm_stage_tiles": (_st, [_vp, _u32]),
    "pt_comm_assemble": (_st, [_vp, _P(pt_render_params)]),
    "pt_scenegen": (_st, [_u32, _u32, _u32, _u32, _u32, _P(pt_scene_counts), _vp, _vp, _vp, _vp, _vp, _P(pt_camera), _vp]),
}

for _name, (_res, _args) in SYMBOLS.items():
    _f = getattr(lib, _name)  # AttributeError here = the library does not export what the header declares
    _f.restype = _res
    _f.argtypes = _args

PTRT_ABI_VERSION = 2  # include/ptrt.h this binding was written against
if lib.pt_abi_version() != PTRT_ABI_VERSION:
    raise ImportError(f"{LIB_PATH} has ABI version {lib.pt_abi_version()}, this binding expects {PTRT_ABI_VERSION}: rebuild libptrt.so")
