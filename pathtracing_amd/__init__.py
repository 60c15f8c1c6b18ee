"""pathtracing_amd — MI355X-native wavefront path tracer behind the compute seam of chairclr/PathTracing.

The product is `libptrt.so` (HIP, gfx950; C ABI in include/ptrt.h). This package is its thin host mirror
(App / Renderer, same shape as the reference's C# host). Importing it loads the HIP library and fails loudly
if it is missing; nothing here computes pixels on the CPU.
"""
from . import _native as native  # noqa: F401  (raises ImportError if libptrt.so is not built)
from .host import (App, Comm, Renderer, Window, SceneData, PtException, make_scene, make_params, tile_layout,  # noqa: F401
                   MATERIAL_DTYPE)

__all__ = ["App", "Comm", "Renderer", "Window", "SceneData", "PtException", "make_scene", "make_params", "tile_layout",
           "MATERIAL_DTYPE", "native"]
