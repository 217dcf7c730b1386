"""ctypes binding of tests/emu/libcore_emu.so: the trace kernel's per-lane state machine compiled for the host.

Test-only.  It exists so the CPU suite can check the kernel's control flow against the oracle without a GPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from rt_amd.ffi import Camera, TraceParams

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# RT_CHECKER_LIB_DIR: another directory of built checker libraries (the sanitizer builds of `make san`)
_PATH = os.path.join(os.environ.get("RT_CHECKER_LIB_DIR") or os.path.join(_ROOT, "tests", "emu"), "libcore_emu.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-C", _ROOT, "-s", "tests/emu/libcore_emu.so"])
        lib = C.CDLL(_PATH)
        lib.emu_render.restype = C.c_int
        lib.emu_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                   C.POINTER(Camera), C.POINTER(TraceParams), C.c_uint32, C.c_void_p, C.c_void_p]
        lib.emu_render_mode.restype = C.c_int
        lib.emu_render_mode.argtypes = lib.emu_render.argtypes[:10] + [C.c_uint32, C.c_void_p, C.c_void_p]
        lib.emu_forward_ok.restype = C.c_int
        lib.emu_forward_ok.argtypes = [C.c_void_p, C.c_uint32]
        _lib = lib
    return _lib


def forward_ok(scene):
    """The host's decision whether a scene's paths may be accumulated forward."""
    return bool(load().emu_forward_ok(C.cast(scene.materials, C.c_void_p), scene.material_count))


def render(scene, p, tile_size=0, column_masks=False, forward=False, grid=False, share=False, general_lights=False,
           cones=True):
    lib = load()
    rgb = np.zeros((p.height, p.width, 3), dtype=np.float32)
    stats = np.zeros(6, dtype=np.uint64)
    mode = (1 if column_masks else 0) | (2 if forward else 0) | (4 if grid else 0) | (8 if share else 0) | (16 if general_lights else 0) | (0 if cones else 32)
    rc = lib.emu_render_mode(scene.centers_radii.ctypes.data, scene.material_index.ctypes.data, scene.sphere_count,
                             C.cast(scene.materials, C.c_void_p), scene.material_count,
                             C.cast(scene.lights, C.c_void_p), scene.light_count,
                             C.byref(scene.camera), C.byref(p), tile_size, mode, rgb.ctypes.data, stats.ctypes.data)
    assert rc == 0
    lib.emu_last_split_count.restype = C.c_ulonglong
    keys = ["samples", "ray_queries", "hit_queries", "shaded_bounces", "sphere_tests", "exact_tests"]
    out = dict(zip(keys, (int(x) for x in stats)))
    if share:
        out["split_columns"] = int(lib.emu_last_split_count())
    return rgb, out
