"""ctypes binding of oracle/librt_oracle.so -- the CPU restatement of the reference's trace loop.

TEST INFRASTRUCTURE: imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg only.
The struct layouts are those of oracle/rt_oracle.h, which match include/rt_hip.h field for field, so the
ctypes classes of rt_amd.ffi are reused for the arguments.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from rt_amd.ffi import Camera, Light, Material, Stats, TraceParams  # layouts only

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_CHECKER_LIB_DIR: another directory of built checker libraries (the sanitizer builds of `make san`)
_LIB_DIR = os.environ.get("RT_CHECKER_LIB_DIR") or _HERE
_PATH = os.path.join(_LIB_DIR, "librt_oracle.so")


class OrcHit(C.Structure):
    _fields_ = [("t", C.c_float), ("position", C.c_float * 3), ("normal", C.c_float * 3), ("sphere", C.c_uint32)]


class OrcScene(C.Structure):
    _fields_ = [
        ("spheres", C.c_void_p),
        ("material_index", C.c_void_p),
        ("sphere_count", C.c_uint32),
        ("materials", C.c_void_p),
        ("material_count", C.c_uint32),
        ("lights", C.c_void_p),
        ("light_count", C.c_uint32),
    ]


class OrcStats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64),
        ("ray_queries", C.c_uint64),
        ("hit_queries", C.c_uint64),
        ("shaded_bounces", C.c_uint64),
        ("sphere_tests", C.c_uint64),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


_lib = None
F3 = C.c_float * 3


def build():
    """Compiles the oracle in place (gcc, a second or two)."""
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        lib = C.CDLL(_PATH)
        lib.orc_render.restype = C.c_int
        lib.orc_render.argtypes = [C.POINTER(OrcScene), C.POINTER(Camera), C.POINTER(TraceParams), C.c_uint32,
                                   C.c_void_p, C.POINTER(OrcStats)]
        lib.orc_render_rows.restype = C.c_int
        lib.orc_render_rows.argtypes = [C.POINTER(OrcScene), C.POINTER(Camera), C.POINTER(TraceParams), C.c_uint32,
                                        C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(OrcStats)]
        lib.orc_philox4x32_10.restype = None
        lib.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32 * 4), C.POINTER(C.c_uint32 * 2), C.POINTER(C.c_uint32 * 4)]
        lib.orc_uniform.restype = C.c_float
        lib.orc_uniform.argtypes = [C.c_uint32]
        lib.orc_ball.restype = None
        lib.orc_ball.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(F3)]
        lib.orc_primary_direction.restype = None
        lib.orc_primary_direction.argtypes = [C.POINTER(Camera)] + [C.c_uint32] * 7 + [C.POINTER(F3)]
        lib.orc_sphere_trace_nearest.restype = C.c_int
        lib.orc_sphere_trace_nearest.argtypes = [C.POINTER(F3), C.POINTER(F3), C.POINTER(F3), C.c_float, C.c_float,
                                                 C.c_float, C.POINTER(C.c_float), C.POINTER(F3), C.POINTER(F3)]
        lib.orc_sphere_trace.restype = C.c_int
        lib.orc_sphere_trace.argtypes = [C.POINTER(F3), C.POINTER(F3), C.POINTER(F3), C.c_float, C.c_float, C.c_float,
                                         C.POINTER(OrcHit)]
        lib.orc_trace_all.restype = C.c_uint32
        lib.orc_trace_all.argtypes = [C.POINTER(OrcScene), C.POINTER(F3), C.POINTER(F3), C.c_float, C.c_float,
                                      C.POINTER(OrcHit), C.c_uint32]
        lib.orc_trace_nearest.restype = C.c_int
        lib.orc_trace_nearest.argtypes = [C.POINTER(OrcScene), C.POINTER(F3), C.POINTER(F3), C.c_float, C.c_float,
                                          C.POINTER(C.c_float), C.POINTER(F3), C.POINTER(F3)]
        lib.orc_scatter.restype = None
        lib.orc_scatter.argtypes = [C.POINTER(Material), C.POINTER(F3), C.POINTER(F3), C.POINTER(F3), C.c_float,
                                    C.POINTER(F3), C.POINTER(F3), C.POINTER(C.c_float)]
        lib.orc_trace_light.restype = None
        lib.orc_trace_light.argtypes = [C.POINTER(OrcScene), C.POINTER(Light), C.POINTER(Material), C.POINTER(F3),
                                        C.POINTER(F3), C.c_void_p, C.POINTER(F3), C.POINTER(OrcStats)]
        lib.orc_get_specular.restype = None
        lib.orc_get_specular.argtypes = [C.POINTER(Light), C.POINTER(Material), C.POINTER(F3), C.POINTER(F3),
                                         C.POINTER(F3)]
        lib.orc_accumulate.restype = None
        lib.orc_accumulate.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(F3)]
        lib.orc_ray_trace.restype = None
        lib.orc_ray_trace.argtypes = [C.POINTER(OrcScene), C.POINTER(Camera), C.POINTER(TraceParams), C.c_uint32,
                                      C.c_uint32, C.POINTER(F3), C.POINTER(F3), C.POINTER(OrcStats)]
        _lib = lib
    return _lib


def f3(v):
    return F3(float(v[0]), float(v[1]), float(v[2]))


def orc_scene(scene):
    """OrcScene view of an rt_amd.ffi.Scene (keeps no ownership: the Scene must outlive it)."""
    s = OrcScene()
    s.spheres = scene.centers_radii.ctypes.data
    s.material_index = scene.material_index.ctypes.data
    s.sphere_count = scene.sphere_count
    s.materials = C.cast(scene.materials, C.c_void_p)
    s.material_count = scene.material_count
    s.lights = C.cast(scene.lights, C.c_void_p)
    s.light_count = scene.light_count
    return s


_O3_PATH = os.path.join(_LIB_DIR, "librt_oracle_o3.so")
_o3 = None


def load_o3():
    """The same source built -O3 -march=x86-64-v3 (no FMA contraction either): only for timing the CPU baseline
    with the optimiser let loose; its frames equal the -O2 build's.  OSError if it was not built."""
    global _o3
    if _o3 is None:
        lib = C.CDLL(_O3_PATH)
        lib.orc_render.restype = C.c_int
        lib.orc_render.argtypes = [C.POINTER(OrcScene), C.POINTER(Camera), C.POINTER(TraceParams), C.c_uint32,
                                   C.c_void_p, C.POINTER(OrcStats)]
        _o3 = lib
    return _o3


def render(scene, p, threads=None, build=None):
    """Whole frame on the CPU; returns (rgb[h,w,3] float32, OrcStats).  build="o3": the -O3 build (timing only)."""
    lib = load_o3() if build == "o3" else load()
    if threads is None:
        threads = min(os.cpu_count() or 1, 16)
    rgb = np.zeros((p.height, p.width, 3), dtype=np.float32)
    stats = OrcStats()
    s = orc_scene(scene)
    rc = lib.orc_render(C.byref(s), C.byref(scene.camera), C.byref(p), threads, rgb.ctypes.data, C.byref(stats))
    if rc != 0:
        raise RuntimeError("orc_render failed with %d" % rc)
    return rgb, stats


def render_rows(scene, p, row_begin, row_end, threads=None):
    """Rows [row_begin, row_end) of the frame on the CPU; returns (rgb[row_end - row_begin, w, 3] float32, OrcStats)."""
    lib = load()
    if threads is None:
        threads = min(os.cpu_count() or 1, 16)
    rgb = np.zeros((p.height, p.width, 3), dtype=np.float32)
    stats = OrcStats()
    s = orc_scene(scene)
    rc = lib.orc_render_rows(C.byref(s), C.byref(scene.camera), C.byref(p), threads, row_begin, row_end,
                             rgb.ctypes.data, C.byref(stats))
    if rc != 0:
        raise RuntimeError("orc_render_rows failed with %d" % rc)
    return rgb[row_begin:row_end].copy(), stats


_FRAME_PATH = os.path.join(_LIB_DIR, "liboracle_frame.so")
_frame_lib = None


def load_frame():
    """liboracle_frame.so: the oracle behind the Frame plug-in interface (oracle/frame_cpu.cpp)."""
    global _frame_lib
    if _frame_lib is None:
        load()
        if not os.path.exists(_FRAME_PATH):
            build()
        lib = C.CDLL(_FRAME_PATH)
        lib.orc_frame_render_demo.restype = C.c_int
        lib.orc_frame_render_demo.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                              C.POINTER(OrcStats)]
        lib.orc_frame_render_scene_file.restype = C.c_int
        lib.orc_frame_render_scene_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(OrcStats)]
        _frame_lib = lib
    return _frame_lib


def render_demo_through_frame(index, p, scene_seed=1, spheres=10000, threads=None):
    """A demo built around the CPU checker's Frame (DemoFrameCPU), driven as the reference's App drives one."""
    lib = load_frame()
    if threads is None:
        threads = min(os.cpu_count() or 1, 16)
    rgb = np.zeros((p.height, p.width, 3), dtype=np.float32)
    stats = OrcStats()
    rc = lib.orc_frame_render_demo(index, scene_seed, spheres, C.byref(p), threads, rgb.ctypes.data, C.byref(stats))
    if rc != 0:
        raise RuntimeError("orc_frame_render_demo failed with %d" % rc)
    return rgb, stats


def render_scene_file_through_frame(path, p, threads=None):
    lib = load_frame()
    if threads is None:
        threads = min(os.cpu_count() or 1, 16)
    rgb = np.zeros((p.height, p.width, 3), dtype=np.float32)
    stats = OrcStats()
    rc = lib.orc_frame_render_scene_file(os.fsencode(path), C.byref(p), threads, rgb.ctypes.data, C.byref(stats))
    if rc != 0:
        raise RuntimeError("orc_frame_render_scene_file failed with %d" % rc)
    return rgb, stats
