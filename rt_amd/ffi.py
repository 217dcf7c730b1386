"""ctypes bindings of librt_hip / librt_host (declarations: include/rt_hip.h, include/rt_host.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB_DIR: another directory of built libraries (experiment builds of the kernels); default rt_amd/lib
_LIB_DIR = os.environ.get("RT_AMD_LIB_DIR") or os.path.join(_HERE, "lib")

FLAG_DETERMINISTIC = 1
FLAG_GRID = 2
FLAG_PIPELINED = 4
FLAG_LEAVE_ROOM = 8
FLAG_PIXEL_RUNS = 16
FLAG_KERNEL_SUMS = 32

RT_HIP_OK = 0
RT_HIP_ERR_NOT_READY = 3
RT_HIP_ERR_CANCELLED = 4


class Material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("albedo", C.c_float * 3), ("roughness", C.c_float), ("refraction", C.c_float)]


class Light(C.Structure):
    _fields_ = [("type", C.c_uint32), ("direction", C.c_float * 3), ("color", C.c_float * 3)]


class Camera(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("direction", C.c_float * 3),
        ("fov", C.c_float),
        ("znear", C.c_float),
        ("zfar", C.c_float),
        ("background", C.c_float * 3),
    ]


class TraceParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("samples_in_row", C.c_uint32),
        ("flags", C.c_uint32),
        ("seed", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("slice_index", C.c_uint32),
        ("slice_count", C.c_uint32),
    ]


class HostHit(C.Structure):
    _fields_ = [("t", C.c_float), ("position", C.c_float * 3), ("normal", C.c_float * 3), ("sphere", C.c_uint32)]


class LaunchInfo(C.Structure):
    """rt_hip_launch_info: how the library would trace a frame (rt_hip_describe_launch)."""

    _fields_ = [(name, C.c_uint32) for name in (
        "blocks", "lds_bytes", "batch_items", "batch_min", "tiled", "grid", "forward", "one_light", "library_math",
        "column_masks", "sample_sharing", "kernel_sums", "wave_sums")] + [("reserved", C.c_uint32 * 3)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "reserved"}


class Stats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64),
        ("ray_queries", C.c_uint64),
        ("hit_queries", C.c_uint64),
        ("shaded_bounces", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("exact_tests", C.c_uint64),
        ("lane_trips", C.c_uint64),
        ("lane_rounds", C.c_uint64),
        ("cone_tests", C.c_uint64),
        ("kernel_ms", C.c_float),
        ("reserved", C.c_uint32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "reserved"}


def _load(name):
    path = os.path.join(_LIB_DIR, name)
    if not os.path.exists(path):
        raise ImportError(
            "%s is not built (%s missing): run `make` at the repository root; there is no fallback renderer" % (name, path)
        )
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_hip = None
_host = None

# Every entry point rt_hip.h declares: name -> (restype, argtypes)
HIP_SYMBOLS = {
    "rt_hip_init": (C.c_int, [C.c_int]),
    "rt_hip_shutdown": (None, []),
    "rt_hip_last_error": (C.c_char_p, []),
    "rt_hip_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "rt_hip_scene_create": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)],
    ),
    "rt_hip_scene_destroy": (None, [C.c_void_p]),
    "rt_hip_trace_async": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(TraceParams), C.POINTER(C.c_void_p)]),
    "rt_hip_job_poll": (C.c_int, [C.c_void_p]),
    "rt_hip_job_snapshot": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_hip_job_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "rt_hip_job_cancel": (C.c_int, [C.c_void_p]),
    "rt_hip_trace_device": (
        C.c_int,
        [C.c_void_p, C.POINTER(Camera), C.POINTER(TraceParams), C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "rt_hip_trace_device_static": (
        C.c_int,
        [C.c_void_p, C.POINTER(Camera), C.POINTER(TraceParams), C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "rt_hip_resolve_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "rt_hip_time_trace_kernel": (
        C.c_int,
        [C.c_void_p, C.POINTER(Camera), C.POINTER(TraceParams), C.c_uint32, C.POINTER(C.c_float), C.POINTER(Stats)],
    ),
    "rt_hip_check_exact_math": (C.c_int, [C.POINTER(C.c_uint64)]),
    "rt_hip_check_philox": (C.c_int, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_hip_describe_launch": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(TraceParams), C.POINTER(LaunchInfo)]),
}

HOST_SYMBOLS = {
    "rt_host_scene_build": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "rt_host_scene_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "rt_host_scene_save": (
        C.c_int,
        [C.c_char_p, C.POINTER(Camera), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32],
    ),
    "rt_host_scene_counts": (None, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_host_scene_copy": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Camera)]),
    "rt_host_scene_free": (None, [C.c_void_p]),
    "rt_host_render_demo": (
        C.c_int,
        [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(TraceParams), C.c_void_p, C.POINTER(Stats), C.c_char_p, C.c_size_t],
    ),
    "rt_host_trace_hits": (
        C.c_uint32,
        [C.c_void_p, C.c_uint32, C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), C.c_float, C.c_float, C.c_uint32,
         C.POINTER(HostHit), C.c_uint32],
    ),
    "rt_host_update_texture": (None, [C.c_void_p, C.c_size_t, C.c_void_p]),
}


def _bind(lib, table):
    for name, (restype, argtypes) in table.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def load_hip():
    global _hip
    if _hip is None:
        _hip = _bind(_load("librt_hip.so"), HIP_SYMBOLS)
    return _hip


def load_host():
    global _host
    if _host is None:
        load_hip()
        _host = _bind(_load("librt_host.so"), HOST_SYMBOLS)
    return _host


class Scene:
    """A flattened scene: the arrays rt_hip_scene_create takes, plus the camera."""

    def __init__(self, centers_radii, material_index, materials, lights, camera):
        self.centers_radii = np.ascontiguousarray(centers_radii, dtype=np.float32).reshape(-1, 4)
        self.material_index = np.ascontiguousarray(material_index, dtype=np.uint32)
        self.materials = materials  # ctypes array of Material
        self.lights = lights  # ctypes array of Light
        self.camera = camera

    @property
    def sphere_count(self):
        return int(self.material_index.shape[0])

    @classmethod
    def _from_handle(cls, host, handle):
        try:
            n, m, l = C.c_uint32(), C.c_uint32(), C.c_uint32()
            host.rt_host_scene_counts(handle, C.byref(n), C.byref(m), C.byref(l))
            cr = np.zeros((n.value, 4), dtype=np.float32)
            mi = np.zeros((n.value,), dtype=np.uint32)
            mats = (Material * max(m.value, 1))()
            lights = (Light * max(l.value, 1))()
            cam = Camera()
            host.rt_host_scene_copy(handle, cr.ctypes.data, mi.ctypes.data, mats, lights, C.byref(cam))
        finally:
            host.rt_host_scene_free(handle)
        scene = cls(cr, mi, mats, lights, cam)
        scene.material_count = m.value
        scene.light_count = l.value
        return scene

    @classmethod
    def demo(cls, index, seed=1, spheres=10000):
        """Demo `index` (1 pretty, 2 glass, 3 synthetic) as the C++ host builds and flattens it."""
        host = load_host()
        handle = C.c_void_p()
        rc = host.rt_host_scene_build(index, seed, spheres, C.byref(handle))
        if rc != 0:
            raise RuntimeError("rt_host_scene_build(%d) failed with %d" % (index, rc))
        return cls._from_handle(host, handle)

    @classmethod
    def load(cls, path):
        """A scene file (rt_amd/host/demos/scene_file.hpp), parsed and flattened by the C++ host."""
        host = load_host()
        handle = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = host.rt_host_scene_load(os.fsencode(path), C.byref(handle), err, 512)
        if rc != 0:
            raise ValueError("scene file: %s" % err.value.decode())
        return cls._from_handle(host, handle)

    def save(self, path):
        host = load_host()
        rc = host.rt_host_scene_save(os.fsencode(path), C.byref(self.camera), self.centers_radii.ctypes.data,
                                     self.material_index.ctypes.data, self.sphere_count,
                                     C.cast(self.materials, C.c_void_p), self.material_count,
                                     C.cast(self.lights, C.c_void_p), self.light_count)
        if rc != 0:
            raise OSError("cannot write scene file %s" % path)

    @classmethod
    def build(cls, spheres, materials, lights, camera):
        """spheres: [(cx, cy, cz, r, material_index)], materials: [(type, (r,g,b), roughness, refraction)],
        lights: [((dx,dy,dz), (r,g,b))], camera: dict of Camera fields."""
        cr = np.array([s[:4] for s in spheres], dtype=np.float32).reshape(-1, 4)
        mi = np.array([s[4] for s in spheres], dtype=np.uint32)
        mats = (Material * max(len(materials), 1))()
        for i, (t, albedo, rough, refr) in enumerate(materials):
            mats[i].type = t
            mats[i].albedo[:] = albedo
            mats[i].roughness = rough
            mats[i].refraction = refr
        ls = (Light * max(len(lights), 1))()
        for i, (d, c) in enumerate(lights):
            ls[i].type = 0
            ls[i].direction[:] = d
            ls[i].color[:] = c
        cam = Camera()
        cam.position[:] = camera["position"]
        d = np.asarray(camera["direction"], dtype=np.float32)
        # Frame::SetCameraDirection stores the direction normalised: v * (1 / sqrt(dot(v, v))) in float.
        inv = np.float32(1.0) / np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), dtype=np.float32)
        cam.direction[:] = [float(np.float32(x) * inv) for x in d]
        cam.fov = camera.get("fov", float(np.float32(2.0) * np.float32(np.float32(3.14159265358979323846)) / np.float32(3.0)))
        cam.znear = camera.get("znear", 0.001)
        cam.zfar = camera.get("zfar", 1000.0)
        cam.background[:] = camera.get("background", (1.0, 1.0, 1.0))
        scene = cls(cr, mi, mats, ls, cam)
        scene.material_count = len(materials)
        scene.light_count = len(lights)
        return scene


def params(width, height, samples_in_row, flags=0, seed=1, max_depth=0, slice_index=0, slice_count=1):
    p = TraceParams()
    p.width, p.height, p.samples_in_row = width, height, samples_in_row
    p.flags, p.seed, p.max_depth = flags, seed, max_depth
    p.slice_index, p.slice_count = slice_index, slice_count
    return p


class HipError(RuntimeError):
    pass


class HipRenderer:
    """Thin owner of the rt_hip handles.  Every call goes through the C ABI."""

    def __init__(self, device=-1):
        self.lib = load_hip()
        self._check(self.lib.rt_hip_init(device), "rt_hip_init")
        self._scenes = {}

    def _check(self, rc, what):
        if rc != RT_HIP_OK:
            raise HipError("%s failed (%d): %s" % (what, rc, self.lib.rt_hip_last_error().decode()))

    def device_info(self):
        cus, khz = C.c_int(), C.c_int()
        name = C.create_string_buffer(256)
        self._check(self.lib.rt_hip_device_info(C.byref(cus), C.byref(khz), name, 256), "rt_hip_device_info")
        return {"compute_units": cus.value, "clock_khz": khz.value, "name": name.value.decode()}

    def upload(self, scene):
        key = id(scene)
        if key not in self._scenes:
            handle = C.c_void_p()
            self._check(
                self.lib.rt_hip_scene_create(
                    scene.centers_radii.ctypes.data,
                    scene.material_index.ctypes.data,
                    scene.sphere_count,
                    C.cast(scene.materials, C.c_void_p),
                    scene.material_count,
                    C.cast(scene.lights, C.c_void_p),
                    scene.light_count,
                    C.byref(handle),
                ),
                "rt_hip_scene_create",
            )
            self._scenes[key] = (handle, scene)
        return self._scenes[key][0]

    def start(self, scene, p):
        job = C.c_void_p()
        self._check(self.lib.rt_hip_trace_async(self.upload(scene), C.byref(scene.camera), C.byref(p), C.byref(job)),
                    "rt_hip_trace_async")
        return job

    def poll(self, job):
        return self.lib.rt_hip_job_poll(job)

    def snapshot(self, job, p):
        rgb = np.empty((p.height, p.width, 3), dtype=np.float32)
        self._check(self.lib.rt_hip_job_snapshot(job, rgb.ctypes.data), "rt_hip_job_snapshot")
        return rgb

    def cancel(self, job):
        self._check(self.lib.rt_hip_job_cancel(job), "rt_hip_job_cancel")

    def finish(self, job, p, allow_cancelled=False):
        rgb = np.empty((p.height, p.width, 3), dtype=np.float32)
        stats = Stats()
        rc = self.lib.rt_hip_job_finish(job, rgb.ctypes.data, C.byref(stats))
        if not (allow_cancelled and rc == RT_HIP_ERR_CANCELLED):
            self._check(rc, "rt_hip_job_finish")
        return rgb, stats

    def render(self, scene, p):
        """One whole frame through the asynchronous job API; returns (rgb[h,w,3] float32, Stats)."""
        return self.finish(self.start(scene, p), p)

    def trace_device(self, scene, p, d_planes_ptr, d_stats_ptr=None, stream_ptr=None):
        self._check(
            self.lib.rt_hip_trace_device(self.upload(scene), C.byref(scene.camera), C.byref(p), d_planes_ptr,
                                         d_stats_ptr, stream_ptr),
            "rt_hip_trace_device",
        )

    def trace_device_static(self, scene, p, d_planes_ptr, d_stats_ptr=None, stream_ptr=None):
        """rt_hip_trace_device over the library's static scratch array: allocates nothing, bounded frame size."""
        self._check(
            self.lib.rt_hip_trace_device_static(self.upload(scene), C.byref(scene.camera), C.byref(p), d_planes_ptr,
                                                d_stats_ptr, stream_ptr),
            "rt_hip_trace_device_static",
        )

    def resolve_device(self, d_planes_ptr, d_rgb_ptr, p, stream_ptr=None):
        self._check(
            self.lib.rt_hip_resolve_device(d_planes_ptr, d_rgb_ptr, p.width, p.height, p.samples_in_row, stream_ptr),
            "rt_hip_resolve_device",
        )

    def time_trace_kernel(self, scene, p, launches):
        ms = C.c_float()
        stats = Stats()
        self._check(
            self.lib.rt_hip_time_trace_kernel(self.upload(scene), C.byref(scene.camera), C.byref(p), launches,
                                              C.byref(ms), C.byref(stats)),
            "rt_hip_time_trace_kernel",
        )
        return ms.value, stats

    def describe(self, scene, p):
        """How the library would trace this frame: a LaunchInfo (nothing is traced)."""
        info = LaunchInfo()
        self._check(self.lib.rt_hip_describe_launch(self.upload(scene), C.byref(scene.camera), C.byref(p), C.byref(info)),
                    "rt_hip_describe_launch")
        return info

    def check_exact_math(self):
        """Mismatch counts of the device self-check of the short correctly rounded sqrt / quotient sequences."""
        out = (C.c_uint64 * 4)()
        self._check(self.lib.rt_hip_check_exact_math(out), "rt_hip_check_exact_math")
        return [int(v) for v in out]

    def check_philox(self):
        """The device's Philox4x32-10 on the Random123 known-answer inputs (three blocks of four words) and the fold
        of a long run of blocks as the device and as the host computed it."""
        words = (C.c_uint32 * 12)()
        dev, host = C.c_uint32(), C.c_uint32()
        self._check(self.lib.rt_hip_check_philox(words, C.byref(dev), C.byref(host)), "rt_hip_check_philox")
        return [int(w) for w in words], int(dev.value), int(host.value)

    def close(self):
        for handle, _ in self._scenes.values():
            self.lib.rt_hip_scene_destroy(handle)
        self._scenes.clear()


def render_demo(index, p, scene_seed=1, spheres=10000):
    """A demo through the C++ Frame API (DemoFrameHIP), as rt_cli drives it."""
    host = load_host()
    rgb = np.empty((p.height, p.width, 3), dtype=np.float32)
    stats = Stats()
    err = C.create_string_buffer(512)
    rc = host.rt_host_render_demo(index, scene_seed, spheres, C.byref(p), rgb.ctypes.data, C.byref(stats), err, 512)
    if rc != 0:
        raise HipError("rt_host_render_demo failed (%d): %s" % (rc, err.value.decode()))
    return rgb, stats


def trace_hits(scene, origin, direction, tmin, tmax, nearest_only=False, capacity=64):
    """The object model's ray queries on the host (TraceInHitableCollection / TraceNearestInHitableCollection):
    returns (number of hits found, [HostHit, ...] the first `capacity` of them)."""
    host = load_host()
    out = (HostHit * capacity)()
    o = (C.c_float * 3)(*[float(x) for x in origin])
    d = (C.c_float * 3)(*[float(x) for x in direction])
    n = host.rt_host_trace_hits(scene.centers_radii.ctypes.data, scene.sphere_count, C.byref(o), C.byref(d), tmin, tmax,
                                1 if nearest_only else 0, out, capacity)
    return n, [out[i] for i in range(min(n, capacity))]


def update_texture(rgb):
    """App::UpdateTexture's float -> RGBA8 conversion, as the C++ host implements it."""
    host = load_host()
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    pixels = rgb.size // 3
    out = np.empty((pixels, 4), dtype=np.uint8)
    host.rt_host_update_texture(rgb.ctypes.data, pixels, out.ctypes.data)
    return out.reshape(rgb.shape[:-1] + (4,))
