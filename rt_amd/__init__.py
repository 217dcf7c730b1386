"""rt_amd -- MI355X (gfx950) ray-tracing backend for rokuz/rt.

The product is two native libraries built in-tree by the top-level Makefile:

* ``rt_amd/lib/librt_hip.so``  -- the HIP kernels behind the C ABI of ``include/rt_hip.h``;
* ``rt_amd/lib/librt_host.so`` -- the C++ host: the reference's Frame plug-in API, ``DemoFrameHIP`` and the
  demo scenes (``include/rt_host.h``).

This package is only the ctypes plumbing that tests and the benchmark use to reach them.  There is no
Python or CPU implementation of the renderer here: if a library is missing, importing its binding raises.
"""
from .ffi import (  # noqa: F401
    Camera,
    HipRenderer,
    Light,
    Material,
    Scene,
    Stats,
    TraceParams,
    FLAG_DETERMINISTIC,
    load_hip,
    load_host,
    update_texture,
)
