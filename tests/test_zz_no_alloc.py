"""rt_hip_trace_device_static -- the entry point that allocates nothing -- against rt_hip_trace_device.

Both take the same arguments and must write the same sum planes and counters, bit for bit; the static variant
keeps its scratch in a file-scope device array of the library, so its frame size is bounded and consecutive
calls reuse (and must not be confused by) what the previous call left there.
"""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def test_static_scratch_entry_point_matches_the_allocating_one(hip, ffi, scenes):
    import torch

    stream = torch.cuda.current_stream().cuda_stream
    # calls in a row, each with a smaller frame than the one before (and another scene, seed and a slice; the last
    # one a slice by runs of pixels)
    cases = [
        (scenes(1), ffi.params(1024, 768, 2, seed=3)),
        (scenes(2), ffi.params(320, 200, 3, seed=9, slice_index=1, slice_count=2)),
        (scenes(1), ffi.params(200, 120, 4, seed=4, slice_index=2, slice_count=3, flags=ffi.FLAG_PIXEL_RUNS)),
    ]
    expected = []
    for scene, p in cases:
        n = 3 * p.width * p.height
        planes = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
        stats = torch.zeros(16, dtype=torch.int64, device="cuda")
        hip.trace_device(scene, p, planes.data_ptr(), stats.data_ptr(), stream)
        torch.cuda.synchronize()
        expected.append((planes, stats[:6].clone()))
    outputs = [torch.full((3 * p.width * p.height,), float("nan"), dtype=torch.float32, device="cuda") for _, p in cases]
    counters = [torch.zeros(16, dtype=torch.int64, device="cuda") for _ in cases]
    for scene, _ in cases:
        hip.upload(scene)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for (scene, p), out, st in zip(cases, outputs, counters):
        hip.trace_device_static(scene, p, out.data_ptr(), st.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before  # nothing was allocated on the device
    for (want, want_stats), out, st in zip(expected, outputs, counters):
        assert torch.equal(out, want)
        assert torch.equal(st[:6], want_stats)
    # the first frame again after the smaller one, this time with the counters in the scratch array
    again = torch.full_like(outputs[0], float("nan"))
    hip.trace_device_static(cases[0][0], cases[0][1], again.data_ptr(), None, stream)
    torch.cuda.synchronize()
    assert torch.equal(again, expected[0][0])


def test_static_scratch_entry_point_refuses_what_does_not_fit(hip, ffi, scenes):
    import torch

    scene = scenes(1)
    handle = hip.upload(scene)
    planes = torch.empty(3 * 1920 * 1080, dtype=torch.float32, device="cuda")
    # 10 samples per row: no whole pixels in a 64-column window, so the kernel cannot add up the pixels itself and
    # the frame's column sums would have to fit the array
    too_big = ffi.params(1920, 1080, 10)
    rc = hip.lib.rt_hip_trace_device_static(handle, C.byref(scene.camera), C.byref(too_big), planes.data_ptr(), None, None)
    assert rc == 1
    message = hip.lib.rt_hip_last_error()
    assert b"limit is 5570538" in message and b"20736000" in message
    fits = ffi.params(1920, 1080, 2, seed=1)  # 4.1 M columns
    hip.trace_device_static(scene, fits, planes.data_ptr(), None, None)
    torch.cuda.synchronize()
    want = torch.empty_like(planes)
    hip.trace_device(scene, fits, want.data_ptr(), None, None)
    torch.cuda.synchronize()
    assert torch.equal(planes, want)


def test_static_scratch_entry_point_takes_the_headline_frame(hip, ffi, scenes):
    """1920x1080 at 16x16 has 33 M columns, six times what the array holds as column sums: the call switches to the
    kernels that add up the pixels themselves (RT_HIP_FLAG_KERNEL_SUMS), whose only scratch is the room of sample
    sharing.  Same planes and counters as rt_hip_trace_device, which leaves column sums."""
    import torch

    scene = scenes(1)
    p = ffi.params(1920, 1080, 16, seed=1)
    want = torch.full((3 * 1920 * 1080,), float("nan"), dtype=torch.float32, device="cuda")
    want_stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    hip.trace_device(scene, p, want.data_ptr(), want_stats.data_ptr(), None)
    torch.cuda.synchronize()
    out = torch.full_like(want, float("nan"))
    stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    free_before = torch.cuda.mem_get_info()[0]
    hip.trace_device_static(scene, p, out.data_ptr(), stats.data_ptr(), None)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    assert torch.equal(out, want)
    assert torch.equal(stats[:4], want_stats[:4])
    assert int(stats[10]) == 0  # no wave gave up waiting for a window slot


def test_static_scratch_entry_point_under_graph_capture(hip, ffi, scenes):
    """The use the entry point exists for: inside a HIP graph capture, where an allocation (or a synchronising call)
    fails the capture.  Two calls of different size are captured into one graph, the graph is replayed twice, and the
    planes are those of rt_hip_trace_device, bit for bit."""
    import torch

    stream = torch.cuda.current_stream().cuda_stream
    cases = [
        (scenes(1), ffi.params(640, 480, 4, seed=2)),
        (scenes(2), ffi.params(200, 120, 3, seed=5, slice_index=1, slice_count=2)),
    ]
    want = []
    for scene, p in cases:
        planes = torch.full((3 * p.width * p.height,), float("nan"), dtype=torch.float32, device="cuda")
        st = torch.zeros(16, dtype=torch.int64, device="cuda")
        hip.trace_device(scene, p, planes.data_ptr(), st.data_ptr(), stream)
        torch.cuda.synchronize()
        want.append((planes, st[:4].clone()))
    outs = [torch.full_like(w, float("nan")) for w, _ in want]
    stats = [torch.zeros(16, dtype=torch.int64, device="cuda") for _ in cases]
    # once outside the capture: what the library works out per scene (the launch shape) is known from here on
    for (scene, p), out, st in zip(cases, outs, stats):
        hip.trace_device_static(scene, p, out.data_ptr(), st.data_ptr(), stream)
    torch.cuda.synchronize()
    for out in outs:
        out.fill_(float("nan"))
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        capturing = torch.cuda.current_stream().cuda_stream
        for (scene, p), out, st in zip(cases, outs, stats):
            hip.trace_device_static(scene, p, out.data_ptr(), st.data_ptr(), capturing)
    torch.cuda.synchronize()
    assert all(bool(torch.isnan(out).all()) for out in outs)  # captured, not run
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        for (planes, counters), out, st in zip(want, outs, stats):
            assert torch.equal(out, planes)
            assert torch.equal(st[:4], counters)
            out.fill_(float("nan"))
        torch.cuda.synchronize()
