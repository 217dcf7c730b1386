"""The drop-in boundary on a real GPU: the asynchronous job API, the Frame plug-in on top of it, the CLI."""
import ctypes as C
import json
import os
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_is_an_mi355x(hip):
    info = hip.device_info()
    assert "gfx950" in info["name"]
    assert info["compute_units"] == 256


def test_job_runs_asynchronously_and_snapshots_progress(hip, ffi, scenes):
    scene = scenes(2)
    p = ffi.params(3840, 2160, 32, seed=3)  # long enough (0.5 s) for the job to keep room for snapshots
    t0 = time.perf_counter()
    job = hip.start(scene, p)
    launch = time.perf_counter() - t0
    assert hip.poll(job) == ffi.RT_HIP_ERR_NOT_READY
    snap = hip.snapshot(job, p)  # must not wait for the trace (the first one also makes the snapshot buffers)
    t1 = time.perf_counter()
    snap = hip.snapshot(job, p)
    snapshot_ms = (time.perf_counter() - t1) * 1e3
    still_running = hip.poll(job) == ffi.RT_HIP_ERR_NOT_READY
    print("snapshot of a running 3840x2160 frame: %.2f ms of host time (accumulate + resolve on the copy stream, "
          "99.5 MB into pinned memory, one memcpy)" % snapshot_ms)
    # (no bound on the host time itself: pinned-memory copies on a shared box vary several-fold; that the job is
    # still running after two snapshots, checked below, is what shows that a snapshot does not wait for the trace)
    final, st = hip.finish(job, p)
    assert launch < 0.2  # returned long before the frame finished (0.5 s)
    # a snapshot holds finished columns only: every pixel is at most its final value, unfinished ones darker
    assert (snap <= final * (1 + 1e-5) + 1e-6).all()
    assert still_running and (snap.sum(axis=2) < final.sum(axis=2) * 0.999).any()
    assert st.samples == 3840 * 2160 * 1024 and st.kernel_ms > 0


def test_cancel_stops_the_frame_early(hip, ffi, scenes):
    scene = scenes(1)
    p = ffi.params(3840, 2160, 8, seed=3)
    job = hip.start(scene, p)
    hip.cancel(job)
    _, st = hip.finish(job, p, allow_cancelled=True)
    assert st.samples < 3840 * 2160 * 64


def test_invalid_arguments_are_reported(hip, ffi, scenes):
    lib = hip.lib
    scene = scenes(1)
    handle = hip.upload(scene)
    job = C.c_void_p()
    bad = ffi.params(0, 10, 2)
    assert lib.rt_hip_trace_async(handle, C.byref(scene.camera), C.byref(bad), C.byref(job)) == 1
    assert b"positive" in lib.rt_hip_last_error()
    bad = ffi.params(8, 8, 2, slice_index=3, slice_count=2)
    assert lib.rt_hip_trace_async(handle, C.byref(scene.camera), C.byref(bad), C.byref(job)) == 1
    mats = (ffi.Material * 1)()
    mats[0].type = 7
    out = C.c_void_p()
    assert lib.rt_hip_scene_create(None, None, 0, mats, 1, None, 0, C.byref(out)) == 1
    assert b"material type" in lib.rt_hip_last_error()


def test_device_entry_points_match_the_job_api(hip, ffi, scenes):
    import torch

    scene = scenes(1)
    p = ffi.params(200, 120, 3, seed=4)
    ref, rs = hip.render(scene, p)
    planes = torch.empty(3 * 200 * 120, dtype=torch.float32, device="cuda")
    rgb = torch.empty(200 * 120 * 3, dtype=torch.float32, device="cuda")
    stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    hip.trace_device(scene, p, planes.data_ptr(), stats.data_ptr(), stream)
    hip.resolve_device(planes.data_ptr(), rgb.data_ptr(), p, stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rgb.cpu().numpy().reshape(120, 200, 3), ref)
    assert int(stats[0]) == rs.samples and int(stats[1]) == rs.ray_queries


def test_frame_plugin_renders_the_demo(hip, ffi, scenes):
    """DemoFrameHIP driven as the reference's App drives a Frame equals the C ABI called directly."""
    p = ffi.params(160, 120, 2, seed=6)
    direct, _ = hip.render(scenes(1), p)
    via_frame, st = ffi.render_demo(1, p, scene_seed=1)
    np.testing.assert_array_equal(via_frame, direct)
    assert st.samples == 160 * 120 * 4


def test_cli_writes_the_frame(hip, ffi, scenes, tmp_path):
    out = tmp_path / "frame.pfm"
    cmd = [os.path.join(ROOT, "rt_amd", "bin", "rt_cli"), "-w", "96", "-h", "64", "-s", "2", "-d", "2", "--seed", "5",
           "--out", str(out), "--progress"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["samples"] == 96 * 64 * 4 and info["backend"] == "hip"
    raw = out.read_bytes()
    header, data = raw.split(b"-1.0\n", 1)
    assert header.split() == [b"PF", b"96", b"64"]
    img = np.frombuffer(data, dtype="<f4").reshape(64, 96, 3)[::-1]  # PFM rows run bottom-up
    ref, _ = hip.render(scenes(2), ffi.params(96, 64, 2, seed=5))
    np.testing.assert_array_equal(img, ref)
    # 8-bit output goes through the viewer's conversion
    ppm = tmp_path / "frame.ppm"
    subprocess.run(cmd[:-3] + ["--out", str(ppm)], check=True, capture_output=True, timeout=120)
    body = ppm.read_bytes().split(b"255\n", 1)[1]
    np.testing.assert_array_equal(np.frombuffer(body, dtype=np.uint8).reshape(64, 96, 3), ffi.update_texture(ref)[..., :3])


def test_cli_renders_a_scene_file(hip, ffi, scenes, tmp_path):
    """--dump-scene writes the chosen demo; --scene renders that file to the same frame.  (The light direction
    and the already normalised camera direction survive the text form; re-normalising the latter moves it by an
    ulp at most, hence the tolerance.)"""
    cli = os.path.join(ROOT, "rt_amd", "bin", "rt_cli")
    scene_path, a_path, b_path = tmp_path / "demo1.rtscene", tmp_path / "a.pfm", tmp_path / "b.pfm"
    common = ["-w", "96", "-h", "64", "-s", "2", "--deterministic"]
    subprocess.run([cli, "-d", "1", "--dump-scene", str(scene_path), "--out", str(a_path)] + common, check=True,
                   capture_output=True, timeout=120)
    res = subprocess.run([cli, "--scene", str(scene_path), "--out", str(b_path)] + common, capture_output=True,
                         text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr

    def read(path):
        return np.frombuffer(path.read_bytes().split(b"-1.0\n", 1)[1], dtype="<f4").reshape(64, 96, 3)

    np.testing.assert_allclose(read(b_path), read(a_path), rtol=1e-4, atol=1e-6)
    assert ffi.Scene.load(str(scene_path)).sphere_count == 41
    bad = tmp_path / "bad.rtscene"
    bad.write_text("rt-scene 1\nsphere 0 0 0 1 nope\n")
    res = subprocess.run([cli, "--scene", str(bad)] + common, capture_output=True, text=True, timeout=120)
    assert res.returncode == 1 and "unknown material" in res.stdout


@pytest.mark.parametrize("world", [2, 3, 8])
def test_device_path_slices_reduce_to_the_frame(hip, ffi, scenes, world):
    """What bench.py does on several GPUs, replayed on one: every rank's slice traced into its own planes with
    rt_hip_trace_device, the planes summed (the RCCL reduce), the sum resolved."""
    import torch

    from rt_amd import sharding

    scene = scenes(1)
    w, h, s = 160, 90, 4
    full = ffi.params(w, h, s, seed=4)
    ref, rs = hip.render(scene, full)
    total = torch.zeros(3 * w * h, dtype=torch.float32, device="cuda")
    stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    samples = 0
    stream = torch.cuda.current_stream().cuda_stream
    for rank in range(world):
        p = sharding.slice_params(lambda: ffi.params(w, h, s, seed=4), rank, world)
        planes = torch.full((3 * w * h,), float("nan"), dtype=torch.float32, device="cuda")  # must be overwritten
        hip.trace_device(scene, p, planes.data_ptr(), stats.data_ptr(), stream)
        torch.cuda.synchronize()
        total += planes
        samples += int(stats[0])
    rgb = torch.empty(w * h * 3, dtype=torch.float32, device="cuda")
    hip.resolve_device(total.data_ptr(), rgb.data_ptr(), full, stream)
    torch.cuda.synchronize()
    assert samples == rs.samples
    np.testing.assert_allclose(rgb.cpu().numpy().reshape(h, w, 3), ref, rtol=2e-6, atol=1e-7)


def test_device_path_calls_on_two_streams_overlap_safely(hip, ffi, scenes):
    """rt_hip_trace_device keeps a workspace per stream: frames in flight on two streams at once (different seeds
    and scenes, several rounds) come out exactly as when traced one at a time."""
    import torch

    w, h, s = 640, 360, 4
    jobs = [(scenes(1), ffi.params(w, h, s, seed=3)), (scenes(2), ffi.params(w, h, s, seed=8))]
    alone = []
    stats = [torch.zeros(16, dtype=torch.int64, device="cuda") for _ in jobs]
    planes = [torch.empty(3 * w * h, dtype=torch.float32, device="cuda") for _ in jobs]
    default = torch.cuda.current_stream().cuda_stream
    for i, (scene, p) in enumerate(jobs):
        hip.trace_device(scene, p, planes[i].data_ptr(), stats[i].data_ptr(), default)
        torch.cuda.synchronize()
        alone.append((planes[i].clone(), stats[i][:6].clone()))
    streams = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)]
    for _ in range(3):
        for i, (scene, p) in enumerate(jobs):
            planes[i].fill_(float("nan"))
            streams[i].wait_stream(torch.cuda.current_stream())
            hip.trace_device(scene, p, planes[i].data_ptr(), stats[i].data_ptr(), streams[i].cuda_stream)
        torch.cuda.synchronize()
        for i in range(len(jobs)):
            assert torch.equal(planes[i], alone[i][0])
            assert torch.equal(stats[i][:6], alone[i][1])


def test_device_path_serves_more_streams_than_it_has_workspaces(hip, ffi, scenes):
    import torch

    scene = scenes(1)
    w, h, s = 96, 54, 3
    p = ffi.params(w, h, s, seed=2)
    planes = torch.empty(3 * w * h, dtype=torch.float32, device="cuda")
    stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    first = None
    for _ in range(12):  # the library keeps eight sets of buffers
        stream = torch.cuda.Stream()
        hip.trace_device(scene, p, planes.data_ptr(), stats.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        if first is None:
            first = planes.clone()
        assert torch.equal(planes, first)


def test_reduce_pipeline_over_rccl_with_one_rank(tmp_path):
    """The asynchronous reduces bench.py issues from two streams, over the nccl (= RCCL) backend: all a one-GPU box
    can exercise of it is a group of one rank, which still goes through the collective calls."""
    import socket
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_pipeline_check.py"), str(port)],
                         capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout + res.stderr
    assert "pipeline ok" in res.stdout


def test_bench_prints_one_json_line_with_the_contract_fields(tmp_path):
    """bench.py on a small frame: one JSON line on stdout with the fields the driver reads."""
    import sys

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--width", "320", "--height", "180", "--samples", "4"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["unit"] == "Msamples/s" and out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1
    assert out["vs_baseline"] is None and out["dtype"] == "f32" and out["higher_is_better"] is True
    assert "workload" in out["config"] and "model" not in out["config"]
    assert out["value"] == pytest.approx(320 * 180 * 16 / (out["ms_per_step"] * 1e-3) / 1e6, rel=5e-3)
    roof = out["roofline"]
    # the top-level fraction is executed work (vector instructions issued against the issue peak): at most 1
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], abs=1e-3) and roof["unit"] == "T lane-instructions/s"
    assert 0.0 < roof["frac"] <= 1.0 and roof["frac_source"].startswith("model")  # no counter profile of this workload
    assert roof["algorithmic_frac"] == pytest.approx(roof["algorithmic_tflops"] / roof["algorithmic_peak_tflops"], abs=2e-4)
    assert roof["kernel_ms"] > 0 and roof["traffic"] is None  # traffic is only known for the profiled workloads
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0


def test_two_rank_bench_rehearsal_on_one_gpu(tmp_path):
    """The multi-rank flow of bench.py (slices, reduce to rank 0, resolve, counters, one JSON line from rank 0)
    with two processes that share this GPU and reduce over gloo: same frame as one rank."""
    import sys

    common = ["--steps", "2", "--warmup", "1", "--width", "320", "--height", "180", "--samples", "4",
              "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common,
                         capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert one.returncode == 0, one.stdout + one.stderr
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--rehearse"] + common,
                         capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert two.returncode == 0, two.stdout + two.stderr
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and "cpu_baseline" not in b
    assert b["mean_radiance"] == pytest.approx(a["mean_radiance"], abs=2e-5)
    assert b["ray_queries_per_sample"] == pytest.approx(a["ray_queries_per_sample"], abs=1e-3)


def test_oversized_frames_are_refused_before_anything_is_allocated(hip, ffi, scenes):
    lib = hip.lib
    scene = scenes(1)
    handle = hip.upload(scene)
    job = C.c_void_p()
    for bad, word in ((ffi.params(8, 8, 5000), b"4096"), (ffi.params(65536, 65536, 1), b"2^30"),
                      (ffi.params(32768, 32768, 4), b"0xF0000000")):
        assert lib.rt_hip_trace_async(handle, C.byref(scene.camera), C.byref(bad), C.byref(job)) == 1
        assert word in lib.rt_hip_last_error(), lib.rt_hip_last_error()
        assert not job.value
    # the largest supersampling the interface takes, on a tiny frame
    p = ffi.params(2, 2, 4096, seed=1)
    out, st = hip.render(scene, p)
    assert st.samples == 4 * 4096 * 4096 and np.isfinite(out).all()


@pytest.mark.parametrize("demo", [1, 2])
def test_same_demo_around_either_frame(oracle, ffi, demo):
    """The drop-in, end to end: one scene builder (demo::PrettySpheres / GlassSpheres), once around DemoFrameHIP and
    once around the CPU checker's DemoFrameCPU, both driven through the Frame virtuals only."""
    p = ffi.params(128, 96, 2, flags=ffi.FLAG_DETERMINISTIC)
    hip_rgb, hs = ffi.render_demo(demo, p, scene_seed=1)
    cpu_rgb, cs = oracle.render_demo_through_frame(demo, p, scene_seed=1)
    err = np.abs(cpu_rgb - hip_rgb) / np.maximum(np.abs(cpu_rgb), 1e-3)
    assert err.max() < 1e-4
    assert (hs.samples, hs.ray_queries, hs.hit_queries) == (cs.samples, cs.ray_queries, cs.hit_queries)


def test_jobs_from_two_threads_on_one_scene(hip, ffi, scenes):
    """Every job has its own camera table and buffers, so two threads may render different views of one scene at
    the same time (the scene itself is immutable): each frame equals the one rendered alone."""
    import copy
    import threading

    base = scenes(1)
    views = []
    for k, shift in enumerate((0.0, 1.5)):
        s = ffi.Scene(base.centers_radii, base.material_index, base.materials, base.lights, copy.copy(base.camera))
        s.material_count, s.light_count = base.material_count, base.light_count
        s.camera.position[0] = base.camera.position[0] + shift
        views.append((s, ffi.params(320 + 64 * k, 200, 3, seed=5 + k)))
    handle = hip.upload(base)
    for s, _ in views:
        hip._scenes[id(s)] = (handle, s)  # the same device scene, another camera
    alone = [hip.render(s, p)[0] for s, p in views]
    results = [[None] * 6 for _ in views]

    def work(i):
        s, p = views[i]
        for n in range(6):
            results[i][n] = hip.render(s, p)[0]

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(views))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for s, _ in views:
        del hip._scenes[id(s)]
    for i in range(len(views)):
        for frame in results[i]:
            np.testing.assert_array_equal(frame, alone[i])
