"""The C-ABI libraries load and export every symbol their headers declare; no compute call is made here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_rt_hip_exports_every_declared_symbol(ffi):
    names = declared("rt_hip.h", "rt_hip_")
    assert len(names) >= 14
    lib = C.CDLL(os.path.join(ROOT, "rt_amd", "lib", "librt_hip.so"))
    for name in names:
        assert hasattr(lib, name), name
    assert set(names) == set(ffi.HIP_SYMBOLS)  # the Python binding covers the whole header


def test_rt_host_exports_every_declared_symbol(ffi):
    names = declared("rt_host.h", "rt_host_")
    lib = ffi.load_host()
    for name in names:
        assert hasattr(lib, name), name
    assert set(names) == set(ffi.HOST_SYMBOLS)


def test_struct_layouts_match_the_header(ffi):
    assert C.sizeof(ffi.Material) == 24 and C.sizeof(ffi.Light) == 28 and C.sizeof(ffi.Camera) == 48
    assert C.sizeof(ffi.TraceParams) == 32 and C.sizeof(ffi.Stats) == 80


def test_calls_before_init_fail_with_a_message(ffi):
    lib = ffi.load_hip()
    handle = C.c_void_p()
    rc = lib.rt_hip_scene_create(None, None, 0, None, 0, None, 0, C.byref(handle))
    if rc != 0:  # not initialised (no GPU here) -> a clear error, nothing created
        assert handle.value is None
        assert b"rt_hip_init" in lib.rt_hip_last_error()


def test_no_cpu_fallback_anywhere():
    """The shipped sources never reach for the oracle or the host emulation."""
    offenders = []
    for top in ("rt_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for name in files:
                if name.endswith((".so", ".pyc")) or "bin" in dirpath.split(os.sep):
                    continue
                text = open(os.path.join(dirpath, name), errors="ignore").read()
                if re.search(r"rt_oracle|oracle_ffi|core_emu|emu_ffi", text):
                    offenders.append(os.path.join(dirpath, name))
    assert offenders == []


def test_cli_refuses_to_run_without_its_backend():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the refusal path is exercised on CPU-only hosts")
    cli = os.path.join(ROOT, "rt_amd", "bin", "rt_cli")
    out = subprocess.run([cli, "-w", "32", "-h", "24"], capture_output=True, text=True)
    assert out.returncode == 2 and "HIP backend error" in out.stdout
    out = subprocess.run([cli, "--backend", "cpu"], capture_output=True, text=True)
    assert out.returncode == 2  # there is no CPU renderer in the product
