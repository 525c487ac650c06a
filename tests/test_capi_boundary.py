"""The C-ABI library loads and exports every symbol include/dsp_gpu.h declares (CPU only:
no compute call is made here), and the product path never touches the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dsp_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsp_gpu_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from dsp_amd import capi
    assert declared_symbols() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    from dsp_amd import capi
    assert os.path.exists(capi.LIB_PATH), "libdsp_gpu.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    capi.lib()
    assert capi.lib().dsp_gpu_version().decode().startswith("dsp_gpu")


def test_argument_errors_without_a_gpu():
    """Validation happens before any HIP call, so it can be checked on CPU."""
    from dsp_amd import capi
    L = capi.lib()
    assert L.dsp_gpu_fft_batch(None, None, 4, 1, 0, 1, None) == capi.EINVAL
    assert b"null" in L.dsp_gpu_last_error_string()
    buf = (ctypes.c_float * 8)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.dsp_gpu_fft_batch(p, p, 0, 1, 0, 1, None) == capi.EINVAL
    assert L.dsp_gpu_fft_batch(p, p, 31, 1, 0, 1, None) == capi.EINVAL
    assert L.dsp_gpu_fft_batch(p, p, 2, 1, 7, 1, None) == capi.EINVAL
    assert L.dsp_gpu_fft_batch(p, p, 2, 0, 0, 1, None) == capi.OK  # empty batch is a no-op
    assert L.dsp_gpu_convolve(p, p, 1, 8, p, 0, 0, 1, None) == capi.EINVAL
    assert L.dsp_gpu_convolve(p, p, 1, 8, p, (1 << 18) + 1, capi.CONV_OVERLAP_SAVE, 1, None) == capi.EINVAL
    assert L.dsp_gpu_convolve(p, p, 1, 8, p, 5000, capi.CONV_DIRECT, 1, None) == capi.EINVAL
    assert L.dsp_gpu_convolve(p, p, 0, 8, p, 3, 0, 1, None) == capi.OK
    assert L.dsp_gpu_moving_average(p, p, 1, 4, 5, 1, None) == capi.EINVAL  # n < npoints
    assert L.dsp_gpu_iir(p, p, 1, 8, p, p, 0, 1.0, 0.0, None, 0, 1, None) == capi.EINVAL
    assert L.dsp_gpu_iir(p, p, 1, 8, p, p, 21, 1.0, 0.0, None, 0, 1, None) == capi.EINVAL
    assert L.dsp_gpu_iir_multi(p, p, 1, 8, p, p, 9, None, None, None, 1, None) == capi.EINVAL
    assert L.dsp_gpu_iir_multi(p, p, 0, 8, p, p, 2, None, None, None, 1, None) == capi.OK
    assert L.dsp_gpu_biquad_swept(p, None, p, 1, 8, 8, 0, 0.01, None, 1, None) == capi.EINVAL   # no sweep
    assert L.dsp_gpu_biquad_swept(p, p, p, 1, 8, 8, 2, 0.01, None, 1, None) == capi.EINVAL      # kind
    assert L.dsp_gpu_biquad_swept(p, p, p, 1, 8, 4, 0, 0.01, None, 1, None) == capi.EINVAL      # fc_stride not 0 / n
    assert L.dsp_gpu_biquad_swept(p, p, p, 1, 8, 8, 0, 0.01, None, 7, None) == capi.EINVAL      # mem
    assert L.dsp_gpu_biquad_swept(p, p, p, 0, 8, 8, 0, 0.01, None, 1, None) == capi.OK
    with pytest.raises(capi.DspGpuError):
        capi.check(capi.EINVAL)


def test_product_path_does_not_use_the_oracle():
    """dsp_amd/ and include/ must not import, link or call anything under oracle/."""
    bad = []
    for base in ("dsp_amd", "include", "bench_support"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "__pycache__" in dirpath or os.sep + "build" in dirpath:
                continue
            for f in files:
                if not f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                    continue
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"(from|import)\s+oracle|oracle/|dsp_oracle|libdsp_ref", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch):
    from dsp_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libdsp_gpu.so")
    with pytest.raises(capi.DspGpuError):
        capi.lib()
