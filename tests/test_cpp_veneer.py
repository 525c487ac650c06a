"""The C++ drop-in layer (include/dsp_mi355x/dsp.hpp) driven through tests/cpp/veneer_check.

CPU: the host-side designs it computes are the reference's, bit for bit (golden fixture), and the
header compiles against the reference's own SampleBuffer where /root/reference exists.
GPU: every veneer call, used the way the reference's eval programs use the reference, against the oracle.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "veneer_check")


def _exe():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True, capture_output=True)
    return EXE


def test_designs_match_golden(golden):
    out = subprocess.run([_exe(), "design"], check=True, capture_output=True, text=True).stdout
    got = {ln.split()[0]: np.array([float(v) for v in ln.split()[1:]]) for ln in out.splitlines()}
    for key in ("sinc_0p14_10", "sinc_0p1_511", "sinc_0p14_200", "sinc_0p2_64_hamming_inv", "sinc_0p2_64_blackman_rev",
                "sinc_0p3_31_identity"):
        assert got[key].tobytes() == golden[key].tobytes(), key
    assert got["filter_kernel_0p14_200"].tobytes() == golden["sinc_0p14_200"].tobytes()
    for tag in ("lp2", "lp4", "lp6", "hp4", "lp4_lo", "lp8"):
        assert got[f"cheby_{tag}_a"].tobytes() == golden[f"cheby_{tag}_a"].tobytes(), tag
        assert got[f"cheby_{tag}_b"].tobytes() == golden[f"cheby_{tag}_b"].tobytes(), tag
    for tag in ("bp", "br", "ls", "hs", "pk"):
        for part in ("a", "b", "cd"):
            assert got[f"bq_{tag}_{part}"].tobytes() == golden[f"bq_{tag}_{part}"].tobytes(), (tag, part)


def test_compiles_against_the_reference_container():
    if not os.path.isdir("/root/reference/dsp/include/dsp"):
        pytest.skip("no /root/reference here")
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "syntax_with_reference"], check=True,
                   capture_output=True)


@pytest.mark.gpu
def test_veneer_on_gpu(tmp_path, orc, golden, dev):
    rng = np.random.default_rng(77)
    d = str(tmp_path)
    fft_in = (rng.standard_normal((3, 1024)) + 1j * rng.standard_normal((3, 1024))).astype(np.complex64)
    fft_in.tofile(os.path.join(d, "fft_in.c64"))
    fir_in = rng.standard_normal((2, 6000)).astype(np.float32)
    fir_in.tofile(os.path.join(d, "fir_in.f32"))
    iir_in = golden["iir_noise_in"]
    iir_in.tofile(os.path.join(d, "iir_in.f32"))
    golden["swept_wh0_fc"].tofile(os.path.join(d, "swept_fc.f32"))
    dist_in = (rng.standard_normal(1 << 16) + 1j * rng.standard_normal(1 << 16)).astype(np.complex64)
    dist_in.tofile(os.path.join(d, "dist_in.c64"))
    p = subprocess.run([_exe(), "run", d], capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stderr
    rd = lambda name, dt: np.fromfile(os.path.join(d, name), dtype=dt)

    assert rel_l2(rd("dist_fwd.c64", np.complex64), orc.fft0(dist_in.astype(np.complex128))) <= 1.6e-4
    assert rel_l2(rd("dist_inv.c64", np.complex64), orc.fft0(dist_in.astype(np.complex128), True)) <= 1.6e-4

    fwd = orc.fft0(fft_in.astype(np.complex128))
    assert rel_l2(rd("fft_fwd.c64", np.complex64).reshape(3, 1024), fwd) <= 1e-4
    assert rel_l2(rd("fft_back.c64", np.complex64).reshape(3, 1024), 1024 * fft_in.astype(np.complex128)) <= 2e-4
    assert rel_l2(rd("fft_fwd.c128", np.complex128).reshape(3, 1024), fwd) <= 1e-13  # double route: the reference's own type
    assert rel_l2(rd("fft_real_double.c128", np.complex128), orc.fft_real(fft_in[0].real[:1000].astype(np.float64), 1024)) <= 1e-13

    assert rel_l2(rd("chain_response.f32", np.float32), orc.iir(np.eye(1, 1024, 512)[0], golden["cheby_lp4_a"], golden["cheby_lp4_b"])) <= 2e-6
    resp = rd("chain_response.f32", np.float32).astype(np.float64)
    assert rel_l2(rd("chain_fft.c128", np.complex128), orc.fft_real(resp, 1024)) <= 1e-4

    h = golden["sinc_0p1_511"]
    want = orc.fir(fir_in.astype(np.float64), h)
    assert rel_l2(rd("fir_direct.f32", np.float32).reshape(2, 6000), want) <= 1e-6
    assert rel_l2(rd("conv_time.f32", np.float32), want[0]) <= 1e-6
    assert rel_l2(rd("fir_streamed.f32", np.float32), want[0]) <= 1e-6  # 256-sample blocks, tail carried by the object
    assert rel_l2(rd("conv_freq.f32", np.float32), want[0]) <= 1.2e-4
    assert rel_l2(rd("mavg.f32", np.float32), orc.moving_average(fir_in[0].astype(np.float64), 511)) <= 1e-6
    assert rel_l2(rd("mavg8.f32", np.float32), orc.moving_average(fir_in[0].astype(np.float64), 8)) <= 1e-6

    assert rel_l2(rd("iir_lp4.f32", np.float32), golden["cheby_lp4_y64"]) <= 2e-6
    st = rd("iir_lp4_state.f64", np.float64)
    ref = golden["cheby_lp4_state_end"]
    assert np.array_equal(st[1:5], ref[1:5]) and st[0] == 0
    assert np.allclose(st[6:10], ref[6:10], rtol=1e-5, atol=1e-7)
    assert rel_l2(rd("iir_ring.f32", np.float32), golden["cheby_lp4_y64"]) <= 2e-6   # 256-frame blocks, object history
    assert rel_l2(rd("iir_ls.f32", np.float32), golden["bq_ls_y64"]) <= 2e-6
    # the retuned band-pass against what the reference build produced for the same voice (golden), blocks of 256
    swept, want = rd("swept.f32", np.float32), golden["swept_wh0_y32"]
    assert np.abs(swept.astype(np.float64) - want).max() <= np.spacing(np.abs(want).max()) and rel_l2(swept, want.astype(np.float64)) <= 1e-7
    members = rd("swept_members.f64", np.float64)
    assert np.allclose(members[:4], golden["swept_wh0_state"], rtol=1e-9, atol=1e-12)
    a, b, _, _ = orc.biquad_design("bandpass", float(golden["swept_wh0_fc"][-1]), float(golden["swept_bw"][0]))
    assert np.array_equal(members[4:7], a) and np.array_equal(members[7:9], b[1:])  # leaves with the last design
    rows = iir_in.reshape(4, -1).astype(np.float64)
    assert rel_l2(rd("iir_rows.f32", np.float32).reshape(4, -1), orc.iir(rows, golden["cheby_lp4_a"], golden["cheby_lp4_b"])) <= 2e-6


@pytest.mark.gpu
def test_reference_callers_on_gpu(tmp_path, orc, dev):
    """tests/cpp/reference_callers_check.cpp: the numeric work of eval/filter_compare, eval/recursive_filters (createDate /
    applyFilter), eval/moving_average_filter_demo and eval/windowed_sinc_filter on SampleBuffer<double>, written against
    `namespace dsp` exactly as those programs call the reference, against the oracle's double functions."""
    exe = os.path.join(ROOT, "tests", "cpp", "reference_callers_check")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "reference_callers_check"], check=True, capture_output=True)
    d = str(tmp_path)
    rng = np.random.default_rng(2024)
    noise = rng.standard_normal(5000)
    noise.tofile(os.path.join(d, "noise.f64"))
    p = subprocess.run([exe, d], capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stderr
    rd = lambda name: np.fromfile(os.path.join(d, name), dtype=np.float64)

    imp = np.zeros(401)
    imp[300] = 1.0
    assert np.array_equal(rd("filter_compare.f64"), orc.fir(imp, orc.sinc(0.14, 200)))  # one product per output: exact

    f, bw, g = float(np.float32(0.25)), float(np.float32(0.006)), 1.0  # the callers hold these in float
    designs = {
        "low_pass": orc.chebyshev(f, 2) + (1.0, 0.0),
        "high_pass": orc.chebyshev(f, 2, highpass=True) + (1.0, 0.0),
        "band_pass": orc.biquad_design("bandpass", f, bw),
        "band_reject": orc.biquad_design("bandreject", f, bw),
        "low_shelf": orc.biquad_design("lowshelf", f, g),
        "high_shelf": orc.biquad_design("highshelf", f, g),
        "peaking": orc.biquad_design("peaking", f, g, bw),
    }
    impulse = np.zeros(1024)
    impulse[512] = 1.0
    for kind, (a, b, c0, d0) in designs.items():
        want = orc.biquad(impulse, a, b, c0, d0)
        got = rd(f"recursive_{kind}_response.f64")
        assert rel_l2(got, want) <= 2e-12, kind
        mag = np.abs(orc.fft_real(got, 1024))
        assert rel_l2(rd(f"recursive_{kind}_magnitude.f64"), mag) <= 1e-13, kind

    a4, b4 = orc.chebyshev(0.1, 4)
    assert rel_l2(rd("noise_lp4.f64"), orc.iir(noise, a4, b4)) <= 2e-12
    assert np.array_equal(rd("noise_lp4_ref.f64"), orc.iir(noise, a4, b4))          # DSP_GPU_IIR_BATCH_REF: bit for bit
    a, b, c0, d0 = designs["peaking"]
    assert np.array_equal(rd("noise_peaking_ref.f64"), orc.biquad(noise, a, b, c0, d0))
    for size in (1, 8, 51):
        assert rel_l2(rd(f"mavg_{size}.f64"), orc.moving_average(noise, size)) <= 1e-12, size
    assert rel_l2(rd("sinc_hp.f64"), orc.fir(noise, orc.sinc(0.2, 200, inversion="inversion"))) <= 1e-13
    assert rel_l2(rd("sinc_lp_streamed.f64"), orc.fir(noise, orc.sinc(0.14, 200))) <= 1e-13
    # per-sample calls: the reference's step bit for bit (generic section and BiQuad; float samples accumulate in float)
    assert np.array_equal(rd("per_sample_lp4.f64"), orc.iir(noise[:64], a4, b4))
    abp, bbp, c0bp, d0bp = designs_bp = orc.biquad_design("bandpass", 0.1, 0.01)
    assert np.array_equal(rd("per_sample_bp.f64"), orc.biquad(noise[:64], abp, bbp, c0bp, d0bp))
    assert np.array_equal(rd("per_sample_lp4_float.f64"), orc.iir(noise[:64].astype(np.float32), a4, b4).astype(np.float64))
    shifted = rd("spectrum_shifted.c64").view(np.complex128)
    assert rel_l2(shifted, orc.shift(orc.fft_real(noise[:1000], 1024))) <= 1e-13


def test_swept_cos_matches_libm():
    """dsp_amd/csrc/swept_cos.h (the cosine inside the per-sample band-pass design, recursive_filters.h:169) compiled
    for the host, against the C library's cos over 2^24 frequencies in [0, 0.5], the quadrant edges and tiny
    frequencies: never more than one ulp (of the result, and 2^-53 absolute) away."""
    exe = os.path.join(ROOT, "tests", "cpp", "swept_cos_check")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "swept_cos_check"], check=True, capture_output=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    got = dict(zip(out[0::2], out[1::2]))
    assert float(got["max_abs_ulp1"]) <= 1.0 and float(got["max_rel_ulp"]) <= 1.0, got
    assert int(got["samples"]) > (1 << 24) and float(got["wild_abs"]) < 1e-12, got
