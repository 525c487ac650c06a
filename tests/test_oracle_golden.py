"""The CPU oracle against the golden fixture (outputs of the reference's own headers).

Bit-exact: the fixture was produced by oracle/_ref (the reference compiled unmodified) and
the restatement performs the same arithmetic in the same order and precision.
"""
import numpy as np
import pytest


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def test_kat_fft8(orc, golden):
    # fft0([1..8]) with the e^{+i} FORWARD sign, unnormalised both ways
    fwd = orc.fft0(golden["kat_fft8_in"])
    assert same(fwd, golden["kat_fft8_fwd"])
    s = 4 * np.sqrt(0.5) * 2
    want = np.array([36, -4 - 1j * (4 + s), -4 - 4j, -4 - 1j * (s - 4), -4, -4 + 1j * (s - 4), -4 + 4j, -4 + 1j * (4 + s)])
    assert np.allclose(fwd, want, atol=1e-12)
    back = orc.fft0(fwd, inverse=True)
    assert same(back, golden["kat_fft8_inv_of_fwd"])
    assert np.allclose(back, 8 * golden["kat_fft8_in"], atol=1e-12)  # no 1/N anywhere (fft.h:40-54)


def test_bit_reverse(orc, golden):
    got = np.array([orc.bit_reverse(i, 4096) for i in range(4096)], dtype=np.int32)
    assert same(got, golden["kat_bitrev_4096"])


@pytest.mark.parametrize("n", [2, 8, 64, 1024, 4096])
def test_fft0_golden(orc, golden, n):
    x = golden[f"fft_{n}_in"].astype(np.complex128)
    assert same(orc.fft0(x), golden[f"fft_{n}_fwd"])
    assert same(orc.fft0(x, inverse=True), golden[f"fft_{n}_inv"])


def test_fft0_matches_numpy_convention(orc, golden):
    # independent cross-check: reference FORWARD == N * ifft, INVERSE == fft
    x = golden["fft_1024_in"].astype(np.complex128)
    assert np.allclose(orc.fft0(x), 1024 * np.fft.ifft(x), rtol=0, atol=1e-9)
    assert np.allclose(orc.fft0(x, inverse=True), np.fft.fft(x), rtol=0, atol=1e-9)


@pytest.mark.parametrize("tag,nf", [("impulse", 64), ("rect", 128), ("sinc", 128), ("pad", 300)])
def test_fft_real_pairs(orc, golden, tag, nf):
    assert same(orc.fft_real(golden[f"pairs_{tag}_in"], nf), golden[f"pairs_{tag}_out"])


def test_fft_real_inverse_and_shift(orc, golden):
    assert same(orc.fft_real(golden["pairs_pad_in"], 300, inverse=True), golden["pairs_pad_out_inv"])
    assert same(orc.shift(golden["shift_in"]), golden["shift_out"])


def test_sinc_designs(orc, golden):
    assert same(orc.sinc(0.14, 10), golden["sinc_0p14_10"])
    assert golden["sinc_0p14_10"].size == 11  # odd length forced (dsp.h:60)
    assert same(orc.sinc(0.1, 511), golden["sinc_0p1_511"])
    assert same(orc.sinc(0.14, 200), golden["sinc_0p14_200"])
    assert golden["sinc_0p14_200"].size == 201
    assert same(orc.sinc(0.2, 64, "hamming", "inversion"), golden["sinc_0p2_64_hamming_inv"])
    assert same(orc.sinc(0.2, 64, "blackman", "reversal"), golden["sinc_0p2_64_blackman_rev"])
    assert same(orc.sinc(0.3, 31, "identity", "none"), golden["sinc_0p3_31_identity"])
    assert abs(golden["sinc_0p1_511"].sum() - 1.0) < 1e-12


def test_fir(orc, golden):
    h10 = golden["sinc_0p14_10"]
    y = orc.fir(golden["fir_kat_in"], h10)
    assert same(y, golden["fir_kat_out"])
    assert not y[:10].any()  # first M outputs stay zero (filter.h:133)
    assert np.allclose(y[15:26], h10.astype(np.float32))
    x = golden["fir_noise_in"]
    assert same(orc.fir(x, golden["sinc_0p1_511"]), golden["fir_noise_511_f32"])
    assert same(orc.fir(x.astype(np.float64), golden["sinc_0p1_511"]), golden["fir_noise_511_f64"])
    assert same(orc.fir(x, golden["sinc_0p2_64_hamming_inv"]), golden["fir_noise_hp64_f32"])
    assert same(orc.fir(golden["fir_tone_in"], golden["sinc_0p14_200"]), golden["fir_tone_out_f64"])
    assert same(orc.fir(golden["fir_tone_in"].astype(np.float32), golden["sinc_0p14_200"]), golden["fir_tone_out_f32"])


def test_moving_average(orc, golden):
    y = orc.moving_average(np.arange(16, dtype=np.float32), 5)
    assert same(y, golden["mavg_kat_out"])
    assert y.tolist() == [0, 0] + list(range(2, 14)) + [0, 0]
    x = golden["fir_noise_in"]
    assert same(orc.moving_average(x, 511), golden["mavg_noise_511_f32"])
    assert same(orc.moving_average(x.astype(np.float64), 511), golden["mavg_noise_511_f64"])
    assert same(orc.moving_average(x, 8), golden["mavg_noise_even8_f32"])


CHEBY = [("lp2", 0.1, 2, False), ("lp4", 0.1, 4, False), ("lp6", 0.1, 6, False), ("hp4", 0.1, 4, True),
         ("lp4_lo", 0.01, 4, False), ("lp8", 0.2, 8, False)]


@pytest.mark.parametrize("tag,fc,poles,hp", CHEBY)
def test_chebyshev(orc, golden, tag, fc, poles, hp):
    a, b = orc.chebyshev(fc, poles, hp)
    assert same(a, golden[f"cheby_{tag}_a"]) and same(b, golden[f"cheby_{tag}_b"])
    x = golden["iir_noise_in"]
    run = orc.biquad if poles == 2 else orc.iir
    assert same(run(x, a, b), golden[f"cheby_{tag}_y32"])
    assert same(run(x.astype(np.float64), a, b), golden[f"cheby_{tag}_y64"])


def test_chebyshev_kat_values(golden):
    # SURVEY.md section 4 KAT for lowPassFilter<4>(0.1)
    assert np.allclose(golden["cheby_lp4_a"], [0.002780756868, 0.01112302747, 0.01668454121, 0.01112302747, 0.002780756868], rtol=1e-9)
    assert np.allclose(golden["cheby_lp4_b"], [0, 2.764030505, -3.122852678, 1.664553024, -0.3502229603], rtol=1e-9)
    assert np.allclose(golden["cheby_lp4_impulse32"][:6], [0.00278076, 0.01880912, 0.05998964, 0.12282682, 0.18527392, 0.22180125], rtol=1e-6)


def test_iir_carried_state(orc, golden):
    a, b = golden["cheby_lp4_a"], golden["cheby_lp4_b"]
    x = golden["iir_noise_in"]
    st = np.zeros(10)
    y = np.concatenate([orc.iir(x[:1500], a, b, state=st), orc.iir(x[1500:], a, b, state=st)])
    assert same(y, golden["cheby_lp4_y32_split"])
    assert same(st, golden["cheby_lp4_state_end"])
    assert same(y, golden["cheby_lp4_y32"])  # chunking with carried state changes nothing


BQ = [("bp", "bandpass", (0.1, 0.01)), ("br", "bandreject", (0.2, 0.02)), ("ls", "lowshelf", (0.3, 6.0)),
      ("hs", "highshelf", (0.5, -3.0)), ("pk", "peaking", (0.4, 4.0, 0.7))]


@pytest.mark.parametrize("tag,kind,args", BQ)
def test_biquads(orc, golden, tag, kind, args):
    a, b, c0, d0 = orc.biquad_design(kind, *args)
    assert same(a, golden[f"bq_{tag}_a"]) and same(b, golden[f"bq_{tag}_b"])
    assert same(np.array([c0, d0]), golden[f"bq_{tag}_cd"])
    x = golden["iir_noise_in"]
    assert same(orc.biquad(x, a, b, c0, d0), golden[f"bq_{tag}_y32"])
    assert same(orc.biquad(x.astype(np.float64), a, b, c0, d0), golden[f"bq_{tag}_y64"])


def test_chain_impulse_filter_fft(orc, golden):
    # eval/recursive_filters/main.cpp:104-121: impulse at N/2 -> filter -> 1024-point fft
    imp = np.zeros(1024)
    imp[512] = 1
    resp = orc.biquad(imp, golden["cheby_lp2_a"], golden["cheby_lp2_b"])
    assert same(resp, golden["chain_lp2_response"])
    assert same(orc.fft_real(resp, 1024), golden["chain_lp2_fft"])
    a, b, c0, d0 = orc.biquad_design("peaking", 0.4, 4.0, 0.7)
    resp = orc.biquad(imp, a, b, c0, d0)
    assert same(resp, golden["chain_pk_response"])
    assert same(orc.fft_real(resp, 1024), golden["chain_pk_fft"])


@pytest.mark.parametrize("tag", ["wh0", "wh1"])
def test_swept_bandpass(orc, golden, tag):
    # eval/signals/wind_generator.h:150-167: design from fc[t], overwrite a and b, one BiQuad step -- per sample
    x, fc, bw = golden["iir_noise_in"], golden[f"swept_{tag}_fc"], float(golden["swept_bw"][0])
    st = np.zeros(4)
    assert same(orc.biquad_swept(x, fc, bw, "bandpass", state=st), golden[f"swept_{tag}_y32"])
    assert same(st, golden[f"swept_{tag}_state"])
    # fed in blocks with the BiQuad's members carried: nothing changes
    st = np.zeros(4)
    y = np.concatenate([orc.biquad_swept(x[:999], fc[:999], bw, state=st), orc.biquad_swept(x[999:], fc[999:], bw, state=st)])
    assert same(y, golden[f"swept_{tag}_y32"]) and same(st, golden[f"swept_{tag}_state"])
    # a constant sweep is the fixed band-pass
    fconst = np.full(x.size, fc[0], dtype=np.float32)
    a, b, c0, d0 = orc.biquad_design("bandpass", float(fc[0]), bw)
    assert same(orc.biquad_swept(x, fconst, bw), orc.biquad(x, a, b, c0, d0))


def test_swept_bandreject(orc, golden):
    x, fc, bw = golden["iir_noise_in"], golden["swept_wh0_fc"], float(golden["swept_bw"][0])
    assert same(orc.biquad_swept(x, fc, bw, "bandreject"), golden["swept_wh0_reject_y32"])
