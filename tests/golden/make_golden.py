#!/usr/bin/env python3
"""Generate tests/golden/dsp_golden.npz by EXECUTING the reference (oracle/_ref).

Run from the repo root in the build container (needs /root/reference):

    make -C oracle && python tests/golden/make_golden.py

The fixture holds only data: seeded inputs and the outputs the reference's own headers
(compiled unmodified into oracle/_ref/libdsp_ref.so) produced for them.  The recipes
follow what the reference's eval programs look at (file:line relative to /root/reference):

  kat_*        smallest known-answer vectors
  fft_*        fft0 forward+inverse on float-exact N(0,1) inputs, N in {2,8,64,1024,4096}
  pairs_*      eval/fft_pairs/main.cpp:56-89   impulse / rect(+-5) / sinc through dsp::fft
  sinc_*       dsp.h:47-79 tap sets; fir_* = filter.h:123-140 on noise / impulse / 5-tone
               (5-tone recipe: eval/windowed_sinc_filter/main.cpp:48-65, 1 kHz sample rate)
  mavg_*       filter.h:81-101
  cheby_*      recursive_filters.h:153-163 designs + 4096-sample noise responses
  bq_*         recursive_filters.h:165-263 designs + responses (shelves exercise c0/d0)
  chain_*      eval/recursive_filters/main.cpp:104-121  impulse@N/2 -> filter -> 1024-pt fft
  swept_*      eval/signals/wind_generator.h:150-167    band-pass redesigned from fc[t] before every step
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import load  # noqa: E402


def main():
    ref = load("ref")
    g = {}
    rng = np.random.default_rng(20261111)

    def noise32(*shape):
        return rng.standard_normal(shape).astype(np.float32)

    # --- KATs ---------------------------------------------------------------
    g["kat_fft8_in"] = (np.arange(1, 9) + 0j).astype(np.complex128)
    g["kat_fft8_fwd"] = ref.fft0(g["kat_fft8_in"])
    g["kat_fft8_inv_of_fwd"] = ref.fft0(g["kat_fft8_fwd"], inverse=True)
    g["kat_bitrev_4096"] = np.array([ref.bit_reverse(i, 4096) for i in range(4096)], dtype=np.int32)

    # --- fft0 on float-exact inputs ---------------------------------------------
    for n in (2, 8, 64, 1024, 4096):
        x = (noise32(n) + 1j * noise32(n)).astype(np.complex64)
        g[f"fft_{n}_in"] = x
        g[f"fft_{n}_fwd"] = ref.fft0(x.astype(np.complex128))
        g[f"fft_{n}_inv"] = ref.fft0(x.astype(np.complex128), inverse=True)

    # --- eval/fft_pairs recipes (real wrapper) ------------------------------------
    imp = np.zeros(64)
    imp[3] = 1
    g["pairs_impulse_in"] = imp
    g["pairs_impulse_out"] = ref.fft_real(imp, 64)
    rect = np.zeros(128)
    for off in range(-5, 6):
        rect[(128 + 0 + off) % 128] = 1
    g["pairs_rect_in"] = rect
    g["pairs_rect_out"] = ref.fft_real(rect, 128)
    xs = ((np.arange(128, dtype=np.float32) / np.float32(128) - np.float32(0.5)) * 10).astype(np.float64)
    xx = np.pi * 1.0 * xs
    snc = np.where(xx == 0, 1.0, np.sin(xx) / np.where(xx == 0, 1.0, xx))
    g["pairs_sinc_in"] = snc
    g["pairs_sinc_out"] = ref.fft_real(snc, 128)
    # zero padding: 300 samples into nf=300 -> 512 points; inverse wrapper too
    pad = rng.standard_normal(300)
    g["pairs_pad_in"] = pad
    g["pairs_pad_out"] = ref.fft_real(pad, 300)
    g["pairs_pad_out_inv"] = ref.fft_real(pad, 300, inverse=True)
    g["shift_in"] = rng.standard_normal(64) + 1j * rng.standard_normal(64)
    g["shift_out"] = ref.shift(g["shift_in"])

    # --- FIR -----------------------------------------------------------------------
    g["sinc_0p14_10"] = ref.sinc(0.14, 10)
    g["sinc_0p1_511"] = ref.sinc(0.1, 511)
    g["sinc_0p14_200"] = ref.sinc(0.14, 200)
    g["sinc_0p2_64_hamming_inv"] = ref.sinc(0.2, 64, "hamming", "inversion")
    g["sinc_0p2_64_blackman_rev"] = ref.sinc(0.2, 64, "blackman", "reversal")
    g["sinc_0p3_31_identity"] = ref.sinc(0.3, 31, "identity", "none")
    d = np.zeros(32, dtype=np.float32)
    d[15] = 1
    g["fir_kat_in"] = d
    g["fir_kat_out"] = ref.fir(d, g["sinc_0p14_10"])
    x8k = noise32(8192)
    g["fir_noise_in"] = x8k
    g["fir_noise_511_f32"] = ref.fir(x8k, g["sinc_0p1_511"])
    g["fir_noise_511_f64"] = ref.fir(x8k.astype(np.float64), g["sinc_0p1_511"])
    g["fir_noise_hp64_f32"] = ref.fir(x8k, g["sinc_0p2_64_hamming_inv"])
    # 5-tone cosine, 1 kHz, padded by the kernel length (201)
    sr, padn = 1000, 201
    t = (1.0 / sr) * np.arange(sr + padn) - (1.0 / sr) * padn
    tone = (np.cos(2 * np.pi * 2 * t) + 0.5 * np.cos(2 * np.pi * 10 * t) + 0.25 * np.cos(2 * np.pi * 50 * t)
            + 0.125 * np.cos(2 * np.pi * 250 * t) + 0.0625 * np.cos(2 * np.pi * 450 * t))
    g["fir_tone_in"] = tone
    g["fir_tone_out_f64"] = ref.fir(tone, g["sinc_0p14_200"])
    g["fir_tone_out_f32"] = ref.fir(tone.astype(np.float32), g["sinc_0p14_200"])

    # --- moving average ----------------------------------------------------------------
    g["mavg_kat_out"] = ref.moving_average(np.arange(16, dtype=np.float32), 5)
    g["mavg_noise_511_f32"] = ref.moving_average(x8k, 511)
    g["mavg_noise_511_f64"] = ref.moving_average(x8k.astype(np.float64), 511)
    g["mavg_noise_even8_f32"] = ref.moving_average(x8k, 8)  # numPoints() does not re-force odd

    # --- IIR ---------------------------------------------------------------------------
    x4k = noise32(4096)
    g["iir_noise_in"] = x4k
    for tag, fc, poles, hp in (("lp2", 0.1, 2, False), ("lp4", 0.1, 4, False), ("lp6", 0.1, 6, False),
                               ("hp4", 0.1, 4, True), ("lp4_lo", 0.01, 4, False), ("lp8", 0.2, 8, False)):
        a, b = ref.chebyshev(fc, poles, hp)
        g[f"cheby_{tag}_a"], g[f"cheby_{tag}_b"] = a, b
        if poles == 2:
            g[f"cheby_{tag}_y32"] = ref.biquad(x4k, a, b)
            g[f"cheby_{tag}_y64"] = ref.biquad(x4k.astype(np.float64), a, b)
        else:
            g[f"cheby_{tag}_y32"] = ref.iir(x4k, a, b)
            g[f"cheby_{tag}_y64"] = ref.iir(x4k.astype(np.float64), a, b)
    impulse = np.zeros(32, dtype=np.float32)
    impulse[0] = 1
    g["cheby_lp4_impulse32"] = ref.iir(impulse, g["cheby_lp4_a"], g["cheby_lp4_b"])
    # carried state: two halves with the reference's own x[]/y[] history handed over
    st = np.zeros(10)
    g["cheby_lp4_y32_split"] = np.concatenate([ref.iir(x4k[:1500], g["cheby_lp4_a"], g["cheby_lp4_b"], state=st),
                                               ref.iir(x4k[1500:], g["cheby_lp4_a"], g["cheby_lp4_b"], state=st)])
    g["cheby_lp4_state_end"] = st.copy()

    for tag, kind, args in (("bp", "bandpass", (0.1, 0.01)), ("br", "bandreject", (0.2, 0.02)),
                            ("ls", "lowshelf", (0.3, 6.0)), ("hs", "highshelf", (0.5, -3.0)),
                            ("pk", "peaking", (0.4, 4.0, 0.7))):
        a, b, c0, d0 = ref.biquad_design(kind, *args)
        g[f"bq_{tag}_a"], g[f"bq_{tag}_b"], g[f"bq_{tag}_cd"] = a, b, np.array([c0, d0])
        g[f"bq_{tag}_y32"] = ref.biquad(x4k, a, b, c0, d0)
        g[f"bq_{tag}_y64"] = ref.biquad(x4k.astype(np.float64), a, b, c0, d0)

    # --- impulse -> filter -> 1024-pt fft magnitude chain -----------------------------------
    imp1k = np.zeros(1024)
    imp1k[512] = 1
    a, b = g["cheby_lp2_a"], g["cheby_lp2_b"]
    resp = ref.biquad(imp1k, a, b)
    g["chain_lp2_response"] = resp
    g["chain_lp2_fft"] = ref.fft_real(resp, 1024)
    a, b, c0, d0 = ref.biquad_design("peaking", 0.4, 4.0, 0.7)
    resp = ref.biquad(imp1k, a, b, c0, d0)
    g["chain_pk_response"] = resp
    g["chain_pk_fft"] = ref.fft_real(resp, 1024)

    # --- the band-pass retuned every sample: eval/signals/wind_generator.h:150-167 -------------------
    # Whistling wh0{400, 600, ...} and wh1{1000, 1000, ...} of eval/signals/main.cpp:33-34 at 44.1 kHz.  The wind
    # speed they follow comes out of choc's oscillator (absent here), so a 20 Hz raised sine stands in for it; the
    # centre frequency is computed in float as wind_generator.h:157-158 does, the bandwidth is its 60.f/sampleRate.
    # (No new random draws: everything above keeps its values.)
    sr44 = np.float32(44100)
    tt = np.arange(4096, dtype=np.float32)
    ws = (np.float32(0.5) + np.float32(0.5) * np.sin(np.float32(2 * np.pi * 20) * tt / sr44)).astype(np.float32)
    bw = float(np.float32(60) / sr44)
    g["swept_bw"] = np.array([bw])
    for tag, lower, upper in (("wh0", 400, 600), ("wh1", 1000, 1000)):
        fc = ((ws * np.float32(lower) + np.float32(upper)) / sr44).astype(np.float32)
        g[f"swept_{tag}_fc"] = fc
        st = np.zeros(4)
        g[f"swept_{tag}_y32"] = ref.biquad_swept(x4k, fc, bw, "bandpass", state=st)
        g[f"swept_{tag}_state"] = st.copy()
    g["swept_wh0_reject_y32"] = ref.biquad_swept(x4k, g["swept_wh0_fc"], bw, "bandreject")

    out = os.path.join(ROOT, "tests", "golden", "dsp_golden.npz")
    np.savez_compressed(out, **g)
    print(f"wrote {out}: {len(g)} arrays, {os.path.getsize(out)/1024:.1f} KiB")


if __name__ == "__main__":
    main()
