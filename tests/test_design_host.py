"""Host-side designers (dsp_amd/design.py) against the golden fixture: the tap / coefficient
arrays the kernels consume must be the reference's (dsp.h:47-79, recursive_filters.h:24-263)."""
import numpy as np
import pytest

from dsp_amd import design


def same(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def test_sinc(golden):
    assert same(design.sinc(0.14, 10), golden["sinc_0p14_10"])
    assert same(design.sinc(0.1, 511), golden["sinc_0p1_511"])
    assert same(design.sinc(0.14, 200), golden["sinc_0p14_200"])
    assert same(design.sinc(0.2, 64, "hamming", "inversion"), golden["sinc_0p2_64_hamming_inv"])
    assert same(design.sinc(0.2, 64, "blackman", "reversal"), golden["sinc_0p2_64_blackman_rev"])
    assert same(design.sinc(0.3, 31, "identity", "none"), golden["sinc_0p3_31_identity"])


@pytest.mark.parametrize("tag,fc,poles,hp", [("lp2", 0.1, 2, False), ("lp4", 0.1, 4, False), ("lp6", 0.1, 6, False),
                                             ("hp4", 0.1, 4, True), ("lp4_lo", 0.01, 4, False), ("lp8", 0.2, 8, False)])
def test_chebyshev(golden, tag, fc, poles, hp):
    a, b = design.chebyshev(fc, poles, hp)
    assert same(a, golden[f"cheby_{tag}_a"]) and same(b, golden[f"cheby_{tag}_b"])


def test_chebyshev_rejects_bad_arguments():
    for bad in ((0.1, 3), (0.1, 22), (0.6, 4), (-0.1, 4)):
        with pytest.raises(ValueError):
            design.chebyshev(*bad)


@pytest.mark.parametrize("tag,fn,args", [("bp", design.band_pass, (0.1, 0.01)), ("br", design.band_reject, (0.2, 0.02)),
                                         ("ls", design.low_shelf, (0.3, 6.0)), ("hs", design.high_shelf, (0.5, -3.0)),
                                         ("pk", design.peaking, (0.4, 4.0, 0.7))])
def test_biquads(golden, tag, fn, args):
    a, b, c0, d0 = fn(*args)
    assert same(a, golden[f"bq_{tag}_a"]) and same(b, golden[f"bq_{tag}_b"])
    assert same([c0, d0], golden[f"bq_{tag}_cd"])
