"""dsp_gpu_fft_batch_noalloc (the allocation-free entry point) against dsp_gpu_fft_batch, through the C ABI.

Same kernels, different scratch buffer (a fixed device array instead of the growing workspace): results must be
bit-identical.  Two calls in a row, the second on a smaller input (the shared array keeps the first call's contents),
plus the size limit and the scratch-free small sizes.  Collected last (file name) so a failure here hides nothing.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(L, capi, fn, x, log2n, batch, direction):
    import torch
    out = torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    capi.check(fn(x.data_ptr(), out.data_ptr(), log2n, batch, direction, *( (capi.MEM_DEVICE,) if fn is L.dsp_gpu_fft_batch else () ), stream))
    torch.cuda.synchronize()
    return out


def test_noalloc_matches_fft_batch_twice(dev, orc):
    import torch
    from dsp_amd import capi
    L = capi.lib()
    g = torch.Generator(device=dev).manual_seed(77)
    for log2n, batch in ((17, 8), (16, 3)):  # second call smaller than the first (both sizes use the scratch array)
        n = 1 << log2n
        x = torch.view_as_complex(torch.randn(batch, n, 2, device=dev, generator=g))
        for direction in (capi.FORWARD, capi.INVERSE):
            got = _run(L, capi, L.dsp_gpu_fft_batch_noalloc, x, log2n, batch, direction)
            want = _run(L, capi, L.dsp_gpu_fft_batch, x, log2n, batch, direction)
            assert torch.equal(got.view(torch.float32), want.view(torch.float32)), (log2n, batch, direction)
        # and against the oracle (the reference's double result), within the FFT budget
        ref = orc.fft0(x[0].cpu().numpy().astype(np.complex128), False)
        got0 = _run(L, capi, L.dsp_gpu_fft_batch_noalloc, x, log2n, batch, capi.FORWARD)[0].cpu().numpy()
        assert np.linalg.norm(got0 - ref) / np.linalg.norm(ref) <= 1e-5 * log2n


def test_noalloc_three_pass_and_in_place(dev):
    import torch
    from dsp_amd import capi
    L = capi.lib()
    g = torch.Generator(device=dev).manual_seed(78)
    for log2n, batch in ((21, 1), (18, 5), (17, 16)):  # 2^21 points is the limit exactly
        n = 1 << log2n
        x = torch.view_as_complex(torch.randn(batch, n, 2, device=dev, generator=g))
        want = _run(L, capi, L.dsp_gpu_fft_batch, x, log2n, batch, capi.FORWARD)
        got = _run(L, capi, L.dsp_gpu_fft_batch_noalloc, x, log2n, batch, capi.FORWARD)
        assert torch.equal(got.view(torch.float32), want.view(torch.float32))
        buf = x.clone()
        stream = torch.cuda.current_stream().cuda_stream
        capi.check(L.dsp_gpu_fft_batch_noalloc(buf.data_ptr(), buf.data_ptr(), log2n, batch, capi.FORWARD, stream))
        torch.cuda.synchronize()
        assert torch.equal(buf.view(torch.float32), want.view(torch.float32))


def test_noalloc_small_sizes_and_limits(dev):
    import torch
    from dsp_amd import capi
    L = capi.lib()
    g = torch.Generator(device=dev).manual_seed(79)
    for log2n, batch in ((12, 1000), (14, 300), (15, 100), (3, 17)):  # single-pass sizes: no scratch, any batch
        x = torch.view_as_complex(torch.randn(batch, 1 << log2n, 2, device=dev, generator=g))
        want = _run(L, capi, L.dsp_gpu_fft_batch, x, log2n, batch, capi.INVERSE)
        got = _run(L, capi, L.dsp_gpu_fft_batch_noalloc, x, log2n, batch, capi.INVERSE)
        assert torch.equal(got.view(torch.float32), want.view(torch.float32))
    x = torch.zeros(1 << 22, dtype=torch.complex64, device=dev)
    y = torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    assert L.dsp_gpu_fft_batch_noalloc(x.data_ptr(), y.data_ptr(), 16, 33, capi.FORWARD, stream) == capi.EINVAL  # 33 * 2^16 > 2^21
    assert b"exceed" in L.dsp_gpu_last_error_string()
    assert L.dsp_gpu_fft_batch_noalloc(x.data_ptr(), y.data_ptr(), 22, 1, capi.FORWARD, stream) == capi.EINVAL
    assert L.dsp_gpu_fft_batch_noalloc(x.data_ptr(), y.data_ptr(), 16, 32, capi.FORWARD, stream) == capi.OK  # exactly the limit
    torch.cuda.synchronize()
