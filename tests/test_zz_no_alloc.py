"""glv_peak_normalize_noalloc - the entry point for callers that cannot allocate - through the
C ABI against glv_peak_normalize: the same bits, on two calls in a row, the second smaller (the
fixed device array still holds the first call's maxima), then on rows that end early, and the
size limit."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def test_peak_normalize_without_allocation_equals_the_plan_based_call():
    from tts_amd import _lib
    from tts_amd.vocoder import GriffinLimPlan
    lib = _lib.load()
    plan = GriffinLimPlan(1024, 256, 1024, 80, 32, 64)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(12)

    def both(y, lens):
        b, stride = y.shape
        want = plan.peak_normalize(torch.from_numpy(y.copy()).cuda(), lens).cpu().numpy()
        got = torch.from_numpy(y.copy()).cuda()
        n_dev = torch.tensor(lens, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        # free device memory as the driver sees it: a hipMalloc inside the library would show here
        # (torch's own counters only see torch's allocator)
        free_before, _ = torch.cuda.mem_get_info()
        _lib.check(lib.glv_peak_normalize_noalloc(ctypes.c_void_p(got.data_ptr()), ctypes.c_void_p(n_dev.data_ptr()),
                                                  b, stride, 1e-6, stream))
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info()
        assert free_after == free_before
        return got.cpu().numpy(), want

    # first call: 64 rows of large values; second call: fewer, shorter rows of small values -
    # stale maxima from the first call would divide them far too small
    y1 = (rng.standard_normal((64, 30000)) * 50.0).astype(np.float32)
    lens1 = [int(n) for n in rng.integers(1, 30001, 64)]
    lens1[0], lens1[1] = 30000, 1
    got, want = both(y1, lens1)
    assert np.array_equal(got, want)
    y2 = (rng.standard_normal((5, 777)) * 1e-3).astype(np.float32)
    lens2 = [777, 100, 0, 1, 776]
    got, want = both(y2, lens2)
    assert np.array_equal(got, want)
    for b, n in enumerate(lens2):
        if n:
            peak = float(np.max(np.abs(y2[b, :n])))
            assert np.max(np.abs(got[b, :n])) == pytest.approx(peak / (peak + 1e-6), rel=1e-6)
        assert np.array_equal(got[b, n:], y2[b, n:])        # past its length a row is untouched
    # sizes beyond the fixed array are refused, nothing is launched
    rc = lib.glv_peak_normalize_noalloc(ctypes.c_void_p(1), ctypes.c_void_p(1), 32769, 16, 1e-6, stream)
    assert rc == -1 and b"32768" in lib.glv_last_error()
    assert lib.glv_peak_normalize_noalloc(None, None, 1, 16, 1e-6, stream) == -1
    plan.close()


def test_peak_normalize_without_allocation_under_stream_capture():
    """The advertised use: the call recorded into a graph.  Capture runs in the mode that rejects
    every allocation and synchronisation made anywhere in the process while it lasts, so a
    hipMalloc / hipFree inside the entry point would fail the capture; the replayed graph must give
    the bits of the plan-based call.  (The address of the library's fixed array was looked up when
    the plan was created - the look-up is not a stream operation and must not fall into the capture.)"""
    from tts_amd import _lib
    from tts_amd.vocoder import GriffinLimPlan
    lib = _lib.load()
    plan = GriffinLimPlan(1024, 256, 1024, 80, 32, 8)
    rng = np.random.default_rng(13)
    y = (rng.standard_normal((8, 5000)) * 3.0).astype(np.float32)
    lens = [5000, 1, 4999, 2500, 17, 5000, 300, 4096]
    want = plan.peak_normalize(torch.from_numpy(y.copy()).cuda(), lens).cpu().numpy()
    got = torch.from_numpy(y.copy()).cuda()
    n_dev = torch.tensor(lens, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side, capture_error_mode="global"):
        rc = lib.glv_peak_normalize_noalloc(ctypes.c_void_p(got.data_ptr()), ctypes.c_void_p(n_dev.data_ptr()),
                                            8, 5000, 1e-6, ctypes.c_void_p(side.cuda_stream))
    assert rc == 0, lib.glv_last_error()
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), y)          # captured, not executed
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    plan.close()
