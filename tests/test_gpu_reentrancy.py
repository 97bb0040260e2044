"""SURVEY 8(b) boundary contract: "no global mutable state ... re-entrant from multiple host threads on distinct streams"
(VERDICT r02 next #8).  The library owns no streams or events: launches that fan out take them from a caller-owned
sar_context (include/sar_hip.h).  Here two host threads, each with its own torch stream and its own context, drive the
C ABI at the same time (ctypes releases the GIL inside a call): the 3x3 / stride-2 data gradient (four parity-class
launches over the context's side streams), a forward convolution with BatchNorm partial sums, a weight gradient and an
ST-GCN temporal convolution.  Every result must equal, BIT FOR BIT, the result of the same calls made serially."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from sar_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randn(shape, generator=g, device=dev) * scale


def _work(dev, seed, ctx, reps):
    """one thread's call sequence on the CURRENT stream; returns its outputs"""
    from sar_amd import ops, _lib as L
    B, cin, cout, H = 4, 64, 128, 32
    Ho = H // 2
    X = _rand((cin, B * H * H), dev, seed)
    dO = _rand((cout, B * Ho * Ho), dev, seed + 1)
    Wf = _rand((9, cin, cout), dev, seed + 2, 0.05)
    Wb = Wf.permute(0, 2, 1).contiguous()
    geo = dict(B=B, Kc=cin, M=cout, H_src=H, W_src=H, H_out=Ho, W_out=Ho, KH=3, KW=3, stride=2, pad=1)
    dgeo = dict(B=B, Kc=cout, M=cin, H_src=Ho, W_src=Ho, H_out=H, W_out=H, KH=3, KW=3, stride=2, pad=1)
    # ST-GCN temporal conv on the same stream
    f, T, V, Bs = 64, 40, 25, 4
    G = _rand((f, Bs * T * V), dev, seed + 3)
    Wt, bt = _rand((9, 1, f, f), dev, seed + 4, 0.05), _rand((f,), dev, seed + 5, 0.1)
    outs = []
    for _ in range(reps):
        out = torch.empty((cout, B * Ho * Ho), device=dev)
        r = ops.conv2d_gemm(X, out, Wf, cin * cout, cout, epi=L.SAR_EPI_STATS, ctx=ctx, **geo)
        dx = torch.empty((cin, B * H * H), device=dev)
        ops.conv2d_gemm(dO, dx, Wb, cout * cin, cin, epi=L.SAR_EPI_ADD, aux=X, transposed=True, ctx=ctx, **dgeo)
        gw = torch.empty(9 * cin * cout, device=dev)
        ops.conv2d_wgrad(X, dO, gw, **geo)
        u = torch.empty((f, Bs * T * V), device=dev)
        ops.conv_gemm(L.SAR_CONV_TEMPORAL, G, u, Wt, f * f, f, B=Bs, V=V, T_src=T, T_out=T, Kc=f, M=f, taps=9, stride=1, pad=4,
                      bias=bt, epi=L.SAR_EPI_NONE)
        outs = [out, r[0], dx, gw, u]
    torch.cuda.current_stream().synchronize()
    return [o.clone() for o in outs]


def test_two_host_threads_on_distinct_streams_are_reentrant(dev):
    from sar_amd import _lib as L
    seeds = (10, 20)
    serial = [_work(dev, s, L.Context(), 1) for s in seeds]
    results, errors = [None, None], []

    def run(i):
        try:
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(device=dev)
            ctx = L.Context()                       # this thread's own side streams
            with torch.cuda.stream(st):
                results[i] = _work(dev, seeds[i], ctx, 25)
        except Exception as e:                      # surfaced below (a thread's exception is otherwise lost)
            errors.append(e)

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for i in range(2):
        for a, b in zip(results[i], serial[i]):
            assert torch.equal(a, b)


def test_without_a_context_the_fan_out_runs_on_the_callers_stream(dev):
    """ctx = NULL is the documented serial fallback: same bits."""
    from sar_amd import _lib as L
    a = _work(dev, 30, None, 1)
    b = _work(dev, 30, L.Context(), 1)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_library_exports_no_debug_entry_points():
    import ctypes
    from sar_amd import _lib as L
    lib = ctypes.CDLL(L.LIB_PATH)
    assert not hasattr(lib, "sar_debug_poison_lds") and not hasattr(lib, "sar_debug_occupancy")


def test_stream_handle_follows_the_current_torch_stream():
    """sar_amd._lib.stream_ptr() asks torch's C module for the raw stream (no Stream object per launch): it must name the stream
    torch considers current -- default, inside a stream context, on a side stream of an engine -- and follow set_device."""
    import torch
    from sar_amd import _lib
    torch.cuda.set_device(0)
    assert _lib.stream_ptr() == torch.cuda.current_stream().cuda_stream
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(s1):
        assert _lib.stream_ptr() == s1.cuda_stream
        with torch.cuda.stream(s2):
            assert _lib.stream_ptr() == s2.cuda_stream != s1.cuda_stream
        assert _lib.stream_ptr() == s1.cuda_stream
    assert _lib.stream_ptr() == torch.cuda.current_stream().cuda_stream
