"""Tensor-level wrappers of the bf16 configuration's kernels (include/sar_hip.h, "CN8" section; csrc/cn8.h).

A CN8 activation is a torch.bfloat16 tensor of shape (G, ld, 8): G = ceil(C/8) planes of ld 16-byte units, unit
(g, col) = channels 8g..8g+7 of column col = (b*T + t)*V + v.  torch is used for device memory and the stream only.
"""
import ctypes as C

import os

import torch

from . import _lib as L
from . import profiler
from ._lib import ConvDesc, WgradDesc, check, ptr, stream_ptr

_MFMA_GATHER = os.environ.get("SAR_CN8_MFMA_GATHER", "1") == "1"   # A/B switch: 0 keeps the vector-ALU gather of the dense adjacency slice


def empty(channels, n, device):
    return torch.empty(((channels + 7) // 8, n, 8), dtype=torch.bfloat16, device=device)


def _cn8(t):
    assert t is None or (t.dtype == torch.bfloat16 and t.is_cuda and t.is_contiguous() and t.dim() == 3 and t.shape[2] == 8), \
        "need a contiguous cuda CN8 tensor (G, ld, 8) of bfloat16"
    return t


def _f32(t):
    assert t is None or (t.dtype == torch.float32 and t.is_cuda and t.is_contiguous()), "need contiguous cuda float32"
    return t


def from_cn(x):
    """fp32 CN matrix [C][n] -> CN8 (rounded to bfloat16, nearest even)"""
    Cc, n = x.shape
    out = empty(Cc, n, x.device)
    check(L.load().sar_cn_to_cn8(ptr(_f32(x)), x.stride(0), ptr(out), n, Cc, n, stream_ptr()), "sar_cn_to_cn8")
    return out


def to_cn(x8, channels):
    """CN8 -> fp32 CN matrix [channels][n]"""
    n = x8.shape[1]
    out = torch.empty((channels, n), dtype=torch.float32, device=x8.device)
    check(L.load().sar_cn8_to_cn(ptr(_cn8(x8)), n, ptr(out), n, channels, n, stream_ptr()), "sar_cn8_to_cn")
    return out


def conv_gemm(mode, src, out, packed, *, B, V, T_src, T_out, Kc, M, taps, stride=1, pad=0, transposed=False, bias=None,
              pro=None, pro_relu=False, tables=None, epi=L.SAR_EPI_NONE, aux=None, aux_affine=None, aux_mean=None, aux2=None,
              aux_mask=None):
    """sar_conv_gemm_cn8: src / out / aux CN8, `packed` the bf16 weight image (ops.PackedWeights.image).  Returns
    (partials, nparts) when the epilogue reduces.  SAR_EPI_ADD_GATE (graph data gradient): out = gate(acc + aux) with the gate
    bytes `aux_mask`, partial sums (sum out, sum out (aux2 - aux_mean))."""
    lib = L.load()
    d = ConvDesc()
    d.mode, d.transposed, d.B, d.V = mode, int(transposed), B, V
    d.T_src, d.T_out, d.Kc, d.M = T_src, T_out, Kc, M
    d.taps, d.stride, d.pad, d.pro_relu, d.epi = taps, stride, pad, int(pro_relu), epi
    _cn8(src), _cn8(out), _cn8(aux)
    assert src.shape[0] == (Kc + 7) // 8 and out.shape[0] == (M + 7) // 8
    d.src, d.ld_src = ptr(src), src.shape[1]
    d.out, d.ld_out = ptr(out), out.shape[1]
    d.bias = ptr(_f32(bias))
    if pro is not None:
        d.pro_scale, d.pro_shift = ptr(_f32(pro[0])), ptr(_f32(pro[1]))
    if tables is not None:
        d.g_idx, d.g_wt, d.g_colsum = ptr(tables.idx), ptr(tables.wt), ptr(tables.colsum)
        d.g_flags = getattr(tables, "g_flags", 0) if _MFMA_GATHER else 0
        for i in range(3):
            d.nz[i] = tables.nz[i]
    if aux is not None:
        assert aux.shape[0] == out.shape[0]
        d.aux, d.ld_aux = ptr(aux), aux.shape[1]
    if aux_affine is not None:
        d.aux_scale, d.aux_shift = ptr(_f32(aux_affine[0])), ptr(_f32(aux_affine[1]))
    d.aux_mean = ptr(_f32(aux_mean))
    if epi == L.SAR_EPI_ADD_GATE:
        assert aux2 is not None and aux_mask is not None and aux_mean is not None and aux2.shape[0] == out.shape[0]
        assert aux_mask.dtype == torch.uint8 and aux_mask.shape == (out.shape[0], aux2.shape[1]) and aux_mask.is_contiguous()
        _cn8(aux2)
        d.aux2, d.ld_aux2, d.aux_mask = ptr(aux2), aux2.shape[1], ptr(aux_mask)
    partials, nparts = None, 0
    if epi in (L.SAR_EPI_STATS, L.SAR_EPI_MASK, L.SAR_EPI_ADD_GATE):
        nparts = lib.sar_conv_gemm_cn8_nparts(C.byref(d))
        if nparts <= 0:
            check(nparts or -1, "sar_conv_gemm_cn8_nparts")
        partials = torch.empty((M, nparts, 2), dtype=torch.float32, device=src.device)
        d.partials = ptr(partials)
    n_conv = B * (T_src if transposed else T_out) * V
    tag = ("gemm_graph" if mode == L.SAR_CONV_GRAPH else ("gemm_temporal%d%s" % (taps, "_dgrad" if transposed else ""))) + "_cn8"
    with profiler.region(tag, 2.0 * M * Kc * taps * n_conv, 2.0 * (Kc * B * T_src * V + M * B * T_out * V)):
        check(lib.sar_conv_gemm_cn8(C.byref(d), ptr(packed), stream_ptr()), "sar_conv_gemm_cn8")
    return (partials, nparts) if partials is not None else None


_WGRAD_SLOTS = int(__import__("os").environ.get("SAR_WGRAD8_SLOTS", "512"))


def conv_wgrad(mode, src, dout, dW_out, *, B, V, T_src, T_out, Kc, M, taps, stride=1, pad=0, pro=None, pro_relu=False,
               tables=None, w_stride_tap, w_stride_c, wsize, bsize, nsplit=None, slabs=None):
    """sar_conv_wgrad_cn8 + sar_slab_reduce_f32: dW (and dbias right behind it) -> dW_out[0 : wsize + bsize] (flat fp32).
    slabs (ops.SlabBatch): the slabs are summed by the batch's next flush() instead of by a launch of their own."""
    lib = L.load()
    d = WgradDesc()
    d.mode, d.B, d.V, d.T_src, d.T_out, d.Kc, d.M = mode, B, V, T_src, T_out, Kc, M
    d.taps, d.stride, d.pad, d.pro_relu = taps, stride, pad, int(pro_relu)
    ft = lib.sar_conv_wgrad_cn8_tile_frames(mode)
    ntiles = B * ((T_out + ft - 1) // ft)
    cb = 32 if (mode == L.SAR_CONV_TEMPORAL and taps == 9) else 64
    blocks = ((M + 63) // 64) * ((Kc + cb - 1) // cb)
    if nsplit is None:           # two resident workgroups per CU (SAR_WGRAD8_SLOTS: sweep knob, tools)
        nsplit = max(1, min(ntiles, (_WGRAD_SLOTS + blocks - 1) // blocks))
    d.nsplit = nsplit
    _cn8(src), _cn8(dout)
    d.src, d.ld_src, d.dout, d.ld_dout = ptr(src), src.shape[1], ptr(dout), dout.shape[1]
    if pro is not None:
        d.pro_scale, d.pro_shift = ptr(_f32(pro[0])), ptr(_f32(pro[1]))
    ident = 0
    if tables is not None:
        d.g_idx, d.g_wt, d.g_colsum = ptr(tables.idx), ptr(tables.wt), ptr(tables.colsum)
        d.g_flags = getattr(tables, "g_flags", 0) if _MFMA_GATHER else 0
        for i in range(3):
            d.nz[i] = tables.nz[i]
        ident = int(getattr(tables, "slice0_identity", False))
    d.w_stride_tap, d.w_stride_c, d.wsize, d.bsize = w_stride_tap, w_stride_c, wsize, bsize
    assert dW_out.numel() >= wsize + bsize and dW_out.is_contiguous()
    slab = (slabs.slab(dW_out, nsplit, wsize + bsize) if slabs is not None
            else torch.empty((nsplit, wsize + bsize), dtype=torch.float32, device=src.device))
    d.slab = ptr(slab)
    tag = ("wgrad_graph" if mode == L.SAR_CONV_GRAPH else "wgrad_temporal%d" % taps) + "_cn8"
    with profiler.region(tag, 2.0 * M * Kc * taps * B * T_out * V, 2.0 * (Kc * B * T_src * V + M * B * T_out * V)):
        check(lib.sar_conv_wgrad_cn8(C.byref(d), ident, stream_ptr()), "sar_conv_wgrad_cn8")
    if slabs is not None:
        slabs.add(slab, nsplit, wsize + bsize, dW_out)
        return
    check(lib.sar_slab_reduce_f32(ptr(slab), nsplit, wsize + bsize, wsize + bsize, ptr(dW_out), stream_ptr()),
          "sar_slab_reduce_f32")


def relu_mask(channels, n, device):
    """one byte per unit: the ReLU mask of a block tail's output (sar_bn_add_relu_fwd_mask_cn8)"""
    return torch.empty(((channels + 7) // 8, n), dtype=torch.uint8, device=device)


def bn_add_relu_fwd(u, sc, sh, res_kind, r, rsc, rsh, y, channels, mask=None):
    if mask is not None:
        check(L.load().sar_bn_add_relu_fwd_mask_cn8(ptr(_cn8(u)), ptr(sc), ptr(sh), res_kind, ptr(_cn8(r)), ptr(rsc), ptr(rsh),
                                                    ptr(_cn8(y)), ptr(mask), channels, u.shape[1], u.shape[1], stream_ptr()),
              "sar_bn_add_relu_fwd_mask_cn8")
        return
    check(L.load().sar_bn_add_relu_fwd_cn8(ptr(_cn8(u)), ptr(sc), ptr(sh), res_kind, ptr(_cn8(r)), ptr(rsc), ptr(rsh), ptr(_cn8(y)),
                                           channels, u.shape[1], u.shape[1], stream_ptr()), "sar_bn_add_relu_fwd_cn8")


_REDUCE_CHUNK = int(__import__("os").environ.get("SAR_BWD_REDUCE_CHUNK8", "8192"))     # units per workgroup of the BN-backward reduction


def bn_add_relu_bwd_reduce(dy, y, u, r, channels, mu=None, mr=None, tail=None, mask=None):
    """mask (relu_mask written by bn_add_relu_fwd): read instead of y"""
    n = u.shape[1]
    nparts = max(1, min(4096, (n + _REDUCE_CHUNK - 1) // _REDUCE_CHUNK))
    partials = torch.empty((channels, nparts, 4), dtype=torch.float32, device=u.device)
    if mask is not None:
        assert tail is None
        check(L.load().sar_bn_add_relu_bwd_reduce_mask_cn8(ptr(_cn8(dy)), ptr(mask), ptr(_cn8(u)), ptr(_cn8(r)), ptr(mu), ptr(mr),
                                                           ptr(partials), nparts, channels, n, n, stream_ptr()),
              "sar_bn_add_relu_bwd_reduce_mask_cn8")
        return partials, nparts
    if tail is not None:      # ops.make_bn_tail: the last workgroup of every plane finalises its channels
        import ctypes
        check(L.load().sar_bn_add_relu_bwd_reduce_tail_cn8(ptr(_cn8(dy)), ptr(_cn8(y)), ptr(_cn8(u)), ptr(_cn8(r)), ptr(mu), ptr(mr),
                                                           ptr(partials), nparts, channels, n, n, ctypes.byref(tail), stream_ptr()),
              "sar_bn_add_relu_bwd_reduce_tail_cn8")
        return partials, nparts
    check(L.load().sar_bn_add_relu_bwd_reduce_cn8(ptr(_cn8(dy)), ptr(_cn8(y)), ptr(_cn8(u)), ptr(_cn8(r)), ptr(mu), ptr(mr),
                                                  ptr(partials), nparts, channels, n, n, stream_ptr()),
          "sar_bn_add_relu_bwd_reduce_cn8")
    return partials, nparts


def bn_add_relu_bwd_apply(dy, y, u, r, k, rk, du, dr, dz_out, channels, mask=None):
    rk = rk or (None, None, None)
    n = u.shape[1]
    if mask is not None:
        check(L.load().sar_bn_add_relu_bwd_apply_mask_cn8(ptr(_cn8(dy)), ptr(mask), ptr(_cn8(u)), ptr(_cn8(r)), ptr(k[0]), ptr(k[1]),
                                                          ptr(k[2]), ptr(rk[0]), ptr(rk[1]), ptr(rk[2]), ptr(_cn8(du)), ptr(_cn8(dr)),
                                                          ptr(_cn8(dz_out)), channels, n, n, stream_ptr()),
              "sar_bn_add_relu_bwd_apply_mask_cn8")
        return
    check(L.load().sar_bn_add_relu_bwd_apply_cn8(ptr(_cn8(dy)), ptr(_cn8(y)), ptr(_cn8(u)), ptr(_cn8(r)), ptr(k[0]), ptr(k[1]),
                                                 ptr(k[2]), ptr(rk[0]), ptr(rk[1]), ptr(rk[2]), ptr(_cn8(du)), ptr(_cn8(dr)),
                                                 ptr(_cn8(dz_out)), channels, n, n, stream_ptr()), "sar_bn_add_relu_bwd_apply_cn8")


def affine2(a, b, k, out, channels):
    n = a.shape[1]
    check(L.load().sar_affine2_cn8(ptr(_cn8(a)), ptr(_cn8(b)), ptr(k[0]), ptr(k[1]), ptr(k[2]), ptr(_cn8(out)), channels, n, n,
                                   stream_ptr()), "sar_affine2_cn8")


def data_bn_apply(x, bone_parent, scale, shift, out, motion=False):
    N, C_, T, V, M = x.shape
    check(L.load().sar_data_bn_apply_cn8(ptr(x), N, C_, T, V, M, ptr(bone_parent), int(motion), ptr(scale), ptr(shift),
                                         ptr(_cn8(out)), out.shape[1], stream_ptr()), "sar_data_bn_apply_cn8")


def data_bn_bwd_reduce(x, bone_parent, dy, mean, partials, motion=False):
    N, C_, T, V, M = x.shape
    check(L.load().sar_data_bn_bwd_reduce_cn8(ptr(x), N, C_, T, V, M, ptr(bone_parent), int(motion), ptr(_cn8(dy)), dy.shape[1],
                                              ptr(mean), ptr(partials), stream_ptr()), "sar_data_bn_bwd_reduce_cn8")


def pool_fwd(y, channels, B, TV, Mp, feat):
    check(L.load().sar_pool_fwd_cn8(ptr(_cn8(y)), y.shape[1], channels, B, TV, Mp, ptr(feat), stream_ptr()), "sar_pool_fwd_cn8")


def pool_bwd(dfeat, channels, B, TV, Mp, dy):
    check(L.load().sar_pool_bwd_cn8(ptr(dfeat), dy.shape[1], channels, B, TV, Mp, ptr(_cn8(dy)), stream_ptr()), "sar_pool_bwd_cn8")
