"""Thin tensor-level wrappers over the C ABI (include/sar_hip.h).  torch is used only for device
memory and the current HIP stream; all arithmetic happens in libsar_hip.so.

Activations use the CN layout: a 2-D float32 tensor [C][B*T*V] (see include/sar_hip.h).
"""
import ctypes as C
import os

import torch

from . import _lib as L
from . import profiler
from ._lib import ConvDesc, WgradDesc, check, ptr, stream_ptr


_PROFILE_SHAPES = __import__("os").environ.get("SAR_PROFILE_SHAPES", "0") == "1"


def _f32(t):
    assert t is None or (t.dtype == torch.float32 and t.is_cuda and t.is_contiguous()), "need contiguous cuda float32"
    return t


class GraphTables:
    """Device copies of the gather lists (graph_tables.gather_lists)."""

    def __init__(self, A, device, transpose=False):
        from .graph_tables import gather_lists
        idx, wt, nz, colsum = gather_lists(A, transpose)
        self.idx = torch.from_numpy(idx).to(device)
        self.wt = torch.from_numpy(wt).to(device)
        self.colsum = torch.from_numpy(colsum).to(device)
        self.nz = nz
        self.K, self.V = idx.shape[0], idx.shape[1]
        # slice 0 is the identity (the 'spatial' strategy's self-links, graph/tools.py:22-30): gather list {(v, 1.0)}
        import numpy as np
        self.slice0_identity = bool(nz[0] == 1 and np.array_equal(idx[0, :, 0], np.arange(self.V)) and np.all(wt[0, :, 0] == 1.0))
        # every gather weight exactly representable in bfloat16 (NTU: 0.25 / 0.5 / 1): the CN8 kernels may then apply a dense
        # slice on the matrix cores (SAR_GRAPH_WT_BF16_EXACT, include/sar_hip.h)
        w32 = torch.from_numpy(np.ascontiguousarray(wt, dtype=np.float32))
        self.g_flags = L.SAR_GRAPH_WT_BF16_EXACT if bool(torch.equal(w32.to(torch.bfloat16).to(torch.float32), w32)) else 0
        # gather lists that are neither {one entry of weight 1} nor empty: the CN8 graph convolution builds only those and reads
        # every other operand straight from the raw tile (SAR_GRAPH_FEW_DENSE, csrc/conv_graph_cn8.hip)
        wnp = np.asarray(wt, dtype=np.float32)
        cnt = (wnp != 0).sum(axis=2)
        first = np.take_along_axis(wnp, np.argmax(wnp != 0, axis=2)[..., None], axis=2)[..., 0]
        self.n_dense_lists = int(((cnt > 1) | ((cnt == 1) & (first != 1.0))).sum())
        if self.slice0_identity:
            self.g_flags |= L.SAR_GRAPH_SLICE0_IDENTITY
        # (<= 12: what the read-gather kernels hold as virtual joints of a V = 25 tile -- conv_graph_cn8.hip needs 2 FT nv <= 256 --;
        # denser tables keep the unit-builder / fp32 kernels and the unfused block tail)
        if self.n_dense_lists <= 12:
            self.g_flags |= L.SAR_GRAPH_FEW_DENSE | (self.n_dense_lists << L.SAR_GRAPH_FEW_DENSE_SHIFT)


class PackedWeights:
    """bf16 operand images of many weight tensors, refreshed from the flat parameter buffer by ONE launch
    (sar_pack_weights_bf16_batch).  add() registers element (tap, c, m) at base[src_off + tap*st + c*sc + m*sm]; a data
    gradient registers the same tensor with the roles of c and m exchanged (no transposed copy)."""

    ITEM = [("src_off", "<i8"), ("st", "<i8"), ("sc", "<i8"), ("sm", "<i8"), ("dst_unit", "<i8"),
            ("taps", "<i4"), ("Kc", "<i4"), ("M", "<i4"), ("G", "<i4")]

    def __init__(self):
        self.items, self.index, self.units = [], {}, 0

    def add(self, key, src_off, st, sc, sm, taps, Kc, M):
        G = 2 * ((Kc + 15) // 16)
        n = taps * G * M
        self.index[key] = (self.units, n)
        self.items.append((src_off, st, sc, sm, self.units, taps, Kc, M, G))
        self.units += n

    def finalize(self, device):
        import numpy as np
        tab = np.array(self.items, dtype=np.dtype(self.ITEM, align=True))
        assert tab.dtype.itemsize == 56          # sizeof(sar_pack_item)
        self.table = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(device)
        self.max_units = max(it[5] * it[8] * it[7] for it in self.items)
        self.buf = torch.empty(self.units * 16, dtype=torch.uint8, device=device)

    def refresh(self, flat):
        check(L.load().sar_pack_weights_bf16_batch(ptr(flat), ptr(self.table), len(self.items), self.max_units, ptr(self.buf),
                                                   stream_ptr()), "sar_pack_weights_bf16_batch")

    def image(self, key):
        off, n = self.index[key]
        return self.buf[off * 16:(off + n) * 16]


class PackedSplitWeights(PackedWeights):
    """Term images of many weight tensors for the split-arithmetic kernels (sar_conv_gemm_split, include/sar_hip.h): the same
    items as PackedWeights with channel groups of 8 and `terms` images per item, refreshed by ONE call per step; the fp16
    arithmetics also get every item's amax (the bits of a float, device memory) = its `w_bound`."""

    TERMS = {"bf16x1": 1, "bf16x3": 2, "bf16x6": 3, "bf16x9": 3, "f16x3": 2, "f16x3s": 2, "f16x3a": 3}

    def __init__(self, arith):
        super().__init__()
        self.arith, self.terms, self.item_of = arith, self.TERMS[arith], {}

    def add(self, key, src_off, st, sc, sm, taps, Kc, M):
        G = (Kc + 7) // 8
        n = self.terms * taps * G * M
        self.index[key] = (self.units, n)
        self.item_of[key] = len(self.items)
        self.items.append((src_off, st, sc, sm, self.units, taps, Kc, M, G))
        self.units += n

    def finalize(self, device):
        super().finalize(device)
        self.amax = torch.zeros(len(self.items), dtype=torch.int32, device=device)

    def refresh(self, flat):
        check(L.load().sar_pack_weights_split_batch(ptr(flat), ptr(self.table), len(self.items), self.max_units,
                                                    L.SAR_SPLIT[self.arith], ptr(self.buf), ptr(self.amax), stream_ptr()),
              "sar_pack_weights_split_batch")

    def bound(self, key):
        i = self.item_of[key]
        return self.amax[i:i + 1]


_WGRADS_SLOTS = int(os.environ.get("SAR_WGRAD_SPLIT_SLOTS", "512"))   # workgroups of a split weight-gradient launch (default: two per CU)
_side_streams = {}
GRAPH_ONE_TILE_WG = False     # tests / A-B: the round-5 graph kernel of the split arithmetic (include/sar_hip.h: SAR_GRAPH_ONE_TILE_WG)
SIDE_CU_MASK_DEFAULT = "off"


def shared_side_stream(device, priority=0):
    """ONE weight-gradient stream per (device, host thread, priority) for every engine of the process.  HIP maps streams onto a few
    hardware queues round-robin in creation order: engines that each created their own stream (a bench process builds seven, one after
    the other) sooner or later got one that shares its hardware queue with the main stream -- the two-stream overlap was gone and a
    4.7 ms Path B step took 5.3 (found as a leg that was slower inside the default bench line than alone)."""
    import threading
    key = (torch.device(device).index or 0, threading.get_ident(), int(priority))
    st = _side_streams.get(key)
    if st is None:
        mask = side_stream_cu_mask(device)
        if mask is None:
            st = torch.cuda.Stream(device=device, priority=int(priority))
        else:
            # a stream that leaves some CUs to the main chain (sar_stream_create_cu_mask), wrapped for torch's stream API
            import ctypes as _C
            words = (_C.c_uint32 * len(mask))(*mask)
            h = _C.c_void_p()
            with torch.cuda.device(device):
                check(L.load().sar_stream_create_cu_mask(words, len(mask), _C.byref(h)), "sar_stream_create_cu_mask")
            st = torch.cuda.ExternalStream(h.value, device=device)
        _side_streams[key] = st
    return st


def shared_aux_stream(device, tag="aux"):
    """ONE further stream per (device, host thread, tag) for the branch launches an engine forks off its main chain (the resnet's
    down-sampling branch and weight images: "aux"; the radar front-end of a resident batch: "front"; SAR_PATHB_DS_STREAM): shared for
    the reason shared_side_stream gives"""
    import threading
    key = (torch.device(device).index or 0, threading.get_ident(), tag)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


def side_stream_cu_mask(device):
    """the CU mask of the weight-gradient stream as a list of 32-bit words, or None = every CU.  SAR_SIDE_CU_MASK:
    'off' | 'skip:<n>' (every n-th CU left to the main chain) | 'first:<k>' / 'last:<k>' (k contiguous CUs left out) | hex words"""
    spec = os.environ.get("SAR_SIDE_CU_MASK", SIDE_CU_MASK_DEFAULT)
    if not spec or spec == "off":
        return None
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    nw = (ncu + 31) // 32
    bits = [1] * ncu
    kind, _, arg = spec.partition(":")
    if kind == "skip":
        n = int(arg)
        for i in range(ncu):
            if i % n == n - 1:
                bits[i] = 0
    elif kind == "first":
        for i in range(int(arg)):
            bits[i] = 0
    elif kind == "last":
        for i in range(int(arg)):
            bits[ncu - 1 - i] = 0
    else:
        return [int(w, 16) for w in spec.split(",")]
    words = [0] * nw
    for i, b in enumerate(bits):
        if b:
            words[i // 32] |= 1 << (i % 32)
    return words


def amax(x, cell):
    """cell (1-element int32 view, zeroed by the caller) = max(cell, bits of max |x|) -- sar_amax_f32, no host sync"""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1      # rows may be strided (ld >= n)
    check(L.load().sar_amax_f32(ptr(x), x.shape[0], x.shape[1], x.stride(0), ptr(cell), stream_ptr()), "sar_amax_f32")


def bn_bound(gamma, beta, count, cell):
    """Samuelson bound of a train-mode BatchNorm output (sar_bn_bound_f32)"""
    check(L.load().sar_bn_bound_f32(ptr(_f32(gamma)), ptr(_f32(beta)), gamma.numel(), float(count), ptr(cell), stream_ptr()),
          "sar_bn_bound_f32")


def affine_bound(scale, shift, src_cell, cell):
    check(L.load().sar_affine_bound_f32(ptr(_f32(scale)), ptr(_f32(shift)), scale.numel(), ptr(src_cell), ptr(cell), stream_ptr()),
          "sar_affine_bound_f32")


# the arithmetic a conv_gemm / conv_wgrad call WITHOUT an explicit `split` argument takes (None = the fp32 kernels): the kernel
# parity suites run once per value (tests/conftest.py)
DEFAULT_SPLIT = None


TAP1_SPLIT = __import__("os").environ.get("SAR_TAP1_SPLIT", "1") == "1"      # the 1-tap temporal operator on conv_tap1_split_kernel (A/B switch)


def split_applicable(mode, V, Kc, M, taps, stride, tables=None, pro=None, transposed=False, pad=0):
    """shapes csrc/conv_gemm_split.hip is built for (the others stay on the fp32 kernels)"""
    if mode == L.SAR_CONV_GRAPH:       # read-gather graph kernel: few non-trivial gather lists, no folded prologue, 16-channel stages
        return (taps == 3 and V == 25 and 16 <= Kc <= 256 and Kc % 16 == 0 and M % 8 == 0 and pro is None and tables is not None
                and bool(tables.g_flags & L.SAR_GRAPH_FEW_DENSE) and tables.n_dense_lists <= 16)
    if mode == L.SAR_CONV_TEMPORAL and taps == 1:      # 1x1 (residual) convolution, forward form: conv_tap1_split_kernel (bf16x6 / f16x3a)
        return (TAP1_SPLIT and V == 25 and Kc >= 16 and M % 8 == 0 and stride in (1, 2) and pad == 0 and pro is None
                and not transposed)
    return mode == L.SAR_CONV_TEMPORAL and taps == 9 and V == 25 and 8 <= Kc <= 256 and M % 8 == 0 and stride in (1, 2)


def _pack_split_single(W, w_stride_tap, w_stride_c, taps, Kc, M, arith):
    """one tensor, packed on the spot (kernel tests / probes; the engines keep a PackedSplitWeights): (image, w_bound)"""
    pk = PackedSplitWeights(arith)
    pk.add("w", 0, w_stride_tap, w_stride_c, 1, taps, Kc, M)
    pk.finalize(W.device)
    pk.refresh(W)
    return pk.image("w"), pk.bound("w")


def _src_bound_single(src, pro):
    """the bound of pro(src) for a stand-alone call: amax of the tensor, through the folded affine when there is one"""
    cells = torch.zeros(2, dtype=torch.int32, device=src.device)
    amax(src, cells[0:1])
    if pro is None:
        return cells[0:1]
    affine_bound(pro[0], pro[1], cells[0:1], cells[1:2])
    return cells[1:2]


def conv_gemm(mode, src, out, W, w_stride_tap, w_stride_c, *, B, V, T_src, T_out, Kc, M, taps, stride=1, pad=0,
              transposed=False, bias=None, pro=None, pro_relu=False, tables=None, epi=L.SAR_EPI_NONE, aux=None,
              aux_affine=None, aux_mean=None, bf16=False, packed=None, partials_out=None, aux2=None, aux_mask=None,
              split="default", bounds=None, aux_even_frames=False):
    """Launch sar_conv_gemm_f32 -- or, with bf16=True, sar_conv_gemm_bf16 (M % 8 == 0, Kc >= 16: bf16
    MFMA operands, fp32 everything else; other shapes stay on the fp32 kernel); or, with split="bf16x6", the fp32-accurate
    split arithmetic on the bf16 / fp16 matrix pipe (sar_conv_gemm_split; `packed` = the PackedSplitWeights image or None = pack
    here; bounds = (src_bound, w_bound) cells for the fp16 arithmetics, None = computed here by device kernels).  Returns
    (partials, nparts) when the epilogue reduces, else None."""
    lib = L.load()
    if split == "default":
        split = DEFAULT_SPLIT
    split = split if (split and split_applicable(mode, V, Kc, M, taps, stride, tables, pro, transposed, pad)
                      and (epi != L.SAR_EPI_ADD_GATE or mode == L.SAR_CONV_GRAPH)) else None
    if split and (mode == L.SAR_CONV_GRAPH or taps == 1) and split not in ("bf16x6", "f16x3a"):
        split = None
    if split:
        bf16 = False
    d = ConvDesc()
    d.mode, d.transposed, d.B, d.V = mode, int(transposed), B, V
    d.T_src, d.T_out, d.Kc, d.M = T_src, T_out, Kc, M
    d.taps, d.stride, d.pad, d.pro_relu, d.epi = taps, stride, pad, int(pro_relu), epi
    _f32(src), _f32(out), _f32(W)
    d.src, d.ld_src = ptr(src), src.stride(0)
    d.out, d.ld_out = ptr(out), out.stride(0)
    bf16 = bf16 and M % 8 == 0 and Kc >= 16
    use_packed = (bf16 or split) and packed is not None      # packed: the operand image from PackedWeights (W is then not read)
    w_bound = bounds[1] if bounds is not None else None
    if split and packed is None:
        packed, w_bound = _pack_split_single(W, w_stride_tap, w_stride_c, taps, Kc, M, split)
        use_packed = True
    d.W, d.w_stride_tap, d.w_stride_c = (None if use_packed else ptr(W)), w_stride_tap, w_stride_c
    d.bias = ptr(_f32(bias))
    if pro is not None:
        d.pro_scale, d.pro_shift = ptr(_f32(pro[0])), ptr(_f32(pro[1]))
    if tables is not None:
        d.g_idx, d.g_wt, d.g_colsum = ptr(tables.idx), ptr(tables.wt), ptr(tables.colsum)
        for i in range(3):
            d.nz[i] = tables.nz[i]
        # aux_even_frames: `aux` holds the even output frames only (include/sar_hip.h SAR_GRAPH_AUX_EVEN_FRAMES: the skip gradient through a
        # stride-2 residual convolution, computed compactly)
        d.g_flags = (tables.g_flags | (L.SAR_GRAPH_ONE_TILE_WG if GRAPH_ONE_TILE_WG else 0)
                     | (L.SAR_GRAPH_AUX_EVEN_FRAMES if aux_even_frames else 0))
    if aux is not None:
        d.aux, d.ld_aux = ptr(_f32(aux)), aux.stride(0)
    if aux_affine is not None:
        d.aux_scale, d.aux_shift = ptr(_f32(aux_affine[0])), ptr(_f32(aux_affine[1]))
    d.aux_mean = ptr(_f32(aux_mean))
    if epi == L.SAR_EPI_ADD_GATE:      # fp32 kernel only: aux2 [M][ld] floats, aux_mask [M][ld / 4] bytes (relu_mask layout)
        assert not bf16 and aux2 is not None and aux_mask is not None and aux_mask.dtype == torch.uint8
        assert aux2.stride(0) % 4 == 0 and aux_mask.shape == (M, aux2.stride(0) // 4) and aux_mask.is_contiguous()
        d.aux2, d.ld_aux2, d.aux_mask = ptr(_f32(aux2)), aux2.stride(0), ptr(aux_mask)
    partials = None
    nparts = 0
    if epi in (L.SAR_EPI_STATS, L.SAR_EPI_MASK, L.SAR_EPI_ADD_GATE):
        nparts = lib.sar_conv_gemm_split_nparts(C.byref(d)) if split else lib.sar_conv_gemm_nparts(C.byref(d))
        if nparts <= 0:
            check(nparts or -1, "sar_conv_gemm_nparts")
        if partials_out is not None:      # caller-owned rows of a stacked partials tensor (sar_amd/stgin.py)
            assert partials_out.shape == (M, nparts, 2) and partials_out.is_contiguous()
            partials = partials_out
        else:
            partials = torch.empty((M, nparts, 2), dtype=torch.float32, device=src.device)
        d.partials = ptr(partials)
    # algorithmic work: the forward conv's MACs (a data gradient costs the same MACs as its forward conv)
    n_conv = B * (T_src if transposed else T_out) * V
    flops = 2.0 * M * Kc * taps * n_conv
    tag = ("gemm_graph" if mode == L.SAR_CONV_GRAPH else ("gemm_temporal%d%s" % (taps, "_dgrad" if transposed else "")))
    if split:
        src_bound = bounds[0] if bounds is not None else None
        if split.startswith("f16") and src_bound is None:
            src_bound = _src_bound_single(src, pro)
        with profiler.region(tag + "_split", flops, 4.0 * (Kc * B * T_src * V + M * B * T_out * V)):
            check(lib.sar_conv_gemm_split(C.byref(d), L.SAR_SPLIT[split], ptr(packed), ptr(src_bound), ptr(w_bound), stream_ptr()),
                  "sar_conv_gemm_split")
    elif bf16:
        ws = packed if use_packed else torch.empty(lib.sar_conv_gemm_bf16_workspace_bytes(C.byref(d)), dtype=torch.uint8,
                                                   device=src.device)
        with profiler.region(tag + "_bf16", flops, 4.0 * (Kc * B * T_src * V + M * B * T_out * V)):
            check(lib.sar_conv_gemm_bf16(C.byref(d), ptr(ws), stream_ptr()), "sar_conv_gemm_bf16")
    else:
        with profiler.region(tag, flops, 4.0 * (Kc * B * T_src * V + M * B * T_out * V)):
            check(lib.sar_conv_gemm_f32(C.byref(d), stream_ptr()), "sar_conv_gemm_f32")
    return (partials, nparts) if partials is not None else None


_WGRAD1_SLOTS = int(__import__("os").environ.get("SAR_WGRAD1_SLOTS", "1024"))      # workgroups in flight of the 1-tap weight gradients
_WGRAD9_SLOTS = int(__import__("os").environ.get("SAR_WGRAD9_SLOTS", "1024"))      # ... of the 9-tap temporal ones (sweep knobs)
_WGRADG_SLOTS = int(__import__("os").environ.get("SAR_WGRADG_SLOTS", "512"))       # ... of the graph ones


# columns per workgroup of the first layer's streaming weight gradient (graph_wgrad_small4_kernel: 256-column iterations)
_WGRAD_SMALL_COLS = int(__import__("os").environ.get("SAR_WGRAD_SMALL_COLS", "4096"))

# SAR_SLAB_BATCH=0: every weight gradient reduces its slabs by its own launch again (A/B switch)
SLAB_BATCH = __import__("os").environ.get("SAR_SLAB_BATCH", "1") == "1"
# when an engine flushes its batch: "end" = at the end of backward (and before a bucket goes to the all-reduce), "bucket" = at every
# bucket boundary also without a data-parallel callback, "block" = behind every block
SLAB_FLUSH = __import__("os").environ.get("SAR_SLAB_FLUSH", "end")


class SlabBatch:
    """The partial slabs of the weight gradients of a backward pass, summed by ONE launch per flush() (sar_slab_reduce_batch_f32:
    per element the additions of sar_slab_reduce_f32 in the same order -- bit-identical) instead of one launch behind every
    weight-gradient kernel: a step had 20-22 of them, 3-4 % of the fp32-storage steps (tools/skip_probe.py).  The slab buffers are
    owned here and re-used every step (key = the destination slice of the flat gradient buffer), so a flush of the same items finds
    its device table in the cache.  Everything -- weight-gradient kernels and flush() -- must be issued on ONE stream (the engines'
    weight-gradient stream); an engine flushes before a gradient bucket is handed to the all-reduce and at the end of backward."""

    ITEM = [("slab", "<u8"), ("out", "<u8"), ("stride", "<i8"), ("n", "<i8"), ("nsplit", "<i4"), ("reserved", "<i4")]

    def __init__(self):
        self._slabs, self._pending, self._tables = {}, [], {}

    def slab(self, out, nsplit, n):
        """the (nsplit, n) slab buffer of the weight gradient that ends in out[0:n]"""
        key = (out.data_ptr(), nsplit, n)
        t = self._slabs.get(key)
        if t is None:
            if len(self._slabs) >= 512:      # a caller that keeps changing its batch geometry: do not grow without bound
                assert not self._pending, "SlabBatch: slab buffers would be dropped with reductions pending"
                self._slabs.clear()
                self._tables.clear()
            t = self._slabs[key] = torch.empty((nsplit, n), dtype=torch.float32, device=out.device)
        return t

    def add(self, slab, nsplit, n, out):
        assert out.numel() >= n and out.is_contiguous() and slab.stride(0) >= n
        self._pending.append((slab.data_ptr(), out.data_ptr(), slab.stride(0), n, nsplit, 0))
        self._device = slab.device

    def flush(self):
        if not self._pending:
            return
        key, self._pending = tuple(self._pending), []
        ent = self._tables.get(key)
        if ent is None:
            import numpy as np
            tab = np.array(list(key), dtype=np.dtype(self.ITEM, align=True))
            assert tab.dtype.itemsize == 40          # sizeof(sar_slab_item)
            ent = self._tables[key] = (torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(self._device), max(it[3] for it in key))
        check(L.load().sar_slab_reduce_batch_f32(ptr(ent[0]), len(key), ent[1], stream_ptr()), "sar_slab_reduce_batch_f32")


def conv_wgrad(mode, src, dout, dW_out, *, B, V, T_src, T_out, Kc, M, taps, stride=1, pad=0, pro=None, pro_relu=False,
               tables=None, w_stride_tap, w_stride_c, wsize, bsize, nsplit=None, bf16=False, split="default", bounds=None,
               slabs=None):
    """dW (and dbias, stored right behind it) -> dW_out[0 : wsize+bsize] (flat float32 view).  bf16=True routes the
    9-tap temporal operator (V = 25; stride 1, or stride 2 with the even-T SAME padding 3) to sar_conv_wgrad_bf16 (bf16 MFMA operands, fp32 accumulation and bias
    sums); every other shape stays on the fp32 kernel."""
    lib = L.load()
    d = WgradDesc()
    d.mode, d.B, d.V, d.T_src, d.T_out, d.Kc, d.M = mode, B, V, T_src, T_out, Kc, M
    d.taps, d.stride, d.pad, d.pro_relu = taps, stride, pad, int(pro_relu)
    _f32(src), _f32(dout)
    d.src, d.ld_src, d.dout, d.ld_dout = ptr(src), src.stride(0), ptr(dout), dout.stride(0)
    if pro is not None:
        d.pro_scale, d.pro_shift = ptr(_f32(pro[0])), ptr(_f32(pro[1]))
    if tables is not None:
        d.g_idx, d.g_wt, d.g_colsum = ptr(tables.idx), ptr(tables.wt), ptr(tables.colsum)
        for i in range(3):
            d.nz[i] = tables.nz[i]
        d.g_flags = tables.g_flags
    # split="bf16x6" / "f16x3a": the split arithmetic of csrc/conv_wgrad_split.hip where it is built (bounds = (src_bound,
    # dout_bound) cells for the fp16 arithmetic, None = computed here by device kernels); other shapes stay on the fp32 kernel
    if split == "default":
        split = DEFAULT_SPLIT
    if taps == 1 and mode == L.SAR_CONV_TEMPORAL and not TAP1_SPLIT:
        split = None
    sp_blocks = 0
    if split in ("bf16x6", "f16x3a"):
        wk, kt = C.c_int(0), C.c_int(0)
        sp_blocks = lib.sar_conv_wgrad_split_blocks(C.byref(d), L.SAR_SPLIT[split], C.byref(wk), C.byref(kt))
        if sp_blocks > 0:
            bf16 = False
            if nsplit is None:      # one round of the 512 resident workgroups (two per CU); SAR_WGRAD_SPLIT_SLOTS: experiment
                ntiles = B * ((T_out * V + kt.value - 1) // kt.value)
                nsplit = max(1, min(ntiles, _WGRADS_SLOTS // sp_blocks)) * wk.value
            else:
                nsplit = (nsplit + wk.value - 1) // wk.value * wk.value
    split = split if sp_blocks > 0 else None
    ct = 32 if (mode == L.SAR_CONV_TEMPORAL and taps == 9) else 64
    # (a bf16 graph weight gradient was built and measured: with the adjacency gather in its stager it ran 1.7x SLOWER
    # than the fp32 kernel -- 9.0 vs 5.4 ms per step -- so the graph weight gradient stays on the fp32 kernel)
    bf16 = bf16 and mode == L.SAR_CONV_TEMPORAL and taps == 9 and V == 25 and (stride == 1 or (stride == 2 and pad == 3))
    if bf16 and nsplit is None:     # 64 (m) x 64 / 32 (c, stride 1 / 2) weight blocks, 8-frame tiles; one round of the 512 resident workgroups
        ntiles = B * ((T_out + 7) // 8)
        cb = 64 if stride == 1 else 32
        nsplit = max(1, min(ntiles, 512 // (((M + 63) // 64) * ((Kc + cb - 1) // cb))))
    if nsplit is None and mode == L.SAR_CONV_GRAPH and Kc <= 4 and pro is None:
        # streaming kernel of the 3-channel first layer (conv_wgrad.hip: graph_wgrad_small_kernel): ~8 chunks of 64 columns per
        # workgroup (its loop is latency-bound: many short workgroups, not few long ones)
        nsplit = max(1, min(4096, (B * T_out * V + _WGRAD_SMALL_COLS - 1) // _WGRAD_SMALL_COLS))
    if nsplit is None:
        ft = max(2, min((128 // V) & ~1, (T_out + 1) & ~1))
        bf = 64
        if mode == L.SAR_CONV_GRAPH and V == 25 and Kc >= 32 and pro is None:      # fixed-geometry kernel: 2-frame tiles, 128 x 64 blocks when M > 64
            ft, bf = 2, (128 if M > 64 else 64)
        ntiles = B * ((T_out + ft - 1) // ft)
        wgs = ((M + bf - 1) // bf) * ((Kc + ct - 1) // ct)
        # workgroups in flight: two rounds of the 512 resident slots for the temporal kernels, one for the graph kernel
        # (tools/nsplit_sweep.py)
        target = _WGRADG_SLOTS if (mode == L.SAR_CONV_GRAPH and ft == 2) else (_WGRAD1_SLOTS if taps == 1 else _WGRAD9_SLOTS)
        nsplit = max(1, min(ntiles, (target + wgs - 1) // wgs))
    d.nsplit = nsplit
    d.w_stride_tap, d.w_stride_c, d.wsize, d.bsize = w_stride_tap, w_stride_c, wsize, bsize
    assert dW_out.numel() >= wsize + bsize and dW_out.is_contiguous()
    # slabs (SlabBatch): the slabs are summed by the batch's next flush() instead of by a launch of their own
    slab = (slabs.slab(dW_out, nsplit, wsize + bsize) if slabs is not None
            else torch.empty((nsplit, wsize + bsize), dtype=torch.float32, device=src.device))
    d.slab = ptr(slab)
    tag = "wgrad_graph" if mode == L.SAR_CONV_GRAPH else "wgrad_temporal%d" % taps
    if split:
        src_bound, dout_bound = bounds if bounds is not None else (None, None)
        if split.startswith("f16") and bounds is None:
            src_bound = _src_bound_single(src, pro)
            dout_bound = _src_bound_single(dout, None)
    with profiler.region(tag + ("_split" if split else "_bf16" if bf16 else ""), 2.0 * M * Kc * taps * B * T_out * V,
                         4.0 * (Kc * B * T_src * V + M * B * T_out * V)):
        if split:
            check(lib.sar_conv_wgrad_split(C.byref(d), L.SAR_SPLIT[split], ptr(src_bound), ptr(dout_bound), stream_ptr()),
                  "sar_conv_wgrad_split")
        elif bf16:
            check(lib.sar_conv_wgrad_bf16(C.byref(d), stream_ptr()), "sar_conv_wgrad_bf16")
        else:
            check(lib.sar_conv_wgrad_f32(C.byref(d), stream_ptr()), "sar_conv_wgrad_f32")
    if slabs is not None:
        slabs.add(slab, nsplit, wsize + bsize, dW_out)
        return
    check(lib.sar_slab_reduce_f32(ptr(slab), nsplit, wsize + bsize, wsize + bsize, ptr(dW_out), stream_ptr()),
          "sar_slab_reduce_f32")


def bn_finalize(partials, nparts, C_, count, eps, momentum, unbiased_running, gamma, beta, running_mean, running_var,
                mean, rstd, scale, shift):
    check(L.load().sar_bn_finalize_f32(ptr(partials), nparts, C_, float(count), eps, momentum, int(unbiased_running),
                                       ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(mean), ptr(rstd),
                                       ptr(scale), ptr(shift), stream_ptr()), "sar_bn_finalize_f32")


def bn_eval_affine(gamma, beta, running_mean, running_var, eps, scale, shift):
    check(L.load().sar_bn_eval_affine_f32(ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), eps,
                                          running_mean.numel(), ptr(scale), ptr(shift), stream_ptr()),
          "sar_bn_eval_affine_f32")


def bn_bwd_finalize(partials, nparts, chan_stride, part_stride, off1, off2, C_, count, gamma, mean, rstd, dgamma, dbeta,
                    k1=None, k2=None, k3=None, centered=True):
    check(L.load().sar_bn_bwd_finalize_f32(ptr(partials), nparts, chan_stride, part_stride, off1, off2, int(centered), C_,
                                           float(count),
                                           ptr(gamma), ptr(mean), ptr(rstd), ptr(dgamma), ptr(dbeta), ptr(k1), ptr(k2),
                                           ptr(k3), stream_ptr()), "sar_bn_bwd_finalize_f32")


def data_bn_stats(x, bone_parent, partials, motion=False):
    N, C_, T, V, M = x.shape
    check(L.load().sar_data_bn_stats_f32(ptr(x), N, C_, T, V, M, ptr(bone_parent), int(motion), ptr(partials), stream_ptr()),
          "sar_data_bn_stats_f32")


def data_bn_apply(x, bone_parent, scale, shift, out, motion=False):
    N, C_, T, V, M = x.shape
    check(L.load().sar_data_bn_apply_f32(ptr(x), N, C_, T, V, M, ptr(bone_parent), int(motion), ptr(scale), ptr(shift), ptr(out),
                                         out.stride(0), stream_ptr()), "sar_data_bn_apply_f32")


def data_bn_bwd_reduce(x, bone_parent, dy, mean, partials, motion=False):
    N, C_, T, V, M = x.shape
    check(L.load().sar_data_bn_bwd_reduce_f32(ptr(x), N, C_, T, V, M, ptr(bone_parent), int(motion), ptr(dy), dy.stride(0), ptr(mean),
                                              ptr(partials), stream_ptr()), "sar_data_bn_bwd_reduce_f32")


RELU_MASK = __import__("os").environ.get("SAR_RELU_MASK", "1") == "1"   # block tails write / read a 1-bit ReLU mask (A/B switch)


def relu_mask(u):
    """the mask tensor of a (C, n) fp32 block-tail output: one byte per float4, or None when the rows are not 4-element groups"""
    Cc, n = u.shape
    if not RELU_MASK or n % 4 or u.stride(0) % 4:
        return None
    return torch.empty((Cc, u.stride(0) // 4), dtype=torch.uint8, device=u.device)


def bn_add_relu_fwd(u, sc, sh, res_kind, r, rsc, rsh, y, mask=None, amax_cell=None):
    """amax_cell: the bound cell of y (split arithmetic): raised by the pass itself when the masked kernel runs, else by a
    separate sar_amax_f32 pass"""
    if mask is not None and amax_cell is not None:
        check(L.load().sar_bn_add_relu_fwd_mask_amax_f32(ptr(u), ptr(sc), ptr(sh), res_kind, ptr(r), ptr(rsc), ptr(rsh), ptr(y), ptr(mask),
                                                         ptr(amax_cell), u.shape[0], u.shape[1], u.stride(0), stream_ptr()),
              "sar_bn_add_relu_fwd_mask_amax_f32")
        return
    if amax_cell is not None:
        bn_add_relu_fwd(u, sc, sh, res_kind, r, rsc, rsh, y, mask)
        amax(y, amax_cell)
        return
    if mask is not None:
        check(L.load().sar_bn_add_relu_fwd_mask_f32(ptr(u), ptr(sc), ptr(sh), res_kind, ptr(r), ptr(rsc), ptr(rsh), ptr(y), ptr(mask),
                                                    u.shape[0], u.shape[1], u.stride(0), stream_ptr()), "sar_bn_add_relu_fwd_mask_f32")
        return
    check(L.load().sar_bn_add_relu_fwd_f32(ptr(u), ptr(sc), ptr(sh), res_kind, ptr(r), ptr(rsc), ptr(rsh), ptr(y),
                                           u.shape[0], u.shape[1], u.stride(0), stream_ptr()), "sar_bn_add_relu_fwd_f32")


_REDUCE_CHUNK = int(__import__("os").environ.get("SAR_BWD_REDUCE_CHUNK", "8192"))      # row elements per workgroup of the BN-backward reductions


# SAR_BN_TAIL=1: the BatchNorm-backward finalisation of every block tail runs in the reduce kernel's last workgroup instead of in
# its own launch(es).  Measured (three interleaved rounds each): fp32 59.44 vs 59.49 ms, bf16 13.37 vs 13.32, Path B 6.35 vs 6.37
# -- the launches it removes were waiting beside useful work of the other stream, not in front of it.  OFF by default: the sc1
# hand-off it relies on is measured behaviour of this part (MI355X_MICROARCH.md), not an architectural guarantee, and buys nothing.
BN_TAIL = __import__("os").environ.get("SAR_BN_TAIL", "0") == "1"
_tickets = {}


def bn_tail_tickets(device):
    """the ticket array of the folded finalisations, one per (device, stream): launches on ONE stream run one after the other and
    share it (4096 ints, zero; every launch leaves it zero); two host threads on two streams never share one"""
    key = (str(device), stream_ptr())
    if key not in _tickets:
        _tickets[key] = torch.zeros(4096, dtype=torch.int32, device=device)
    return _tickets[key]


def make_bn_tail(device, count, gamma, bn, dgamma, dbeta, rgamma=None, rbn=None, rdgamma=None, rdbeta=None):
    """sar_bn_tail for a block tail: bn / rbn are _BN records (rstd, k1, k2, k3)"""
    t = L.BnTail()
    t.ticket, t.count = ptr(bn_tail_tickets(device)), float(count)
    t.gamma, t.rstd, t.dgamma, t.dbeta = ptr(_f32(gamma)), ptr(bn.rstd), ptr(_f32(dgamma)), ptr(_f32(dbeta))
    t.k1, t.k2, t.k3 = ptr(bn.k1), ptr(bn.k2), ptr(bn.k3)
    if rbn is not None:
        t.rgamma, t.rrstd, t.rdgamma, t.rdbeta = ptr(_f32(rgamma)), ptr(rbn.rstd), ptr(_f32(rdgamma)), ptr(_f32(rdbeta))
        t.rk1, t.rk2, t.rk3 = ptr(rbn.k1), ptr(rbn.k2), ptr(rbn.k3)
    return t


def bn_add_relu_bwd_reduce(dy, y, u, r, mu=None, mr=None, tail=None, mask=None):
    """tail (make_bn_tail): the reduce kernel's last workgroup per channel also finalises (dgamma, dbeta, k1..k3): no
    sar_bn_bwd_finalize launch behind it.  mask (relu_mask, written by bn_add_relu_fwd): read instead of y."""
    Cc, n = u.shape
    nparts = max(1, min(4096, (n + _REDUCE_CHUNK - 1) // _REDUCE_CHUNK))
    partials = torch.empty((Cc, nparts, 4), dtype=torch.float32, device=u.device)
    if mask is not None and tail is None:
        check(L.load().sar_bn_add_relu_bwd_reduce_mask_f32(ptr(dy), ptr(mask), ptr(u), ptr(r), ptr(mu), ptr(mr), ptr(partials), nparts,
                                                           Cc, n, u.stride(0), stream_ptr()), "sar_bn_add_relu_bwd_reduce_mask_f32")
        return partials, nparts
    if tail is not None:
        assert Cc <= 4096
        check(L.load().sar_bn_add_relu_bwd_reduce_tail_f32(ptr(dy), ptr(y), ptr(u), ptr(r), ptr(mu), ptr(mr), ptr(partials), nparts, Cc,
                                                           n, u.stride(0), C.byref(tail), stream_ptr()),
              "sar_bn_add_relu_bwd_reduce_tail_f32")
        return partials, nparts
    check(L.load().sar_bn_add_relu_bwd_reduce_f32(ptr(dy), ptr(y), ptr(u), ptr(r), ptr(mu), ptr(mr), ptr(partials), nparts, Cc, n,
                                                  u.stride(0), stream_ptr()), "sar_bn_add_relu_bwd_reduce_f32")
    return partials, nparts


def bn_add_relu_bwd_apply(dy, y, u, r, k, rk, du, dr, dz_out, mask=None, amax_cell=None, amax_dr_cell=None):
    """amax_cell: the bound cell of du (split arithmetic), see bn_add_relu_fwd; amax_dr_cell: the same for dr (the operand of the
    residual branch's dense 1x1 data gradient)"""
    Cc, n = u.shape
    rk = rk or (None, None, None)
    if mask is not None and amax_cell is not None:
        check(L.load().sar_bn_add_relu_bwd_apply_mask_amax_f32(ptr(dy), ptr(mask), ptr(u), ptr(r), ptr(k[0]), ptr(k[1]), ptr(k[2]),
                                                               ptr(rk[0]), ptr(rk[1]), ptr(rk[2]), ptr(du), ptr(dr), ptr(dz_out),
                                                               ptr(amax_cell), ptr(amax_dr_cell if dr is not None else None),
                                                               Cc, n, u.stride(0), stream_ptr()),
              "sar_bn_add_relu_bwd_apply_mask_amax_f32")
        return
    if amax_cell is not None or amax_dr_cell is not None:
        bn_add_relu_bwd_apply(dy, y, u, r, k, rk, du, dr, dz_out, mask)
        if amax_cell is not None:
            amax(du, amax_cell)
        if amax_dr_cell is not None and dr is not None:
            amax(dr, amax_dr_cell)
        return
    if mask is not None:
        check(L.load().sar_bn_add_relu_bwd_apply_mask_f32(ptr(dy), ptr(mask), ptr(u), ptr(r), ptr(k[0]), ptr(k[1]), ptr(k[2]),
                                                          ptr(rk[0]), ptr(rk[1]), ptr(rk[2]), ptr(du), ptr(dr), ptr(dz_out), Cc, n,
                                                          u.stride(0), stream_ptr()), "sar_bn_add_relu_bwd_apply_mask_f32")
        return
    check(L.load().sar_bn_add_relu_bwd_apply_f32(ptr(dy), ptr(y), ptr(u), ptr(r), ptr(k[0]), ptr(k[1]), ptr(k[2]),
                                                 ptr(rk[0]), ptr(rk[1]), ptr(rk[2]), ptr(du), ptr(dr), ptr(dz_out), Cc, n,
                                                 u.stride(0), stream_ptr()), "sar_bn_add_relu_bwd_apply_f32")


def affine2(a, b, k, out, amax_cell=None):
    Cc, n = a.shape
    if amax_cell is not None:
        check(L.load().sar_affine2_amax_f32(ptr(a), ptr(b), ptr(k[0]), ptr(k[1]), ptr(k[2]), ptr(out), ptr(amax_cell), Cc, n, a.stride(0),
                                            stream_ptr()), "sar_affine2_amax_f32")
        return
    check(L.load().sar_affine2_f32(ptr(a), ptr(b), ptr(k[0]), ptr(k[1]), ptr(k[2]), ptr(out), Cc, n, a.stride(0),
                                   stream_ptr()), "sar_affine2_f32")


def pool_fwd(y, B, TV, Mp, feat):
    check(L.load().sar_pool_fwd_f32(ptr(y), y.stride(0), y.shape[0], B, TV, Mp, ptr(feat), stream_ptr()),
          "sar_pool_fwd_f32")


def fc_fwd(feat, W, bias, logits):
    N, Cc = feat.shape
    check(L.load().sar_fc_fwd_f32(ptr(feat), ptr(W), ptr(bias), N, Cc, logits.shape[1], ptr(logits), stream_ptr()),
          "sar_fc_fwd_f32")


def softmax_ce(logits, labels, inv_global_batch, loss_sum=None, dlogits=None, probs=None):
    N, K = logits.shape
    assert labels.dtype == torch.int64 and labels.is_cuda
    check(L.load().sar_softmax_ce_f32(ptr(logits), ptr(labels), N, K, inv_global_batch, ptr(loss_sum), ptr(dlogits),
                                      ptr(probs), stream_ptr()), "sar_softmax_ce_f32")


def fc_bwd(feat, W, dlogits, dW, dbias, dfeat):
    N, Cc = feat.shape
    check(L.load().sar_fc_bwd_f32(ptr(feat), ptr(W), ptr(dlogits), N, Cc, dlogits.shape[1], ptr(dW), ptr(dbias),
                                  ptr(dfeat), stream_ptr()), "sar_fc_bwd_f32")


def pool_bwd(dfeat, B, TV, Mp, dy):
    check(L.load().sar_pool_bwd_f32(ptr(dfeat), dy.stride(0), dy.shape[0], B, TV, Mp, ptr(dy), stream_ptr()),
          "sar_pool_bwd_f32")


def sgd_nesterov(w, v, g, lr_dev, momentum):
    check(L.load().sar_sgd_nesterov_f32(ptr(w), ptr(v), ptr(g), w.numel(), ptr(lr_dev), momentum, stream_ptr()),
          "sar_sgd_nesterov_f32")


def transpose(inp, out, batch, R, Cc):
    check(L.load().sar_transpose_f32(ptr(inp), ptr(out), batch, R, Cc, stream_ptr()), "sar_transpose_f32")


# ------------------------------------------------------------------------------------------------ dense adjacency
def graph_dense_fwd(y, A, out, K, F, V, nframes, stats=False, add=None):
    """out[m] = sum_k y[k F + m] . A_k (+ add[m]) (sar_graph_dense_fwd_f32); returns (partials, nparts) when stats."""
    lib = L.load()
    partials, nparts = None, 0
    if stats:
        nparts = lib.sar_graph_dense_nparts(nframes)
        partials = torch.empty((F, nparts, 2), dtype=torch.float32, device=y.device)
    check(lib.sar_graph_dense_fwd_f32(ptr(_f32(y)), y.stride(0), ptr(_f32(A)), ptr(_f32(out)), out.stride(0), K, F, V, nframes,
                                      ptr(partials), ptr(_f32(add)), add.stride(0) if add is not None else 0, stream_ptr()),
          "sar_graph_dense_fwd_f32")
    return (partials, nparts) if stats else None


def graph_dense_bwd_data(dout, A, dy, K, F, V, nframes):
    check(L.load().sar_graph_dense_bwd_data_f32(ptr(_f32(dout)), dout.stride(0), ptr(_f32(A)), ptr(_f32(dy)), dy.stride(0), K, F, V,
                                                nframes, stream_ptr()), "sar_graph_dense_bwd_data_f32")


def graph_dense_dA(y, dout, dA, K, F, V, nframes, nsplit=64):
    lib = L.load()
    nsplit = max(1, min(nsplit, (nframes + 7) // 8))
    slab = torch.empty(lib.sar_graph_dense_dadj_slab_floats(K, F, V, nsplit), dtype=torch.float32, device=y.device)
    check(lib.sar_graph_dense_dadj_f32(ptr(_f32(y)), y.stride(0), ptr(_f32(dout)), dout.stride(0), K, F, V, nframes, nsplit,
                                     ptr(slab), ptr(_f32(dA)), stream_ptr()), "sar_graph_dense_dadj_f32")


# ------------------------------------------------------------------------------------------------ graph isomorphism conv
def conv_gemm_nparts(B, V, T_src, T_out, Kc, M, taps=1, stride=1, pad=0, transposed=False, epi=L.SAR_EPI_STATS):
    """partial sums per output row that sar_conv_gemm_f32 (TEMPORAL) writes for this geometry"""
    d = ConvDesc()
    d.mode, d.transposed, d.B, d.V, d.T_src, d.T_out, d.Kc, d.M = L.SAR_CONV_TEMPORAL, int(transposed), B, V, T_src, T_out, Kc, M
    d.taps, d.stride, d.pad, d.epi = taps, stride, pad, epi
    n = L.load().sar_conv_gemm_nparts(C.byref(d))
    if n <= 0:
        check(n or -1, "sar_conv_gemm_nparts")
    return n


def gin_adjacency(A, eps, table, scale, slice_scale=None, Km1=None, V=None):
    """table [Km1+1][V][V] (or None), scale[C] = 1 + eps, slice_scale[Km1+1] = (1, .., 1, 1 + eps) (or None)"""
    if A is not None:
        Km1, V = A.shape[0], A.shape[1]
    check(L.load().sar_gin_adjacency_f32(ptr(_f32(A)), Km1, V, ptr(_f32(eps)), ptr(_f32(table)), ptr(_f32(scale)), scale.numel(),
                                         ptr(_f32(slice_scale)), stream_ptr()), "sar_gin_adjacency_f32")


def graph_gather_sum(inp, tables, scale, K, F, V, out, add=None, k0=0):
    """out[m] = sum_k scale[k] (in[k F + m] gathered with slice k0 + k of `tables`) (+ add[m])"""
    n = out.shape[1]
    nz = (C.c_int32 * K)(*tables.nz[k0:k0 + K])
    check(L.load().sar_graph_gather_sum_f32(ptr(_f32(inp)), inp.stride(0), ptr(tables.idx[k0:]), ptr(tables.wt[k0:]),
                                            C.cast(nz, C.c_void_p), ptr(_f32(scale)),
                                            K, F, V, n, ptr(_f32(out)), out.stride(0), ptr(_f32(add)),
                                            add.stride(0) if add is not None else 0, stream_ptr()), "sar_graph_gather_sum_f32")


def graph_gather_expand(inp, tables, K, F, V, out, k0=0):
    """out[k F + m] = in[m] gathered with slice k0 + k of `tables`"""
    n = inp.shape[1]
    nz = (C.c_int32 * K)(*tables.nz[k0:k0 + K])
    check(L.load().sar_graph_gather_expand_f32(ptr(_f32(inp)), inp.stride(0), ptr(tables.idx[k0:]), ptr(tables.wt[k0:]),
                                               C.cast(nz, C.c_void_p), K, F, V, n, ptr(_f32(out)), out.stride(0), stream_ptr()),
          "sar_graph_gather_expand_f32")


def gin_sum_fwd(a, scale, shift, K, s_out, stats=False):
    Cc, n = s_out.shape
    partials, nparts = None, 0
    if stats:
        nparts = L.load().sar_gin_nparts(n)
        partials = torch.empty((Cc, nparts, 2), dtype=torch.float32, device=a.device)
    check(L.load().sar_gin_sum_fwd_f32(ptr(_f32(a)), a.stride(0), ptr(_f32(scale)), ptr(_f32(shift)), K, Cc, n, ptr(_f32(s_out)),
                                       s_out.stride(0), ptr(partials), stream_ptr()), "sar_gin_sum_fwd_f32")
    return (partials, nparts) if stats else None


def gin_bwd_reduce(ds, a, scale, shift, mean, K):
    Cc, n = ds.shape
    nparts = L.load().sar_gin_nparts(n)
    partials = torch.empty((K * Cc, nparts, 2), dtype=torch.float32, device=a.device)
    check(L.load().sar_gin_bwd_reduce_f32(ptr(_f32(ds)), ds.stride(0), ptr(_f32(a)), a.stride(0), ptr(_f32(scale)), ptr(_f32(shift)),
                                          ptr(_f32(mean)), K, Cc, n, ptr(partials), stream_ptr()), "sar_gin_bwd_reduce_f32")
    return partials, nparts


def gin_bwd_apply(ds, a, scale, shift, k, K, da):
    Cc, n = ds.shape
    check(L.load().sar_gin_bwd_apply_f32(ptr(_f32(ds)), ds.stride(0), ptr(_f32(a)), a.stride(0), ptr(_f32(scale)), ptr(_f32(shift)),
                                         ptr(_f32(k[0])), ptr(_f32(k[1])), ptr(_f32(k[2])), K, Cc, n, ptr(_f32(da)), da.stride(0),
                                         stream_ptr()), "sar_gin_bwd_apply_f32")


def gin_eps_grad(G, W, eps, deps):
    assert G.is_contiguous() and W.is_contiguous() and G.numel() == W.numel()
    check(L.load().sar_gin_eps_grad_f32(ptr(G), ptr(W), G.numel(), ptr(eps), ptr(deps), stream_ptr()), "sar_gin_eps_grad_f32")


# ------------------------------------------------------------------------------------------------ ResNet-18 ops
def _shape_tag(geo):
    """SAR_PROFILE_SHAPES=1 (diagnostic, tools/pathb_layers.py): one profiler bucket per layer geometry"""
    if not _PROFILE_SHAPES:
        return ""
    return " Kc%d M%d %dx%d->%dx%d s%d" % (geo["Kc"], geo["M"], geo["H_src"], geo["W_src"], geo["H_out"], geo["W_out"], geo["stride"])


def _conv2d_desc(src, *, B, Kc, M, H_src, W_src, H_out, W_out, KH, KW, stride, pad, transposed=False, pro=None,
                 pro_relu=False):
    d = L.Conv2dDesc()
    d.transposed, d.B, d.Kc, d.M = int(transposed), B, Kc, M
    d.H_src, d.W_src, d.H_out, d.W_out = H_src, W_src, H_out, W_out
    d.KH, d.KW, d.stride, d.pad, d.pro_relu = KH, KW, stride, pad, int(pro_relu)
    d.src, d.ld_src = ptr(_f32(src)), src.stride(0)
    if pro is not None:
        d.pro_scale, d.pro_shift = ptr(_f32(pro[0])), ptr(_f32(pro[1]))
    return d


def conv2d_split_applicable(*, KH, KW, stride, pad, Kc, M, H_src, W_src, H_out, W_out, aux_even_pixels=False, transposed=False,
                            epi=None, **_):
    """shapes csrc/conv2d_split.hip is built for: 3x3 / stride 1 / pad 1 (forward and data gradient; windows of <= 512 staged pixels)
    and the 3x3 / stride 2 / pad 1 DATA GRADIENT onto an image of exactly twice the size; the others stay fp32"""
    if KH == 3 and KW == 3 and stride == 2 and pad == 1 and transposed:
        if not (H_out == 2 * H_src and W_out == 2 * W_src and 16 <= Kc <= 512 and M % 8 == 0 and W_src <= 128):
            return False
        if aux_even_pixels and epi != L.SAR_EPI_ADD:
            return False
        cpix = H_src * W_src
        rw = (128 // cpix) * (H_src + 1) * (W_src + 1) if cpix <= 64 else (min(H_src, 128 // W_src) + 1) * (W_src + 1)
        return rw <= 256
    if not (KH == 3 and KW == 3 and stride == 1 and pad == 1 and H_src == H_out and W_src == W_out):
        return False
    if not (8 <= Kc <= 512 and M % 8 == 0 and not aux_even_pixels):
        return False
    opix = H_out * W_out
    rw = (256 // opix) * (H_out + 2) * (W_out + 2) if opix <= 128 else (min(H_out, 256 // W_out) + 2) * (W_out + 2) if W_out <= 256 else 1 << 30
    return rw <= 512


def _pack_split_conv2d(W, w_stride_tap, w_stride_c, Kc, M, arith, transposed):
    """one 3x3 tensor stored (tap, c, m), packed on the spot (kernel tests): (image, w_bound).  transposed: W is the DATA-GRADIENT
    operand layout (tap, m, c) of the forward tensor as the fp32 kernel takes it (element (tap, c', m') of the call = weight of
    forward tap `tap`); the stride-1 split kernel (transposed="mirror") wants the mirrored taps: item element (tap, c', m') =
    W[8 - tap][c'][m']; the stride-2 data gradient addresses the forward taps directly (no mirroring)."""
    pk = PackedSplitWeights(arith)
    if transposed == "mirror":
        pk.add("w", 8 * w_stride_tap, -w_stride_tap, w_stride_c, 1, 9, Kc, M)
    else:
        pk.add("w", 0, w_stride_tap, w_stride_c, 1, 9, Kc, M)
    pk.finalize(W.device)
    pk.refresh(W)
    return pk.image("w"), pk.bound("w")


def conv2d_gemm(src, out, W, w_stride_tap, w_stride_c, *, epi=L.SAR_EPI_NONE, aux=None, aux_affine=None, aux_mean=None,
                aux_even_pixels=False, ctx=None, split="default", packed=None, bounds=None, **geo):
    """sar_conv2d_gemm_f32 -- or, with split="f16x3a" / "bf16x6" on the shapes conv2d_split_applicable() names, the fp32-accurate
    split arithmetic on the fp16 / bf16 matrix pipe (sar_conv2d_gemm_split; `packed` = the PackedSplitWeights image of the launch's
    view of the weights or None = pack here from W; bounds = (src_bound, w_bound) cells, None = computed here by device kernels).
    Returns (partials, nparts) when the epilogue reduces.  aux_even_pixels: SAR_C2D_AUX_EVEN_PIXELS
    (aux is the compact data gradient of the parallel 1x1 / stride 2 convolution, added at the even pixels only).
    ctx: an L.Context whose side streams the call may fan out over (None: the current stream only)."""
    lib = L.load()
    if split == "default":
        split = DEFAULT_SPLIT
    if split not in ("f16x3a", "bf16x6") or not conv2d_split_applicable(aux_even_pixels=aux_even_pixels, epi=epi, **geo):
        split = None
    d = _conv2d_desc(src, **geo)
    d.ctx = ctx.handle if ctx is not None else None
    d.flags = L.SAR_C2D_AUX_EVEN_PIXELS if aux_even_pixels else 0
    d.out, d.ld_out = ptr(_f32(out)), out.stride(0)
    d.W, d.w_stride_tap, d.w_stride_c, d.epi = ptr(_f32(W)), w_stride_tap, w_stride_c, epi
    if aux is not None:
        d.aux, d.ld_aux = ptr(_f32(aux)), aux.stride(0)
    if aux_affine is not None:
        d.aux_scale, d.aux_shift = ptr(_f32(aux_affine[0])), ptr(_f32(aux_affine[1]))
    d.aux_mean = ptr(_f32(aux_mean))
    partials = None
    nparts = 0
    kslab = None
    if split:      # small feature maps: workspace of the K-split (sar_hip.h), 0 bytes = not planned for this shape
        nb = lib.sar_conv2d_gemm_split_slab_bytes(C.byref(d))
        if nb > 0:
            kslab = torch.empty(nb // 4, dtype=torch.float32, device=src.device)
            d.slab = ptr(kslab)
    if epi in (L.SAR_EPI_STATS, L.SAR_EPI_MASK):
        nparts = lib.sar_conv2d_gemm_split_nparts(C.byref(d)) if split else lib.sar_conv2d_nparts(C.byref(d))
        if nparts <= 0:
            check(nparts or -1, "sar_conv2d_nparts")
        partials = torch.empty((geo["M"], nparts, 2), dtype=torch.float32, device=src.device)
        d.partials = ptr(partials)
    n_conv = geo["B"] * (geo["H_src"] * geo["W_src"] if geo.get("transposed") else geo["H_out"] * geo["W_out"])
    flops = 2.0 * geo["M"] * geo["Kc"] * geo["KH"] * geo["KW"] * n_conv
    tag = "conv2d_%dx%d%s" % (geo["KH"], geo["KW"], "_dgrad" if geo.get("transposed") else "")
    if split:
        w_bound = bounds[1] if bounds is not None else None
        if packed is None:
            packed, w_bound = _pack_split_conv2d(W, w_stride_tap, w_stride_c, geo["Kc"], geo["M"], split,
                                                 "mirror" if (geo.get("transposed") and geo["stride"] == 1) else False)
        src_bound = bounds[0] if bounds is not None else None
        if split.startswith("f16") and src_bound is None:
            src_bound = _src_bound_single(src, geo.get("pro"))
        with profiler.region(tag + "_split" + _shape_tag(geo), flops):
            check(lib.sar_conv2d_gemm_split(C.byref(d), L.SAR_SPLIT[split], ptr(packed), ptr(src_bound), ptr(w_bound), stream_ptr()),
                  "sar_conv2d_gemm_split")
    else:
        with profiler.region(tag + _shape_tag(geo), flops):
            check(lib.sar_conv2d_gemm_f32(C.byref(d), stream_ptr()), "sar_conv2d_gemm_f32")
    return (partials, nparts) if partials is not None else None


_C2D_WG_SLOTS = int(__import__("os").environ.get("SAR_C2D_WG_SLOTS", "512"))


def conv2d_wgrad(src, dout, dW_tcm, *, split="default", bounds=None, slabs=None, **geo):
    """dW in (tap, c, m) layout -> dW_tcm (flat, taps*Kc*M floats): sar_conv2d_wgrad_f32 -- or, with split="f16x3a" / "bf16x6" on 3x3 /
    stride 1 / pad 1 at image widths 8 / 16 / 32 / 64, sar_conv2d_wgrad_split (fp32 results on the fp16 / bf16 matrix pipe; bounds =
    (src_bound, dout_bound) cells, None = computed here by device kernels).  Slabs are summed in slab order either way."""
    lib = L.load()
    if split == "default":
        split = DEFAULT_SPLIT
    if split not in ("f16x3a", "bf16x6"):
        split = None
    d = _conv2d_desc(src, **geo)
    d.dout, d.ld_dout = ptr(_f32(dout)), dout.stride(0)
    taps = geo["KH"] * geo["KW"]
    n = taps * geo["Kc"] * geo["M"]
    flops = 2.0 * geo["M"] * geo["Kc"] * taps * geo["B"] * geo["H_out"] * geo["W_out"]
    if split:
        wk, kt = C.c_int(0), C.c_int(0)
        wgs = lib.sar_conv2d_wgrad_split_blocks(C.byref(d), L.SAR_SPLIT[split], C.byref(wk), C.byref(kt))
        if wgs == L.SAR_E_UNSUP:
            split = None
        else:
            check(0 if wgs > 0 else (wgs or -1), "sar_conv2d_wgrad_split_blocks")
    if split:
        ntiles = (geo["B"] * geo["H_out"] * geo["W_out"] + kt.value - 1) // kt.value
        groups = max(1, min(ntiles, (_C2D_WG_SLOTS + wgs - 1) // wgs))      # one round of the resident workgroups (2 per CU)
        nsplit = groups * wk.value
        d.nsplit = nsplit
        slab = slabs.slab(dW_tcm, nsplit, n) if slabs is not None else torch.empty((nsplit, n), dtype=torch.float32, device=src.device)
        d.slab = ptr(slab)
        sb, db = bounds if bounds is not None else (None, None)
        if split.startswith("f16"):
            if sb is None:
                sb = _src_bound_single(src, geo.get("pro"))
            if db is None:
                db = _src_bound_single(dout, None)
        with profiler.region("conv2d_wgrad_3x3_split" + _shape_tag(geo), flops):
            check(lib.sar_conv2d_wgrad_split(C.byref(d), L.SAR_SPLIT[split], ptr(sb), ptr(db), stream_ptr()), "sar_conv2d_wgrad_split")
        if slabs is not None:
            slabs.add(slab, nsplit, n, dW_tcm)
        else:
            check(lib.sar_slab_reduce_f32(ptr(slab), nsplit, n, n, ptr(dW_tcm), stream_ptr()), "sar_slab_reduce_f32")
        return
    wgs = ((geo["M"] + 63) // 64) * max(1, (geo["Kc"] + 31) // 32)
    # one round of the 512 resident workgroups (2 per CU): measured 22 % faster than 768 / 1024 in isolation (the kernels do not
    # fit 3 per CU, so a larger grid runs a second, partly filled round)
    nsplit = max(1, min(geo["B"] * max(1, geo["H_out"] // 2), (_C2D_WG_SLOTS + wgs - 1) // wgs))
    d.nsplit = nsplit
    slab = slabs.slab(dW_tcm, nsplit, n) if slabs is not None else torch.empty((nsplit, n), dtype=torch.float32, device=src.device)
    d.slab = ptr(slab)
    with profiler.region("conv2d_wgrad_%dx%d" % (geo["KH"], geo["KW"]) + _shape_tag(geo), flops):
        check(lib.sar_conv2d_wgrad_f32(C.byref(d), stream_ptr()), "sar_conv2d_wgrad_f32")
    if slabs is not None:
        slabs.add(slab, nsplit, n, dW_tcm)
    else:
        check(lib.sar_slab_reduce_f32(ptr(slab), nsplit, n, n, ptr(dW_tcm), stream_ptr()), "sar_slab_reduce_f32")


def permute3(inp, out, d0, d1, d2, s0, s1, s2):
    check(L.load().sar_permute3_f32(ptr(inp), ptr(out), d0, d1, d2, s0, s1, s2, stream_ptr()), "sar_permute3_f32")


class PermuteBatch:
    """Many 3-d re-layouts between two flat fp32 buffers in ONE launch (sar_permute3_batch_f32)."""

    ITEM = [("src_off", "<i8"), ("dst_off", "<i8"), ("s0", "<i8"), ("s1", "<i8"), ("s2", "<i8"),
            ("d0", "<i4"), ("d1", "<i4"), ("d2", "<i4"), ("reserved", "<i4")]

    def __init__(self):
        self.items = []

    def add(self, src_off, dst_off, d0, d1, d2, s0, s1, s2):
        self.items.append((src_off, dst_off, s0, s1, s2, d0, d1, d2, 0))

    def finalize(self, device):
        import numpy as np
        tab = np.array(self.items, dtype=np.dtype(self.ITEM, align=True))
        assert tab.dtype.itemsize == 56          # sizeof(sar_permute_item)
        self.table = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(device)
        self.max_elems = max(it[5] * it[6] * it[7] for it in self.items)

    def run(self, src, dst):
        check(L.load().sar_permute3_batch_f32(ptr(_f32(src)), ptr(_f32(dst)), ptr(self.table), len(self.items), self.max_elems,
                                              stream_ptr()), "sar_permute3_batch_f32")


def bn_relu_maxpool_fwd(x, scale, shift, y, B, H, W):
    check(L.load().sar_bn_relu_maxpool_fwd_f32(ptr(x), ptr(scale), ptr(shift), ptr(y), x.shape[0], B, H, W, x.stride(0),
                                               y.stride(0), stream_ptr()), "sar_bn_relu_maxpool_fwd_f32")


def bn_relu_maxpool_bwd(x, scale, shift, mean, dy, dz, B, H, W):
    Cc = x.shape[0]
    nparts = L.load().sar_bn_relu_maxpool_bwd_nparts(B, H, W)
    partials = torch.empty((Cc, nparts, 2), dtype=torch.float32, device=x.device)
    check(L.load().sar_bn_relu_maxpool_bwd_f32(ptr(x), ptr(scale), ptr(shift), ptr(mean), ptr(dy), ptr(dz), ptr(partials),
                                               nparts, Cc, B, H, W, x.stride(0), dy.stride(0), stream_ptr()),
          "sar_bn_relu_maxpool_bwd_f32")
    return partials, nparts


def adam(w, m, v, g, lr_dev, step_dev, beta1=0.9, beta2=0.999, eps=1e-8):
    check(L.load().sar_adam_f32(ptr(w), ptr(m), ptr(v), ptr(g), w.numel(), ptr(lr_dev), ptr(step_dev), beta1, beta2, eps,
                                stream_ptr()), "sar_adam_f32")


def conv2d_stem_dgrad(dout, w_packed, dx, *, B, H, W, H_out, W_out, M, KH, KW, stride, pad):
    """Image gradient of the one-input-channel stem conv (dout CN [M][B*Ho*Wo], w_packed [KH*KW][1][M])."""
    check(L.load().sar_conv2d_stem_dgrad_f32(ptr(_f32(dout)), dout.stride(0), ptr(_f32(w_packed)), B, H, W, H_out, W_out, M,
                                             KH, KW, stride, pad, ptr(dx), stream_ptr()), "sar_conv2d_stem_dgrad_f32")
