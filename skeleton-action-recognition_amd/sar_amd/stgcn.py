"""ST-GCN training engine on the HIP kernels (one process = one MI355X).

Mirrors the reference model (models/stgcn.py:101-160, blocks :11-64, GraphConvTD models/gcn.py:187-209)
and its train step (main_gnn.py:219-239): forward, hand-scheduled backward, gradients into ONE flat
fp32 buffer (so data parallelism is a single RCCL all-reduce, main_gnn.py:257 MirroredStrategy), fused
Nesterov SGD (main_gnn.py:312-314).  Parameters keep the reference's Keras layouts and names.

HBM layout: activations are [C][B*T*V] matrices (CN layout, include/sar_hip.h); per block the tensors
that cross a BatchNorm barrier (g = graph conv out, u = temporal conv out, r = residual conv out,
y = block out) are materialised once in forward and kept for backward; BN+ReLU is folded into the
consumer's operand load, BN statistics into the producer's epilogue.
"""
import math

import os

import numpy as np
import torch

from . import _lib as L
from . import ops

BN_EPS = 1e-3        # Keras BatchNormalization defaults (models/stgcn.py:27,37,56,111)
BN_MOMENTUM = 0.99
KS, KT = 3, 9        # kernel_size=[3, 9], models/stgcn.py:14
# A block's weight gradients are ISSUED behind its data gradients in the fp32 engines (side stream as before): they then run
# beside the NEXT block's element-wise BatchNorm passes instead of beside this block's data-gradient GEMMs.  Three interleaved
# rounds: fp32 59.22 -> 58.99 ms per step; bf16 13.26 -> 13.55 (slower: there the order stays as it was).  SAR_WGRAD_DEFER=0/1 forces it.
_WGRAD_DEFER_ENV = __import__("os").environ.get("SAR_WGRAD_DEFER")
_FUSE_TAIL_F32 = __import__("os").environ.get("SAR_F32_FUSE_TAIL", "1") == "1"   # SAR_EPI_ADD_GATE in the fp32 graph data gradient
# the arithmetic an STGCN built without an explicit `mfma` uses (the parity suites run once per value: tests/conftest.py)
DEFAULT_MFMA = __import__("os").environ.get("SAR_MFMA", "fp32")
SPLIT_ARITH = {"f32_split": "f16x3a", "f32_split_bf16x6": "bf16x6"}      # engine mode -> arithmetic of csrc/conv_gemm_split.hip
# A/B switch: which kernel families of a split engine take the split kernels (default all that are built)
# the skip gradient of a conv-residual block as its even frames only: "fp32" (default) = the fp32-arithmetic engine, where it measured
# +0.4 % (58.70 -> 58.47 ms, interleaved); "1" = the split engines too (single stream -0.29 ms, two streams 34.34 -> 34.44 ms: not taken);
# "0" = never (gpurun_out/compact_skip_ab*.txt)
# SAR_STGCN_AUX_STREAM=0: the split engine's term images, the zeroing of its bound cells and its Samuelson cells on the main stream again
# instead of on a third stream beside the data_bn stage and the first layer (round 6, as sar_amd/resnet.py; bit-identical).  One bench.py
# process per entry (profiles/r06_stgcn_aux_stream_ab.txt): f32_split 1 929 -> 1 936 clips/s (+0.3 %); the bf16 engine's single pack launch
# forked the same way: 5 497 -> 5 466 (-0.5 %), not forked
AUX_STREAM = __import__("os").environ.get("SAR_STGCN_AUX_STREAM", "1") == "1"
_COMPACT_SKIP = __import__("os").environ.get("SAR_COMPACT_SKIP", "fp32")
_SPLIT_KINDS = set(__import__("os").environ.get("SAR_SPLIT_KINDS", "tfwd,tdgrad,twgrad,gfwd,gdgrad,gwgrad,rfwd,rdgrad,rwgrad").split(","))
# (filters, stride, residual) -- models/stgcn.py:113-123
BLOCKS = [(64, 1, False), (64, 1, True), (64, 1, True), (64, 1, True), (128, 2, True), (128, 1, True), (128, 1, True),
          (256, 2, True), (256, 1, True), (256, 1, True)]


def same_pad(T, k, s):
    """TF 'SAME': (out, pad_begin, pad_end); the extra pad goes at the end."""
    out = -(-T // s)
    total = max((out - 1) * s + k - T, 0)
    return out, total // 2, total - total // 2


def ntu_adjacency():
    """graph/ntu_rgb_d.py:6-40 + graph/tools.py:4-30 ('spatial' labelling) -> (3,25,25) float64."""
    from graph.ntu_rgb_d import Graph  # the package's drop-in mirror of the reference module
    return Graph().A


class _BN:
    """Per-BatchNorm device state: parameters are views of the flat buffer, statistics are buffers."""

    def __init__(self, C, device):
        z = lambda: torch.zeros(C, dtype=torch.float32, device=device)
        self.moving_mean, self.moving_var = z(), torch.ones(C, dtype=torch.float32, device=device)
        self.mean, self.rstd, self.scale, self.shift = z(), z(), z(), z()
        self.k1, self.k2, self.k3 = z(), z(), z()


class STGCN:
    # split arithmetic (mfma="f32_split*"): off unless __init__ turns it on (subclasses with their own __init__ -- ST-GIN -- are fp32)
    mfma, split, spacked, _cells = "fp32", None, None, None
    _slabs, _slab_flush = None, "end"     # (ops.SlabBatch of the weight gradients: set by __init__; ST-GIN keeps its per-gradient reductions)
    _aux, _aux_pending, _bounds_forked = None, False, False     # (third stream: set by __init__; engines with their own __init__ have none)

    def __init__(self, num_classes=60, in_channels=3, num_node=25, A=None, device="cuda", seed=0, bone_pairs=None,
                 blocks=None, motion=False, mfma=None, trainable_adjacency=False):
        L.load()  # fail loudly if the HIP library is missing
        # mfma="fp32" (default) is the reference's arithmetic.
        # mfma="bf16" is SURVEY.md 8d config 3: activations and activation gradients are stored in HBM as bfloat16 (CN8
        # layout, csrc/cn8.h), every convolution / data gradient / weight gradient multiplies bf16 operands with fp32
        # accumulation (sar_conv_gemm_cn8, sar_conv_wgrad_cn8); BatchNorm statistics, master weights, gradients and the
        # optimizer stay fp32 (sar_amd/stgcn8.py holds the step).
        # mfma="bf16_operands" is the round-1 intermediate kept for A/B runs: bf16 MFMA operands, fp32 activations in HBM.
        # mfma="f32_split": fp32 RESULTS on the fp16 matrix pipe -- storage, BatchNorm statistics, epilogues and parameters are the
        # fp32 engine's, the contraction of the GEMM-shaped kernels multiplies two fp16 terms per operand (three products, fp32
        # accumulation: csrc/conv_gemm_split.hip; same parity tolerances as "fp32").  "f32_split_bf16x6": three bfloat16 terms,
        # six products (no operand scaling).  Kernels without a split form, and inference, stay on the fp32 kernels.
        if mfma is None:
            mfma = DEFAULT_MFMA if (type(self) is STGCN and not trainable_adjacency) else "fp32"
        assert mfma in ("fp32", "bf16", "bf16_operands", "f32_split", "f32_split_bf16x6")
        self.mfma = mfma
        self.cn8 = mfma == "bf16"
        self.bf16 = mfma == "bf16_operands"
        self.split = SPLIT_ARITH.get(mfma)
        # trainable_adjacency: the stacked adjacency becomes the trainable variable `adjacency_matrix`
        # (models/gcn.py:212-238 AdjGraphConv) shared by all blocks; the graph convolution then keeps the reference's
        # order -- 1x1 convolution to 3F channels, dense contraction with A (csrc/graph_dense.hip) -- and the backward pass
        # produces dA.  `train_adjacency` gates its gradient per step (main_gnn.py:228-232: trained only while
        # epoch > --freeze-graph-until).  fp32 only.
        self.dense_A = bool(trainable_adjacency)
        self.train_adjacency = True
        assert not (self.dense_A and mfma != "fp32"), "trainable adjacency is built for the fp32 engine"
        self.device = torch.device(device)
        self.num_classes, self.C_in, self.V = num_classes, in_channels, num_node
        self.blocks = list(blocks if blocks is not None else BLOCKS)
        A = np.asarray(ntu_adjacency() if A is None else A, dtype=np.float64)
        assert A.shape == (KS, num_node, num_node)
        self.A_host = A.astype(np.float32)
        self.A = torch.from_numpy(self.A_host).to(self.device)     # 'adjacency_matrix', non-trainable (stgcn.py:105-109)
        self.tab_fwd = ops.GraphTables(self.A_host, self.device, transpose=False)
        self.tab_bwd = ops.GraphTables(self.A_host, self.device, transpose=True)
        # the weight-gradient stream; SAR_WGRAD_PRIO: its priority (torch convention: lower = more urgent; default 0 = same as
        # the main chain)
        self._side = (ops.shared_side_stream(self.device, int(os.environ.get("SAR_WGRAD_PRIO", "0")))
                      if self.device.type == "cuda" and os.environ.get("SAR_WGRAD_STREAM", "1") == "1" else None)
        self._slabs = ops.SlabBatch() if ops.SLAB_BATCH else None     # one slab reduction per gradient bucket (ops.SlabBatch)
        self._slab_flush = ops.SLAB_FLUSH
        self._aux = ops.shared_aux_stream(self.device) if (AUX_STREAM and self.device.type == "cuda" and type(self) is STGCN) else None
        self.motion = bool(motion)   # motion stream (data_gen/gen_motion_data.py:24-27) of the joint / bone data, on the fly
        self.bone_parent = None
        if bone_pairs is not None:
            bp = np.full(num_node, -1, dtype=np.int32)
            for v1, v2 in bone_pairs:          # data_gen/gen_bone_data.py:36-41 (1-based pairs)
                bp[v1 - 1] = v2 - 1
            self.bone_parent = torch.from_numpy(bp).to(self.device)

        # ---- parameter table (Keras layouts), flat storage
        self.shapes = {}
        nch = num_node * in_channels
        self._add("data_bn.gamma", (nch,)), self._add("data_bn.beta", (nch,))
        cin = in_channels
        self.kinds = []
        for i, (f, s, res) in enumerate(self.blocks):
            pre = "l%d." % i
            kind = "none" if not res else ("identity" if (cin == f and s == 1) else "conv")   # stgcn.py:41-56
            self.kinds.append(kind)
            self._add(pre + "gcn.kernel", (1, 1, cin, KS * f)), self._add(pre + "gcn.bias", (KS * f,))
            self._add(pre + "bn1.gamma", (f,)), self._add(pre + "bn1.beta", (f,))
            self._add(pre + "tcn.kernel", (KT, 1, f, f)), self._add(pre + "tcn.bias", (f,))
            self._add(pre + "bn2.gamma", (f,)), self._add(pre + "bn2.beta", (f,))
            if kind == "conv":
                self._add(pre + "res.kernel", (1, 1, cin, f)), self._add(pre + "res.bias", (f,))
                self._add(pre + "res_bn.gamma", (f,)), self._add(pre + "res_bn.beta", (f,))
            cin = f
        self.C_last = cin
        self._add("logits.kernel", (1, 1, cin, num_classes)), self._add("logits.bias", (num_classes,))
        if self.dense_A:
            self._add("adjacency_matrix", (KS, num_node, num_node))
        # flat storage: every offset is a multiple of 4 floats so that each weight view is 16-byte aligned
        # (the GEMM kernels stage weight rows as float4); a bias stays glued to its kernel because the
        # weight-gradient slabs are reduced as one contiguous [kernel | bias] range.
        total, self.offsets = 0, {}
        for k, shp in self.shapes.items():
            n = int(np.prod(shp))
            if k.endswith(".kernel"):
                assert n % 4 == 0, k
            self.offsets[k] = total
            total += n if k.endswith(".kernel") else (n + 3) // 4 * 4
        self.n_params = sum(int(np.prod(shp)) for shp in self.shapes.values())
        dev = self.device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.velocity = torch.zeros(total, dtype=torch.float32, device=dev)
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._lr_host = None      # the value lr_dev holds (sgd_step skips the fill while the schedule keeps the rate); None = unknown
        self.packed = None
        if self.cn8:       # bf16 operand images of EVERY conv weight, both orientations, refreshed by one launch per forward
            pk = ops.PackedWeights()
            cin = in_channels
            for i, (f, s_, res) in enumerate(self.blocks):
                pre = "l%d." % i
                og, ot = self.offsets[pre + "gcn.kernel"], self.offsets[pre + "tcn.kernel"]
                pk.add(pre + "gcn.f", og, f, KS * f, 1, KS, cin, f)          # (k, c, m) = kernel[c][k*F + m]
                pk.add(pre + "gcn.b", og, f, 1, KS * f, KS, f, cin)          # (k, c', m') = kernel[m'][k*F + c']
                pk.add(pre + "tcn.f", ot, f * f, f, 1, KT, f, f)             # (tap, c, m) = kernel[tap][c][m]
                pk.add(pre + "tcn.b", ot, f * f, 1, f, KT, f, f)             # (tap, c', m') = kernel[tap][m'][c']
                if self.kinds[i] == "conv":
                    orr = self.offsets[pre + "res.kernel"]
                    pk.add(pre + "res.f", orr, 0, f, 1, 1, cin, f)
                    pk.add(pre + "res.b", orr, 0, 1, f, 1, f, cin)
                cin = f
            pk.finalize(dev)
            self.packed = pk
        if self.bf16:      # bf16 operand images of every conv weight, both orientations, refreshed by one launch per forward
            pk = ops.PackedWeights()
            cin = in_channels
            for i, (f, s_, res) in enumerate(self.blocks):
                pre = "l%d." % i
                og, ot = self.offsets[pre + "gcn.kernel"], self.offsets[pre + "tcn.kernel"]
                if f % 8 == 0 and cin >= 16:
                    pk.add(pre + "gcn.f", og, f, KS * f, 1, KS, cin, f)          # (k, c, m) = kernel[c][k*F + m]
                if cin % 8 == 0 and f >= 16:
                    pk.add(pre + "gcn.b", og, f, 1, KS * f, KS, f, cin)          # (k, c', m') = kernel[m'][k*F + c']
                if f % 8 == 0 and f >= 16:
                    pk.add(pre + "tcn.f", ot, f * f, f, 1, KT, f, f)             # (tap, c, m) = kernel[tap][c][m]
                    pk.add(pre + "tcn.b", ot, f * f, 1, f, KT, f, f)             # (tap, c', m') = kernel[tap][m'][c']
                if self.kinds[i] == "conv":
                    orr = self.offsets[pre + "res.kernel"]
                    if f % 8 == 0 and cin >= 16:
                        pk.add(pre + "res.f", orr, 0, f, 1, 1, cin, f)
                    if cin % 8 == 0 and f >= 16:
                        pk.add(pre + "res.b", orr, 0, 1, f, 1, f, cin)
                cin = f
            if pk.items:
                pk.finalize(dev)
                self.packed = pk
        self.spacked, self._cells = None, None
        if self.split:     # term images of the conv weights the split kernels take, both orientations, one refresh per step
            pk = ops.PackedSplitWeights(self.split)
            cin = in_channels
            for i, (f, s_, res) in enumerate(self.blocks):
                pre = "l%d." % i
                og, ot = self.offsets[pre + "gcn.kernel"], self.offsets[pre + "tcn.kernel"]
                if ops.split_applicable(L.SAR_CONV_GRAPH, num_node, cin, f, KS, 1, self.tab_fwd):
                    pk.add(pre + "gcn.f", og, f, KS * f, 1, KS, cin, f)          # (k, c, m) = kernel[c][k*F + m]
                if ops.split_applicable(L.SAR_CONV_GRAPH, num_node, f, cin, KS, 1, self.tab_bwd):
                    pk.add(pre + "gcn.b", og, f, 1, KS * f, KS, f, cin)          # (k, c', m') = kernel[m'][k*F + c']
                if self.kinds[i] == "conv" and self.split in ("f16x3a", "bf16x6"):
                    # the strided 1x1 residual convolution and the dense 1x1 product of its data gradient (conv_tap1_split_kernel, round 6)
                    orr = self.offsets[pre + "res.kernel"]
                    if ops.split_applicable(L.SAR_CONV_TEMPORAL, num_node, cin, f, 1, s_):
                        pk.add(pre + "res.f", orr, 0, f, 1, 1, cin, f)           # (0, c, m) = kernel[c][m]
                    if s_ == 2 and ops.split_applicable(L.SAR_CONV_TEMPORAL, num_node, f, cin, 1, 1):
                        pk.add(pre + "res.b", orr, 0, 1, f, 1, f, cin)           # (0, c', m') = kernel[m'][c']
                cin = f
                if ops.split_applicable(L.SAR_CONV_TEMPORAL, num_node, f, f, KT, s_):
                    pk.add(pre + "tcn.f", ot, f * f, f, 1, KT, f, f)             # (tap, c, m) = kernel[tap][c][m]
                    pk.add(pre + "tcn.b", ot, f * f, 1, f, KT, f, f)             # (tap, c', m') = kernel[tap][m'][c']
            if pk.items:
                pk.finalize(dev)
                self.spacked = pk
            # operand bounds of the fp16 arithmetic (include/sar_hip.h: cells), zeroed at the start of every training step
            self._cells = torch.zeros(8 * len(self.blocks), dtype=torch.int32, device=dev)
        # fp32 operands of the data-gradient GEMMs ((tap, f, c) / (k*F + f, c) / (f, c) transposes of the kernels): ONE
        # re-layout launch at the start of backward() instead of one small dependent launch in front of every data gradient
        # (22 per step, each on the critical chain)
        self._wT_off, self._wT_perm, self._wT = {}, None, None
        if not self.cn8:
            pb, off, cin = ops.PermuteBatch(), 0, in_channels
            for i, (f, s_, res) in enumerate(self.blocks):
                pre = "l%d." % i
                pb.add(self.offsets[pre + "tcn.kernel"], off, KT, f, f, f * f, 1, f)        # [tap][c][f] -> [tap][f][c]
                self._wT_off[pre + "tcn"], off = off, off + KT * f * f
                pb.add(self.offsets[pre + "gcn.kernel"], off, 1, KS * f, cin, 0, 1, KS * f)  # [c][k*F+f] -> [k*F+f][c]
                self._wT_off[pre + "gcn"], off = off, off + KS * f * cin
                if self.kinds[i] == "conv":
                    pb.add(self.offsets[pre + "res.kernel"], off, 1, f, cin, 0, 1, f)       # [c][f] -> [f][c]
                    self._wT_off[pre + "res"], off = off, off + f * cin
                cin = f
            pb.finalize(dev)
            self._wT_perm, self._wT = pb, torch.zeros(off, dtype=torch.float32, device=dev)
        self.p = {k: self._view(self.flat, k) for k in self.shapes}
        self.g = {k: self._view(self.grad, k) for k in self.shapes}
        if self.dense_A:
            self.A = self.p["adjacency_matrix"]          # the trainable copy (initialised from the graph in _init_params)
        self.bn = {"data_bn": _BN(nch, dev)}
        for i, (f, s, res) in enumerate(self.blocks):
            self.bn["l%d.bn1" % i], self.bn["l%d.bn2" % i] = _BN(f, dev), _BN(f, dev)
            if self.kinds[i] == "conv":
                self.bn["l%d.res_bn" % i] = _BN(f, dev)
        self._init_params(seed)
        self._saved = None
        self._deferred, self._flushing = [], False
        self._buckets = self._make_buckets(total)

    # ------------------------------------------------------------------ gradient buckets (data-parallel exchange)
    def _make_buckets(self, total):
        """Contiguous slices of the flat gradient buffer in the order backward() completes them (main_gnn.py:234,239: the
        all-reduce inside apply_gradients).  Parameters are laid out data_bn, l0 .. l9, logits(, adjacency_matrix); backward
        produces logits, l9 .. l0, data_bn(, adjacency): bucket 0 = [first block of the last stage .. logits], one bucket per
        earlier stage boundary (a stride-2 block), the last bucket = [data_bn .. end of the first stage]; a trainable
        adjacency is summed over the blocks at the very end and forms its own slice.  Each entry: (block index after whose
        backward the slice is complete, or -1 = at the end of backward; lo; hi)."""
        nb = len(self.blocks)
        starts = [i for i, (f, s, res) in enumerate(self.blocks) if s != 1 and i > 0]     # a new stage begins here
        hi = self.offsets.get("adjacency_matrix", total)
        out = []
        for i in reversed(starts):
            lo = min(o for k, o in self.offsets.items() if k.startswith("l%d." % i))
            if lo < hi:
                out.append((i, lo, hi))
                hi = lo
        out.append((-1, 0, hi))
        if "adjacency_matrix" in self.offsets:
            out.append((-1, self.offsets["adjacency_matrix"], total))
        assert sum(h - l for _, l, h in out) == total and nb > 0
        return out

    def _bucket_done(self, bi, cb):
        """Every gradient of bucket bi has been ISSUED: weight gradients on the side stream, BatchNorm / bias gradients on the
        main stream.  cb(bi, flat slice, events) may start the slice's all-reduce once the events have completed."""
        _, lo, hi = self._buckets[bi]
        events = []
        self._flush_slabs()
        if self._side is not None:
            ev = torch.cuda.Event()
            ev.record(self._side)
            events.append(ev)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        events.append(ev)
        cb(bi, self.grad[lo:hi], events)

    def _flush_slabs(self):
        """the slabs of every weight gradient issued since the last flush are summed by ONE launch on the weight-gradient stream
        (ops.SlabBatch): before a bucket is handed to the all-reduce and at the end of backward()"""
        slabs = getattr(self, "_slabs", None)
        if slabs is None:
            return
        if self._side is None:
            slabs.flush()
        else:
            with torch.cuda.stream(self._side):
                slabs.flush()

    def _buckets_after_block(self, i, cb):
        if cb is not None:
            for bi, (blk, _, _) in enumerate(self._buckets):
                if blk == i:
                    self._bucket_done(bi, cb)
        elif getattr(self, "_slabs", None) is not None and (
                self._slab_flush == "block" or (self._slab_flush == "bucket" and any(blk == i for blk, _, _ in self._buckets))):
            self._flush_slabs()      # (experiment switch SAR_SLAB_FLUSH: more, smaller slab reductions than one at the end of backward)

    def _finish_backward(self, cb):
        """common tail of backward(): nothing deferred is left behind, the side stream is joined, the remaining buckets go"""
        if self._deferred:
            self._flush_deferred()
        if cb is not None:
            self._buckets_after_block(-1, cb)
        self._flush_slabs()
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)      # every weight gradient is in self.grad
        self._saved = None

    # ------------------------------------------------------------------ parameters
    def _add(self, name, shape):
        self.shapes[name] = tuple(shape)

    def _view(self, flat, name):
        o = self.offsets[name]
        return flat[o:o + int(np.prod(self.shapes[name]))].view(self.shapes[name])

    def _init_params(self, seed):
        """models/stgcn.py:7-8 VarianceScaling(2, fan_out, truncated_normal); biases 0; BN gamma 1, beta 0."""
        gen = torch.Generator().manual_seed(seed)
        for k, shp in self.shapes.items():
            if k == "adjacency_matrix":
                self.p[k].copy_(torch.from_numpy(self.A_host))
            elif k.endswith(".kernel"):
                fan_out = shp[0] * shp[1] * shp[3]
                std = math.sqrt(2.0 / fan_out) / .87962566103423978
                w = torch.empty(shp, dtype=torch.float64)
                torch.nn.init.trunc_normal_(w, 0.0, std, -2 * std, 2 * std, generator=gen)
                self.p[k].copy_(w.to(torch.float32))
            elif k.endswith(".gamma"):
                self.p[k].fill_(1.0)
            else:
                self.p[k].zero_()

    def load_params(self, params):
        """params: dict name -> tensor in the oracle / Keras layouts (incl. optional moving stats, 'A' ignored)."""
        self._lr_host = None      # (a restored engine re-writes the device-side learning rate at its next step)
        for k, v in params.items():
            if k in self.p:
                self.p[k].copy_(v.to(torch.float32).reshape(self.shapes[k]))
            elif k.endswith(".moving_mean"):
                self.bn[k[:-len(".moving_mean")]].moving_mean.copy_(v.to(torch.float32))
            elif k.endswith(".moving_var"):
                self.bn[k[:-len(".moving_var")]].moving_var.copy_(v.to(torch.float32))

    def state_dict(self):
        out = {k: v.detach().cpu().clone() for k, v in self.p.items()}
        for k, b in self.bn.items():
            out[k + ".moving_mean"] = b.moving_mean.cpu().clone()
            out[k + ".moving_var"] = b.moving_var.cpu().clone()
        out["A"] = self.A.cpu().clone()
        return out

    def grads(self):
        return {k: v for k, v in self.g.items()}

    # ------------------------------------------------------------------ forward
    def _bn_forward_stats(self, name, partials, nparts, count, training, unbiased):
        b = self.bn[name]
        ops.bn_finalize(partials, nparts, b.mean.numel(), count, BN_EPS, BN_MOMENTUM, unbiased, self.p[name + ".gamma"],
                        self.p[name + ".beta"], b.moving_mean if training else None, b.moving_var if training else None,
                        b.mean, b.rstd, b.scale, b.shift)

    def _bn_eval(self, name):
        b = self.bn[name]
        ops.bn_eval_affine(self.p[name + ".gamma"], self.p[name + ".beta"], b.moving_mean, b.moving_var, BN_EPS, b.scale,
                           b.shift)

    def forward(self, x, training=True, keep=None):
        """x: (N, C, T, V, M) float32 cuda -> logits (N, classes).  keep: optional dict receiving
        intermediate activations (tests)."""
        if self.cn8:
            from . import stgcn8
            return stgcn8.forward(self, x, training, keep)
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 5
        x = x.contiguous()
        N, Cin, T, V, M = x.shape
        assert Cin == self.C_in and V == self.V
        dev, B = x.device, N * M
        saved = {"x": x, "N": N, "M": M, "T": T, "blocks": [], "training": training}
        if self.packed is not None:
            self.packed.refresh(self.flat)       # the parameters may have changed since the last step
        self._bounds_forked = False
        if self.spacked is not None and training:
            if self._aux is not None:
                # term images, bound cells and the Samuelson cells (functions of gamma / beta and the sample counts) beside the data_bn
                # stage and the 3-channel layer: joined in front of the first launch that reads an image or a cell (_join_aux)
                self._aux.wait_stream(torch.cuda.current_stream())      # behind the optimizer step that produced self.flat
                with torch.cuda.stream(self._aux):
                    self.spacked.refresh(self.flat)
                    self._cells.zero_()
                    Tb = T
                    for bi_, (f_, s_, _r) in enumerate(self.blocks):
                        if self._cell_live("l%d.tcn.f" % bi_, "tfwd", "twgrad"):
                            ops.bn_bound(self.p["l%d.bn1.gamma" % bi_], self.p["l%d.bn1.beta" % bi_], B * Tb * V, self._cell(bi_, 0))
                        Tb = same_pad(Tb, KT, s_)[0]
                self._aux_pending = self._bounds_forked = True
            else:
                self.spacked.refresh(self.flat)
                self._cells.zero_()
        # ---- data_bn (models/stgcn.py:142-147)
        nch = V * Cin
        if training:
            part = torch.empty((nch, N, 2), dtype=torch.float32, device=dev)
            ops.data_bn_stats(x, self.bone_parent, part, self.motion)
            self._bn_forward_stats("data_bn", part, N, N * M * T, True, False)
        else:
            self._bn_eval("data_bn")
        dbn = self.bn["data_bn"]
        h = torch.empty((Cin, B * T * V), dtype=torch.float32, device=dev)
        ops.data_bn_apply(x, self.bone_parent, dbn.scale, dbn.shift, h, self.motion)
        if keep is not None:
            keep["x0"] = h
        Tc, cin = T, Cin
        for i, (f, s, res) in enumerate(self.blocks):
            h, Tc = self._block_forward(i, h, cin, f, s, B, Tc, training, saved, keep)
            cin = f
        # ---- head (models/stgcn.py:153-158)
        feat = torch.empty((N, cin), dtype=torch.float32, device=dev)
        ops.pool_fwd(h, B, Tc * V, M, feat)
        logits = torch.empty((N, self.num_classes), dtype=torch.float32, device=dev)
        ops.fc_fwd(feat, self.p["logits.kernel"].view(cin, self.num_classes), self.p["logits.bias"], logits)
        saved.update(feat=feat, T_last=Tc, y_last_shape=(cin, B * Tc * V))
        self._saved = saved if training else None
        if keep is not None:
            keep["feat"] = feat
        return logits

    def _block_forward(self, i, X, cin, f, s, B, T, training, saved, keep):
        V, dev = self.V, X.device
        pre = "l%d." % i
        kind = self.kinds[i]
        To, pad, _ = same_pad(T, KT, s)
        n_in, n_out = B * T * V, B * To * V
        epi = L.SAR_EPI_STATS if training else L.SAR_EPI_NONE
        # sgcn: GraphConvTD (models/gcn.py:199-209)
        g = torch.empty((f, n_in), dtype=torch.float32, device=dev)
        y3 = None
        if self.dense_A:   # models/gcn.py:229-237: Conv2D(3F, 1x1) then einsum 'nkctv,kvw->nctw' with the trainable A
            y3 = torch.empty((KS * f, n_in), dtype=torch.float32, device=dev)
            ops.conv_gemm(L.SAR_CONV_TEMPORAL, X, y3, self.p[pre + "gcn.kernel"], 0, KS * f, B=B, V=V, T_src=T, T_out=T, Kc=cin,
                          M=KS * f, taps=1, stride=1, pad=0, bias=self.p[pre + "gcn.bias"])
            r1 = ops.graph_dense_fwd(y3, self.A, g, KS, f, V, B * T, stats=training)
        else:
            gimg = self._simg(pre + "gcn.f") if training else None
            if gimg is not None or (training and self._cell_live(pre + "gcn.f", "gfwd", "gwgrad")):
                self._join_aux()                    # first reader of a term image / first writer of a cell on the main chain
            if training and i == 0 and self._cell_live(pre + "gcn.f", "gfwd", "gwgrad"):
                ops.amax(X, self._cell(i, 3))       # the bound of the block input (blocks > 0: written by the previous block's tail)
            r1 = ops.conv_gemm(L.SAR_CONV_GRAPH, X, g, self.p[pre + "gcn.kernel"], f, KS * f, B=B, V=V, T_src=T, T_out=T,
                               Kc=cin, M=f, taps=KS, bias=self.p[pre + "gcn.bias"], tables=self.tab_fwd, epi=epi,
                               **self._split_args(gimg, self._cell(i, 3), self._img(pre + "gcn.f")))
        if training:
            self._bn_forward_stats(pre + "bn1", r1[0], r1[1], n_in, True, True)
        else:
            self._bn_eval(pre + "bn1")
        bn1 = self.bn[pre + "bn1"]
        # tgcn: BN -> ReLU folded into the operand load, Conv2D [9,1] stride s SAME (models/stgcn.py:26-36)
        u = torch.empty((f, n_out), dtype=torch.float32, device=dev)
        simg = self._simg(pre + "tcn.f") if training else None        # inference keeps the fp32 kernels (no batch statistics to bound with)
        self._join_aux()           # (the 3-channel layer's graph convolution above reads neither images nor cells: the fork ends here)
        if training and self._cell_live(pre + "tcn.f", "tfwd", "twgrad") and not self._bounds_forked:     # (a cell is raised when ANY split consumer of it is on)
            ops.bn_bound(self.p[pre + "bn1.gamma"], self.p[pre + "bn1.beta"], n_in, self._cell(i, 0))
        r2 = ops.conv_gemm(L.SAR_CONV_TEMPORAL, g, u, self.p[pre + "tcn.kernel"], f * f, f, B=B, V=V, T_src=T, T_out=To,
                           Kc=f, M=f, taps=KT, stride=s, pad=pad, bias=self.p[pre + "tcn.bias"],
                           pro=(bn1.scale, bn1.shift), pro_relu=True, epi=epi,
                           **self._split_args(simg, self._cell(i, 0), self._img(pre + "tcn.f")))
        if training:
            self._bn_forward_stats(pre + "bn2", r2[0], r2[1], n_out, True, True)
        else:
            self._bn_eval(pre + "bn2")
        bn2 = self.bn[pre + "bn2"]
        r = None
        rbn = None
        if kind == "conv":  # models/stgcn.py:47-56
            r = torch.empty((f, n_out), dtype=torch.float32, device=dev)
            rsimg = self._simg(pre + "res.f") if training else None     # split arithmetic: X's bound is cell 3 (raised by the block below)
            r3 = ops.conv_gemm(L.SAR_CONV_TEMPORAL, X, r, self.p[pre + "res.kernel"], 0, f, B=B, V=V, T_src=T, T_out=To,
                               Kc=cin, M=f, taps=1, stride=s, pad=0, bias=self.p[pre + "res.bias"], epi=epi,
                               **self._split_args(rsimg, self._cell(i, 3), self._img(pre + "res.f")))
            if training:
                self._bn_forward_stats(pre + "res_bn", r3[0], r3[1], n_out, True, True)
            else:
                self._bn_eval(pre + "res_bn")
            rbn = self.bn[pre + "res_bn"]
        y = torch.empty((f, n_out), dtype=torch.float32, device=dev)
        res_kind = {"none": 0, "identity": 1, "conv": 2}[kind]
        ymask = ops.relu_mask(y) if training else None     # 1 bit per element: what the BatchNorm-backward passes read instead of y
        # (split arithmetic: y is the next block's graph-convolution operand; its bound is a by-product of this pass)
        ycell = self._cell(i + 1, 3) if (training and i + 1 < len(self.blocks)
                                         and (self._cell_live("l%d.gcn.f" % (i + 1), "gfwd", "gwgrad")
                                              or self._cell_live("l%d.res.f" % (i + 1), "rfwd", "rwgrad"))) else None
        ops.bn_add_relu_fwd(u, bn2.scale, bn2.shift, res_kind, X if kind == "identity" else r,
                            rbn.scale if rbn else None, rbn.shift if rbn else None, y, mask=ymask, amax_cell=ycell)
        if training:
            saved["blocks"].append(dict(X=X, g=g, u=u, r=r, y=y, ymask=ymask, T=T, To=To, pad=pad, cin=cin, f=f, s=s, kind=kind, y3=y3))
        if keep is not None:
            keep[pre + "g"], keep[pre + "u"], keep[pre + "y"] = g, u, y
        return y, To

    def _join_aux(self):
        """the main chain waits for what forward() forked onto the third stream (once per step)"""
        if self._aux_pending:
            torch.cuda.current_stream().wait_stream(self._aux)
            self._aux_pending = False

    def _img(self, key):
        """packed bf16 operand image of a conv weight (bf16 mode), else None"""
        return self.packed.image(key) if self.packed is not None and key in self.packed.index else None

    # ---- split arithmetic (mfma="f32_split*")
    @property
    def _f16(self):
        return bool(self.split) and self.split.startswith("f16")

    def _simg(self, key):
        """(term images, w_bound cell) of a conv weight in split mode, else None"""
        pk = self.spacked
        kind = {"tcn.f": "tfwd", "tcn.b": "tdgrad", "gcn.f": "gfwd", "gcn.b": "gdgrad", "res.f": "rfwd", "res.b": "rdgrad"}[key.split(".", 1)[1]]
        if kind not in _SPLIT_KINDS:
            return None
        return (pk.image(key), pk.bound(key)) if pk is not None and key in pk.index else None

    def _cell_live(self, key, *kinds):
        """does operand-bound cell of weight `key`'s layer have a split consumer that is switched on?  A cell is raised by its
        producer whenever ANY of its consumers runs on the fp16 split kernels -- the forward launch or a weight gradient (ADVICE r05:
        with SAR_SPLIT_KINDS=tdgrad,twgrad the weight gradient read a source bound that only the forward launch used to raise)."""
        return bool(self._f16 and self._split_has(key) and any(k in _SPLIT_KINDS for k in kinds))

    def _split_has(self, key):
        """the split kernels take this conv weight's shape (its term images exist), whatever SAR_SPLIT_KINDS switches on"""
        return self.spacked is not None and key in self.spacked.index

    def _cell(self, i, j):
        """operand-bound cell j of block i: 0 relu(bn1(g)), 1 du, 2 dg, 3 X, 4 dr"""
        return self._cells[8 * i + j:8 * i + j + 1] if self._cells is not None else None

    def _split_args(self, simg, src_cell, packed=None):
        """the arithmetic keyword arguments of ops.conv_gemm: a launch on the split kernels (simg from _simg), else the fp32 /
        bf16-operand kernel with its `packed` image"""
        if simg is None:
            return dict(split=None, bf16=self.bf16, packed=packed)
        return dict(split=self.split, packed=simg[0], bounds=(src_cell, simg[1]))

    # ------------------------------------------------------------------ backward
    def _off_critical_path(self, fn, *tensors):
        """Weight-gradient kernels feed nothing but the optimizer: they run on a second stream, ordered after
        everything issued so far, so that the MFMA-bound reductions overlap the HBM-bound BatchNorm / ReLU passes of
        the main chain.  `tensors` are the inputs whose memory the caching allocator must not hand out again before
        the side stream is done with them.  On by default (SAR_WGRAD_STREAM=0 turns it off): measured +3.5 % clips/s at
        bs = 64 in fp32 -- the MFMA kernels fill the register file, so only part of the element-wise work can co-reside.
        Per-kernel timings of overlapping kernels are inflated; bench.py therefore also reports the dominant family
        measured in a short pass with the side stream off (roofline.isolated)."""
        if self._side is None:
            fn()
            return
        if self._wgrad_defer() and not self._flushing:   # see _WGRAD_DEFER_ENV above
            self._deferred.append((fn, tensors))
            return
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            fn()
        for t in tensors:
            if t is not None:
                t.record_stream(self._side)

    def _wgrad_defer(self):
        return (not self.cn8) if _WGRAD_DEFER_ENV is None else _WGRAD_DEFER_ENV == "1"

    def _flush_deferred(self):
        q, self._deferred = self._deferred, []
        self._flushing = True
        try:
            for fn, tensors in q:
                self._off_critical_path(fn, *tensors)
        finally:
            self._flushing = False

    def backward(self, dlogits, bucket_cb=None):
        """dlogits (N, classes) -> fills self.grad (every trainable parameter).  main_gnn.py:233.  bucket_cb: see
        _bucket_done (the data-parallel exchange starts per bucket while backward is still running)."""
        if self.cn8:
            from . import stgcn8
            return stgcn8.backward(self, dlogits, bucket_cb)
        sv = self._saved
        assert sv is not None, "backward() needs a preceding forward(training=True)"
        dev, V = dlogits.device, self.V
        if self._wT_perm is not None:
            self._wT_perm.run(self.flat, self._wT)      # every data-gradient operand of this step
        N, M = sv["N"], sv["M"]
        B = N * M
        c_last = self.C_last
        feat = sv["feat"]
        dfeat = torch.empty_like(feat)
        ops.fc_bwd(feat, self.p["logits.kernel"].view(c_last, self.num_classes), dlogits.contiguous(),
                   self.g["logits.kernel"].view(c_last, self.num_classes), self.g["logits.bias"], dfeat)
        dY = torch.empty(sv["y_last_shape"], dtype=torch.float32, device=dev)
        ops.pool_bwd(dfeat, B, sv["T_last"] * V, M, dY)
        if self.dense_A:
            self._dA_layers = torch.zeros((len(self.blocks), KS * V * V), dtype=torch.float32, device=dev)
        # fp32 engine, gather tables: the graph data gradient of block i gates the output gradient of block i - 1 with that block's
        # ReLU mask and reduces its BatchNorm-backward sums in the same epilogue (SAR_EPI_ADD_GATE; sar_amd/stgcn8.py does the same
        # for the bf16 engine) -- the bn_add_relu_bwd_reduce pass and the masked-gradient write of the apply pass are gone for
        # every block whose successor's skip path is not a convolution
        fuse = self._fuse_tail_f32()
        gated = None
        for i in reversed(range(len(self.blocks))):
            if fuse:
                sbb = sv["blocks"][i - 1] if i >= 1 else None
                below = sbb if (sbb is not None and self.kinds[i - 1] != "conv" and sbb.get("ymask") is not None
                                and sv["blocks"][i]["kind"] != "none" and sv["blocks"][i]["cin"] % 8 == 0) else None
                dY, gated = self._block_backward(i, sv["blocks"][i], dY, B, gated, below)
            else:
                dY = self._block_backward(i, sv["blocks"][i], dY, B)
            if self._deferred:
                self._flush_deferred()
            self._buckets_after_block(i, bucket_cb)
        if self.dense_A:       # the adjacency is shared by all blocks: dA = sum over the layers (fixed order); zero while frozen
            n = KS * V * V
            ops.check(L.load().sar_slab_reduce_f32(ops.ptr(self._dA_layers), len(self.blocks), n, n,
                                                   ops.ptr(self.g["adjacency_matrix"]), ops.stream_ptr()), "sar_slab_reduce_f32")
            self._dA_layers = None
        # data_bn gamma/beta (the input needs no gradient)
        x = sv["x"]
        nch = V * self.C_in
        part = torch.empty((nch, N, 2), dtype=torch.float32, device=dev)
        dbn = self.bn["data_bn"]
        ops.data_bn_bwd_reduce(x, self.bone_parent, dY, dbn.mean, part, self.motion)
        ops.bn_bwd_finalize(part, N, N * 2, 2, 0, 1, nch, N * M * sv["T"], self.p["data_bn.gamma"], dbn.mean, dbn.rstd,
                            self.g["data_bn.gamma"], self.g["data_bn.beta"])
        self._finish_backward(bucket_cb)

    def _fuse_tail_f32(self):
        """SAR_EPI_ADD_GATE in the fp32 graph data gradient (SAR_F32_FUSE_TAIL=0 turns it off): plain ST-GCN engine, fp32 MFMA
        operands, gather tables, ReLU masks written by the forward tails"""
        return (_FUSE_TAIL_F32 and type(self)._block_backward is STGCN._block_backward and not self.bf16 and not self.dense_A
                and not getattr(self, "cn8", False) and ops.RELU_MASK and not ops.BN_TAIL)

    def _block_backward(self, i, sb, dY, B, gated=None, below=None):
        """gated: this block's BatchNorm-backward partial sums when dY already carries the ReLU gate (produced by the graph data
        gradient of the block above); below: the saved tensors of block i - 1 when THIS block's graph data gradient is to do the
        same for it.  With the fused tail on (backward()) the return value is (dX, sums-for-the-block-below), else dX."""
        fused_call = self._fuse_tail_f32()
        V, dev = self.V, dY.device
        pre = "l%d." % i
        X, g, u, r, y = sb["X"], sb["g"], sb["u"], sb["r"], sb["y"]
        T, To, pad, cin, f, s, kind = sb["T"], sb["To"], sb["pad"], sb["cin"], sb["f"], sb["s"], sb["kind"]
        n_in, n_out = B * T * V, B * To * V
        bn1, bn2 = self.bn[pre + "bn1"], self.bn[pre + "bn2"]
        rbn = self.bn.get(pre + "res_bn")
        # ---- tail: y = relu(bn2(u) + res)   (models/stgcn.py:37,62-63)
        rk = (rbn.k1, rbn.k2, rbn.k3) if kind == "conv" else None
        if gated is not None:    # the sums came with dY: (sum dz, sum dz (u - mean)) per channel and partial
            assert kind != "conv"
            ops.bn_bwd_finalize(gated[0], gated[1], gated[1] * 2, 2, 0, 1, f, n_out, self.p[pre + "bn2.gamma"], bn2.mean, bn2.rstd,
                                self.g[pre + "bn2.gamma"], self.g[pre + "bn2.beta"], bn2.k1, bn2.k2, bn2.k3)
        elif ops.BN_TAIL:      # the reduce kernel's last workgroup per channel finalises BN2 (and the residual BN): no launch between
            tail = ops.make_bn_tail(dev, n_out, self.p[pre + "bn2.gamma"], bn2, self.g[pre + "bn2.gamma"], self.g[pre + "bn2.beta"],
                                    *((self.p[pre + "res_bn.gamma"], rbn, self.g[pre + "res_bn.gamma"], self.g[pre + "res_bn.beta"])
                                      if kind == "conv" else ()))
            ops.bn_add_relu_bwd_reduce(dY, y, u, r if kind == "conv" else None, bn2.mean, rbn.mean if kind == "conv" else None,
                                       tail=tail)
        else:
            part, nparts = ops.bn_add_relu_bwd_reduce(dY, y, u, r if kind == "conv" else None, bn2.mean,
                                                      rbn.mean if kind == "conv" else None, mask=sb.get("ymask"))
            ops.bn_bwd_finalize(part, nparts, nparts * 4, 4, 0, 1, f, n_out, self.p[pre + "bn2.gamma"], bn2.mean, bn2.rstd,
                                self.g[pre + "bn2.gamma"], self.g[pre + "bn2.beta"], bn2.k1, bn2.k2, bn2.k3)
            if kind == "conv":
                ops.bn_bwd_finalize(part, nparts, nparts * 4, 4, 0, 2, f, n_out, self.p[pre + "res_bn.gamma"], rbn.mean,
                                    rbn.rstd, self.g[pre + "res_bn.gamma"], self.g[pre + "res_bn.beta"], rbn.k1, rbn.k2,
                                    rbn.k3)
        du = torch.empty_like(u)
        dr = torch.empty_like(r) if kind == "conv" else None
        dz = dY if (kind == "identity" and gated is None) else None  # in place: dY becomes the pre-ReLU gradient for the skip path (already gated: nothing to write)
        ducell = self._cell(i, 1) if self._cell_live(pre + "tcn.b", "tdgrad", "twgrad") else None   # the bound of du: by-product
        # (the bound of dr: operand of the residual branch's dense 1x1 data gradient on conv_tap1_split_kernel; by-product too)
        drcell = self._cell(i, 4) if (kind == "conv" and self._f16 and (
            (self._simg(pre + "res.b") is not None and self._compact_skip(s, T, i)) or self._res_wgrad_split(pre))) else None
        ops.bn_add_relu_bwd_apply(dY, y, u, r if kind == "conv" else None, (bn2.k1, bn2.k2, bn2.k3), rk, du, dr, dz,
                                  mask=sb.get("ymask"), amax_cell=ducell, amax_dr_cell=drcell)
        # ---- temporal conv: weight / bias gradient, then data gradient fused with ReLU-mask + BN1 reductions
        wt = self.g[pre + "tcn.kernel"]
        flat_w = self.grad[self.offsets[pre + "tcn.kernel"]:self.offsets[pre + "tcn.bias"] + f]
        simg = self._simg(pre + "tcn.b")
        self._off_critical_path(lambda: ops.conv_wgrad(
            L.SAR_CONV_TEMPORAL, g, du, flat_w, B=B, V=V, T_src=T, T_out=To, Kc=f, M=f, taps=KT, stride=s, pad=pad,
            pro=(bn1.scale, bn1.shift), pro_relu=True, w_stride_tap=f * f, w_stride_c=f, wsize=wt.numel(), bsize=f,
            bf16=self.bf16, split=self.split if ("twgrad" in _SPLIT_KINDS and self._split_has(pre + "tcn.f") and self._split_has(pre + "tcn.b")) else None,
            bounds=(self._cell(i, 0), self._cell(i, 1)) if self._f16 else None, slabs=self._slabs), g, du)
        wimg = self._img(pre + "tcn.b")
        wT = None
        if wimg is None:
            o = self._wT_off[pre + "tcn"]
            wT = self._wT[o:o + KT * f * f]                               # [tap][f][c], re-laid at the start of backward()
        dz1 = torch.empty((f, n_in), dtype=torch.float32, device=dev)
        pm = ops.conv_gemm(L.SAR_CONV_TEMPORAL, du, dz1, wT, f * f, f, B=B, V=V, T_src=To, T_out=T, Kc=f, M=f, taps=KT,
                           stride=s, pad=pad, transposed=True, epi=L.SAR_EPI_MASK, aux=g,
                           aux_affine=(bn1.scale, bn1.shift), aux_mean=bn1.mean,
                           **self._split_args(simg, self._cell(i, 1), wimg))
        ops.bn_bwd_finalize(pm[0], pm[1], pm[1] * 2, 2, 0, 1, f, n_in, self.p[pre + "bn1.gamma"], bn1.mean, bn1.rstd,
                            self.g[pre + "bn1.gamma"], self.g[pre + "bn1.beta"], bn1.k1, bn1.k2, bn1.k3)
        dg = dz1
        dgcell = self._cell(i, 2) if self._cell_live(pre + "gcn.b", "gdgrad", "gwgrad") else None      # the bound of dg: by-product
        ops.affine2(dz1, g, (bn1.k1, bn1.k2, bn1.k3), dg, amax_cell=dgcell)   # BN1 backward apply (in place)
        flat_g = self.grad[self.offsets[pre + "gcn.kernel"]:self.offsets[pre + "gcn.bias"] + KS * f]
        if self.dense_A:
            return self._graph_backward_dense(i, sb, dg, dY, dr, B, flat_g)
        # ---- graph conv: weight / bias gradient
        gw_split = self.split if ("gwgrad" in _SPLIT_KINDS and self._split_has(pre + "gcn.f")
                                  and self._split_has(pre + "gcn.b")) else None      # (both bounds exist: X from the forward, dg above)
        self._off_critical_path(lambda: ops.conv_wgrad(
            L.SAR_CONV_GRAPH, X, dg, flat_g, B=B, V=V, T_src=T, T_out=T, Kc=cin, M=f, taps=KS, tables=self.tab_fwd,
            w_stride_tap=f, w_stride_c=KS * f, wsize=cin * KS * f, bsize=KS * f, bf16=self.bf16, split=gw_split,
            bounds=(self._cell(i, 3), self._cell(i, 2)) if self._f16 else None, slabs=self._slabs), X, dg)
        dXres = self._residual_backward(i, sb, dr, B)
        # ---- graph conv data gradient (+ skip-path gradient)
        gimg = self._img(pre + "gcn.b")
        gT = None
        if gimg is None:
            o = self._wT_off[pre + "gcn"]
            gT = self._wT[o:o + KS * f * cin]                             # [k][f][c]
        dX = torch.empty((cin, n_in), dtype=torch.float32, device=dev)
        aux = dY if kind == "identity" else dXres
        even = kind == "conv" and aux is not None and self._compact_skip(s, T, i)   # dXres holds the even frames only
        sgimg = self._simg(pre + "gcn.b")
        if below is not None and aux is not None:
            # dX = gate_{i-1}(W^T dg . A^T + skip gradient) and block i - 1's BatchNorm-backward sums in one epilogue
            bn2b = self.bn["l%d.bn2" % (i - 1)]
            pm = ops.conv_gemm(L.SAR_CONV_GRAPH, dg, dX, gT, f * cin, cin, B=B, V=V, T_src=T, T_out=T, Kc=f, M=cin, taps=KS,
                               tables=self.tab_bwd, epi=L.SAR_EPI_ADD_GATE, aux=aux, aux2=below["u"], aux_mask=below["ymask"],
                               aux_mean=bn2b.mean, aux_even_frames=even, **self._split_args(sgimg, self._cell(i, 2), None))
            return dX, pm
        ops.conv_gemm(L.SAR_CONV_GRAPH, dg, dX, gT, f * cin, cin, B=B, V=V, T_src=T, T_out=T, Kc=f, M=cin, taps=KS,
                      tables=self.tab_bwd, epi=L.SAR_EPI_ADD if aux is not None else L.SAR_EPI_NONE, aux=aux, aux_even_frames=even,
                      **self._split_args(sgimg, self._cell(i, 2), gimg))
        return (dX, None) if fused_call else dX

    def _residual_backward(self, i, sb, dr, B):
        """weight gradient and data gradient of the strided 1x1 residual convolution (blocks 5 and 8); None otherwise"""
        if sb["kind"] != "conv":
            return None
        V, dev, pre = self.V, dr.device, "l%d." % i
        X, T, To, cin, f, s = sb["X"], sb["T"], sb["To"], sb["cin"], sb["f"], sb["s"]
        flat_r = self.grad[self.offsets[pre + "res.kernel"]:self.offsets[pre + "res.bias"] + f]
        rw_split = self._res_wgrad_split(pre) and T == s * To      # wgrad_tap1_split_kernel: bounds = X's cell 3, dr's cell 4
        self._off_critical_path(lambda: ops.conv_wgrad(
            L.SAR_CONV_TEMPORAL, X, dr, flat_r, B=B, V=V, T_src=T, T_out=To, Kc=cin, M=f, taps=1, stride=s, pad=0,
            w_stride_tap=0, w_stride_c=f, wsize=cin * f, bsize=f, slabs=self._slabs, split=self.split if rw_split else None,
            bounds=(self._cell(i, 3), self._cell(i, 4)) if (rw_split and self._f16) else None), X, dr)
        rimg = self._img(pre + "res.b")
        rT = None
        if rimg is None and pre + "res" in self._wT_off:
            o = self._wT_off[pre + "res"]
            rT = self._wT[o:o + f * cin]
        elif rimg is None:             # engines without the batched re-layout (ST-GIN)
            rT = torch.empty((f, cin), dtype=torch.float32, device=dev)
            ops.transpose(self.p[pre + "res.kernel"], rT, 1, cin, f)
        if self._compact_skip(s, T, i):
            # the gradient through a stride-2 1x1 convolution is non-zero on EVEN input frames only: a dense 1x1 product over the To
            # frames (half the matrix work of the strided data gradient, half the bytes written), added by the graph data gradient's
            # epilogue on even frames (SAR_GRAPH_AUX_EVEN_FRAMES) -- the zeros are neither written nor read back
            dXc = torch.empty((cin, B * To * V), dtype=torch.float32, device=dev)
            rb = self._simg(pre + "res.b")
            ops.conv_gemm(L.SAR_CONV_TEMPORAL, dr, dXc, rT, 0, cin, B=B, V=V, T_src=To, T_out=To, Kc=f, M=cin, taps=1, stride=1, pad=0,
                          **self._split_args(rb, self._cell(i, 4), None))
            return dXc
        dXres = torch.empty((cin, B * T * V), dtype=torch.float32, device=dev)
        ops.conv_gemm(L.SAR_CONV_TEMPORAL, dr, dXres, rT, 0, cin, B=B, V=V, T_src=To, T_out=T, Kc=f, M=cin, taps=1,
                      stride=s, pad=0, transposed=True, bf16=self.bf16, packed=rimg)
        return dXres

    def _res_wgrad_split(self, pre):
        """the residual 1x1 convolution's weight gradient on wgrad_tap1_split_kernel: the layer has its forward term images (the same
        shape test) and the kind is switched on"""
        return bool(self.split in ("f16x3a", "bf16x6") and "rwgrad" in _SPLIT_KINDS and ops.TAP1_SPLIT and self._split_has(pre + "res.f"))

    def _compact_skip(self, s, T, i=None):
        """the skip gradient of a conv-residual block as its even frames only (SAR_COMPACT_SKIP=0: the strided data gradient over all
        T frames): fp32-storage engines with gather tables and fp32 / split arithmetic.  Default ("fp32"): the fp32 engine, and the
        blocks of a split engine whose dense 1x1 product runs on conv_tap1_split_kernel (round 6)"""
        on = _COMPACT_SKIP == "1" or (_COMPACT_SKIP == "fp32" and (not self.split or (
            i is not None and self._simg("l%d.res.b" % i) is not None)))
        return (on and s == 2 and not self.bf16 and not getattr(self, "cn8", False)
                and not self.dense_A and type(self)._block_backward is STGCN._block_backward)

    def _graph_backward_dense(self, i, sb, dg, dY, dr, B, flat_g):
        """Backward of Conv2D(3F, 1x1) -> einsum with the trainable adjacency (models/gcn.py:229-237): dy3 = dg . A^T per
        slice, dA from (y3, dg), the 1x1 convolution's weight / bias / data gradients from (X, dy3)."""
        V, dev, pre = self.V, dg.device, "l%d." % i
        X, y3, T, cin, f, kind = sb["X"], sb["y3"], sb["T"], sb["cin"], sb["f"], sb["kind"]
        n_in = B * T * V
        dy3 = torch.empty_like(y3)
        ops.graph_dense_bwd_data(dg, self.A, dy3, KS, f, V, B * T)
        if self.train_adjacency:       # one V x V x K block per layer; summed over the layers at the end of backward()
            ops.graph_dense_dA(y3, dg, self._dA_layers[i], KS, f, V, B * T)
        self._off_critical_path(lambda: ops.conv_wgrad(
            L.SAR_CONV_TEMPORAL, X, dy3, flat_g, B=B, V=V, T_src=T, T_out=T, Kc=cin, M=KS * f, taps=1, stride=1, pad=0,
            w_stride_tap=0, w_stride_c=KS * f, wsize=cin * KS * f, bsize=KS * f, slabs=self._slabs), X, dy3)
        dXres = self._residual_backward(i, sb, dr, B)
        o = self._wT_off[pre + "gcn"]
        gT = self._wT[o:o + KS * f * cin]
        dX = torch.empty((cin, n_in), dtype=torch.float32, device=dev)
        aux = dY if kind == "identity" else dXres
        ops.conv_gemm(L.SAR_CONV_TEMPORAL, dy3, dX, gT, 0, cin, B=B, V=V, T_src=T, T_out=T, Kc=KS * f, M=cin, taps=1, stride=1,
                      pad=0, transposed=True, epi=L.SAR_EPI_ADD if aux is not None else L.SAR_EPI_NONE, aux=aux)
        return dX

    # ------------------------------------------------------------------ training step
    def loss_and_grad(self, x, labels, global_batch_size=None, bucket_cb=None):
        """main_gnn.py:221-233: loss = sum CE / global_batch; gradients of every trainable variable.  bucket_cb: backward()."""
        logits = self.forward(x, training=True)
        N = x.shape[0]
        gbs = global_batch_size or N
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dlogits = torch.empty_like(logits)
        ops.softmax_ce(logits, labels, 1.0 / gbs, loss, dlogits)
        self.backward(dlogits, bucket_cb)
        return logits, loss

    def sgd_step(self, lr, momentum=0.9):
        """tf.keras SGD(momentum, nesterov=True) over the flat buffers (main_gnn.py:312-314)."""
        if getattr(self, "_lr_host", None) != float(lr):      # one launch less per step while the schedule holds the rate
            self.lr_dev.fill_(float(lr))
            self._lr_host = float(lr)
        ops.sgd_nesterov(self.flat, self.velocity, self.grad, self.lr_dev, momentum)

    def predict(self, x):
        """main_gnn.py:205-208: softmax(model(features, training=False))."""
        logits = self.forward(x, training=False)
        probs = torch.empty_like(logits)
        labels = torch.zeros(x.shape[0], dtype=torch.int64, device=x.device)
        ops.softmax_ce(logits, labels, 1.0, None, None, probs)
        return probs
