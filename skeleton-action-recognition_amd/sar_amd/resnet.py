"""ResNet-18 (1 input channel, configurable width) training engine on the HIP kernels -- the spectrogram path's
classifier (reference models/resnet18.py:131-254: stem Conv7x7/2 + BN + ReLU + MaxPool3x3/2, four stages of two
BasicBlocks :26-72, AdaptiveAvgPool, Linear) and its train step (main_spectrogram.py:105-111,152-158: mean
CrossEntropyLoss, Adam).

Parameters keep the reference's state_dict names and torch shapes (conv OIHW, fc (classes, 512)) as views of one flat
fp32 buffer; a conv weight is STORED in the forward GEMM's operand order (tap, c, m) -- its OIHW tensor is a strided
view -- so the forward reads the master weights and the weight-gradient kernels write the gradient buffer in place;
only the (tap, m, c) layout of the data-gradient GEMMs is re-packed, once per step on the side stream.  Activations are CN matrices [C][B*H*W]; per BasicBlock the tensors crossing a BatchNorm barrier
(c1, c2, downsample out, block out) are materialised once; BN+ReLU is folded into the consumer conv's operand load,
BN statistics into the producer's epilogue, the block tail reuses the ST-GCN BN+add+ReLU kernels.
"""
import math

import numpy as np
import torch

from . import _lib as L
from . import ops
from .stgcn import _BN, SPLIT_ARITH

BN_EPS = 1e-5            # torch BatchNorm2d defaults (models/resnet18.py norm_layer = nn.BatchNorm2d)
BN_KEEP = 0.9            # running = 0.9*running + 0.1*batch  (torch momentum 0.1)
LAYERS = [2, 2, 2, 2]    # models/resnet18.py:274
# arithmetic of an engine built without an explicit `mfma` (the parity suites run once per value, tests/conftest.py):
# "fp32" = fp32 MFMA kernels; "f32_split" / "f32_split_bf16x6" = fp32 results on the fp16 / bf16 matrix pipe for the 3x3 / stride-1
# convolutions (csrc/conv2d_split.hip), everything else unchanged
DEFAULT_MFMA = __import__("os").environ.get("SAR_MFMA_PATHB", "fp32")
_SPLIT_KINDS = set(__import__("os").environ.get("SAR_SPLIT_KINDS_PATHB", "fwd,dgrad,wgrad").split(","))


class _Conv:
    def __init__(self, name, cin, cout, k, stride, pad):
        self.name, self.cin, self.cout, self.k, self.stride, self.pad = name, cin, cout, k, stride, pad
        self.taps = k * k


SLAB_BATCH_PATHB = __import__("os").environ.get("SAR_SLAB_BATCH_PATHB", "0") == "1"
# Fork / join of what does not depend on the main chain (round 6; SAR_PATHB_DS_STREAM=0: everything on the main stream again).  At bs = 32
# a step is a serial chain of ~150 launches of 4-50 us: the down-sampling branch of the three stride-2 blocks (1x1 convolution + its
# BatchNorm finalisation forward; its backward finalisation and the dense 1x1 product of its data gradient backward), the nine Samuelson
# bound cells and -- through prepack(), called by the trainer in front of the radar front-end -- the step's weight images run on a third
# stream and are joined where their results are needed.  Same launches, same order per tensor: bit-identical
# (tests/test_gpu_slab_batch.py); interleaved processes (tools/ab_pathb2.sh, profiles/r06_pathB_fork_ab.txt): fp32 4 976 -> 5 222 clips/s
# (+4.9 %), f32_split 7 098 -> 7 386 (+4.1 %)
DS_STREAM = __import__("os").environ.get("SAR_PATHB_DS_STREAM", "1") == "1"
# The BatchNorm-backward finalisation of a block tail (bn2 and the down-sampling BN) in the reduce kernel's last workgroup per channel
# (ops.BN_TAIL) instead of one or two launches behind it.  One bench.py process per entry (profiles/r06_pathB_fork_ab.txt):
# fp32 5 214 -> 5 220 clips/s (neutral), f32_split 7 208 -> 7 330 (+1.7 %): on for the split arithmetic ("split"; "1" / "0": always / never;
# SAR_BN_TAIL=1 still turns it on everywhere)
BN_TAIL_PATHB = __import__("os").environ.get("SAR_BN_TAIL_PATHB", "split")


class ResNet18:
    def __init__(self, num_classes=60, num_filters=64, device="cuda", seed=0, mfma=None):
        L.load()
        self.device = torch.device(device)
        self.mfma = mfma or DEFAULT_MFMA
        assert self.mfma == "fp32" or self.mfma in SPLIT_ARITH, self.mfma
        self.split = SPLIT_ARITH.get(self.mfma)
        self.num_classes, self.nf = num_classes, num_filters
        self.shapes, self.convs, self.bn_names, self.blocks = {}, {}, [], []

        def conv(name, cin, cout, k, stride):
            self.shapes[name + ".weight"] = (cout, cin, k, k)
            self.convs[name] = _Conv(name, cin, cout, k, stride, k // 2)

        def bn(name, c):
            self.shapes[name + ".weight"] = (c,)
            self.shapes[name + ".bias"] = (c,)
            self.bn_names.append((name, c))

        conv("conv1", 1, num_filters, 7, 2)
        bn("bn1", num_filters)
        inpl = num_filters
        for li, nb in enumerate(LAYERS):
            planes = num_filters * 2 ** li
            for bi in range(nb):
                pre = "layer%d.%d." % (li + 1, bi)
                stride = 2 if (li > 0 and bi == 0) else 1
                conv(pre + "conv1", inpl, planes, 3, stride)
                bn(pre + "bn1", planes)
                conv(pre + "conv2", planes, planes, 3, 1)
                bn(pre + "bn2", planes)
                ds = stride != 1 or inpl != planes
                if ds:
                    conv(pre + "downsample.0", inpl, planes, 1, stride)
                    bn(pre + "downsample.1", planes)
                self.blocks.append((pre, inpl, planes, stride, ds))
                inpl = planes
        self.c_last = inpl
        self.shapes["fc.weight"] = (num_classes, inpl)
        self.shapes["fc.bias"] = (num_classes,)
        total, self.offsets = 0, {}
        for k, shp in self.shapes.items():
            self.offsets[k] = total
            total += (int(np.prod(shp)) + 3) // 4 * 4
        self.n_params = sum(int(np.prod(s)) for s in self.shapes.values())
        dev = self.device
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat, self.grad, self.adam_m, self.adam_v = z(total), z(total), z(total), z(total)
        self.lr_dev, self.step_dev = z(1), z(1)
        self.p = {k: self._view(self.flat, k) for k in self.shapes}
        self.g = {k: self._view(self.grad, k) for k in self.shapes}
        self.bn = {name: _BN(c, dev) for name, c in self.bn_names}
        self._init_params(seed)
        self._saved = None
        # weight-gradient kernels feed only the optimizer: second stream, as in sar_amd/stgcn.py (SAR_WGRAD_STREAM=0: off)
        import os
        self._side = (ops.shared_side_stream(dev) if dev.type == "cuda" and os.environ.get("SAR_WGRAD_STREAM", "1") == "1"
                      else None)
        # one slab reduction per gradient bucket (ops.SlabBatch) -- OFF here: at bs = 32 the per-gradient reductions hide beside the
        # main chain, a batched one lands on the tail of backward (interleaved, profiles/r06_slab_batch_ab.txt: 4.62 -> 4.74 ms in
        # the split arithmetic, 6.40 -> 6.44 / 6.33 at the end / per stage in fp32); SAR_SLAB_BATCH_PATHB=1 turns it on
        self._slabs = ops.SlabBatch() if (ops.SLAB_BATCH and SLAB_BATCH_PATHB) else None
        self._slab_flush = ops.SLAB_FLUSH
        self._aux = ops.shared_aux_stream(dev) if (DS_STREAM and dev.type == "cuda") else None
        self._bounds_forked = False
        self._prepacked = None
        # side streams of the launches that fan out (stride-2 data gradients): owned by this engine, not by the library
        with torch.cuda.device(dev):
            self._ctx = L.Context() if dev.type == "cuda" else None
        # The data-gradient GEMMs read (tap, m, c): ONE re-layout launch per training step, on the side stream (the forward
        # reads the stored (tap, c, m) weights in place and the weight-gradient kernels write self.grad in place).
        self._woff, off = {}, 0
        pb = ops.PermuteBatch()
        for name, cv in self.convs.items():
            n = cv.taps * cv.cin * cv.cout
            src = self.offsets[name + ".weight"]
            self._woff[name] = (off, n)
            if name != "conv1":
                pb.add(src, off, cv.taps, cv.cout, cv.cin, cv.cin * cv.cout, 1, cv.cout)
            off += (n + 3) // 4 * 4
        pb.finalize(dev)
        self._perm_bwd = pb
        self._wb = z(off)
        # Split arithmetic: the term images of every 3x3 / stride-1 weight tensor -- forward view (tap, c, m) and data-gradient view
        # (mirrored taps, c and m exchanged: include/sar_hip.h sar_conv2d_gemm_split) -- written by ONE launch per step from the
        # master weights, and the bound cells of the source operands (zeroed once per step; raised by the producing passes)
        self.spacked, self._cells, self._cell_of = None, None, {}
        if self.split and dev.type == "cuda":
            sp = ops.PackedSplitWeights(self.split)
            for name, cv in self.convs.items():
                if cv.k == 3 and cv.stride == 1 and 8 <= cv.cin <= 512 and cv.cout % 8 == 0 and cv.cout <= 512:
                    src, cc = self.offsets[name + ".weight"], cv.cin * cv.cout
                    sp.add((name, "f"), src, cc, cv.cout, 1, 9, cv.cin, cv.cout)
                    sp.add((name, "b"), src + 8 * cc, -cc, 1, cv.cout, 9, cv.cout, cv.cin)
                    for kind in ("f", "b"):
                        self._cell_of[(name, kind)] = len(self._cell_of)
                elif cv.k == 3 and cv.stride == 2 and 8 <= cv.cin and cv.cin % 8 == 0 and 16 <= cv.cout <= 512:
                    # the stride-2 data gradient (four parity classes, csrc/conv2d_split.hip): the forward taps of the (tap, m, c) view
                    src, cc = self.offsets[name + ".weight"], cv.cin * cv.cout
                    sp.add((name, "b"), src, cc, 1, cv.cout, 9, cv.cout, cv.cin)
                    self._cell_of[(name, "b")] = len(self._cell_of)
            sp.finalize(dev)
            self.spacked = sp
            self._cells = torch.zeros(max(1, len(self._cell_of)), dtype=torch.int32, device=dev)
        # Gradient buckets for the data-parallel exchange (main_spectrogram.py:118-119), in the order backward() completes
        # them: [layer4 + fc], [layer3], [layer2], [conv1 + bn1 + layer1].  Each is a contiguous slice of the flat gradient
        # buffer (parameters are laid out in declaration order), so that a bucket can be all-reduced while the earlier layers are
        # still in backward.
        names = list(self.shapes)
        stage_of = lambda k: (4 if k.startswith(("layer4.", "fc.")) else 3 if k.startswith("layer3.") else
                              2 if k.startswith("layer2.") else 1)
        self._buckets = []
        for st in (4, 3, 2, 1):
            ks = [k for k in names if stage_of(k) == st]
            lo = min(self.offsets[k] for k in ks)
            hi = max(self.offsets[k] + (int(np.prod(self.shapes[k])) + 3) // 4 * 4 for k in ks)
            self._buckets.append(dict(stage=st, lo=lo, hi=hi))
        assert sorted((b["lo"], b["hi"]) for b in self._buckets)[0][0] == 0 and max(b["hi"] for b in self._buckets) == total

    def _view(self, flat, name):
        """The named tensor in torch's shape.  Conv weights are STORED in the forward GEMM's operand order (kh, kw, c, m)
        -- Adam is element-wise, so the optimizer does not care, the forward kernels read the master weights in place and
        the weight-gradient kernels write the gradient buffer in place: the OIHW tensor is a strided view of it."""
        o = self.offsets[name]
        shp = self.shapes[name]
        t = flat[o:o + int(np.prod(shp))]
        if len(shp) == 4:
            return t.view(shp[2], shp[3], shp[1], shp[0]).permute(3, 2, 0, 1)
        return t.view(shp)

    def _init_params(self, seed):
        """models/resnet18.py:187-194: kaiming_normal_(fan_out, relu) for convs, BN weight 1 / bias 0;
        nn.Linear default init for fc."""
        gen = torch.Generator().manual_seed(seed)
        for k, shp in self.shapes.items():
            if len(shp) == 4:
                fan_out = shp[0] * shp[2] * shp[3]
                self.p[k].copy_(torch.randn(shp, generator=gen) * math.sqrt(2.0 / fan_out))
            elif k == "fc.weight":
                bound = 1.0 / math.sqrt(shp[1])
                self.p[k].copy_((torch.rand(shp, generator=gen) * 2 - 1) * bound)
            elif k == "fc.bias":
                bound = 1.0 / math.sqrt(self.shapes["fc.weight"][1])
                self.p[k].copy_((torch.rand(shp, generator=gen) * 2 - 1) * bound)
            elif k.endswith(".weight"):
                self.p[k].fill_(1.0)
            else:
                self.p[k].zero_()

    def load_params(self, params):
        self._prepacked = None           # images issued by prepack() would be those of the old weights
        for k, v in params.items():
            if k in self.p:
                self.p[k].copy_(v.to(torch.float32).reshape(self.shapes[k]))
            elif k.endswith(".running_mean"):
                self.bn[k[:-13]].moving_mean.copy_(v.to(torch.float32))
            elif k.endswith(".running_var"):
                self.bn[k[:-12]].moving_var.copy_(v.to(torch.float32))

    def state_dict(self):
        out = {k: v.detach().cpu().clone() for k, v in self.p.items()}
        for k, b in self.bn.items():
            out[k + ".running_mean"] = b.moving_mean.cpu().clone()
            out[k + ".running_var"] = b.moving_var.cpu().clone()
        return out

    # ------------------------------------------------------------------ weights
    def _pack(self, need_bwd):
        """(tap, m, c) data-gradient layout of every conv weight: one launch per training step."""
        if need_bwd and self.spacked is not None:
            self.spacked.refresh(self.flat)
            self._cells.zero_()
        if need_bwd:            # only the backward pass reads the data-gradient layouts: off the forward's critical path
            if self._side is None:
                self._perm_bwd.run(self.flat, self._wb)
            else:
                main = torch.cuda.current_stream()
                self._side.wait_stream(main)                  # after the optimizer step that produced self.flat
                with torch.cuda.stream(self._side):
                    self._perm_bwd.run(self.flat, self._wb)
                self._wb_ready = torch.cuda.Event()
                self._wb_ready.record(self._side)

    def prepack(self, training=True):
        """The coming step's weight images (term images and amax cells of the split arithmetic, the data-gradient layouts) issued NOW
        on the third stream, so that they run beside whatever the caller issues next on the main stream -- the radar front-end, which
        does not read the resnet's weights (sar_amd/train.py).  forward() then joins instead of packing.  No-op without the stream."""
        if self._aux is None or self._prepacked is not None:
            return
        self._aux.wait_stream(torch.cuda.current_stream())      # behind the optimizer step that produced self.flat
        with torch.cuda.stream(self._aux):
            self._pack(training)
        self._prepacked = bool(training)

    def _w(self, name, bwd=False):
        if bwd:
            o, n = self._woff[name]
            return self._wb[o:o + n]
        o, n = self.offsets[name + ".weight"], self._woff[name][1]
        return self.flat[o:o + n]                      # the stored weights ARE the forward operand

    def _cell(self, name, kind):
        """bound cell of the source operand of conv `name`'s forward ("f") / data gradient ("b") on the split kernels, else None"""
        i = self._cell_of.get((name, kind)) if self._cells is not None else None
        return None if i is None else self._cells[i:i + 1]

    def _split_args(self, name, kind, training=True):
        """arithmetic keyword arguments of ops.conv2d_gemm for conv `name` (training steps only: inference keeps the fp32 kernels)"""
        want = {"f": "fwd", "b": "dgrad"}[kind] in _SPLIT_KINDS
        if not training or self.spacked is None or (name, kind) not in self._cell_of or not want:
            return dict(split=None)
        return dict(split=self.split, packed=self.spacked.image((name, kind)),
                    bounds=(self._cell(name, kind), self.spacked.bound((name, kind))))

    def _conv_fwd(self, name, X, B, H, W, training, pro=None, bn_src=None, out=None):
        """bn_src: (BatchNorm name, sample count) of the folded prologue -- the source bound of the split kernels is then the Samuelson
        bound of that train-mode BatchNorm (no pass over the data); without a prologue the producer of X has raised the cell."""
        cv = self.convs[name]
        Ho, Wo = (H + 2 * cv.pad - cv.k) // cv.stride + 1, (W + 2 * cv.pad - cv.k) // cv.stride + 1
        if out is None:
            out = torch.empty((cv.cout, B * Ho * Wo), dtype=torch.float32, device=X.device)
        assert out.shape == (cv.cout, B * Ho * Wo)
        sa = self._split_args(name, "f", training)
        # (raised whenever the cell exists: the forward launch AND the weight gradient read it -- ADVICE r05: with
        # SAR_SPLIT_KINDS_PATHB=dgrad,wgrad the weight gradient read a bound that only a split forward launch used to raise)
        if training and bn_src is not None and self._cell(name, "f") is not None and not self._bounds_forked:
            ops.bn_bound(self.p[bn_src[0] + ".weight"], self.p[bn_src[0] + ".bias"], bn_src[1], self._cell(name, "f"))
        r = ops.conv2d_gemm(X, out, self._w(name), cv.cin * cv.cout, cv.cout,
                            epi=L.SAR_EPI_STATS if training else L.SAR_EPI_NONE, B=B, Kc=cv.cin, M=cv.cout, H_src=H,
                            W_src=W, H_out=Ho, W_out=Wo, KH=cv.k, KW=cv.k, stride=cv.stride, pad=cv.pad, pro=pro,
                            pro_relu=pro is not None, **sa)
        return out, r, Ho, Wo

    def _all_bn_bounds(self, B, H, W):
        """the Samuelson bound cells of a training step -- the stem's (source of the first block) and every block's bn1 (folded into
        conv2's operand) -- from the geometry alone: what forward() raises layer by layer when they are not forked"""
        def osz(n, cv):
            return (n + 2 * cv.pad - cv.k) // cv.stride + 1
        cv = self.convs["conv1"]
        H1, W1 = osz(H, cv), osz(W, cv)
        first = self._cell(self.blocks[0][0] + "conv1", "f")
        if first is not None:
            ops.bn_bound(self.p["bn1.weight"], self.p["bn1.bias"], B * H1 * W1, first)
        Hc, Wc = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        for pre, inpl, planes, stride, ds in self.blocks:
            c1 = self.convs[pre + "conv1"]
            Ho, Wo = osz(Hc, c1), osz(Wc, c1)
            cell = self._cell(pre + "conv2", "f")
            if cell is not None:
                ops.bn_bound(self.p[pre + "bn1.weight"], self.p[pre + "bn1.bias"], B * Ho * Wo, cell)
            Hc, Wc = Ho, Wo

    def _bn_stats(self, name, r, count, training):
        b = self.bn[name]
        if training:
            ops.bn_finalize(r[0], r[1], b.mean.numel(), count, BN_EPS, BN_KEEP, True, self.p[name + ".weight"],
                            self.p[name + ".bias"], b.moving_mean, b.moving_var, b.mean, b.rstd, b.scale, b.shift)
        else:
            ops.bn_eval_affine(self.p[name + ".weight"], self.p[name + ".bias"], b.moving_mean, b.moving_var, BN_EPS,
                               b.scale, b.shift)
        return b

    # ------------------------------------------------------------------ forward
    def forward(self, x, training=True, keep=None):
        """x: (B, 1, H, W) float32 cuda -> logits (B, classes).  models/resnet18.py:235-251."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 1
        x = x.contiguous()
        B, _, H, W = x.shape
        dev = x.device
        # prepack(): the weight images were issued on the third stream by the trainer.  The stem (an fp32 7x7 kernel on the stored
        # weights) needs none of them: they are joined in front of the first block, so the pack runs beside the stem
        join_aux = self._prepacked is not None and self._prepacked == bool(training)
        if not join_aux:
            self._pack(training)
        self._prepacked = None
        self._bounds_forked = False
        if training and self._aux is not None and self._cells is not None:
            # every Samuelson bound of the step (they depend on gamma / beta and the sample counts only) on the third stream, beside
            # the stem: nine 4-us launches off the serial chain; joined in front of the first block
            if not join_aux:
                self._aux.wait_stream(torch.cuda.current_stream())  # behind the zeroing of the cells (prepacked: zeroed on that stream)
            with torch.cuda.stream(self._aux):
                self._all_bn_bounds(B, H, W)
            self._bounds_forked = join_aux = True
        X0 = x.view(1, B * H * W)
        c0, r, H1, W1 = self._conv_fwd("conv1", X0, B, H, W, training)
        bn0 = self._bn_stats("bn1", r, B * H1 * W1, training)
        H2, W2 = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        h = torch.empty((self.nf, B * H2 * W2), dtype=torch.float32, device=dev)
        ops.bn_relu_maxpool_fwd(c0, bn0.scale, bn0.shift, h, B, H1, W1)
        if keep is not None:
            keep["conv1"], keep["pool"] = c0, h
        saved = dict(x0=X0, c0=c0, B=B, H=H, W=W, H1=H1, W1=W1, H2=H2, W2=W2, blocks=[])
        Hc, Wc = H2, W2
        if join_aux:
            torch.cuda.current_stream().wait_stream(self._aux)
        if self._bounds_forked:
            pass
        elif training and self._cell(self.blocks[0][0] + "conv1", "f") is not None:
            # the stem tail is max-pool(relu(bn1(c0))): bounded by bn1's Samuelson bound
            ops.bn_bound(self.p["bn1.weight"], self.p["bn1.bias"], B * H1 * W1, self._cell(self.blocks[0][0] + "conv1", "f"))
        for bi_, (pre, inpl, planes, stride, ds) in enumerate(self.blocks):
            dsc = bd = None
            forked = ds and self._aux is not None
            if forked:       # the down-sampling branch on the third stream; its output lives in main-stream memory
                cd = self.convs[pre + "downsample.0"]
                Hd, Wd = (Hc + 2 * cd.pad - cd.k) // cd.stride + 1, (Wc + 2 * cd.pad - cd.k) // cd.stride + 1
                dsc = torch.empty((cd.cout, B * Hd * Wd), dtype=torch.float32, device=dev)
                self._aux.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._aux):
                    _, rd, _, _ = self._conv_fwd(pre + "downsample.0", h, B, Hc, Wc, training, out=dsc)
                    bd = self._bn_stats(pre + "downsample.1", rd, B * Hd * Wd, training)
                h.record_stream(self._aux)
            c1, r1, Ho, Wo = self._conv_fwd(pre + "conv1", h, B, Hc, Wc, training)
            b1 = self._bn_stats(pre + "bn1", r1, B * Ho * Wo, training)
            c2, r2, _, _ = self._conv_fwd(pre + "conv2", c1, B, Ho, Wo, training, pro=(b1.scale, b1.shift),
                                          bn_src=(pre + "bn1", B * Ho * Wo))
            b2 = self._bn_stats(pre + "bn2", r2, B * Ho * Wo, training)
            if ds and not forked:
                dsc, rd, _, _ = self._conv_fwd(pre + "downsample.0", h, B, Hc, Wc, training)
                bd = self._bn_stats(pre + "downsample.1", rd, B * Ho * Wo, training)
            if forked:
                torch.cuda.current_stream().wait_stream(self._aux)
            y = torch.empty_like(c2)
            ymask = ops.relu_mask(y) if training else None  # 1 bit per element: what the BatchNorm-backward passes read instead of y
            nxt = self.blocks[bi_ + 1][0] + "conv1" if bi_ + 1 < len(self.blocks) else None
            ops.bn_add_relu_fwd(c2, b2.scale, b2.shift, 2 if ds else 1, dsc if ds else h, bd.scale if ds else None,
                                bd.shift if ds else None, y, mask=ymask,
                                amax_cell=self._cell(nxt, "f") if (training and nxt) else None)   # the next block's source bound
            if training:
                saved["blocks"].append(dict(X=h, c1=c1, c2=c2, dsc=dsc, y=y, ymask=ymask, H=Hc, W=Wc, Ho=Ho, Wo=Wo))
            if keep is not None:
                keep[pre + "out"], keep[pre + "c1"] = y, c1
            h, Hc, Wc = y, Ho, Wo
        feat = torch.empty((B, self.c_last), dtype=torch.float32, device=dev)
        ops.pool_fwd(h, B, Hc * Wc, 1, feat)
        wt = torch.empty((self.c_last, self.num_classes), dtype=torch.float32, device=dev)
        ops.transpose(self.p["fc.weight"], wt, 1, self.num_classes, self.c_last)
        logits = torch.empty((B, self.num_classes), dtype=torch.float32, device=dev)
        ops.fc_fwd(feat, wt, self.p["fc.bias"], logits)
        saved.update(feat=feat, wt=wt, Hl=Hc, Wl=Wc)
        self._saved = saved if training else None
        return logits

    # ------------------------------------------------------------------ backward
    def _conv_wgrad(self, name, X, dout, B, H, W, Ho, Wo, pro=None):
        cv = self.convs[name]
        o, n = self.offsets[name + ".weight"], self._woff[name][1]

        sa = dict(split=None)      # split kernels: the source bound is the forward's cell, the dout bound the data gradient's
        if self.spacked is not None and (name, "f") in self._cell_of and "wgrad" in _SPLIT_KINDS:
            sa = dict(split=self.split, bounds=(self._cell(name, "f"), self._cell(name, "b")))

        def run():
            ops.conv2d_wgrad(X, dout, self.grad[o:o + n], B=B, Kc=cv.cin, M=cv.cout, H_src=H, W_src=W, H_out=Ho, W_out=Wo, KH=cv.k,
                             KW=cv.k, stride=cv.stride, pad=cv.pad, pro=pro, pro_relu=pro is not None, slabs=self._slabs, **sa)
        if self._side is None:
            run()
        else:       # ordered after everything issued so far; X / dout must outlive the side stream's use
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                run()
            X.record_stream(self._side)
            dout.record_stream(self._side)

    def _flush_slabs(self):
        """the slabs of every weight gradient issued since the last flush are summed by ONE launch on the weight-gradient stream
        (ops.SlabBatch): before a bucket is handed to the all-reduce and at the end of backward()"""
        if self._slabs is None:
            return
        if self._side is None:
            self._slabs.flush()
        else:
            with torch.cuda.stream(self._side):
                self._slabs.flush()

    def _conv_dgrad(self, name, dout, B, H, W, Ho, Wo, **epi):
        """gradient w.r.t. the conv's input (H, W) from dout at (Ho, Wo)."""
        cv = self.convs[name]
        dx = torch.empty((cv.cin, B * H * W), dtype=torch.float32, device=dout.device)
        r = ops.conv2d_gemm(dout, dx, self._w(name, True), cv.cout * cv.cin, cv.cin, B=B, Kc=cv.cout, M=cv.cin, H_src=Ho,
                            W_src=Wo, H_out=H, W_out=W, KH=cv.k, KW=cv.k, stride=cv.stride, pad=cv.pad, transposed=True,
                            ctx=self._ctx, **self._split_args(name, "b"), **epi)
        return dx, r

    def _bn_bwd(self, name, part, nparts, chan_stride, part_stride, off2, count):
        b = self.bn[name]
        ops.bn_bwd_finalize(part, nparts, chan_stride, part_stride, 0, off2, b.mean.numel(), count,
                            self.p[name + ".weight"], b.mean, b.rstd, self.g[name + ".weight"], self.g[name + ".bias"],
                            b.k1, b.k2, b.k3)
        return b

    def _bucket_done(self, bi, cb):
        """Every gradient of bucket bi has been ISSUED: its weight gradients on the side stream, its BatchNorm / fc gradients
        on the main stream.  cb(bi, flat slice, events) may start the
        slice's all-reduce as soon as the events have completed -- the main stream goes on with the earlier layers."""
        bk = self._buckets[bi]
        events = []
        self._flush_slabs()
        if self._side is not None:
            ev = torch.cuda.Event()
            ev.record(self._side)
            events.append(ev)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        events.append(ev)
        cb(bi, self.grad[bk["lo"]:bk["hi"]], events)

    def backward(self, dlogits, need_dx=False, bucket_cb=None):
        """Gradients of every parameter into self.g; need_dx additionally returns d loss / d image (B,1,H,W) -- only
        wanted when the image depends on trainable VirtualRadar parameters.  bucket_cb: see _bucket_done (data-parallel
        training: the gradient exchange overlaps the rest of backward)."""
        sv = self._saved
        assert sv is not None
        dev, B = dlogits.device, sv["B"]
        if self._side is not None and getattr(self, "_wb_ready", None) is not None:
            torch.cuda.current_stream().wait_event(self._wb_ready)      # the data-gradient weight layouts (packed on the side stream)
            self._wb_ready = None
        dwt = torch.empty_like(sv["wt"])
        dfeat = torch.empty_like(sv["feat"])
        dlogits = dlogits.contiguous()
        if self._aux is not None and bucket_cb is None:
            # the classifier's parameter gradients feed nothing but the optimizer: beside the chain (joined at the end of backward);
            # under data parallelism they belong to the first bucket, whose events are recorded on the main stream: serial there
            self._aux.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._aux):
                ops.fc_bwd(sv["feat"], sv["wt"], dlogits, dwt, self.g["fc.bias"], None)
                ops.transpose(dwt, self.g["fc.weight"], 1, self.c_last, self.num_classes)
            dwt.record_stream(self._aux)
            dlogits.record_stream(self._aux)
            ops.fc_bwd(sv["feat"], sv["wt"], dlogits, None, None, dfeat)
            self._fc_forked = True
        else:
            ops.fc_bwd(sv["feat"], sv["wt"], dlogits, dwt, self.g["fc.bias"], dfeat)
            ops.transpose(dwt, self.g["fc.weight"], 1, self.c_last, self.num_classes)
        dY = torch.empty((self.c_last, B * sv["Hl"] * sv["Wl"]), dtype=torch.float32, device=dev)
        ops.pool_bwd(dfeat, B, sv["Hl"] * sv["Wl"], 1, dY)
        nblk = len(self.blocks)
        for bidx, ((pre, inpl, planes, stride, ds), sb) in enumerate(zip(reversed(self.blocks), reversed(sv["blocks"]))):
            X, c1, c2, dsc, y = sb["X"], sb["c1"], sb["c2"], sb["dsc"], sb["y"]
            H, W, Ho, Wo = sb["H"], sb["W"], sb["Ho"], sb["Wo"]
            n_out = B * Ho * Wo
            b1, b2 = self.bn[pre + "bn1"], self.bn[pre + "bn2"]
            bd = self.bn.get(pre + "downsample.1")
            rk = (bd.k1, bd.k2, bd.k3) if ds else None
            if ops.BN_TAIL or BN_TAIL_PATHB == "1" or (BN_TAIL_PATHB == "split" and self.split):   # the reduce kernel's last workgroup per channel finalises bn2 (and the downsample BN)
                n2, nd = pre + "bn2", pre + "downsample.1"
                tail = ops.make_bn_tail(dY.device, n_out, self.p[n2 + ".weight"], b2, self.g[n2 + ".weight"], self.g[n2 + ".bias"],
                                        *((self.p[nd + ".weight"], bd, self.g[nd + ".weight"], self.g[nd + ".bias"]) if ds else ()))
                ops.bn_add_relu_bwd_reduce(dY, y, c2, dsc, b2.mean, bd.mean if ds else None, tail=tail)
            else:
                part, nparts = ops.bn_add_relu_bwd_reduce(dY, y, c2, dsc, b2.mean, bd.mean if ds else None, mask=sb.get("ymask"))
                if ds and self._aux is not None:     # the two finalisations read the same partial sums: side by side
                    self._aux.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(self._aux):
                        self._bn_bwd(pre + "downsample.1", part, nparts, nparts * 4, 4, 2, n_out)
                    part.record_stream(self._aux)
                self._bn_bwd(pre + "bn2", part, nparts, nparts * 4, 4, 1, n_out)
                if ds and self._aux is not None:
                    torch.cuda.current_stream().wait_stream(self._aux)
                elif ds:
                    self._bn_bwd(pre + "downsample.1", part, nparts, nparts * 4, 4, 2, n_out)
            dc2 = torch.empty_like(c2)
            ddsc = torch.empty_like(dsc) if ds else None
            ops.bn_add_relu_bwd_apply(dY, y, c2, dsc, (b2.k1, b2.k2, b2.k3), rk, dc2, ddsc, None if ds else dY, mask=sb.get("ymask"),
                                      amax_cell=self._cell(pre + "conv2", "b"))      # bound of dc2 for conv2's split data gradient
            if ds and self._aux is not None:
                ddsc_ready = torch.cuda.Event()
                ddsc_ready.record(torch.cuda.current_stream())
            # conv2 (input = relu(bn1(c1)), folded)
            self._conv_wgrad(pre + "conv2", c1, dc2, B, Ho, Wo, Ho, Wo, pro=(b1.scale, b1.shift))
            dz1, pm = self._conv_dgrad(pre + "conv2", dc2, B, Ho, Wo, Ho, Wo, epi=L.SAR_EPI_MASK, aux=c1,
                                       aux_affine=(b1.scale, b1.shift), aux_mean=b1.mean)
            self._bn_bwd(pre + "bn1", pm[0], pm[1], pm[1] * 2, 2, 1, n_out)
            ops.affine2(dz1, c1, (b1.k1, b1.k2, b1.k3), dz1, amax_cell=self._cell(pre + "conv1", "b"))      # dc1 in place (+ its bound)
            self._conv_wgrad(pre + "conv1", X, dz1, B, H, W, Ho, Wo)
            aux, even = dY, False
            if ds:
                self._conv_wgrad(pre + "downsample.0", X, ddsc, B, H, W, Ho, Wo)
                cd = self.convs[pre + "downsample.0"]
                if cd.k == 1 and cd.stride == 2 and cd.pad == 0 and stride == 2 and H == 2 * Ho and W == 2 * Wo:
                    # the 1x1 / stride 2 data gradient is non-zero at the even pixels only: computed at the small resolution
                    # (a plain stride-1 product) and added there by the parity-class launches of conv1's data gradient
                    aux = torch.empty((cd.cin, B * Ho * Wo), dtype=torch.float32, device=dev)
                    if self._aux is not None:     # forked: ddsc has been ready since the apply pass; joined in front of conv1's data gradient
                        self._aux.wait_event(ddsc_ready)
                        with torch.cuda.stream(self._aux):
                            ops.conv2d_gemm(ddsc, aux, self._w(pre + "downsample.0", True), cd.cout * cd.cin, cd.cin, B=B, Kc=cd.cout,
                                            M=cd.cin, H_src=Ho, W_src=Wo, H_out=Ho, W_out=Wo, KH=1, KW=1, stride=1, pad=0, transposed=True)
                        ddsc.record_stream(self._aux)
                        torch.cuda.current_stream().wait_stream(self._aux)
                    else:
                        ops.conv2d_gemm(ddsc, aux, self._w(pre + "downsample.0", True), cd.cout * cd.cin, cd.cin, B=B, Kc=cd.cout,
                                        M=cd.cin, H_src=Ho, W_src=Wo, H_out=Ho, W_out=Wo, KH=1, KW=1, stride=1, pad=0, transposed=True)
                    even = True
                else:
                    aux, _ = self._conv_dgrad(pre + "downsample.0", ddsc, B, H, W, Ho, Wo)
            dY, _ = self._conv_dgrad(pre + "conv1", dz1, B, H, W, Ho, Wo, epi=L.SAR_EPI_ADD, aux=aux, aux_even_pixels=even)
            if bucket_cb is not None and pre.endswith(".0.") and pre[5] in "432":   # first block of layer 4 / 3 / 2: that stage is done
                self._bucket_done({"4": 0, "3": 1, "2": 2}[pre[5]], bucket_cb)
            elif bucket_cb is None and (self._slab_flush == "block" or (self._slab_flush == "bucket" and pre.endswith(".0.") and pre[5] in "432")):
                self._flush_slabs()  # (experiment switch SAR_SLAB_FLUSH)
        # stem: maxpool + relu + bn backward, then the 7x7 weight gradient (the image needs no gradient)
        bn0 = self.bn["bn1"]
        c0 = sv["c0"]
        dz0 = torch.empty_like(c0)
        part, nparts = ops.bn_relu_maxpool_bwd(c0, bn0.scale, bn0.shift, bn0.mean, dY, dz0, B, sv["H1"], sv["W1"])
        self._bn_bwd("bn1", part, nparts, nparts * 2, 2, 1, B * sv["H1"] * sv["W1"])
        ops.affine2(dz0, c0, (bn0.k1, bn0.k2, bn0.k3), dz0)
        self._conv_wgrad("conv1", sv["x0"], dz0, B, sv["H"], sv["W"], sv["H1"], sv["W1"])
        if bucket_cb is not None:
            self._bucket_done(3, bucket_cb)                      # conv1 + bn1 + layer1
            if self._side is not None:
                torch.cuda.current_stream().wait_stream(self._side)
        else:
            self._flush_slabs()
            if self._side is not None:
                torch.cuda.current_stream().wait_stream(self._side)
        if getattr(self, "_fc_forked", False):
            torch.cuda.current_stream().wait_stream(self._aux)
            self._fc_forked = False
        dx = None
        if need_dx:
            cv = self.convs["conv1"]
            dx = torch.empty((B, 1, sv["H"], sv["W"]), dtype=torch.float32, device=dev)
            ops.conv2d_stem_dgrad(dz0, self._w("conv1"), dx, B=B, H=sv["H"], W=sv["W"], H_out=sv["H1"], W_out=sv["W1"],
                                  M=cv.cout, KH=cv.k, KW=cv.k, stride=cv.stride, pad=cv.pad)
        self._saved = None
        return dx

    # ------------------------------------------------------------------ training step
    def loss_and_grad(self, x, labels, need_dx=False, grad_scale=1.0, bucket_cb=None):
        """main_spectrogram.py:152-157: CrossEntropyLoss() (mean) and backward.  With need_dx also returns
        d loss / d image as a third value.  grad_scale (1 / world under data parallelism) scales every gradient -- the mean
        over ranks is then a plain SUM all-reduce -- but not the returned loss; bucket_cb: backward()."""
        logits = self.forward(x, training=True)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dlogits = torch.empty_like(logits)
        ops.softmax_ce(logits, labels, grad_scale / x.shape[0], loss, dlogits)
        if grad_scale != 1.0:
            loss.mul_(1.0 / grad_scale)
        dx = self.backward(dlogits, need_dx, bucket_cb)
        return (logits, loss, dx) if need_dx else (logits, loss)

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8):
        """torch.optim.Adam(lr) (main_spectrogram.py:106) over the flat buffers."""
        self._prepacked = None
        self.step_dev += 1.0
        if getattr(self, "_lr_host", None) != float(lr):      # one launch less per step while the schedule holds the rate
            self.lr_dev.fill_(float(lr))
            self._lr_host = float(lr)
        ops.adam(self.flat, self.adam_m, self.adam_v, self.grad, self.lr_dev, self.step_dev, betas[0], betas[1], eps)
