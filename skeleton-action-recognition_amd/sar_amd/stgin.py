"""ST-GIN training engine: the sibling model of ST-GCN whose spatial operator is the graph isomorphism
convolution -- models/stgin.py:11-140, GraphIsoConvTD models/gcn.py:112-163 -- on the same HIP kernels, flat parameter /
gradient buffers and train step as sar_amd/stgcn.py (main_gnn.py:219-239 is model-agnostic: `--model stgin`).

Per block (models/stgin.py:58-66):
    x' = einsum('nctv,kvw->nkctw', x, concat(A[:2], diag(1 + epsilon)))           <= 4-entry gathers (sar_graph_gather_expand_f32)
                                                                                  for the fixed sparse A; identity slices and the
                                                                                  self slice read x itself (prologue scale 1 + eps);
                                                                                  dense A: sar_graph_dense_bwd_data_f32
    per slice k:  Conv2D(h,1x1) -> BN -> ReLU -> Conv2D(h,1x1) -> BN -> ReLU      sar_conv_gemm_f32 (taps = 1, BN statistics in
                                                                                  the epilogue, BN + ReLU folded into the next
                                                                                  convolution's operand load), h = filters / 2
    s = sum_k (...)                                                               sar_gin_sum_fwd_f32 (+ statistics of s)
    tgcn (BN -> ReLU -> Conv2D(filters,[9,1],stride) -> BN), residual, ReLU       exactly sar_amd/stgcn.py's
The K = 3 branch MLPs of a block are stacked along the channel axis (a1, a2: [3h][B*T*V]; one 3h-channel BatchNorm state per
MLP layer) so that the BatchNorm finalisations and the element-wise backward passes are one launch per block, not three.
`epsilon` is a trainable scalar per block: its gradient is <dout . x^T, W> of the self slice's first convolution
(sar_gin_eps_grad_f32).  fp32 only.
"""
import numpy as np
import torch

from . import _lib as L
from . import ops
from .stgcn import STGCN, _BN, BLOCKS, KS, KT, same_pad, ntu_adjacency


class STGIN(STGCN):
    def __init__(self, num_classes=60, in_channels=3, num_node=25, A=None, device="cuda", seed=0, bone_pairs=None,
                 blocks=None, motion=False, mfma="fp32"):
        L.load()  # fail loudly if the HIP library is missing
        assert mfma == "fp32", "the ST-GIN engine is fp32"
        import os
        self.cn8 = self.bf16 = self.dense_A = False
        self.train_adjacency = False
        self.packed = None
        self._wT_off, self._wT_perm, self._wT = {}, None, None      # STGCN's batched data-gradient operands: not used here
        self.device = torch.device(device)
        self.num_classes, self.C_in, self.V = num_classes, in_channels, num_node
        self.blocks = list(blocks if blocks is not None else BLOCKS)
        A = np.asarray(ntu_adjacency()[:KS - 1] if A is None else A, dtype=np.float64)       # models/stgin.py:87-90: Graph().A[:2]
        assert A.shape == (KS - 1, num_node, num_node)
        self.A_host = A.astype(np.float32)
        self.A = torch.from_numpy(self.A_host).to(self.device).contiguous()    # 'adjacency_matrix', non-trainable
        # The adjacency is fixed, so the contractions x . A_k run as <= 4-entry gathers (sar_graph_gather_*_f32) when every
        # column / row of [A_0, .., diag(1)] has <= 4 non-zeros (the NTU graph); a slice that IS the identity (slice 0 of the
        # 'spatial' strategy) is not materialised at all.  Denser adjacencies take the dense kernels (csrc/graph_dense.hip).
        a_ext = np.concatenate([self.A_host, np.eye(num_node, dtype=np.float32)[None]])
        try:
            self.tab_f = ops.GraphTables(a_ext, self.device, transpose=False)
            self.tab_b = ops.GraphTables(a_ext, self.device, transpose=True)
            self.identity_slice = [bool(np.array_equal(self.A_host[k], np.eye(num_node, dtype=np.float32))) for k in range(KS - 1)]
        except ValueError:
            self.tab_f = self.tab_b = None
            self.identity_slice = [False] * (KS - 1)
        self._side = (ops.shared_side_stream(self.device, int(os.environ.get("SAR_WGRAD_PRIO", "0")))
                      if self.device.type == "cuda" and os.environ.get("SAR_WGRAD_STREAM", "1") == "1" else None)
        self.motion = bool(motion)
        self.bone_parent = None
        if bone_pairs is not None:
            bp = np.full(num_node, -1, dtype=np.int32)
            for v1, v2 in bone_pairs:
                bp[v1 - 1] = v2 - 1
            self.bone_parent = torch.from_numpy(bp).to(self.device)

        # ---- parameter table (Keras layouts).  Branch parameters of one kind are registered back to back so that the
        # stacked [K*h] views used by the kernels are contiguous ranges of the flat buffer (h % 4 == 0).
        self.shapes = {}
        nch = num_node * in_channels
        self._add("data_bn.gamma", (nch,)), self._add("data_bn.beta", (nch,))
        cin = in_channels
        self.kinds = []
        for i, (f, s, res) in enumerate(self.blocks):
            pre, h = "l%d." % i, f // 2
            assert f % 8 == 0, "filters / 2 must be a multiple of 4"
            self.kinds.append("none" if not res else ("identity" if (cin == f and s == 1) else "conv"))   # stgin.py:41-56
            for k in range(KS):
                self._add(pre + "mlp%d.c1.kernel" % k, (1, 1, cin, h)), self._add(pre + "mlp%d.c1.bias" % k, (h,))
            for k in range(KS):
                self._add(pre + "mlp%d.c2.kernel" % k, (1, 1, h, h)), self._add(pre + "mlp%d.c2.bias" % k, (h,))
            for bn in ("bn1", "bn2"):
                for part in ("gamma", "beta"):
                    for k in range(KS):
                        self._add(pre + "mlp%d.%s.%s" % (k, bn, part), (h,))
            self._add(pre + "epsilon", ())
            self._add(pre + "bn1.gamma", (h,)), self._add(pre + "bn1.beta", (h,))
            self._add(pre + "tcn.kernel", (KT, 1, h, f)), self._add(pre + "tcn.bias", (f,))
            self._add(pre + "bn2.gamma", (f,)), self._add(pre + "bn2.beta", (f,))
            if self.kinds[i] == "conv":
                self._add(pre + "res.kernel", (1, 1, cin, f)), self._add(pre + "res.bias", (f,))
                self._add(pre + "res_bn.gamma", (f,)), self._add(pre + "res_bn.beta", (f,))
            cin = f
        self.C_last = cin
        self._add("logits.kernel", (1, 1, cin, num_classes)), self._add("logits.bias", (num_classes,))
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
        self.p = {k: self._view(self.flat, k) for k in self.shapes}
        self.g = {k: self._view(self.grad, k) for k in self.shapes}
        self.bn = {"data_bn": _BN(nch, dev)}
        cmax = in_channels
        for i, (f, s, res) in enumerate(self.blocks):
            pre, h = "l%d." % i, f // 2
            for bn in ("bn1", "bn2"):      # stacked views / state of the K branch BatchNorms
                for part in ("gamma", "beta"):
                    o = self.offsets[pre + "mlp0.%s.%s" % (bn, part)]
                    assert self.offsets[pre + "mlp%d.%s.%s" % (KS - 1, bn, part)] == o + (KS - 1) * h
                    self.p[pre + "mlp.%s.%s" % (bn, part)] = self.flat[o:o + KS * h]
                    self.g[pre + "mlp.%s.%s" % (bn, part)] = self.grad[o:o + KS * h]
                self.bn[pre + "mlp." + bn] = _BN(KS * h, dev)
            self.bn[pre + "bn1"], self.bn[pre + "bn2"] = _BN(h, dev), _BN(f, dev)
            if self.kinds[i] == "conv":
                self.bn[pre + "res_bn"] = _BN(f, dev)
            cmax = max(cmax, f)
        self._zeros = torch.zeros(cmax, dtype=torch.float32, device=dev)
        self._init_params(seed)
        self._saved = None
        self._deferred, self._flushing = [], False
        self._buckets = self._make_buckets(total)

    # ------------------------------------------------------------------ parameters
    @staticmethod
    def _branch_stat(name):
        """'l3.mlp1.bn2.moving_mean' -> ('l3.mlp.bn2', 1, 'moving_mean'); None for every other name"""
        parts = name.split(".")
        if len(parts) == 4 and parts[1].startswith("mlp") and parts[1][3:].isdigit() and parts[3].startswith("moving_"):
            return "%s.mlp.%s" % (parts[0], parts[2]), int(parts[1][3:]), parts[3]
        return None

    def load_params(self, params):
        """params: dict name -> tensor in the oracle / Keras layouts (oracle/stgin.py names; 'A' ignored)."""
        rest = {}
        for k, v in params.items():
            b = self._branch_stat(k)
            if b is None:
                rest[k] = v
            else:
                st = getattr(self.bn[b[0]], b[2])
                h = st.numel() // KS
                st[b[1] * h:(b[1] + 1) * h].copy_(v.to(torch.float32))
        super().load_params({k: v for k, v in rest.items() if k in self.shapes or k.endswith((".moving_mean", ".moving_var"))})

    def state_dict(self):
        out = {k: self.p[k].detach().cpu().clone() for k in self.shapes}
        for k, b in self.bn.items():
            parts = k.split(".")
            if len(parts) == 3 and parts[1] == "mlp":
                h = b.moving_mean.numel() // KS
                for j in range(KS):
                    out["%s.mlp%d.%s.moving_mean" % (parts[0], j, parts[2])] = b.moving_mean[j * h:(j + 1) * h].cpu().clone()
                    out["%s.mlp%d.%s.moving_var" % (parts[0], j, parts[2])] = b.moving_var[j * h:(j + 1) * h].cpu().clone()
            else:
                out[k + ".moving_mean"] = b.moving_mean.cpu().clone()
                out[k + ".moving_var"] = b.moving_var.cpu().clone()
        out["A"] = self.A.cpu().clone()
        return out

    def grads(self):
        return {k: self.g[k] for k in self.shapes}

    # ------------------------------------------------------------------ forward
    def _block_forward(self, i, X, cin, f, s, B, T, training, saved, keep):
        V, dev = self.V, X.device
        pre, h, K = "l%d." % i, f // 2, KS
        kind = self.kinds[i]
        To, pad, _ = same_pad(T, KT, s)
        n_in, n_out = B * T * V, B * To * V
        epi = L.SAR_EPI_STATS if training else L.SAR_EPI_NONE
        new = lambda rows, n: torch.empty((rows, n), dtype=torch.float32, device=dev)
        # ---- sgcn: GraphIsoConvTD (models/gcn.py:149-163)
        escale = torch.empty(cin, dtype=torch.float32, device=dev)           # 1 + eps per input channel
        sscale = torch.empty(K, dtype=torch.float32, device=dev)             # per slice: (1, .., 1, 1 + eps)
        if self.tab_f is not None:                                            # gather lists; identity slices read x itself
            table = None
            ops.gin_adjacency(None, self.p[pre + "epsilon"], None, escale, sscale, Km1=K - 1, V=V)
            live = [k for k in range(K - 1) if not self.identity_slice[k]]
            z = new(max(1, len(live)) * cin, n_in)                            # z_k = x . A_k for the slices that need it
            src = [X] * K
            for j, k in enumerate(live):
                src[k] = z[j * cin:(j + 1) * cin]
                ops.graph_gather_expand(X, self.tab_f, 1, cin, V, src[k], k0=k)
        else:
            table = torch.empty((K, V, V), dtype=torch.float32, device=dev)  # [A_0^T, A_1^T, (1 + eps) I]
            ops.gin_adjacency(self.A, self.p[pre + "epsilon"], table, escale, sscale)
            z = new((K - 1) * cin, n_in)                                      # z[k cin + c] = x[c] . A_k
            ops.graph_dense_bwd_data(X, table, z, K - 1, cin, V, B * T)
            src = [z[k * cin:(k + 1) * cin] for k in range(K - 1)] + [X]
        a1, a2 = new(K * h, n_in), new(K * h, n_in)
        geo = dict(B=B, V=V, T_src=T, T_out=T, taps=1, stride=1, pad=0)
        part1 = part2 = None
        if training:
            np1 = ops.conv_gemm_nparts(Kc=cin, M=h, **geo)
            np2 = ops.conv_gemm_nparts(Kc=h, M=h, **geo)
            part1 = torch.empty((K * h, np1, 2), dtype=torch.float32, device=dev)
            part2 = torch.empty((K * h, np2, 2), dtype=torch.float32, device=dev)
        rows = lambda t, k: t[k * h:(k + 1) * h] if t is not None else None
        for k in range(K):
            ops.conv_gemm(L.SAR_CONV_TEMPORAL, src[k], rows(a1, k), self.p[pre + "mlp%d.c1.kernel" % k], 0, h, Kc=cin, M=h,
                          bias=self.p[pre + "mlp%d.c1.bias" % k], pro=(escale, self._zeros[:cin]) if k == K - 1 else None,
                          epi=epi, partials_out=rows(part1, k), **geo)
        m1 = self.bn[pre + "mlp.bn1"]
        if training:
            self._bn_forward_stats(pre + "mlp.bn1", part1, np1, n_in, True, True)
        else:
            self._bn_eval(pre + "mlp.bn1")
        for k in range(K):
            ops.conv_gemm(L.SAR_CONV_TEMPORAL, rows(a1, k), rows(a2, k), self.p[pre + "mlp%d.c2.kernel" % k], 0, h, Kc=h, M=h,
                          bias=self.p[pre + "mlp%d.c2.bias" % k], pro=(rows(m1.scale, k), rows(m1.shift, k)), pro_relu=True,
                          epi=epi, partials_out=rows(part2, k), **geo)
        m2 = self.bn[pre + "mlp.bn2"]
        if training:
            self._bn_forward_stats(pre + "mlp.bn2", part2, np2, n_in, True, True)
        else:
            self._bn_eval(pre + "mlp.bn2")
        g = new(h, n_in)
        r1 = ops.gin_sum_fwd(a2, m2.scale, m2.shift, K, g, stats=training)
        if training:
            self._bn_forward_stats(pre + "bn1", r1[0], r1[1], n_in, True, True)
        else:
            self._bn_eval(pre + "bn1")
        bn1 = self.bn[pre + "bn1"]
        # ---- tgcn (models/stgin.py:27-39), residual (:41-56), ReLU
        u = new(f, n_out)
        r2 = ops.conv_gemm(L.SAR_CONV_TEMPORAL, g, u, self.p[pre + "tcn.kernel"], h * f, f, B=B, V=V, T_src=T, T_out=To, Kc=h,
                           M=f, taps=KT, stride=s, pad=pad, bias=self.p[pre + "tcn.bias"], pro=(bn1.scale, bn1.shift),
                           pro_relu=True, epi=epi)
        if training:
            self._bn_forward_stats(pre + "bn2", r2[0], r2[1], n_out, True, True)
        else:
            self._bn_eval(pre + "bn2")
        bn2 = self.bn[pre + "bn2"]
        r = rbn = None
        if kind == "conv":
            r = new(f, n_out)
            r3 = ops.conv_gemm(L.SAR_CONV_TEMPORAL, X, r, self.p[pre + "res.kernel"], 0, f, B=B, V=V, T_src=T, T_out=To, Kc=cin,
                               M=f, taps=1, stride=s, pad=0, bias=self.p[pre + "res.bias"], epi=epi)
            if training:
                self._bn_forward_stats(pre + "res_bn", r3[0], r3[1], n_out, True, True)
            else:
                self._bn_eval(pre + "res_bn")
            rbn = self.bn[pre + "res_bn"]
        y = new(f, n_out)
        ops.bn_add_relu_fwd(u, bn2.scale, bn2.shift, {"none": 0, "identity": 1, "conv": 2}[kind], X if kind == "identity" else r,
                            rbn.scale if rbn else None, rbn.shift if rbn else None, y)
        if training:
            saved["blocks"].append(dict(X=X, g=g, u=u, r=r, y=y, T=T, To=To, pad=pad, cin=cin, f=f, s=s, kind=kind, src=src, a1=a1,
                                        a2=a2, table=table, sscale=sscale))
        if keep is not None:
            keep[pre + "g"], keep[pre + "u"], keep[pre + "y"], keep[pre + "a1"], keep[pre + "a2"] = g, u, y, a1, a2
        return y, To

    # ------------------------------------------------------------------ backward
    def _block_backward(self, i, sb, dY, B):
        V, dev = self.V, dY.device
        pre, K = "l%d." % i, KS
        X, g, u, r, y, src, a1, a2 = sb["X"], sb["g"], sb["u"], sb["r"], sb["y"], sb["src"], sb["a1"], sb["a2"]
        T, To, pad, cin, f, s, kind = sb["T"], sb["To"], sb["pad"], sb["cin"], sb["f"], sb["s"], sb["kind"]
        h = f // 2
        n_in, n_out = B * T * V, B * To * V
        bn1, bn2, m1, m2 = self.bn[pre + "bn1"], self.bn[pre + "bn2"], self.bn[pre + "mlp.bn1"], self.bn[pre + "mlp.bn2"]
        rbn = self.bn.get(pre + "res_bn")
        new = lambda rows, n: torch.empty((rows, n), dtype=torch.float32, device=dev)
        rows = lambda t, k: t[k * h:(k + 1) * h]
        # ---- tail: y = relu(bn2(u) + res)
        rk = (rbn.k1, rbn.k2, rbn.k3) if kind == "conv" else None
        if ops.BN_TAIL:      # the reduce kernel's last workgroup per channel finalises BN2 (and the residual BN)
            tail = ops.make_bn_tail(dev, n_out, self.p[pre + "bn2.gamma"], bn2, self.g[pre + "bn2.gamma"], self.g[pre + "bn2.beta"],
                                    *((self.p[pre + "res_bn.gamma"], rbn, self.g[pre + "res_bn.gamma"], self.g[pre + "res_bn.beta"])
                                      if kind == "conv" else ()))
            ops.bn_add_relu_bwd_reduce(dY, y, u, r if kind == "conv" else None, bn2.mean, rbn.mean if kind == "conv" else None,
                                       tail=tail)
        else:
            part, nparts = ops.bn_add_relu_bwd_reduce(dY, y, u, r if kind == "conv" else None, bn2.mean,
                                                      rbn.mean if kind == "conv" else None)
            ops.bn_bwd_finalize(part, nparts, nparts * 4, 4, 0, 1, f, n_out, self.p[pre + "bn2.gamma"], bn2.mean, bn2.rstd,
                                self.g[pre + "bn2.gamma"], self.g[pre + "bn2.beta"], bn2.k1, bn2.k2, bn2.k3)
            if kind == "conv":
                ops.bn_bwd_finalize(part, nparts, nparts * 4, 4, 0, 2, f, n_out, self.p[pre + "res_bn.gamma"], rbn.mean, rbn.rstd,
                                    self.g[pre + "res_bn.gamma"], self.g[pre + "res_bn.beta"], rbn.k1, rbn.k2, rbn.k3)
        du = torch.empty_like(u)
        dr = torch.empty_like(r) if kind == "conv" else None
        dz = dY if kind == "identity" else None
        ops.bn_add_relu_bwd_apply(dY, y, u, r if kind == "conv" else None, (bn2.k1, bn2.k2, bn2.k3), rk, du, dr, dz)
        # ---- temporal conv (h -> f channels)
        flat_w = self.grad[self.offsets[pre + "tcn.kernel"]:self.offsets[pre + "tcn.bias"] + f]
        self._off_critical_path(lambda: ops.conv_wgrad(
            L.SAR_CONV_TEMPORAL, g, du, flat_w, B=B, V=V, T_src=T, T_out=To, Kc=h, M=f, taps=KT, stride=s, pad=pad,
            pro=(bn1.scale, bn1.shift), pro_relu=True, w_stride_tap=h * f, w_stride_c=f, wsize=KT * h * f, bsize=f), g, du)
        wT = torch.empty((KT, f, h), dtype=torch.float32, device=dev)
        ops.transpose(self.p[pre + "tcn.kernel"], wT, KT, h, f)               # [tap][c][m] -> [tap][m][c]
        ds = new(h, n_in)
        pm = ops.conv_gemm(L.SAR_CONV_TEMPORAL, du, ds, wT, f * h, h, B=B, V=V, T_src=To, T_out=T, Kc=f, M=h, taps=KT, stride=s,
                           pad=pad, transposed=True, epi=L.SAR_EPI_MASK, aux=g, aux_affine=(bn1.scale, bn1.shift),
                           aux_mean=bn1.mean)
        ops.bn_bwd_finalize(pm[0], pm[1], pm[1] * 2, 2, 0, 1, h, n_in, self.p[pre + "bn1.gamma"], bn1.mean, bn1.rstd,
                            self.g[pre + "bn1.gamma"], self.g[pre + "bn1.beta"], bn1.k1, bn1.k2, bn1.k3)
        ops.affine2(ds, g, (bn1.k1, bn1.k2, bn1.k3), ds)                      # gradient w.r.t. s = sum of the branches
        # ---- branches, last BN + ReLU (all K in one launch each)
        part, nparts = ops.gin_bwd_reduce(ds, a2, m2.scale, m2.shift, m2.mean, K)
        ops.bn_bwd_finalize(part, nparts, nparts * 2, 2, 0, 1, K * h, n_in, self.p[pre + "mlp.bn2.gamma"], m2.mean, m2.rstd,
                            self.g[pre + "mlp.bn2.gamma"], self.g[pre + "mlp.bn2.beta"], m2.k1, m2.k2, m2.k3)
        da2 = a2                                                               # in place: a2 is not read again
        ops.gin_bwd_apply(ds, a2, m2.scale, m2.shift, (m2.k1, m2.k2, m2.k3), K, da2)
        # ---- second 1x1 convolution of every branch
        geo = dict(B=B, V=V, T_src=T, T_out=T, taps=1, stride=1, pad=0)
        npm = ops.conv_gemm_nparts(Kc=h, M=h, transposed=True, epi=L.SAR_EPI_MASK, **geo)
        pm1 = torch.empty((K * h, npm, 2), dtype=torch.float32, device=dev)
        da1 = new(K * h, n_in)
        w2T = torch.empty((K, h, h), dtype=torch.float32, device=dev)
        for k in range(K):
            kn, bi = pre + "mlp%d.c2.kernel" % k, pre + "mlp%d.c2.bias" % k
            flat = self.grad[self.offsets[kn]:self.offsets[bi] + h]
            self._off_critical_path(lambda k=k, flat=flat: ops.conv_wgrad(
                L.SAR_CONV_TEMPORAL, rows(a1, k), rows(da2, k), flat, Kc=h, M=h, pro=(rows(m1.scale, k), rows(m1.shift, k)),
                pro_relu=True, w_stride_tap=0, w_stride_c=h, wsize=h * h, bsize=h, **geo), a1, da2)
            ops.transpose(self.p[kn], w2T[k], 1, h, h)
            ops.conv_gemm(L.SAR_CONV_TEMPORAL, rows(da2, k), rows(da1, k), w2T[k], 0, h, Kc=h, M=h, transposed=True,
                          epi=L.SAR_EPI_MASK, aux=rows(a1, k), aux_affine=(rows(m1.scale, k), rows(m1.shift, k)),
                          aux_mean=rows(m1.mean, k), partials_out=rows(pm1, k), **geo)
        ops.bn_bwd_finalize(pm1, npm, npm * 2, 2, 0, 1, K * h, n_in, self.p[pre + "mlp.bn1.gamma"], m1.mean, m1.rstd,
                            self.g[pre + "mlp.bn1.gamma"], self.g[pre + "mlp.bn1.beta"], m1.k1, m1.k2, m1.k3)
        ops.affine2(da1, a1, (m1.k1, m1.k2, m1.k3), da1)
        # ---- first 1x1 convolution of every branch; the self slice also yields d epsilon
        dzz = new(K * cin, n_in)
        w1T = torch.empty((K, h, cin), dtype=torch.float32, device=dev)
        for k in range(K):
            kn, bi = pre + "mlp%d.c1.kernel" % k, pre + "mlp%d.c1.bias" % k
            flat = self.grad[self.offsets[kn]:self.offsets[bi] + h]

            def wgrad(k=k, flat=flat, kn=kn):
                ops.conv_wgrad(L.SAR_CONV_TEMPORAL, src[k], rows(da1, k), flat, Kc=cin, M=h, w_stride_tap=0, w_stride_c=h,
                               wsize=cin * h, bsize=h, **geo)
                if k == K - 1:     # taken on the un-scaled x: d eps = <G, W>, dW = (1 + eps) G
                    ops.gin_eps_grad(self.g[kn], self.p[kn], self.p[pre + "epsilon"], self.g[pre + "epsilon"])
            self._off_critical_path(wgrad, src[k], da1)
            ops.transpose(self.p[kn], w1T[k], 1, cin, h)
            ops.conv_gemm(L.SAR_CONV_TEMPORAL, rows(da1, k), dzz[k * cin:(k + 1) * cin], w1T[k], 0, cin, Kc=h, M=cin,
                          transposed=True, **geo)
        dXres = self._residual_backward(i, sb, dr, B)
        # ---- dX = sum_k dz_k . A_k^T + (1 + eps) dz_self (+ the skip-path gradient)
        dX = new(cin, n_in)
        aux = dY if kind == "identity" else dXres
        if sb["table"] is None:
            ops.graph_gather_sum(dzz, self.tab_b, sb["sscale"], K, cin, V, dX, add=aux)
        else:
            ops.graph_dense_fwd(dzz, sb["table"], dX, K, cin, V, B * T, add=aux)
        return dX
