"""CPU checks of oracle/stgin.py (the restatement of models/stgin.py + GraphIsoConvTD, models/gcn.py:112-163) -- parity
unpinned like oracle/stgcn.py (TensorFlow is absent), so what can be checked here is internal consistency: the einsum against
explicit matrix products per slice, the structure / parameter count the reference's constructor implies, and the analytic
gradients (epsilon included) against central finite differences in float64."""
import torch

from oracle import stgcn as S
from oracle import stgin as G


def test_structure_and_parameter_count():
    p = G.init_params(60)
    assert tuple(p["A"].shape) == (2, 25, 25)                      # models/stgin.py:87-90: Graph().A[:2]
    assert torch.equal(p["A"][0], torch.eye(25))                   # 'spatial' strategy: slice 0 = self links
    names = S.trainable_names(p)
    assert "A" not in names and sum(n.endswith("epsilon") for n in names) == 10
    # per block: 3 x (conv cin->h + BN + conv h->h + BN) + epsilon + BN(h) + conv9 h->f + BN(f) [+ 1x1 residual conv + BN]
    def block(cin, f, res_conv):
        h = f // 2
        n = 3 * (cin * h + h + 2 * h + h * h + h + 2 * h) + 1 + 2 * h + 9 * h * f + f + 2 * f
        return n + (cin * f + f + 2 * f if res_conv else 0)
    want = 2 * 75 + block(3, 64, False) + 3 * block(64, 64, False) + block(64, 128, True) + 2 * block(128, 128, False) \
        + block(128, 256, True) + 2 * block(256, 256, False) + 256 * 60 + 60
    assert sum(p[k].numel() for k in names) == want == 1778172


def test_graph_iso_conv_is_the_sum_of_per_slice_mlps():
    blocks = [(16, 1, False)]
    p = S.randomize_affine(G.init_params(5, seed=1, dtype=torch.float64, blocks=blocks), seed=2)
    p["l0.epsilon"] = torch.tensor(0.25, dtype=torch.float64)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 6, 25, generator=g, dtype=torch.float64)
    out = G.graph_iso_conv(x, p, "l0.", p["A"], False, None, None, None)
    ref = 0
    slices = [p["A"][0], p["A"][1], torch.eye(25, dtype=torch.float64) * 1.25]
    for k, Ak in enumerate(slices):
        z = x @ Ak                                                  # (n, c, t, v) . (v, w)
        q = "l0.mlp%d." % k
        for c, bn in (("c1", "bn1"), ("c2", "bn2")):
            z = torch.einsum("nctv,cm->nmtv", z, p[q + c + ".kernel"][0, 0]) + p[q + c + ".bias"].view(1, -1, 1, 1)
            z = (z - p[q + bn + ".moving_mean"].view(1, -1, 1, 1)) / torch.sqrt(p[q + bn + ".moving_var"].view(1, -1, 1, 1) + 1e-3)
            z = torch.relu(z * p[q + bn + ".gamma"].view(1, -1, 1, 1) + p[q + bn + ".beta"].view(1, -1, 1, 1))
        ref = ref + z
    assert (out - ref).abs().max() < 1e-12


def test_gradients_against_finite_differences():
    blocks = [(8, 1, False), (16, 2, True)]
    p = S.randomize_affine(G.init_params(4, seed=3, dtype=torch.float64, blocks=blocks), seed=4)
    p["l0.epsilon"], p["l1.epsilon"] = torch.tensor(0.2, dtype=torch.float64), torch.tensor(-0.1, dtype=torch.float64)
    x, y = S.synthetic_batch(2, seed=5, T=8, M=1, num_classes=4, dtype=torch.float64)
    _, loss, grads, _, _ = G.loss_and_grads(p, x, y, blocks=blocks)
    # the ReLU pattern of the unperturbed point is held fixed so that the finite difference never crosses a kink
    taps = {}
    G.forward(p, x, True, {}, taps, blocks)
    masks = {k[:-4]: (v > 0) for k, v in taps.items() if k.endswith("_pre")}
    _, _, grads_m, _, _ = G.loss_and_grads(p, x, y, blocks=blocks, masks=masks)
    for k in grads:
        assert (grads[k] - grads_m[k]).abs().max() < 1e-12

    def loss_at(name, idx, delta):
        q = dict(p)
        t = p[name].clone()
        t.view(-1)[idx] += delta
        q[name] = t
        return S.loss_fn(G.forward(q, x, True, None, None, blocks, masks), y, 2).item()

    for name, idx in (("l0.epsilon", 0), ("l1.epsilon", 0), ("l0.mlp2.c1.kernel", 5), ("l1.mlp0.c2.kernel", 11),
                      ("l1.mlp1.bn1.gamma", 3), ("l0.tcn.kernel", 17), ("l1.res.kernel", 2)):
        h = 1e-6
        fd = (loss_at(name, idx, h) - loss_at(name, idx, -h)) / (2 * h)
        an = grads[name].reshape(-1)[idx].item()
        assert abs(fd - an) < 1e-6 * max(1.0, abs(an)) + 1e-8, (name, fd, an)
