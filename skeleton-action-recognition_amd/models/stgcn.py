"""Drop-in for the reference's models/stgcn.py: `Model(num_classes=60)` called as `model(x, training)`.

x: (N, in_channels=3, T, V=25, M) float32 on the GPU -> logits (N, num_classes)   (models/stgcn.py:135-160).
The arithmetic runs in libsar_hip.so through sar_amd.stgcn.STGCN; this module only adapts it to the
torch.nn.Module / autograd interfaces so that `loss.backward()` and any torch optimizer work, and keeps the
reference's attribute names used by its training script (main_gnn.py:228-232,311,315):
  * `trainable_variables` -- objects with a `.name`; the adjacency is exposed as the NON-trainable
    variable `adjacency_matrix` (models/stgcn.py:105-109),
  * block hyper-parameters (models/stgcn.py:113-123) are fixed as in the reference.
"""
import torch

from sar_amd.stgcn import STGCN, BLOCKS  # noqa: F401


class _STGCNFunction(torch.autograd.Function):
    """Whole-network forward/backward on the HIP engine (one autograd node)."""

    @staticmethod
    def forward(ctx, x, engine, training, *params):
        ctx.engine = engine
        ctx.training = training
        return engine.forward(x, training=training)

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.engine
        if not ctx.training:
            raise RuntimeError("backward through model(x, training=False) is not supported (inference path)")
        eng.backward(dlogits.contiguous())
        grads = tuple(eng.g[k].clone() for k in eng.shapes)
        return (None, None, None) + grads


class _Variable:
    """Keras-like view of a parameter: `.name`, `.numpy()`; `.tensor` is the torch parameter."""

    def __init__(self, name, tensor, trainable=True):
        self.name, self.tensor, self.trainable = name, tensor, trainable

    def numpy(self):
        return self.tensor.detach().cpu().numpy()


class Model(torch.nn.Module):
    def __init__(self, num_classes=60, in_channels=3, device="cuda", seed=0, stream="joint", mfma="fp32",
                 trainable_adjacency=False):
        """stream (not in the reference's constructor, which reads pre-computed files): 'joint', 'bone', 'joint_motion'
        or 'bone_motion' -- the bone (data_gen/gen_bone_data.py) and motion (data_gen/gen_motion_data.py) transforms are
        applied on the fly to JOINT input inside the data_bn prologue, bit-exactly.
        mfma: 'fp32' (the reference's arithmetic on the fp32 MFMA), 'f32_split' (fp32 storage and results, the contractions as three
        products of fp16 terms on the fp16 matrix pipe: same parity tolerances, 1.6x the training rate; 'f32_split_bf16x6': six
        products of bf16 terms) or 'bf16' (bf16 activations and MFMA operands, fp32 everything else: sar_amd/stgcn.py)."""
        super().__init__()
        assert stream in ("joint", "bone", "joint_motion", "bone_motion"), stream
        from sar_amd.bone import NTU_BONE_PAIRS
        self.engine = STGCN(num_classes=num_classes, in_channels=in_channels, device=device, seed=seed,
                            bone_pairs=NTU_BONE_PAIRS if stream.startswith("bone") else None,
                            motion=stream.endswith("motion"), mfma=mfma, trainable_adjacency=trainable_adjacency)
        # parameters are views into the engine's flat fp32 buffer (one all-reduce bucket, fused optimizer)
        self._names = list(self.engine.shapes)
        for k in self._names:
            self.register_parameter(k.replace(".", "_"), torch.nn.Parameter(self.engine.p[k]))
        if not trainable_adjacency:
            self.register_buffer("adjacency_matrix", self.engine.A)      # non-trainable, models/stgcn.py:105-109
        # trainable_adjacency=True (models/gcn.py:212-238 AdjGraphConv's variable, shared by the blocks): `adjacency_matrix`
        # is one of the parameters registered above and main_gnn.py's --freeze-graph-until gates its gradient
        self.A = self.adjacency_matrix

    @property
    def trainable_variables(self):
        return [_Variable(k, getattr(self, k.replace(".", "_"))) for k in self._names]

    @property
    def variables(self):
        return self.trainable_variables + [_Variable("adjacency_matrix", self.adjacency_matrix, False)]

    def forward(self, x, training=None):
        if training is None:
            training = self.training
        params = [getattr(self, k.replace(".", "_")) for k in self._names]
        if training and torch.is_grad_enabled():
            return _STGCNFunction.apply(x, self.engine, True, *params)
        return self.engine.forward(x, training=training)

    # the engine's fused paths, for scripts that want the reference's exact train step without autograd
    def train_step(self, x, labels, lr, global_batch_size=None, momentum=0.9):
        logits, loss = self.engine.loss_and_grad(x, labels, global_batch_size)
        self.engine.sgd_step(lr, momentum)
        return logits, loss
