"""Drop-in for the reference's models/stgin.py: `Model(num_classes=60)` called as `model(x, training)` -- ST-GCN's sibling
with the graph isomorphism convolution (models/gcn.py:112-163) as its spatial operator (`main_gnn.py --model stgin`).

x: (N, in_channels=3, T, V=25, M) float32 on the GPU -> logits (N, num_classes)   (models/stgin.py:117-140).
The arithmetic runs in libsar_hip.so through sar_amd.stgin.STGIN; this module adapts it to torch.nn.Module / autograd exactly
like models/stgcn.py does for ST-GCN.  The adjacency `adjacency_matrix` = Graph().A[:2] is NON-trainable
(models/stgin.py:87-90); each block owns the trainable scalar `epsilon` (models/gcn.py:144-147).
"""
import torch

from sar_amd.stgin import STGIN, BLOCKS  # noqa: F401
from models.stgcn import _STGCNFunction, _Variable


class Model(torch.nn.Module):
    def __init__(self, num_classes=60, in_channels=3, device="cuda", seed=0, stream="joint", mfma="fp32",
                 trainable_adjacency=False):
        super().__init__()
        assert stream in ("joint", "bone", "joint_motion", "bone_motion"), stream
        assert mfma == "fp32", "models.stgin runs in fp32 (the bf16 configuration is built for models.stgcn)"
        assert not trainable_adjacency, "models.stgin keeps its adjacency fixed (models/stgin.py:87-90)"
        from sar_amd.bone import NTU_BONE_PAIRS
        self.engine = STGIN(num_classes=num_classes, in_channels=in_channels, device=device, seed=seed,
                            bone_pairs=NTU_BONE_PAIRS if stream.startswith("bone") else None, motion=stream.endswith("motion"))
        self._names = list(self.engine.shapes)
        for k in self._names:
            self.register_parameter(k.replace(".", "_"), torch.nn.Parameter(self.engine.p[k]))
        self.register_buffer("adjacency_matrix", self.engine.A)
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

    def train_step(self, x, labels, lr, global_batch_size=None, momentum=0.9):
        logits, loss = self.engine.loss_and_grad(x, labels, global_batch_size)
        self.engine.sgd_step(lr, momentum)
        return logits, loss
