"""Drop-in for the reference's models/resnet18.py: `resnet18(num_classes=..., num_filters=...)` -> a torch module whose
forward takes (B, 1, H, W) images and returns logits (models/resnet18.py:235-254,266-275).  Parameter names follow
the reference's state_dict (conv1.weight, bn1.weight, layer1.0.conv1.weight, ..., fc.bias) with dots replaced by
underscores for attribute access; `state_dict_reference()` gives them back with the original keys.
The arithmetic runs in libsar_hip.so through sar_amd.resnet.ResNet18."""
import torch

from sar_amd.resnet import ResNet18


class _ResNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, engine, *params):
        ctx.engine = engine
        return engine.forward(x, training=True)

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.engine
        dx = eng.backward(dlogits.contiguous(), need_dx=ctx.needs_input_grad[0])
        return (dx, None) + tuple(eng.g[k].clone() for k in eng.shapes)


class ResNet(torch.nn.Module):
    def __init__(self, num_classes=1000, num_filters=64, device="cuda", seed=0, mfma=None):
        """mfma (not in the reference): "fp32" | "f32_split" | "f32_split_bf16x6" -- sar_amd.resnet.ResNet18; None = its default"""
        super().__init__()
        self.engine = ResNet18(num_classes=num_classes, num_filters=num_filters, device=device, seed=seed, mfma=mfma)
        self._names = list(self.engine.shapes)
        for k in self._names:
            self.register_parameter(k.replace(".", "_"), torch.nn.Parameter(self.engine.p[k]))

    def state_dict_reference(self):
        return self.engine.state_dict()

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return _ResNetFunction.apply(x, self.engine, *[getattr(self, k.replace(".", "_")) for k in self._names])
        return self.engine.forward(x, training=self.training)


def resnet18(pretrained=False, progress=True, **kwargs):
    """models/resnet18.py:266-275 (the reference's `pretrained` branch references undefined names and cannot work)."""
    if pretrained:
        raise NotImplementedError("no pretrained weights exist for the 1-channel ResNet-18")
    return ResNet(**kwargs)
