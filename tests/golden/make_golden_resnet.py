"""Generates tests/golden/resnet18_tiny.npz by running the REFERENCE's models/resnet18.py (imported by file path
from /root/reference; `import models` itself fails upstream) with num_filters=8, seeded init and input.
Build container only."""
import importlib.util
import os

import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_resnet18", "/root/reference/models/resnet18.py")
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
torch.manual_seed(0)
net = m.resnet18(num_classes=7, num_filters=8)
g = torch.Generator().manual_seed(1)
with torch.no_grad():       # move BN affine / running stats and fc bias away from their trivial init
    for name, t in list(net.named_parameters()) + list(net.named_buffers()):
        if name.endswith("bn1.weight") or name.endswith("bn2.weight") or name.endswith("downsample.1.weight"):
            t.copy_(1 + 0.2 * torch.randn(t.shape, generator=g))
        elif name.endswith(".bias"):
            t.copy_(0.2 * torch.randn(t.shape, generator=g))
        elif name.endswith("running_mean"):
            t.copy_(0.1 * torch.randn(t.shape, generator=g))
        elif name.endswith("running_var"):
            t.copy_(1 + torch.rand(t.shape, generator=g))
x = torch.randn(3, 1, 64, 64, generator=g)
y = torch.tensor([1, 5, 2])
out = {"x": x.numpy(), "y": y.numpy()}
for k, v in net.state_dict().items():
    if not k.endswith("num_batches_tracked"):
        out["param:" + k] = v.numpy().copy()
net.train()
logits = net(x)
loss = torch.nn.CrossEntropyLoss()(logits, y)
loss.backward()
out["logits_train"] = logits.detach().numpy()
out["loss"] = np.array(loss.item(), dtype=np.float32)
for k, v in net.named_parameters():
    out["grad:" + k] = v.grad.numpy()
for k, v in net.state_dict().items():
    if k.endswith("running_mean") or k.endswith("running_var"):
        out["after:" + k] = v.numpy().copy()
net.eval()
with torch.no_grad():
    out["logits_eval"] = net(x).numpy()
np.savez_compressed(os.path.join(here, "resnet18_tiny.npz"), **out)
print("saved", len(out), "arrays;", sum(p.numel() for p in net.parameters()), "parameters; loss", loss.item())
# full-size parameter count of the reference (SURVEY: 11 201 020 for 60 classes, 64 filters)
print(sum(p.numel() for p in m.resnet18(num_classes=60, num_filters=64).parameters()))
