"""Drop-in for the reference's `models` package (models/__init__.py:1-6).  Only the north-star models are
provided: `models.stgcn` (ST-GCN), its sibling `models.stgin` (graph isomorphism convolution) and `models.resnet` /
`models.resnet18` (VirtualRadar + ResNet-18).
Sub-modules are imported lazily so that `import models` works on a CPU-only box."""
