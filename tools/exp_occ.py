import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")): sys.path.insert(0, p)
import torch
from sar_amd import _lib
lib = _lib.load()
pr = torch.cuda.get_device_properties(0)
print(pr)
for a in dir(pr):
    if 'shared' in a.lower() or 'regs' in a.lower() or 'multi' in a.lower(): print(a, getattr(pr, a))
for lds in (8, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80):
    print("lds %2d KB -> blocks/CU: 128x128 tile %d, 64x256 tile %d" % (lds, lib.sar_debug_occupancy(0, lds*1024), lib.sar_debug_occupancy(1, lds*1024)))
