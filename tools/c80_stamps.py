#!/usr/bin/env python3
"""Phase stamps of the 80-channel conv kernel inside a DRN-L forward at the C3 shape (diagnostic build of the kernel:
SRAD_C80_STAMP=1 makes the launcher run the stamped instance, synchronously, and print medians to stderr).
    SRAD_C80_STAMP=1 python tools/c80_stamps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from srad_amd.nets import DRN


class Opt:
    n_colors, n_blocks, n_feats, negval, rgb_range, scale = 3, 40, 20, 0.2, 255.0, [2, 4]
    precision, use_graph = "bf16", False


torch.manual_seed(1)
m = DRN(Opt()).cuda().eval()
x = torch.rand(8, 3, 64, 64, device="cuda") * 255
with torch.no_grad():
    y = m(x)
torch.cuda.synchronize()
print("done", [tuple(t.shape) for t in y])
