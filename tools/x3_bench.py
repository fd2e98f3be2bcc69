#!/usr/bin/env python3
"""C2 forward (DRCT-L x4, batch 4, 32x32 LR) in the three precisions: ms per step (hipGraph replay), per-kernel-class
event times of an eager pass, and the error against the fp32 mode.  SRAD_X3_FM=16|32 picks mlp_block's rows per workgroup."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import Opt
from srad_amd import _lib as L
from srad_amd.nets import DRCT


def run(prec, x, sd=None, steps=50):
    o = Opt()
    o.precision, o.use_graph = prec, True
    torch.manual_seed(1)
    m = DRCT(o).cuda().eval()
    if sd is not None:
        m.load_state_dict(sd)
    with torch.no_grad():
        for _ in range(4):
            y = m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        m.use_graph = False
        L.prof_enable(True)
        m(x)
        torch.cuda.synchronize()
        L.prof_collect()
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        prof = L.prof_collect()
        L.prof_enable(False)
    ks = {k: (v["launches"] // 5, round(v["ms"] * 1e3 / v["launches"], 2), round(v["ms"] / 5, 3)) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    return m, y.clone(), dt * 1e3, ks


if __name__ == "__main__":
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(4, 1, 32, 32, generator=g) * 255).cuda()
    m32, y32, t32, k32 = run("fp32", x)
    out = {"fp32": {"ms": round(t32, 3), "kernels": k32}}
    for prec in sys.argv[1:] or ["bf16", "bf16x3"]:
        m, y, t, ks = run(prec, x, m32.state_dict())
        out[prec] = {"ms": round(t, 3), "max_rel_vs_fp32_mode": float(f"{float((y - y32).abs().max() / y32.abs().max()):.3e}"), "kernels": ks}
        del m
    print(json.dumps(out, indent=1))
