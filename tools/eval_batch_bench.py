#!/usr/bin/env python3
"""Anomaly-eval images/s (bench.py's leg: 78 pairs at 128 px) against the forward batch size of evaluate.super_resolve_u8."""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import Opt
from srad_amd import evaluate as E
from srad_amd.nets import DRCT
from srad_amd.spec import synth_pairs


class EvalOpt:
    rgb_range = 255.0


y, sr_u8, hr_u8 = synth_pairs(21, 57, 128, 1, seed=0)
pairs = []
for s_img, h_img in zip(sr_u8, hr_u8):
    lr = s_img.reshape(32, 4, 32, 4, 1).astype(np.float32).mean((1, 3))
    pairs.append((np.clip(np.rint(lr), 0, 255).astype(np.uint8), h_img))
good, bad = pairs[:21], pairs[21:]
orig = E.super_resolve_u8
for prec in sys.argv[1:] or ["bf16x3", "bf16"]:
    o = Opt()
    o.precision, o.use_graph = prec, False
    torch.manual_seed(1)
    m = DRCT(o).cuda().eval()
    for batch in (8, 13, 16, 26, 39, 78):
        E.super_resolve_u8 = lambda model, lr, hr, rr, batch=batch: orig(model, lr, hr, rr, batch=batch)
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            E.evaluate_on_test(EvalOpt, m, good, bad)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                E.evaluate_on_test(EvalOpt, m, good, bad)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
        print(f"{prec} batch {batch}: {78 / dt:.0f} images/s ({dt * 1e3:.1f} ms per split)", flush=True)
    del m
