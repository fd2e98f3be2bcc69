"""C2 with the batch handed over as HOST buffers: pinned LR batch -> HBM, forward (one hipGraph replay), HR batch -> pinned host
memory, one synchronisation per step.  bench.py's `value` has the inputs resident in HBM (the C ABI takes device pointers); this
is the PCIe-inclusive rate DESIGN.md quotes beside it.  Run on the GPU box:  python tools/pcie_rate.py [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from srad_amd.nets import DRCT  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    opt = bench.Opt()
    model = DRCT(opt).to(dev).eval()
    B, H, W, s = 4, 32, 32, 4
    xh = (torch.rand(B, 1, H, W) * 255.0).pin_memory()
    yh = torch.empty(B, 1, H * s, W * s).pin_memory()
    xd = torch.empty(B, 1, H, W, device=dev)
    out = {}
    with torch.no_grad():
        for mode in ("resident", "host_buffers"):
            for _ in range(5):
                xd.copy_(xh, non_blocking=True)
                y = model(xd)
                yh.copy_(y, non_blocking=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                if mode == "host_buffers":
                    xd.copy_(xh, non_blocking=True)
                y = model(xd)
                if mode == "host_buffers":
                    yh.copy_(y, non_blocking=True)
                    torch.cuda.synchronize()          # the caller reads the HR batch before handing over the next one
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            out[mode] = {"ms_per_step": round(dt * 1e3, 4), "hr_mpixels_per_s": round(B * H * s * W * s / dt / 1e6, 2)}
    out["bytes_per_step"] = {"h2d": xh.numel() * 4, "d2h": yh.numel() * 4}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
