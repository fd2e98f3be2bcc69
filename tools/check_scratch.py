#!/usr/bin/env python3
"""No kernel of libsrad.so may use scratch (private memory): a spilled or dynamically indexed register array is a silent
4x slowdown (found on the 64-row mlp_block instances in round 2: a lambda that stopped being inlined sent its array
arguments to scratch).  Reads the code objects' metadata (`.private_segment_fixed_size`); exit code 1 on a hit.

    python tools/check_scratch.py [path/to/libsrad.so]
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_with_scratch(so_path):
    tmp = tempfile.mkdtemp(prefix="srad_scratch_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(so_path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        found, total = [], 0
        for co in sorted(glob.glob(local + ".*gfx950*")):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
            name = None
            for line in notes.splitlines():
                m = re.search(r"\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", line)
                if m and name is not None:
                    total += 1
                    if int(m.group(1)) > 0:
                        found.append((name, int(m.group(1))))
                    name = None
        return found, total
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "anomaly-detection-super-resolution_amd", "libsrad.so")
    found, total = kernels_with_scratch(so)
    print(f"{total} kernels in {so}; {len(found)} use scratch")
    for name, size in found:
        print(f"  {size:6d} bytes/lane  {name}")
    return 1 if found or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
