import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from srad_amd import _lib as L
dev = torch.device("cuda:0")
for M in (4096, 8192):
    attn = torch.randn(M, 320, device=dev).to(torch.bfloat16); short = torch.randn(M, 320, device=dev); y = torch.empty(M, 320, device=dev)
    w = torch.randn(512 * 512, device=dev) * 0.05
    scratch = torch.empty(16 << 20, dtype=torch.uint8, device=dev); off = (-scratch.data_ptr()) % 256
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for d, m, no in [(180, 360, 32), (212, 424, 32), (244, 488, 32), (276, 276, 32), (308, 308, 180)]:
            row = []
            for fm in (16, 32, 64):
                us = C.c_float()
                L.check(L.lib().srad_bench_mlp_block(M, d, m, no, L.dptr(attn), L.dptr(short), L.dptr(y), L.dptr(w), C.c_void_p(scratch.data_ptr() + off),
                                                     C.c_size_t(scratch.numel() - off), fm << 8, 100, C.byref(us), L.current_stream_ptr()), "bench")
                row.append(f"fm{fm}={us.value:6.1f}")
            print(f"M={M} d={d}: " + "  ".join(row))
torch.cuda.synchronize()
