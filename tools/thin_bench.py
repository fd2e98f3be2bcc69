import os, sys
sys.path.insert(0, "/root/repo/tools")
sys.argv = [sys.argv[0], "bf16", "8192"]
import gemm_bench as G
import torch
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for name, K, N, B, H, W, taps in [("head 4->20 @256", 4, 20, 8, 256, 256, 9), ("tail 40->4 @256", 40, 4, 8, 256, 256, 9),
                                      ("tail 80->4 @128", 80, 4, 8, 128, 128, 9), ("tail 80->4 @64", 80, 4, 8, 64, 64, 9),
                                      ("down 20->20 @256 s1", 20, 20, 8, 256, 256, 9), ("up1x1 80->20 @256", 80, 20, 8, 256, 256, 1),
                                      ("up1x1 80->40 @128", 80, 40, 8, 128, 128, 1), ("down 40->80 @64", 40, 80, 8, 64, 64, 9),
                                      ("down 20->40 @128", 20, 40, 8, 128, 128, 9), ("up 80->320 @128", 80, 320, 8, 128, 128, 9)]:
        G.M = B * H * W
        t = G.gemm(K, N, ntaps=taps, B=B, H=H, W=W, iters=20)
        byt = 4.0 * B * H * W * (K + N)
        print(f"{name:24s}: {t:8.1f} us   {byt / t / 1e6:7.2f} TB/s on in+out bytes")
