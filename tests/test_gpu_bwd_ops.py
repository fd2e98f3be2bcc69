"""GPU: the backward kernels one by one against torch autograd of the same op (fp32 statement on the CPU).
Tolerances as for the forward operators: fp32 mode 1e-4 relative (exact-fp32 MFMA, atomics change only the
summation order), bf16 mode 3e-2 relative."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {"fp32": 1e-4, "bf16": 3e-2}


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("M,K,N", [(4096, 180, 540), (1000, 212, 32), (77, 308, 180), (8192, 360, 180), (130, 36, 4), (5000, 80, 80)])
def test_wgrad_linear(dev, prec, M, K, N):
    from srad_amd import ops
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K + 8, generator=g)[:, :K]
    dy = torch.randn(M, N + 4, generator=g)[:, :N]
    rs = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25])[: max(1, M // 1024 + 1)]
    dw, db = ops.wgrad(dy.to(dev), x.to(dev), N, K, precision=prec)
    assert _rel(dw[:, :, 0], dy.t() @ x) < TOL[prec]
    assert _rel(db, dy.sum(0)) < TOL[prec]
    # alpha and accumulation into a non-zero buffer are exercised by the engine tests; per-sample row scale here
    if M % 1024 == 0:
        B = M // 1024
        rs = (torch.arange(B) % 2).float() * 1.25
        import ctypes as C
        from srad_amd import _lib as L
        dw2 = torch.zeros(N, K, 1, device=dev)
        db2 = torch.zeros(N, device=dev)
        xd, dyd, rsd = x.to(dev), dy.to(dev), rs.to(dev)
        L.check(L.lib().srad_op_wgrad(L.PRECISIONS[prec], L.dptr(dyd), dyd.stride(0), L.dptr(xd), xd.stride(0), B, 32, 32, N, K, 1, 1,
                                      L.dptr(rsd), 0.5, L.dptr(dw2), L.dptr(db2), ops.wgrad_workspace(dev), L.current_stream_ptr()), "op_wgrad")
        sc = rs.repeat_interleave(1024)[:, None] * 0.5
        assert _rel(dw2[:, :, 0], (dy * sc).t() @ x) < TOL[prec]
        assert _rel(db2, (dy * sc).sum(0)) < TOL[prec]


@pytest.mark.parametrize("storage", ["f32", "xh", "xh_yh"])
@pytest.mark.parametrize("M,N,K", [(2048, 540, 180), (1024, 180, 360), (2048, 32, 180), (1024, 360, 244), (3072, 128, 64), (1536, 180, 180)])
def test_wgrad_linear_deferred(dev, storage, M, N, K):
    """The training step's route (queue -> one deferred launch -> reduce), with the operands as fp32, X as bf16, or both as
    bf16 with dY pre-multiplied by its per-sample factor; whole 128-row steps (mask-free path) and not (M = 1536 ... )."""
    from srad_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(M, N, generator=g).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    rps = 256
    rs = (torch.floor(0.7 + torch.rand(M // rps, generator=g)) / 0.7).to(dev)
    dys = dy * rs.repeat_interleave(rps).unsqueeze(1)
    dw_ref = dys.double().t() @ x.double()
    db_ref = dys.double().sum(0)
    if storage == "f32":
        dw, db = ops.wgrad_linear_deferred(dy, x, row_scale=rs, rps=rps, alpha=0.5)
    elif storage == "xh":
        dw, db = ops.wgrad_linear_deferred(dy, x.to(torch.bfloat16), row_scale=rs, rps=rps, alpha=0.5)
    else:
        dw, db = ops.wgrad_linear_deferred(dys.to(torch.bfloat16), x.to(torch.bfloat16), alpha=0.5)
    # operands rounded to bf16 (relative 2^-9 each), fp32 accumulation over M rows
    assert _rel(dw, 0.5 * dw_ref.float()) < 6e-3
    assert _rel(db, 0.5 * db_ref.float()) < 6e-3
    # accumulation: a second pass adds onto the first
    if storage == "f32":
        L = ops.L
        L.check(L.lib().srad_op_wgrad_deferred(L.PRECISIONS["bf16"], L.dptr(dy), N, 0, L.dptr(x), K, 0, M, N, K, L.dptr(rs), rps, 0.5, L.dptr(dw),
                                               L.dptr(db), ops.wgrad_workspace(dy.device), L.current_stream_ptr()), "op_wgrad_deferred")
        assert _rel(dw, dw_ref.float()) < 6e-3 and _rel(db, db_ref.float()) < 6e-3


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", [(2, 32, 32, 180, 64, 1), (1, 17, 23, 4, 180, 1), (2, 16, 12, 64, 4, 1),
                                                   (1, 32, 32, 20, 20, 2), (1, 15, 11, 40, 80, 2),
                                                   (2, 64, 64, 80, 80, 1), (1, 17, 23, 80, 80, 1),       # 80 -> 80: the 80-wide tile kernel (bf16)
                                                   # C -> C, W % 32 == 0, >= 8192 pixels: the nine-taps-per-workgroup kernel (bf16)
                                                   (3, 64, 64, 80, 80, 1), (2, 64, 64, 40, 40, 1), (1, 96, 96, 20, 20, 1),
                                                   (1, 70, 128, 64, 64, 1), (1, 67, 128, 80, 80, 1), (2, 64, 64, 12, 12, 1)])
def test_wgrad_conv3x3(dev, prec, B, H, W, Cin, Cout, stride):
    from srad_amd import ops
    g = torch.Generator().manual_seed(B * H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, requires_grad=True)
    b = torch.zeros(Cout, requires_grad=True)
    y = F.conv2d(x, w, b, stride=stride, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    x_rows = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous()
    dy_rows = dy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous()
    dw, db = ops.wgrad(dy_rows.to(dev), x_rows.to(dev), Cout, Cin, ntaps=9, B=B, H=H, W=W, stride=stride, precision=prec)
    assert _rel(dw.reshape(Cout, Cin, 3, 3), w.grad) < TOL[prec]
    assert _rel(db, b.grad) < TOL[prec]


def test_wgrad_conv3x3_strided_operands(dev):
    """The nine-taps-per-workgroup kernel on operands that are column blocks of wider row-major buffers (DRN's [up | skip]
    concatenation buffers: row stride 2 C; the gradient block starts at a column offset)."""
    from srad_amd import ops
    B, H, W, Cc = 2, 64, 64, 80
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, Cc, H, W, generator=g)
    w = torch.randn(Cc, Cc, 3, 3, generator=g, requires_grad=True)
    b = torch.zeros(Cc, requires_grad=True)
    y = F.conv2d(x, w, b, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    x_wide = torch.randn(B * H * W, 2 * Cc, generator=g)
    dy_wide = torch.randn(B * H * W, Cc + 8, generator=g)
    x_wide[:, Cc:] = x.permute(0, 2, 3, 1).reshape(-1, Cc)
    dy_wide[:, 4:4 + Cc] = dy.permute(0, 2, 3, 1).reshape(-1, Cc)
    xd, dyd = x_wide.to(dev), dy_wide.to(dev)
    dw, db = ops.wgrad(dyd[:, 4:4 + Cc], xd[:, Cc:], Cc, Cc, ntaps=9, B=B, H=H, W=W, precision="bf16")
    assert _rel(dw.reshape(Cc, Cc, 3, 3), w.grad) < TOL["bf16"]
    assert _rel(db, b.grad) < TOL["bf16"]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_dgrad_linear_and_activation_modes(dev, prec):
    from srad_amd import ops
    g = torch.Generator().manual_seed(3)
    M, K, N = 2048, 212, 424
    dy = torch.randn(M, K, generator=g)           # gradient of fc2's output (K = d)
    w2 = torch.randn(K, N, generator=g) / math.sqrt(N)   # fc2.weight [d, hidden]
    h = torch.randn(M, N, generator=g, requires_grad=True)
    F.linear(F.gelu(h), w2).backward(dy)
    rs = ((torch.arange(2) % 2).float() * 1.1 + 0.5)
    dx = ops.dgrad(dy.to(dev), w2.to(dev), precision=prec)
    assert _rel(dx, dy @ w2) < TOL[prec]
    dh = ops.dgrad(dy.to(dev), w2.to(dev), r=h.detach().to(dev), rmode=1, precision=prec)
    assert _rel(dh, h.grad) < TOL[prec]
    y = torch.randn(M, N, generator=g)
    dl = ops.dgrad(dy.to(dev), w2.to(dev), r=y.to(dev), rmode=2, slope=0.2, alpha=0.5, precision=prec)
    assert _rel(dl, (dy @ w2) * 0.5 * torch.where(y > 0, 1.0, 0.2)) < TOL[prec]
    import ctypes as C
    from srad_amd import _lib as L
    # per-sample scale (two samples of 1024 rows)
    dyd, wd, rsd = dy.to(dev), w2.reshape(K, N, 1).contiguous().to(dev), rs.to(dev)
    out = torch.empty(M, N, device=dev)
    nb = L.lib().srad_op_gemm_scratch_bytes(L.PRECISIONS[prec], N, K, 1)
    sc = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
    off = (-sc.data_ptr()) % 256
    L.check(L.lib().srad_op_dgrad(L.PRECISIONS[prec], L.dptr(dyd), K, 2, 32, 32, K, L.dptr(wd), N, 1, None, 0, 0, 0.0, 1.0,
                                  L.dptr(rsd), L.dptr(out), N, C.c_void_p(sc.data_ptr() + off), C.c_size_t(nb),
                                  L.current_stream_ptr()), "op_dgrad")
    assert _rel(out, (dy @ w2) * rs.repeat_interleave(1024)[:, None]) < TOL[prec]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 32, 180, 64), (1, 16, 24, 64, 256), (2, 8, 8, 4, 180)])
def test_dgrad_conv3x3(dev, prec, B, H, W, Cin, Cout):
    from srad_amd import ops
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    y = F.conv2d(x, w, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dy_rows = dy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous()
    dx = ops.dgrad(dy_rows.to(dev), w.to(dev), B=B, H=H, W=W, precision=prec)
    assert _rel(dx, x.grad.permute(0, 2, 3, 1).reshape(-1, Cin)) < TOL[prec]


@pytest.mark.parametrize("rows,C", [(4096, 180), (1000, 308), (37, 64)])
def test_layernorm_bwd(dev, rows, C):
    from srad_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C + 4, generator=g) * 2 + 0.3)[:, :C].clone().requires_grad_(True)
    gam = torch.randn(C, generator=g, requires_grad=True)
    bet = torch.randn(C, generator=g, requires_grad=True)
    dy = torch.randn(rows, C, generator=g)
    dres = torch.randn(rows, C, generator=g)
    prev = torch.randn(rows, C, generator=g)
    F.layer_norm(x, (C,), gam, bet, 1e-5).backward(dy)
    xd = torch.zeros(rows, C + 4, device=dev)
    xd[:, :C] = x.detach().to(dev)
    out, dg, db = ops.layernorm_bwd(dy.to(dev), xd[:, :C], gam.detach().to(dev))
    assert _rel(out, x.grad) < 1e-4 and _rel(dg, gam.grad) < 1e-4 and _rel(db, bet.grad) < 1e-4
    out2, _, _ = ops.layernorm_bwd(dy.to(dev), xd[:, :C], gam.detach().to(dev), dres=dres.to(dev), out=prev.to(dev).clone())
    assert _rel(out2, x.grad + dres + prev) < 1e-4


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16io"])
@pytest.mark.parametrize("d,heads,shift", [(180, 6, 0), (180, 6, 4), (212, 4, 4), (244, 2, 0), (276, 6, 4), (308, 4, 0), (32, 2, 3),
                                           (48, 2, 5), (64, 2, 0)])
def test_window_attention_bwd(dev, d, heads, shift, prec):
    if prec == "bf16io" and d // heads > 128:
        pytest.skip("the all-bf16 kernel takes head dims <= 128")
    from oracle import sr_ref as R
    from srad_amd import ops
    B, H, W, ws = 2, 16, 24, 8
    g = torch.Generator().manual_seed(d + shift)
    T = B * H * W
    qkv = (torch.randn(T, 3 * d, generator=g) * 0.7).requires_grad_(True)
    table = (torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5).requires_grad_(True)
    dout = torch.randn(T, d, generator=g)

    def ref(qkv, table):
        x = qkv.view(B, H, W, 3 * d)
        if shift:
            x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
        xw = R.window_partition(x, ws).view(-1, ws * ws, 3, heads, d // heads).permute(2, 0, 3, 1, 4)
        q, k, v = xw[0] * (d // heads) ** -0.5, xw[1], xw[2]
        a = q @ k.transpose(-2, -1)
        bias = table[R.rel_pos_index(ws).view(-1)].view(ws * ws, ws * ws, -1).permute(2, 0, 1)
        a = a + bias.unsqueeze(0)
        if shift:
            mask = R.calculate_mask(H, W, ws, shift)
            nW = mask.shape[0]
            a = (a.view(B, nW, heads, ws * ws, ws * ws) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, ws * ws, ws * ws)
        o = (torch.softmax(a, -1) @ v).transpose(1, 2).reshape(-1, ws, ws, d)
        o = R.window_reverse(o, ws, H, W)
        if shift:
            o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
        return o.reshape(T, d)

    out_ref = ref(qkv, table)
    out_ref.backward(dout)
    out = ops.window_attention(qkv.detach().to(dev), table.detach().to(dev), B, H, W, ws, shift, heads)
    assert _rel(out, out_ref.detach()) < 1e-4
    if prec == "bf16io":      # the training step's kernel: bf16 operands in per-head slots (NaN in dO's padding), bf16 results
        dqkv, dtable = ops.window_attention_bwd_bf16io(qkv.detach().to(dev), dout.to(dev), table.detach().to(dev), B, H, W, ws, shift, heads)
    else:
        dqkv, dtable = ops.window_attention_bwd(qkv.detach().to(dev), dout.to(dev), table.detach().to(dev), B, H, W, ws, shift, heads,
                                                precision=prec)
    # bf16 mode: q, k, v, dO, P and dS are rounded to bf16 for the MFMA (fp32 accumulation, fp32 softmax statistics)
    tol = 2e-4 if prec == "fp32" else 2e-2
    assert _rel(dqkv, qkv.grad) < tol
    assert _rel(dtable, table.grad) < tol


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("ws,shift,B,H,W,d,heads", [(2, 1, 2, 8, 6, 180, 6), (2, 0, 1, 4, 4, 212, 4), (4, 2, 2, 8, 12, 180, 6), (4, 0, 1, 8, 8, 244, 2),
                                                    (16, 8, 1, 32, 48, 180, 6), (16, 0, 1, 16, 32, 212, 4), (16, 8, 1, 32, 32, 244, 2),
                                                    (16, 8, 1, 32, 16, 276, 6), (16, 0, 1, 16, 16, 308, 4), (12, 6, 1, 24, 24, 60, 2),
                                                    (3, 1, 1, 9, 6, 32, 2), (1, 0, 1, 4, 4, 32, 2)])
def test_window_attention_bwd_other_window_sizes(dev, prec, ws, shift, B, H, W, d, heads):
    """The reference's CLI builds windows of 2, 4, 8 and 16 (window_size = img_size // 4, src/main.py:218-219,286): the general
    attention backward (window_attn_bwd_gen_kernel: N = ws^2 tokens padded to blocks of 64, one to four blocks) against autograd
    of the oracle's attention; several windows per side with the shift mask, every DRCT-L head dim.  fp32 MFMAs in both modes."""
    from oracle import sr_ref as R
    from srad_amd import ops
    g = torch.Generator().manual_seed(ws * 1000 + d + shift)
    T, N, hd = B * H * W, ws * ws, d // heads
    qkv = (torch.randn(T, 3 * d, generator=g) * 0.7).requires_grad_(True)
    table = (torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5).requires_grad_(True)
    dout = torch.randn(T, d, generator=g)
    x = qkv.view(B, H, W, 3 * d)
    if shift:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = R.window_partition(x, ws).view(-1, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    mask = R.calculate_mask(H, W, ws, shift) if shift else None
    o = R.attention_from_qkv(xw[0] * hd ** -0.5, xw[1], xw[2], table, ws, mask)             # src/drct.py:282-299
    o = R.window_reverse(o.transpose(1, 2).reshape(-1, ws, ws, d), ws, H, W)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    o.reshape(T, d).backward(dout)
    dqkv, dtable = ops.window_attention_bwd(qkv.detach().to(dev), dout.to(dev), table.detach().to(dev), B, H, W, ws, shift, heads, precision=prec)
    e1, e2 = _rel(dqkv, qkv.grad), _rel(dtable, table.grad)
    print(f"attention backward ws={ws} shift={shift} d={d} heads={heads} {prec}: dqkv {e1:.2e} dtable {e2:.2e}")
    assert e1 < 2e-4 and e2 < 2e-4


def test_adam_and_l1_grad_match_torch(dev):
    import ctypes as C
    from srad_amd import _lib as L
    g = torch.Generator().manual_seed(11)
    n = 100003
    p0 = torch.randn(n, generator=g)
    for wd in (0.0, 1e-2):
        p_ref = p0.clone().requires_grad_(True)
        opt = torch.optim.Adam([p_ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
        p = p0.clone().to(dev)
        m = torch.zeros(n, device=dev)
        v = torch.zeros(n, device=dev)
        for step in range(1, 4):
            grad = torch.randn(n, generator=g)
            p_ref.grad = grad.clone()
            opt.step()
            gd = (grad * 4).to(dev)
            L.check(L.lib().srad_adam_step(L.dptr(p), L.dptr(gd), L.dptr(m), L.dptr(v), n, 1e-3, 0.9, 0.999, 1e-8, wd, step, 0.25,
                                           L.current_stream_ptr()), "adam")
            assert float((p.cpu() - p_ref.detach()).abs().max()) < 2e-6
    a, b = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    b[:10] = a[:10]
    a.requires_grad_(True)
    F.l1_loss(a, b).backward()
    out = torch.empty(1000, device=dev)
    ad, bd = a.detach().to(dev), b.to(dev)
    L.check(L.lib().srad_l1_grad(L.dptr(ad), L.dptr(bd), L.dptr(out), 1000, 1.0 / 1000, L.current_stream_ptr()), "l1_grad")
    assert torch.equal(out.cpu(), a.grad)


@pytest.mark.parametrize("M,d,m,KA", [(64, 180, 360, 32), (8192, 212, 424, 32), (96, 244, 488, 0), (8192 + 32, 276, 276, 32),
                                      (48, 308, 308, 180)])
def test_fused_mlp_backward_matches_autograd(dev, M, d, m, KA):
    """mlp_bwd_kernel (adjust data gradient -> fc2 / GELU' / fc1 data gradients -> LayerNorm2 backward -> projection data
    gradient) against fp32 autograd of the same chain (src/drct.py:300, 509-510, 184-190, 389-396).  Operands are rounded
    to bf16 for the MFMAs (fp32 accumulation): tolerance 2e-2 of each tensor's largest value."""
    from srad_amd import ops
    g = torch.Generator().manual_seed(M + d)
    rps = M // 4 if M % 4 == 0 else M
    nsmp = M // rps
    attn = torch.randn(M, d, generator=g, requires_grad=True)
    x0 = torch.randn(M, d, generator=g)
    w_proj = (torch.randn(d, d, generator=g) / math.sqrt(d)).requires_grad_(True)
    w1 = torch.randn(m, d, generator=g) / math.sqrt(d)
    b1 = torch.randn(m, generator=g) * 0.1
    w2 = torch.randn(d, m, generator=g) / math.sqrt(m)
    gam = (1 + 0.2 * torch.randn(d, generator=g)).requires_grad_(True)
    bet = (0.1 * torch.randn(d, generator=g)).requires_grad_(True)
    rs1 = torch.floor(0.7 + torch.rand(nsmp, generator=g)) / 0.7
    rs2 = torch.floor(0.7 + torch.rand(nsmp, generator=g)) / 0.7
    r1 = rs1.repeat_interleave(rps)[:, None]
    r2 = rs2.repeat_interleave(rps)[:, None]
    x1 = (x0 + r1 * (attn @ w_proj.t())).detach().requires_grad_(True)
    hpre = (F.layer_norm(x1, (d,), gam, bet, 1e-5) @ w1.t() + b1)
    hpre.retain_grad()
    x2 = x1 + r2 * (F.gelu(hpre) @ w2.t())
    x2.retain_grad()
    if KA:
        w_adj = torch.randn(KA, d, generator=g) / math.sqrt(d)
        slope, alpha = (0.2, 1.0) if KA == 32 else (1.0, 0.2)        # adjust1-4: LeakyReLU 0.2; adjust5: * 0.2, no activation
        y = F.leaky_relu(x2 @ w_adj.t(), slope) if KA == 32 else x2 @ w_adj.t()
        dA = torch.randn(M, KA, generator=g)
        (y * alpha).backward(dA)
        adjust = (dA.to(dev), y.detach().to(dev) if KA == 32 else None, slope, alpha, w_adj.to(dev))
        dx2_in = None
    else:
        dx2_ref = torch.randn(M, d, generator=g)
        x2.backward(dx2_ref)
        adjust, dx2_in = None, dx2_ref.to(dev)
    dx1_ref = x1.grad                                               # total gradient at x1 (residual + MLP branch)
    (x0 + r1 * (attn @ w_proj.t())).backward(dx1_ref)               # ... pushed through the projection
    out = ops.mlp_bwd(dx2_in, hpre.detach().to(dev), x1.detach().to(dev), gam.detach().to(dev), w1.to(dev), w2.to(dev),
                      rs2.to(dev), rps, adjust=adjust, proj=(w_proj.detach().to(dev), rs1.to(dev)))
    tol = 2e-2

    def close(a, b):
        return float((a.cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert close(out["dx2"], x2.grad) < tol
    assert close(out["dh"], hpre.grad) < tol
    assert close(out["dx1"], dx1_ref) < tol
    assert close(out["dO"], attn.grad) < tol
    assert close(out["dgamma"], gam.grad) < tol and close(out["dbeta"], bet.grad) < tol
    if KA == 32:
        assert close(out["dA"], dA * torch.where(y.detach() > 0, 1.0, slope)) < 1e-6


@pytest.mark.parametrize("M,d", [(64, 180), (8192, 244), (8192 + 16, 276), (32, 308)])
def test_fused_qkv_layernorm_backward_matches_autograd(dev, M, d):
    """lin_ln_bwd_kernel: out += dres + LayerNorm1'(dqkv @ Wqkv) with dgamma / dbeta, against fp32 autograd."""
    from srad_amd import ops
    g = torch.Generator().manual_seed(M + d)
    D = 308
    xbuf = torch.randn(M, D, generator=g)                             # rows of the RDG's dense buffer: LN reads [:, :d]
    x = xbuf[:, :d].clone().requires_grad_(True)
    gam = (1 + 0.2 * torch.randn(d, generator=g)).requires_grad_(True)
    bet = (0.1 * torch.randn(d, generator=g)).requires_grad_(True)
    w = torch.randn(3 * d, d, generator=g) / math.sqrt(d)
    dqkv = torch.randn(M, 3 * d, generator=g)
    dres = torch.randn(M, d, generator=g)
    prev = torch.randn(M, D, generator=g)
    (F.layer_norm(x, (d,), gam, bet, 1e-5) @ w.t()).backward(dqkv)
    xd = xbuf.to(dev)
    outbuf = prev.to(dev).clone()
    out, dg, db = ops.lin_ln_bwd(dqkv.to(dev), w.to(dev), xd[:, :d], gam.detach().to(dev), dres=dres.to(dev), out=outbuf[:, :d])
    ref = x.grad + dres + prev[:, :d]
    rel = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert rel(outbuf[:, :d], ref) < 2e-2
    assert torch.equal(outbuf[:, d:].cpu(), prev[:, d:])             # columns beyond d untouched
    assert rel(dg, gam.grad) < 2e-2 and rel(db, bet.grad) < 2e-2
