#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running THE REFERENCE ITSELF
(/root/reference, imported read-only in the build container) on deterministic synthetic
weights and inputs.  The fixtures hold only data (configs, inputs, expected outputs); the
weights are re-synthesised from ``srad_amd.spec.synth_state`` by name+seed on both sides, so
no reference source or checkpoint is stored.

Run (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_golden.py

The reference imports skimage / imageio / torchvision at module top without using them on
this path; they are absent from the image, so empty stand-in modules are
registered before the import.  Nothing is written to /root/reference.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DRN_GAIN = 0.5     # conv gain for DRN fixtures: 80 residual RCABs at gain 1 overflow the useful range
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SRAD_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)


def _stub_unused_imports():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    for n in ["skimage", "skimage.color", "skimage.exposure", "imageio", "imageio.v2",
              "torchvision", "torchvision.transforms", "torchvision.datasets"]:
        stub(n)
    stub("skimage.metrics", structural_similarity=None, peak_signal_noise_ratio=None)
    for parent, child in [("skimage", "color"), ("skimage", "metrics"), ("skimage", "exposure"),
                          ("imageio", "v2"), ("torchvision", "transforms"), ("torchvision", "datasets")]:
        setattr(sys.modules[parent], child, sys.modules[f"{parent}.{child}"])


def main():
    _stub_unused_imports()
    sys.path.insert(0, REF)
    import torch
    from srad_amd import spec as S
    from src.main import DRCT as DRCTOpt, DRN as DRNOpt, setup_opt_drct, setup_opt_drn
    from src.drct import DRCT
    from src.drn import DRN
    from src.model import DownBlock
    from src import metrics as RM
    from src.trainer import quantize
    from sklearn.metrics import roc_auc_score

    torch.set_num_threads(8)
    torch.manual_seed(0)

    def ref_drct(cfg: S.DRCTConfig):
        opt = DRCTOpt()
        opt = setup_opt_drct(opt, 0.0, 11, "mvtec", "grid", False, cfg.upscale, True, cfg.in_chans, 1, 1,
                             cfg.img_size * cfg.upscale, cfg.img_size, "", "", "", 1, 1, 1, 0.0, 0, ".", "1*L1")
        assert opt.window_size == cfg.window_size
        opt.depths = (6,) * cfg.n_rdg
        opt.num_heads = (cfg.num_heads,) * cfg.n_rdg
        m = DRCT(opt)
        sp = S.drct_spec(cfg)
        sd = m.state_dict()
        assert list(sd.keys()) == list(sp.keys()), "state-dict key order differs from spec"
        for k, (shp, _) in sp.items():
            assert tuple(sd[k].shape) == tuple(shp), (k, sd[k].shape, shp)
        return m, sp

    def load_synth(m, sp, seed, cfg):
        st = S.synth_state(sp, seed=seed, gain=1.0, cfg=cfg)
        # integer / mask buffers must equal what the reference computes itself
        for k, (_, kind) in sp.items():
            if kind in ("index", "mask"):
                assert np.array_equal(st[k], m.state_dict()[k].numpy()), k
        m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
        return st

    out = {}

    # ---------------------------------------------------------------- DRCT cases
    drct_cases = {
        # name: (cfg, B, H, W, seed)
        "drct_full_gray_x4": (S.DRCTConfig(in_chans=1, img_size=32, window_size=8, upscale=4, n_rdg=12), 1, 32, 32, 11),
        "drct_r2_rgb_x4": (S.DRCTConfig(in_chans=3, img_size=32, window_size=8, upscale=4, n_rdg=2), 2, 32, 32, 12),
        "drct_r2_gray_x4_dyn64": (S.DRCTConfig(in_chans=1, img_size=32, window_size=8, upscale=4, n_rdg=2), 1, 64, 32, 13),
        "drct_r1_gray_x4_ws4": (S.DRCTConfig(in_chans=1, img_size=16, window_size=4, upscale=4, n_rdg=1), 2, 16, 16, 14),
        "drct_r1_gray_x8_ws2": (S.DRCTConfig(in_chans=1, img_size=8, window_size=2, upscale=8, n_rdg=1), 1, 8, 8, 15),
        "drct_r1_gray_x4_ws16": (S.DRCTConfig(in_chans=1, img_size=64, window_size=16, upscale=4, n_rdg=1), 1, 64, 64, 16),
    }
    for name, (cfg, B, H, W, seed) in drct_cases.items():
        m, sp = ref_drct(cfg)
        load_synth(m, sp, seed, cfg)
        m.eval()
        x = torch.from_numpy(S.synth_image(name, (B, cfg.in_chans, H, W), seed=1))
        taps = {}
        if cfg.n_rdg <= 2:
            hooks = [m.patch_embed.register_forward_hook(lambda mod, i, o: taps.__setitem__("embed", o.detach().clone())),
                     m.layers[0].register_forward_hook(lambda mod, i, o: taps.__setitem__("rdg0", o.detach().clone()))]
        with torch.no_grad():
            y = m(x)
        print(name, "out", tuple(y.shape), "mean %.4f std %.4f absmax %.4f" % (y.mean(), y.std(), y.abs().max()))
        out[name + "/cfg"] = np.array([cfg.in_chans, cfg.img_size, cfg.window_size, cfg.upscale, cfg.n_rdg, seed], dtype=np.int64)
        out[name + "/x"] = x.numpy()
        out[name + "/y"] = y.numpy()
        for k, v in taps.items():
            step = max(1, v.shape[1] // 128)           # keep fixtures small: every step-th token
            out[f"{name}/tap_{k}"] = v[:, ::step].numpy()
            out[f"{name}/tap_{k}_step"] = np.array(step)
            print("   tap", k, "std %.4f" % v.std())
        if name == "drct_r2_rgb_x4":
            # G7: gradients under L1 loss (eval mode: DropPath off)
            m.zero_grad()
            xg = x.clone().requires_grad_(True)
            hr = torch.from_numpy(S.synth_image(name + "/hr", (B, cfg.in_chans, H * 4, W * 4), seed=2))
            loss = torch.nn.L1Loss(reduction="mean")(m(xg), hr)
            loss.backward()
            out[name + "/hr"] = hr.numpy()
            out[name + "/loss"] = np.array(loss.item(), dtype=np.float64)
            out[name + "/grad_x"] = xg.grad.numpy()
            gn = {k: p.grad for k, p in m.named_parameters()}
            keys = ["conv_first.weight", "layers.0.swin2.attn.relative_position_bias_table",
                    "layers.0.swin1.attn.proj.weight", "layers.1.swin5.mlp.fc2.bias",
                    "layers.0.adjust1.weight", "layers.0.swin1.norm1.weight", "conv_last.weight", "upsample.0.bias"]
            for k in keys:
                out[f"{name}/grad/{k}"] = gn[k].numpy()
            out[name + "/grad_names"] = np.array(list(gn.keys()))
            out[name + "/grad_l2"] = np.array([float(g.double().pow(2).sum().sqrt()) for g in gn.values()])

    # ---------------------------------------------------------------- DRN cases
    drn_cases = {
        "drn_x2_gray": (S.DRNConfig.for_scale(2, 1), 2, 16, 16, 21),
        "drn_x4_rgb": (S.DRNConfig.for_scale(4, 3), 1, 16, 12, 22),
        "drn_x4_gray": (S.DRNConfig.for_scale(4, 1), 1, 8, 8, 23),
        "drn_x8_gray": (S.DRNConfig.for_scale(8, 1), 1, 4, 4, 24),
    }
    for name, (cfg, B, H, W, seed) in drn_cases.items():
        opt = DRNOpt()
        opt = setup_opt_drn(opt, 0.0, 11, "mvtec", "carpet", False, cfg.scale, True, cfg.n_colors, 1, 1, 64, "", "", "",
                            1, 1, 1, 0.0, 0, ".", ".", "1*L1")
        assert (opt.n_blocks, opt.n_feats) == (cfg.n_blocks, cfg.n_feats)
        m = DRN(opt)
        sp = S.drn_spec(cfg)
        sd = m.state_dict()
        assert list(sd.keys()) == list(sp.keys()), "DRN key order differs"
        for k, (shp, _) in sp.items():
            assert tuple(sd[k].shape) == tuple(shp), (k, sd[k].shape, shp)
        st = S.synth_state(sp, seed=seed, gain=DRN_GAIN, cfg=cfg)
        for k in ("sub_mean.weight", "sub_mean.bias", "add_mean.weight", "add_mean.bias"):
            assert np.allclose(st[k], sd[k].numpy()), k          # MeanShift constants
        m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
        m.eval()
        x = torch.from_numpy(S.synth_image(name, (B, cfg.n_colors, H, W), seed=1))
        with torch.no_grad():
            ys = m(x)
        out[name + "/cfg"] = np.array([cfg.n_colors, cfg.scale, seed], dtype=np.int64)
        out[name + "/x"] = x.numpy()
        for j, y in enumerate(ys):
            out[f"{name}/y{j}"] = y.numpy()
            print(name, j, tuple(y.shape), "mean %.3f std %.3f" % (y.mean(), y.std()))
        # G6: dual model on the finest output
        dsp = S.dual_spec(cfg)
        d = DownBlock(opt, 2)
        assert list(d.state_dict().keys()) == list(dsp.keys())
        dst = S.synth_state(dsp, seed=seed + 100, gain=DRN_GAIN, cfg=cfg)
        d.load_state_dict({k: torch.from_numpy(v) for k, v in dst.items()})
        with torch.no_grad():
            out[name + "/dual"] = d(ys[-1]).numpy()

    if "--grad-only" not in sys.argv:
        np.savez_compressed(os.path.join(HERE, "sr_golden.npz"), **out)

    # ---------------------------------------------------------------- G7 for the other window sizes (round 3): sr_grad_golden.npz
    # The reference's remaining CLI presets build windows of 2, 4 and 16 (window_size = img_size // 4, src/main.py:218-219,286);
    # the same cases as above (1 RDG; x8 for the window-2 one), gradients of the reference's autograd under nn.L1Loss, eval
    # mode (DropPath off): dLoss/dx, the L2 norm of every parameter gradient and a few whole tensors.
    gr = {}
    for name in ("drct_r1_gray_x4_ws4", "drct_r1_gray_x8_ws2", "drct_r1_gray_x4_ws16"):
        cfg, B, H, W, seed = drct_cases[name]
        m, sp = ref_drct(cfg)
        load_synth(m, sp, seed, cfg)
        m.eval()
        x = torch.from_numpy(S.synth_image(name, (B, cfg.in_chans, H, W), seed=1))
        xg = x.clone().requires_grad_(True)
        hr = torch.from_numpy(S.synth_image(name + "/hr", (B, cfg.in_chans, H * cfg.upscale, W * cfg.upscale), seed=2))
        loss = torch.nn.L1Loss(reduction="mean")(m(xg), hr)
        loss.backward()
        gn = {k: p.grad for k, p in m.named_parameters()}
        gr[name + "/hr"] = hr.numpy()
        gr[name + "/loss"] = np.array(loss.item(), dtype=np.float64)
        gr[name + "/grad_x"] = xg.grad.numpy()
        for k in ["conv_first.weight", "layers.0.swin2.attn.relative_position_bias_table", "layers.0.swin4.attn.relative_position_bias_table",
                  "layers.0.swin1.attn.qkv.weight", "layers.0.swin3.attn.proj.weight", "layers.0.swin5.mlp.fc2.bias",
                  "layers.0.adjust1.weight", "layers.0.swin1.norm1.weight", "conv_last.weight"]:
            gr[f"{name}/grad/{k}"] = gn[k].numpy()
        gr[name + "/grad_names"] = np.array(list(gn.keys()))
        gr[name + "/grad_l2"] = np.array([float(g.double().pow(2).sum().sqrt()) for g in gn.values()])
        print(name, "loss %.5f" % loss.item(), "grad_x absmax %.3e" % xg.grad.abs().max())
    np.savez_compressed(os.path.join(HERE, "sr_grad_golden.npz"), **gr)
    if "--grad-only" in sys.argv:
        return

    # ---------------------------------------------------------------- scorer goldens (G9, G10)
    sc = {}
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import scorer_ref as O
    for tag, ch, size in [("gray", 1, 48), ("rgb", 3, 40)]:
        y_true, sr, hr = O.synth_pairs(3, 3, size, ch, seed=5)
        sc[f"{tag}/sr"] = np.stack(sr)
        sc[f"{tag}/hr"] = np.stack(hr)
        sc[f"{tag}/y"] = np.array(y_true)
        wss = [3, 11, 13, 23, size - 5 if (size - 5) % 2 else size - 4]
        sc[f"{tag}/ws"] = np.array(wss)
        vals = np.zeros((len(sr), len(wss)))
        for i, (s, h) in enumerate(zip(sr, hr)):
            for j, ws in enumerate(wss):
                vals[i, j] = RM.ssim_numpy(h.astype(np.float32) / 255.0, s.astype(np.float32) / 255.0, ws)
        sc[f"{tag}/ssim"] = vals
        sc[f"{tag}/psnr"] = np.array([RM.psnr_numpy(h.astype(np.float32) / 255.0, s.astype(np.float32) / 255.0)
                                      for s, h in zip(sr, hr)])
        sc[f"{tag}/ssim_u8"] = np.array([RM.ssim_numpy(h, s, 7) for s, h in zip(sr, hr)])   # integer-input branch
        # validation metrics (torch path incl. the H3 constants) on float tensors in [0,255]
        st = torch.from_numpy(np.stack(sr)).permute(0, 3, 1, 2).float() + 0.37
        ht = torch.from_numpy(np.stack(hr)).permute(0, 3, 1, 2).float()
        sc[f"{tag}/val_sr"] = st.numpy()
        sc[f"{tag}/val_quant"] = quantize(st * 1.003 - 0.2, 255).numpy()
        sc[f"{tag}/val_psnr"] = np.array([RM.psnr_torch(st[i:i + 1], ht[i:i + 1], 255) for i in range(len(sr))])
        sc[f"{tag}/val_ssim"] = np.array([RM.ssim_torch(st[i:i + 1], ht[i:i + 1], 255) for i in range(len(sr))])
    rng = np.random.RandomState(3)
    auc_cases = {
        "random": (rng.randint(0, 2, 40), rng.rand(40)),
        "ties": (np.array([0, 0, 1, 1, 0, 1, 1, 0, 1, 0]), np.array([.1, .2, .2, .3, .3, .3, .9, .1, .2, .9])),
        "all_equal": (np.array([0, 1, 0, 1, 1]), np.zeros(5)),
        "separated": (np.array([0, 0, 0, 1, 1]), np.array([.1, .2, .3, .7, .9])),
        "inverted": (np.array([1, 1, 0, 0]), np.array([.1, .2, .3, .4])),
        "grid_like": (np.r_[np.zeros(21, int), np.ones(57, int)], np.r_[rng.rand(21) * .6, rng.rand(57) * .8 + .2]),
    }
    for k, (y, s) in auc_cases.items():
        sc[f"auc/{k}/y"] = y
        sc[f"auc/{k}/s"] = s
        sc[f"auc/{k}/auc"] = np.array(roc_auc_score(y, s))
    np.savez_compressed(os.path.join(HERE, "scorer_golden.npz"), **sc)
    for f in ("sr_golden.npz", "scorer_golden.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
