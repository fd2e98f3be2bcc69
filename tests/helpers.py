"""Shared test helpers: rebuild configs / synthetic state from a golden fixture entry."""
import numpy as np

from srad_amd import spec as S

DRN_GAIN = 0.5   # must match tests/golden/make_golden.py


def drct_case(g, name):
    in_chans, img_size, ws, upscale, n_rdg, seed = [int(v) for v in g[name + "/cfg"]]
    cfg = S.DRCTConfig(in_chans=in_chans, img_size=img_size, window_size=ws, upscale=upscale, n_rdg=n_rdg)
    sd = S.synth_state(S.drct_spec(cfg), seed=seed, gain=1.0, cfg=cfg)
    return cfg, sd, g[name + "/x"], g[name + "/y"]


def drn_case(g, name):
    n_colors, scale, seed = [int(v) for v in g[name + "/cfg"]]
    cfg = S.DRNConfig.for_scale(scale, n_colors)
    sd = S.synth_state(S.drn_spec(cfg), seed=seed, gain=DRN_GAIN, cfg=cfg)
    dual = S.synth_state(S.dual_spec(cfg), seed=seed + 100, gain=DRN_GAIN, cfg=cfg)
    ys = [g[f"{name}/y{j}"] for j in range(cfg.phase + 1)]
    return cfg, sd, dual, g[name + "/x"], ys, g[name + "/dual"]


DRCT_CASES = ["drct_full_gray_x4", "drct_r2_rgb_x4", "drct_r2_gray_x4_dyn64", "drct_r1_gray_x4_ws4",
              "drct_r1_gray_x8_ws2", "drct_r1_gray_x4_ws16"]
DRN_CASES = ["drn_x2_gray", "drn_x4_rgb", "drn_x4_gray", "drn_x8_gray"]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
