#!/usr/bin/env python3
"""Generate tests/golden/legacy_golden.npz by running THE REFERENCE's dataset-preparation script
(scripts/prepare_mvtec_data.py) and its legacy folder scorers / threshold finders (src/helpers.py) on small synthetic PNG
trees.  Data only: the input images, file names, listing orders and the reference's outputs.  Build container only:

    python tests/golden/make_legacy_golden.py

Import shim as in make_golden.py (empty stand-ins for the absent skimage / imageio / torchvision).  scikit-image's
``structural_similarity`` / ``peak_signal_noise_ratio`` are placeholders that are not callable, so ``calculate_ssim`` /
``calculate_psnr`` take the reference's own fall-back branch (its "unified implementation", src/helpers.py:107-134) - the
branch this build mirrors."""
import importlib.util
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SRAD_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)


def _stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    for n in ["skimage", "skimage.color", "skimage.exposure", "imageio", "imageio.v2", "torchvision", "torchvision.transforms",
              "torchvision.datasets"]:
        stub(n)
    stub("skimage.metrics", structural_similarity=None, peak_signal_noise_ratio=None)
    for parent, child in [("skimage", "color"), ("skimage", "metrics"), ("skimage", "exposure"), ("imageio", "v2"),
                          ("torchvision", "transforms"), ("torchvision", "datasets")]:
        setattr(sys.modules[parent], child, sys.modules[f"{parent}.{child}"])
    for n in ("matplotlib", "matplotlib.pyplot"):
        try:
            __import__(n)
        except Exception:
            stub(n)
    if "matplotlib" in sys.modules and "matplotlib.pyplot" in sys.modules:
        setattr(sys.modules["matplotlib"], "pyplot", sys.modules["matplotlib.pyplot"])


def texture(rng, h, w, c):
    yy, xx = np.mgrid[0:h, 0:w]
    base = 120 + 70 * np.sin(xx / 3.1) * np.cos(yy / 4.3)
    img = base[:, :, None] + 18 * rng.randn(h, w, max(c, 1))
    img = np.clip(img, 0, 255).astype(np.uint8)
    return img[:, :, 0] if c == 0 else img


def main():
    _stubs()
    sys.path.insert(0, REF)
    out = {}
    rng = np.random.RandomState(11)
    tmp = Path(tempfile.mkdtemp(prefix="legacy_golden_"))

    # ------------------------------------------------------------------ A. scripts/prepare_mvtec_data.py
    spec = importlib.util.spec_from_file_location("ref_prepare", os.path.join(REF, "scripts", "prepare_mvtec_data.py"))
    P = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(P)
    src = tmp / "mvtec"
    names_in = []
    for cls, gray in (("carpet", False), ("grid", True)):
        tree = {"train/good": 6, "test/good": 2, "test/crack": 2, "test/bent": 1}
        for sub, n in tree.items():
            d = src / cls / sub
            d.mkdir(parents=True)
            for i in range(n):
                img = texture(rng, 50, 44, 0 if gray else 3)
                name = f"{i:03d}.png"
                Image.fromarray(img).save(d / name)
                key = f"{cls}/{sub}/{name}"
                names_in.append(key)
                out["prep/in/" + key] = img
    dst = tmp / "mvtec_32"
    P.prepare_mvtec_dataset(str(src), str(dst), scale_factors=(2, 4), target_hr=(32, 32), val_ratio=0.34, seed=42)
    files = sorted(str(p.relative_to(dst)) for p in dst.rglob("*.png"))
    out["prep/in_names"] = np.array(names_in)
    out["prep/out_names"] = np.array(files)
    for f in files:
        out["prep/out/" + f] = np.array(Image.open(dst / f))
    # the split as a function of the listing order the reference saw (a glob's order is the file system's)
    for cls in ("carpet", "grid"):
        listing = [p.name for p in (src / cls / "train" / "good").glob("*.png")]
        out[f"prep/listing/{cls}"] = np.array(listing)
        out[f"prep/val/{cls}"] = np.array(sorted(p.name for p in (dst / cls / "val" / "good" / "HR").glob("*.png")))
    # the split rule on explicit lists
    for tag, n, ratio, seed in (("a", 10, 0.1, 42), ("b", 7, 0.34, 3), ("c", 1, 0.5, 42), ("d", 5, 0.0, 42), ("e", 2, 0.01, 7)):
        lst = [f"{i:02d}.png" for i in range(n)]
        r = np.random.RandomState(seed)
        sh = list(lst)
        r.shuffle(sh)
        v = int(len(sh) * float(ratio))
        v = max(1, v) if len(sh) > 1 and ratio > 0 else 0
        out[f"prep/split/{tag}/args"] = np.array([n, ratio, seed], dtype=np.float64)
        out[f"prep/split/{tag}/val"] = np.array(sh[:v])
        out[f"prep/split/{tag}/train"] = np.array(sh[v:])

    # ------------------------------------------------------------------ B. src/helpers.py folder scorers + threshold finders
    from src import helpers as RH
    fold = tmp / "folders"
    sets = {"good": 4, "bad": 5}
    for label, n in sets.items():
        (fold / f"{label}_orig").mkdir(parents=True)
        (fold / f"{label}_rec").mkdir(parents=True)
        for i in range(n):
            o = texture(rng, 40, 36, 3)
            r = np.clip(o.astype(np.int32) + rng.randint(-7, 8, o.shape), 0, 255)
            if label == "bad":
                y0, x0 = rng.randint(4, 24), rng.randint(4, 20)
                r[y0:y0 + 9, x0:x0 + 9] += rng.randint(25, 60)
                r = np.clip(r, 0, 255)
            name = f"{label}{i}.png"
            Image.fromarray(o).save(fold / f"{label}_orig" / name)
            Image.fromarray(r.astype(np.uint8)).save(fold / f"{label}_rec" / name)
            out[f"legacy/{label}_orig/{name}"] = o
            out[f"legacy/{label}_rec/{name}"] = r.astype(np.uint8)
    args = [str(fold / "good_orig"), str(fold / "good_rec"), str(fold / "bad_orig"), str(fold / "bad_rec")]
    for label in sets:
        out[f"legacy/listing/{label}"] = np.array(os.listdir(fold / f"{label}_orig"))
    aw = RH.analyze_window_sizes(*args, min_size=3, max_size=None, step=10)
    for k, v in aw.items():
        out["legacy/analyze/" + k] = np.asarray(v, dtype=np.float64)
    y, s1, s2, s3 = RH.process_images(*args, str(tmp / "log.txt"), 11)
    out["legacy/process/y"] = np.array(y)
    out["legacy/process/ssim"] = np.array(s1, dtype=np.float64)
    out["legacy/process/mse"] = np.array(s2, dtype=np.float64)
    out["legacy/process/psnr"] = np.array(s3, dtype=np.float64)
    one = (out["legacy/bad_orig/bad0.png"], out["legacy/bad_rec/bad0.png"])
    out["legacy/single"] = np.array([RH.calculate_ssim(one[0], one[1], 7), RH.calculate_mse(*one), RH.calculate_psnr(*one),
                                     RH.calculate_ssim(one[0][:, :, 0], one[1][:, :, 0], 5)], dtype=np.float64)
    th = {}
    cases = {"proc_ssim": (y, s1), "proc_mse": (y, s2),
             "ties": ([0, 0, 1, 1, 0, 1, 1, 0, 1, 0], [.1, .2, .2, .3, .3, .3, .9, .1, .2, .9]),
             "random": (rng.randint(0, 2, 40).tolist(), np.round(rng.rand(40), 2).tolist())}
    for k, (yy, ss) in cases.items():
        out[f"legacy/thr/{k}/y"] = np.array(yy)
        out[f"legacy/thr/{k}/s"] = np.array(ss, dtype=np.float64)
        out[f"legacy/thr/{k}/out"] = np.array([RH.find_optimal_threshold_YoudenJ(yy, ss), RH.find_optimal_threshold(yy, ss),
                                               RH.find_threshold_for_perfect_recall(yy, ss)], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "legacy_golden.npz"), **out)
    print("wrote legacy_golden.npz:", len(out), "arrays;", "analyze windows", aw["window_sizes"], "best", aw["best_window_size"], aw["best_auc_window_size"])


if __name__ == "__main__":
    main()
