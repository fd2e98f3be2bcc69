"""MVTec AD dataset preparation - host-side mirror of reference ``scripts/prepare_mvtec_data.py`` (22-33, 43-159, 161-205, 269-318):
the offline step that produces the on-disk layout ``data.py`` reads.

    python -m srad_amd.prepare_mvtec_data --hr-size 128 --scales 4 [--val-ratio 0.1] [--seed 42] [--source data/mvtec] [--target data/mvtec_128]

For every class (``carpet``, ``grid``):

    <target>/<class>/{train,val}/good/HR/<name>.png      train/good images, LANCZOS-resized to hr x hr (converted to RGB first)
    <target>/<class>/{train,val}/good/LR_<s>/<name>.png  the HR image LANCZOS-resized to hr // s, for every s of the scale set
    <target>/<class>/test/good/{HR,LR_<s>}/<name>.png
    <target>/<class>/test/bad/{HR,LR_<s>}/<defect>_<name>.png   every defect class merged into ``bad``, the defect name as prefix

The scale set is progressive (what DRN-L's intermediate outputs are trained against): always 2, plus 4 when 8 is asked for.
The train / val split shuffles the directory listing with ``numpy.random.RandomState(seed)`` and takes the first
``max(1, int(n * val_ratio))`` names as validation - the listing is used in the order the file system returns it, as in the
reference, so the same tree on the same file system gives the same split.  Everything here is PIL + numpy on the host: no
GPU work, nothing on the hot path."""
from __future__ import annotations

import argparse
import shutil
from pathlib import Path
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
from PIL import Image

CLASSES = ("carpet", "grid")


def resize_image(image_path, target_size=(128, 128), resample=Image.LANCZOS):
    """The image at ``image_path`` as RGB, resized to ``target_size`` (scripts/prepare_mvtec_data.py:22-28)."""
    with Image.open(image_path) as img:
        rgb = img if img.mode == 'RGB' else img.convert('RGB')
        return rgb.resize(target_size, resample)


def create_lr_image(hr_image, scale_factor=4, resample=Image.LANCZOS):
    """``hr_image`` resized to ``size // scale_factor`` per side (scripts/prepare_mvtec_data.py:30-33)."""
    w, h = hr_image.size
    return hr_image.resize((w // scale_factor, h // scale_factor), resample)


def split_train_val(files: Sequence, val_ratio: float = 0.1, seed: int = 42) -> Tuple[List, List]:
    """(train, val) of an ordered file list (scripts/prepare_mvtec_data.py:69-74): in-place RandomState shuffle, the first
    ``int(n * val_ratio)`` (at least one when there are two or more files and the ratio is positive) are validation."""
    files = list(files)
    np.random.RandomState(seed).shuffle(files)
    n_val = int(len(files) * float(val_ratio))
    n_val = max(1, n_val) if len(files) > 1 and val_ratio > 0 else 0
    return files[n_val:], files[:n_val]


def _write_pyramid(src: Path, split_dir: Path, name: str, scale_factors: Iterable[int], target_hr) -> None:
    """One source image -> <split_dir>/HR/<name> and <split_dir>/LR_<s>/<name> for every s."""
    hr = resize_image(src, target_size=target_hr)
    (split_dir / "HR").mkdir(parents=True, exist_ok=True)
    hr.save(split_dir / "HR" / name)
    for s in scale_factors:
        (split_dir / f"LR_{s}").mkdir(parents=True, exist_ok=True)
        create_lr_image(hr, scale_factor=s).save(split_dir / f"LR_{s}" / name)


def process_training_data(source_dir, train_target_dir, val_target_dir, scale_factors=(4,), target_hr=(128, 128), val_ratio=0.1, seed=42):
    """train/good of one class -> the train and val trees (scripts/prepare_mvtec_data.py:43-93)."""
    source_dir, train_target_dir, val_target_dir = Path(source_dir), Path(train_target_dir), Path(val_target_dir)
    print(f"Processing training data: {source_dir.name}")
    for base in (train_target_dir, val_target_dir):                 # both trees exist even when a split is empty
        for sub in ["HR"] + [f"LR_{s}" for s in scale_factors]:
            (base / "good" / sub).mkdir(parents=True, exist_ok=True)
    files = list(source_dir.glob("*.png"))
    print(f"  Found {len(files)} training images")
    if not files:
        print("  WARNING No training images found. Skipping train/val split.")
        return
    train, val = split_train_val(files, val_ratio, seed)
    for split, base in ((train, train_target_dir), (val, val_target_dir)):
        for f in split:
            _write_pyramid(f, base / "good", f.name, scale_factors, target_hr)
    print(f"  Created {len(train)} train pairs and {len(val)} val pairs")


def process_test_data(source_dir, target_dir, scale_factors=(4,), target_hr=(128, 128)):
    """test/<good | defect classes> of one class -> test/good and test/bad (scripts/prepare_mvtec_data.py:95-159)."""
    source_dir, target_dir = Path(source_dir), Path(target_dir)
    print(f"Processing test data: {source_dir.name}")
    for label in ("good", "bad"):
        for sub in ["HR"] + [f"LR_{s}" for s in scale_factors]:
            (target_dir / label / sub).mkdir(parents=True, exist_ok=True)
    if (source_dir / "good").exists():
        good = list((source_dir / "good").glob("*.png"))
        print(f"  Processing good: {len(good)} images")
        for f in good:
            _write_pyramid(f, target_dir / "good", f.name, scale_factors, target_hr)
    for defect in [d for d in source_dir.iterdir() if d.is_dir() and d.name != "good"]:
        images = list(defect.glob("*.png"))
        print(f"  Processing {defect.name}: {len(images)} images")
        for f in images:
            _write_pyramid(f, target_dir / "bad", f"{defect.name}_{f.name}", scale_factors, target_hr)   # unique across defect classes
    print(f"  Good test images: {len(list((target_dir / 'good' / 'HR').glob('*.png')))}")
    print(f"  Bad test images: {len(list((target_dir / 'bad' / 'HR').glob('*.png')))}")


def prepare_mvtec_dataset(source_base="data/mvtec", target_base="data/mvtec_128", scale_factors=(4,), target_hr=(128, 128), val_ratio=0.1, seed=42):
    """Both classes, train + val + test; an existing target tree is replaced (scripts/prepare_mvtec_data.py:161-205)."""
    source_base, target_base = Path(source_base), Path(target_base)
    print(f"Preparing MVTec AD dataset for {target_hr[0]}x{target_hr[1]} training")
    if target_base.exists():
        shutil.rmtree(target_base)
        print("Cleaned existing target directory")
    for cls in CLASSES:
        print(f"\nProcessing class: {cls}")
        train_src, test_src = source_base / cls / "train" / "good", source_base / cls / "test"
        if train_src.exists():
            process_training_data(train_src, target_base / cls / "train", target_base / cls / "val", scale_factors, target_hr=target_hr,
                                  val_ratio=val_ratio, seed=seed)
        else:
            print(f"  ERROR: Training data not found: {train_src}")
        if test_src.exists():
            process_test_data(test_src, target_base / cls / "test", scale_factors, target_hr=target_hr)
        else:
            print(f"  ERROR: Test data not found: {test_src}")
    print(f"\nDataset preparation complete!\nOutput directory: {target_base}")


def verify_dataset_structure(base_dir) -> Dict[str, Dict[str, int]]:
    """Image counts per folder of a prepared tree, printed and returned (scripts/prepare_mvtec_data.py:207-267)."""
    base, counts = Path(base_dir), {}
    print(f"\nVerifying dataset structure: {base_dir}")
    for cls in CLASSES:
        counts[cls] = {}
        print(f"\n  {cls}/")
        for split, label in (("train", "good"), ("val", "good"), ("test", "good"), ("test", "bad")):
            folder = base / cls / split / label
            for sub in [folder / "HR"] + sorted(folder.glob("LR_*")):
                key = f"{split}/{label}/{sub.name}"
                if sub.exists():
                    counts[cls][key] = len(list(sub.glob("*.png")))
                    print(f"    {key}: {counts[cls][key]} images")
                else:
                    print(f"    ERROR: {key}: missing")
    print("Dataset verification complete!")
    return counts


def progressive_scales(user_scales: Iterable[int]) -> Tuple[int, ...]:
    """The LR levels written for the requested final scales (scripts/prepare_mvtec_data.py:296-302): always 2, and 4 under 8."""
    scales = set(int(s) for s in user_scales)
    bad = [s for s in scales if s not in (4, 8)]
    if bad:
        raise ValueError("Only scales 4 and/or 8 are supported")
    scales.add(2)
    if 8 in scales:
        scales.add(4)
    return tuple(sorted(scales))


def main(argv=None):
    ap = argparse.ArgumentParser(description="MVTec AD dataset preparation")
    ap.add_argument("--hr-size", type=int, default=128, choices=[256, 128, 64, 32])
    ap.add_argument("--scales", type=str, default="4", help="Comma-separated downscale factors: 4,8")
    ap.add_argument("--val-ratio", type=float, default=0.1)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--source", type=str, default="data/mvtec", help="(this build) where the original MVTec AD tree is")
    ap.add_argument("--target", type=str, default="", help="(this build) output tree; default data/mvtec_<hr-size>")
    args = ap.parse_args(argv)
    print(f"MVTec AD Dataset Preparation ({args.hr_size}x{args.hr_size})")
    if not Path(args.source).exists():
        print(f"ERROR: Source data not found. Please ensure MVTec dataset is in {args.source}/")
        return 1
    try:
        scales = progressive_scales(int(s) for s in args.scales.split(',') if s.strip())
    except ValueError as e:
        print(f"ERROR: {e if 'supported' in str(e) else 'Invalid --scales. Use comma-separated integers from {4,8}'}")
        return 1
    target = args.target or f"data/mvtec_{args.hr_size}"
    prepare_mvtec_dataset(args.source, target, scales, (args.hr_size, args.hr_size), args.val_ratio, args.seed)
    verify_dataset_structure(target)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
