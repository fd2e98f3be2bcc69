"""Legacy folder scoring and threshold finders - host-side mirror of reference ``src/helpers.py`` (107-134, 158-230, 321-370,
453-481): the functions its older scripts and notebooks call on folders of original / reconstructed images.

    calculate_ssim / calculate_mse / calculate_psnr      per image pair
    analyze_window_sizes(...)                             SSIM over a window-size range for four folders -> the dict of the reference
    process_images(...)                                   (y_true, 1 - SSIM, MSE, -PSNR) of four folders at one window size
    find_optimal_threshold_YoudenJ / find_optimal_threshold / find_threshold_for_perfect_recall

Which SSIM: the reference calls scikit-image's ``structural_similarity`` and falls back to its own ``metrics.ssim_numpy`` /
``psnr_numpy`` ("a unified implementation to avoid dependency variance", src/helpers.py:107-134) when that raises.  scikit-image
is not part of this image, so the fallback IS the behaviour here: uniform ws x ws window, reflect padding, luminance of RGB,
``data_range`` 255 for the uint8 images the folders hold - the arithmetic of the engine's scorer (``srad_score_pairs``), which
evaluates every window size of a folder in one launch sequence instead of a Python loop per pixel.  SSIM is invariant to the common
scale, so the engine's [0, 1]-ranged evaluation equals the reference's [0, 255]-ranged one up to float32 rounding (pinned at 2e-6
by tests/test_host_golden.py against the imported reference).  The scikit-image branch is parity-unpinned (stated in DESIGN.md).

The MSE of this module is on the raw 0..255 values (src/helpers.py:124-127), not on [0, 1] like the evaluator's."""
from __future__ import annotations

import logging
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from PIL import Image

from . import metrics as M


def setup_logger(log_file_path):
    """src/helpers.py:102-105."""
    logging.basicConfig(filename=log_file_path, level=logging.INFO, format='%(asctime)s - %(message)s', datefmt='%Y-%m-%d %H:%M:%S')


def _load_rgb(path) -> np.ndarray:
    img = Image.open(path)
    return np.array(img if img.mode == 'RGB' else img.convert('RGB'))


def _stack(images: Sequence[np.ndarray]) -> torch.Tensor:
    a = np.stack([im if im.ndim == 3 else im[:, :, None] for im in images])
    if a.dtype != np.uint8:
        raise ValueError("the folder scorers take uint8 images (what PIL loads)")
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _score_stack(originals: Sequence[np.ndarray], reconstructed: Sequence[np.ndarray], sizes: Sequence[int]):
    """SSIM [n, len(sizes)], MSE on 0..255 values [n], PSNR [n] of same-shape uint8 pairs, one scorer call."""
    ssim, mse, psnr = M.score_pairs(_stack(reconstructed), _stack(originals), list(sizes))
    return ssim.cpu().numpy(), mse.cpu().numpy() * (255.0 * 255.0), psnr.cpu().numpy()


def calculate_ssim(original, reconstructed, win_size):
    """src/helpers.py:107-122 (the unified-implementation branch)."""
    original, reconstructed = np.asarray(original), np.asarray(reconstructed)
    if original.ndim != reconstructed.ndim or original.ndim not in (2, 3):
        raise ValueError("Input images must have the same dimensions (both 2D or both 3D)")
    return float(_score_stack([original], [reconstructed], [win_size])[0][0, 0])


def calculate_mse(original, reconstructed):
    """src/helpers.py:124-127: mean squared difference of the raw values, float32."""
    o, r = np.asarray(original, dtype=np.float32), np.asarray(reconstructed, dtype=np.float32)
    return float(np.mean((o - r) ** 2))


def calculate_psnr(original, reconstructed):
    """src/helpers.py:129-134 (data range 255 for integer images)."""
    return float(_score_stack([np.asarray(original)], [np.asarray(reconstructed)], [])[2][0])


def _folder_pairs(folder_original, folder_reconstructed) -> Tuple[List[str], List[np.ndarray], List[np.ndarray]]:
    names = os.listdir(folder_original)                             # the listing order is the order of every returned list
    return (names, [_load_rgb(os.path.join(folder_original, n)) for n in names],
            [_load_rgb(os.path.join(folder_reconstructed, n)) for n in names])


def _sweep_sizes(shape, min_size, max_size, step) -> Tuple[List[int], int]:
    min_dim = min(shape[0], shape[1])
    top = min(max_size, min_dim - 3) if max_size else min_dim - 3
    top = top if top % 2 != 0 else top - 1
    return list(range(min_size, top + 1, step)), top


def analyze_window_sizes(good_original_folder, good_reconstructed_folder, bad_original_folder, bad_reconstructed_folder,
                         min_size=3, max_size=None, step=10) -> Dict:
    """src/helpers.py:158-230: SSIM of every image at every window size, folder means, their difference and the AUC of 1 - SSIM per
    window size; the best window by difference and by AUC (first maximum)."""
    def folder(fo, fr):
        _, orig, rec = _folder_pairs(fo, fr)
        sizes, top = _sweep_sizes(orig[-1].shape, min_size, max_size, step)        # the reference keeps the LAST image's bound
        return _score_stack(orig, rec, sizes)[0], top
    good, good_top = folder(good_original_folder, good_reconstructed_folder)
    bad, bad_top = folder(bad_original_folder, bad_reconstructed_folder)
    window_sizes = list(range(min_size, min(good_top, bad_top) + 1, step))
    good, bad = good[:, :len(window_sizes)], bad[:, :len(window_sizes)]
    avg_good, avg_bad = np.mean(good, axis=0), np.mean(bad, axis=0)
    diff = avg_good - avg_bad
    y_true = [0] * len(good) + [1] * len(bad)
    aucs = [M.roc_auc(y_true, np.concatenate([1 - good[:, i], 1 - bad[:, i]])) for i in range(len(window_sizes))]
    return {'window_sizes': window_sizes, 'avg_good_scores': avg_good.tolist(), 'avg_bad_scores': avg_bad.tolist(),
            'score_differences': diff.tolist(), 'best_window_size': window_sizes[int(np.argmax(diff))], 'max_difference': np.max(diff),
            'auc_scores': aucs, 'best_auc_window_size': window_sizes[int(np.argmax(aucs))], 'max_auc': np.max(aucs)}


def process_images(good_original_folder, good_reconstructed_folder, bad_original_folder, bad_reconstructed_folder, log_file_path, window_size):
    """src/helpers.py:321-370: (y_true, 1 - SSIM, MSE, -PSNR) over the good (label 0) then the bad (label 1) folder, one log line
    per image."""
    setup_logger(log_file_path)
    y_true, s_ssim, s_mse, s_psnr = [], [], [], []
    for label, (fo, fr) in enumerate(((good_original_folder, good_reconstructed_folder), (bad_original_folder, bad_reconstructed_folder))):
        names, orig, rec = _folder_pairs(fo, fr)
        ssim, mse, psnr = _score_stack(orig, rec, [window_size])
        for i, name in enumerate(names):
            y_true.append(label)
            s_ssim.append(1 - float(ssim[i, 0]))
            s_mse.append(float(mse[i]))
            s_psnr.append(-float(psnr[i]))
            logging.info(f"Image: {name}, Label: {'Anomalous' if label else 'Normal'}, SSIM (window size {window_size}): "
                         f"{float(ssim[i, 0]):.4f}, MSE: {float(mse[i]):.4f}, PSNR: {float(psnr[i]):.4f}")
    return y_true, s_ssim, s_mse, s_psnr


def roc_curve(y_true, y_scores, drop_intermediate: bool = True):
    """``sklearn.metrics.roc_curve`` for binary labels (the call sites are src/helpers.py:454, 461): thresholds in decreasing order
    starting at +inf, one point per distinct score, collinear points dropped."""
    y = np.asarray(y_true).astype(bool)
    s = np.asarray(y_scores, dtype=np.float64)
    order = np.argsort(s, kind="mergesort")[::-1]
    y, s = y[order], s[order]
    distinct = np.r_[np.where(np.diff(s))[0], y.size - 1]
    tps = np.cumsum(y)[distinct].astype(np.float64)
    fps = (1 + distinct - tps).astype(np.float64)
    thr = s[distinct]
    if drop_intermediate and len(fps) > 2:
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    tps, fps, thr = np.r_[0, tps], np.r_[0, fps], np.r_[np.inf, thr]
    return fps / fps[-1], tps / tps[-1], thr


def find_optimal_threshold_YoudenJ(y_true, y_scores):
    """src/helpers.py:453-458: the threshold maximising tpr - fpr."""
    fpr, tpr, thr = roc_curve(y_true, y_scores)
    return thr[int(np.argmax(tpr - fpr))]


def find_optimal_threshold(y_true, y_scores):
    """src/helpers.py:460-469: the ROC point closest to (0, 1)."""
    fpr, tpr, thr = roc_curve(y_true, y_scores)
    return thr[int(np.argmin(np.sqrt(fpr ** 2 + (1 - tpr) ** 2)))]


def find_threshold_for_perfect_recall(y_true, y_scores):
    """src/helpers.py:471-481: the lowest score of a positive sample."""
    y, s = np.asarray(y_true), np.asarray(y_scores)
    return s[y == 1].min()
