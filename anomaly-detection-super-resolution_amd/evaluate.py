"""Standalone evaluator: SR every test image -> truncating u8 -> SSIM window sweep -> AUCs, the
flow of reference src/evaluate.py:138-267 with the SR forward and the whole scorer on the GPU.
Images of the test split are independent, so with ``--gpus N`` (one process per GPU under
``torch.distributed.run``) rank r scores images r::N and only the per-image score rows are gathered
(SURVEY.md §8(e)); there is no collective on the data path.

    python -m srad_amd.evaluate --run-dir <run> [--checkpoint f.pt] [--dtype bf16]

Deliberate, documented departures from the reference (SURVEY.md §8 hazards): the model is put in
eval mode (H1: the reference leaves DRCT's DropPath active, so its scores are not reproducible); the
u8 conversion truncates exactly as the reference's evaluator does (H2).
"""
from __future__ import annotations

import os
import re
from pathlib import Path
from typing import Iterable, List, Sequence, Tuple

import numpy as np
import torch

from . import metrics as M
from .model import Model
from .options import build_opt, parse_eval_args


def infer_from_run_dir(run_dir: str) -> dict:
    """src/evaluate.py:48-122: directory-name pattern first, then config.txt overrides."""
    result = {'model_type': None, 'dataset': None, 'classe': None, 'resolution': None, 'scale': None}
    for seg in Path(run_dir).parts:
        if seg in ('drct', 'drn-l'):
            result['model_type'] = seg
            break
    m = re.match(r"(?P<ds>\w+)_(?P<cls>\w+)_(?P<res>\d+)_X(?P<scale>\d+)", Path(run_dir).name)
    if m:
        result.update(dataset=m.group('ds'), classe=m.group('cls'), resolution=int(m.group('res')), scale=int(m.group('scale')))
    cfg_path = Path(run_dir) / 'config.txt'
    if cfg_path.exists():
        lines = cfg_path.read_text().splitlines()

        def read_val(key):
            for line in lines:
                if line.strip().startswith(f"{key}:"):
                    return line.split(':', 1)[1].strip()
            return None
        for key, dst in (('model_name', 'model_type'), ('dataset', 'dataset'), ('classe', 'classe')):
            v = read_val(key)
            if v:
                result[dst] = v
        res = read_val('patch_size')
        if res and res.isdigit():
            result['resolution'] = int(res)
        scale_val = read_val('upscale') or read_val('scale')
        if scale_val:
            ms = re.findall(r"\d+", scale_val)
            if ms:
                result['scale'] = int(ms[-1])
    return result


def resolve_checkpoint(args) -> str:
    """src/evaluate.py:125-135"""
    if args.checkpoint:
        return args.checkpoint
    if args.run_dir:
        for name in ('model_best.pt', 'model_latest.pt'):
            cand = os.path.join(args.run_dir, 'model', name)
            if os.path.isfile(cand):
                return cand
    raise FileNotFoundError('Please provide --checkpoint or a valid --run-dir containing model/*.pt')


def iter_split(data_root: str, classe: str, split: str, scale, n_colors: int, rgb_range: float = 255.0
               ) -> Iterable[Tuple[str, np.ndarray, np.ndarray]]:
    """(name, LR u8 HWC, HR u8 HWC) of ``{root}/{class}/test/{split}`` read the way the reference's evaluator reads it
    (src/evaluate.py:140-150,204-217): the MVTec test loader (``data.MVTec(train=False)``: LR_{s} / LR_bicubic/X{s} / LR
    next to HR, ``set_channel``, HR cropped to LR * scale) and the TRUNCATING u8 conversion of the HR tensor.  The LR image
    is what the loader feeds the model, kept as u8 when it is integral (PNG input), so the forward sees the same values."""
    from .data import MVTec

    class _O:
        pass
    o = _O()
    o.scale = list(scale) if isinstance(scale, (list, tuple)) else [scale]
    o.data_dir = str(Path(data_root) / classe / 'test' / split)
    o.n_colors, o.rgb_range, o.no_augment, o.patch_size, o.batch_size, o.test_every = n_colors, 255, True, 0, 1, 1
    ds = MVTec(o, train=False)
    for i in range(len(ds)):
        lr, hr, name = ds.sample(i)
        hr_u8 = hr.clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()          # .byte(): truncation
        lr0 = lr[0].permute(1, 2, 0).numpy()
        yield name, (lr0.astype(np.uint8) if np.array_equal(lr0, np.floor(lr0)) and lr0.min() >= 0 and lr0.max() <= 255 else lr0), hr_u8


def save_sr_image(sr_u8_hwc: np.ndarray, name: str, split: str, scale_value: int, output_dir: str) -> None:
    """src/evaluate.py:193-202: ``<output_dir>/<split>/x<scale>/<name>.png`` of the truncated u8 SR image."""
    from PIL import Image
    out_dir = Path(output_dir) / split / f"x{scale_value}"
    out_dir.mkdir(parents=True, exist_ok=True)
    img = Image.fromarray(sr_u8_hwc[:, :, 0]) if sr_u8_hwc.shape[2] == 1 else Image.fromarray(np.ascontiguousarray(sr_u8_hwc))
    img.save(str(out_dir / f"{name}.png"))


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """Images rank ``rank`` of ``world`` scores: r, r + world, ... (image-parallel, no data-path collective)."""
    return list(range(rank, n, world))


def gather_score_rows(mine: Sequence[int], rows: np.ndarray, total: int, rank: int, world: int):
    """Assemble the [total, n_ws + 2] score table on rank 0 from every rank's rows (the only
    cross-rank exchange of the evaluator: a few floats per image).  Other ranks get None."""
    if world == 1:
        return rows
    import torch.distributed as dist
    gathered = [None] * world
    dist.all_gather_object(gathered, (list(mine), rows))
    if rank != 0:
        return None
    full = np.zeros((total, rows.shape[1]), dtype=np.float64)
    seen = np.zeros(total, dtype=bool)
    for idx, r in gathered:
        full[idx] = r
        seen[idx] = True
    if not seen.all():
        raise RuntimeError("score rows missing after the gather")
    return full


@torch.no_grad()
def super_resolve_u8(model, lr_u8: Sequence[np.ndarray], hr_u8: Sequence[np.ndarray], rgb_range: float, batch: int = 0
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """collect_pairs (src/evaluate.py:204-224): forward, crop to the HR size, truncate to u8.
    Returns (sr, hr) uint8 stacks [n,H,W,C] on the GPU.  ``batch`` images per forward; 0 = as many as make ~64 k LR pixels
    (64 images at 128 px / x4, one 1024 px tile): the reference forwards one image at a time, the outputs per image are the same
    and the chip is only filled from a few thousand tokens up (78 pairs at 128 px: 1830 -> 2120 images/s in split-bf16 mode)."""
    dev = model.device if hasattr(model, 'device') else next(model.parameters()).device
    if batch <= 0 and len(lr_u8):
        batch = max(1, min(64, 65536 // max(1, lr_u8[0].shape[0] * lr_u8[0].shape[1])))
    sr_out: List[torch.Tensor] = []
    for i in range(0, len(lr_u8), batch):
        lr = torch.from_numpy(np.stack(lr_u8[i:i + batch])).to(dev).permute(0, 3, 1, 2).float() * (rgb_range / 255.0)   # np2Tensor
        sr = model(lr)
        if isinstance(sr, list):
            sr = sr[-1]
        h, w = hr_u8[i].shape[:2]
        sr_out.append(M.to_u8_hwc(sr[..., :h, :w].contiguous(), rgb_range))
    hr = torch.from_numpy(np.stack(hr_u8)).to(dev)
    return torch.cat(sr_out), hr


def evaluate_on_test(opt, model, good: Sequence[Tuple[np.ndarray, np.ndarray]], bad: Sequence[Tuple[np.ndarray, np.ndarray]],
                     rank: int = 0, world: int = 1, names: Sequence[str] = (), output_dir: str = '', save_images: bool = False) -> dict:
    """src/evaluate.py:138-267 for in-memory (LR, HR) u8 pairs.  With world > 1 every rank scores its
    share r::world; rank 0 gathers the score rows and returns the AUCs (others return {}).  ``save_images``: every rank
    writes the SR images it produced under ``output_dir/{good,bad}/x{scale}`` (src/evaluate.py:190-224)."""
    model.eval()                                              # H1: deterministic scoring
    y_true = [0] * len(good) + [1] * len(bad)
    pairs = list(good) + list(bad)
    if len(set(y_true)) < 2:
        print('Test set lacks both classes; AUC not available')
        return {}
    mine = shard_indices(len(pairs), rank, world)
    sr, hr = super_resolve_u8(model, [pairs[i][0] for i in mine], [pairs[i][1] for i in mine], float(opt.rgb_range))
    if save_images and output_dir:
        scale_value = opt.scale[-1] if isinstance(opt.scale, list) else int(opt.scale)
        sr_host = sr.cpu().numpy()
        for k, i in enumerate(mine):
            save_sr_image(sr_host[k], names[i] if i < len(names) else f"{i:05d}", 'good' if y_true[i] == 0 else 'bad', scale_value, output_dir)
    H, W = hr.shape[1:3]
    sizes = M.sweep_window_sizes(min(H, W))
    ssim, mse, psnr = M.score_pairs(sr, hr, sizes)
    rows = torch.cat([ssim, mse[:, None], psnr[:, None]], dim=1)          # [n_mine, n_ws + 2] float64
    full = gather_score_rows(mine, rows.cpu().numpy(), len(pairs), rank, world)
    if full is None:
        return {}
    best_ws, best_auc, best_j = sizes[0], -1.0, 0
    for j, ws in enumerate(sizes):
        a = M.roc_auc(y_true, 1.0 - full[:, j])
        if a > best_auc:
            best_auc, best_ws, best_j = a, ws, j
    out = dict(best_ws=best_ws, auc_ssim=M.roc_auc(y_true, 1.0 - full[:, best_j]), auc_mse=M.roc_auc(y_true, full[:, -2]),
               auc_psnr=M.roc_auc(y_true, -full[:, -1]), n_images=len(pairs), window_sizes=sizes)
    print(f"Test AUCs - SSIM(best ws={best_ws}): {out['auc_ssim']:.4f}, MSE: {out['auc_mse']:.4f}, PSNR: {out['auc_psnr']:.4f}")
    return out


def main(argv=None):
    args = parse_eval_args(argv)
    from .launch import spawn
    if spawn(_run, getattr(args, "gpus", 1), (args,)):      # --gpus N without a launcher: N fresh ranks (image-parallel)
        return
    _run(args)


def _run(args):
    model_type, class_name, resolution, scale = args.model_type, args.classe, args.resolution, args.scale
    if args.run_dir:
        inf = infer_from_run_dir(args.run_dir)
        model_type = inf.get('model_type') or model_type
        class_name = inf.get('classe') or class_name
        resolution = inf.get('resolution') or resolution
        scale = inf.get('scale') or scale
    if args.device == 'cpu':
        raise SystemExit("--device cpu is the reference's own path; this build has no CPU fallback")
    ckpt = resolve_checkpoint(args)
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    opt = build_opt(model_type, class_name, resolution, scale, args.batch_size, args.dtype, pre_train=ckpt,
                    data_root=args.data_root)
    opt.test_only = True
    model = Model(opt, None, dual_model=(model_type == 'drn-l'))
    g = list(iter_split(opt.data_root, class_name, 'good', opt.scale, opt.n_colors))
    b = list(iter_split(opt.data_root, class_name, 'bad', opt.scale, opt.n_colors))
    out_dir = args.output_dir or (os.path.join(args.run_dir, 'eval_results') if args.run_dir else './workspace/eval_results')
    evaluate_on_test(opt, model, [(lr, hr) for _, lr, hr in g], [(lr, hr) for _, lr, hr in b], rank, world,
                     names=[n for n, _, _ in g] + [n for n, _, _ in b], output_dir=out_dir, save_images=args.save_images)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
