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


def _load_png(path: str, n_colors: int) -> np.ndarray:
    from PIL import Image
    im = Image.open(path)
    im = im.convert('L') if n_colors == 1 else im.convert('RGB')
    a = np.asarray(im, dtype=np.uint8)
    return a[:, :, None] if a.ndim == 2 else a


def iter_split(data_root: str, classe: str, split: str, scale: int, n_colors: int) -> Iterable[Tuple[str, np.ndarray, np.ndarray]]:
    """(name, LR u8 HWC, HR u8 HWC) of ``{root}/{class}/test/{split}/{HR,LR_{s}}/*.png`` - the layout
    scripts/prepare_mvtec_data.py writes and src/data.py:109-147 reads (LR_{s}, LR_bicubic/X{s} or LR)."""
    base = Path(data_root) / classe / 'test' / split
    hr_dir = base / 'HR'
    lr_dir = next((d for d in (base / f'LR_{scale}', base / 'LR_bicubic' / f'X{scale}', base / 'LR') if d.is_dir()), None)
    if not hr_dir.is_dir() or lr_dir is None:
        raise FileNotFoundError(f"expected {hr_dir} and an LR folder next to it")
    for hp in sorted(hr_dir.glob('*.png')):
        lp = lr_dir / hp.name
        if not lp.is_file():
            raise FileNotFoundError(str(lp))
        yield hp.stem, _load_png(str(lp), n_colors), _load_png(str(hp), n_colors)


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
def super_resolve_u8(model, lr_u8: Sequence[np.ndarray], hr_u8: Sequence[np.ndarray], rgb_range: float, batch: int = 8
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """collect_pairs (src/evaluate.py:204-224): forward, crop to the HR size, truncate to u8.
    Returns (sr, hr) uint8 stacks [n,H,W,C] on the GPU."""
    dev = model.device if hasattr(model, 'device') else next(model.parameters()).device
    sr_out: List[torch.Tensor] = []
    for i in range(0, len(lr_u8), batch):
        lr = torch.from_numpy(np.stack(lr_u8[i:i + batch])).to(dev).permute(0, 3, 1, 2).float() * (rgb_range / 255.0)
        sr = model(lr)
        if isinstance(sr, list):
            sr = sr[-1]
        h, w = hr_u8[i].shape[:2]
        sr_out.append(M.to_u8_hwc(sr[..., :h, :w].contiguous(), rgb_range))
    hr = torch.from_numpy(np.stack(hr_u8)).to(dev)
    return torch.cat(sr_out), hr


def evaluate_on_test(opt, model, good: Sequence[Tuple[np.ndarray, np.ndarray]], bad: Sequence[Tuple[np.ndarray, np.ndarray]],
                     rank: int = 0, world: int = 1) -> dict:
    """src/evaluate.py:138-267 for in-memory (LR, HR) u8 pairs.  With world > 1 every rank scores its
    share r::world; rank 0 gathers the score rows and returns the AUCs (others return {})."""
    model.eval()                                              # H1: deterministic scoring
    y_true = [0] * len(good) + [1] * len(bad)
    pairs = list(good) + list(bad)
    if len(set(y_true)) < 2:
        print('Test set lacks both classes; AUC not available')
        return {}
    mine = shard_indices(len(pairs), rank, world)
    sr, hr = super_resolve_u8(model, [pairs[i][0] for i in mine], [pairs[i][1] for i in mine], float(opt.rgb_range))
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
    good = [(lr, hr) for _, lr, hr in iter_split(opt.data_root, class_name, 'good', scale, opt.n_colors)]
    bad = [(lr, hr) for _, lr, hr in iter_split(opt.data_root, class_name, 'bad', scale, opt.n_colors)]
    evaluate_on_test(opt, model, good, bad, rank, world)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
