"""Option objects and argument parsing with the reference's names, defaults and quirks
(src/main.py:35-294; src/evaluate.py:20-45).  The choices are a superset: --scale 2 and
--resolution 512/1024 are accepted for the BASELINE configs, plus --dtype and --gpus."""
from __future__ import annotations

import argparse
import os
import sys
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import yaml


@dataclass
class DRN:
    model_name: str = 'drn-l'
    n_threads: int = -2
    cpu: bool = False
    n_GPUs: int = 1
    seed: int = 1
    data_dir: str = './workspace/gkd/DC2/unlabeled/HR_512_grayscale/'
    data_train: str = ''
    data_test: str = ''
    data_range: str = '1-224/225-280'
    scale: object = 4
    patch_size: int = 512
    rgb_range: int = 255
    n_colors: int = 1
    no_augment: bool = False
    pre_train: str = '.'
    pre_train_dual: str = '.'
    n_blocks: int = 40
    n_feats: int = 20
    negval: float = 0.2
    test_every: int = 10
    epochs: int = 10
    batch_size: int = 4
    self_ensemble: bool = False
    test_only: bool = False
    lr: float = 1e-4
    eta_min: float = 1e-7
    beta1: float = 0.9
    beta2: float = 0.999
    epsilon: float = 1e-8
    weight_decay: float = 1e-8
    loss: str = '1*L1'
    skip_threshold: float = 1.5
    dual_weight: float = 0.1
    save: str = './workspace/experiment/drn-l/gkd_dc2_unlabeld_X4_10_grayscale/'
    print_every: int = 10
    save_results: bool = True
    dual: bool = True
    patience: int = 10
    min_delta: float = 0.0
    dataset: str = ''
    classe: str = ''
    slurm: bool = False
    ssim_window_size: int = 11
    best_auc: float = 1.0
    precision: str = 'fp32'
    use_graph: bool = True


@dataclass
class DRCT:
    model_name: str = 'drct'
    n_threads: int = 1
    cpu: bool = False
    n_GPUs: int = 1
    seed: int = 1
    data_dir: str = './workspace/gkd/DC2/unlabeled/HR_512_grayscale/'
    data_train: str = ''
    data_test: str = ''
    data_range: str = '1-260/261-299'
    scale: object = 4
    patch_size: int = 512
    rgb_range: int = 255
    n_colors: int = 1
    no_augment: bool = False
    pre_train: str = '.'
    pre_train_dual: str = '.'
    negval: float = 0.2
    test_every: int = 30
    epochs: int = 10
    batch_size: int = 2
    self_ensemble: bool = False
    test_only: bool = False
    lr: float = 1e-4
    eta_min: float = 1e-7
    beta1: float = 0.9
    beta2: float = 0.999
    epsilon: float = 1e-8
    loss: str = '1*L1'
    skip_threshold: float = 1e6
    dual_weight: float = 0.1
    save: str = './workspace/experiment/drct/gkd_dc2_unlabeled_X4_10_test_grayscale/'
    print_every: int = 10
    save_results: bool = True
    dual: bool = False
    upscale: int = 4
    img_size: int = 128
    window_size: int = 16
    compress_ratio: int = 3
    squeeze_factor: int = 30
    conv_scale: float = 0.01
    overlap_ratio: float = 0.5
    img_range: float = 1.0
    depths: tuple = (6,) * 12
    embed_dim: int = 180
    num_heads: tuple = (6,) * 12
    mlp_ratio: int = 2
    upsampler: str = 'pixelshuffle'
    resi_connection: str = '1conv'
    ema_decay: float = 0.999
    weight_decay: float = 0.0
    betas: tuple = (0.9, 0.99)
    patience: int = 10
    min_delta: float = 0.0
    dataset: str = ''
    classe: str = ''
    slurm: bool = False
    ssim_window_size: int = 11
    best_auc: float = 1.0
    precision: str = 'fp32'
    use_graph: bool = True


def setup_opt_drn(opt: DRN, best_auc, ssim_window_size, dataset, classe, slurm, scale, no_augment, n_colors, epochs,
                  batch_size, patch_size, data_dir, save, data_range, test_every, print_every, patience, min_delta,
                  n_threads, pre_trained, pre_trained_dual, loss) -> DRN:
    """src/main.py:144-205 (same positional signature)."""
    opt.scale = [pow(2, s + 1) for s in range(int(np.log2(scale)))]
    if scale == 2:
        opt.n_blocks, opt.n_feats = 44, 40
    elif scale == 4:
        opt.n_blocks, opt.n_feats = 40, 20
    elif scale == 8:
        opt.n_blocks, opt.n_feats = 36, 10
    else:
        print(f"No setup for this scale: {scale}")
    opt.no_augment, opt.n_colors, opt.epochs, opt.batch_size = no_augment, n_colors, epochs, batch_size
    opt.patch_size, opt.data_dir, opt.save = patch_size, data_dir, save
    opt.test_every, opt.print_every, opt.patience, opt.min_delta = test_every, print_every, patience, min_delta
    opt.n_threads, opt.pre_train, opt.pre_train_dual, opt.loss = n_threads, pre_trained, pre_trained_dual, loss
    opt.dataset, opt.classe, opt.slurm = dataset, classe, slurm
    opt.ssim_window_size, opt.best_auc = ssim_window_size, best_auc
    return opt


def setup_opt_drct(opt: DRCT, best_auc, ssim_window_size, dataset, classe, slurm, scale, no_augment, n_colors, epochs,
                   batch_size, patch_size, img_size, data_dir, save, data_range, test_every, print_every, patience,
                   min_delta, n_threads, pre_trained, loss) -> DRCT:
    """src/main.py:243-294: note ``window_size = img_size // 4``."""
    opt.upscale = scale
    opt.scale = [scale]
    opt.no_augment, opt.n_colors, opt.epochs, opt.batch_size = no_augment, n_colors, epochs, batch_size
    opt.patch_size, opt.data_dir, opt.data_range, opt.save = patch_size, data_dir, data_range, save
    opt.test_every, opt.print_every, opt.img_size = test_every, print_every, img_size
    opt.patience, opt.min_delta, opt.n_threads, opt.pre_train = patience, min_delta, n_threads, pre_trained
    opt.window_size = img_size // 4
    opt.loss, opt.dataset, opt.classe, opt.slurm = loss, dataset, classe, slurm
    opt.ssim_window_size, opt.best_auc = ssim_window_size, best_auc
    return opt


def _with_config(parser: argparse.ArgumentParser, pre_args) -> None:
    if pre_args.config is not None and os.path.isfile(pre_args.config):
        with open(pre_args.config, 'r') as f:
            cfg = yaml.safe_load(f) or {}
        parser.set_defaults(**{k.replace('-', '_'): v for k, v in cfg.items()})


def parse_train_args(argv: Optional[List[str]] = None) -> argparse.Namespace:
    """src/main.py:207-241 (+ --dtype, --gpus; wider --scale / --resolution choices)."""
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument('--config', type=str, default=None)
    pre_args, _ = pre.parse_known_args(argv)
    p = argparse.ArgumentParser(description='Training/Evaluation entrypoint', parents=[pre])
    p.add_argument('--model-type', type=str, default='drct', choices=['drct', 'drn-l'])
    p.add_argument('--dataset', type=str, default='mvtec', choices=['mvtec'])
    p.add_argument('--classe', type=str, default='grid', choices=['grid', 'carpet'])
    p.add_argument('--scale', type=int, default=4, choices=[2, 4, 8])
    p.add_argument('--resolution', type=int, default=128, choices=[32, 64, 128, 256, 512, 1024])
    p.add_argument('--epochs', type=int, default=2)
    p.add_argument('--batch-size', type=int, default=4)
    p.add_argument('--lr', type=float, default=2e-4)      # parsed but never applied, like the reference (H5)
    p.add_argument('--no-augment', action='store_true')
    p.add_argument('--device', type=str, default='auto', choices=['auto', 'cuda', 'mps', 'cpu'])
    p.add_argument('--data-root', type=str, default='auto')
    p.add_argument('--save-dir', type=str, default='./workspace/experiment')
    p.add_argument('--pretrain', action='store_true')
    p.add_argument('--test-only', action='store_true')
    p.add_argument('--workers', type=int, default=0 if sys.platform == 'darwin' else 4)
    p.add_argument('--dtype', type=str, default='fp32', choices=['fp32', 'bf16'], help="training precision (the evaluation at the end of "
                   "a run uses the evaluator's split-bf16 mode when this is fp32)")
    p.add_argument('--gpus', type=int, default=1)
    _with_config(p, pre_args)
    return p.parse_args(argv)


def parse_eval_args(argv=None) -> argparse.Namespace:
    """src/evaluate.py:20-45 (+ --dtype, --gpus)."""
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument('--config', type=str, default=None)
    pre_args, _ = pre.parse_known_args(argv)
    p = argparse.ArgumentParser(description='Evaluation entrypoint', parents=[pre])
    p.add_argument('--model-type', type=str, default='drct', choices=['drct', 'drn-l'])
    p.add_argument('--dataset', type=str, default='mvtec', choices=['mvtec'])
    p.add_argument('--classe', type=str, default='grid')
    p.add_argument('--scale', type=int, default=4)
    p.add_argument('--resolution', type=int, default=128)
    p.add_argument('--device', type=str, default='auto', choices=['auto', 'cuda', 'mps', 'cpu'])
    p.add_argument('--data-root', type=str, default='auto')
    p.add_argument('--run-dir', type=str, default='')
    p.add_argument('--checkpoint', type=str, default='')
    p.add_argument('--batch-size', type=int, default=1)
    p.add_argument('--output-dir', type=str, default='')
    p.add_argument('--save-images', action='store_true', default=True)
    p.add_argument('--workers', type=int, default=0 if sys.platform == 'darwin' else 4)
    p.add_argument('--dtype', type=str, default='bf16x3', choices=['fp32', 'bf16', 'bf16x3'],
                   help="bf16x3 = split-bf16 (fp32-grade outputs on the bf16 matrix pipe; the default: AUC parity), fp32 = exact-fp32 MFMA, "
                        "bf16 = fastest, outside the +-0.002 AUC bar")
    p.add_argument('--gpus', type=int, default=1)
    _with_config(p, pre_args)
    return p.parse_args(argv)


def build_opt(model_type: str, class_name: str, resolution: int, scale: int, batch_size: int = 1, dtype: str = 'fp32',
              pre_train: str = '.', pre_train_dual: str = '.', data_root: str = 'auto', save: str = './workspace/eval',
              epochs: int = 1, no_augment: bool = True):
    """The option object main.py / evaluate.py build before calling Model (src/main.py:398-471,
    src/evaluate.py:293-334)."""
    n_colors = 3 if class_name == 'carpet' else 1
    img_size = resolution // scale
    if data_root == 'auto':
        data_root = f"data/mvtec_{resolution}"
    data_dir = f"{data_root}/{class_name}/train/good"
    if model_type == 'drn-l':
        opt = setup_opt_drn(DRN(), 0.0, 11, 'mvtec', class_name, False, scale, no_augment, n_colors, epochs, batch_size,
                            resolution, data_dir, save, '', 1, 1, 1, 0.0, 4, pre_train, pre_train_dual, '1*L1')
    else:
        opt = setup_opt_drct(DRCT(), 0.0, 11, 'mvtec', class_name, False, scale, no_augment, n_colors, epochs, batch_size,
                             resolution, img_size, data_dir, save, '', 1, 1, 1, 0.0, 4, pre_train, '1*L1')
    opt.model_name = model_type
    opt.data_root = data_root
    opt.precision = dtype
    return opt
