"""CLI entry with the reference's flags and run structure (src/main.py:296-474): build the option object, then
``train_drct`` / ``train_drn`` = Checkpoint -> Data -> Model -> Loss -> Trainer -> ``while not t.terminate(): t.train()``
-> post-training validation PSNR / SSIM on ``val/good`` -> ``checkpoint.save``.

``--model-type drct`` trains on the HIP engine (forward + L1 + backward + fused Adam); ``--model-type drn-l`` trains
DRN-L with its dual regression models (x2 / x4 / x8); ``--test-only`` evaluates an
existing run.  Data parallel (BASELINE config C4): ``--gpus N`` starts N ranks (or launch under
``python -m torch.distributed.run``); ``--batch-size`` stays the GLOBAL minibatch, every rank takes ``rank::N`` of it
and the gradients are all-reduced over RCCL while the backward runs.  DropPath draws are seeded ``seed + rank``."""
from __future__ import annotations

import copy
import datetime
import os
import random
import sys
import time

from .options import DRCT, DRN, parse_train_args, setup_opt_drct, setup_opt_drn


def set_seed(seed: int) -> None:
    """src/main.py:26-33."""
    import numpy as np
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def _ranks():
    import torch.distributed as dist
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def _train(opt, dual_model: bool) -> None:
    """src/main.py train_drn (296-335) / train_drct (337-388)."""
    import torch
    from .checkpoint import Checkpoint
    from .data import Data
    from .loss import Loss
    from .model import Model
    from .trainer import Trainer
    rank, world = _ranks()
    set_seed(opt.seed)
    ckp = Checkpoint(opt) if rank == 0 else None               # one writer; the other ranks only train
    loader = Data(opt, rank, world)
    model = Model(opt, ckp, dual_model=dual_model)             # same seed on every rank: identical replicas
    torch.manual_seed(opt.seed + rank)                         # per-rank DropPath streams
    loss = Loss(opt, ckp) if not opt.test_only else None
    t = Trainer(opt, loader, model, loss, ckp, dual_model=dual_model)
    start_time = time.time()
    while not t.terminate():
        t.train()
    print("Training completed")
    if ckp is not None:
        ckp.write_log(f"Total Training Time: {((time.time() - start_time) / 3600):.2f}")
    if rank == 0 and not opt.test_only:
        # post-training evaluation on val/good (PSNR / SSIM), src/main.py:369-380 - same resolution as the run
        try:
            eval_opt = copy.deepcopy(opt)
            eval_opt.test_only, eval_opt.no_augment, eval_opt.batch_size = True, True, 1
            eval_opt.data_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.normpath(opt.data_dir))), 'val', 'good')
            eval_opt.data_test = 'mvtec_val_good'               # eval_opt only (src/main.py:322,374): the run's log lines / results dir keep its own data_test
            t.loader_test = Data(eval_opt).loader_test
            t.test()
        except Exception as e:                                  # the reference prints and carries on
            print(f"Evaluation skipped due to error: {e}")
        ckp.write_log("Skipping anomaly AUC on validation (good-only split)")
        ckp.save(t, opt.epochs, is_best=True, dual_model=dual_model)
        ckp.done()


def train_drn(opt_drn: DRN) -> None:
    _train(opt_drn, dual_model=True)


def train_drct(opt_drct: DRCT) -> None:
    _train(opt_drct, dual_model=False)


def build_train_opt(args):
    """src/main.py:398-471: paths, virtual dataset length (256 samples per epoch), the option object."""
    n_colors = 3 if (args.dataset == 'mvtec' and args.classe == 'carpet') else 1
    res, scale = args.resolution, args.scale
    date_string = datetime.datetime.now().strftime("%H:%M:%S")
    data_root = f"data/mvtec_{res}" if args.data_root == 'auto' else args.data_root
    data_dir = f"{data_root}/{args.classe}/train/good"
    save = f"{args.save_dir}/{args.model_type}/mvtec_{args.classe}_{res}_X{scale}{date_string}/"
    data_range = '1-210/211-264' if args.classe == 'grid' else '1-224/225-280'
    test_every = 256 // args.batch_size
    if args.model_type == 'drn-l':
        pre, pre_dual = ((f'workspace/pretrained_model_weights/DRNL{scale}x.pt',
                          f'workspace/pretrained_model_weights/DRNL{scale}x_dual_model.pt') if args.pretrain else ('.', '.'))
        opt = setup_opt_drn(DRN(), 0.0, 11, args.dataset, args.classe, False, scale, args.no_augment, n_colors, args.epochs,
                            args.batch_size, res, data_dir, save, data_range, test_every, test_every, 1, 0.005, 4, pre, pre_dual,
                            '1*L1')
    else:
        pre = 'workspace/pretrained_model_weights/net_g_latest.pth' if args.pretrain else '.'
        opt = setup_opt_drct(DRCT(), 0.0, 11, args.dataset, args.classe, False, scale, args.no_augment, n_colors, args.epochs,
                             args.batch_size, res, res // scale, data_dir, save, data_range, test_every, test_every, 1, 0.005, 4,
                             pre, '1*L1')
    opt.model_name = args.model_type
    opt.test_only = args.test_only
    opt.precision = args.dtype
    opt.data_root = data_root
    return opt


def run(args):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")                          # RCCL over xGMI
    opt = build_train_opt(args)
    if world > 1:                                                # one run directory: rank 0's timestamp
        box = [opt.save]
        dist.broadcast_object_list(box, src=0)
        opt.save = box[0]
    (train_drn if args.model_type == 'drn-l' else train_drct)(opt)
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    args = parse_train_args(argv)
    print(f"Model: {args.model_type}\nDataset: {args.dataset}\nClass: {args.classe}\n"
          f"Resolution: {args.resolution}\nScale: {args.scale}")
    if args.device == 'cpu':
        raise SystemExit("--device cpu is the reference's own path; this build has no CPU fallback")
    if args.test_only:
        from . import evaluate
        return evaluate.main([a for a in (argv if argv is not None else sys.argv[1:]) if a != '--test-only'])
    from .launch import spawn
    if spawn(run, args.gpus, (args,)):       # --gpus N without a launcher: N fresh ranks, this process only waits
        return None
    return run(args)


if __name__ == "__main__":
    main()
