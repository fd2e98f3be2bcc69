"""CLI entry with the reference's flags (src/main.py:390-474).  ``--model-type drct`` trains on the HIP engine
(forward + L1 + backward + fused Adam; under ``python -m torch.distributed.run`` the minibatch is sharded over the
ranks and gradients are all-reduced over RCCL while the backward runs); ``--model-type drn-l`` trains DRN-L with its
dual regression models (x2 / x4; the x8 preset is inference-only).  ``--test-only`` evaluates an existing run."""
from __future__ import annotations

import os
import sys

from .options import build_opt, parse_train_args


def main(argv=None):
    args = parse_train_args(argv)
    print(f"Model: {args.model_type}\nDataset: {args.dataset}\nClass: {args.classe}\n"
          f"Resolution: {args.resolution}\nScale: {args.scale}")
    if args.device == 'cpu':
        raise SystemExit("--device cpu is the reference's own path; this build has no CPU fallback")
    if args.test_only:
        from . import evaluate
        return evaluate.main([a for a in (argv if argv is not None else sys.argv[1:]) if a != '--test-only'])
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    torch.manual_seed(1)                                     # src/main.py:89
    data_root = args.data_root if getattr(args, "data_root", None) else "auto"
    save = os.path.join(args.save_dir, args.model_type, f"{args.dataset}_{args.classe}_{args.resolution}_X{args.scale}")
    opt = build_opt(args.model_type, args.classe, args.resolution, args.scale, batch_size=args.batch_size,
                    dtype=getattr(args, "dtype", "bf16"), data_root=data_root, save=save, epochs=args.epochs,
                    no_augment=getattr(args, "no_augment", True))
    from .trainer import train_drct, train_drn
    out = train_drn(opt) if args.model_type == 'drn-l' else train_drct(opt)
    if world > 1:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
