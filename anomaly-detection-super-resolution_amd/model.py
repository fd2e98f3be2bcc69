"""Model registry + wrapper: drop-in for reference src/model.py (``make_model`` 46-52, ``Model``
54-170, dual ``DownBlock`` 8-44) whose networks run on the HIP engine."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .nets import DRCT, DRN, DownBlock  # noqa: F401  (DownBlock re-exported like the reference module)


def make_model(opt):
    """src/model.py:46-52 - unknown names print and return None, as the reference does."""
    if opt.model_name == 'drct':
        return DRCT(opt)
    elif opt.model_name == 'drn-l':
        return DRN(opt)
    else:
        print(f"No model with this name: {opt.model_name}")


class _NullLog:
    log_file = None

    def write_log(self, msg, refresh=False):
        print(msg)


class Model(nn.Module):
    """Same attributes and methods as the reference wrapper: ``forward(x, idx_scale=0)``, ``device``,
    ``model``, ``dual_models``, ``get_model()``, ``state_dict()``, ``save(path, is_best)``,
    ``load(pre_train, pre_train_dual, cpu)``, ``count_parameters(model)``."""

    def __init__(self, opt, ckp=None, dual_model=False):
        super().__init__()
        print('Making model...')
        ckp = ckp or _NullLog()
        self.opt = opt
        self.scale = opt.scale
        self.idx_scale = 0
        self.self_ensemble = getattr(opt, 'self_ensemble', False)
        self.cpu = getattr(opt, 'cpu', False)
        if self.cpu:
            raise RuntimeError("--device cpu: this build runs the SR path on the HIP engine only; the CPU path is "
                               "the reference itself (or oracle/ for tests).  There is no CPU fallback.")
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the HIP engine cannot run (there is no CPU fallback)")
        self.device = torch.device('cuda', torch.cuda.current_device())
        ckp.write_log(f"Using device: {self.device}")
        self.n_GPUs = getattr(opt, 'n_GPUs', 1)
        self.dual_model = dual_model
        self.model = make_model(opt).to(self.device)
        if self.dual_model:
            self.dual_models = []
            for _ in self.opt.scale:
                self.dual_models.append(DownBlock(opt, 2).to(self.device))
        self.load(getattr(opt, 'pre_train', '.'), getattr(opt, 'pre_train_dual', '.'), cpu=False)
        if not getattr(opt, 'test_only', False) and getattr(ckp, 'log_file', None) is not None:
            print(self.model, file=ckp.log_file)
            if self.dual_model:
                print(self.dual_models, file=ckp.log_file)
        num_parameter = self.count_parameters(self.model)
        ckp.write_log(f"The number of parameters is {num_parameter / 1000 ** 2:.2f}M")

    def forward(self, x, idx_scale=0):
        self.idx_scale = idx_scale
        return self.model(x)

    def get_model(self):
        return self.model

    def get_dual_model(self, idx):
        return self.dual_models[idx]

    def state_dict(self, **kwargs):
        return self.get_model().state_dict(**kwargs)

    def count_parameters(self, model):
        return sum(p.numel() for p in model.parameters() if p.requires_grad)

    def save(self, path, is_best=False):
        """model/model_latest.pt (+ model_best.pt) = raw state_dict; dual models as a LIST of
        state_dicts in dual_model_{latest,best}.pt (src/model.py:123-147)."""
        target = self.get_model()
        os.makedirs(os.path.join(path, 'model'), exist_ok=True)
        torch.save(target.state_dict(), os.path.join(path, 'model', 'model_latest.pt'))
        if is_best:
            torch.save(target.state_dict(), os.path.join(path, 'model', 'model_best.pt'))
        if self.dual_model:
            dual_models = [self.get_dual_model(i).state_dict() for i in range(len(self.dual_models))]
            torch.save(dual_models, os.path.join(path, 'model', 'dual_model_latest.pt'))
            if is_best:
                torch.save(dual_models, os.path.join(path, 'model', 'dual_model_best.pt'))

    def load(self, pre_train='.', pre_train_dual='.', cpu=False):
        """``torch.load(weights_only=True)`` + ``load_state_dict(strict=False)`` (src/model.py:149-170)."""
        if pre_train != '.':
            print('Loading model from {}'.format(pre_train))
            self.get_model().load_state_dict(torch.load(pre_train, weights_only=True, map_location=self.device),
                                             strict=False)
        if self.dual_model and pre_train_dual != '.':
            print('Loading dual model from {}'.format(pre_train_dual))
            dual_models = torch.load(pre_train_dual, weights_only=True, map_location=self.device)
            for i in range(len(self.dual_models)):
                self.get_dual_model(i).load_state_dict(dual_models[i], strict=False)
