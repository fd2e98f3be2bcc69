"""Run directory, logs and result images - host-side mirror of reference ``src/checkpoint.py`` (``Checkpoint(opt)``, 9-125).
The on-disk format is the interface: ``evaluate --run-dir`` re-parses ``config.txt`` (src/evaluate.py:84-118) and loads
``model/model_{best,latest}.pt`` (src/evaluate.py:125-135), so every file keeps the reference's name and text layout:

    <save>/config.txt            timestamp, blank line, one ``key: value`` line per option field, blank line
    <save>/log.txt               everything ``write_log`` printed
    <save>/model/*.pt            written by ``Model.save`` (model.py)
    <save>/loss_log.pt           ``Loss.save``
    <save>/psnr_ssim_log.pt      the [epochs, 2] validation table ``Trainer.test`` fills
    <save>/optimizer.pt          ``trainer.optimizer.state_dict()``
    <save>/dual_optimizers.pt    DRN: {i: state_dict of dual optimizer i}  (the reference pickles the optimizer objects)
    <save>/results/<data_test>/x<scale>/<name>.png    ``save_results_nopostfix``: mul(255 / rgb_range) then ``.byte()`` =
                                 TRUNCATION (src/checkpoint.py:113-114; SURVEY.md hazard H2), via the engine's u8 kernel

The PDF plots (``plot_psnr_ssim``, 63-105) are out of scope; the method exists and logs that."""
from __future__ import annotations

import datetime
import os

import numpy as np
import torch


class Checkpoint():
    def __init__(self, opt):
        self.opt = opt
        self.ok = True
        self.log = torch.Tensor()
        now = datetime.datetime.now().strftime('%Y-%m-%d-%H:%M:%S')
        if opt.save == '.':
            opt.save = '../experiment/EXP/' + now
        self.dir = opt.save
        for sub in ('', '/model', '/results'):
            os.makedirs(self.dir + sub, exist_ok=True)
        open_type = 'a' if os.path.exists(self.dir + '/log.txt') else 'w'
        self.log_file = open(self.dir + '/log.txt', open_type)
        with open(self.dir + '/config.txt', open_type) as f:
            f.write(now + '\n\n')
            for arg in vars(opt):
                f.write('{}: {}\n'.format(arg, getattr(opt, arg)))
            f.write('\n')

    def save(self, trainer, epochs, is_best=False, dual_model=False):
        trainer.model.save(self.dir, is_best=is_best)
        if getattr(trainer, 'loss', None) is not None:
            trainer.loss.save(self.dir)
        self.plot_psnr_ssim(trainer.get_last_epoch())
        torch.save(self.log, os.path.join(self.dir, 'psnr_ssim_log.pt'))
        torch.save(trainer.optimizer.state_dict(), os.path.join(self.dir, 'optimizer.pt'))
        if dual_model:
            torch.save({i: o.state_dict() for i, o in enumerate(trainer.dual_optimizers)},
                       os.path.join(self.dir, 'dual_optimizers.pt'))

    def add_log(self, log):
        self.log = torch.cat([self.log, log])

    def write_log(self, log, refresh=False):
        print(log)
        self.log_file.write(log + '\n')
        if refresh:
            self.log_file.close()
            self.log_file = open(self.dir + '/log.txt', 'a')

    def done(self):
        self.log_file.close()

    def plot_psnr_ssim(self, epoch):
        if self.log.numel() == 0 or self.log.dim() < 2 or self.log.shape[1] < 2:
            self.write_log('No evaluation logs available; skipping PSNR/SSIM plot')

    def save_results_nopostfix(self, filename, sr, scale):
        from PIL import Image
        from . import metrics as M
        apath = '{}/results/{}/x{}'.format(self.dir, self.opt.data_test, scale)
        os.makedirs(apath, exist_ok=True)
        nd = M.to_u8_hwc(sr[0:1], self.opt.rgb_range)[0].cpu().numpy()        # truncating, like .byte()
        im = Image.fromarray(nd[:, :, 0]) if nd.shape[2] == 1 else Image.fromarray(np.ascontiguousarray(nd))
        im.save('{}.png'.format(os.path.join(apath, filename)))
