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

import time
from pathlib import Path

import numpy as np
import torch

_STAMP = '%Y-%m-%d-%H:%M:%S'                      # the timestamp config.txt opens with


def _config_text(opt, stamp: str) -> str:
    """config.txt's wire format: stamp, blank, ``key: value`` per option field in declaration order, blank."""
    fields = ''.join(f'{key}: {value}\n' for key, value in vars(opt).items())
    return f'{stamp}\n\n{fields}\n'


class Checkpoint():
    """Attributes the callers use: ``opt``, ``ok``, ``dir``, ``log`` ([epochs, 2 * n_scales] PSNR | SSIM table), ``log_file``."""

    def __init__(self, opt):
        stamp = time.strftime(_STAMP)
        if opt.save == '.':                                      # the reference's fall-back location
            opt.save = '../experiment/EXP/' + stamp
        self.opt, self.ok, self.dir = opt, True, opt.save
        self.log = torch.Tensor()
        root = Path(self.dir)
        for folder in (root, root / 'model', root / 'results'):
            folder.mkdir(parents=True, exist_ok=True)
        resumed = (root / 'log.txt').exists()                    # an existing run directory is appended to, never truncated
        mode = 'a' if resumed else 'w'
        self.log_file = open(self._log_path, mode)
        with open(root / 'config.txt', mode) as f:
            f.write(_config_text(opt, stamp))

    @property
    def _log_path(self) -> str:
        return self.dir + '/log.txt'

    def save(self, trainer, epochs, is_best=False, dual_model=False):
        trainer.model.save(self.dir, is_best=is_best)
        if getattr(trainer, 'loss', None) is not None:
            trainer.loss.save(self.dir)
        self.plot_psnr_ssim(trainer.get_last_epoch())
        out = Path(self.dir)
        torch.save(self.log, out / 'psnr_ssim_log.pt')
        torch.save(trainer.optimizer.state_dict(), out / 'optimizer.pt')
        if dual_model:
            torch.save({i: o.state_dict() for i, o in enumerate(trainer.dual_optimizers)}, out / 'dual_optimizers.pt')

    def add_log(self, log):
        """Append one row block to the validation table."""
        self.log = log if self.log.numel() == 0 else torch.cat((self.log, log), dim=0)

    def write_log(self, log, refresh=False):
        """One line to stdout and to log.txt; ``refresh`` re-opens the file so the line is on disk."""
        print(log)
        print(log, file=self.log_file)
        if refresh:
            self.log_file.close()
            self.log_file = open(self._log_path, 'a')

    def done(self):
        if not self.log_file.closed:
            self.log_file.close()

    def plot_psnr_ssim(self, epoch):
        if self.log.numel() == 0 or self.log.dim() < 2 or self.log.shape[1] < 2:
            self.write_log('No evaluation logs available; skipping PSNR/SSIM plot')

    def save_results_nopostfix(self, filename, sr, scale):
        from PIL import Image
        from . import metrics as M
        folder = Path(f'{self.dir}/results/{self.opt.data_test}/x{scale}')
        folder.mkdir(parents=True, exist_ok=True)
        pixels = M.to_u8_hwc(sr[0:1], self.opt.rgb_range)[0].cpu().numpy()        # truncating, like .byte()
        image = Image.fromarray(pixels[:, :, 0] if pixels.shape[2] == 1 else np.ascontiguousarray(pixels))
        image.save(folder / f'{filename}.png')
