"""Paired LR/HR PNG data for training and testing - host-side mirror of reference ``src/data.py`` (same names,
arguments and sampling rules; the images feed the HIP engine, nothing here runs on the GPU).

What is kept from the reference, with the line it follows:
  * folder layout ``{data_dir}/HR/*.png`` with the LR twin under ``LR_bicubic/X{s}/{name}x{s}.png``, ``LR_{s}/{name}.png``
    or ``LR/{name}.png`` (src/data.py:109-136), scales listed coarsest first (``args.scale`` reversed, 74-75);
  * the VIRTUAL epoch: ``test_every * batch_size`` samples per epoch whatever the folder holds - index ``i`` is image
    ``i % n`` below ``n * (len // n)`` and a uniformly drawn image above it (101-105, 148-155);
  * training samples: ``set_channel`` -> one random HR patch of ``patch_size`` snapped to a multiple of the largest
    scale, the matching LR patches of every scale (21-36) -> hflip / vflip / transpose drawn ONCE per sample and applied
    to every scale and the HR image together (38-50) -> ``np2Tensor`` = CHW float32 times ``rgb_range / 255`` (11-19);
  * test samples: the HR image cropped to ``LR * scale`` (177-181).

``get_patch`` / ``augment`` draw from Python's ``random`` in the reference's order (tx, ty; hflip, vflip, rot90), so with
the same ``random.seed`` they return the reference's own patches (fixtures: tests/golden/host_golden.npz).

Differences, all deliberate: images are decoded once and kept as arrays (an MVTec class is a few hundred tiles); the
loader is a plain iterable with a per-(seed, epoch, position) ``random.Random`` instead of DataLoader worker processes
(the reference's worker seeding makes its sample order irreproducible anyway); under ``torch.distributed`` every rank
draws the same global batch and keeps ``rank::world`` of it, so the union over ranks is the reference's minibatch.
RGB -> Y uses scikit-image's published ``rgb2ycbcr`` luma (Y = 16 + 65.481 R + 128.553 G + 24.966 B on [0,1] inputs);
scikit-image is not importable here, so that one conversion is parity-unpinned.
"""
from __future__ import annotations

import glob
import os
import random
from typing import Iterator, List, Sequence, Tuple

import numpy as np
import torch

_Y_FROM_RGB = np.array([65.481, 128.553, 24.966], dtype=np.float64)     # skimage.color.rgb2ycbcr, first row


def np2Tensor(*args, rgb_range=255):
    """(list of HWC arrays, HWC array) -> (list of CHW float tensors, CHW float tensor) scaled by rgb_range / 255."""
    def convert(img):
        t = torch.from_numpy(np.array(np.asarray(img).transpose((2, 0, 1)))).float()        # a copy: cached images stay untouched
        return t.mul_(rgb_range / 255)
    return [convert(a) for a in args[0]], convert(args[1])


def get_patch(*args, patch_size=96, scale=(2,), multi_scale=False, rng=random):
    """Random HR patch + the LR patches of every scale (src/data.py:21-36).  ``args = (lr_list, hr)``."""
    lrs, hr = args[0], args[-1]
    th, tw = hr.shape[:2]
    tx = rng.randrange(0, tw - patch_size + 1)
    ty = rng.randrange(0, th - patch_size + 1)
    tx -= tx % scale[0]
    ty -= ty % scale[0]
    out = []
    for img, s in zip(lrs, scale):
        ip, ix, iy = patch_size // s, tx // s, ty // s
        out.append(img[iy:iy + ip, ix:ix + ip, :])
    return [out, hr[ty:ty + patch_size, tx:tx + patch_size, :]]


def augment(*args, hflip=True, rot=True, rng=random):
    """One draw of (hflip, vflip, transpose) applied to every LR scale and the HR image (src/data.py:38-50)."""
    do_h = hflip and rng.random() < 0.5
    do_v = rot and rng.random() < 0.5
    do_t = rot and rng.random() < 0.5

    def apply(img):
        if do_h:
            img = img[:, ::-1, :]
        if do_v:
            img = img[::-1, :, :]
        if do_t:
            img = img.transpose(1, 0, 2)
        return img
    return [apply(a) for a in args[0]], apply(args[-1])


def rgb2y(img: np.ndarray) -> np.ndarray:
    """Luma of skimage.color.rgb2ycbcr for an HxWx3 image (uint8 -> /255 first, like img_as_float): float64 in [16, 235]."""
    a = np.asarray(img)
    f = a.astype(np.float64) / 255.0 if a.dtype == np.uint8 else a.astype(np.float64)
    return 16.0 + f @ _Y_FROM_RGB


def set_channel(*args, n_channels=3):
    """HW -> HW1; RGB -> Y when one channel is asked for; gray -> 3 copies when three are (src/data.py:52-65)."""
    def fix(img):
        if img.ndim == 2:
            img = img[:, :, None]
        c = img.shape[2]
        if n_channels == 1 and c == 3:
            img = rgb2y(img)[:, :, None]
        elif n_channels == 3 and c == 1:
            img = np.concatenate([img] * n_channels, 2)
        return img
    return [fix(a) for a in args[0]], fix(args[-1])


def _imread(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("L", "RGB"):
            im = im.convert("RGB" if im.mode in ("P", "RGBA", "CMYK", "YCbCr") else "L")
        return np.asarray(im)


class SRData:
    """Dataset of (LR list coarsest-first, HR, filename) - src/data.py:67-183."""

    def __init__(self, args, name='', train=True, benchmark=False):
        self.args, self.name, self.train, self.benchmark = args, name, train, benchmark
        self.split = 'train' if train else 'test'
        self.do_eval = True
        self.scale = list(args.scale)[::-1]
        self._set_filesystem(args.data_dir)
        self.images_hr, self.images_lr = self._scan()
        if not self.images_hr:
            raise FileNotFoundError(f"no HR images under {self.dir_hr}")
        self._set_dataset_length()
        self._cache = {}

    def _set_filesystem(self, data_dir):
        self.apath = data_dir
        self.dir_hr = os.path.join(self.apath, 'HR')
        self.dir_lr = self.apath
        self.ext = ('.png', '.png')

    def _scan(self):
        names_hr = sorted(glob.glob(os.path.join(self.dir_hr, '*' + self.ext[0])))
        names_lr = [[] for _ in self.scale]
        for f in names_hr:
            stem = os.path.splitext(os.path.basename(f))[0]
            for si, s in enumerate(self.scale):
                cands = (os.path.join(self.dir_lr, 'LR_bicubic', f'X{s}', f'{stem}x{s}{self.ext[1]}'),
                         os.path.join(self.apath, f'LR_{s}', f'{stem}{self.ext[1]}'),
                         os.path.join(self.apath, 'LR', f'{stem}{self.ext[1]}'))
                hit = next((c for c in cands if os.path.exists(c)), None)
                if hit is None:
                    raise FileNotFoundError(f"LR image not found for {stem} at scale {s}: tried {', '.join(cands)}")
                names_lr[si].append(hit)
        return names_hr, names_lr

    def _set_dataset_length(self):
        if self.train:
            self.dataset_length = self.args.test_every * self.args.batch_size
            self.random_border = len(self.images_hr) * (self.dataset_length // len(self.images_hr))
        else:
            self.dataset_length = len(self.images_hr)

    def __len__(self):
        return self.dataset_length

    def _get_index(self, idx, rng=None):
        if not self.train or idx < self.random_border:
            return idx % len(self.images_hr) if self.train else idx
        return (rng or random).randrange(len(self.images_hr))        # reference: np.random.randint (uniform)

    def _load_file(self, idx, rng=None):
        idx = self._get_index(idx, rng)
        if idx not in self._cache:
            f_hr = self.images_hr[idx]
            self._cache[idx] = ([_imread(self.images_lr[si][idx]) for si in range(len(self.scale))], _imread(f_hr),
                                os.path.splitext(os.path.basename(f_hr))[0])
        return self._cache[idx]

    def get_patch(self, lr, hr, rng=random):
        if self.train:
            lr, hr = get_patch(lr, hr, patch_size=self.args.patch_size, scale=self.scale,
                               multi_scale=len(self.scale) > 1, rng=rng)
            if not self.args.no_augment:
                lr, hr = augment(lr, hr, rng=rng)
        else:
            ih, iw = lr[0].shape[:2]
            hr = hr[0:ih * self.scale[0], 0:iw * self.scale[0]]
        return lr, hr

    def sample(self, idx, rng=random):
        lr, hr, filename = self._load_file(idx, rng)
        lr, hr = set_channel(lr, hr, n_channels=self.args.n_colors)
        lr, hr = self.get_patch(lr, hr, rng)
        lr_t, hr_t = np2Tensor(lr, hr, rgb_range=self.args.rgb_range)
        return lr_t, hr_t, filename

    def __getitem__(self, idx):
        return self.sample(idx)


class MVTec(SRData):
    def __init__(self, args, name='MVTec', train=True, benchmark=False):
        super().__init__(args, name=name, train=train, benchmark=benchmark)


class Loader:
    """What the trainer needs of ``torch.utils.data.DataLoader``: ``len()``, ``.dataset``, iteration yielding
    ``(list of [b,C,h,w] LR tensors, [b,C,H,W] HR tensor, filenames)``.  ``set_epoch`` reseeds the shuffle; with
    ``world > 1`` every rank walks the same global batches and keeps positions ``rank::world`` of each."""

    def __init__(self, dataset: SRData, batch_size: int, shuffle: bool, seed: int = 1, rank: int = 0, world: int = 1):
        if world > 1 and (batch_size % world or batch_size < world):
            raise ValueError(f"batch size {batch_size} cannot be sharded over {world} ranks: it is the GLOBAL minibatch of the "
                             f"reference (src/main.py --batch-size) and must be a positive multiple of the world size")
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), shuffle
        self.seed, self.rank, self.world, self.epoch = int(seed), int(rank), int(world), 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Tuple[List[torch.Tensor], torch.Tensor, Sequence[str]]]:
        n = len(self.dataset)
        order = list(range(n))
        if self.shuffle:
            random.Random(f"{self.seed}/{self.epoch}/order").shuffle(order)
        for b0 in range(0, n, self.batch_size):
            picks = list(enumerate(order[b0:b0 + self.batch_size]))[self.rank::self.world]
            if not picks:
                continue
            samples = [self.dataset.sample(idx, random.Random(f"{self.seed}/{self.epoch}/{b0 + j}")) for j, idx in picks]
            lrs = [torch.stack([s[0][k] for s in samples]) for k in range(len(samples[0][0]))]
            yield lrs, torch.stack([s[1] for s in samples]), tuple(s[2] for s in samples)


class Data:
    """src/data.py:195-219: ``loader_train`` (None with ``test_only``) and ``loader_test``."""

    def __init__(self, args, rank: int = 0, world: int = 1):
        self.loader_train = None
        seed = int(getattr(args, 'seed', 1))
        if not getattr(args, 'test_only', False):
            self.loader_train = Loader(MVTec(args, train=True), args.batch_size, shuffle=True, seed=seed, rank=rank, world=world)
        self.loader_test = Loader(MVTec(args, train=False), args.batch_size, shuffle=False, seed=seed)
