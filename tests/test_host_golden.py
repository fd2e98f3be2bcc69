"""CPU: the host-side restatements against outputs of THE REFERENCE's own functions (tests/golden/host_golden.npz,
written by tests/golden/make_host_golden.py from the imported reference):
  * oracle/host_ref.py - loss values + autograd gradients of every loss type (src/loss.py), get_patch / augment under
    fixed random seeds (src/data.py:21-50);
  * the product's data.py (no GPU involved): the same patches, flips, channel handling and tensor scaling, plus the
    virtual-epoch rule, folder fall-backs and rank sharding of the loader."""
import random

import numpy as np
import pytest
import torch

from oracle import host_ref as O

LOSS_CASES = ["gray48", "rgb40", "gray16"]
SPECS = ["1*L1", "1*MSE", "1*PSNR", "1*SSIM", "0.7*L1+0.3*SSIM", "1*MSE+0.05*PSNR"]


def _key(spec):
    return spec.replace("*", "x").replace("+", "_")


@pytest.mark.parametrize("tag", LOSS_CASES + ["gray_crop"])
def test_loss_oracle_matches_reference_values_and_gradients(host_golden, tag):
    g = host_golden
    sr, hr = torch.from_numpy(g[f"loss/{tag}/sr"]), torch.from_numpy(g[f"loss/{tag}/hr"])
    for spec in (["1*SSIM"] if tag == "gray_crop" else SPECS):
        x = sr.clone().requires_grad_(True)
        val = O.total_loss(spec, x, hr, batch_size=3, rgb_range=255)
        val.backward()
        ref_v, ref_g = float(g[f"loss/{tag}/{_key(spec)}/value"]), g[f"loss/{tag}/{_key(spec)}/grad"]
        assert abs(float(val.detach()) - ref_v) <= 2e-6 * max(1.0, abs(ref_v)), (tag, spec, float(val.detach()), ref_v)
        assert np.abs(x.grad.numpy() - ref_g).max() <= 2e-5 * np.abs(ref_g).max() + 1e-12, (tag, spec)
        log = g[f"loss/{tag}/{_key(spec)}/log"]                       # one row: the weighted terms (+ total)
        assert log.shape == (1, spec.count("+") + 1 + (1 if "+" in spec else 0))
        assert abs(log[0, -1] - ref_v) <= 1e-5 * max(1.0, abs(ref_v))


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_get_patch_and_augment_match_reference_under_fixed_seeds(host_golden, impl):
    g = host_golden
    hr, lr4, lr2 = g["data/hr"], g["data/lr4"], g["data/lr2"]
    from srad_amd import data as D
    for seed in range(6):
        random.seed(seed)
        if impl == "oracle":
            pl, ph = O.get_patch([lr4, lr2], hr, 32, [4, 2])
            al, ah = O.augment(pl, ph)
        else:
            pl, ph = D.get_patch([lr4, lr2], hr, patch_size=32, scale=[4, 2], multi_scale=True)
            al, ah = D.augment(pl, ph)
        assert np.array_equal(pl[0], g[f"data/seed{seed}/patch_lr4"]) and np.array_equal(pl[1], g[f"data/seed{seed}/patch_lr2"])
        assert np.array_equal(ph, g[f"data/seed{seed}/patch_hr"])
        assert np.array_equal(al[0], g[f"data/seed{seed}/aug_lr4"]) and np.array_equal(al[1], g[f"data/seed{seed}/aug_lr2"])
        assert np.array_equal(ah, g[f"data/seed{seed}/aug_hr"])
        # the LR patches are the HR patch's region: pixel (y, x) of LR_s covers HR (s y, s x)
        hy, hx = divmod(int(ph[0, 0, 0]), 72)
        assert divmod(int(pl[0][0, 0, 0]) - 100000, 18) == (hy // 4, hx // 4) and divmod(int(pl[1][0, 0, 0]) - 200000, 36) == (hy // 2, hx // 2)
    random.seed(3)
    pl, ph = (O.get_patch([lr4], hr[:, :64], 64, [4]) if impl == "oracle"
              else D.get_patch([lr4], hr[:, :64], patch_size=64, scale=[4]))
    assert np.array_equal(pl[0], g["data/full/patch_lr4"]) and np.array_equal(ph, g["data/full/patch_hr"])


def test_set_channel_and_np2tensor_match_reference(host_golden):
    from srad_amd import data as D
    g = host_golden
    cl, ch = D.set_channel([g["data/setchan/in"]], g["data/setchan/in"], n_channels=3)
    assert np.array_equal(cl[0], g["data/setchan/lr"]) and np.array_equal(ch, g["data/setchan/hr"])
    g2 = g["data/setchan/in"]
    tl, th = D.np2Tensor([g2[:, :, None]], g2[:, :, None], rgb_range=1)
    assert torch.equal(tl[0], torch.from_numpy(g["data/np2tensor/lr"])) and torch.equal(th, torch.from_numpy(g["data/np2tensor/hr"]))
    # RGB -> Y: scikit-image's published luma (parity unpinned: skimage is not importable here); white -> 235, black -> 16
    rgb = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0]]], dtype=np.uint8)
    y = D.set_channel([rgb], rgb, n_channels=1)[1][:, :, 0]
    assert np.allclose(y, [[235.0, 16.0, 16 + 65.481]]) and np.allclose(y, O.rgb2y(rgb))


def _write_class(tmp, n, sizes=(16, 16), scales=(4,), layout="LR_s"):
    from PIL import Image
    d = tmp / "good"
    (d / "HR").mkdir(parents=True)
    for i in range(n):
        hr = (np.arange(sizes[0] * sizes[1]).reshape(sizes) % 200 + i).astype(np.uint8)
        Image.fromarray(hr).save(d / "HR" / f"{i:03d}.png")
        for s in scales:
            lr = hr[::s, ::s]
            if layout == "LR_s":
                (d / f"LR_{s}").mkdir(exist_ok=True)
                Image.fromarray(lr).save(d / f"LR_{s}" / f"{i:03d}.png")
            elif layout == "bicubic":
                (d / "LR_bicubic" / f"X{s}").mkdir(parents=True, exist_ok=True)
                Image.fromarray(lr).save(d / "LR_bicubic" / f"X{s}" / f"{i:03d}x{s}.png")
            else:
                (d / "LR").mkdir(exist_ok=True)
                Image.fromarray(lr).save(d / "LR" / f"{i:03d}.png")
    return str(d)


class _Args:
    n_colors, rgb_range, no_augment, patch_size, batch_size, test_every, test_only, seed = 1, 255, False, 16, 4, 8, False, 1

    def __init__(self, data_dir, scale):
        self.data_dir, self.scale = data_dir, list(scale)


@pytest.mark.parametrize("layout", ["LR_s", "bicubic", "LR"])
def test_dataset_virtual_epoch_and_folder_fallbacks(tmp_path, layout):
    """src/data.py:101-105,109-155: an epoch is test_every * batch_size samples whatever the folder holds; index i is image
    i % n below n * (len // n) and a random image above; LR is looked up as LR_bicubic/X{s}/{name}x{s}.png, LR_{s}/, LR/."""
    from srad_amd import data as D
    d = _write_class(tmp_path, 5, layout=layout)
    a = _Args(d, [4])
    ds = D.MVTec(a, train=True)
    assert len(ds) == 32 and ds.random_border == 30 and ds.scale == [4]
    assert [ds._get_index(i) for i in range(12)] == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 0, 1]
    assert all(0 <= ds._get_index(i, random.Random(i)) < 5 for i in (30, 31))
    for i in range(32):
        assert O.virtual_index(i, 5, 8, 4) == (i % 5 if i < 30 else None, 32)
    lr, hr, name = ds[7]
    assert lr[0].shape == (1, 4, 4) and hr.shape == (1, 16, 16) and lr[0].dtype == torch.float32 and name == "002"
    test = D.MVTec(a, train=False)
    assert len(test) == 5 and test[4][2] == "004"
    if layout != "LR":                                                     # the LR/ fall-back serves every scale
        with pytest.raises(FileNotFoundError, match="LR image not found"):
            D.MVTec(_Args(d, [8]), train=True)


def test_loader_batches_epoch_length_and_rank_sharding(tmp_path):
    """The loader yields len(dataset) / batch_size global batches per epoch (src/main.py:448: test_every = 256 //
    batch_size); with world 2 every rank gets rank::2 of each global batch - same samples, same per-sample draws."""
    from srad_amd import data as D
    d = _write_class(tmp_path, 6, sizes=(32, 32), scales=(2, 4))
    a = _Args(d, [2, 4])
    a.patch_size = 16
    full = D.Data(a).loader_train
    assert len(full) == 8 and len(full.dataset) == 32
    full.set_epoch(2)
    whole = list(full)
    assert len(whole) == 8 and [t.shape for t in whole[0][0]] == [(4, 1, 4, 4), (4, 1, 8, 8)] and whole[0][1].shape == (4, 1, 16, 16)
    parts = []
    for r in range(2):
        ld = D.Data(a, rank=r, world=2).loader_train
        ld.set_epoch(2)
        parts.append(list(ld))
    for b in range(8):
        merged_hr = torch.stack([parts[b2 % 2][b][1][b2 // 2] for b2 in range(4)])
        assert torch.equal(merged_hr, whole[b][1])                        # union over ranks == the reference's minibatch
        assert tuple(parts[0][b][2]) + tuple(parts[1][b][2]) == (whole[b][2][0], whole[b][2][2], whole[b][2][1], whole[b][2][3])
    full.set_epoch(3)
    assert not torch.equal(list(full)[0][1], whole[0][1])                 # reshuffled every epoch
    with pytest.raises(ValueError, match="GLOBAL minibatch"):
        D.Data(a, rank=0, world=3)
    # test loader: HR cropped to LR * scale, batch of one, folder order
    a.batch_size = 1
    t = list(D.Data(a).loader_test)
    assert len(t) == 6 and t[0][1].shape == (1, 1, 32, 32) and t[3][2] == ("003",)
