"""GPU: the evaluator end to end (SR forward -> truncating u8 -> window sweep -> AUCs) against the same
pipeline run through the CPU oracle, and the reference checkpoint surface of ``Model``."""
import os

import numpy as np
import pytest
import torch

from oracle import scorer_ref as O
from oracle import sr_ref as R
from srad_amd import spec as S

pytestmark = pytest.mark.gpu


def _pairs(n_good, n_bad, hr_size, scale, ch):
    y, sr, hr = O.synth_pairs(n_good, n_bad, hr_size, ch, seed=3)
    out = []
    for s_img, h_img in zip(sr, hr):
        src = s_img if True else h_img                      # "bad" images carry the planted blob in the LR input
        lr = src.reshape(hr_size // scale, scale, hr_size // scale, scale, ch).astype(np.float32).mean((1, 3))
        out.append((np.clip(np.rint(lr), 0, 255).astype(np.uint8), h_img))
    return y, out[:n_good], out[n_good:]


@pytest.mark.parametrize("model_type", ["drct", "drn-l"])
def test_evaluator_matches_oracle_pipeline(model_type):
    from srad_amd import evaluate as E
    from srad_amd import options as Opt
    from srad_amd.model import Model
    scale, hr_size = 4, 64
    opt = Opt.build_opt(model_type, 'grid', hr_size, scale)
    opt.use_graph = False
    if model_type == 'drct':
        opt.depths, opt.num_heads = (6,), (6,)
        cfg = S.DRCTConfig(in_chans=1, img_size=16, window_size=4, upscale=4, n_rdg=1)
        sd = S.synth_state(S.drct_spec(cfg), seed=9, gain=0.7, cfg=cfg)
    else:
        cfg = S.DRNConfig.for_scale(4, 1)
        sd = S.synth_state(S.drn_spec(cfg), seed=9, gain=0.4, cfg=cfg)
    model = Model(opt, None, dual_model=(model_type == 'drn-l'))
    model.get_model().load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    y, good, bad = _pairs(6, 8, hr_size, scale, 1)
    got = E.evaluate_on_test(opt, model, good, bad)
    # oracle pipeline on the CPU
    sr_u8, hr_u8 = [], []
    with torch.no_grad():
        for lr, hr in good + bad:
            x = torch.from_numpy(lr).permute(2, 0, 1)[None].float()
            out = R.drct_forward(sd, x, cfg) if model_type == 'drct' else R.drn_forward(sd, x, cfg)[-1]
            sr_u8.append(np.transpose(O.to_u8_trunc(out.numpy()[0]), (1, 2, 0)))
            hr_u8.append(hr)
    ref = O.evaluate_pairs(y, sr_u8, hr_u8)
    assert got["window_sizes"] == ref["window_sizes"]
    for k in ("auc_ssim", "auc_mse", "auc_psnr"):
        assert abs(got[k] - ref[k]) <= 0.002, (k, got[k], ref[k])          # north_star: AUC within +-0.002
        assert round(got[k], 3) == round(ref[k], 3)


def test_model_checkpoint_surface(tmp_path):
    from srad_amd import options as Opt
    from srad_amd.model import Model, make_model
    opt = Opt.build_opt('drn-l', 'carpet', 64, 2)
    m = Model(opt, None, dual_model=True)
    assert m.device.type == 'cuda' and len(m.dual_models) == 1 and m.get_model() is m.model
    m.save(str(tmp_path), is_best=True)
    files = sorted(os.listdir(tmp_path / 'model'))
    assert files == ['dual_model_best.pt', 'dual_model_latest.pt', 'model_best.pt', 'model_latest.pt']
    sd = torch.load(tmp_path / 'model' / 'model_best.pt', weights_only=True)
    assert list(sd.keys()) == list(S.drn_spec(S.DRNConfig.for_scale(2, 3)).keys())
    duals = torch.load(tmp_path / 'model' / 'dual_model_latest.pt', weights_only=True)
    assert isinstance(duals, list) and list(duals[0].keys()) == ['dual_module.0.0.weight', 'dual_module.1.weight']
    # round trip through load(): outputs identical
    x = torch.rand(1, 3, 16, 16, device='cuda') * 255
    with torch.no_grad():
        m.eval()
        a = m(x)[-1].clone()
        opt2 = Opt.build_opt('drn-l', 'carpet', 64, 2, pre_train=str(tmp_path / 'model' / 'model_best.pt'),
                             pre_train_dual=str(tmp_path / 'model' / 'dual_model_best.pt'))
        m2 = Model(opt2, None, dual_model=True).eval()
        assert torch.equal(m2(x)[-1], a)          # bit-reproducible: no atomics anywhere in the forward
    opt.model_name = 'nope'
    assert make_model(opt) is None


def test_training_mode_through_the_model_wrapper():
    """Model(opt) in train() mode: DRCT and DRN-L x4 return tensors with a graph and fill the flat gradient buffer;
    so does the DRN x8 preset (n_feats = 10, level 0 stored zero-padded)."""
    from srad_amd import options as Opt
    from srad_amd.model import Model
    opt = Opt.build_opt('drct', 'grid', 64, 4)
    opt.depths, opt.num_heads = (6,), (6,)
    opt.img_size, opt.window_size = 32, 8
    m = Model(opt, None)
    m.train()
    y = m(torch.rand(1, 1, 16, 16, device='cuda') * 255)
    assert y.requires_grad and tuple(y.shape) == (1, 1, 64, 64)
    y.mean().backward()
    assert float(m.model.flat_grads.abs().sum()) > 0
    opt = Opt.build_opt('drn-l', 'grid', 64, 4)
    opt.n_blocks = 2
    m = Model(opt, None)
    m.train()
    ys = m(torch.rand(1, 1, 16, 16, device='cuda') * 255)
    assert len(ys) == 3 and ys[-1].requires_grad
    (ys[-1].mean() + ys[0].mean()).backward()
    assert float(m.model.flat_grads.abs().sum()) > 0
    opt = Opt.build_opt('drn-l', 'grid', 64, 8)
    opt.n_blocks = 2
    m = Model(opt, None)
    m.train()
    ys = m(torch.rand(1, 1, 8, 8, device='cuda') * 255)
    assert len(ys) == 4 and ys[-1].requires_grad and tuple(ys[-1].shape) == (1, 1, 64, 64)
    (ys[-1].mean() + ys[1].mean()).backward()
    assert float(m.model.flat_grads.abs().sum()) > 0
