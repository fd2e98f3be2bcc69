"""The §8(f) remainders against the reference's own outputs (tests/golden/legacy_golden.npz, written by
tests/golden/make_legacy_golden.py from the imported reference):

  * ``prepare_mvtec_data`` (scripts/prepare_mvtec_data.py:22-33, 43-205): the prepared tree - file set, LANCZOS pixels, defect
    classes merged into ``bad`` with the class name as prefix, the train / val split rule;                                  [CPU]
  * the threshold finders of src/helpers.py:453-481 (and the roc_curve they stand on);                                      [CPU]
  * the legacy folder scorers ``analyze_window_sizes`` / ``process_images`` / ``calculate_*`` (src/helpers.py:107-134,
    158-230, 321-370) on the engine's scorer.                                                                               [GPU]"""
import os

import numpy as np
import pytest
from PIL import Image

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "legacy_golden.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


def _write_source_tree(g, root):
    for key in [str(k) for k in g["prep/in_names"]]:
        path = root / key
        path.parent.mkdir(parents=True, exist_ok=True)
        Image.fromarray(g["prep/in/" + key]).save(path)


def test_prepared_tree_equals_the_reference_scripts(g, tmp_path, capsys):
    from srad_amd import prepare_mvtec_data as P
    src, dst = tmp_path / "mvtec", tmp_path / "mvtec_32"
    _write_source_tree(g, src)
    (dst / "stale").mkdir(parents=True)                                  # an existing target tree is replaced
    P.prepare_mvtec_dataset(str(src), str(dst), scale_factors=(2, 4), target_hr=(32, 32), val_ratio=0.34, seed=42)
    assert not (dst / "stale").exists()
    mine = sorted(str(p.relative_to(dst)) for p in dst.rglob("*.png"))
    ref = [str(k) for k in g["prep/out_names"]]
    # which training images land in val depends on the order the file system lists them in (the reference shuffles the raw glob):
    # compare the split-independent structure first ...
    norm = lambda names: sorted(n.replace("/val/", "/train/") for n in names)
    assert norm(mine) == norm(ref)
    for cls in ("carpet", "grid"):
        assert len([n for n in mine if n.startswith(f"{cls}/val/good/HR/")]) == len(g[f"prep/val/{cls}"]) == 2
        bad = [os.path.basename(n) for n in mine if n.startswith(f"{cls}/test/bad/HR/")]
        assert sorted(bad) == ["bent_000.png", "crack_000.png", "crack_001.png"]
    # ... then every file's pixels (LANCZOS to 32 px as RGB, LR_2 / LR_4 from the HR image), wherever the split put it
    by_key = {n.replace("/val/", "/train/"): n for n in ref}
    for n in mine:
        want = g["prep/out/" + by_key[n.replace("/val/", "/train/")]]
        got = np.array(Image.open(dst / n))
        assert got.shape == want.shape and got.dtype == np.uint8 and np.array_equal(got, want), n
    assert np.array(Image.open(dst / "grid/train/good/LR_4" / sorted(os.listdir(dst / "grid/train/good/LR_4"))[0])).shape == (8, 8, 3)
    # and with the same listing order, the same split
    for cls in ("carpet", "grid"):
        listing = [str(x) for x in g[f"prep/listing/{cls}"]]
        _, val = P.split_train_val(listing, 0.34, 42)
        assert sorted(val) == [str(x) for x in g[f"prep/val/{cls}"]]
    counts = P.verify_dataset_structure(str(dst))
    assert counts["grid"]["test/bad/HR"] == 3 and counts["carpet"]["train/good/LR_2"] == 4 and counts["carpet"]["val/good/HR"] == 2


def test_split_rule_and_progressive_scales(g):
    from srad_amd import prepare_mvtec_data as P
    for tag in "abcde":
        n, ratio, seed = g[f"prep/split/{tag}/args"]
        train, val = P.split_train_val([f"{i:02d}.png" for i in range(int(n))], float(ratio), int(seed))
        assert val == [str(x) for x in g[f"prep/split/{tag}/val"]] and train == [str(x) for x in g[f"prep/split/{tag}/train"]]
    assert P.progressive_scales([4]) == (2, 4) and P.progressive_scales([8]) == (2, 4, 8) and P.progressive_scales([4, 8]) == (2, 4, 8)
    with pytest.raises(ValueError):
        P.progressive_scales([3])


def test_cli_reports_a_missing_source(tmp_path, capsys):
    from srad_amd import prepare_mvtec_data as P
    assert P.main(["--hr-size", "32", "--source", str(tmp_path / "nothing")]) == 1
    assert "Source data not found" in capsys.readouterr().out


def test_threshold_finders_match_reference(g):
    from srad_amd import helpers as H
    for k in ("proc_ssim", "proc_mse", "ties", "random"):
        y, s, want = g[f"legacy/thr/{k}/y"], g[f"legacy/thr/{k}/s"], g[f"legacy/thr/{k}/out"]
        got = [H.find_optimal_threshold_YoudenJ(y, s), H.find_optimal_threshold(y, s), H.find_threshold_for_perfect_recall(y, s)]
        assert np.allclose(got, want, rtol=0, atol=1e-12), (k, got, want)


def _write_folders(g, root):
    for label in ("good", "bad"):
        for kind in ("orig", "rec"):
            d = root / f"{label}_{kind}"
            d.mkdir(parents=True)
            for key in [k for k in g.files if k.startswith(f"legacy/{label}_{kind}/")]:
                Image.fromarray(g[key]).save(d / os.path.basename(key))
    return [str(root / "good_orig"), str(root / "good_rec"), str(root / "bad_orig"), str(root / "bad_rec")]


@pytest.mark.gpu
def test_legacy_folder_scorers_match_reference(g, tmp_path):
    from srad_amd import helpers as H
    args = _write_folders(g, tmp_path)
    aw = H.analyze_window_sizes(*args, min_size=3, max_size=None, step=10)
    assert aw["window_sizes"] == [int(x) for x in g["legacy/analyze/window_sizes"]]
    for k in ("avg_good_scores", "avg_bad_scores", "score_differences", "max_difference"):
        assert np.allclose(aw[k], g["legacy/analyze/" + k], rtol=0, atol=2e-6), k
    assert np.allclose(aw["auc_scores"], g["legacy/analyze/auc_scores"], atol=1e-12) and abs(aw["max_auc"] - float(g["legacy/analyze/max_auc"])) < 1e-12
    assert aw["best_window_size"] == int(g["legacy/analyze/best_window_size"]) and aw["best_auc_window_size"] == int(g["legacy/analyze/best_auc_window_size"])
    y, s1, s2, s3 = H.process_images(*args, str(tmp_path / "log.txt"), 11)
    # the lists follow each folder's listing order, which is the file system's: compare by file name
    def by_name(values, golden_values):
        mine = dict(zip(os.listdir(args[0]) + os.listdir(args[2]), values))
        ref = dict(zip([str(x) for x in g["legacy/listing/good"]] + [str(x) for x in g["legacy/listing/bad"]], golden_values))
        assert sorted(mine) == sorted(ref)
        return np.array([mine[k] for k in sorted(mine)]), np.array([ref[k] for k in sorted(ref)])
    assert y == [0] * 4 + [1] * 5
    a, b = by_name(s1, g["legacy/process/ssim"]); assert np.allclose(a, b, rtol=0, atol=2e-6)
    a, b = by_name(s2, g["legacy/process/mse"]); assert np.allclose(a, b, rtol=2e-6)
    a, b = by_name(s3, g["legacy/process/psnr"]); assert np.allclose(a, b, rtol=0, atol=2e-5)
    o, r = g["legacy/bad_orig/bad0.png"], g["legacy/bad_rec/bad0.png"]
    got = [H.calculate_ssim(o, r, 7), H.calculate_mse(o, r), H.calculate_psnr(o, r), H.calculate_ssim(o[:, :, 0], r[:, :, 0], 5)]
    assert np.allclose(got, g["legacy/single"], rtol=2e-6, atol=2e-6), (got, g["legacy/single"])
