"""FluidNPZSequenceDataset / augmentation mirror (fluidnet_amd/dataset) against the oracle's independent restatement of
ml/dataset/npz_sequence.py + augmentations.py on a synthetic seq_*.npz set.  The reference's dataset module itself is not
importable here (pydantic_settings is absent), so this row's parity is "unpinned" by the reference; what is checked is the
layout the reference's text prescribes, including its two mask conventions (on-disk: K+1 frames t..t+K, preloaded: K
frames t+1..t+K)."""
import numpy as np
import pytest
import torch

import fluidnet_oracle as orc
from fluidnet_amd.dataset.augmentations import apply_horizontal_flip, apply_rollout_augmentation
from fluidnet_amd.dataset.npz_sequence import FluidNPZSequenceDataset, _load_rollout_sample, _load_sample


@pytest.fixture()
def npz_root(tmp_path):
    rs = np.random.RandomState(0)
    seqs = []
    d = tmp_path / "npz" / "train"
    d.mkdir(parents=True)
    for i, T in enumerate((12, 9)):
        seq = {k: rs.rand(T, 16, 16).astype(np.float32) for k in ("density", "velx", "velz")}
        seq["emitter"] = (rs.rand(T, 16, 16) > 0.9).astype(np.float32)
        seq["collider"] = (rs.rand(T, 16, 16) > 0.9).astype(np.float32)
        np.savez_compressed(d / f"seq_{i:04d}.npz", **seq)
        seqs.append(seq)
    (tmp_path / "_field_stats.yaml").write_text("normalization_scales:\n  S_density: 2.0\n  S_velx: 0.5\n  S_velz: 4.0\n")
    return tmp_path / "npz", seqs


@pytest.mark.parametrize("K,stride", [(1, 1), (4, 2)])
@pytest.mark.parametrize("preload", [False, True])
@pytest.mark.parametrize("normalize", [False, True])
def test_samples_match_oracle(npz_root, K, stride, preload, normalize):
    root, seqs = npz_root
    ds = FluidNPZSequenceDataset(root, "train", normalize=normalize, preload=preload, rollout_steps=K, stride=stride)
    scales = {"S_density": 2.0, "S_velx": 0.5, "S_velz": 4.0} if normalize else None
    for epoch in range(stride):
        ds.set_epoch(epoch)
        want = orc.build_indices([12, 9], K, stride, epoch % stride)
        assert len(ds) == len(want)
        for i, (si, t) in enumerate(want):
            got = ds[i]
            exp = orc.make_sample(seqs[si], t, K, preload=preload, scales=scales)
            assert len(got) == len(exp) == (2 if K == 1 else 3)
            for a, b in zip(got, exp):
                assert a.dtype == torch.float32 and tuple(a.shape) == b.shape
                np.testing.assert_allclose(a.numpy(), b, rtol=1e-6, atol=0)
    if K > 1:
        assert ds[0][2].shape[0] == (K if preload else K + 1)


def test_module_level_loaders(npz_root):
    root, seqs = npz_root
    p = sorted((root / "train").iterdir())[0]
    x, y = _load_sample(p, 3, False, None)
    ex, ey = orc.make_sample(seqs[0], 3, 1, preload=False)
    np.testing.assert_array_equal(x.numpy(), ex)
    np.testing.assert_array_equal(y.numpy(), ey)
    x0, ys, mk = _load_rollout_sample(p, 2, 3, False, None)
    e0, eys, emk = orc.make_sample(seqs[0], 2, 3, preload=False)
    np.testing.assert_array_equal(mk.numpy(), emk)
    assert mk.shape[0] == 4


def test_flip_matches_oracle_and_is_involution(npz_root):
    root, seqs = npz_root
    x0, ys, mk = (torch.from_numpy(a) for a in orc.make_sample(seqs[0], 2, 3, preload=True))
    torch.manual_seed(0)
    fx, fy, fm = apply_rollout_augmentation(x0.clone(), ys.clone(), mk.clone(), flip_prob=1.0)
    ex, ey, em = orc.flip_sample(x0.numpy(), ys.numpy(), mk.numpy())
    np.testing.assert_array_equal(fx.numpy(), ex)
    np.testing.assert_array_equal(fy.numpy(), ey)
    np.testing.assert_array_equal(fm.numpy(), em)
    bx, by, bm = apply_rollout_augmentation(fx.clone(), fy.clone(), fm.clone(), flip_prob=1.0)
    assert torch.equal(bx, x0) and torch.equal(by, ys) and torch.equal(bm, mk)
    nx, ny, nm = apply_rollout_augmentation(x0.clone(), ys.clone(), mk.clone(), flip_prob=0.0)
    assert torch.equal(nx, x0)
    x, y = (torch.from_numpy(a) for a in orc.make_sample(seqs[0], 2, 1, preload=True))
    hx, hy = apply_horizontal_flip(x, y)
    e1, e2 = orc.flip_sample(x.numpy(), y.numpy())
    np.testing.assert_array_equal(hx.numpy(), e1)
    np.testing.assert_array_equal(hy.numpy(), e2)


def test_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        FluidNPZSequenceDataset(tmp_path, "train")
    (tmp_path / "train").mkdir()
    with pytest.raises(FileNotFoundError):
        FluidNPZSequenceDataset(tmp_path, "train")
    np.savez(tmp_path / "train" / "seq_0000.npz", density=np.zeros((2, 4, 4), np.float32), velx=np.zeros((2, 4, 4), np.float32),
             velz=np.zeros((2, 4, 4), np.float32))
    with pytest.raises(ValueError):
        FluidNPZSequenceDataset(tmp_path, "train")


def test_rollout_evaluation_starting_frames_rule():
    """ml/training/test_evaluation.py:173-189"""
    from fluidnet_amd.training.test_evaluation import _get_starting_frames
    assert _get_starting_frames(120) == [20, 60, 89]
    assert _get_starting_frames(40) == [9]                      # 20 + 30 does not fit (twice: the middle is 20 too), 9 + 30 < 40 does
    assert _get_starting_frames(16, steps=5, points=[3, "middle", -6]) == [3, 8, 10]
    assert _get_starting_frames(16, steps=5, points=[0, 11]) == []       # t >= 1 and t + steps < T


