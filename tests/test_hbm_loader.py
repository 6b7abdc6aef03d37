"""HBM-resident dataset: host-side batching logic on CPU (sharding, determinism, epoch offsets) and, on the GPU, the
GATHER_BATCH kernel against FluidNPZSequenceDataset.__getitem__ (bit-exact: it only moves, mirrors and negates fp32 values)."""
import numpy as np
import pytest
import torch

from fluidnet_amd.dataset import FluidNPZSequenceDataset
from fluidnet_amd.dataset.augmentations import apply_horizontal_flip
from fluidnet_amd.dataset.hbm_resident import HbmBatchLoader, HbmSequenceStore


@pytest.fixture()
def npz_root(tmp_path):
    rs = np.random.RandomState(3)
    d = tmp_path / "npz" / "train"
    d.mkdir(parents=True)
    for i, T in enumerate((14, 9, 11)):
        seq = {k: rs.randn(T, 12, 20).astype(np.float32) for k in ("density", "velx", "velz")}
        seq["emitter"] = (rs.rand(T, 12, 20) > 0.9).astype(np.float32)
        seq["collider"] = (rs.rand(T, 12, 20) > 0.9).astype(np.float32)
        np.savez_compressed(d / f"seq_{i:04d}.npz", **seq)
    (tmp_path / "_field_stats.yaml").write_text("normalization_scales:\n  S_density: 2.0\n  S_velx: 0.5\n  S_velz: 4.0\n")
    return tmp_path / "npz"


class _FakeStore:          # batches() only needs the dataset: lets the sharding logic run without a GPU
    def __init__(self, ds):
        self.dataset = ds
        self.frames = torch.zeros(1)


def test_batches_shard_disjointly_and_deterministically(npz_root):
    ds = FluidNPZSequenceDataset(npz_root, "train", rollout_steps=2, stride=2, is_training=True,
                                 augmentation_config={"enable_augmentation": True, "flip_probability": 0.5})
    loaders = [HbmBatchLoader(_FakeStore(ds), 2, seed=5, rank=r, world=2) for r in range(2)]
    for epoch in (0, 1):
        for ld in loaders:
            ld.set_epoch(epoch)
        b0, b1 = loaders[0].batches(), loaders[1].batches()
        assert len(b0) == len(b1) == len(ds) // 4 == len(loaders[0])
        seen = np.concatenate([r for r, _ in b0] + [r for r, _ in b1])
        assert len(set(seen.tolist())) == len(seen) and seen.max() < len(ds)          # no sample twice in an epoch, across ranks
        again = loaders[0].batches()
        assert all((a[0] == b[0]).all() and (a[1] == b[1]).all() for a, b in zip(b0, again))
    loaders[0].set_epoch(0); e0 = loaders[0].batches()
    loaders[0].set_epoch(2); e2 = loaders[0].batches()
    assert any((a[0] != b[0]).any() for a, b in zip(e0, e2))                          # reshuffled per epoch
    flips = np.concatenate([f for _, f in e0])
    assert set(flips.tolist()) <= {0, 1}
    noaug = HbmBatchLoader(_FakeStore(FluidNPZSequenceDataset(npz_root, "train", rollout_steps=2)), 2)
    assert all(f.sum() == 0 for _, f in noaug.batches())
    with pytest.raises(ValueError):
        HbmBatchLoader(_FakeStore(ds), 2, mask_frames=5)


def test_store_requires_gpu(npz_root):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HbmSequenceStore(FluidNPZSequenceDataset(npz_root, "train"), device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("K,normalize,preload", [(1, False, True), (3, True, True), (3, True, False)])
def test_gather_matches_dataset_getitem(npz_root, K, normalize, preload):
    ds = FluidNPZSequenceDataset(npz_root, "train", normalize=normalize, preload=preload, rollout_steps=K, stride=2)
    store = HbmSequenceStore(ds)
    Km = K if (preload or K == 1) else K + 1                  # the reference's two mask conventions (preloaded / on-disk)
    ld = HbmBatchLoader(store, 4, shuffle=True, seed=1, mask_frames=Km)
    for epoch in (0, 1):
        ld.set_epoch(epoch)
        for rows, _ in ld.batches():
            flips = np.array([0, 1, 0, 1], dtype=np.int32)
            x0 = torch.empty(4, 6, 12, 20, device="cuda")
            y = torch.empty(4, K, 3, 12, 20, device="cuda")
            mk = torch.empty(4, Km, 2, 12, 20, device="cuda")
            ld.gather(rows, flips, x0, y, mk)
            for j, i in enumerate(rows):
                s = ds[int(i)]
                if K == 1:
                    ex, ey = s
                    if flips[j]:
                        ex, ey = apply_horizontal_flip(ex, ey)
                    assert torch.equal(x0[j].cpu(), ex) and torch.equal(y[j, 0].cpu(), ey)
                else:
                    ex, ey, em = s
                    if flips[j]:
                        ex, ey, em = torch.flip(ex, dims=[2]), torch.flip(ey, dims=[3]), torch.flip(em, dims=[3])
                        ex[1] = -ex[1]
                        ey[:, 1] = -ey[:, 1]
                    assert torch.equal(x0[j].cpu(), ex) and torch.equal(y[j].cpu(), ey) and torch.equal(mk[j].cpu(), em)
    # DataLoader-style iteration yields the dataset's tuple shapes
    b = next(iter(ld))
    assert len(b) == (2 if K == 1 else 3) and b[0].shape == (4, 6, 12, 20)


@pytest.mark.gpu
@pytest.mark.parametrize("K,preload", [(1, True), (4, True), (4, False)])
def test_gather_matches_reference_fixture_g6(golden_dir, tmp_path, K, preload):
    """GATHER_BATCH against what the REAL ml/dataset/npz_sequence.py returned for the same files (tests/golden/make_golden.py g6):
    both mask conventions, bit for bit, and the flipped tuples of ml/dataset/augmentations.py."""
    import os

    import fixture_inputs as fi
    g = np.load(os.path.join(golden_dir, "g6_dataset.npz"))
    root = fi.write_synth_npz(tmp_path)
    ds = FluidNPZSequenceDataset(root, "train", rollout_steps=K, preload=preload, stride=1)
    Km = K if (preload or K == 1) else K + 1
    ld = HbmBatchLoader(HbmSequenceStore(ds), 3, shuffle=False, mask_frames=Km)
    pre = f"K{K}/pre{int(preload)}/s1"
    rows = np.array([0, len(ds) // 2, len(ds) - 1])
    x0 = torch.empty(3, 6, 16, 20, device="cuda")
    y = torch.empty(3, K, 3, 16, 20, device="cuda")
    mk = torch.empty(3, Km, 2, 16, 20, device="cuda")
    ld.gather(rows, np.zeros(3, np.int32), x0, y, mk)
    for j, idx in enumerate(rows):
        assert np.array_equal(x0[j].cpu().numpy(), g[f"{pre}/item{idx}/0"])
        if K == 1:
            assert np.array_equal(y[j, 0].cpu().numpy(), g[f"{pre}/item{idx}/1"])
            # single-step batches: the model input and the loss masks are those of frame t (npz_sequence.py:144-153, trainer.py:187-188)
            assert np.array_equal(mk[j, 0].cpu().numpy(), g[f"{pre}/item{idx}/0"][4:6])
        else:
            assert np.array_equal(y[j].cpu().numpy(), g[f"{pre}/item{idx}/1"])
            assert np.array_equal(mk[j].cpu().numpy(), g[f"{pre}/item{idx}/2"])
    if K == 4 and not preload:          # the flipped rollout tuple of the fixture is item 2 of this very dataset
        ld.gather(np.array([2, 2, 2]), np.ones(3, np.int32), x0, y, mk)
        for j, t in enumerate((x0, y, mk)):
            assert np.array_equal(t[0].cpu().numpy(), g[f"flip_on/rollout/{j}"])


@pytest.mark.gpu
def test_epoch_table_gather_equals_per_batch_gather(npz_root):
    """upload_epoch() + gather_uploaded(b): the whole epoch's indices / flip flags in one upload must form the same batches, bit for bit,
    as gather(rows, flips) with per-batch uploads; a stale table (another epoch) is refused."""
    ds = FluidNPZSequenceDataset(npz_root, "train", rollout_steps=3, preload=True, stride=1, is_training=True,
                                 augmentation_config={"enable_augmentation": True, "flip_probability": 0.5})
    ld = HbmBatchLoader(HbmSequenceStore(ds), 4, shuffle=True, seed=9)
    bufs = lambda: (torch.empty(4, 6, 12, 20, device="cuda"), torch.empty(4, 3, 3, 12, 20, device="cuda"), torch.empty(4, 3, 2, 12, 20, device="cuda"))  # noqa: E731
    for epoch in (0, 3):
        ld.set_epoch(epoch)
        bs = ld.batches()
        assert ld.upload_epoch() == len(bs) > 1 and any(f.sum() for _, f in bs)
        for b, (rows, flips) in enumerate(bs):
            a, c = bufs(), bufs()
            ld.gather(rows, flips, *a)
            ld.gather_uploaded(b, *c)
            assert all(torch.equal(p, q) for p, q in zip(a, c))
    ld.set_epoch(4)
    with pytest.raises(RuntimeError, match="upload_epoch"):
        ld.gather_uploaded(0, *bufs())
