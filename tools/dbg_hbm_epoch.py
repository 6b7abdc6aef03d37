"""Debug: the HBM-loader epoch vs the host-batch epoch, each run twice in one process; pairwise parameter differences."""
import sys, pathlib, tempfile, itertools
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / "tests"))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / "oracle"))
import torch
import test_gpu_trainer as T
from fluidnet_amd.dataset.npz_sequence import FluidNPZSequenceDataset
from fluidnet_amd.dataset.hbm_resident import HbmBatchLoader, HbmSequenceStore
from fluidnet_amd.training.trainer import Trainer

tmp = pathlib.Path(tempfile.mkdtemp())
root = T._npz(tmp)
cfgT = T._config(tmp, 2)
ds = FluidNPZSequenceDataset(root, "train", rollout_steps=2, preload=True, stride=1)
loader = HbmBatchLoader(HbmSequenceStore(ds), 2, shuffle=True, seed=3)
host_batches = []
for rows, _ in loader.batches():
    samples = [ds[int(i)] for i in rows]
    host_batches.append(tuple(torch.stack([s[j] for s in samples]) for j in range(3)))
out = {}
order = sys.argv[1].split(",") if len(sys.argv) > 1 else ["hbm", "host", "hbm", "host"]
for n, kind in enumerate(order):
    model, _ = T._model()
    feed = loader if kind == "hbm" else host_batches
    tr = Trainer(model, feed, host_batches[:1], cfgT, "cuda", engine="plan")
    d = tr.train_epoch()
    out[f"{kind}{n}"] = (d, model.flat_params.detach().cpu().clone())
    print(kind, n, {k: round(v, 7) for k, v in d.items()}, flush=True)
for a, b in itertools.combinations(out, 2):
    diff = (out[a][1] - out[b][1]).abs()
    print(f"{a} vs {b}: max {float(diff.max()):.2e} frac>1e-4 {float((diff > 1e-4).float().mean()):.3f} frac>1e-6 {float((diff > 1e-6).float().mean()):.3f}")
