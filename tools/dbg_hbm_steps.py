"""Debug: the same host-batch epoch run several times, parameters recorded after every step; where two runs first part."""
import sys, pathlib, tempfile
R = pathlib.Path(__file__).resolve().parents[1]
for p in (R, R / "tests", R / "oracle"):
    sys.path.insert(0, str(p))
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
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 4
runs = []
import os
POISON = os.environ.get("POISON")
for n in range(NR):
    if POISON and n:
        junk = torch.empty(1 << 28, device="cuda").fill_(float(POISON))
        del junk
    model, _ = T._model()
    snaps, grads = [], []
    for b in host_batches:
        tr = Trainer(model, [b], host_batches[:1], cfgT, "cuda", engine="plan") if not snaps else tr
        tr.train_loader = [b]
        tr.train_epoch()
        snaps.append(model.flat_params.detach().cpu().clone())
        grads.append(model._gflat.detach().cpu().clone())
    runs.append((snaps, grads))
    offs = model._offsets
names = sorted(offs, key=lambda k: offs[k])
def name_of(i):
    last = names[0]
    for k in names:
        if offs[k] <= i:
            last = k
    return f"{last}[{i - offs[last]}]"
for r in range(1, NR):
    for s in range(len(host_batches)):
        d = (runs[0][0][s] - runs[r][0][s]).abs()
        gd = (runs[0][1][s] - runs[r][1][s]).abs()
        big = torch.nonzero(d > 1e-4).flatten().tolist()
        print(f"run0 vs run{r} step {s}: max dp {float(d.max()):.2e} n>1e-4 {len(big)} n>1e-6 {int((d > 1e-6).sum())}  max dg {float(gd.max()):.2e} |g| {float(runs[0][1][s].norm()):.3e}")
        for i in big[:6]:
            print("     ", name_of(i), "p", float(runs[0][0][s][i]), float(runs[r][0][s][i]), "g", float(runs[0][1][s][i]), float(runs[r][1][s][i]))
print("---- step-0 flips (param diff > 1e-4 vs run0) per run; per-tensor max gradient diff at step 1")
for r in range(1, NR):
    d = (runs[0][0][0] - runs[r][0][0]).abs()
    print(f"run{r}:", [name_of(i) for i in torch.nonzero(d > 1e-4).flatten().tolist()])
    gd = (runs[0][1][1] - runs[r][1][1]).abs()
    per = []
    for k in names:
        o = offs[k]
        nxt = min([offs[j] for j in names if offs[j] > o], default=gd.numel())
        per.append((float(gd[o:nxt].max()), k, int((gd[o:nxt] > 1e-6).sum()), nxt - o))
    per.sort(reverse=True)
    print("   ", [(f"{a:.1e}", k, n, m) for a, k, n, m in per[:8]])
