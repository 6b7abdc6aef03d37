"""Debug: sensitivity of the step-1 gradient to +-lr moves of the zero-gradient biases (conv biases in front of InstanceNorm)."""
import sys, pathlib, tempfile, os
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
cfgT = T._config(tmp, int(os.environ.get("K", "2")))
ds = FluidNPZSequenceDataset(root, "train", rollout_steps=cfgT.rollout_step, preload=True, stride=1)
loader = HbmBatchLoader(HbmSequenceStore(ds), 2, shuffle=True, seed=3)
host_batches = []
for rows, _ in loader.batches():
    samples = [ds[int(i)] for i in rows]
    host_batches.append(tuple(torch.stack([s[j] for s in samples]).cuda() for j in range(3)))
model, _ = T._model()
tr = Trainer(model, [host_batches[0]], host_batches[:1], cfgT, "cuda", engine="plan")
tr.train_epoch()
plan = next(iter(tr._plans.values()))
b = host_batches[1]
plan.x0.tensor.copy_(b[0]); plan.y.tensor.copy_(b[1]); plan.masks.tensor.copy_(b[2])
offs = model._offsets
P, G = model.flat_params, model._gflat
base = P.clone()

def grad():
    plan.run(("fwd", "fin", "bwd"))
    torch.cuda.synchronize()
    return G.clone(), plan.loss_dict()

g0, d0 = grad()
g1, d1 = grad()
print("repeat:", float((g0 - g1).abs().max()), d0, d1)
from lockstep import OUT_SLOTS
from fluidnet_amd.engine import runtime as rt
k = "downs.1.down.bias"
trig = None
for e in range(32):
    for sgn in (1.0, -1.0):
        P.copy_(base)
        P[offs[k] + e] += sgn * 1e-3
        g, d = grad()
        dg = float((g - g0).abs().max())
        print(f"element {e} {sgn:+.0f}: max dg {dg:.2e}")
        if dg > 3e-5 and trig is None:
            trig = (e, sgn)
print("trigger", trig)
trig = trig or (0, 1.0)

def run_ops_collect():
    outs = []
    for sec in ("fwd", "fin", "bwd"):
        recs = plan.sections[sec]
        for j in range(len(recs)):
            rt.run_ops(plan._lowered[sec][j:j + 1])
            torch.cuda.synchronize()
            name = rt.OPNAME[recs[j].code]
            for sl in OUT_SLOTS[name]:
                if sl < len(recs[j].p) and recs[j].p[sl] is not None:
                    outs.append((sec, j, name, recs[j].tag, recs[j].p[sl].buf.name, recs[j].p[sl].buf.tensor.double().clone()))
    return outs
P.copy_(base)
A = run_ops_collect()
P.copy_(base)
P[offs[k] + trig[0]] += trig[1] * 1e-3
Bo = run_ops_collect()
for (sec, j, name, tag, bn, a), (_, _, _, _, _, c) in zip(A, Bo):
    if bn in ("grads_flat",):
        continue
    rel = float((a - c).abs().max()) / (float(a.abs().max()) + 1e-30)
    if rel > 0:
        print(f"{sec} op#{j} {name} [{tag}] {bn} rel {rel:.3e} scale {float(a.abs().max()):.3e}")
P.copy_(base)
for (sec, j, name, tag, bn, a), (_, _, _, _, _, c) in zip(A, Bo):
    if bn.endswith("/z"):
        for nm, t in (("base", a), ("perturbed", c)):
            d = torch.minimum(torch.minimum(t.abs(), (t - 1).abs()), (t + 1).abs())
            d = torch.where(t == 0, torch.ones_like(d), d)
            i = int(d.argmin())
            print(f"{bn} {nm}: shape {tuple(t.shape)} closest value to a clamp bound: z = {float(t.flatten()[i])!r} (distance {float(d.flatten()[i]):.3e}) at flat index {i}")
