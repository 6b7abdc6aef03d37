"""Debug: two identical trainers, the second created after the allocator's free blocks were filled with a poison value; step 0 run
whole, step 1 run one plan op at a time on both with the outputs of every op compared."""
import sys, pathlib, tempfile, os
R = pathlib.Path(__file__).resolve().parents[1]
for p in (R, R / "tests", R / "oracle"):
    sys.path.insert(0, str(p))
import torch
import test_gpu_trainer as T
from lockstep import OUT_SLOTS
from fluidnet_amd.engine import runtime as rt
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
    host_batches.append(tuple(torch.stack([s[j] for s in samples]).cuda() for j in range(3)))
POISON = float(os.environ.get("POISON", "1e3"))
trs = []
for n in range(2):
    if n:
        junk = torch.empty(1 << 28, device="cuda").fill_(POISON)
        del junk
    model, _ = T._model()
    tr = Trainer(model, [host_batches[0]], host_batches[:1], cfgT, "cuda", engine="plan")
    tr.train_epoch()
    trs.append(tr)
torch.cuda.synchronize()
plans = [next(iter(tr._plans.values())) for tr in trs]
print("after step 0: max param diff", float((trs[0].model.flat_params - trs[1].model.flat_params).abs().max()),
      "grad diff", float((trs[0].model._gflat - trs[1].model._gflat).abs().max()))
rows = []
for b0, b1 in zip(plans[0].pb.bufs, plans[1].pb.bufs):
    if b0.tensor is None or b0.parent is not None:
        continue
    a, c = b0.tensor.double(), b1.tensor.double()
    bad = int((~torch.isfinite(c)).sum())
    rel = float((a - c).abs().max()) / (float(a.abs().max()) + 1e-30)
    rows.append((rel, b0.name, tuple(b0.shape), float(a.abs().max()), bad))
rows.sort(reverse=True)
for r in rows[:25]:
    print("buf after step 0:", r)
if os.environ.get("SYNC", "1") == "1":
    trs[1].model.flat_params.copy_(trs[0].model.flat_params)
    s0, s1 = trs[0].optimizer._state(), trs[1].optimizer._state()
    s1[2].copy_(s0[2]); s1[3].copy_(s0[3])
b = host_batches[1]
for tr, plan in zip(trs, plans):
    plan.x0.tensor.copy_(b[0]); plan.y.tensor.copy_(b[1]); plan.masks.tensor.copy_(b[2])
    tr.optimizer.step_count += 1
    plan.set_adam_scalars(tr.optimizer.param_groups[0]["lr"], tr.optimizer.step_count)
shown = 0
for sec in ("fwd", "fin", "bwd", "opt"):
    recs = [plans[0].sections[sec], plans[1].sections[sec]]
    for k in range(len(recs[0])):
        for plan in plans:
            rt.run_ops(plan._lowered[sec][k:k + 1])
        torch.cuda.synchronize()
        name = rt.OPNAME[recs[0][k].code]
        for j in OUT_SLOTS[name]:
            r0, r1 = recs[0][k], recs[1][k]
            if j >= len(r0.p) or r0.p[j] is None:
                continue
            a, c = r0.p[j].buf.tensor.double(), r1.p[j].buf.tensor.double()
            rel = float((a - c).abs().max()) / (float(a.abs().max()) + 1e-30)
            if rel > float(os.environ.get("THR", "1e-6")) and shown < 40:
                shown += 1
                print(f"{sec} op#{k} {name} [{r0.tag}] slot{j} buf={r0.p[j].buf.name} rel diff {rel:.3e} scale {float(a.abs().max()):.3e}")
print("done")
