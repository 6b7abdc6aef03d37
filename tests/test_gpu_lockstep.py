"""GPU parity, kernel by kernel: the full K-step training plan (weight re-layout, K forwards, losses, BPTT backward,
clip + Adam) is replayed op by op through the C ABI against the CPU twins.  Covers fp32 (reference kernels),
bf16 with reference convs, and bf16 with the MFMA conv/wgrad kernels."""
import numpy as np
import pytest
import torch

import fluidnet_oracle as orc
from fluidnet_amd.engine.train_plan import RolloutTrainPlan
from helpers import loss_c2, pack_params, tiny_cfg
from lockstep import lockstep

pytestmark = pytest.mark.gpu


def _build(cfg, lc, B, R, K, dtype, impl, seed=3, **kw):
    params = orc.make_params(cfg, seed=seed)
    x0, y, mk = orc.synth_batch(B, K, R, seed=50 + seed)
    plan = RolloutTrainPlan(cfg, lc, B, R, R, K, dtype=dtype, conv_impl=impl, clip_norm=0.05, seed=77, **kw)
    pb = plan.pb
    plan.bind("cpu", pack_params(pb, params), torch.zeros(pb.n_params_padded))
    plan.x0.tensor.copy_(torch.from_numpy(x0))
    plan.y.tensor.copy_(torch.from_numpy(y))
    plan.masks.tensor.copy_(torch.from_numpy(mk))
    plan.set_adam_scalars(3e-3, 1)
    recs = [r for s in ("fwd", "fin", "bwd", "opt") for r in plan.sections[s]]
    return plan, recs


CASES = [
    ("fp32-ref-replicate", dict(), "replicate", "fp32", "ref", 8),
    ("bf16-ref-replicate", dict(dropout=0.2), "replicate", "bf16", "ref", 8),
    ("bf16-mfma-replicate", dict(dropout=0.2), "replicate", "bf16", "mfma", 16),
    ("fp32-ref-reflect-central", dict(act="silu", upsample="nearest"), "reflect", "fp32", "ref", 8),
    ("bf16-mfma-circular", dict(act="relu", output_activation="sigmoid_tanh", use_residual=False), "circular", "bf16", "mfma", 16),
    ("bf16-mfma-zeros-depth3", dict(depth=3, act="leaky_relu"), "zeros", "bf16", "mfma", 16),
    ("fp32-ref-avgpool", dict(downsample="avgpool"), "replicate", "fp32", "ref", 8),
    ("bf16-mfma-maxpool-reflect", dict(downsample="maxpool", dropout=0.1), "reflect", "bf16", "mfma", 16),
    ("bf16-mfma-avgpool-depth3", dict(downsample="avgpool", depth=3), "zeros", "bf16", "mfma", 16),
    ("fp32-ref-batchnorm-small", dict(norm="batch", downsample="avgpool", upsample="nearest"), "zeros", "fp32", "ref", 8),
    ("bf16-mfma-batchnorm-small", dict(norm="batch", downsample="avgpool", upsample="nearest", dropout=0.1), "zeros", "bf16", "mfma", 16),
    ("fp32-ref-groupnorm", dict(norm="group", act="silu", use_residual=False), "zeros", "fp32", "ref", 8),
    ("bf16-mfma-groupnorm", dict(norm="group", act="silu", group_norm_groups=4), "replicate", "bf16", "mfma", 16),
    ("fp32-ref-transpose-up", dict(upsample="transpose"), "reflect", "fp32", "ref", 8),
    ("bf16-mfma-transpose-up", dict(upsample="transpose", dropout=0.1), "replicate", "bf16", "mfma", 16),
    ("bf16-mfma-nonorm", dict(norm="none", dropout=0.2), "replicate", "bf16", "mfma", 16),
    ("fp32-ref-base20-small-medium", dict(downsample="avgpool", dropout=0.1), "replicate", "fp32", "ref", 20),
    ("bf16-auto-base20-small-medium", dict(downsample="avgpool", depth=3), "replicate", "bf16", "auto", 20),
    ("bf16-auto-base12-groupnorm", dict(norm="group", group_norm_groups=3), "zeros", "bf16", "auto", 12),
]


@pytest.mark.parametrize("name,over,pm,dtype,impl,base", CASES, ids=[c[0] for c in CASES])
def test_lockstep_train_plan(name, over, pm, dtype, impl, base):
    cfg = tiny_cfg(base_channels=base, padding_mode=pm, **over)
    stencil = "central" if "central" in name else "forward"
    plan, recs = _build(cfg, loss_c2(32, padding_mode=pm, stencil_mode=stencil), 2, 32, 2, dtype, impl)
    failures = lockstep(plan.pb, recs)
    assert not failures, f"{len(failures)} of {len(recs)} ops differ:\n" + "\n".join(failures[:40])
