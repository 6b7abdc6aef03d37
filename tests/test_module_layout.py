"""Host-side checks of the nn.Module mirror that need no GPU: state-dict keys of every supported option (they must be the reference's:
ml/models/unet.py:204-279), the flat parameter / norm-state layouts behind them, and what check_supported accepts."""
import numpy as np
import pytest
import torch

import fluidnet_oracle as orc
from fluidnet_amd.engine.plan import check_supported, cpad, kpad, norm_state_layout, param_layout
from fluidnet_amd.models.unet import UNet, UNetConfig
from helpers import tiny_cfg


@pytest.mark.parametrize("over", [
    dict(), dict(norm="batch", downsample="avgpool"), dict(norm="group", upsample="transpose"), dict(norm="none", downsample="maxpool"),
    dict(base_channels=20, depth=3), dict(use_residual=False, bottleneck_blocks=0, depth=1),
])
def test_state_dict_keys_and_flat_views(over):
    cfg = tiny_cfg(**over)
    m = UNet(UNetConfig(**cfg.__dict__), precision="fp32")
    want = orc.param_shapes(cfg)
    got = {k: tuple(v.shape) for k, v in m.named_parameters()}
    assert list(got) == list(want) and all(got[k] == tuple(want[k]) for k in want)
    sd = m.state_dict()
    bn = [k for k in sd if "running_" in k or k.endswith("num_batches_tracked")]
    assert bool(bn) == (cfg.norm == "batch")
    # flat layout: every tensor 32-byte aligned, per-channel vectors with room for the padded channel count, nothing overlaps
    shapes, offs, total = param_layout(cfg)
    end = 0
    for k, s in shapes.items():
        assert offs[k] % 8 == 0 and offs[k] >= end
        end = offs[k] + (cpad(s[0]) if len(s) == 1 else int(np.prod(s)))
    assert total >= end
    # parameters become views of the flat buffer, and loading a state dict writes through them
    flat = m.flat_params
    p = orc.make_params(cfg, seed=2)
    m.load_state_dict(p, strict=cfg.norm != "batch")
    for k, v in p.items():
        assert torch.equal(flat[offs[k]: offs[k] + v.numel()].view(v.shape), v), k
    gaps = torch.ones(total, dtype=torch.bool)
    for k, s in shapes.items():
        gaps[offs[k]: offs[k] + int(np.prod(s))] = False
    assert float(flat[gaps].abs().max() if gaps.any() else 0.0) == 0.0          # the zero padding a padded layer reads as gamma / beta / bias
    if cfg.norm == "batch":
        noffs, n = norm_state_layout(cfg)
        ns = m.flat_norm_state
        assert ns.numel() == max(n, 8)
        for k, o in noffs.items():
            mod, leaf = k.rsplit(".", 1)
            buf = getattr(m.get_submodule(mod), leaf)
            assert buf.data_ptr() == ns.data_ptr() + 4 * o
            assert float(buf[0]) == (1.0 if leaf == "running_var" else 0.0)
        m.count_batches(3)
        assert int(m.state_dict()["downs.0.block.n1.num_batches_tracked"]) == 3


def test_channel_padding_rules():
    assert [cpad(c) for c in (8, 16, 20, 24, 40, 48)] == [8, 16, 24, 24, 40, 48]
    assert [kpad(c) for c in (8, 16, 24, 40, 64)] == [16, 16, 32, 48, 64]


def test_check_supported_matrix():
    ok = UNetConfig(in_channels=6, base_channels=32, depth=3, norm="instance", act="gelu", upsample="bilinear", padding_mode="replicate", use_residual=True)
    check_supported(ok, 128, 128)
    check_supported(UNetConfig(), 32, 32)                                      # the reference's own defaults
    for bad, hw in ((dict(base_channels=18), (32, 32)), (dict(base_channels=96), (32, 32)), (dict(in_channels=17), (32, 32)),
                    (dict(out_channels=5), (32, 32)), (dict(depth=3), (36, 36)), (dict(depth=4), (16, 16))):
        with pytest.raises(NotImplementedError):
            check_supported(UNetConfig(**{**ok.__dict__, **bad}), *hw)
    with pytest.raises(ValueError, match="depth must be >= 1"):
        check_supported(UNetConfig(**{**ok.__dict__, "depth": 0}), 32, 32)
