"""CPU: host-side logic of the drop-in modules (no compute: that needs the GPU and fails loudly without it)."""
import types

import pytest
import torch

from helpers import fusion_inputs, load


def _args(D=128, H=2, L=1, ffn=0, I=400):
    return types.SimpleNamespace(hidden_size=D, num_attention_heads=H, num_hidden_layers=L, use_intermediate=ffn, intermediate_size=I)


def test_state_dict_keys_and_shapes_match_reference():
    """SURVEY.md par. 8b: checkpoint keys/shapes, probed from the reference and stored in the golden files."""
    from snag_amd.fusion import MformerFusion
    for case, ffn, M in (("g1_fusion_m4_d128_h2_ffn.npz", 1, 4), ("g1_fusion_m3_d128_h2.npz", 0, 3)):
        _, params, cfg = fusion_inputs(load(case))
        sd = MformerFusion(_args(ffn=ffn), M).state_dict()
        assert set(sd) == set(params)
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(params[k].shape), k
    assert sum(p.numel() for p in MformerFusion(_args(), 3).parameters()) == 117894


def test_reference_checkpoint_loads():
    from snag_amd.fusion import MformerFusion
    _, params, cfg = fusion_inputs(load("g1_fusion_m3_d64_h4_l2.npz"))
    mod = MformerFusion(_args(D=64, H=4, L=2, ffn=1, I=96), 3)
    mod.load_state_dict(params, strict=True)
    assert torch.equal(mod.fusion_layer[1].attention.self.key.weight, params["fusion_layer.1.attention.self.key.weight"])
    assert torch.equal(mod.weight_raw, params["weight_raw"])


def test_default_init_matches_reference_conventions():
    from snag_amd.fusion import MformerFusion
    mod = MformerFusion(_args(), 3)
    assert torch.equal(mod.weight_raw, torch.ones(6))                                    # SNAG_tools.py:28
    ln = mod.fusion_layer[0].attention.output.LayerNorm
    assert ln.eps == 1e-12 and torch.equal(ln.weight, torch.ones(128)) and torch.equal(ln.bias, torch.zeros(128))
    w = mod.fusion_layer[0].attention.self.query.weight
    assert float(w.abs().max()) <= 1 / 128 ** 0.5 + 1e-6                                 # nn.Linear kaiming-uniform bound


def test_constructor_errors_mirror_reference():
    from snag_amd.fusion import MformerFusion
    with pytest.raises(AssertionError):                  # reference: assert hidden_size % heads == 0 (SNAG_tools.py:161)
        MformerFusion(_args(H=3), 3)
    with pytest.raises(ValueError):
        MformerFusion(_args(), 3, joint_way="nope")


def test_no_cpu_fallback():
    """Without a GPU every compute entry refuses loudly instead of silently running somewhere else."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from snag_amd import _lib, evalsim
    from snag_amd.fusion import MformerFusion
    from snag_amd.mkgc import MKGCConfig, MKGCHotPath
    mod = MformerFusion(_args(), 3)
    with pytest.raises(_lib.SnagHipError):
        mod([torch.zeros(4, 128)] * 3)
    with pytest.raises(_lib.SnagHipError):
        evalsim.pairwise_distances(torch.zeros(3, 4), torch.zeros(3, 4))
    with pytest.raises(_lib.SnagHipError):
        MKGCHotPath(MKGCConfig(n_ent=8, n_rel=2, feat_dims=[4, 4]), [torch.zeros(8, 4), torch.zeros(8, 4)])


def test_product_never_imports_oracle():
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "snag_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_stacked_qkv_view_is_free_when_adjacent():
    from snag_amd.fusion import _stacked
    flat = torch.arange(3 * 4 * 4, dtype=torch.float32).view(12, 4)
    q, k, v = flat[0:4], flat[4:8], flat[8:12]
    s = _stacked([q, k, v])
    assert s.data_ptr() == flat.data_ptr() and torch.equal(s, flat)
    s2 = _stacked([q.clone(), k.clone(), v.clone()])
    assert torch.equal(s2, flat)


def test_lr_schedules_match_transformers():
    """cosine_lr / linear_lr == transformers' get_*_schedule_with_warmup, which the reference's set_optim installs (src/utils.py:71-76)."""
    import torch
    from transformers import get_cosine_schedule_with_warmup, get_linear_schedule_with_warmup
    from snag_amd.mkgc import cosine_lr, linear_lr
    for make, mine in ((get_cosine_schedule_with_warmup, cosine_lr), (get_linear_schedule_with_warmup, linear_lr)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=5e-4)
        sch = make(opt, num_warmup_steps=7, num_training_steps=50)
        for step in range(55):
            assert abs(sch.get_last_lr()[0] - mine(step, 50, 7, 5e-4)) < 1e-12, (make.__name__, step)
            opt.step()
            sch.step()


def test_balanced_splits_of_grouped_long_k_launches():
    """MKGCHotPath._balanced_splits (host arithmetic only): the K splits the engine requests when two long-K products share one launch of
    256-row tiles -- all row tiles x splits fit one round of 256 workgroups, no split is finer than 64 k, and the workgroups' k-ranges are as
    even as the row-tile granularity allows."""
    from snag_amd.mkgc import MKGCHotPath
    bs = MKGCHotPath._balanced_splits
    assert bs([(15000, 4096), (15000, 768)]) == [3, 1]               # both first projections at the DB15K shape: 177 + 59 workgroups
    assert bs([(4096, 15000), (768, 15000)]) == [13, 13]             # their weight gradients: 208 + 39
    for shapes in ([(15000, 4096), (15000, 768)], [(4096, 15000), (768, 15000)], [(1875, 4096), (1875, 768)], [(4096, 1875), (768, 1875)],
                   [(3750, 2048), (3750, 2048)], [(1024, 512), (300, 640)], [(100000, 1024), (2048, 600)]):
        s = bs(shapes)
        tiles = [(m + 255) // 256 for m, _ in shapes]
        assert all(1 <= x <= max(1, k // 64) for x, (_, k) in zip(s, shapes)), (shapes, s)
        total = sum(t * x for t, x in zip(tiles, s))
        assert total <= 256 or all(x == 1 for x in s), (shapes, s, total)          # (more row tiles than workgroups: nothing to split)
        if total <= 256:     # no problem could take one more split without leaving the round
            i = max(range(len(s)), key=lambda j: ((shapes[j][1] + 15) // 16) / s[j])
            assert s[i] >= max(1, shapes[i][1] // 64) or total + tiles[i] > 256, (shapes, s)
