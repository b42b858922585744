"""GPU: seeded random sweep of the Mformer fusion (modalities, hidden size, heads, layers, FFN on/off, None slots, explicit dropout
masks) -- all four outputs and every gradient against the oracle's autograd."""
import random
import types

import pytest
import torch

from helpers import assert_close
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("seed", range(10))
def test_fusion_random_configs(seed):
    _run(seed)


@pytest.mark.parametrize("train", [True, False])
def test_fusion_reference_default_heads_with_all_six_modalities(train):
    """config.py:108 defaults num_attention_heads to 5; with --use_surface all six modalities are stacked: H*M*M = 180 attention
    entries per entity (the kernels keep them in three registers per lane; up to 192)."""
    _run(77, forced=dict(H=5, D=300, M=6, L=1, ffn=True, I=400, N=63, train=train))


def _run(seed, forced=None):
    from snag_amd.fusion import MformerFusion
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(1000 + seed)
    H = rng.choice([1, 2, 4])
    D = H * rng.choice([8, 13, 25, 32, 64, 75])            # includes widths that are not multiples of 4 / 64 (D = 13, 26, 52, 75, 150, 300)
    M = rng.randint(2, 6)
    L = rng.choice([1, 1, 2])
    ffn = rng.random() < 0.5
    I = rng.choice([24, 50, 96])
    N = rng.choice([1, 5, 63, 257])
    train = rng.random() < 0.6
    if forced:
        H, D, M, L, ffn, I, N, train = (forced[k] for k in ("H", "D", "M", "L", "ffn", "I", "N", "train"))
    args = types.SimpleNamespace(hidden_size=D, num_attention_heads=H, num_hidden_layers=L, use_intermediate=int(ffn), intermediate_size=I)
    mod = MformerFusion(args, modal_num=M)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if "LayerNorm.weight" in n:
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5)
            else:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
    slots = sorted(rng.sample(range(6), M))
    embs_cpu = [torch.randn(N, D, generator=g) * rng.choice([0.3, 1.0, 3.0]) if i in slots else None for i in range(6)]
    masks = None
    if train:
        masks = []
        for _ in range(L):
            m = {"probs": torch.rand(N, H, M, M, generator=g) >= 0.1, "attn_out": torch.rand(N, M, D, generator=g) >= 0.1}
            if ffn:
                m["ffn_out"] = torch.rand(N, M, D, generator=g) >= 0.1
            masks.append(m)
    cw = [torch.randn(N, M * D, generator=g), torch.randn(N, M * D, generator=g), torch.randn(N, M, D, generator=g), torch.randn(N, M, generator=g)]
    # oracle
    params = {k: v.detach().clone().requires_grad_(True) for k, v in mod.state_dict().items()}
    e_ref = [e.clone().requires_grad_(True) if e is not None else None for e in embs_cpu]
    outs_ref = O.fusion_forward(e_ref, params, H, L, ffn, drop_masks=masks)
    sum((o * c).sum() for o, c in zip(outs_ref, cw)).backward()
    # HIP
    mod = mod.to(DEV)
    mod.train(train)
    e_gpu = [e.to(DEV).requires_grad_(True) if e is not None else None for e in embs_cpu]
    dmasks = [{k: v.to(torch.uint8).to(DEV) for k, v in m.items()} for m in masks] if masks else None
    outs = mod(e_gpu, drop_masks=dmasks)
    sum((o * c.to(DEV)).sum() for o, c in zip(outs, cw)).backward()
    tag = f"seed={seed} N={N} M={M} D={D} H={H} L={L} ffn={ffn} I={I} train={train}"
    for nm, a, b in zip(("joint", "joint_fz", "hidden", "weight_norm"), outs, outs_ref):
        assert_close(a, b.detach(), atol=3e-5, rtol=1e-5, what=f"{tag} {nm}")
    for i, (a, b) in enumerate(zip(e_gpu, e_ref)):
        if a is not None:
            assert_close(a.grad, b.grad, atol=2e-4 * max(1.0, float(b.grad.abs().max())), rtol=1e-4, what=f"{tag} d emb{i}")
    for k, p in mod.named_parameters():
        ref = params[k].grad
        ref = ref if ref is not None else torch.zeros_like(params[k])
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        assert_close(got, ref, atol=2e-4 * max(1.0, float(ref.abs().max())), rtol=1e-4, what=f"{tag} d {k}")
