"""GPU parity: noise injection (golden G2), projections (golden G6) and the encoder slice (golden G7)."""
import types

import numpy as np
import pytest
import torch

from helpers import T, assert_close, load
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_noise_golden_replays_reference_rng():
    from snag_amd import noise, ops
    z = load("g2_noise.npz")
    eps, rho = float(z["mask_ratio"]), float(z["noise_ratio"])
    img, rel, att = (T(z[k], DEV) for k in ("img", "rel", "att"))
    m, s = noise.mean_std(img, z["ent_wo_img"].tolist())
    assert_close(m, z["img_mean"], atol=1e-6), assert_close(s, z["img_std"], atol=1e-6)
    for nm, x in (("rel", rel), ("att", att)):
        m, s = noise.mean_std(x)
        assert_close(m, z[f"{nm}_mean"], atol=1e-5, rtol=1e-5), assert_close(s, z[f"{nm}_std"], atol=1e-5, rtol=1e-5)
    # explicit (mask, z) from the reference's RNG stream -> bit-comparable blend (1e-6: fma contraction differs)
    for nm, x in (("rel", rel), ("att", att), ("img", img)):
        mask = torch.from_numpy(z[f"{nm}_mask"])
        sel = torch.full((x.shape[0],), -1, dtype=torch.int32)
        sel[mask] = torch.arange(int(mask.sum()), dtype=torch.int32)
        out = ops.noise_apply(x, T(z[f"{nm}_mean"], DEV), T(z[f"{nm}_std"], DEV), mask.to(torch.uint8).to(DEV), eps,
                              z=T(z[f"{nm}_z"], DEV), sel_index=sel.to(DEV))
        assert_close(out, z[f"{nm}_noisy"], atol=1e-6, what=nm)
        assert torch.equal(out.cpu()[~mask], torch.from_numpy(z[nm])[~mask])          # untouched rows are bit-identical
    # the whole update_noise() call order, seeded like the reference run
    inj = noise.NoiseInjector({"rel": rel, "att": att, "img": img}, rho, eps, exclude_rows={"img": z["ent_wo_img"].tolist()}, mode="parity")
    torch.manual_seed(int(z["seed"]))
    noisy = inj.update_noise(entity_weight=T(z["ent_w"], DEV))
    for nm in ("rel", "att", "img"):
        assert_close(noisy[nm], z[f"{nm}_noisy"], atol=2e-6, what=f"update_noise {nm}")
    assert_close(inj.entity_noise, z["entity_noise"], atol=1e-6)
    assert np.array_equal(inj.entity_noise_mask.cpu().numpy().astype(bool), z["entity_noise_mask"])


def test_noise_philox_statistics():
    """Perf mode has no reference stream to match: check the distributional contract instead."""
    from snag_amd import noise, ops
    g = torch.Generator().manual_seed(1)
    N, F = 20000, 64
    x = (torch.randn(N, F, generator=g) * 2 + 1).to(DEV)
    m, s = noise.mean_std(x)
    out = noise.add_noise_to_embeddings(x, m, s, 0.2, 0.7, mode="philox", seed=5)
    changed = (out != x).any(dim=1)
    frac = float(changed.float().mean())
    assert abs(frac - 0.2) < 0.015, frac                                   # Bernoulli(0.2) rows
    zrec = ((out[changed] - 0.3 * x[changed]) / 0.7 - m) / s                # recover z
    assert abs(float(zrec.mean())) < 0.01 and abs(float(zrec.std()) - 1.0) < 0.01
    out2 = noise.add_noise_to_embeddings(x, m, s, 0.2, 0.7, mode="philox", seed=5)
    assert torch.equal(out, out2)                                          # counter-based: same seed, same table
    keep = ops.dropout_keep_mask((1 << 20,), 0.1, 3)
    assert abs(float(keep.float().mean()) - 0.9) < 2e-3


def test_projections_golden():
    from snag_amd.projection import Linear, ProjectionHead
    z = load("g6_proj.npz")
    x = T(z["x"], DEV)
    lin = Linear(x.shape[1], 128).to(DEV)
    lin.load_state_dict({"weight": T(z["lin.weight"]), "bias": T(z["lin.bias"])})
    y = lin(x)
    assert_close(y, z["lin.y"], atol=2e-5)
    (y * T(z["cw"], DEV)).sum().backward()
    assert_close(lin.weight.grad, z["lin.g.weight"], atol=1e-4, rtol=1e-4)
    assert_close(lin.bias.grad, z["lin.g.bias"], atol=1e-4, rtol=1e-4)
    ph = ProjectionHead(x.shape[1], 96, 128, dropout=float(z["ph.p"])).to(DEV)
    ph.load_state_dict({"l1.weight": T(z["ph.l1"]), "l2.weight": T(z["ph.l2"])})
    for mode in ("eval", "train"):
        ph.train(mode == "train")
        ph.zero_grad()
        xr = x.clone().requires_grad_(True)
        y = ph(xr, keep_mask=T(z["ph.mask"], DEV) if mode == "train" else None)
        assert_close(y, z[f"ph.{mode}.y"], atol=2e-5, what=mode)
        (y * T(z["cw"], DEV)).sum().backward()
        assert_close(ph.l1.weight.grad, z[f"ph.{mode}.g.l1"], atol=1e-4, rtol=1e-4)
        assert_close(ph.l2.weight.grad, z[f"ph.{mode}.g.l2"], atol=1e-4, rtol=1e-4)
        assert_close(xr.grad, z[f"ph.{mode}.g.x"], atol=1e-4, rtol=1e-4)
    assert ph(None) is None


@pytest.mark.parametrize("tag", ["clean", "noisy"])
def test_encoder_slice_golden(tag):
    """MultiModalEncoder.forward (SNAG_tools.py:122-156, GAT stubbed): projections + entity-noise blend + fusion."""
    from snag_amd import ops
    from snag_amd.fusion import MformerFusion
    from snag_amd.projection import linear
    z = load("g7_encoder.npz")
    P = {k[2:]: T(v, DEV) for k, v in z.items() if k.startswith("p.")}
    ent = P["entity_emb.weight"]
    img = linear(T(z["img"], DEV), P["img_fc.weight"], P["img_fc.bias"])
    rel = linear(T(z["rel"], DEV), P["rel_fc.weight"], P["rel_fc.bias"])
    att = linear(T(z["att"], DEV), P["att_fc.weight"], P["att_fc.bias"])
    for nm, v in (("img", img), ("rel", rel), ("att", att)):
        assert_close(v, z[f"{tag}.{nm}"], atol=2e-5, what=nm)
    noise = T(z["entity_noise"], DEV) if tag == "noisy" else None
    nmask = T(z["entity_noise_mask"].astype(np.uint8), DEV) if tag == "noisy" else None
    x0 = ops.pack_tokens([img, att, rel, ent], g_slot=3, noise=noise, noise_mask=nmask, mask_ratio=float(z["mask_ratio"]))
    assert_close(x0[:, 3], z[f"{tag}.gph"], atol=1e-6, what="gph token")
    args = types.SimpleNamespace(hidden_size=128, num_attention_heads=2, num_hidden_layers=1, use_intermediate=0, intermediate_size=400)
    fus = MformerFusion(args, 4).to(DEV).eval()
    fus.load_state_dict({k[len("fusion."):]: v for k, v in P.items() if k.startswith("fusion.")})
    outs = fus.forward_stacked(x0)
    for nm, o in zip(("joint", "joint_fz", "hidden", "weight_norm"), outs):
        assert_close(o, z[f"{tag}.{nm}"], atol=2e-5, what=nm)


@pytest.mark.parametrize("R,C,pad", [(1, 1, 0), (63, 65, 0), (700, 37, 5), (15000, 768, 8), (129, 4096, 0)])
def test_transpose_kernel(R, C, pad):
    """snag_transpose_f32 (the per-epoch refresh of the transposed feature copies) is an exact copy, strides honoured."""
    from snag_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(R + C)
    src = torch.randn(R, C + pad, generator=g).to(DEV)[:, :C]
    dst = torch.full((C, R + pad), -7.0, device=DEV)
    call("snag_transpose_f32", ptr(src), src.stride(0), R, C, ptr(dst), dst.stride(0), stream())
    assert torch.equal(dst[:, :R], src.t())
    assert float((dst[:, R:] + 7.0).abs().sum()) == 0.0          # padding untouched


@pytest.mark.parametrize("R,C,pad", [(15000, 128, 256), (1, 5, 0), (333, 300, 3), (4096, 64, 0)])
def test_colsum_and_its_pair_form(R, C, pad):
    """snag_colsum (nn.Linear bias gradient: column sums of a row-strided matrix, fixed-order partial sums) against an fp64 sum, with
    accumulate; snag_colsum_pair = two such problems in one pair of launches, bit-equal to two single calls."""
    from snag_amd import ops
    from snag_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(R + C)
    bufs = [torch.randn(R, C + pad, generator=g).to(DEV) for _ in range(2)]
    xs = [b[:, :C] for b in bufs]                                     # row stride C + pad
    work = torch.empty(2 * 256 * C, device=DEV)
    singles = []
    for x in xs:
        out = torch.full((C,), 3.0, device=DEV)
        call("snag_colsum", ptr(x), R, C, x.stride(0), ptr(out), 1, ptr(work), stream())          # accumulate onto 3
        ref = x.double().sum(0) + 3.0
        assert float((out.double() - ref).abs().max()) <= 1e-5 * max(1.0, R ** 0.5)
        out0 = torch.empty(C, device=DEV)
        call("snag_colsum", ptr(x), R, C, x.stride(0), ptr(out0), 0, ptr(work), stream())
        singles.append(out0)
    o0, o1 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    call("snag_colsum_pair", ptr(xs[0]), ptr(xs[1]), R, C, xs[0].stride(0), ptr(o0), ptr(o1), 0, ptr(work), stream())
    assert torch.equal(o0, singles[0]) and torch.equal(o1, singles[1])
    assert torch.equal(ops.colsum(xs[0].contiguous()), singles[0])


@pytest.mark.parametrize("with_z", [False, True])
def test_noise_apply_quad_path_equals_the_element_path(with_z):
    """snag_noise_apply handles a quad of columns per thread (one dwordx4 per operand, ONE Philox draw for the quad) when the rows are
    whole quads and every pointer is 16-byte aligned, and goes element by element otherwise (drawing the quad's normals for each of
    its elements and keeping one).  Same inputs behind aligned and behind misaligned pointers: identical outputs."""
    from snag_amd._lib import call, ptr, stream
    N, F = 333, 40
    g = torch.Generator().manual_seed(4)
    x = torch.randn(N, F, generator=g)
    mean, std = torch.randn(F, generator=g), torch.rand(F, generator=g) + 0.1
    mask = (torch.rand(N, generator=g) < 0.4).to(torch.uint8).to(DEV)
    nsel = int(mask.sum())
    sel = (torch.cumsum(mask.cpu().to(torch.int32), 0) - 1).to(torch.int32).to(DEV)
    z = torch.randn(max(nsel, 1), F, generator=g)
    outs = []
    for off in (0, 1):                      # 1: every float buffer starts 4 bytes past a 16-byte boundary
        def put(t):
            b = torch.zeros(t.numel() + 4, device=DEV)
            v = b[off:off + t.numel()].view(t.shape)
            v.copy_(t.to(DEV))
            return v
        xd, md, sd, zd = put(x), put(mean), put(std), put(z)
        od = put(torch.zeros(N, F))
        call("snag_noise_apply", ptr(xd), N, F, ptr(md), ptr(sd), ptr(mask), ptr(sel) if with_z else None, ptr(zd) if with_z else None, 12345, 0.7,
             ptr(od), stream())
        outs.append(od.clone())
    assert torch.equal(outs[0], outs[1])
    untouched = mask.cpu() == 0
    assert torch.equal(outs[0].cpu()[untouched], x[untouched])
    if with_z:
        want = x.clone()
        rows = (~untouched).nonzero().flatten()
        want[rows] = (1.0 - 0.7) * x[rows] + 0.7 * (mean + std * z[sel.cpu()[rows].long()])
        torch.testing.assert_close(outs[0].cpu(), want, atol=1e-6, rtol=1e-6)
