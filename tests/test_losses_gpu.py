"""GPU parity: icl_loss (HIP) vs golden G8 (reference icl_loss + autograd) and the oracle at the run_snag.sh batch size."""
import pytest
import torch

from helpers import T, assert_close, load
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("tag", ["plain", "weighted", "asym"])
def test_icl_loss_golden(tag):
    from snag_amd.losses import icl_loss
    z = load("g8_icl_loss.npz")
    emb = T(z["emb"], DEV).requires_grad_(True)
    wn = T(z["weight_norm"], DEV).requires_grad_(True) if tag == "weighted" else None
    crit = icl_loss(tau=float(z[f"{tag}.tau"]), ab_weight=float(z[f"{tag}.ab"]), n_view=2)
    loss = crit(emb, z["links"], weight_norm=wn)
    assert_close(loss, z[f"{tag}.loss"], atol=1e-4, rtol=1e-5)
    loss.backward()
    assert_close(emb.grad, z[f"{tag}.g.emb"], atol=1e-5, rtol=1e-4)
    if wn is not None:
        e2 = T(z["emb"]).requires_grad_(True)
        w2 = T(z["weight_norm"]).requires_grad_(True)
        O.icl_loss(e2, T(z["links"]), float(z[f"{tag}.tau"]), float(z[f"{tag}.ab"]), weight_norm=w2).backward()
        assert_close(wn.grad, w2.grad, atol=1e-5, rtol=1e-4, what="d weight_norm")


def test_icl_loss_training_shape_vs_oracle():
    """B = 3500 pairs (run_snag.sh:14), d = 300: one unimodal loss call of the MMEA step."""
    from snag_amd.losses import icl_loss
    g = torch.Generator().manual_seed(2)
    N, d, B = 9000, 300, 3500
    emb = torch.randn(N, d, generator=g)
    perm = torch.randperm(N, generator=g)
    links = torch.stack([perm[:B], perm[B:2 * B]], 1)
    e1 = emb.clone().requires_grad_(True)
    ref = O.icl_loss(e1, links, 0.1, 0.5)
    ref.backward()
    e2 = emb.to(DEV).requires_grad_(True)
    loss = icl_loss(tau=0.1, ab_weight=0.5)(e2, links.to(DEV))
    loss.backward()
    assert_close(loss, ref, atol=1e-4, rtol=1e-5)
    assert_close(e2.grad, e1.grad, atol=2e-6, rtol=1e-3)
    # entities outside the batch receive exactly zero gradient
    outside = torch.ones(N, dtype=torch.bool)
    outside[links.flatten()] = False
    assert float(e2.grad.cpu()[outside].abs().max()) == 0.0


def test_multi_loss_layer_matches_golden():
    from snag_amd.losses import CustomMultiLossLayer
    z = load("g8_icl_loss.npz")
    ml = CustomMultiLossLayer(6).to(DEV)
    with torch.no_grad():
        ml.log_vars.copy_(T(z["ml.log_vars"], DEV))
    ls = [l.clone().requires_grad_(True) for l in T(z["ml.losses"], DEV)]
    tot = ml(ls)
    assert_close(tot, z["ml.total"], atol=1e-6)
    tot.backward()
    assert_close(ml.log_vars.grad, z["ml.g.log_vars"], atol=1e-6)


@pytest.mark.parametrize("B,weighted", [(8, True), (64, False), (132, True)])
def test_infonce_grad_dwordx4_rows_equal_scalar_rows(B, weighted):
    """snag_infonce_grad transforms the logits in place one dwordx4 per lane when rows, lse and weights are 16-byte addressable and B is
    a multiple of 4; any other layout goes element by element.  Same numbers in both layouts -> identical output."""
    from snag_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(B)
    n = 2 * B
    logits = torch.randn(n, n, generator=g) * 3
    lse = torch.logsumexp(logits, dim=1).to("cuda")
    w = torch.rand(B, generator=g).to("cuda") if weighted else None
    outs = []
    for pitch in (n + 32 - n % 32, n + 1):
        buf = torch.zeros(n, pitch, device="cuda")
        buf[:, :n] = logits.to("cuda")
        S = buf[:, :n]
        call("snag_infonce_grad", ptr(S), S.stride(0), B, ptr(lse), ptr(w), 0.3 / B, 0.7 / B, None, stream())
        outs.append(S.clone())
    assert outs[0].stride(0) == n and torch.equal(outs[0], outs[1])
    assert float(outs[0].diagonal().abs().max()) == 0.0
