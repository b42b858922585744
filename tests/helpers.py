"""Shared helpers for the parity tests (golden loading, tolerances)."""
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star: scores within 1e-4 fp32, ranking indices bit-exact.
ATOL = 1e-4


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def fusion_cases():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g1_fusion_*.npz")))


def T(a, device="cpu"):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device)


def fusion_inputs(z, device="cpu", requires_grad=False):
    """-> (embs list with None slots, params dict keyed by state_dict names, cfg dict)"""
    embs = []
    for i in range(6):
        if int(z["pattern"][i]):
            e = T(z[f"emb{i}"], device)
            if requires_grad:
                e.requires_grad_(True)
            embs.append(e)
        else:
            embs.append(None)
    params = {}
    for k in z:
        if k.startswith("p."):
            p = T(z[k], device)
            if requires_grad:
                p.requires_grad_(True)
            params[k[2:]] = p
    cfg = dict(N=int(z["N"]), M=int(z["M"]), D=int(z["D"]), H=int(z["H"]), I=int(z["I"]),
               ffn=bool(int(z["ffn"])), L=int(z["L"]))
    return embs, params, cfg


def fusion_masks(z, L, device="cpu"):
    masks = []
    for l in range(L):
        m = {"probs": T(z[f"mask.{l}.probs"], device), "attn_out": T(z[f"mask.{l}.attn_out"], device)}
        if f"mask.{l}.ffn_out" in z:
            m["ffn_out"] = T(z[f"mask.{l}.ffn_out"], device)
        masks.append(m)
    return masks


def assert_close(a, b, atol=ATOL, rtol=0.0, what=""):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}: max abs err {err.max().item():.3e} (tol {atol:g}+{rtol:g}*|ref|), {int(bad.sum())} bad"
