"""Gaussian modality noise masking (GMNM) on the GPU.

Reference: ``SNAG.add_noise_to_embeddings`` / ``get_mean_std`` / ``update_noise`` (model/SNAG.py:66-98) and the
entity blend inside ``MultiModalEncoder.forward`` (model/SNAG_tools.py:127-128).  Two modes:

* ``mode="parity"``: the row masks and normal draws come from torch's CPU generator in the reference's exact call
  order (``rand(N)`` then ``randn_like(X[mask])`` per table, then ``randn_like(E)`` and ``rand(N)`` for the entity table),
  so a seeded run reproduces the reference's noisy tables; only the arithmetic runs in HIP.
* ``mode="philox"``: masks and normals are generated on the device (counter-based Philox), nothing crosses PCIe.
"""
import torch

from . import ops

f32 = torch.float32


def mean_std(x, exclude_rows=None):
    """Per-column mean / unbiased std, skipping ``exclude_rows`` (model/SNAG.py:77-84: img stats skip ent_wo_img)."""
    ex = None
    n_valid = x.shape[0]
    if exclude_rows is not None and len(exclude_rows) > 0:
        ex = torch.zeros(x.shape[0], device=x.device, dtype=torch.uint8)
        idx = torch.as_tensor(exclude_rows, device=x.device, dtype=torch.int64)
        ex[idx] = 1
        n_valid = x.shape[0] - int(torch.unique(idx).numel())
    return ops.colstats(x, ex, n_valid)


def add_noise_to_embeddings(x, mean, std, noise_ratio, mask_ratio, *, mode="parity", seed=0):
    """-> noisy copy of ``x`` (model/SNAG.py:66-75).  Parity mode consumes torch's global CPU RNG like the reference."""
    N, F = x.shape
    if mode == "parity":
        mask = torch.rand(N) < noise_ratio                         # :68
        n_sel = int(mask.sum())
        z = torch.randn(n_sel, F)                                  # :70 randn_like(embeddings[mask])
        sel = torch.full((N,), -1, dtype=torch.int32)
        sel[mask] = torch.arange(n_sel, dtype=torch.int32)
        return ops.noise_apply(x, mean, std, mask.to(torch.uint8).to(x.device), mask_ratio,
                               z=z.to(x.device) if n_sel else torch.zeros(1, F, device=x.device), sel_index=sel.to(x.device))
    keep = ops.dropout_keep_mask((N,), noise_ratio, seed * 2 + 1, device=x.device)
    return ops.noise_apply(x, mean, std, keep ^ 1, mask_ratio, seed=seed * 2 + 2)


class NoiseInjector:
    """Holds the clean tables, their statistics and the per-epoch noisy copies (``update_noise`` once per epoch, main.py:253-254)."""

    def __init__(self, tables, noise_ratio, mask_ratio, exclude_rows=None, mode="parity"):
        """tables: dict name -> [N, F] CUDA tensor in the reference's update order (rel, att, img); exclude_rows: dict name -> rows."""
        self.tables = dict(tables)
        self.noise_ratio, self.mask_ratio, self.mode = noise_ratio, mask_ratio, mode
        ex = exclude_rows or {}
        self.stats = {k: mean_std(v, ex.get(k)) for k, v in self.tables.items()}
        self.noisy = {}
        self.entity_noise = None
        self.entity_noise_mask = None
        self.epoch = 0

    def update_noise(self, entity_weight=None, seed=0):
        self.epoch += 1
        for i, (k, x) in enumerate(self.tables.items()):
            m, s = self.stats[k]
            self.noisy[k] = add_noise_to_embeddings(x, m, s, self.noise_ratio, self.mask_ratio, mode=self.mode,
                                                    seed=seed * 131 + self.epoch * 17 + i)
        if entity_weight is not None:
            w = entity_weight.detach()
            m, s = ops.colstats(w)                                 # :94-95 (current weights, every epoch)
            N = w.shape[0]
            if self.mode == "parity":
                z = torch.randn(N, w.shape[1])                     # :96
                u = torch.rand(N)                                  # :98
                self.entity_noise = ops.noise_apply(torch.zeros_like(w), m, s, torch.ones(N, device=w.device, dtype=torch.uint8), 1.0,
                                                    z=z.to(w.device), sel_index=torch.arange(N, device=w.device, dtype=torch.int32))
                self.entity_noise_mask = (u < self.noise_ratio * 0.5).to(torch.uint8).to(w.device)
            else:
                self.entity_noise = ops.gaussian_fill(m, s, N, seed * 977 + self.epoch)
                self.entity_noise_mask = ops.dropout_keep_mask((N,), self.noise_ratio * 0.5, seed * 977 + self.epoch + 5, device=w.device) ^ 1
        return self.noisy
