"""Error of snag_gemm_f32 against an fp64 product (the library in use is chosen by SNAG_HIP_LIB): max and rms of |c - c64| / (|a| |b|)_row,
i.e. relative to the sum of the products' magnitudes, the scale on which f32 accumulation itself errs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snag_amd import ops  # noqa: E402


def main():
    g = torch.Generator(device="cuda").manual_seed(1)
    for (M, N, K, dist) in ((3000, 128, 4096, "randn"), (3000, 128, 4096, "uniform01"), (2048, 256, 128, "randn"), (512, 128, 15000, "randn"),
                            (3000, 128, 4096, "wide")):
        if dist == "randn":
            a, b = torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g)
        elif dist == "uniform01":
            a, b = torch.rand(M, K, device="cuda", generator=g), torch.rand(N, K, device="cuda", generator=g)
        else:       # magnitudes over 12 decades
            a = torch.randn(M, K, device="cuda", generator=g) * torch.pow(10.0, torch.rand(M, K, device="cuda", generator=g) * 12 - 6)
            b = torch.randn(N, K, device="cuda", generator=g) * torch.pow(10.0, torch.rand(N, K, device="cuda", generator=g) * 12 - 6)
        c = ops.gemm(a, b)
        c64 = a.double() @ b.double().t()
        scale = a.double().abs() @ b.double().abs().t()
        err = (c.double() - c64).abs() / scale
        c_t = (a @ b.t()).double()          # rocBLAS f32 for comparison
        err_t = (c_t - c64).abs() / scale
        rel = (c.double() - c64).abs().max() / c64.abs().max()
        print(f"[{M}x{N}x{K} {dist}] snag: max {float(err.max()):.3e} rms {float(err.pow(2).mean().sqrt()):.3e} (max|err|/max|c| {float(rel):.3e});"
              f"  torch f32: max {float(err_t.max()):.3e} rms {float(err_t.pow(2).mean().sqrt()):.3e}", flush=True)


if __name__ == "__main__":
    main()
