#!/bin/bash
# Sample the shader clock and power while the dominant GEMM shape runs back to back (evidence for the f32-MFMA practical ceiling).
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
from snag_amd import ops
a = torch.randn(4096, 15008, device="cuda")[:, :15000]; b = torch.randn(4096, 128, device="cuda"); out = torch.empty(15000, 128, device="cuda")
t0 = time.time()
n = 0
while time.time() - t0 < 8.0:
    for _ in range(200): ops.gemm(a, b, a_kcontig=False, b_kcontig=False, out=out)
    torch.cuda.synchronize(); n += 200
dt = time.time() - t0
print(f"gemm loop: {n} launches, {dt / n * 1e6:.1f} us each (incl. fix-up), {2 * 15000 * 4096 * 128 * n / dt / 1e12:.1f} TFLOP/s sustained")
PY
PID=$!
sleep 4
for i in 1 2 3; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|mclk|power" | head -6
  sleep 1
done
wait $PID
echo "--- idle ---"
sleep 2
rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|mclk|power" | head -6
