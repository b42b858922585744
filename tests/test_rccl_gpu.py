"""GPU: the collective path on the backend that ships.  torch.distributed "nccl" IS RCCL on ROCm; with one GPU on the box the
communicator has world_size 1, which still runs the real start()/finish() machinery of GradAllReduce (async RCCL work on RCCL's
stream behind the compute stream, the optimiser waiting on it) -- what the gloo tests cannot cover.  Runs in a fresh child process."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_engine_step_through_rccl_equals_the_plain_step_bit_for_bit():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_child.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])     # RCCL writes banner lines to stdout too
    assert out["backend"] == "nccl"
    for jw in ("Mformer_hd_graph", "Mformer_hd_mean"):
        o = out[jw]
        # summing over ONE rank is the identity, so with bit-reproducible steps any mis-ordering between the collective (RCCL's
        # stream) and the clip / AdamW kernels (compute stream), or a bucket exchanged before its gradients were final, shows up as a bit
        assert o["params_bit_equal"] and o["grads_bit_equal"] and o["moments_bit_equal"] and o["losses_equal"], o
        assert o["loss3"] < o["loss0"]
        assert o["rows_params_bit_equal"] and o["rows_grads_bit_equal"] and o["rows_losses_equal"], o      # row-sharded data path over RCCL
        fa, fb = o["ln_tail_losses"]              # LayerNorm backward inside the scorer's pass vs its own launch: same step up to summation order
        assert all(abs(x - y) <= 1e-4 * abs(x) for x, y in zip(fa, fb)), o
        # the default single-rank engine (and its one-launch form) vs the row-sharded run over RCCL: first-step gradients agree to summation order,
        # the four-step loss trajectories to 1e-4 (Adam turns last-bit differences of near-zero gradient entries into +-lr: parameters are not compared)
        assert o["default_vs_rows_grad_rel"] <= 2e-5 and o["one_launch_vs_rows_grad_rel"] <= 2e-5, o
        assert all(abs(x - y) <= 1e-4 * abs(x) for x, y in zip(o["one_launch_losses"], o["rows_losses"])), o
    la, lb = out["fused_losses"]                  # default step (fused scorer, fp32 atomics): equal up to the atomics' arrival order
    assert all(abs(a - b) <= 1e-4 * abs(a) for a, b in zip(la, lb)) and la[2] < la[0], (la, lb)
    assert out["steps_agreed"] == 4 and out["epoch_loss_finite"]
