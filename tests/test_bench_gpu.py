"""GPU: bench.py's contract.  One JSON line on stdout with the required keys at N = 1; and the N > 1 code path (torch.distributed.run,
row-sharded fusion, strong and weak scaling, max-over-ranks timing) as a two-rank dry run sharing the one device over gloo
(BENCH_BACKEND=gloo -- functional only; on a multi-GPU node the same code runs over RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "3", "--cpu-seconds", "0", "--no-extras"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    o = _json_line(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "roofline_score"):
        assert k in o, k
    assert o["n_gpus"] == 1 and o["steps"] == 6 and o["dtype"] == "f32" and o["value"] > 1e5 and "workload" in o["config"]
    for rf in (o["roofline"], o["roofline_score"]):
        assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] < 1.5 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9


def test_bench_default_line_carries_hits_parity_fusion_roofline_and_the_pool_variant():
    """The driver-run form (extras and the cpu_baseline leg on): the metric's second half -- Hits@1/10 of the timed engine against the float64
    oracle, with the counts behind "bit-exact ranking indices" --, the fusion roofline object (SURVEY par. 8d), the configs[3] check at full
    size and the README POOL=1 variant are all in the ONE JSON line."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--cpu-seconds", "2"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    o = _json_line(p.stdout)
    hp = o["hits_parity"]
    assert hp["n"] == 512 and hp["n_unexplained"] == 0 and hp["n_rank_diffs"] == hp["n_near_tie_explained"] and hp["max_abs_score_err_vs_f64"] < 1e-4
    assert abs(hp["hits1_gpu"] - hp["hits1_oracle"]) <= hp["n_hits_affecting_diffs"] / 512 + 0.5e-4
    assert abs(hp["hits10_gpu"] - hp["hits10_oracle"]) <= hp["n_hits_affecting_diffs"] / 512 + 0.5e-4
    rf = o["roofline_fusion"]
    assert rf["loop_equals_module_forward"] and rf["per_entity"]["bytes"] == 4620 and 0 < rf["fwd"]["frac_hbm"] < 1 and 0 < rf["bwd"]["frac_f32_mfma_peak"] < 2
    assert rf["cpu_probe"]["max_abs_err_gpu_vs_oracle"] < 1e-4
    full = o["roofline_sim"]["oracle_check"]["full_size"]
    assert full["n"] == 10500 and full["window_covers_2_delta"] and full["l2r"]["n_unexplained"] == 0 and full["r2l"]["n_unexplained"] == 0
    pv = o["pool_variant"]
    assert pv["feat_dims"] == [256, 256] and pv["pooled_from"] == [4096, 768] and 0 < pv["ms_per_step"] < o["ms_per_step"]
    assert o["cpu_baseline"]["kind"] == "port" and o["cpu_baseline"]["value"] > 0


def test_bench_pool_and_atten_weight_flags():
    """`--pool` (README POOL=1) and `--joint-way atten_weight` run the same loop; the workload string says what was timed."""
    for extra, want in ((["--pool"], "POOL=1"), (["--joint-way", "atten_weight"], "atten_weight")):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--cpu-seconds", "0", "--no-extras"] + extra,
                           capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        o = _json_line(p.stdout)
        assert want in o["config"]["workload"] and o["value"] > 1e5 and o["config"]["final_loss"] > 0


def test_bench_noise_level_step_refreshes_the_noise_in_every_timed_step():
    """README "noise-level: step": --noise-level step puts a refresh into every timed step; the line also carries the measured cost of one refresh."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "3", "--cpu-seconds", "0", "--no-kernel-timers",
                        "--noise-level", "step"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, BENCH_PHASES="1"))
    assert p.returncode == 0, p.stderr[-3000:]
    o = _json_line(p.stdout)
    assert o["config"]["noise_level"] == "step" and 0 < o["noise_refresh_us"] < 2000
    assert o["phases_us"]["noise_refresh"] > 0 and "noise_refresh" in o["phases_us"]["order"]


@pytest.mark.parametrize("extra,scaling,per_gpu", [([], "weak", 256), (["--scaling", "strong"], "strong", 128), (["--shard-fusion"], "weak", 256),
                                                     (["--shard-fusion", "--scaling", "strong"], "strong", 128)])
def test_bench_two_rank_dry_run(extra, scaling, per_gpu):
    env = dict(os.environ, BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "256", "--cpu-seconds", "0",
           "--no-extras", "--no-kernel-timers"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    o = _json_line(p.stdout)
    assert o["n_gpus"] == 2 and o["scaling"] == scaling and o["config"]["batch_per_gpu"] == per_gpu and o["config"]["global_batch"] == 2 * per_gpu
    assert ("row-sharded" in o["config"]["fusion"]) == ("--shard-fusion" in extra)          # default: plain data parallelism (full replicas + all-reduce)
    assert o["value"] == pytest.approx(2 * per_gpu * 4 / (o["ms_per_step"] * 4e-3), rel=1e-6) and o["config"]["final_loss"] > 0
    # the N > 1 line splits the step into phases (HIP events on the compute stream): what a first real 8-GPU run is diagnosed with
    ph = o["phases_us"]
    want = ["fusion_fwd", "score", "backward", "allreduce_wait", "clip_adamw"] + (["all_gather", "reduce_scatter"] if "--shard-fusion" in extra else [])
    for k in want:
        assert ph[k] > 0, (k, ph)
    assert ph["comm_exposed_us"] > 0 and ph["compute_us"] > 0 and ph["steps_sampled"] >= 1
    assert abs(ph["comm_exposed_us"] + ph["compute_us"] - sum(ph[k] for k in ph["order"])) < 1e-6 * (ph["compute_us"] + ph["comm_exposed_us"]) + 1e-3


def test_bench_gpus_n_without_the_launcher_starts_it_itself():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver types for --gpus 1): bench.py starts
    torch.distributed.run as a child process before touching the GPU and forwards the one JSON line and the return code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "256", "--cpu-seconds", "0",
                        "--no-extras", "--no-kernel-timers"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    o = _json_line(p.stdout)
    assert o["n_gpus"] == 2 and o["steps"] == 3 and o["config"]["global_batch"] == 512
    assert o["config"]["dist"] == {"backend": "gloo", "world_size": 2} and o["phases_us"]["compute_us"] > 0
    # a launcher that started the wrong number of ranks is refused, not silently benchmarked
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and "--nproc-per-node" in p.stderr


def test_bench_stress_config_single_and_two_rank_dry_run():
    """bench.py --config stress (BASELINE configs[4]: 1M x 256 bf16 table, 128 negatives): one rank, and two ranks sharing the device over gloo
    (row-owned optimiser state, reduce-scatter / all-gather phases reported separately)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "stress", "--steps", "4", "--warmup", "2", "--cpu-seconds", "0"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    o = _json_line(p.stdout)
    assert o["n_gpus"] == 1 and o["config"]["n_ent"] == 1_000_000 and o["roofline"]["bound"] == "hbm" and 0 < o["roofline"]["frac"] < 1
    assert o["phases_us"]["compute_us"] > 0 and o["phases_us"]["comm_bytes_per_rank"]["all_gather_bf16"] == 0
    # SURVEY par. 8d's second index distribution: Zipf(1.0) entity popularity (hot rows: contended gradient rows)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "stress", "--index-dist", "zipf", "--steps", "3", "--warmup", "1",
                        "--cpu-seconds", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    z = _json_line(p.stdout)
    assert "Zipf" in z["config"]["workload"] and z["config"]["final_loss"] > 0 and 0 < z["roofline"]["frac"] < 1
    env["BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "stress", "--steps", "3", "--warmup", "1", "--batch", "256",
           "--cpu-seconds", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    o = _json_line(p.stdout)
    assert o["n_gpus"] == 2 and o["config"]["global_batch"] == 512
    assert o["phases_us"]["comm_bytes_per_rank"]["reduce_scatter_fp32"] == 500_000 * 256 * 4 and o["phases_us"]["comm_us"] > 0
