"""bench.py end to end on the GPU box: the one-line JSON contract at N=1 (reduced size) and the N-rank launcher
(`--gpus 2` over gloo on the one device: RCCL refuses two ranks on one GPU, the launcher and the exchange code
are the same)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return lines[0]


def test_bench_line_contract_single_gpu():
    r = _bench(["--gpus", "1", "--steps", "6", "--warmup", "3", "--gaussians", "200000", "--cpu-sample", "20000", "--cpu-reps", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_compute", "cpu_baseline", "step_ms"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 6 and r["warmup"] == 3 and r["dtype"] == "f32" and r["vs_baseline"] is None
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s"
    assert roof["n_isects_processed"] == r["config"]["n_isects"] <= r["config"]["n_isects_gsplat_lists"]
    per_isect = 128 if roof["kernel"] == "blend_bwd_kernel" else 40
    per_px = 24 if roof["kernel"] == "blend_bwd_kernel" else 20
    assert roof["algorithmic_bytes"] == per_isect * r["config"]["n_isects"] + per_px * 1920 * 1080
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    assert abs(roof["achieved"] - roof["algorithmic_bytes"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * roof["achieved"]
    assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["cores"] >= 1 and len(r["cpu_baseline"]["reps_s"]) == 1
    assert r["host"]["blocked_on_readback_ms_per_step"] >= 0.0
    # round 4: the timed loop has the reference's shape (a different shuffled view every step, per-step means-LR); the static
    # camera and the end-to-end loop with refinement are reported beside it
    assert "view_schedule" in r["config"] and "shuffle" in r["config"]["view_schedule"]
    assert r["static_view"]["train_iters_per_s"] > 0
    rl = r["real_loop"]
    assert "error" not in rl, rl
    for mode in ("captured", "eager"):
        assert rl[mode]["finite"] and rl[mode]["steps"] == 600 and len(rl[mode]["n_gaussians"]) == 6
    assert rl["captured"]["n_gaussians"] == rl["eager"]["n_gaussians"]            # the same refinement decisions in both modes
    assert abs(rl["captured"]["loss_mean_last_50"] - rl["eager"]["loss_mean_last_50"]) <= 1e-4 * abs(rl["eager"]["loss_mean_last_50"])
    assert rl["captured"]["captures"] >= 2 and rl["captured"]["overflows"] <= 2   # (a refinement that changes nothing re-captures nothing)
    assert r["roofline_compute"].get("clock_mhz") and isinstance(r["roofline"]["traffic"], list)
    assert "deferred_size_check" in r["drop_in"], r["drop_in"]


def test_bench_gpus_2_launches_two_ranks():
    r = _bench(["--gpus", "2", "--steps", "3", "--warmup", "2", "--gaussians", "100000", "--no-cpu-baseline"],
               {"GS_BENCH_BACKEND": "gloo"})
    assert r["n_gpus"] == 2 and r["config"]["parallelism"] == "view-dp2" and r["config"]["exchange"] == "factorised"
    assert r["config"]["exchange_bytes_per_rank"]["all_gather_view_record"] == 16 * 100000 + 64
    assert r["config"]["exchange_bytes_per_rank"]["collectives_per_step"] == 2
    assert r["value"] > 0
