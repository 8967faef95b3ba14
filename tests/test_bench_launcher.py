"""bench.py's N-rank launcher on the CPU: `--gpus N` without WORLD_SIZE must start N fresh rank processes that
rendezvous (GS_BENCH_DRYRUN=1: gloo, no GPU, no product code) and report n_gpus == N; a WORLD_SIZE that
disagrees with --gpus must fail loudly.  (The real thing runs in tests/test_gpu_bench.py.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_spawns_two_ranks():
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"GS_BENCH_DRYRUN": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout            # rank 0 prints ONE line
    assert lines[0]["n_gpus"] == 2 and lines[0]["rank_sum"] == 1.0


def test_world_size_mismatch_fails_loudly():
    p = _run(["--gpus", "4"], {"GS_BENCH_DRYRUN": "1", "WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_failing_rank_fails_the_launch():
    # a rank that cannot start (no HIP device in this container, dry-run off) must surface as a non-zero exit
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {"GS_BENCH_BACKEND": "gloo"})
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0
