"""RCCL actually executes (VERDICT r2 "missing" #1; SURVEY.md section 4 "world_size=1 RCCL smoke on GPU").

The reference is single-GPU (/root/reference/train.py:36-43); the one-view-per-GPU exchange is north_star's requirement.
Every >1-rank test of this suite runs on gloo.  Here a world_size-1 `nccl` (= RCCL) group on the box's GPU drives the
*whole* exchange path -- collectives issued async, one of them from inside `backward()`, on RCCL's own streams -- and
the result must equal, bit for bit, the same path over gloo, and equal the plain one-rank step (no process group) to
rounding.  Each mode is a fresh child process (tests/rccl_child.py): nothing that touched the GPU is ever re-executed."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _child(mode, out):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_child.py"), mode, str(out)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0 and "rccl_child ok" in p.stdout, f"{mode} child failed:\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    return torch.load(out, weights_only=False)


def test_full_exchange_path_on_rccl_world_size_1(tmp_path):
    nccl = _child("nccl", tmp_path / "nccl.pt")
    gloo = _child("gloo", tmp_path / "gloo.pt")
    plain = _child("plain", tmp_path / "plain.pt")
    assert nccl["backend"] == "nccl" and nccl["world"] == 1 and nccl["vp_exchange"] is True and plain["vp_exchange"] is False
    # 3 steps x (all-gather of the per-view records, geometry + statistics all-reduce)
    assert nccl["vp_collectives"] == gloo["vp_collectives"] == 6 and plain["vp_collectives"] == 0
    bitwise_vs_plain = True
    for k, v in nccl["vp"].items():
        assert torch.equal(v, gloo["vp"][k]), f"RCCL and gloo disagree on {k}"
        ref = plain["vp"][k]
        if not torch.equal(v, ref):
            bitwise_vs_plain = False
        # the exchange path rebuilds the SH gradients from the 3 colour gradients (gs_sh_grad_views) and packs the
        # statistics in another kernel: same numbers to fp32 rounding of one product
        scale = float(ref.abs().max()) + 1e-30
        assert float((v - ref).abs().max()) <= 2e-6 * scale + 1e-12, (k, float((v - ref).abs().max()), scale)
    print("factorised exchange over RCCL vs plain one-rank step: bitwise" if bitwise_vs_plain else
          "factorised exchange over RCCL vs plain one-rank step: equal to fp32 rounding (not bitwise)")
    # plain scheme: a one-rank SUM all-reduce + division by 1 is the identity
    for name in ("plain_fused", "plain_bucket"):
        for k, v in nccl[name].items():
            assert torch.equal(v, gloo[name][k]) and torch.equal(v, plain[name][k]), (name, k)
