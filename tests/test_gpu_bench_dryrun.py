"""bench.py's N > 1 code path, executed before the driver's 8-GPU run does it for the first time: one rank per
process under torch.distributed.run with `--backend gloo`, all ranks on the one visible GPU -- the same shard logic, weight-arena
broadcast, bit-for-bit `broadcast_verified` probe, config-4 leg (ViT-g bf16, global batch 64, model teardown and reload on rank 0),
early return of ranks != 0 and final barriers as under RCCL.  The JSON line is labelled "gloo-dryrun": a rehearsal, not a scaling
number."""
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


def _run(n, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
           "--windows", "2", "--warm-seconds", "0", "--backend", "gloo", "--no-cpu-baseline", "--no-latency"] + extra
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # stdout carries the ONE JSON line (of rank 0) and nothing else that looks like one
    return json.loads(lines[0])


@pytest.mark.parametrize("n,batch", [(2, 32), (8, 8)])
def test_bench_multi_rank_dry_run(n, batch):
    j = _run(n, ["--batch", str(batch)])
    assert j["n_gpus"] == n and j["backend"] == "gloo-dryrun" and j["scaling"] == "weak"
    assert j["config"]["global_batch"] == n * batch and j["config"]["parallelism"] == f"dp{n}"
    assert j["broadcast_verified"] is True and j["weight_broadcast_ms"] > 0
    assert j["value"] > 0 and j["windows"] == 2 and len(j["window_values"]) == 2 and j["value_min"] <= j["value"] <= j["value_max"]
    c4 = j["config4"]  # BASELINE configs[3]: ViT-g/14 SwiGLU bf16, global batch 64 = n x 64/n, weights by broadcast
    assert c4 and c4["finite"] is True and c4["value"] > 0 and c4["dtype"] == "bf16" and f"= {n} x {64 // n}" in c4["workload"]
    assert c4["weight_broadcast_ms"] > 0 and c4["arena_mb"] > 2000
    assert j["roofline"]["kernel"] == "gemm_ffn_in" and j["roofline"]["achieved"] > 0  # rank 0's post-run legs still ran


def test_bench_single_rank_line_has_spread_and_clock():
    """N = 1, the driver's own invocation shape (small steps here): the line carries the window spread and the in-kernel clock."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--windows", "3", "--warm-seconds", "0.2",
                        "--no-cpu-baseline", "--no-latency"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 1 and "backend" not in j and j["windows"] == 3 and len(j["window_values"]) == 3
    assert j["value_min"] <= j["value"] <= j["value_max"] and abs(j["ms_per_step"] * j["value"] / 1e3 - 32) < 0.5
    assert 1.0 < j["effective_clock_ghz"] < 2.6, j["effective_clock_ghz"]
