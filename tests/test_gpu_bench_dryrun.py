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
    # the line explains itself: every rank's own rate, step time, in-kernel clock and broadcast time
    pr = j["per_rank"]
    assert [r["rank"] for r in pr] == list(range(n)) and all(r["images_per_sec"] > 0 and r["ms_per_step"] > 0 for r in pr)
    assert all(r["weight_broadcast_ms"] > 0 and 1.0 < r["effective_clock_ghz"] < 2.6 for r in pr)
    assert all(r["images_per_sec_min"] <= r["images_per_sec"] <= r["images_per_sec_max"] for r in pr)
    assert max(r["ms_per_step"] for r in pr) <= j["ms_per_step"] * 1.5  # `value` is the max over ranks per window: no rank far beyond it
    assert j["value_host_buffers"] is None  # (N = 1 only)
    # N > 1: rank 0 also times the one-process group front over all N devices -- in a child process, so that a crash there costs this field only
    gf = j["group_front"]
    assert gf and "error" not in gf, gf
    assert gf["value"] > 0 and gf["global_batch"] == n * batch and len(gf["devices"]) == n and len(gf["topology"]) == n
    assert 0.3 < gf["ratio_to_value"] < 1.3


def test_bench_group_front_eight_entries():
    """`--front group`: the C-ABI's own multi-device front end (one process, dinov2_hip_group_submit / _wait, page-locked host buffers) as
    the headline, rehearsed as an 8-entry group on the one visible device with a global batch of 8 x 4."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--front", "group", "--gpus", "8", "--devices", "0,0,0,0,0,0,0,0",
                        "--batch", "4", "--steps", "3", "--warmup", "1", "--windows", "2"], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["front"] == "group" and j["n_gpus"] == 8 and j["config"]["global_batch"] == 32 and j["devices"] == [0] * 8
    assert j["value"] > 0 and len(j["window_values"]) == 2
    assert "group_broadcast_ms" in j  # (None here: a group of duplicates of ONE device has nothing to broadcast over; > 0 on real peers)


def test_bench_watchdog_names_the_stage():
    """A rendezvous that cannot complete (world size 2, one process) must end with exit code 4 and a message naming the stage and the rank,
    not hang until the caller's timeout."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dist-timeout", "5", "--steps", "1",
                        "--no-cpu-baseline", "--no-latency"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "init_process_group" in r.stderr or "watchdog" in r.stderr, r.stderr[-2000:]


def test_bench_single_rank_line_has_spread_and_clock():
    """N = 1, the driver's own invocation shape (small steps here): the line carries the window spread and the in-kernel clock."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--windows", "3", "--warm-seconds", "0.2",
                        "--no-cpu-baseline", "--no-latency"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 1 and "backend" not in j and j["windows"] == 3 and len(j["window_values"]) == 3
    assert j["value_min"] <= j["value"] <= j["value_max"] and abs(j["ms_per_step"] * j["value"] / 1e3 - 32) < 0.5
    assert 1.0 < j["effective_clock_ghz"] < 2.6, j["effective_clock_ghz"]
    assert sorted(j["kernel_clocks_ghz"]) == ["attention", "gemm_attn_out", "gemm_ffn_in", "gemm_ffn_out", "gemm_qkv"]
    assert all(1.0 < c < 2.6 for c in j["kernel_clocks_ghz"].values()) and "value_at_nominal_clock_if_clock_bound" not in j
    assert j["images_per_sec_per_ghz"] > 0 and j["per_rank"] is None
    hb = j["value_host_buffers"]  # host buffers in and out through dinov2_hip_group_submit / _wait, never the headline
    assert hb and hb["value"] > 0 and 0.5 < hb["ratio_to_value"] < 1.1 and hb["in_flight"] == 2
