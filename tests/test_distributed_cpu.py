"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding, weight broadcast, output gather and
max-over-ranks timing helpers bench.py uses on RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.environ["REPO_ROOT"])
    import numpy as np, torch, torch.distributed as dist
    from __graft_entry__ import load_package, PKG_NAME
    load_package()
    from importlib import import_module
    D = import_module(PKG_NAME + ".dist")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # 1. weight arena broadcast (chunked): every rank ends up with rank 0's bytes
    n = 3_000_001
    arena = torch.arange(n, dtype=torch.int64).to(torch.uint8) if rank == 0 else torch.zeros(n, dtype=torch.uint8)
    D.broadcast_weights(dist, arena, src=0, chunk_bytes=1 << 20)
    assert torch.equal(arena, torch.arange(n, dtype=torch.int64).to(torch.uint8))
    # 2. contiguous batch sharding covers the global batch exactly once
    lo, hi = D.shard_range(13, world, rank)
    rows = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 4)
    pad = torch.full((7 - rows.shape[0], 4), -1.0)
    got = D.gather_rows(dist, torch, torch.cat([rows, pad]), world)
    got = got[got[:, 0] >= 0]
    assert torch.equal(got[:, 0], torch.arange(13, dtype=torch.float32)), got
    # 3. max-over-ranks timing
    m = D.max_over_ranks(dist, torch, 1.0 + rank, "cpu")
    assert m == float(world)
    dist.barrier()
    dist.destroy_process_group()
    print("OK", rank)
""")


def test_two_rank_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   REPO_ROOT=ROOT, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0 and f"OK {r}" in out, out


def test_shard_range_properties(pkg):
    from importlib import import_module
    from __graft_entry__ import PKG_NAME
    D = import_module(PKG_NAME + ".dist")
    for gb in (1, 7, 64, 65):
        for world in (1, 2, 4, 8):
            spans = [D.shard_range(gb, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert D.shard_range(64, 8, 3) == (24, 32)  # BASELINE config 4: 64 images = 8 x 8


def test_bench_watchdog_exits_with_stage_and_rank():
    """bench.py's watchdog: a collective set-up stage that overruns its budget ends the process with exit code 4 and a
    message that names the stage and the rank -- instead of a hang that only the driver's timeout would end."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "wd = bench.Watchdog(3)\n"
            "with wd.stage('quick stage', 5.0):\n    pass\n"
            "with wd.stage('weight-arena broadcast (test)', 0.6):\n    time.sleep(30)\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 4, (r.returncode, r.stderr[-500:])
    assert "rank 3" in r.stderr and "weight-arena broadcast (test)" in r.stderr and "quick stage" not in r.stderr
