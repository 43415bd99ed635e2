#!/usr/bin/env python3
"""bench.py -- images/sec of the dino_predict hot path on N MI355X GPUs (BASELINE.json metric).

A "step" is one pass of the hot path (patch-embed .. classifier head, C-ABI `dinov2_hip_predict`) over one batch of
synthetic preprocessed 518x518 images that already sit in HBM; logits/probs stay on the device.  Workload at every N:
BASELINE.json configs[2] = ViT-L/14 + 4 registers, f16, 518x518, batch 32 PER GPU (weak scaling: independent images,
no data-path collective; the only collective is the one-time RCCL broadcast of rank 0's converted weight arena).

  python bench.py                                   # N=1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W      # one rank per GPU over RCCL

Timing: after the warm-up steps (and at least `--warm-seconds` of them, so that clocks and caches are in their loaded state) the
script times `--windows` (default 5) windows of EXACTLY `--steps` steps each, every window bracketed by a barrier +
synchronize on both sides and taken as the MAX over ranks; `value` / `ms_per_step` are the MEDIAN window (`value_min`,
`value_max`, `window_values` carry the spread), and `effective_clock_ghz` is the shader clock the power-limited part sustained during
the timed windows: s_memtime against the 100 MHz s_memrealtime, summed by workgroup 0 over EVERY launch of each of the five heavy kernel
kinds (`kernel_clocks_ghz`: QKV / attn-out / FFN-in / FFN-out GEMMs, attention), weighted by each kind's share of the step -- boxes
differ by several per cent in exactly that clock (`images_per_sec_per_ghz` is the figure to compare across boxes).

`--backend gloo` (or DINOV2_BENCH_BACKEND=gloo) is a DRY RUN of the N > 1 path on however many GPUs are visible: all ranks share
the visible device(s) and the collectives go through gloo -- same shard logic, weight broadcast, `broadcast_verified`, config-4
leg with its teardown / reload, early return of ranks != 0 and final barriers as under RCCL.  Its JSON line says
`"backend": "gloo-dryrun"`: it is a correctness rehearsal, never a scaling number.

`value_host_buffers` (N = 1): the same workload with HOST buffers on both sides of the boundary -- page-locked f32 images in, logits out,
through `dinov2_hip_group_submit` / `_wait` with two batches in flight from one host thread -- i.e. the timing definition of the
reference's `inference.cpp:64-68` (wall time around the whole predict call, input upload included); never `value`.
`--front group` times that front end as the headline instead (one process, N devices behind the C-ABI, `group_broadcast_ms`).
With N > 1 the line carries `per_rank` (every rank's own window times, in-kernel clock and broadcast time: a straggler is visible) and
every collective set-up stage runs under a watchdog that names the stage and the rank on stderr and exits non-zero instead of hanging.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (the FFN-in GEMM + GELU epilogue, 27 % of all
FLOPs) from HIP events recorded around each of its launches on the session's own stream; `cpu_baseline` times the
CPU oracle (restatement of the reference graph -- the reference itself cannot be built offline) on one image, at the
reference's default `-t 4` and at the best of a few larger OpenMP teams.

With N > 1 the line also carries `broadcast_verified` (every rank runs one shared probe image after the weight broadcast and
rank 0 checks that all ranks return rank 0's logits bit for bit: the broadcast arena is usable) and `config4`: BASELINE.json
configs[3], ViT-g/14 SwiGLU bf16 with a GLOBAL batch of 64 sharded 64/N per GPU, weights broadcast over RCCL, timed the same way.

Parity statement used everywhere in this repository: max|d_logit| <= 1e-3 * max(1, max|logit|) against the CPU oracle (f16).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0  # dense f16/bf16, /opt/skills/guides/MI355X_MICROARCH.md


def commit_id() -> str:
    """The commit the measured tree is at: .head_sha (written by tools/gpurun_measure.sh before the snapshot travels to the GPU box, which
    has no .git), `git rev-parse` where a repository is present, else the id compiled into libdinov2_hip.so at build time
    (dinov2_hip_build_id -- what the driver's box reports); "+dirty" when uncommitted changes were in the tree."""
    try:
        f = os.path.join(ROOT, ".head_sha")
        if os.path.exists(f):
            return open(f).read().strip() or "unknown"
        if not os.path.exists(os.path.join(ROOT, ".git")):
            from importlib import import_module
            from __graft_entry__ import PKG_NAME, load_package
            load_package()
            return import_module(PKG_NAME + ".api").build_id() + " (library build id)"
        import subprocess
        sha = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True, timeout=10).stdout.strip()
        dirty = subprocess.run(["git", "status", "--porcelain", "--untracked-files=no"], cwd=ROOT, capture_output=True, text=True, timeout=10).stdout.strip()
        return (sha + ("+dirty" if dirty else "")) if sha else "unknown"
    except Exception:
        return "unknown"


def parity_bound(args) -> float:
    """Stated logit tolerance of a bench configuration, relative to max(1, max|logit|)."""
    if args.dtype != "f16" or args.wtype != "f16":
        return 2e-2
    return 2e-3 if args.model == "giant" else 1e-3


class Watchdog:
    """`with wd.stage("name", seconds):` -- if the block is still running after `seconds`, print which stage of which rank is stuck to
    stderr and exit the process with code 4 (a hung RCCL rendezvous / broadcast then fails the run with a message instead of sitting in
    the driver's timeout).  One daemon thread, polled twice a second."""

    def __init__(self, rank):
        import threading
        self.rank, self.cur, self.lock = rank, None, threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                cur = self.cur
            if cur and time.perf_counter() > cur[1]:
                print(f"bench.py watchdog: rank {self.rank} has been in stage '{cur[0]}' for more than {cur[2]:g} s -- giving up "
                      f"(MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')} "
                      f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')})", file=sys.stderr, flush=True)
                os._exit(4)

    def stage(self, name, seconds):
        wd = self

        class _Ctx:
            def __enter__(self_):
                with wd.lock:
                    wd.cur = (name, time.perf_counter() + seconds, seconds)

            def __exit__(self_, *exc):
                with wd.lock:
                    wd.cur = None
                return False
        return _Ctx()


def group_front(api, path, devices, dt, B, S, num_classes, steps, windows, warmup=2):
    """Throughput through the C-ABI's multi-device front end with HOST buffers on both sides (page-locked f32 images in, logits out):
    one process, `len(devices)` devices, the global batch of len(devices) * B images split contiguously, two batches in flight from this
    one host thread (dinov2_hip_group_submit / _wait).  Returns the median window and the group's own facts."""
    G = len(devices)
    grp = api.Group(path, devices=list(devices), dtype=dt, classify=True, broadcast=G > 1, streams_per_device=2)
    try:
        rng = np.random.default_rng(7)
        pin = api.pinned_empty((G * B, 3, S, S), np.float32)
        pin[:B] = rng.standard_normal((B, 3, S, S), dtype=np.float32)
        for g in range(1, G):
            pin[g * B:(g + 1) * B] = pin[:B]
        kw = dict(classify=True, want=("logits",))
        for _ in range(warmup):
            out = grp.predict(pin, **kw)
        if not np.isfinite(out["logits"]).all():
            raise SystemExit("group front: non-finite logits")
        win = []
        for _ in range(windows):
            q = []
            t0 = time.perf_counter()
            for _ in range(steps):
                if len(q) == 2:
                    grp.wait(q.pop(0))
                q.append(grp.submit(pin, **kw))
            while q:
                grp.wait(q.pop(0))
            win.append(time.perf_counter() - t0)
        med = float(np.median(win))
        return {"value": round(G * B * steps / med, 2), "unit": "images/sec", "ms_per_step": round(med / steps * 1e3, 3), "steps": steps, "windows": windows,
                "window_values": [round(G * B * steps / w, 2) for w in win], "devices": list(devices), "global_batch": G * B,
                "in_flight": 2, "host_buffers": "page-locked f32 [B, 3, S, S] in, f32 logits out",
                "group_broadcast_ms": round(grp.broadcast_ms, 2) if G > 1 and grp.broadcast_ms >= 0 else None, "topology": grp.topology,
                "definition": "wall time around whole predict calls with the input upload and the result download inside (the reference's "
                              "inference.cpp:64-68), two batches in flight from one host thread"}
    finally:
        grp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="large")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=518)
    ap.add_argument("--registers", type=int, default=4)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--wtype", default="f16", help="GGUF storage type of the 2-D weights (f16, q8_0, q4_0, ...)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-config4", action="store_true", help="N > 1 only: skip the ViT-g bf16 global-batch-64 leg")
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each; value = the median window")
    ap.add_argument("--warm-seconds", type=float, default=1.0, help="minimum wall time of the warm-up before the first window")
    ap.add_argument("--backend", default=os.environ.get("DINOV2_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="gloo = dry run of the N > 1 path with all ranks on the visible GPU(s) (labelled as such)")
    ap.add_argument("--front", default="session", choices=["session", "group"],
                    help="group = time dinov2_hip_group_submit/_wait (one process, --gpus devices, page-locked host buffers in and out) as the headline")
    ap.add_argument("--devices", default="", help="--front group: comma-separated device ordinals (default 0 .. gpus-1; a 1-GPU box can rehearse with 0,0,0,0)")
    ap.add_argument("--no-host-buffers", action="store_true", help="N = 1: skip the host-buffer leg (value_host_buffers)")
    ap.add_argument("--dist-timeout", type=float, default=180.0, help="seconds a collective set-up stage may take before the watchdog exits")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-arithmetic / ggml-style-mode distances in cpu_baseline")
    args = ap.parse_args()
    args.windows = max(1, args.windows)

    # the pool's host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails (hipIpcGetMemHandle);
    # already exported on the GPU boxes, set here as well so that a bare `torchrun bench.py` does not depend on the shell
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch  # plumbing only: device memory for the inputs, torch.distributed (RCCL) for N > 1
    from __graft_entry__ import PKG_NAME, load_package
    pkg = load_package()
    from importlib import import_module
    api = import_module(PKG_NAME + ".api")
    D = import_module(PKG_NAME + ".dist")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    wd = Watchdog(rank)

    if args.front == "group":
        # ---- the C-ABI's own multi-device front end as the headline: ONE process, N devices, host buffers in and out ----
        if world > 1:
            raise SystemExit("--front group is one process for all devices: run it without torch.distributed.run")
        devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
        cfg = pkg.synth.CONFIGS[args.model]
        path = os.path.join(tempfile.gettempdir(), f"dinov2_{args.model}_r{args.registers}_{args.wtype}_seed42.gguf")
        if not os.path.exists(path):
            tmp = path + f".{os.getpid()}.tmp"
            pkg.synth.write_synthetic_gguf(tmp, args.model, registers=args.registers, num_classes=1000, seed=42, wtype=args.wtype)
            os.replace(tmp, path)
        dtg = api.F16 if args.dtype == "f16" else api.BF16
        with wd.stage("group front (create + weight broadcast + timed windows)", max(600.0, args.dist_timeout)):
            g = group_front(api, path, devices, dtg, args.batch, args.size, 1000, args.steps, args.windows, warmup=max(2, args.warmup))
        gflop_img = pkg.synth.flops_per_image(cfg, args.size, args.size, args.registers, 1000) / 1e9
        out = {"metric": ("images/sec (518x518), ViT-L/14 fp16" if (args.model, args.size, args.dtype) == ("large", 518, "f16") else
                          f"images/sec ({args.size}x{args.size}), ViT-{args.model[0].upper()}/14 {'fp16' if args.dtype == 'f16' else args.dtype}"),
               "value": g["value"], "unit": "images/sec", "commit": commit_id(), "n_gpus": len(devices), "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": g["ms_per_step"], "windows": args.windows, "window_values": g["window_values"], "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "front": "group",
               "front_note": "dinov2_hip_group_submit/_wait: one process, one host thread, two batches in flight, page-locked host buffers in "
                             "and out (PCIe-inclusive: NOT comparable with the device-resident `value` of the default front)",
               "devices": g["devices"], "group_broadcast_ms": g["group_broadcast_ms"], "topology": g["topology"],
               "config": {"workload": f"dinov2-{args.model} (ViT-{args.model[0].upper()}/14, {args.registers} registers) {args.wtype} GGUF, "
                                      f"{args.size}x{args.size}, batch={args.batch} per GPU, classify head, random-init weights",
                          "global_batch": g["global_batch"], "parallelism": f"dp{len(devices)}", "gflop_per_image": round(gflop_img, 1)},
               "whole_forward_tflops_per_gpu": round(g["value"] / len(devices) * gflop_img / 1e3, 1)}
        print(json.dumps(out), flush=True)
        return

    dist = None
    dryrun = args.backend == "gloo"
    if dryrun:  # rehearsal: every rank on the GPU(s) this box has
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    # DINOV2_BENCH_FORCE_DIST=1 exercises the RCCL path (process group, arena broadcast, max-over-ranks) on a 1-GPU box
    if world > 1 or os.environ.get("DINOV2_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # RCCL prints a version banner with printf -- to STDOUT, flushed when the process exits, i.e. after the JSON line this
        # script owes the driver.  Point fd 1 at stderr while the communicator comes up, flush C stdio, then restore it, so
        # that stdout carries the one JSON line and nothing else.
        import ctypes
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            import datetime
            tmo = datetime.timedelta(seconds=args.dist_timeout)
            with wd.stage("init_process_group (rendezvous)", args.dist_timeout + 30):
                if dryrun:
                    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
                else:
                    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=tmo)
            with wd.stage("first barrier (communicator creation over xGMI)", args.dist_timeout + 30):
                dist.barrier()  # forces the communicator (and the banner) now
                torch.cuda.synchronize()
            # The LAST rendezvous goes through a second, CPU-side (gloo) group with a long timeout: ranks != 0 reach it minutes before rank 0
            # (which still has its per-kernel profile, the latency legs and -- round 6 -- the group front over all N devices to run), and an
            # RCCL barrier would (a) time out after --dist-timeout and take the whole job down, (b) keep a spinning kernel on every other GPU
            # while rank 0's group front measures them.
            tail_pg = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=3600))
        finally:
            try:
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    cfg = pkg.synth.CONFIGS[args.model]
    num_classes = 1000
    # ---- weights: rank 0 writes + loads + converts the synthetic GGUF; the others receive the arena over RCCL/xGMI
    path = os.path.join(tempfile.gettempdir(), f"dinov2_{args.model}_r{args.registers}_{args.wtype}_seed42.gguf")
    if rank == 0 and not os.path.exists(path):
        tmp = path + f".{os.getpid()}.tmp"
        pkg.synth.write_synthetic_gguf(tmp, args.model, registers=args.registers, num_classes=num_classes, seed=42,
                                       wtype=args.wtype)
        os.replace(tmp, path)
    if dist is not None:
        with wd.stage("barrier behind the synthetic GGUF", args.dist_timeout + 120):
            dist.barrier()
    dt = api.F16 if args.dtype == "f16" else api.BF16
    t_load = time.perf_counter()
    model = api.Model(path, device=local, dtype=dt, classify=True, skip_tensor_data=(rank != 0))
    bcast_ms = None
    if dist is not None:
        ptr, nbytes = model.arena()
        arena = torch.as_tensor(D.DevPtr(ptr, nbytes), device=f"cuda:{local}")
        torch.cuda.synchronize()
        with wd.stage("weight-arena broadcast (rank 0 -> all, %.0f MB)" % (nbytes / 1e6), args.dist_timeout):
            dist.barrier()  # every rank's arena exists: what follows is the broadcast alone
            t0 = time.perf_counter()
            D.broadcast_weights(dist, arena, src=0)
            torch.cuda.synchronize()
            bcast_ms = (time.perf_counter() - t0) * 1e3
    load_s = time.perf_counter() - t_load
    sess = api.Session(model)

    # every rank runs ONE shared probe image on its (broadcast) arena; rank 0 checks that all ranks agree with it bit for bit
    bcast_ok = None
    if dist is not None:
        pg = torch.Generator(device=f"cuda:{local}").manual_seed(1234)
        probe = torch.randn((1, 3, args.size, args.size), generator=pg, device=f"cuda:{local}", dtype=torch.float32)
        plog = torch.empty((1, num_classes), device=f"cuda:{local}", dtype=torch.float32)
        torch.cuda.synchronize()
        sess.predict_device(probe.data_ptr(), 1, args.size, args.size, classify=True, layout=api.RGB_CHW, logits_ptr=plog.data_ptr())
        sess.sync()
        with wd.stage("all-gather of the probe image's logits", args.dist_timeout):
            allp = D.gather_rows(dist, torch, plog, world)
        bcast_ok = bool((allp == allp[0:1]).all().item()) and bool(torch.isfinite(allp).all().item())
        # every rank gathered the same rows, so every rank reaches the same verdict and exits together (no rank is left waiting in
        # a later barrier); the max-reduction makes that explicit even if a gather ever became rank-dependent
        bcast_ok = D.max_over_ranks(dist, torch, 0.0 if bcast_ok else 1.0, f"cuda:{local}") == 0.0
        if not bcast_ok:
            if rank == 0:
                print("ranks disagree on the probe image: the broadcast weight arena is not identical everywhere", file=sys.stderr)
            dist.destroy_process_group()
            raise SystemExit(3)

    B, S = args.batch, args.size
    T = model.tokens(S, S)
    gen = torch.Generator(device=f"cuda:{local}").manual_seed(42 + rank)
    imgs = torch.randn((B, 3, S, S), generator=gen, device=f"cuda:{local}", dtype=torch.float32)
    logits = torch.empty((B, num_classes), device=f"cuda:{local}", dtype=torch.float32)
    probs = torch.empty_like(logits)
    torch.cuda.synchronize()  # the inputs come from torch's stream; the session runs on its own non-blocking stream

    def step():
        sess.predict_device(imgs.data_ptr(), B, S, S, classify=True, layout=api.RGB_CHW, logits_ptr=logits.data_ptr(),
                            probs_ptr=probs.data_ptr())

    def fence():
        if dist is not None:
            dist.barrier()
        sess.sync()
        torch.cuda.synchronize()

    # warm-up: W steps, then more until `--warm-seconds` have passed (a cold part clocks higher than a loaded one for the first
    # few hundred ms; the windows below should all see the loaded state)
    t_warm = time.perf_counter()
    for _ in range(args.warmup):
        step()
    sess.sync()
    while time.perf_counter() - t_warm < args.warm_seconds:
        for _ in range(5):
            step()
        sess.sync()
    # `--windows` windows of EXACTLY `--steps` steps, each bracketed by barrier + synchronize and taken as the max over ranks
    def read_clock_sums():
        try:
            import ctypes
            buf = (ctypes.c_uint64 * 18)()
            return list(buf) if api.lib().dinov2_hip_op_clock_slots(buf) == 0 else None
        except Exception:
            return None

    clk0 = read_clock_sums()
    window_s, own_s = [], []
    with wd.stage("timed windows", 600.0 + args.dist_timeout):
        for _ in range(args.windows):
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            sess.sync()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            own_s.append(el)
            if dist is not None:
                el = D.max_over_ranks(dist, torch, el, f"cuda:{local}")
            window_s.append(el)
        if dist is not None:
            dist.barrier()
    elapsed = float(np.median(window_s))
    # the clock this rank's device sustained inside each heavy kernel kind over ALL launches of the timed windows (workgroup 0's s_memtime
    # against the 100 MHz s_memrealtime, running sums on the device, read before and after): FFN-in alone (`dom_clock`, the roofline's
    # kernel) and all five kinds (`slot_clock`)
    dom_clock = dom_dev_ms = None
    slot_clock = {}
    clk1 = read_clock_sums()
    if clk0 and clk1:
        for i, nm in enumerate(("gemm_qkv", "gemm_attn_out", "gemm_ffn_in", "gemm_ffn_out", "attention")):
            dc, dk, dn = clk1[3 * i] - clk0[3 * i], clk1[3 * i + 1] - clk0[3 * i + 1], clk1[3 * i + 2] - clk0[3 * i + 2]
            if dk > 0 and dn > 0:
                slot_clock[nm] = dc / (dk * 10.0)  # cycles per ns
                if nm == "gemm_ffn_in":
                    dom_clock = round(slot_clock[nm], 4)
                    dom_dev_ms = dk / dn * 1e-5  # what workgroup 0 (persistent: first tile to last) spent inside a launch on average, device clock
    # every rank's own numbers (a straggling GPU or a slow link is invisible in a max over ranks): gathered on all, printed by rank 0
    per_rank = None
    if dist is not None:
        own_med = float(np.median(own_s))
        mine = torch.tensor([[float(rank), float(local), B * args.steps / own_med, own_med / args.steps * 1e3, float(dom_clock or 0.0),
                              float(bcast_ms or 0.0), B * args.steps / max(own_s), B * args.steps / min(own_s)]], dtype=torch.float64,
                            device=f"cuda:{local}")
        with wd.stage("all-gather of the per-rank numbers", args.dist_timeout):
            allr = D.gather_rows(dist, torch, mine, world).cpu().numpy()
        per_rank = [{"rank": int(r[0]), "device": int(r[1]), "images_per_sec": round(float(r[2]), 2), "ms_per_step": round(float(r[3]), 3),
                     "effective_clock_ghz": round(float(r[4]), 4) or None, "weight_broadcast_ms": round(float(r[5]), 2),
                     "images_per_sec_min": round(float(r[6]), 2), "images_per_sec_max": round(float(r[7]), 2)} for r in allr]
    if not bool(torch.isfinite(probs).all()):
        raise SystemExit("non-finite probabilities")

    # ---- BASELINE configs[3] (N > 1 only): ViT-g/14 SwiGLU bf16, global batch 64 = N x 64/N, weights by RCCL broadcast ----
    config4 = None
    if dist is not None and not args.no_config4 and 64 % world == 0:
        try:
            del sess
            model.close()
            torch.cuda.empty_cache()
            sess = model = None
            gpath = os.path.join(tempfile.gettempdir(), f"dinov2_giant_r{args.registers}_f16_seed42.gguf")
            if rank == 0 and not os.path.exists(gpath):
                tmp = gpath + f".{os.getpid()}.tmp"
                pkg.synth.write_synthetic_gguf(tmp, "giant", registers=args.registers, num_classes=num_classes, seed=42)
                os.replace(tmp, gpath)
            dist.barrier()
            gm = api.Model(gpath, device=local, dtype=api.BF16, classify=True, skip_tensor_data=(rank != 0))
            gptr, gbytes = gm.arena()
            garena = torch.as_tensor(D.DevPtr(gptr, gbytes), device=f"cuda:{local}")
            torch.cuda.synchronize()
            with wd.stage("config-4 weight-arena broadcast (%.0f MB)" % (gbytes / 1e6), args.dist_timeout + 60):
                dist.barrier()
                t0 = time.perf_counter()
                D.broadcast_weights(dist, garena, src=0)
                torch.cuda.synchronize()
                g_bcast_ms = D.max_over_ranks(dist, torch, (time.perf_counter() - t0) * 1e3, f"cuda:{local}")
            gs = api.Session(gm)
            lo, hi = D.shard_range(64, world, rank)
            gB = hi - lo
            gimgs = torch.randn((gB, 3, S, S), generator=gen, device=f"cuda:{local}", dtype=torch.float32)
            glog = torch.empty((gB, num_classes), device=f"cuda:{local}", dtype=torch.float32)
            torch.cuda.synchronize()

            def gstep():
                gs.predict_device(gimgs.data_ptr(), gB, S, S, classify=True, layout=api.RGB_CHW, logits_ptr=glog.data_ptr())

            for _ in range(2):
                gstep()
            dist.barrier()
            gs.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gsteps = max(2, min(args.steps, 5))
            for _ in range(gsteps):
                gstep()
            gs.sync()
            torch.cuda.synchronize()
            g_el = D.max_over_ranks(dist, torch, time.perf_counter() - t0, f"cuda:{local}")
            g_fin = D.max_over_ranks(dist, torch, 0.0 if bool(torch.isfinite(glog).all()) else 1.0, f"cuda:{local}")
            gcfg = pkg.synth.CONFIGS["giant"]
            ggf = pkg.synth.flops_per_image(gcfg, S, S, args.registers, num_classes) / 1e9
            config4 = {"workload": f"dinov2-giant (ViT-g/14, SwiGLU, {args.registers} registers) bf16, {S}x{S}, global batch 64 = {world} x {gB}",
                       "value": round(64 * gsteps / g_el, 2), "unit": "images/sec", "ms_per_step": round(g_el / gsteps * 1e3, 3), "steps": gsteps,
                       "dtype": "bf16", "gflop_per_image": round(ggf, 1), "tflops_per_gpu": round(64 * gsteps / g_el * ggf / 1e3 / world, 1),
                       "weight_broadcast_ms": round(g_bcast_ms, 2), "arena_mb": round(gbytes / 1e6, 1), "finite": g_fin == 0.0}
            del gs
            gm.close()
        except Exception as e:  # a side leg: the headline was measured above and rank 0 still owes the driver its line
            config4 = {"error": repr(e)}
        finally:
            # the ViT-L model again for the rank-0 measurements below, whatever happened in the leg
            if rank == 0 and model is None:
                model = api.Model(path, device=local, dtype=dt, classify=True)
                sess = api.Session(model)

    if rank != 0:
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier(group=tail_pg)
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    gflop_img = pkg.synth.flops_per_image(cfg, S, S, args.registers, num_classes) / 1e9

    # ---- roofline of the dominant kernel: HIP events around every launch, on the stream the kernels run on ----
    sess.profile(True)
    prof_steps = 2
    for _ in range(prof_steps):
        step()
    prof = sess.profile_read()
    sess.profile(False)
    H, F, L = cfg["hidden"], cfg["ffn"], cfg["layers"]
    M = B * T
    flops_launch = {  # algorithmic FLOPs per launch; padding does not count
        "gemm_ffn_in": 2.0 * M * H * (2 * F if cfg["swiglu"] else F),
        "gemm_ffn_out": 2.0 * M * F * H,
        "gemm_qkv": 2.0 * M * H * 3 * H,
        "gemm_attn_out": 2.0 * M * H * H,
        "attention": 4.0 * B * T * T * H,
        "gemm_patch_embed": 2.0 * B * (T - 1 - args.registers) * 588 * H,
    }
    kernels = {}
    total_ms = sum(v[0] for v in prof.values()) or 1.0
    for name, (ms, n) in prof.items():
        if n == 0:
            continue
        avg = ms / n
        k = {"avg_ms": round(avg, 4), "launches_per_step": n // prof_steps, "share": round(ms / total_ms, 4)}
        if name in flops_launch:
            k["tflops"] = round(flops_launch[name] / (avg * 1e-3) / 1e12, 1)
        kernels[name] = k
    dom = "gemm_ffn_in"
    ach = kernels.get(dom, {}).get("tflops", 0.0)
    # time-weighted shader clock of the step: each heavy kernel kind's in-kernel clock (all launches of the timed windows) weighted by
    # the time the step spends in that kind (LayerNorm, head, im2col -- memory-bound, 7 % of the step -- carry no stamp and no weight)
    eff_clock, clk_cover = None, 0.0
    if slot_clock:
        wsum = csum = 0.0
        for nm, c in slot_clock.items():
            kk = kernels.get(nm)
            if kk:
                w = kk["avg_ms"] * kk["launches_per_step"]
                wsum += w
                csum += w * c
        if wsum > 0:
            eff_clock = round(csum / wsum, 4)
            clk_cover = round(wsum / (sum(k["avg_ms"] * k["launches_per_step"] for k in kernels.values()) or 1.0), 3)
    # HBM-side bytes per launch of the dominant kernel: rocprofv3 PMC (FETCH_SIZE and WRITE_SIZE in separate passes, KiB,
    # FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md), collected by tools/hbm_traffic.sh over this same
    # command and committed under profiles/ (a PMC pass cannot run inside the timed process).
    traffic = None
    traffic_src = None
    try:
        tfile = next(f for f in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json")
                     if os.path.exists(os.path.join(ROOT, "profiles", f)))
        tj = json.load(open(os.path.join(ROOT, "profiles", tfile)))
        # FFN-in at this shape runs the mixed 256/192-row launch (gemm.hip plan C): one kernel per GEMM -- gemm4.hip's since round 4
        syms = {"f16": ("gemm4_mixed_kernelIDF16_Li3E", "gemm4_kernelIDF16_Li3E", "gemm2_mixed_kernelIDF16_Li3E", "gemm2_kernelIDF16_Li3E"),
                "bf16": ("gemm4_mixed_kernelIDF16bLi3E", "gemm4_kernelIDF16bLi3E", "gemm2_mixed_kernelIDF16bLi3E", "gemm2_kernelIDF16bLi3E")}[args.dtype]
        hit = [v for sym in syms for k, v in tj.items() if sym in k][:1]
        if hit and args.model == "large" and B == 32 and not cfg["swiglu"]:
            traffic = round(hit[0]["hbm_bytes_per_launch_corrected"])
            traffic_src = {"file": "profiles/" + tfile, "commit": tj.get("commit", "unknown"),
                           "box": "the builder's measurement box of that round (gpurun), not this run's"}
    except Exception:
        traffic = None
    roofline = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "traffic_note": "HBM-side bytes/launch from rocprofv3 FETCH_SIZE x2 + WRITE_SIZE (profiles/r0N_hbm_traffic.json, newest round); "
                                "algorithmic bytes/launch = %d" % int(2 * M * H + 2 * H * F + 2 * M * F),
                "flops_per_launch": flops_launch[dom], "avg_launch_ms": kernels.get(dom, {}).get("avg_ms"),
                # cross-check without the two events a profiled launch carries (they add ~ 5 us of gaps per launch; rocprofv3's average
                # sits between the two): the same kernel's launches of the timed windows, averaged, on the device's own 100 MHz clock
                "in_kernel_clock_ghz": dom_clock,
                "in_kernel_ms_device_clock": None if not dom_dev_ms else round(dom_dev_ms, 4),
                "achieved_by_device_clock": None if not dom_dev_ms else round(flops_launch[dom] / (dom_dev_ms * 1e-3) / 1e12, 1),
                "whole_forward_tflops": round(value / world * gflop_img / 1e3, 1),
                "whole_forward_frac": round(value / world * gflop_img / 1e3 / MFMA_PEAK_TFLOPS, 4),
                # context, not the contract's peak: what v_mfma_f32_32x32x16_f16 alone sustains on random f16 operands on
                # this power-limited part (tools/probes/mfma_peak.hip; 2 480 TF on zeros)
                "sustained_mfma_tflops_random_operands": 1750.0, "frac_of_sustained": round(ach / 1750.0, 4)}

    # ---- p50 latency at batch 1 (the other half of BASELINE.json's metric) ----
    p50 = p99 = p50_224 = p99_224 = p50_fold = p50_224_fold = None
    if not args.no_latency:
        one = imgs[:1].contiguous()

        def latency(sx, img1, side):
            torch.cuda.synchronize()
            lat = []
            for i in range(220):  # 20 warm-up + 200 timed forwards
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sx.predict_device(img1.data_ptr(), 1, side, side, classify=True, layout=api.RGB_CHW, logits_ptr=logits.data_ptr(),
                                  probs_ptr=probs.data_ptr())
                sx.sync()
                lat.append((time.perf_counter() - t0) * 1e3)
            return round(float(np.median(lat[20:])), 3), round(float(np.percentile(lat[20:], 99)), 3)

        # (every plan is batch-invariant: this image alone == this image inside the batch of B, bit for bit)
        p50, p99 = latency(sess, one, S)
        # the reference's own README regime: 224 x 224, batch 1 (device-resident input here; tools/readme_table.py has the host-in /
        # logits-out wall time the reference's table quotes)
        s224 = api.Session(model)
        p50_224, p99_224 = latency(s224, torch.randn((1, 3, 224, 224), generator=gen, device=f"cuda:{local}", dtype=torch.float32), 224)
        del s224
        # the library's latency option (dinov2_hip_load_opts.ln_fold = 1: LayerNorm 1 / 2 carried by the neighbouring GEMM epilogues, five
        # launches per layer instead of seven; opt-in because it costs 2.5 % of the batch-32 rate -- profiles/r06_ln_fold.md).  Rank 0, N = 1 only.
        if world == 1 and os.environ.get("DINOV2_HIP_LN_FOLD") is None:
            try:
                mf = api.Model(path, device=local, dtype=dt, classify=True, ln_fold=1)
                sf = api.Session(mf)
                p50_fold, _ = latency(sf, one, S)
                p50_224_fold, _ = latency(sf, torch.randn((1, 3, 224, 224), generator=gen, device=f"cuda:{local}", dtype=torch.float32), 224)
                del sf, mf
            except api.DinoError:
                pass

    # ---- side measurement, NOT the headline: the same batch split over two sessions (two HIP streams) on this GPU.  The other
    #      stream's kernels fill the idle CUs of a GEMM's last round; per-kernel durations of overlapped launches would mean
    #      nothing against the roofline, so `value` and `roofline` above stay single-stream (profiles/r01_gemm_tuning.md 7).
    two_stream = None
    if world == 1 and B >= 2 and B % 2 == 0 and not args.no_latency:
        pair = [api.Session(model), api.Session(model)]
        hb = B // 2

        def step2():
            for i, sx in enumerate(pair):
                sx.predict_device(imgs[i * hb:(i + 1) * hb].data_ptr(), hb, S, S, classify=True, layout=api.RGB_CHW,
                                  logits_ptr=logits[i * hb:(i + 1) * hb].data_ptr(), probs_ptr=probs[i * hb:(i + 1) * hb].data_ptr())
            for sx in pair:
                sx.sync()

        for _ in range(2):
            step2()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step2()
        torch.cuda.synchronize()
        two_stream = round(B * 5 / (time.perf_counter() - t0), 2)
        del pair

    # ---- the same workload with HOST buffers on both sides of the boundary (never `value`): page-locked f32 images in, logits out, two
    #      batches in flight through dinov2_hip_group_submit / _wait -- the reference's own timing definition (inference.cpp:64-68 times the
    #      whole predict call, upload included)
    host_leg = None
    if world == 1 and not dryrun and not args.no_host_buffers:
        try:
            with wd.stage("host-buffer leg (dinov2_hip_group_*)", 600.0):
                host_leg = group_front(api, path, [local], dt, B, S, num_classes, steps=max(4, min(args.steps, 10)), windows=3)
            host_leg["ratio_to_value"] = round(host_leg["value"] / value, 4)
        except Exception as e:  # a side measurement must not take the headline down
            host_leg = {"error": repr(e)}

    # ---- N > 1: the OTHER multi-GPU front end in the same run: one process (this rank), all N devices behind
    #      dinov2_hip_group_submit / _wait, host buffers in and out, while the other ranks idle at the final barrier -- so that the first
    #      8-GPU run yields both front ends and `per_rank` from one driver command.  A side measurement: an error goes into the field.
    group_leg = None
    if world > 1 and not args.no_host_buffers:
        try:
            devs = [0] * world if dryrun else list(range(world))
            # ... in a CHILD process (this file with --front group): the multi-device lanes of group.cpp have only ever time-shared one
            # device, and a crash or a hang in them must cost this field, not the headline line rank 0 still owes the driver
            import subprocess
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                     "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            cmd = [sys.executable, os.path.abspath(__file__), "--front", "group", "--gpus", str(world), "--devices", ",".join(map(str, devs)),
                   "--model", args.model, "--batch", str(B), "--size", str(S), "--registers", str(args.registers), "--dtype", args.dtype,
                   "--wtype", args.wtype, "--steps", str(max(4, min(args.steps, 10))), "--windows", "3", "--warmup", "2"]
            with wd.stage("group front over %d devices (child process)" % world, 330.0):
                # (a healthy run takes under a minute; the limit bounds what a hang in the never-exercised multi-device path can add to the job)
                cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=270)
            lines = [ln for ln in cp.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
            if cp.returncode != 0 or not lines:
                raise RuntimeError("child exit code %d: %s" % (cp.returncode, cp.stderr.decode(errors="replace")[-400:]))
            child = json.loads(lines[-1])
            group_leg = {k: child.get(k) for k in ("value", "unit", "ms_per_step", "steps", "windows", "window_values", "devices",
                                                   "group_broadcast_ms", "front_note", "whole_forward_tflops_per_gpu")}
            group_leg["global_batch"] = child["config"]["global_batch"]
            group_leg["topology"] = child.get("topology")
            group_leg["ratio_to_value"] = round(group_leg["value"] / value, 4)
        except BaseException as e:
            group_leg = {"error": repr(e)}

    # ---- CPU baseline: the oracle (restatement of the reference graph) on the host cores, bounded sample ----
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle.oracle import OracleModel
        ora = OracleModel(path)
        img1 = imgs[0].cpu().numpy()
        # the box may give the container far fewer CPUs than os.cpu_count() says (256 threads measured 20-40x SLOWER than
        # 16): time a few team sizes on the same image and report the best -- a CPU baseline should not be handicapped
        t0 = time.perf_counter()
        ora.forward(img1, classify=True, nthreads=4)  # the reference's default: dino_params.n_threads = min(4, hw) (dinov2.h:62)
        t4_s = time.perf_counter() - t0
        best_s, cores, exp = None, None, None
        for nt in (8, 16, 32):
            t0 = time.perf_counter()
            e = ora.forward(img1, classify=True, nthreads=nt)
            dt_cpu = time.perf_counter() - t0
            if best_s is None or dt_cpu < best_s:
                best_s, cores, exp = dt_cpu, nt, e
        cpu_s = best_s
        got = None
        # parity spot-check of the timed configuration itself (image 0 of the last step)
        step()
        sess.sync()
        gl = logits[0].cpu().numpy()
        dl = float(np.abs(gl - exp["logits"]).max())
        big = float(np.abs(exp["logits"]).max())
        # The tolerance stated both ways: besides the distance to the ggml-mode oracle, the ABSOLUTE distance of the
        # HIP logits to EXACT arithmetic on the same stored weights (oracle_forward_exact: double, no intermediate rounding), next to the
        # same distance of the oracle's ggml-style modes (f16 activation rounding on / off x f16 GELU table on / off): which of the
        # two approximates the model better cannot be read off "HIP vs oracle" alone.  f16 GGUFs only (the switches are f16 semantics).
        exact = None
        if not args.no_exact and args.wtype == "f16" and args.dtype == "f16":
            try:
                ex = ora.forward_exact(img1, classify=True, nthreads=cores)["logits"]
                d_modes = {"ggml_default": float(np.abs(exp["logits"] - ex).max())}
                act0 = int(ora.c.act_round)
                for name, kw in (("act_round_0", dict(act_round=0, gelu_f16_lut=1)), ("no_gelu_lut", dict(act_round=act0, gelu_f16_lut=0)),
                                 ("act_round_0_no_gelu_lut", dict(act_round=0, gelu_f16_lut=0))):
                    ora.set(**kw)
                    d_modes[name] = float(np.abs(ora.forward(img1, classify=True, nthreads=cores)["logits"] - ex).max())
                ora.set(act_round=act0, gelu_f16_lut=1)
                d_hip = float(np.abs(gl.astype(np.float64) - ex).max())
                worst = max(d_modes.values())
                exact = {"max_abs_logit_diff_vs_exact": round(d_hip, 6), "max_abs_logit_exact": round(float(np.abs(ex).max()), 4),
                         "oracle_modes_max_abs_diff_vs_exact": {k: round(v, 6) for k, v in d_modes.items()},
                         "hip_over_worst_ggml_style": round(d_hip / worst, 3) if worst > 0 else None}
            except Exception as e:  # the checker must not take the measurement down
                exact = {"error": repr(e)}
        cpu = {"value": round(1.0 / cpu_s, 4), "unit": "images/sec", "cores": cores, "kind": "port",
               "sample": f"1 image, {args.model} 518x518 batch 1, full predict (wall time of the whole call, as inference.cpp:64-68 "
                         f"times it), best of OpenMP teams of 8 / 16 / 32 threads (host reports {os.cpu_count()} CPUs)",
               "value_4_threads": round(1.0 / t4_s, 4), "note_4_threads": "the reference's default -t 4 (dinov2.h:62)",
               "max_abs_logit_diff_vs_gpu": round(dl, 6), "max_abs_logit": round(big, 4),
               "rel_logit_diff_vs_gpu": round(dl / max(1.0, big), 6),
               # the bound that applies to THIS run: relative to the largest logit; f16 weights + f16 compute 1e-3
               # (ViT-g 40 layers 2e-3); bf16 compute 2e-2; quantised GGUFs are checked against the ggml-mode oracle here
               # (activations quantised to q8_0 blocks as well), where the stated bound is 2e-2
               "parity_bound": f"max|d_logit| <= {parity_bound(args):g} * max(1, max|logit|)",
               "within_bound": bool(dl <= parity_bound(args) * max(1.0, big)),
               # north_star's wording read literally -- an ABSOLUTE 1e-3 on the logits -- next to the stated (relative) bound.  It is NOT
               # reliably met and cannot be from this side: the ggml-mode oracle is itself 0.96e-3 (mean over images; worst 1.09e-3) from exact
               # arithmetic on this model, i.e. the contract's own f16 roundings already use the whole absolute budget (profiles/r05_parity_attribution.md)
               "within_bound_absolute_1e-3": bool(dl <= 1e-3),
               "max_abs_logit_diff_vs_exact": None if not exact else exact.get("max_abs_logit_diff_vs_exact"),
               "hip_over_worst_ggml_style": None if not exact else exact.get("hip_over_worst_ggml_style"),
               "distance_to_exact": exact}
        del got

    out = {
        # BASELINE.json's metric on its default configuration; other --model / --size / --dtype runs are labelled as what they are
        "metric": ("images/sec (518x518), ViT-L/14 fp16" if (args.model, args.size, args.dtype) == ("large", 518, "f16") else
                   f"images/sec ({args.size}x{args.size}), ViT-{args.model[0].upper()}/14 {'fp16' if args.dtype == 'f16' else args.dtype}"),
        "value": round(value, 2), "unit": "images/sec", "commit": commit_id(),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "windows": args.windows, "value_min": round(world * B * args.steps / max(window_s), 2),
        "value_max": round(world * B * args.steps / min(window_s), 2),
        "window_values": [round(world * B * args.steps / w, 2) for w in window_s],
        "timing": f"median of {args.windows} windows of {args.steps} steps (barrier + synchronize around each, max over ranks), after "
                  f">= {args.warm_seconds:g} s of warm-up",
        "effective_clock_ghz": eff_clock, "nominal_clock_ghz": 2.4,
        "effective_clock_note": "time-weighted over the five heavy kernel kinds (kernel_clocks_ghz; they cover %s of the step's kernel time); "
                                "rank 0's device, averaged over every launch of the timed windows" % clk_cover,
        "kernel_clocks_ghz": {k: round(v, 4) for k, v in slot_clock.items()} or None,
        "images_per_sec_per_ghz": None if not eff_clock else round(value / world / eff_clock, 1),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"dinov2-{args.model} (ViT-{args.model[0].upper()}/14, {args.registers} registers) "
                               f"{args.wtype} GGUF, {S}x{S}, batch={B} per GPU, classify head, random-init weights",
                   "global_batch": world * B, "tokens_per_image": T, "parallelism": f"dp{world}",
                   "gflop_per_image": round(gflop_img, 1)},
        "p50_latency_ms_batch1": p50, "p99_latency_ms_batch1": p99,
        "latency_mode": "batch-invariant kernels (batch-1 bits == the image's bits inside any batch)",
        "p50_latency_ms_batch1_224x224": p50_224, "p99_latency_ms_batch1_224x224": p99_224,
        "p50_latency_ms_batch1_ln_fold": p50_fold, "p50_latency_ms_batch1_224x224_ln_fold": p50_224_fold,
        "ln_fold": os.environ.get("DINOV2_HIP_LN_FOLD", "0") not in ("0", ""),  # (the headline runs the library's default: off)
        "two_sessions_images_per_sec": two_stream, "group_front": group_leg,
        "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
        "load_s": round(load_s, 2), "weight_broadcast_ms": None if bcast_ms is None else round(bcast_ms, 2),
        "broadcast_verified": bcast_ok, "config4": config4, "per_rank": per_rank,
        "value_host_buffers": host_leg,
    }
    if dryrun:
        out["backend"] = "gloo-dryrun"
        out["backend_note"] = ("rehearsal of the N > 1 code path: all ranks time-share the visible GPU(s) and the collectives go through "
                               "gloo -- NOT a scaling measurement")
    try:  # anything a native library still holds in C stdio goes out BEFORE the JSON line, which stays the last one
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(group=tail_pg)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
