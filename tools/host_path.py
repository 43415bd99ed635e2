#!/usr/bin/env python3
"""Throughput when the boundary is handed HOST buffers (PCIe-inclusive), ViT-L/14 518x518, batch 32: f32 preprocessed images
(3.2 MB each) and raw 8-bit images (0.8 MB each, preprocessed on the device)."""
import os, sys, time, tempfile
import numpy as np
import torch
torch.cuda.init()  # (before the library touches the device: torch's lazy init failed when it came second)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
from __graft_entry__ import PKG_NAME, load_package
pkg = load_package(); api = import_module(PKG_NAME + ".api")
path = os.path.join(tempfile.gettempdir(), "hp_large.gguf")
if not os.path.exists(path):
    pkg.synth.write_synthetic_gguf(path, "large", registers=4, num_classes=1000, seed=42)
B = 32
sess = api.Session(api.Model(path, classify=True))
f32 = np.random.default_rng(0).standard_normal((B, 3, 518, 518)).astype(np.float32)
u8 = np.random.default_rng(1).integers(0, 256, (B, 504, 504, 3), dtype=np.uint8)  # -> (504/14 + 1) * 14 = 518
for name, fn in (("host f32 RGB_CHW", lambda: sess.predict(f32, classify=True, want=("logits",))),
                 ("host raw u8 BGR_HWC (features size 518)", lambda: sess.predict(u8, classify=False, layout=api.U8_BGR_HWC, want=("cls",)))):
    for _ in range(2): fn()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {B / dt:.1f} images/s ({dt * 1e3:.1f} ms per batch of {B})")

# two sessions on two host threads: one's host -> device copy runs under the other's forward
import threading
model = sess.model
pair = [api.Session(model), api.Session(model)]
for sx in pair:
    sx.predict(f32, classify=True, want=("logits",))
def _worker(sx):
    for _ in range(5):
        sx.predict(f32, classify=True, want=("logits",))
ts = [threading.Thread(target=_worker, args=(sx,)) for sx in pair]
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
dt = time.perf_counter() - t0
print(f"host f32, two sessions on two threads: {2 * 5 * B / dt:.1f} images/s")


# ---- the native group (dinov2_hip_group_predict: host buffers in, host buffers out) on this ONE GPU, against the same forward with
#      device-resident input and output (what bench.py times): lanes per device 1 (copy, forward, copy in sequence) and 2 (default)
dev = torch.from_numpy(f32).cuda()
logits_d = torch.empty((B, 1000), device="cuda", dtype=torch.float32)
torch.cuda.synchronize()
def _resident():
    sess.predict_device(dev.data_ptr(), B, 518, 518, classify=True, layout=api.RGB_CHW, logits_ptr=logits_d.data_ptr())
    sess.sync()
def _rate(fn, n=8):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return B * n / (time.perf_counter() - t0)
res = {"device_resident_single_session": _rate(_resident)}
pin = api.pinned_empty(f32.shape, np.float32); pin[...] = f32
pin8 = api.pinned_empty(u8.shape, np.uint8); pin8[...] = u8
del pair
def _pipelined(grp, src, depth, n=8, **kw):
    """n batches through submit / wait with `depth` jobs in flight (one host thread)."""
    for _ in range(2): grp.predict(src, **kw)
    t0 = time.perf_counter()
    q = []
    for _ in range(n):
        if len(q) == depth: grp.wait(q.pop(0))
        q.append(grp.submit(src, **kw))
    while q: grp.wait(q.pop(0))
    return B * n / (time.perf_counter() - t0)
grp = api.Group(path, devices=[0], classify=True, broadcast=False, streams_per_device=2)
kw32 = dict(classify=True, want=("logits",)); kw8 = dict(classify=False, layout=api.U8_BGR_HWC, want=("cls",))
res["group_predict_blocking_pageable_f32"] = _rate(lambda: grp.predict(f32, **kw32))
res["group_predict_blocking_pinned_f32"] = _rate(lambda: grp.predict(pin, **kw32))
res["group_predict_blocking_pinned_u8"] = _rate(lambda: grp.predict(pin8, **kw8))
res["group_2_in_flight_pageable_f32"] = _pipelined(grp, f32, 2, **kw32)
res["group_2_in_flight_pinned_f32"] = _pipelined(grp, pin, 2, **kw32)
res["group_2_in_flight_pinned_u8"] = _pipelined(grp, pin8, 2, **kw8)
res["group_2_in_flight_pinned_f32_all_outputs"] = _pipelined(grp, pin, 2, classify=True)  # + 180 MB of tokens back per batch
grp.close()
base = res["device_resident_single_session"]
for k, v in res.items():
    print(f"{k}: {v:.1f} images/s  ({v / base:.3f} x device-resident)")
import json
json.dump({k: round(v, 1) for k, v in res.items()} | {"ratio_2_in_flight_pinned_f32": round(res["group_2_in_flight_pinned_f32"] / base, 4), "ratio_2_in_flight_pageable_f32": round(res["group_2_in_flight_pageable_f32"] / base, 4),
           "workload": "ViT-L/14 + 4 reg, f16, 518x518, batch 32, classify, one MI355X; host buffers in AND out through dinov2_hip_group_predict"},
          open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "host_path.json"), "w"), indent=1)
