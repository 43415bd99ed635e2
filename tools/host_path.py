#!/usr/bin/env python3
"""Throughput when the boundary is handed HOST buffers (PCIe-inclusive), ViT-L/14 518x518, batch 32: f32 preprocessed images
(3.2 MB each) and raw 8-bit images (0.8 MB each, preprocessed on the device)."""
import os, sys, time, tempfile
import numpy as np
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
