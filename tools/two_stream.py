#!/usr/bin/env python3
"""Experiment: one session at batch 32 vs two sessions (two HIP streams) at batch 16 each, launched back to back."""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
from __graft_entry__ import PKG_NAME, load_package
pkg = load_package(); api = import_module(PKG_NAME + ".api")
path = os.path.join(tempfile.gettempdir(), "ts_large.gguf")
if not os.path.exists(path):
    pkg.synth.write_synthetic_gguf(path, "large", registers=4, num_classes=1000, seed=42)
B, S = int(os.environ.get("B", 32)), 518
model = api.Model(path, classify=True)
imgs = torch.randn(B, 3, S, S, device="cuda")
logits = torch.empty(B, 1000, device="cuda"); probs = torch.empty_like(logits)
def run(nsess, steps=10, warm=3):
    sess = [api.Session(model) for _ in range(nsess)]
    per = B // nsess
    def step():
        for i, s in enumerate(sess):
            s.predict_device(imgs[i*per:(i+1)*per].data_ptr(), per, S, S, classify=True, layout=api.RGB_CHW,
                             logits_ptr=logits[i*per:(i+1)*per].data_ptr(), probs_ptr=probs[i*per:(i+1)*per].data_ptr())
        for s in sess: s.sync()
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return B * steps / dt
for n in (1, 2, 4, 1, 2, 4):
    print(f"sessions={n}: {run(n):.1f} img/s", flush=True)
