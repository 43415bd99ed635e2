#!/usr/bin/env python3
"""Bitwise determinism / batch-permutation probe of the forward (run on the GPU box)."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
from __graft_entry__ import PKG_NAME, load_package
pkg = load_package(); api = import_module(PKG_NAME + ".api")
path = os.path.join(tempfile.gettempdir(), "probe_large2.gguf")
if not os.path.exists(path):
    pkg.synth.write_synthetic_gguf(path, "large", registers=4, num_classes=1000, seed=42, layers=2)
imgs = pkg.synth.synthetic_images(3, 518, 518, seed=42)
sess = api.Session(api.Model(path, classify=True))
for layer in (0, 1, 2):
    a = sess.debug_hidden(imgs, layer); b = sess.debug_hidden(imgs, layer); c = sess.debug_hidden(imgs[::-1].copy(), layer)[::-1]
    print(f"layer {layer}: same-input repeat max diff {np.abs(a-b).max():.3e}; permuted-batch max diff {np.abs(a-c).max():.3e}; "
          f"rows differing (perm) {int((np.abs(a-c).max(-1) > 0).sum())} of {a.shape[0]*a.shape[1]}")
    if np.abs(a - c).max() > 0:
        bad = np.argwhere(np.abs(a - c).max(-1) > 0)
        print("   first differing (image, token):", bad[:6].tolist(), "last:", bad[-3:].tolist())
