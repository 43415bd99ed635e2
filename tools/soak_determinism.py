#!/usr/bin/env python3
"""Soak test (run on the GPU box): the full ViT-L/14 forward at batch 32, N times on the same input; every run's logits and patch
tokens must equal the first run's bit for bit.  A stale-LDS race in a persistent kernel shows up as a rare difference.
    python tools/soak_determinism.py [--iters 60] [--batch 32] [--model large] [--size 518]"""
import argparse, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
from __graft_entry__ import PKG_NAME, load_package
pkg = load_package(); api = import_module(PKG_NAME + ".api")
ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=60); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--model", default="large"); ap.add_argument("--size", type=int, default=518); args = ap.parse_args()
path = os.path.join(tempfile.gettempdir(), f"soak_{args.model}.gguf")
if not os.path.exists(path):
    pkg.synth.write_synthetic_gguf(path, args.model, registers=4, num_classes=1000, seed=42)
imgs = pkg.synth.synthetic_images(args.batch, args.size, args.size, seed=7)
sess = api.Session(api.Model(path, classify=True))
ref = sess.predict(imgs, classify=True, want=("logits", "patch_tokens"))
bad = 0
for it in range(args.iters):
    out = sess.predict(imgs, classify=True, want=("logits", "patch_tokens"))
    for k in ("logits", "patch_tokens"):
        if not np.array_equal(out[k], ref[k]):
            d = np.abs(out[k].astype(np.float64) - ref[k].astype(np.float64))
            print(f"iteration {it}: {k} differs, max |d| {d.max():.3e}, {int((d > 0).sum())} elements"); bad += 1
print(f"{args.iters} repeats of batch {args.batch} {args.model}: {'all bit-identical' if bad == 0 else str(bad) + ' differences'}; finite: {bool(np.isfinite(ref['logits']).all())}")
sys.exit(1 if bad else 0)
