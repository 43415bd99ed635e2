#!/usr/bin/env python3
"""The reference's `realtime` loop (/root/reference/realtime.cpp:24-110) without the camera and the window: synthetic 854x480
8-bit frames -> dino_preprocess on the device ((w/14 + 1) * 14 = 868 x 490: 62 x 35 patches) -> feature forward -> patch tokens
back on the host -> 3-component PCA map.  Reports frames per second of the loop and of its parts.

    python tools/realtime_bench.py [--model small|base|large] [--frames 200]
"""
import argparse, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
from __graft_entry__ import PKG_NAME, load_package
pkg = load_package(); api = import_module(PKG_NAME + ".api"); inf = import_module(PKG_NAME + ".inference")
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="small")
ap.add_argument("--frames", type=int, default=200)
args = ap.parse_args()
path = os.path.join(tempfile.gettempdir(), f"rt_{args.model}.gguf")
if not os.path.exists(path):
    pkg.synth.write_synthetic_gguf(path, args.model, registers=4, num_classes=0, seed=42)
model = api.Model(path, classify=False)
sess = api.Session(model)
W, H = 854, 480  # FRAME_WIDTH x FRAME_HEIGHT of the reference
rng = np.random.default_rng(0)
frames = rng.integers(0, 256, (8, H, W, 3), dtype=np.uint8)
oh, ow = api.preprocess_size(0, H, W, model.hparams.patch_size)
n = args.frames
Hd = model.hparams.hidden_size
for label, mode in (("device PCA on resident tokens (dinov2_hip_pca3)", 2), ("device PCA on host tokens", 1), ("host numpy PCA", 0)):
    t_fwd = t_pca = 0.0
    for i in range(n + 10):
        f = frames[i % 8][None]
        t0 = time.perf_counter()
        r = sess.predict(f, classify=False, layout=api.U8_BGR_HWC, want=() if mode == 2 else ("patch_tokens",))
        if mode == 2:
            sess.sync()
        t1 = time.perf_counter()
        vis = inf.pca_visual(Hd if mode == 2 else r["patch_tokens"][0], oh // 14, ow // 14, oh, ow, session=sess if mode else None)
        t2 = time.perf_counter()
        if i >= 10:
            t_fwd += t1 - t0
            t_pca += t2 - t1
    print(f"{args.model}: frame {W}x{H} -> {ow}x{oh} ({(oh // 14) * (ow // 14)} patches); forward incl. H2D/D2H {t_fwd / n * 1e3:.2f} ms, "
          f"{label} {t_pca / n * 1e3:.2f} ms -> {n / (t_fwd + t_pca):.1f} frames/s ({n / t_fwd:.1f} without the PCA)")
