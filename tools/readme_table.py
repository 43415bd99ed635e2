#!/usr/bin/env python3
"""The reference's own benchmark table (/root/reference/README.md:287-307, 363-409: wall time of the whole `dino_predict`
call, `-c` classification path, 224x224, batch 1, 100-run mean) re-run through this library on one MI355X with synthetic
weights of the same architectures, host image in / logits out like the reference.  Next to it the CPU restatement (oracle)
on this box's host cores, as a calibration of how far the restatement is from real ggml speed on the i9-14900HX.

    python tools/readme_table.py [--oracle]      -> gpurun_out/readme_table.json
"""
import argparse, json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from importlib import import_module
from __graft_entry__ import PKG_NAME, load_package
pkg = load_package(); api = import_module(PKG_NAME + ".api")
ap = argparse.ArgumentParser(); ap.add_argument("--oracle", action="store_true"); args = ap.parse_args()
PUBLISHED_MS = {("small", 4, "f16"): 64, ("base", 4, "f16"): 200, ("large", 4, "f16"): 597, ("giant", 4, "f16"): 1995,
                ("small", 0, "f16"): 62, ("base", 0, "f16"): 197, ("large", 0, "f16"): 600, ("giant", 0, "f16"): 1969,
                ("large", 4, "q8_0"): 353, ("large", 4, "q4_0"): 395, ("giant", 4, "q8_0"): 1065, ("giant", 4, "q4_0"): 1275}
rows = []
img = np.random.default_rng(42).standard_normal((1, 3, 224, 224)).astype(np.float32)
for (name, regs, wtype), pub in PUBLISHED_MS.items():
    path = os.path.join(tempfile.gettempdir(), f"rt_{name}_{regs}_{wtype}.gguf")
    if not os.path.exists(path):
        pkg.synth.write_synthetic_gguf(path, name, registers=regs, num_classes=1000, seed=42, wtype=wtype)
    def timed():
        sess = api.Session(api.Model(path, classify=True))
        for _ in range(10):
            sess.predict(img, classify=True, topk=5, want=("probs",))
        lat = []
        for _ in range(100):
            t0 = time.perf_counter()
            sess.predict(img, classify=True, topk=5, want=("probs",))
            lat.append((time.perf_counter() - t0) * 1e3)
        return lat
    lat = timed()
    row = {"model": name, "registers": regs, "weights": wtype, "published_cpu_ms": pub, "mi355x_mean_ms": round(float(np.mean(lat)), 3),
           "mi355x_p50_ms": round(float(np.median(lat)), 3), "speedup_vs_published": round(pub / float(np.mean(lat)), 1)}
    if args.oracle and wtype == "f16" and regs == 4 and name in ("small", "large"):
        from oracle.oracle import OracleModel
        ora = OracleModel(path)
        for nt in (4, 24, os.cpu_count()):
            ora.forward(img[0], classify=True, nthreads=nt)
            t0 = time.perf_counter()
            for _ in range(3):
                ora.forward(img[0], classify=True, nthreads=nt)
            row[f"cpu_restatement_ms_t{nt}"] = round((time.perf_counter() - t0) / 3 * 1e3, 1)
    rows.append(row)
    print(row, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "readme_table.json"), "w"), indent=1)
