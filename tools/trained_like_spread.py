#!/usr/bin/env python3
"""Trained-like ViT-L/14 @518 (synth.write_synthetic_gguf(trained_like=True)): the distance of the HIP path (default, and with ln_fold) and of the
ggml-default oracle to exact arithmetic over SEVERAL images -- round 5 showed that a single image resolves nothing below ~10 % on a 24-layer
network, and tests/test_gpu_trained_like.py asserts on one.  GPU box (the oracle runs on its host cores):
    python tools/trained_like_spread.py [n_images] > gpurun_out/trained_like_spread.json"""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module

from __graft_entry__ import PKG_NAME, load_package

pkg = load_package()
api = import_module(PKG_NAME + ".api")
from oracle.oracle import OracleModel  # noqa: E402  (measurement tool: the oracle is the checker here)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
path = os.path.join(tempfile.gettempdir(), "trained_like_large.gguf")
pkg.synth.write_synthetic_gguf(path, "large", registers=4, num_classes=1000, seed=42, head_std=0.12, trained_like=True)
imgs = pkg.synth.synthetic_images(n, 518, 518, seed=100)
outs = {}
for name, fold in (("hip", -1), ("hip_ln_fold", 1)):
    sess = api.Session(api.Model(path, classify=True, ln_fold=fold))
    outs[name] = sess.predict(imgs, classify=True, want=("logits", "patch_tokens"))
ora = OracleModel(path)
rows = []
for i in range(n):
    ex = ora.forward_exact(imgs[i], classify=True)
    gg = ora.forward(imgs[i], classify=True)
    big = float(np.abs(ex["logits"]).max())
    r = {"image": i, "max_abs_logit_exact": big,
         "ggml_default_vs_exact": float(np.abs(gg["logits"] - ex["logits"]).max()),
         "ggml_default_vs_exact_rms": float(np.sqrt(np.mean((gg["logits"] - ex["logits"]) ** 2)))}
    for name in outs:
        d = outs[name]["logits"][i] - ex["logits"]
        r[f"{name}_vs_exact"] = float(np.abs(d).max())
        r[f"{name}_vs_exact_rms"] = float(np.sqrt(np.mean(d ** 2)))
        r[f"{name}_vs_ggml_default_rel"] = float(np.abs(outs[name]["logits"][i] - gg["logits"]).max() / max(1.0, np.abs(gg["logits"]).max()))
    rows.append(r)
    print(json.dumps(r), file=sys.stderr, flush=True)


def mean(k):
    return float(np.mean([r[k] for r in rows]))


summary = {k: mean(k) for k in rows[0] if k != "image"}
summary["hip_over_ggml_default_max"] = summary["hip_vs_exact"] / summary["ggml_default_vs_exact"]
summary["hip_over_ggml_default_rms"] = summary["hip_vs_exact_rms"] / summary["ggml_default_vs_exact_rms"]
summary["hip_ln_fold_over_ggml_default_max"] = summary["hip_ln_fold_vs_exact"] / summary["ggml_default_vs_exact"]
summary["hip_ln_fold_over_ggml_default_rms"] = summary["hip_ln_fold_vs_exact_rms"] / summary["ggml_default_vs_exact_rms"]
summary["worst_hip_vs_ggml_default_rel"] = max(r["hip_vs_ggml_default_rel"] for r in rows)
summary["worst_hip_ln_fold_vs_ggml_default_rel"] = max(r["hip_ln_fold_vs_ggml_default_rel"] for r in rows)
print(json.dumps({"what": "trained-like ViT-L/14 @518, f16, %d images (seed 100): max / rms |logit - exact| of the HIP path, the HIP path with ln_fold and the ggml-default oracle" % n,
                  "mean_over_images": summary, "rows": rows}, indent=1))
