#!/usr/bin/env python3
"""Where does the HIP path's distance to exact arithmetic come from?  (profiles/r05_parity_attribution.md)

Full-depth ViT-L/14 + 4 registers at 518 x 518 (config 3's model), several images.  Yardstick: oracle_forward_exact (double, no
intermediate rounding, the same stored weights).  The HIP forward is run with ONE rounding at a time removed -- tuning builds of the
library, `make -C dinov2.cpp_amd variant V=prec<bits> VFLAGS=-DDINO_PREC=<bits> VSRC="csrc/gemm.hip csrc/attention.hip csrc/model.cpp"`
(csrc/gemm.hip, "DINO_PREC"): 1 q, 8 k, 16 v carried as hi + lo f16 words (~ 22 bits) into attention, 2 the probabilities as hi + lo into PV,
4 GELU table entries from a double tanh, 31 all of them -- each in its own process (DINOV2_HIP_LIB), on the small-tile GEMM and the
throughput attention kernel ("gemm_tile" = 128, "attn_v" = 1: the kernels that implement the switches; the product's bits are those
kernels' bits, which the run checks first).

  python tools/parity_attribution.py [--images 4] [--out gpurun_out/parity_attribution_r05.json]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = [("product kernels (default dispatch)", None, False), ("same roundings, small-tile GEMM + attention_kernel", None, True),
            ("q -> hi + lo", "prec1", True), ("k -> hi + lo", "prec8", True), ("v -> hi + lo", "prec16", True), ("P -> hi + lo", "prec2", True),
            ("GELU table from double tanh", "prec4", True), ("all five", "prec31", True)]


def worker(args):
    from importlib import import_module
    from __graft_entry__ import PKG_NAME, load_package
    pkg = load_package()
    api = import_module(PKG_NAME + ".api")
    if args.force:
        api.set_tuning("gemm_tile", 128)
        api.set_tuning("attn_v", 1)
    sess = api.Session(api.Model(args.gguf, classify=True))
    lg, tk = [], []
    for seed in range(args.seed0, args.seed0 + args.images):
        img = pkg.synth.synthetic_images(1, 518, 518, seed=seed)
        o = sess.predict(img, classify=True)
        lg.append(o["logits"][0])
        tk.append(o["patch_tokens"][0])
    np.savez(args.worker, logits=np.stack(lg), tokens=np.stack(tk))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--seed0", type=int, default=42)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_attribution_r05.json"))
    ap.add_argument("--worker", default="")
    ap.add_argument("--gguf", default="")
    ap.add_argument("--force", type=int, default=0)
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    from importlib import import_module
    from __graft_entry__ import PKG_NAME, load_package
    pkg = load_package()
    import_module(PKG_NAME + ".api")
    from oracle.oracle import OracleModel
    gguf = os.path.join(tempfile.gettempdir(), "dinov2_large_r4_f16_seed42.gguf")
    if not os.path.exists(gguf):
        pkg.synth.write_synthetic_gguf(gguf, "large", registers=4, num_classes=1000, seed=42)
    imgs = [pkg.synth.synthetic_images(1, 518, 518, seed=s)[0] for s in range(args.seed0, args.seed0 + args.images)]
    ora = OracleModel(gguf)
    exact = [ora.forward_exact(im, classify=True, nthreads=args.threads) for im in imgs]
    ex_l = np.stack([e["logits"] for e in exact]).astype(np.float64)
    ex_t = np.stack([e["patch_tokens"] for e in exact]).astype(np.float64)

    def dist(lg, tk):
        dl, dtk = lg.astype(np.float64) - ex_l, tk.astype(np.float64) - ex_t
        per = np.abs(dl).max(axis=1)
        return {"logits_max_abs_mean_over_images": float(per.mean()), "logits_max_abs_worst_image": float(per.max()),
                "logits_max_abs_per_image": [float(v) for v in per], "logits_rms": float(np.sqrt((dl ** 2).mean())),
                "tokens_max_abs_mean_over_images": float(np.abs(dtk).reshape(len(imgs), -1).max(axis=1).mean()),
                "tokens_rms": float(np.sqrt((dtk ** 2).mean()))}

    res = {"model": "ViT-L/14 + 4 registers, f16 GGUF (synthetic, seed 42), 518 x 518", "images": args.images, "seeds": [args.seed0, args.seed0 + args.images - 1],
           "max_abs_logit_exact": float(np.abs(ex_l).max()), "max_abs_token_exact": float(np.abs(ex_t).max()), "rows": {}}
    # the oracle's modes, for scale
    for name, kw in (("oracle, ggml default (f16 activation rounding + f16 GELU table, f32 attention)", {}),
                     ("oracle, attention operands rounded like the MFMA path", dict(attn_round=1)),
                     ("oracle, f32 everywhere (no activation rounding, no table)", dict(act_round=0, gelu_f16_lut=False))):
        om = OracleModel(gguf, **kw)
        o = [om.forward(im, classify=True, nthreads=args.threads) for im in imgs]
        res["rows"][name] = dist(np.stack([x["logits"] for x in o]), np.stack([x["patch_tokens"] for x in o]))
        print(name, res["rows"][name]["logits_max_abs_mean_over_images"], res["rows"][name]["logits_rms"], flush=True)
    prod = None
    for name, lib, force in VARIANTS:
        env = dict(os.environ)
        if lib:
            env["DINOV2_HIP_LIB"] = os.path.join(ROOT, "dinov2.cpp_amd", "variants", f"libdinov2_hip_v{lib}.so")
            if not os.path.exists(env["DINOV2_HIP_LIB"]):
                print("missing", env["DINOV2_HIP_LIB"], file=sys.stderr)
                continue
        tmp = os.path.join(tempfile.gettempdir(), f"attr_{lib or 'default'}_{int(force)}.npz")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", tmp, "--gguf", gguf, "--force", str(int(force)), "--images", str(args.images),
                        "--seed0", str(args.seed0)], check=True, env=env, cwd=ROOT)
        z = np.load(tmp)
        row = dist(z["logits"], z["tokens"])
        if prod is None:
            prod = (z["logits"].copy(), z["tokens"].copy())
        else:
            row["bit_identical_to_product_kernels"] = bool(np.array_equal(z["logits"], prod[0]) and np.array_equal(z["tokens"], prod[1]))
            row["logits_max_abs_vs_product"] = float(np.abs(z["logits"].astype(np.float64) - prod[0]).max())
        res["rows"]["HIP: " + name] = row
        print("HIP:", name, row["logits_max_abs_mean_over_images"], row["logits_rms"], row.get("bit_identical_to_product_kernels"), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
