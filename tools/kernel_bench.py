#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the ViT-L/14 @518 batch-32 shapes (M = 43968): TFLOP/s of each GEMM (with its fused
epilogue) and of attention, on uniform-random operands.  Run on the GPU box:  python tools/kernel_bench.py [--batch 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module

from __graft_entry__ import PKG_NAME, load_package

load_package()
api = import_module(PKG_NAME + ".api")

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--hidden", type=int, default=1024)
ap.add_argument("--ffn", type=int, default=4096)
ap.add_argument("--tokens", type=int, default=1374)
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--dtype", type=int, default=0)
ap.add_argument("--only", default="")
ap.add_argument("--shape", action="append", default=[], help="name,epi,M,N,K (repeatable): custom GEMM shapes")
args = ap.parse_args()
L = api.lib()
H, F, T, B = args.hidden, args.ffn, args.tokens, args.batch
M = B * T
EPI = {"patch": 0, "qkv": 1, "resid": 2, "gelu": 3, "swiglu": 4, "plain": 5, "resid_ln": 6, "qkv_ln": 7, "gelu_ln": 8, "swiglu_ln": 9}
rows = [("qkv", "qkv", M, 3 * H, H), ("attn_out", "resid", M, H, H), ("ffn_in", "gelu", M, F, H),
        ("ffn_out", "resid", M, H, F), ("plain_ffn_in", "plain", M, F, H),
        # the LN-fold variants of the same launches (csrc/kernels.h, epilogues 6 .. 9)
        ("qkv_ln", "qkv_ln", M, 3 * H, H), ("attn_out_ln", "resid_ln", M, H, H), ("ffn_in_ln", "gelu_ln", M, F, H), ("ffn_out_ln", "resid_ln", M, H, F)]
if args.shape:
    rows = [(a, b, int(c), int(d), int(e)) for a, b, c, d, e in (x.split(",") for x in args.shape)]
    args.only = ""
for name, epi, m, n, k in rows:
    if args.only and name != args.only:
        continue
    ms = L.dinov2_hip_op_gemm_bench(args.dtype, EPI[epi], m, n, k, args.iters)
    print(f"gemm {name:12s} M={m} N={n} K={k} epi={epi:6s} {ms:8.4f} ms  {2.0 * m * n * k / ms / 1e9:8.1f} TFLOP/s", flush=True)
if args.shape or (args.only and args.only != "attention"):
    sys.exit(0)
ms = L.dinov2_hip_op_attention_bench(args.dtype, B, T, H, H // 64, args.iters)
print(f"attention    B={B} T={T} H={H}            {ms:8.4f} ms  {4.0 * B * T * T * H / ms / 1e9:8.1f} TFLOP/s", flush=True)
