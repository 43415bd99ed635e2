#!/usr/bin/env python3
"""Yardstick, not product: what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, plain f16 GEMM, no fused epilogue, uniform
random operands, warm clocks) reaches on the forward's four GEMM shapes at ViT-L/14 batch 32 on this box, next to this library's
kernels WITH their epilogues (tools/kernel_bench.py).  The product never calls a BLAS; this only says how much headroom a
state-of-the-art library leaves on the same shapes.    python tools/vendor_gemm_yardstick.py"""
import sys
import time
import torch

torch.cuda.init()
dev = "cuda"
M = (int(sys.argv[1]) if len(sys.argv) > 1 else 32) * 1374  # optional argument: batch size
shapes = [("qkv", M, 3072, 1024), ("attn_out", M, 1024, 1024), ("ffn_in", M, 4096, 1024), ("ffn_out", M, 1024, 4096), ("4096^3", 4096, 4096, 4096)]
for dt in (torch.float16, torch.bfloat16):
    for name, m, n, k in shapes:
        a = (torch.rand((m, k), device=dev, dtype=torch.float32) * 2 - 1).to(dt)
        w = ((torch.rand((n, k), device=dev, dtype=torch.float32) * 2 - 1) * 0.05).to(dt)
        out = torch.empty((m, n), device=dev, dtype=dt)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1:  # warm clocks
            torch.matmul(a, w.t(), out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            torch.matmul(a, w.t(), out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        print(f"vendor {str(dt).split('.')[-1]:9s} {name:9s} M={m} N={n} K={k}  {ms:.4f} ms  {2.0 * m * n * k / ms / 1e9:7.1f} TFLOP/s", flush=True)
