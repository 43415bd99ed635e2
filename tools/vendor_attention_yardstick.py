#!/usr/bin/env python3
"""Yardstick, not product: torch.nn.functional.scaled_dot_product_attention (the ROCm build's flash / memory-efficient backends) on
the forward's attention shape (batch 32, 16 heads, 1 374 tokens, head dim 64, f16 / bf16, uniform random operands, warm clocks),
next to this library's attention kernel (tools/kernel_bench.py --only attention).    python tools/vendor_attention_yardstick.py"""
import time
import torch
import torch.nn.functional as F

torch.cuda.init()
B, nh, T, hd = 32, 16, 1374, 64
flops = 4.0 * B * nh * T * T * hd
for dt in (torch.float16, torch.bfloat16):
    q, k, v = ((torch.rand((B, nh, T, hd), device="cuda", dtype=torch.float32) * 2 - 1).to(dt) for _ in range(3))
    backends = []
    try:
        from torch.nn.attention import SDPBackend, sdpa_kernel
        backends = [("flash", SDPBackend.FLASH_ATTENTION), ("mem_efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)]
    except Exception as e:  # pragma: no cover
        print("no backend selection:", e)
    for name, be in backends:
        try:
            with sdpa_kernel(be):
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.1:
                    F.scaled_dot_product_attention(q, k, v)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 5 if name == "math" else 50
                e0.record()
                for _ in range(n):
                    F.scaled_dot_product_attention(q, k, v)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / n
            print(f"vendor sdpa {name:14s} {str(dt).split('.')[-1]:9s} {ms:.4f} ms  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)
        except Exception as e:
            print(f"vendor sdpa {name:14s} {str(dt).split('.')[-1]:9s} unavailable: {str(e)[:120]}", flush=True)
