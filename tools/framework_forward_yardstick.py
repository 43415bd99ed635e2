#!/usr/bin/env python3
"""Yardstick, not product: the same architecture (ViT-L/14 + 4 registers, 24 layers, classifier head) as HuggingFace
Dinov2WithRegistersForImageClassification with random weights, f16, 518 x 518, batch 32, eager PyTorch-ROCm (hipBLASLt GEMMs, fused
SDPA attention) on this box -- what the framework itself gets out of the GPU for this forward, next to bench.py's number.
    python tools/framework_forward_yardstick.py"""
import time
import torch

torch.cuda.init()
from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersForImageClassification

cfg = Dinov2WithRegistersConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, mlp_ratio=4, image_size=518, patch_size=14,
                                num_register_tokens=4, num_labels=1000, hidden_act="gelu_pytorch_tanh", attn_implementation="sdpa")
with torch.device("cuda"):
    model = Dinov2WithRegistersForImageClassification(cfg).half().eval()
for B in (32, 1):
    x = torch.randn((B, 3, 518, 518), device="cuda", dtype=torch.float16)
    with torch.no_grad():
        for _ in range(3):
            model(pixel_values=x)
        torch.cuda.synchronize()
        n = 10 if B == 32 else 100
        lat = []
        t0 = time.perf_counter()
        for _ in range(n):
            t1 = time.perf_counter()
            model(pixel_values=x)
            if B == 1:
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t1) * 1e3)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    if B == 32:
        print(f"framework eager f16, batch 32: {B / dt:.1f} images/s ({dt * 1e3:.1f} ms per batch)", flush=True)
    else:
        lat.sort()
        print(f"framework eager f16, batch 1: p50 {lat[len(lat) // 2]:.2f} ms", flush=True)
