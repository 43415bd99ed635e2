#!/usr/bin/env python3
"""Yardstick helper: the vendor GEMM at 4 096^3 / 8 192^3 only (f16, uniform random operands), for PMC passes (tools/pmc_cmd.sh)."""
import torch
for n in (4096, 8192):
    a = (torch.rand((n, n), device="cuda", dtype=torch.float32) * 2 - 1).half()
    w = (torch.rand((n, n), device="cuda", dtype=torch.float32) * 2 - 1).half()
    for _ in range(30):
        c = a @ w.t()
    torch.cuda.synchronize()
