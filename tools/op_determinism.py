#!/usr/bin/env python3
"""Bitwise repeatability of single ops (run on the GPU box): same input twice, outputs must be identical."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from importlib import import_module
from __graft_entry__ import PKG_NAME, load_package
load_package(); api = import_module(PKG_NAME + ".api")
import test_gpu_ops as t
rng = np.random.default_rng(0)
B, T, nh = 3, 1374, 16; H = nh * 64
qkv = t._round(rng.standard_normal((B * T, 3 * H)).astype(np.float32) * 0.7, 0)
outs = []
for i in range(4):
    out = np.zeros((B * T, H), np.float32)
    assert api.lib().dinov2_hip_op_attention(0, t._p(qkv), t._p(out), B, T, H, nh) == 0
    outs.append(out)
print("attention: max diff across 4 runs", max(np.abs(outs[0] - o).max() for o in outs[1:]), "rows differing", int((np.abs(outs[0] - outs[1]).max(-1) > 0).sum()))
for (M, N, K, epi, name) in [(4122, 1024, 1024, t.EPI_PLAIN, "plain N1024 K1024"), (4122, 1024, 4096, t.EPI_PLAIN, "plain N1024 K4096"),
                             (4122, 3072, 1024, t.EPI_PLAIN, "plain N3072"), (1374, 1024, 4096, t.EPI_PLAIN, "plain M1374 K4096")]:
    A = t._round(rng.standard_normal((M, K)), 0); W = t._round(rng.standard_normal((N, K)) * 0.1, 0)
    bias = rng.standard_normal(N).astype(np.float32)
    outs = []
    for i in range(4):
        out = np.full((M, N), np.nan, np.float32)
        t._gemm(api, 0, epi, A, W, bias, None, out, M, N, K, N)
        outs.append(out)
    print(f"gemm {name}: max diff across 4 runs", max(np.abs(outs[0] - o).max() for o in outs[1:]))
