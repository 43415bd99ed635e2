#!/usr/bin/env python3
"""Does a GEMM output row depend on WHERE the row sits in the matrix?  A = [X; Y; X]: both X blocks must match bitwise."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
from __graft_entry__ import PKG_NAME, load_package
load_package(); api = import_module(PKG_NAME + ".api")
fp = C.POINTER(C.c_float)
P = lambda a: a.ctypes.data_as(fp) if a is not None else fp()
rng = np.random.default_rng(0)
T, K = 1374, 1024
for name, epi, N in (("plain", 5, 1024), ("plain", 5, 4096), ("qkv", 1, 3072), ("gelu", 3, 1024), ("gelu", 3, 4096), ("resid", 2, 1024)):
    X = rng.standard_normal((T, K)).astype(np.float16).astype(np.float32)
    Y = rng.standard_normal((T, K)).astype(np.float16).astype(np.float32)
    A = np.ascontiguousarray(np.concatenate([X, Y, X]))
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float16).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    aux = rng.standard_normal(N).astype(np.float32)
    M = A.shape[0]
    for tile in (128, 256, 0):  # forced small-tile kernel, forced 256-row tiles, the dispatcher's own plan
        api.set_tuning("gemm_tile", tile)  # (the library reads DINOV2_HIP_GEMM_TILE once; inside a process the switch is set through the C-ABI)
        out = np.zeros((M, N), np.float32)
        rc = api.lib().dinov2_hip_op_gemm(0, epi, P(A), P(W), P(bias), P(aux), aux.size, P(out), M, N, M, N, K, 0, 0, 0, N // 3, 0.125)
        a, b = out[:T], out[2 * T:]
        d = np.abs(a - b)
        rows = np.where(d.max(1) > 0)[0]
        print(f"{name:6s} N={N} tile={tile or 'auto'} ({api.gemm_plan(0, epi, M, N, K)}) rc={rc}: rows differing {len(rows)} of {T}; max diff {d.max():.3e}; "
              f"first rows {rows[:8].tolist()}; cols of first {np.where(d[rows[0]] > 0)[0][:8].tolist() if len(rows) else []}")
    api.reset_tuning("gemm_tile")
