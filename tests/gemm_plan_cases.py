"""Which kernel plans can the GEMM dispatcher (csrc/gemm.hip launch_gemm) pick for the DINOv2 family, and which of them do the bit-equality
tests run?  Shared by tests/test_gemm_plans.py (CPU: enumerates the reachable plans through dinov2_hip_op_gemm_plan and asserts that every one
is covered) and tests/test_gpu_ops.py::test_gemm_plan_coverage_case_bits (GPU: runs every COVERAGE_CASES entry and compares its rows with
the same rows of an M = 100 launch, bit for bit).  VERDICT r4 item 7c."""
F16, BF16 = 0, 1
EPI_PATCH, EPI_QKV, EPI_RESID, EPI_GELU, EPI_SWIGLU, EPI_PLAIN = range(6)

# hidden, ffn, swiglu  (synth.CONFIGS / the four published checkpoints)
MODELS = {"small": (384, 1536, False), "base": (768, 3072, False), "large": (1024, 4096, False), "giant": (1536, 4096, True)}
BATCHES = (1, 2, 4, 8, 32, 64)
SIDES = (224, 518)
REGISTERS = (0, 4)


def model_gemms(name, batch, side, registers):
    """(epilogue, M, N, K) of every GEMM launch of one forward (csrc/model.cpp forward())."""
    H, F, swiglu = MODELS[name]
    P = (side // 14) ** 2
    T = P + 1 + registers
    M = batch * T
    out = [(EPI_PATCH, batch * P, H, 640), (EPI_QKV, M, 3 * H, H), (EPI_RESID, M, H, H), (EPI_RESID, M, H, F)]
    out.append((EPI_SWIGLU, M, 2 * F, H) if swiglu else (EPI_GELU, M, F, H))
    return out


def reachable(api, dtypes=(F16, BF16)):
    """{(leaf, epilogue, dtype): (M, N, K) of the smallest problem that reaches it} over the whole family."""
    found = {}
    for name in MODELS:
        for b in BATCHES:
            for side in SIDES:
                for r in REGISTERS:
                    for epi, M, N, K in model_gemms(name, b, side, r):
                        if M * max(N, K) * 2 >= 1 << 32:  # dinov2_hip_predict splits such batches into passes (32-bit cursors)
                            continue
                        for dt in dtypes:
                            for leaf in api.gemm_plan(dt, epi, M, N, K).split(";"):
                                key = (leaf, epi, dt)
                                if key not in found or M * N * K < found[key][0] * found[key][1] * found[key][2]:
                                    found[key] = (M, N, K)
    return found


# (dtype, epilogue, M, N, K): the smallest problem of the family for every reachable (leaf, epilogue, dtype), greedily (a split plan covers
# several leaves); regenerate with `python tests/gemm_plan_cases.py` after a dispatcher change (tests/test_gemm_plans.py fails until it is
# current).  The comment behind each case is the plan it had when the list was written.
COVERAGE_CASES = [
    (0, 2, 257, 384, 384),  # small<32x64,w1x4,st3,ks2>
    (1, 2, 257, 384, 384),  # small<32x64,w1x4,st3,ks2>
    (0, 0, 256, 384, 640),  # small<32x64,w1x4,st3,ks2>
    (1, 0, 256, 384, 640),  # small<32x64,w1x4,st3,ks2>
    (0, 1, 257, 1152, 384),  # small<32x64,w1x4,st3,ks2>
    (1, 1, 257, 1152, 384),  # small<32x64,w1x4,st3,ks2>
    (0, 3, 257, 1536, 384),  # small<32x64,w1x4,st3,ks2>
    (1, 3, 257, 1536, 384),  # small<32x64,w1x4,st3,ks2>
    (0, 2, 1370, 384, 384),  # small<64x64,w2x4,st3,ks2>
    (1, 2, 1370, 384, 384),  # small<64x64,w2x4,st3,ks2>
    (0, 1, 514, 1152, 384),  # small<64x64,w2x4,st3,ks2>
    (1, 1, 514, 1152, 384),  # small<64x64,w2x4,st3,ks2>
    (0, 3, 514, 1536, 384),  # small<64x64,w2x4,st3,ks2>
    (1, 3, 514, 1536, 384),  # small<64x64,w2x4,st3,ks2>
    (0, 0, 1369, 384, 640),  # small<64x64,w2x4,st3,ks2>
    (1, 0, 1369, 384, 640),  # small<64x64,w2x4,st3,ks2>
    (0, 2, 2740, 384, 384),  # small<64x128,w2x4,st3,ks2>
    (1, 2, 2740, 384, 384),  # small<64x128,w2x4,st3,ks2>
    (0, 1, 1028, 1152, 384),  # small<64x128,w2x4,st3,ks2>
    (1, 1, 1028, 1152, 384),  # small<64x128,w2x4,st3,ks2>
    (0, 3, 1028, 1536, 384),  # small<64x128,w2x4,st3,ks2>
    (1, 3, 1028, 1536, 384),  # small<64x128,w2x4,st3,ks2>
    (0, 0, 2738, 384, 640),  # small<64x128,w2x4,st3,ks2>
    (1, 0, 2738, 384, 640),  # small<64x128,w2x4,st3,ks2>
    (0, 3, 1370, 1536, 384),  # small<64x128,w2x2,st3,ks1>
    (1, 3, 1370, 1536, 384),  # small<64x128,w2x2,st3,ks1>
    (0, 2, 5480, 384, 384),  # small<64x128,w2x2,st3,ks1>
    (1, 2, 5480, 384, 384),  # small<64x128,w2x2,st3,ks1>
    (0, 1, 2056, 1152, 384),  # small<64x128,w2x2,st3,ks1>
    (1, 1, 2056, 1152, 384),  # small<64x128,w2x2,st3,ks1>
    (0, 1, 2740, 1152, 384),  # small<64x128,w2x2,st2,ks1>
    (1, 1, 2740, 1152, 384),  # small<64x128,w2x2,st2,ks1>
    (0, 3, 2056, 1536, 384),  # small<64x128,w2x2,st2,ks1>
    (1, 3, 2056, 1536, 384),  # small<64x128,w2x2,st2,ks1>
    (0, 2, 8224, 384, 384),  # small<64x128,w2x2,st2,ks1>
    (1, 2, 8224, 384, 384),  # small<64x128,w2x2,st2,ks1>
    (0, 0, 2048, 1024, 640),  # small<64x128,w2x2,st3,ks1>
    (1, 0, 2048, 1024, 640),  # small<64x128,w2x2,st3,ks1>
    (0, 3, 2740, 1536, 384),  # gemm2<128>
    (1, 3, 2740, 1536, 384),  # gemm2<128>
    (0, 0, 8192, 384, 640),  # small<64x128,w2x2,st2,ks1>
    (1, 0, 8192, 384, 640),  # small<64x128,w2x2,st2,ks1>
    (0, 3, 514, 4096, 1024),  # gemm4_short<64>
    (1, 3, 514, 4096, 1024),  # gemm4_short<64>
    (0, 3, 5480, 1536, 384),  # gemm2<256>
    (1, 3, 5480, 1536, 384),  # gemm2<256>
    (0, 2, 5480, 768, 768),  # gemm2<128>
    (1, 2, 5480, 768, 768),  # gemm2<128>
    (0, 1, 1028, 3072, 1024),  # gemm4_short<64>
    (1, 1, 1028, 3072, 1024),  # gemm4_short<64>
    (0, 4, 257, 8192, 1536),  # gemm4_short<64>
    (1, 4, 257, 8192, 1536),  # gemm4_short<64>
    (0, 1, 8224, 1152, 384),  # small<128x128,w2x2,st2,ks1>
    (1, 1, 8224, 1152, 384),  # small<128x128,w2x2,st2,ks1>
    (0, 1, 2056, 2304, 768),  # gemm2<128>
    (1, 1, 2056, 2304, 768),  # gemm2<128>
    (0, 1, 1370, 3072, 1024),  # gemm4_short<96>
    (1, 1, 1370, 3072, 1024),  # gemm4_short<96>
    (0, 3, 1028, 4096, 1024),  # gemm4_short<96>
    (1, 3, 1028, 4096, 1024),  # gemm4_short<96>
    (0, 1, 10960, 1152, 384),  # gemm2<192>;small<64x128,w2x4,st3,ks2>
    (1, 1, 10960, 1152, 384),  # gemm2<192>;small<64x128,w2x4,st3,ks2>
    (0, 0, 8192, 1024, 640),  # small<128x128,w2x2,st2,ks1>
    (1, 0, 8192, 1024, 640),  # small<128x128,w2x2,st2,ks1>
    (0, 2, 43840, 384, 384),  # gemm2<192>;small<64x128,w2x2,st2,ks1>
    (1, 2, 43840, 384, 384),  # gemm2<192>;small<64x128,w2x2,st2,ks1>
    (0, 2, 10960, 768, 768),  # gemm2<256>
    (1, 2, 10960, 768, 768),  # gemm2<256>
    (0, 1, 2056, 3072, 1024),  # gemm4_short<128>
    (1, 1, 2056, 3072, 1024),  # gemm4_short<128>
    (0, 4, 514, 8192, 1536),  # gemm4_short<96>
    (1, 4, 514, 8192, 1536),  # gemm4_short<96>
    (0, 0, 10952, 1024, 640),  # gemm2<192>
    (1, 0, 10952, 1024, 640),  # gemm2<192>
    (0, 1, 16448, 1152, 384),  # gemm2<256>;small<32x64,w1x4,st3,ks2>;small<64x128,w2x2,st3,ks1>
    (1, 1, 16448, 1152, 384),  # gemm2<256>;small<32x64,w1x4,st3,ks2>;small<64x128,w2x2,st3,ks1>
    (0, 1, 2740, 3072, 1024),  # gemm4<256>
    (1, 1, 2740, 3072, 1024),  # gemm4<256>
    (0, 3, 2056, 4096, 1024),  # gemm4<256>
    (1, 3, 2056, 4096, 1024),  # gemm4<256>
    (0, 3, 16448, 1536, 384),  # gemm2_mixed<256+192>
    (1, 3, 16448, 1536, 384),  # gemm2_mixed<256+192>
    (0, 3, 2740, 4096, 1024),  # gemm2<192>
    (1, 3, 2740, 4096, 1024),  # gemm2<192>
    (0, 2, 87680, 384, 384),  # gemm2<192>;small<128x128,w2x2,st2,ks1>
    (1, 2, 87680, 384, 384),  # gemm2<192>;small<128x128,w2x2,st2,ks1>
    (0, 4, 1028, 8192, 1536),  # gemm2<192>
    (1, 4, 1028, 8192, 1536),  # gemm2<192>
    (0, 1, 43840, 1152, 384),  # gemm2_mixed<256+192>;small<64x128,w2x2,st2,ks1>
    (1, 1, 43840, 1152, 384),  # gemm2_mixed<256+192>;small<64x128,w2x2,st2,ks1>
    (0, 2, 10960, 768, 3072),  # gemm4<256>
    (1, 2, 10960, 768, 3072),  # gemm4<256>
    (0, 1, 8224, 3072, 1024),  # gemm4_mixed<256+192>
    (1, 1, 8224, 3072, 1024),  # gemm4_mixed<256+192>
    (0, 4, 2056, 8192, 1536),  # gemm4<256>;small<64x128,w4x2,st3,ks2>
    (1, 4, 2056, 8192, 1536),  # gemm4<256>;small<64x128,w4x2,st3,ks2>
    (0, 2, 16448, 1536, 1536),  # gemm2_mixed<256+192>
    (1, 2, 16448, 1536, 1536),  # gemm2_mixed<256+192>
    (0, 3, 10960, 4096, 1024),  # gemm4_mixed<256+192>
    (1, 3, 10960, 4096, 1024),  # gemm4_mixed<256+192>
    (0, 4, 5480, 8192, 1536),  # gemm4_mixed<256+192>
    (1, 4, 5480, 8192, 1536),  # gemm4_mixed<256+192>
    (0, 2, 16448, 1536, 4096),  # gemm4_mixed<256+192>
    (1, 2, 16448, 1536, 4096),  # gemm4_mixed<256+192>
    (0, 4, 16704, 8192, 1536),  # gemm4<256>;small<64x128,w2x2,st3,ks1>
    (1, 4, 16704, 8192, 1536),  # gemm4<256>;small<64x128,w2x2,st3,ks1>
]


# ---- the LN-fold launches (csrc/kernels.h epilogues 6 .. 9; csrc/model.cpp forward() with dinov2_hip_load_opts.ln_fold) ----
EPI_RESID_LN, EPI_QKV_LN, EPI_GELU_LN, EPI_SWIGLU_LN = 6, 7, 8, 9


def model_gemms_ln(name, batch, side, registers):
    H, F, swiglu = MODELS[name]
    T = (side // 14) ** 2 + 1 + registers
    M = batch * T
    return [(EPI_QKV_LN, M, 3 * H, H), (EPI_RESID_LN, M, H, H), (EPI_RESID_LN, M, H, F), (EPI_SWIGLU_LN, M, 2 * F, H) if swiglu else (EPI_GELU_LN, M, F, H)]


def reachable_ln(api, dtypes=(F16, BF16)):
    found = {}
    for name in MODELS:
        for b in BATCHES:
            for side in SIDES:
                for r in REGISTERS:
                    for epi, M, N, K in model_gemms_ln(name, b, side, r):
                        if M * max(N, K) * 2 >= 1 << 32:
                            continue
                        for dt in dtypes:
                            for leaf in api.gemm_plan(dt, epi, M, N, K).split(";"):
                                key = (leaf, epi, dt)
                                if key not in found or M * N * K < found[key][0] * found[key][1] * found[key][2]:
                                    found[key] = (M, N, K)
    return found


# one problem per reachable (leaf, LN epilogue, dtype), generated like COVERAGE_CASES (`python tests/gemm_plan_cases.py`); run by
# tests/test_gpu_ln_fold.py::test_ln_plan_coverage_case_bits
LN_COVERAGE_CASES = [
    (0, 6, 257, 384, 384),  # small<32x64,w1x4,st3,ks2>
    (1, 6, 257, 384, 384),  # small<32x64,w1x4,st3,ks2>
    (0, 7, 257, 1152, 384),  # small<32x64,w1x4,st3,ks2>
    (1, 7, 257, 1152, 384),  # small<32x64,w1x4,st3,ks2>
    (0, 8, 257, 1536, 384),  # small<32x64,w1x4,st3,ks2>
    (1, 8, 257, 1536, 384),  # small<32x64,w1x4,st3,ks2>
    (0, 6, 1370, 384, 384),  # small<64x64,w2x4,st3,ks2>
    (1, 6, 1370, 384, 384),  # small<64x64,w2x4,st3,ks2>
    (0, 7, 514, 1152, 384),  # small<64x64,w2x4,st3,ks2>
    (1, 7, 514, 1152, 384),  # small<64x64,w2x4,st3,ks2>
    (0, 8, 514, 1536, 384),  # small<64x64,w2x4,st3,ks2>
    (1, 8, 514, 1536, 384),  # small<64x64,w2x4,st3,ks2>
    (0, 6, 2740, 384, 384),  # small<64x128,w2x4,st3,ks2>
    (1, 6, 2740, 384, 384),  # small<64x128,w2x4,st3,ks2>
    (0, 7, 1028, 1152, 384),  # small<64x128,w2x4,st3,ks2>
    (1, 7, 1028, 1152, 384),  # small<64x128,w2x4,st3,ks2>
    (0, 8, 1028, 1536, 384),  # small<64x128,w2x4,st3,ks2>
    (1, 8, 1028, 1536, 384),  # small<64x128,w2x4,st3,ks2>
    (0, 8, 1370, 1536, 384),  # gemm4_short<64>
    (1, 8, 1370, 1536, 384),  # gemm4_short<64>
    (0, 6, 5480, 384, 384),  # small<64x128,w2x2,st3,ks1>
    (1, 6, 5480, 384, 384),  # small<64x128,w2x2,st3,ks1>
    (0, 7, 2056, 1152, 384),  # small<64x128,w2x2,st3,ks1>
    (1, 7, 2056, 1152, 384),  # small<64x128,w2x2,st3,ks1>
    (0, 7, 2740, 1152, 384),  # small<64x128,w2x2,st2,ks1>
    (1, 7, 2740, 1152, 384),  # small<64x128,w2x2,st2,ks1>
    (0, 6, 8224, 384, 384),  # small<64x128,w2x2,st2,ks1>
    (1, 6, 8224, 384, 384),  # small<64x128,w2x2,st2,ks1>
    (0, 8, 2740, 1536, 384),  # gemm4_short<96>
    (1, 8, 2740, 1536, 384),  # gemm4_short<96>
    (0, 7, 1028, 2304, 768),  # gemm4_short<64>
    (1, 7, 1028, 2304, 768),  # gemm4_short<64>
    (0, 8, 5480, 1536, 384),  # gemm4<256>
    (1, 8, 5480, 1536, 384),  # gemm4<256>
    (0, 6, 5480, 768, 768),  # gemm4_short<96>
    (1, 6, 5480, 768, 768),  # gemm4_short<96>
    (0, 9, 257, 8192, 1536),  # gemm4_short<64>
    (1, 9, 257, 8192, 1536),  # gemm4_short<64>
    (0, 7, 8224, 1152, 384),  # small<128x128,w2x2,st2,ks1>
    (1, 7, 8224, 1152, 384),  # small<128x128,w2x2,st2,ks1>
    (0, 7, 2056, 2304, 768),  # gemm4_short<96>
    (1, 7, 2056, 2304, 768),  # gemm4_short<96>
    (0, 7, 10960, 1152, 384),  # gemm4_mixed<0+192>;small<64x128,w2x4,st3,ks2>
    (1, 7, 10960, 1152, 384),  # gemm4_mixed<0+192>;small<64x128,w2x4,st3,ks2>
    (0, 7, 2740, 2304, 768),  # gemm4_short<128>
    (1, 7, 2740, 2304, 768),  # gemm4_short<128>
    (0, 8, 2056, 3072, 768),  # gemm4_short<128>
    (1, 8, 2056, 3072, 768),  # gemm4_short<128>
    (0, 6, 8224, 768, 768),  # gemm4_short<128>
    (1, 6, 8224, 768, 768),  # gemm4_short<128>
    (0, 6, 43840, 384, 384),  # gemm4_mixed<0+192>;small<64x128,w2x2,st2,ks1>
    (1, 6, 43840, 384, 384),  # gemm4_mixed<0+192>;small<64x128,w2x2,st2,ks1>
    (0, 6, 10960, 768, 768),  # gemm4<256>
    (1, 6, 10960, 768, 768),  # gemm4<256>
    (0, 9, 514, 8192, 1536),  # gemm4_short<96>
    (1, 9, 514, 8192, 1536),  # gemm4_short<96>
    (0, 7, 16448, 1152, 384),  # gemm4<256>;small<32x64,w1x4,st3,ks2>;small<64x128,w2x2,st3,ks1>
    (1, 7, 16448, 1152, 384),  # gemm4<256>;small<32x64,w1x4,st3,ks2>;small<64x128,w2x2,st3,ks1>
    (0, 8, 16448, 1536, 384),  # gemm4_mixed<256+192>
    (1, 8, 16448, 1536, 384),  # gemm4_mixed<256+192>
    (0, 8, 2740, 4096, 1024),  # gemm4_mixed<0+192>
    (1, 8, 2740, 4096, 1024),  # gemm4_mixed<0+192>
    (0, 6, 87680, 384, 384),  # gemm4_mixed<0+192>;small<128x128,w2x2,st2,ks1>
    (1, 6, 87680, 384, 384),  # gemm4_mixed<0+192>;small<128x128,w2x2,st2,ks1>
    (0, 9, 1028, 8192, 1536),  # gemm4_mixed<0+192>
    (1, 9, 1028, 8192, 1536),  # gemm4_mixed<0+192>
    (0, 7, 43840, 1152, 384),  # gemm4_mixed<256+192>;small<64x128,w2x2,st2,ks1>
    (1, 7, 43840, 1152, 384),  # gemm4_mixed<256+192>;small<64x128,w2x2,st2,ks1>
    (0, 9, 2056, 8192, 1536),  # gemm4<256>;small<64x128,w4x2,st3,ks2>
    (1, 9, 2056, 8192, 1536),  # gemm4<256>;small<64x128,w4x2,st3,ks2>
    (0, 6, 16448, 1536, 1536),  # gemm4_mixed<256+192>
    (1, 6, 16448, 1536, 1536),  # gemm4_mixed<256+192>
    (0, 9, 5480, 8192, 1536),  # gemm4_mixed<256+192>
    (1, 9, 5480, 8192, 1536),  # gemm4_mixed<256+192>
    (0, 9, 16704, 8192, 1536),  # gemm4<256>;small<64x128,w2x2,st3,ks1>
    (1, 9, 16704, 8192, 1536),  # gemm4<256>;small<64x128,w2x2,st3,ks1>
]


def cases_leaves(api, cases):
    got = set()
    for dt, epi, M, N, K in cases:
        for leaf in api.gemm_plan(dt, epi, M, N, K).split(";"):
            got.add((leaf, epi, dt))
    return got


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from importlib import import_module
    from __graft_entry__ import PKG_NAME, load_package
    load_package()
    api = import_module(PKG_NAME + ".api")
    need = reachable(api)
    print(f"# {len(need)} reachable (leaf, epilogue, dtype) combinations")
    # greedy: smallest problems first; one case may cover several leaves (split plans)
    chosen, have = [], set()
    for key, (M, N, K) in sorted(need.items(), key=lambda kv: kv[1][0] * kv[1][1] * kv[1][2]):
        if key in have:
            continue
        case = (key[2], key[1], M, N, K)
        chosen.append(case)
        have |= cases_leaves(api, [case])
    for c in chosen:
        print(f"    {c},  # {api.gemm_plan(*c)}")
    need = reachable_ln(api)
    print(f"# LN fold: {len(need)} reachable (leaf, epilogue, dtype) combinations")
    chosen, have = [], set()
    for key, (M, N, K) in sorted(need.items(), key=lambda kv: kv[1][0] * kv[1][1] * kv[1][2]):
        if key in have:
            continue
        case = (key[2], key[1], M, N, K)
        chosen.append(case)
        have |= cases_leaves(api, [case])
    for c in chosen:
        print(f"    {c},  # {api.gemm_plan(*c)}")
