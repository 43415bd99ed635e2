"""LN fold (csrc/kernels.h, epilogues 6 .. 9): the LayerNorm between a residual update and the next weight matmul
(/root/reference/dinov2.cpp:694-700, 722-728) carried by the two GEMM epilogues instead of a launch of its own.

Piece by piece through the diagnostic C-ABI (include/dinov2_hip_ops.h), then end to end:
  * the producer (EPI_RESID_LN): x bit-identical to the plain residual epilogue, xg = T(x gamma) exactly, row statistics BIT-EXACT against a
    numpy restatement of the fixed pairwise tree -- from every kernel plan the dispatcher can pick (gemm4 256-row / mixed / + small-tile tail,
    column split, small-tile only);
  * ln_prepare (the same outputs for a residual stream no GEMM has written) and the load-time s / c vectors;
  * the consumers (EPI_QKV_LN / GELU_LN / SWIGLU_LN) against float64, and bit-identical rows from every plan (batch invariance);
  * the model with the fold on against the model with it off and against the oracle (fixtures, full-depth ViT-L incl. the trained-like one).
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.oracle import OracleModel

pytestmark = pytest.mark.gpu

F16, BF16 = 0, 1
EPI_RESID, EPI_RESID_LN, EPI_QKV_LN, EPI_GELU_LN, EPI_SWIGLU_LN = 2, 6, 7, 8, 9
fp = C.POINTER(C.c_float)


def _p(a):
    return a.ctypes.data_as(fp) if a is not None else fp()


def _round(a, dt):
    a = np.asarray(a, np.float32)
    if dt == F16:
        return a.astype(np.float16).astype(np.float32)
    u = a.view(np.uint32)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return u.view(np.float32)


def _slots(H):
    return 12 if H // 64 <= 12 else 24


def tree_stats(x):
    """(sum, sum of squares) per row and 64-column group, in float32, in the producers' fixed order: leaves of four consecutive columns
    (a + b) + (c + d), then adjacent pairs, four levels (device_types.h ln_leaf4 + the DPP / LDS steps).  [M, slots, 2], pad slots zero."""
    x = np.asarray(x, np.float32)
    M, H = x.shape
    g = x.reshape(M, H // 64, 16, 4)
    s = (g[..., 0] + g[..., 1]) + (g[..., 2] + g[..., 3])
    q = (g[..., 0] * g[..., 0] + g[..., 1] * g[..., 1]) + (g[..., 2] * g[..., 2] + g[..., 3] * g[..., 3])
    for _ in range(4):
        s = s[..., 0::2] + s[..., 1::2]
        q = q[..., 0::2] + q[..., 1::2]
    out = np.zeros((M, _slots(H), 2), np.float32)
    out[:, :H // 64, 0] = s[..., 0]
    out[:, :H // 64, 1] = q[..., 0]
    return out


def coeffs(stats, H, eps=1e-6):
    S = stats[:, :, 0].astype(np.float64).sum(1)
    Q = stats[:, :, 1].astype(np.float64).sum(1)
    mean = S / H
    var = np.maximum(Q / H - mean * mean, 0.0)
    r = 1.0 / np.sqrt(var + eps)
    return mean, r


def _resid_ln(api, dt, A, W, bias, ls, gamma, x0):
    M, K = A.shape
    N = W.shape[0]
    x = x0.copy()
    xg = np.full((M, N), np.nan, np.float32)
    st = np.full((M, _slots(N), 2), np.nan, np.float32)
    rc = api.lib().dinov2_hip_op_gemm_resid_ln(dt, _p(A), _p(W), _p(bias), _p(ls), _p(gamma), _p(x), _p(xg), _p(st), M, N, K)
    assert rc == 0
    return x, xg, st


@pytest.mark.parametrize("dt", [F16, BF16])
@pytest.mark.parametrize("H", [128, 384, 1024, 1536])
def test_ln_prepare(api, dt, H):
    rng = np.random.default_rng(H + dt)
    rows = 517
    x = (rng.standard_normal((rows, H)) * 3 + 0.7).astype(np.float32)
    x[5, 9] = 143.0
    gamma = (rng.standard_normal(H) * 0.3 + 1).astype(np.float32)
    xg = np.full((rows, H), np.nan, np.float32)
    st = np.full((rows, _slots(H), 2), np.nan, np.float32)
    assert api.lib().dinov2_hip_op_ln_prepare(dt, _p(x), _p(gamma), _p(xg), _p(st), rows, H) == 0
    assert np.array_equal(xg, _round(x * gamma, dt))
    assert np.array_equal(st, tree_stats(x))


@pytest.mark.parametrize("dt", [F16, BF16])
def test_ln_fold_vectors(api, dt):
    rng = np.random.default_rng(3 + dt)
    N, K = 700, 384
    W = _round(rng.standard_normal((N, K)) * 0.05, dt)
    bias, gamma, beta = (rng.standard_normal(n).astype(np.float32) for n in (N, K, K))
    s, c = np.empty(N, np.float32), np.empty(N, np.float32)
    assert api.lib().dinov2_hip_op_ln_fold_vectors(dt, _p(W), _p(bias), _p(gamma), _p(beta), _p(s), _p(c), N, K) == 0
    Wd = W.astype(np.float64)
    np.testing.assert_allclose(s, Wd @ gamma.astype(np.float64), rtol=0, atol=2e-6)
    np.testing.assert_allclose(c, bias + Wd @ beta.astype(np.float64), rtol=0, atol=2e-6)


# (M, N, K, substring the dispatcher's plan must contain): every plan a producer launch can take
_PRODUCER_CASES = [(41100, 1024, 384, "gemm4_mixed<256+192>"), (98200, 512, 256, "gemm4<256>"), (16700, 1024, 256, "gemm4<256>;small<32x64"),
                   (11500, 1536, 256, "gemm4<256>;small<64x128,w2x4"), (37500, 384, 256, "gemm4_mixed<0+192>;small<64x128,w2x2"),
                   (5496, 1024, 1024, "gemm4_short<96>"), (10992, 1024, 4096, "gemm4_mixed<0+192>"),
                   (1374, 1024, 256, "small<64x128,w2x4"), (300, 128, 128, "small<64x128,w4x2"), (261, 384, 384, "small<32x64"),
                   (20000, 384, 384, "small<64x128,w2x2,st2"), (5000, 512, 256, "small<64x128,w2x2,st3")]


@pytest.mark.parametrize("dt", [F16, BF16])
@pytest.mark.parametrize("M,N,K,plan", _PRODUCER_CASES)
def test_resid_ln_producer(api, dt, M, N, K, plan):
    got_plan = api.gemm_plan(dt, EPI_RESID_LN, M, N, K)
    assert plan in got_plan, got_plan
    rng = np.random.default_rng(M + N + K + dt)
    X = _round(rng.standard_normal((100, K)), dt)  # rows repeat every 100: one reference block for every tile position
    A = np.ascontiguousarray(np.tile(X, ((M + 99) // 100, 1))[:M])
    W = _round(rng.standard_normal((N, K)) * 0.05, dt)
    bias, ls = rng.standard_normal(N).astype(np.float32), (rng.standard_normal(N) * 0.3).astype(np.float32)
    gamma = (rng.standard_normal(N) * 0.3 + 1).astype(np.float32)
    x0 = (rng.standard_normal((100, N)) * 2 + 0.4).astype(np.float32)
    x0[7, 3] = 120.0
    x0 = np.ascontiguousarray(np.tile(x0, ((M + 99) // 100, 1))[:M])
    x, xg, st = _resid_ln(api, dt, A, W, bias, ls, gamma, x0)
    # x: the plain residual epilogue's bits
    ref = x0.copy()
    assert api.lib().dinov2_hip_op_gemm(dt, EPI_RESID, _p(A), _p(W), _p(bias), _p(ls), N, _p(ref), M, N, M, N, K, 0, 0, 0, 0, 1.0) == 0
    assert np.array_equal(x, ref)
    # xg, stats: exact functions of x
    assert np.array_equal(xg, _round(x * gamma, dt))
    assert np.array_equal(st, tree_stats(x))
    # and every block of 100 rows carries the same bits (tile position, tile height, kernel)
    last = M - 100 - M % 100
    for a in (0, (M // 200) * 100, last):
        assert np.array_equal(st[a:a + 100], st[:100]) and np.array_equal(xg[a:a + 100], xg[:100])


def _consumer(api, dt, epi, A, W, s, c, st, out_cols, qcols=0, qscale=1.0, eps=1e-6):
    M, K = A.shape
    N = W.shape[0]
    out = np.full((M, out_cols), np.nan, np.float32)
    rc = api.lib().dinov2_hip_op_gemm_ln_consumer(dt, epi, _p(A), _p(W), _p(s), _p(c), _p(st), eps, _p(out), out_cols, M, N, K, qcols, qscale)
    assert rc == 0
    return out


def _consumer_ref(dt, epi, A, W, s, c, st, qcols, qscale):
    K = A.shape[1]
    mean, r = coeffs(st, K)
    acc = A.astype(np.float64) @ W.astype(np.float64).T
    v = r[:, None] * (acc - mean[:, None] * s.astype(np.float64)[None, :]) + c.astype(np.float64)[None, :]
    if epi == EPI_QKV_LN:
        v[:, :qcols] *= qscale
        return v
    if epi == EPI_GELU_LN:
        xr = v.astype(np.float32).astype(np.float16).astype(np.float64)
        return 0.5 * xr * (1 + np.tanh(0.79788456080286535587989211986876 * xr * (1 + 0.044715 * xr * xr)))
    N = W.shape[0]
    n = np.arange(N)
    x1, x2 = v[:, ((n >> 5) & 1) == 0], v[:, ((n >> 5) & 1) == 1]  # interleaved weights_in rows: 32 x1 units, then their 32 x2 units
    return x1 / (1 + np.exp(-x1)) * x2


_CONSUMER_CASES = [(41100, 1024, 384), (9300, 2048, 1024), (1374, 3072, 1024), (261, 1536, 384), (300, 384, 128), (1374, 4096, 1536)]


@pytest.mark.parametrize("dt", [F16, BF16])
@pytest.mark.parametrize("epi", [EPI_QKV_LN, EPI_GELU_LN, EPI_SWIGLU_LN])
@pytest.mark.parametrize("M,N,K", _CONSUMER_CASES)
def test_ln_consumer(api, dt, epi, M, N, K):
    """v = r (acc - mean s) + c with mean / r from the statistics rows, then the QKV / GELU / SwiGLU epilogue, against float64; the rows
    repeat every 100, and every copy -- whatever tile, tile height or kernel computed it -- must carry the first copy's bits, which are also
    those of a 100-row launch (the small-tile kernel)."""
    rng = np.random.default_rng(M + N + K + dt + epi)
    X = _round(rng.standard_normal((100, K)) * 1.5, dt)
    A = np.ascontiguousarray(np.tile(X, ((M + 99) // 100, 1))[:M])
    W = _round(rng.standard_normal((N, K)) * 0.04, dt)
    s, c = (rng.standard_normal(N) * 0.5).astype(np.float32), rng.standard_normal(N).astype(np.float32)
    # statistics of some residual rows (not tied to A: the epilogue only sees numbers)
    xs = (rng.standard_normal((100, K)) * 2 + 0.3).astype(np.float32)
    st100 = tree_stats(xs)
    st = np.ascontiguousarray(np.tile(st100, ((M + 99) // 100, 1, 1))[:M])
    oc = N // 2 if epi == EPI_SWIGLU_LN else N
    out = _consumer(api, dt, epi, A, W, s, c, st, oc, qcols=N // 4, qscale=0.18)
    ref = _consumer_ref(dt, epi, X, W, s, c, st100, N // 4, 0.18)
    tol = 2e-3 if dt == F16 else 1.6e-2
    err = np.abs(out[:100] - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= tol, (err.max(), np.unravel_index(err.argmax(), err.shape))
    small = _consumer(api, dt, epi, X, W, s, c, st100, oc, qcols=N // 4, qscale=0.18)
    assert np.array_equal(small, out[:100])
    last = M - 100 - M % 100
    for a in ((M // 200) * 100, last):
        assert np.array_equal(out[a:a + 100], out[:100]), a


def test_ln_consumer_plans_cover_gemm4_and_small(api):
    """The cases above really run the kernels they are meant to compare."""
    plans = {(M, N, K): api.gemm_plan(F16, EPI_GELU_LN, M, N, K) for M, N, K in _CONSUMER_CASES}
    assert "gemm4_mixed" in plans[(41100, 1024, 384)] or "gemm4<256>" in plans[(41100, 1024, 384)]
    assert "gemm4<256>" in plans[(9300, 2048, 1024)] and "small<" in plans[(9300, 2048, 1024)]
    assert "gemm4_short" in plans[(1374, 3072, 1024)] and "gemm4_short" in plans[(1374, 4096, 1536)]
    assert plans[(261, 1536, 384)].startswith("small<") and plans[(300, 384, 128)].startswith("small<")


import sys  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_plan_cases import LN_COVERAGE_CASES  # noqa: E402


@pytest.mark.parametrize("case", LN_COVERAGE_CASES, ids=lambda c: "dt%d-epi%d-%dx%dx%d" % c)
def test_ln_plan_coverage_case_bits(api, case):
    """One problem per (kernel, LN epilogue, dtype) the dispatcher can reach for the DINOv2 family with the fold on (the list is generated
    from the library's own plan query; tests/test_gemm_plans.py checks it for completeness on the CPU).  Producers: x = the plain residual
    epilogue's bits, xg and the row statistics exact functions of x.  Consumers: every 100-row copy carries the bits of a 100-row launch."""
    dt, epi, M, N, K = case
    rng = np.random.default_rng(M + 3 * N + 7 * K + 11 * epi + dt)
    X = _round(rng.standard_normal((100, K)), dt)
    A = np.ascontiguousarray(np.tile(X, ((M + 99) // 100, 1))[:M])
    W = _round(rng.standard_normal((N, K)) * 0.05, dt)
    last = M - 100 - M % 100 if M >= 200 else 0
    if epi == EPI_RESID_LN:
        bias, ls = rng.standard_normal(N).astype(np.float32), (rng.standard_normal(N) * 0.3).astype(np.float32)
        gamma = (rng.standard_normal(N) * 0.3 + 1).astype(np.float32)
        x0 = np.ascontiguousarray(np.tile((rng.standard_normal((100, N)) * 2 + 0.4).astype(np.float32), ((M + 99) // 100, 1))[:M])
        x, xg, st = _resid_ln(api, dt, A, W, bias, ls, gamma, x0)
        ref = x0.copy()
        assert api.lib().dinov2_hip_op_gemm(dt, EPI_RESID, _p(A), _p(W), _p(bias), _p(ls), N, _p(ref), M, N, M, N, K, 0, 0, 0, 0, 1.0) == 0
        assert np.array_equal(x, ref)
        assert np.array_equal(xg, _round(x * gamma, dt))
        assert np.array_equal(st, tree_stats(x))
        if M >= 200:
            assert np.array_equal(st[last:last + 100], st[:100]) and np.array_equal(xg[last:last + 100], xg[:100])
        return
    s, c = (rng.standard_normal(N) * 0.5).astype(np.float32), rng.standard_normal(N).astype(np.float32)
    st100 = tree_stats((rng.standard_normal((100, K)) * 2 + 0.3).astype(np.float32))
    st = np.ascontiguousarray(np.tile(st100, ((M + 99) // 100, 1, 1))[:M])
    oc = N // 2 if epi == EPI_SWIGLU_LN else N
    out = _consumer(api, dt, epi, A, W, s, c, st, oc, qcols=N // 3, qscale=0.18)
    small = _consumer(api, dt, epi, X, W, s, c, st100, oc, qcols=N // 3, qscale=0.18)
    assert np.isfinite(out).all()
    assert np.array_equal(small, out[:100])
    if M >= 200:
        assert np.array_equal(out[last:last + 100], out[:100])


# ---- end to end ---------------------------------------------------------------------------------------------------------------------------
def _rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("name", ["tiny_gelu_noreg", "tiny_gelu_reg4", "tiny_swiglu_reg4"])
def test_fold_on_fixtures_vs_oracle_and_unfolded(api, golden_dir, name):
    gguf = os.path.join(golden_dir, name + ".gguf")
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    img = gold["img_56x84"]
    on = api.Session(api.Model(gguf, classify=True, ln_fold=1)).predict(img[None], classify=True)
    off = api.Session(api.Model(gguf, classify=True, ln_fold=-1)).predict(img[None], classify=True)
    exp = OracleModel(gguf).forward(img, classify=True)
    for got in (on, off):
        assert _rel(got["logits"][0], exp["logits"]) <= 1e-3
        assert _rel(got["patch_tokens"][0], exp["patch_tokens"]) <= 5e-3
    assert not np.array_equal(on["logits"], off["logits"])  # (the option really switches paths)
    # per layer: the residual stream of the folded path stays on the oracle's
    sess = api.Session(api.Model(gguf, classify=True, ln_fold=1))
    hid = OracleModel(gguf).forward(img, classify=False, hidden=True)["hidden"]
    for layer in range(hid.shape[0]):
        assert _rel(sess.debug_hidden(img[None], layer)[0], hid[layer]) <= 5e-3, layer


def test_fold_batch_invariance(api, pkg, tmp_path):
    """With the fold on, an image's bits still do not depend on the batch it arrives in (ViT-S at 224: batch 1 runs few-tile plans, batch 24
    persistent 256-row tiles with a column split, N = 384 = 256 + 128)."""
    path = str(tmp_path / "small.gguf")
    pkg.synth.write_synthetic_gguf(path, "small", registers=4, num_classes=100, seed=3)
    imgs = pkg.synth.synthetic_images(24, 224, 224, seed=5)
    sess = api.Session(api.Model(path, classify=True, ln_fold=1))
    big = sess.predict(imgs, classify=True)
    for b in (0, 23):
        one = sess.predict(imgs[b:b + 1], classify=True)
        assert np.array_equal(one["logits"][0], big["logits"][b]) and np.array_equal(one["patch_tokens"][0], big["patch_tokens"][b])
    three = sess.predict(imgs[5:8], classify=True)
    assert np.array_equal(three["logits"], big["logits"][5:8])
