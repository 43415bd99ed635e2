"""End-to-end parity of the HIP forward (through the C-ABI) against the CPU oracle and the committed HF golden
fixtures.  Tolerances (stated here, derived in DESIGN.md "Numerics"):

  f16 compute, logits      max|d| <= 1e-3 * max(1, max|logit|)   (BASELINE.json north_star: "logits within 1e-3")
  f16 compute, tokens      max|d| <= 5e-3 * max(1, max|token|)   (final-LN features are O(1..10))
  bf16 compute             8x the f16 bounds (3 fewer mantissa bits)
  quantised GGUF           vs oracle in "dequant" mode (same contract as the HIP path): f16 bounds;
                           vs oracle in "ggml" mode (q8_0 activation quantisation): 2e-2 (the HIP path is the MORE
                           accurate of the two)
"""
import json
import os

import numpy as np
import pytest

from oracle.oracle import OracleModel, bgr_hwc_to_rgb_chw

pytestmark = pytest.mark.gpu

FIXTURES = ["tiny_gelu_noreg", "tiny_gelu_reg4", "tiny_swiglu_reg4"]


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.fixture(scope="module")
def manifest(golden_dir):
    return json.load(open(os.path.join(golden_dir, "manifest.json")))


@pytest.mark.parametrize("name", FIXTURES)
def test_golden_fixture_vs_oracle_and_hf(api, golden_dir, manifest, name):
    """Every committed fixture, every resolution (identity / non-square / downsampled pos-embed), classify + features."""
    gguf = os.path.join(golden_dir, name + ".gguf")
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    model = api.Model(gguf, classify=True)
    sess = api.Session(model)
    ora = OracleModel(gguf)  # ggml CPU numerics: f16 activation rounding, f32 attention, f16 GELU LUT
    R = manifest[name]["registers"]
    for key in manifest[name]["sizes"]:
        img = gold[f"img_{key}"]
        exp = ora.forward(img, classify=True)
        got = sess.predict(img[None], classify=True, topk=3)
        assert _rel(got["logits"][0], exp["logits"]) <= 1e-3, key
        assert np.abs(got["probs"][0] - exp["probs"]).max() <= 1e-3
        assert _rel(got["cls"][0], exp["cls"]) <= 5e-3
        assert got["patch_tokens"].shape[1] == exp["patch_tokens"].shape[0]  # registers included when classifying
        assert _rel(got["patch_tokens"][0], exp["patch_tokens"]) <= 5e-3
        assert got["topk_ids"][0, 0] == int(np.argmax(exp["probs"]))
        np.testing.assert_allclose(got["probs"][0].sum(), 1.0, atol=1e-5)
        # and against HuggingFace (independent implementation, pure f32): looser, includes ggml-style roundings
        assert _rel(got["logits"][0], gold[f"logits_{key}"]) <= 3e-3
        # feature path strips CLS and registers (dinov2.cpp:770-789)
        feat = sess.predict(img[None], classify=False)
        expf = ora.forward(img, classify=False)
        assert feat["patch_tokens"].shape[1] == exp["patch_tokens"].shape[0] - R
        assert _rel(feat["patch_tokens"][0], expf["patch_tokens"]) <= 5e-3


@pytest.mark.parametrize("name", FIXTURES)
def test_hidden_states_per_layer(api, golden_dir, name):
    """Token stream after the embeddings and after every layer vs the oracle (localises a wrong kernel)."""
    gguf = os.path.join(golden_dir, name + ".gguf")
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    model = api.Model(gguf, classify=False)
    sess = api.Session(model)
    ora = OracleModel(gguf)
    img = gold["img_56x84"]
    exp = ora.forward(img, hidden=True)["hidden"]
    for layer in range(exp.shape[0]):
        got = sess.debug_hidden(img[None], layer)[0]
        assert _rel(got, exp[layer]) <= (1e-5 if layer == 0 else 3e-3), f"layer {layer}"


def test_batch_equals_independent_images(api, golden_dir):
    """B images == B independent batch-1 forwards (the reference is strictly batch 1, dinov2.cpp:630): bit-exact."""
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    model = api.Model(gguf, classify=True)
    sess = api.Session(model)
    rng = np.random.default_rng(0)
    imgs = rng.standard_normal((5, 3, 70, 98)).astype(np.float32)
    full = sess.predict(imgs, classify=True)
    for b in range(5):
        one = sess.predict(imgs[b:b + 1], classify=True)
        assert np.array_equal(one["logits"][0], full["logits"][b])
        assert np.array_equal(one["patch_tokens"][0], full["patch_tokens"][b])


def test_bgr_hwc_layout_matches_reference_repack(api, golden_dir):
    """cv::Mat-style BGR interleaved input == RGB planar input after the dinov2.cpp:914-931 repack."""
    gguf = os.path.join(golden_dir, "tiny_gelu_noreg.gguf")
    model = api.Model(gguf, classify=True)
    sess = api.Session(model)
    rng = np.random.default_rng(1)
    bgr = rng.standard_normal((2, 56, 70, 3)).astype(np.float32)
    a = sess.predict(bgr, classify=True, layout=api.BGR_HWC)
    b = sess.predict(np.stack([bgr_hwc_to_rgb_chw(x) for x in bgr]), classify=True, layout=api.RGB_CHW)
    assert np.array_equal(a["logits"], b["logits"])


def test_pool_quirk_flags(api, golden_dir):
    """HF-style pooling (mean over patch tokens only) when both reference quirks are switched off."""
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    gold = np.load(os.path.join(golden_dir, "tiny_gelu_reg4.npz"))
    model = api.Model(gguf, classify=True, pool_const_divisor=False, pool_includes_registers=False)
    got = api.Session(model).predict(gold["img_56x84"][None], classify=True)
    assert _rel(got["logits"][0], gold["hf_logits_56x84"]) <= 3e-3


def test_mirror_api_prints_and_returns(api, golden_dir, capsys):
    """dino_model_load / dino_predict mirror: same call shapes as inference.cpp:45-65."""
    p = api.dino_params(classify=True, topk=3, model=os.path.join(golden_dir, "tiny_gelu_reg4.gguf"))
    m = api.dino_model()
    assert api.dino_model_load((70, 70), p.model, m, p)
    assert m.hparams.hidden_size == 128 and m.hparams.num_register_tokens == 4
    img = np.random.default_rng(2).standard_normal((70, 70, 3)).astype(np.float32)
    out = api.dino_predict(m, img, p)
    assert out is not None and len(out.preds) == 3
    assert capsys.readouterr().out.count(" > label_") == 3
    p2 = api.dino_params(classify=False)
    feats = api.dino_predict(m, img, p2)
    assert feats.patch_tokens.shape == (25, 128)
    assert not api.dino_model_load((70, 70), "/nonexistent.gguf", api.dino_model(), p)


def test_error_paths(api, golden_dir):
    model = api.Model(os.path.join(golden_dir, "tiny_gelu_reg4.gguf"), classify=False)
    sess = api.Session(model)
    with pytest.raises(api.DinoError) as e:
        sess.predict(np.zeros((1, 3, 60, 70), np.float32))  # 60 is not a multiple of 14
    assert e.value.status == 4
    with pytest.raises(api.DinoError) as e:
        sess.predict(np.zeros((1, 3, 70, 70), np.float32), classify=True)  # loaded without the head
    assert e.value.status == 6


@pytest.mark.parametrize("dtype_name,scale", [("f16", 1.0), ("bf16", 8.0)])
@pytest.mark.parametrize("cfg", ["small", "tiny-swiglu"])
def test_synthetic_model_vs_oracle(api, pkg, tmp_path, cfg, dtype_name, scale):
    """Seeded synthetic checkpoints in the converter's schema (no pretrained weights exist offline): ViT-S/14 at
    224x224 (BASELINE config 1 shape: T = 257, pos-embed 37 -> 16 bicubic) and a SwiGLU model, f16 and bf16 compute."""
    path = str(tmp_path / f"{cfg}.gguf")
    registers = 0 if cfg == "small" else 4
    pkg.synth.write_synthetic_gguf(path, cfg, registers=registers, num_classes=1000 if cfg == "small" else 16, seed=7)
    hw = (224, 224) if cfg == "small" else (84, 56)
    imgs = pkg.synth.synthetic_images(2, *hw, seed=3)
    dt = api.F16 if dtype_name == "f16" else api.BF16
    got = api.Session(api.Model(path, dtype=dt, classify=True)).predict(imgs, classify=True)
    ora = OracleModel(path)
    for b in range(2):
        exp = ora.forward(imgs[b], classify=True)
        assert _rel(got["logits"][b], exp["logits"]) <= 1e-3 * scale
        assert _rel(got["patch_tokens"][b], exp["patch_tokens"]) <= 5e-3 * scale
        assert np.abs(got["probs"][b] - exp["probs"]).max() <= 1e-3 * scale


@pytest.mark.parametrize("wtype", ["q8_0", "q4_0", "q4_1", "q5_0", "q5_1"])
def test_quantised_gguf_dequant_on_load(api, pkg, tmp_path, wtype):
    """BASELINE config 5: quantised GGUF -> dequant-on-load -> f16 MFMA path; logits vs the CPU reference semantics."""
    path = str(tmp_path / f"m_{wtype}.gguf")
    pkg.synth.write_synthetic_gguf(path, "tiny", registers=4, num_classes=32, seed=5, wtype=wtype)
    imgs = pkg.synth.synthetic_images(1, 70, 70, seed=4)
    got = api.Session(api.Model(path, classify=True)).predict(imgs, classify=True)
    same_contract = OracleModel(path, quant_mode="dequant").forward(imgs[0], classify=True)
    ggml_contract = OracleModel(path, quant_mode="ggml").forward(imgs[0], classify=True)
    assert _rel(got["logits"][0], same_contract["logits"]) <= 1e-3
    assert _rel(got["logits"][0], ggml_contract["logits"]) <= 2e-2


def test_full_size_large_properties(api, pkg, tmp_path):
    """ViT-L/14 + 4 registers at 518x518 (the headline shape, 4 layers to keep the oracle in seconds): parity of the
    first image against the oracle, plus size-independent properties at batch 3: permutation equivariance over the
    batch, probabilities sum to 1, finite outputs."""
    path = str(tmp_path / "large4.gguf")
    pkg.synth.write_synthetic_gguf(path, "large", registers=4, num_classes=1000, seed=42, layers=4)
    imgs = pkg.synth.synthetic_images(3, 518, 518, seed=42)
    sess = api.Session(api.Model(path, classify=True))
    got = sess.predict(imgs, classify=True)
    assert got["patch_tokens"].shape == (3, 4 + 1369, 1024)
    exp = OracleModel(path).forward(imgs[0], classify=True)
    assert _rel(got["logits"][0], exp["logits"]) <= 1e-3
    assert _rel(got["patch_tokens"][0], exp["patch_tokens"]) <= 5e-3
    perm = sess.predict(imgs[::-1].copy(), classify=True)
    assert np.array_equal(perm["logits"][::-1], got["logits"])
    np.testing.assert_allclose(got["probs"].sum(-1), 1.0, atol=1e-5)
    assert np.isfinite(got["patch_tokens"]).all()


@pytest.mark.parametrize("name", ["tiny_gelu_reg4", "tiny_swiglu_reg4"])
@pytest.mark.parametrize("h,w", [(14, 14), (14, 70), (154, 98), (406, 406), (490, 854)])
def test_shape_edge_cases(api, golden_dir, name, h, w):
    """Extremes of the token count: a single patch (T = 6: one partial key tile, one partial query block), a 1 x 5 strip,
    a non-square 11 x 7 grid, and 29 x 29 = 841 patches (T = 846: 14 key tiles, 7 query blocks, pos-embed upsampled from
    5 x 5), and the realtime demo's 854 x 480 frame rounded to patches (35 x 61 = 2 135 patches) -- classify and features
    against the oracle."""
    gguf = os.path.join(golden_dir, name + ".gguf")
    sess = api.Session(api.Model(gguf, classify=True))
    ora = OracleModel(gguf)
    img = np.random.default_rng(h * 1000 + w).standard_normal((3, h, w)).astype(np.float32)
    for classify in (True, False):
        exp = ora.forward(img, classify=classify)
        got = sess.predict(img[None], classify=classify)
        assert got["patch_tokens"].shape[1:] == exp["patch_tokens"].shape
        assert _rel(got["patch_tokens"][0], exp["patch_tokens"]) <= 5e-3
        assert _rel(got["cls"][0], exp["cls"]) <= 5e-3
        if classify:
            assert _rel(got["logits"][0], exp["logits"]) <= 1e-3
            np.testing.assert_allclose(got["probs"][0].sum(), 1.0, atol=1e-5)


def test_odd_batch_and_session_reuse_across_shapes(api, golden_dir):
    """Batch 37 (ragged GEMM row tiles, several attention grid rows), then the same session at other shapes and back:
    the workspace is re-carved and the cached interpolated pos-embed is refreshed each time."""
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    sess = api.Session(api.Model(gguf, classify=True))
    ora = OracleModel(gguf)
    rng = np.random.default_rng(37)
    big = rng.standard_normal((37, 3, 70, 70)).astype(np.float32)
    got = sess.predict(big, classify=True)
    for b in (0, 17, 36):
        assert _rel(got["logits"][b], ora.forward(big[b], classify=True)["logits"]) <= 1e-3
    other = rng.standard_normal((2, 3, 98, 42)).astype(np.float32)
    g2 = sess.predict(other, classify=True)
    assert _rel(g2["logits"][1], ora.forward(other[1], classify=True)["logits"]) <= 1e-3
    again = sess.predict(big, classify=True)
    assert np.array_equal(again["logits"], got["logits"])


def test_image_result_independent_of_batch_size(api, pkg, tmp_path):
    """ViT-L/14 shapes (4 layers).  By default (dinov2_hip_load_opts.batch_invariant = 1) an image alone (batch 1: small-tile
    GEMMs, pipelined attention kernel) and the same image inside a batch of 24 (persistent 256x256 GEMM, high-occupancy attention
    kernel) give identical bits -- also when the batch is cut into chunks with a remainder of one image.  The load option
    batch_invariant = 0 (once an opt-in split-K mode) is still accepted and changes nothing."""
    path = str(tmp_path / "large4b.gguf")
    pkg.synth.write_synthetic_gguf(path, "large", registers=4, num_classes=1000, seed=7, layers=4)
    imgs = pkg.synth.synthetic_images(24, 518, 518, seed=7)
    sess = api.Session(api.Model(path, classify=True))  # the default IS batch-invariant
    full = sess.predict(imgs, classify=True)
    for b in (0, 23):
        one = sess.predict(imgs[b:b + 1], classify=True)
        assert np.array_equal(one["logits"][0], full["logits"][b])
        assert np.array_equal(one["patch_tokens"][0], full["patch_tokens"][b])
    # a chunked predict whose last chunk holds ONE image (the remainder chunk must not get other bits)
    os.environ["DINOV2_HIP_MAX_CHUNK"] = "23"
    try:
        chunked = api.Session(api.Model(path, classify=True)).predict(imgs, classify=True)
    finally:
        del os.environ["DINOV2_HIP_MAX_CHUNK"]
    assert np.array_equal(chunked["logits"], full["logits"]) and np.array_equal(chunked["patch_tokens"], full["patch_tokens"])
    fast = api.Session(api.Model(path, classify=True, batch_invariant=False))
    one = fast.predict(imgs[23:24], classify=True)
    assert np.array_equal(one["logits"][0], full["logits"][23]) and np.array_equal(one["patch_tokens"][0], full["patch_tokens"][23])


@pytest.mark.parametrize("model,layers", [("large", 3), ("small", 4), ("base", 2)])
def test_tiny_batches_at_224_few_tile_plans(api, pkg, tmp_path, model, layers):
    """The reference's own regime, 224 x 224, batch 1 (T = 257 + 4 rows): every GEMM takes a few-tile plan of csrc/gemm.hip (32 x 64 and
    64 x 64 tiles, fewer workgroups than CUs).  Against the oracle to the stated bound; an image alone, in a batch of 2, 3 (M = 783) and 9
    (larger tiles) gets the same bits; bf16 compute too; 50 repeats as a race screen; debug_hidden agrees with a 1-layer-shorter stop."""
    path = str(tmp_path / f"{model}{layers}.gguf")
    pkg.synth.write_synthetic_gguf(path, model, registers=4, num_classes=1000, seed=13, layers=layers)
    imgs = pkg.synth.synthetic_images(9, 224, 224, seed=13)
    sess = api.Session(api.Model(path, classify=True))
    exp = OracleModel(path).forward(imgs[0], classify=True)
    one = sess.predict(imgs[:1], classify=True)
    assert _rel(one["logits"][0], exp["logits"]) <= 1e-3 and _rel(one["patch_tokens"][0], exp["patch_tokens"]) <= 5e-3
    for B in (2, 3, 9):
        got = sess.predict(imgs[:B], classify=True)
        assert np.array_equal(got["logits"][0], one["logits"][0]) and np.array_equal(got["patch_tokens"][0], one["patch_tokens"][0]), B
    feats = sess.predict(imgs[:1], classify=False)
    assert _rel(feats["patch_tokens"][0], OracleModel(path).forward(imgs[0], classify=False)["patch_tokens"]) <= 5e-3
    sb = api.Session(api.Model(path, classify=True, dtype=api.BF16))
    b1, b9 = sb.predict(imgs[:1], classify=True), sb.predict(imgs, classify=True)
    assert np.array_equal(b1["logits"][0], b9["logits"][0]) and np.array_equal(b1["patch_tokens"][0], b9["patch_tokens"][0])
    assert _rel(b1["logits"][0], exp["logits"]) <= 8e-3 and _rel(b1["patch_tokens"][0], exp["patch_tokens"]) <= 4e-2
    first = sess.predict(imgs[:2], classify=True)
    for _ in range(50):
        again = sess.predict(imgs[:2], classify=True)
        assert np.array_equal(first["logits"], again["logits"]) and np.array_equal(first["patch_tokens"], again["patch_tokens"])


@pytest.mark.parametrize("dtype_name,scale", [("f16", 1.0), ("bf16", 8.0)])
def test_full_size_giant_swiglu(api, pkg, tmp_path, dtype_name, scale):
    """BASELINE config 4 shapes (ViT-g/14: H = 1536, 24 heads, SwiGLU 8192 -> 4096; 2 of its 40 layers to keep the oracle
    in seconds) at 518x518, batch 6 so that the persistent 256x256 kernel and its SwiGLU epilogue are the ones running:
    first and last image against the oracle, f16 and bf16 compute."""
    path = str(tmp_path / "giant2.gguf")
    pkg.synth.write_synthetic_gguf(path, "giant", registers=4, num_classes=1000, seed=13, layers=2)
    imgs = pkg.synth.synthetic_images(6, 518, 518, seed=13)
    dt = api.F16 if dtype_name == "f16" else api.BF16
    got = api.Session(api.Model(path, dtype=dt, classify=True)).predict(imgs, classify=True)
    ora = OracleModel(path)
    for b in (0, 5):
        exp = ora.forward(imgs[b], classify=True)
        assert _rel(got["logits"][b], exp["logits"]) <= 1e-3 * scale
        assert _rel(got["patch_tokens"][b], exp["patch_tokens"]) <= 5e-3 * scale
    np.testing.assert_allclose(got["probs"].sum(-1), 1.0, atol=1e-5)


def test_graph_replay_matches_eager(golden_dir):
    """DINOV2_HIP_GRAPHS=1 (read once per process, hence the subprocess): eager, capture, replay and replay of the same
    device-resident forward give identical bits; a different shape in between re-uploads its pos-embed and the replayed
    graph of the first shape still sees the right one."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from importlib import import_module
from __graft_entry__ import PKG_NAME, load_package
load_package(); api = import_module(PKG_NAME + ".api")
sess = api.Session(api.Model(sys.argv[1], classify=True))
g = torch.Generator().manual_seed(1)
a = torch.randn(3, 3, 70, 98, generator=g).cuda(); b = torch.randn(2, 3, 42, 56, generator=g).cuda()
def run(x):
    B = x.shape[0]
    lo = torch.empty(B, 10, device="cuda"); pr = torch.empty_like(lo)
    sess.predict_device(x.data_ptr(), B, x.shape[2], x.shape[3], classify=True, layout=api.RGB_CHW, logits_ptr=lo.data_ptr(), probs_ptr=pr.data_ptr())
    sess.sync()
    return lo.cpu().numpy()
r = [run(a), run(a), run(a), run(b), run(a), run(b), run(b), run(a)]
assert all(np.array_equal(r[0], r[i]) for i in (1, 2, 4, 7)), "graph replay differs from eager"
assert np.array_equal(r[3], r[5]) and np.array_equal(r[3], r[6])
print("GRAPH_OK")
'''
    env = dict(os.environ, DINOV2_HIP_GRAPHS="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code, os.path.join(golden_dir, "tiny_gelu_reg4.gguf")], cwd=root, env=env,
                         capture_output=True, text=True, timeout=600)
    assert "GRAPH_OK" in out.stdout, out.stdout + out.stderr


def test_two_sessions_on_two_threads(api, golden_dir):
    """include/dinov2_hip.h: a model is immutable and may be shared by any number of sessions; one session per host thread.
    Two threads, each with its own session (own stream + workspace), run different inputs concurrently (ctypes releases the
    GIL inside the C calls); every result must equal the single-threaded one bit for bit."""
    import threading
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    model = api.Model(gguf, classify=True)
    rng = np.random.default_rng(21)
    inputs = [rng.standard_normal((3, 3, 70, 98)).astype(np.float32), rng.standard_normal((2, 3, 154, 70)).astype(np.float32)]
    ref = [api.Session(model).predict(x, classify=True)["logits"] for x in inputs]
    out = [[None] * 20, [None] * 20]
    errs = []

    def worker(k):
        try:
            sess = api.Session(model)
            for i in range(20):
                out[k][i] = sess.predict(inputs[k], classify=True)["logits"]
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for k in range(2):
        for i in range(20):
            assert np.array_equal(out[k][i], ref[k]), (k, i)


@pytest.mark.parametrize("name", FIXTURES)
def test_random_shapes_vs_oracle(api, golden_dir, name):
    """Twenty random (batch, height, width, classify) per fixture -- every patch-grid shape from 1 x 1 to 25 x 25, one session
    re-carving its workspace each time -- one image of each batch against the oracle."""
    gguf = os.path.join(golden_dir, name + ".gguf")
    sess = api.Session(api.Model(gguf, classify=True))
    ora = OracleModel(gguf)
    rng = np.random.default_rng(sum(map(ord, name)))
    for _ in range(20):
        B, h, w = int(rng.integers(1, 10)), 14 * int(rng.integers(1, 26)), 14 * int(rng.integers(1, 26))
        classify = bool(rng.integers(0, 2))
        x = rng.standard_normal((B, 3, h, w)).astype(np.float32)
        got = sess.predict(x, classify=classify)
        b = int(rng.integers(0, B))
        exp = ora.forward(x[b], classify=classify)
        assert np.isfinite(got["patch_tokens"]).all(), (B, h, w)
        assert _rel(got["patch_tokens"][b], exp["patch_tokens"]) <= 5e-3, (B, h, w, classify)
        if classify:
            assert _rel(got["logits"][b], exp["logits"]) <= 1e-3, (B, h, w)


@pytest.mark.parametrize("P,H", [(256, 384), (1369, 1024), (2170, 1536), (37, 32), (700, 100), (12, 8)])
def test_pca3_matches_svd(api, golden_dir, P, H):
    """dinov2_hip_pca3 (device means, covariance on the matrix cores, block iteration, projection; host Rayleigh-Ritz) against numpy's SVD of the
    centred tokens -- the PCA of inference.cpp:76-81.  Tokens get a clear three-direction structure on top of noise, as patch
    tokens have.  Tolerances: |cos| between matching components >= 0.999 (the covariance is accumulated from f16-rounded
    centred tokens), projections within 1% of the largest projection."""
    sess = api.Session(api.Model(os.path.join(golden_dir, "tiny_gelu_reg4.gguf"), classify=False))
    rng = np.random.default_rng(P * 7 + H)
    basis = np.linalg.qr(rng.standard_normal((H, 3)))[0].T                                  # [3, H] orthonormal
    x = (rng.standard_normal((P, 3)) * np.array([9.0, 5.0, 2.5])) @ basis + 0.3 * rng.standard_normal((P, H)) + 4.0
    x = x.astype(np.float32)
    comp, mean, proj = sess.pca3(x)
    xc = x.astype(np.float64) - x.mean(0, dtype=np.float64)
    _, _, vt = np.linalg.svd(xc, full_matrices=False)
    ref = vt[:3]
    for c in ref:
        if c[np.abs(c).argmax()] < 0:
            c *= -1
    np.testing.assert_allclose(mean, x.mean(0, dtype=np.float64), atol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(comp, axis=1), 1.0, atol=1e-5)
    for k in range(3):
        assert abs(float(comp[k].astype(np.float64) @ ref[k])) >= 0.999, k
        assert comp[k][np.abs(comp[k]).argmax()] > 0
    want = xc @ comp.astype(np.float64).T                                                     # projection is self-consistent
    assert np.abs(proj - want).max() <= 1e-3 * max(1.0, np.abs(want).max())
    assert np.abs(np.abs(proj) - np.abs(xc @ ref.T)).max() <= 1e-2 * np.abs(xc @ ref.T).max()
    with pytest.raises(api.DinoError):
        sess.pca3(np.zeros((2, 8), np.float32))
    # degenerate input (all tokens equal: zero covariance) stays finite: zero projection, like cv::PCA on constant data
    comp0, mean0, proj0 = sess.pca3(np.full((P, H), 2.5, np.float32))
    assert np.isfinite(comp0).all() and np.array_equal(proj0, np.zeros_like(proj0)) and np.allclose(mean0, 2.5)


def test_pca3_on_resident_tokens(api, golden_dir):
    """tokens = NULL: the PCA of the patch tokens the last predict left on the device equals, bit for bit, the PCA of the same
    tokens handed over from the host; a shape that is not theirs is refused."""
    sess = api.Session(api.Model(os.path.join(golden_dir, "tiny_gelu_reg4.gguf"), classify=False))
    img = np.random.default_rng(3).standard_normal((2, 3, 84, 112)).astype(np.float32)
    tok = sess.predict(img, classify=False)["patch_tokens"]
    P, H = tok.shape[1:]
    a = sess.pca3(None, (P, H))
    b = sess.pca3(tok[0])
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    with pytest.raises(api.DinoError):
        sess.pca3(None, (P + 1, H))
    again = sess.pca3(tok[0])
    assert all(np.array_equal(x, y) for x, y in zip(again, b))  # deterministic


def test_large_batches_are_split_transparently(api, golden_dir, monkeypatch):
    """dinov2_hip_predict splits a batch whose widest activation buffer would pass 2^31 bytes (the kernels use 32-bit staging
    offsets).  DINOV2_HIP_MAX_CHUNK forces the split at a small size: same bits as the unsplit call for every output, for host
    f32 and raw 8-bit inputs, classify and features; and the PCA of the resident tokens still refers to image 0 of the call."""
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    sess = api.Session(api.Model(gguf, classify=True))
    rng = np.random.default_rng(11)
    img = rng.standard_normal((11, 3, 56, 70)).astype(np.float32)
    raw = rng.integers(0, 256, (7, 61, 83, 3), dtype=np.uint8)
    want = ("cls", "patch_tokens", "logits", "probs")
    ref_c = sess.predict(img, classify=True, topk=3, want=want)
    ref_f = sess.predict(img, classify=False, want=("cls", "patch_tokens"))
    ref_r = sess.predict(raw, classify=False, layout=api.U8_BGR_HWC, want=("patch_tokens",))
    monkeypatch.setenv("DINOV2_HIP_MAX_CHUNK", "4")
    got_c = sess.predict(img, classify=True, topk=3, want=want)
    got_f = sess.predict(img, classify=False, want=("cls", "patch_tokens"))
    P, H = got_f["patch_tokens"].shape[1:]
    pca_resident = sess.pca3(None, (P, H))
    got_r = sess.predict(raw, classify=False, layout=api.U8_BGR_HWC, want=("patch_tokens",))
    for ref, got in ((ref_c, got_c), (ref_f, got_f), (ref_r, got_r)):
        assert ref.keys() == got.keys()
        for k in ref:
            assert np.array_equal(ref[k], got[k]), k
    for a, b in zip(pca_resident, sess.pca3(ref_f["patch_tokens"][0])):
        assert np.array_equal(a, b)
