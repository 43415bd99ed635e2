"""CPU suite, part 1: the oracle (oracle/dinov2_oracle.c) pinned against the committed golden vectors.

The reference holds no tests or golden vectors and cannot be built offline, so the fixtures
come from HuggingFace DINOv2 (tests/golden/make_golden.py); the oracle must match them to f32 round-off when its
ggml-specific roundings are switched off, and stay within the documented rounding envelope when they are on."""
import json
import os

import numpy as np
import pytest

from oracle.oracle import OracleModel, bgr_hwc_to_rgb_chw

FIXTURES = ["tiny_gelu_noreg", "tiny_gelu_reg4", "tiny_swiglu_reg4"]


@pytest.fixture(scope="module")
def manifest(golden_dir):
    return json.load(open(os.path.join(golden_dir, "manifest.json")))


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_matches_hf_in_f32_mode(golden_dir, manifest, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    m = OracleModel(os.path.join(golden_dir, name + ".gguf"), act_round=0, gelu_f16_lut=False)
    m.set(conv_round=0)
    for key in manifest[name]["sizes"]:
        o = m.forward(g[f"img_{key}"], classify=True, hidden=True)
        assert np.abs(o["hidden"] - g[f"hidden_{key}"]).max() < 2e-5, key
        assert np.abs(o["cls"] - g[f"final_{key}"][0]).max() < 2e-5
        assert np.abs(o["patch_tokens"] - g[f"final_{key}"][1:]).max() < 2e-5
        assert np.abs(o["logits"] - g[f"logits_{key}"]).max() < 5e-6
        assert np.abs(o["probs"] - g[f"probs_{key}"]).max() < 1e-6


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_pos_embed_interpolation(golden_dir, manifest, name):
    """cv::resize(INTER_CUBIC) restatement == torch bicubic (align_corners=False, no antialias) to f32 round-off."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    m = OracleModel(os.path.join(golden_dir, name + ".gguf"))
    for key in manifest[name]["sizes"]:
        hh, ww = map(int, key.split("x"))
        pe = m.interpolate_pos_embed(hh // 14, ww // 14)
        assert np.abs(pe - g[f"pos_{key}"]).max() < 2e-6, key


def test_pos_embed_identity_on_equal_patch_count(golden_dir):
    """The reference returns the table untouched whenever the patch COUNT matches (dinov2.cpp:176-179)."""
    m = OracleModel(os.path.join(golden_dir, "tiny_gelu_noreg.gguf"))
    assert np.array_equal(m.interpolate_pos_embed(5, 5), m.pos.reshape(26, 128))


@pytest.mark.parametrize("name", FIXTURES)
def test_ggml_roundings_stay_in_envelope(golden_dir, name):
    """ggml numerics (f16 activation rounding, f16 GELU LUT) move logits by O(1e-4), not more."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    m = OracleModel(os.path.join(golden_dir, name + ".gguf"))
    o = m.forward(g["img_70x70"], classify=True)
    assert 0 < np.abs(o["logits"] - g["logits_70x70"]).max() < 2e-3


def test_head_quirks(golden_dir):
    """pool = sum over patch tokens INCLUDING registers / 25 (reference) vs mean over patch tokens only (HF)."""
    g = np.load(os.path.join(golden_dir, "tiny_gelu_reg4.npz"))
    p = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    hf = OracleModel(p, act_round=0, gelu_f16_lut=False, pool_const_divisor=False, pool_includes_registers=False)
    hf.set(conv_round=0)
    o = hf.forward(g["img_56x84"], classify=True)
    assert np.abs(o["logits"] - g["hf_logits_56x84"]).max() < 5e-6
    ref = OracleModel(p, act_round=0, gelu_f16_lut=False)
    ref.set(conv_round=0)
    assert np.abs(ref.forward(g["img_56x84"], classify=True)["logits"] - g["hf_logits_56x84"]).max() > 1e-3


def test_feature_vs_classify_views(golden_dir):
    """patch_tokens: features strip CLS + registers; classify keeps registers (dinov2.cpp:770-789)."""
    g = np.load(os.path.join(golden_dir, "tiny_gelu_reg4.npz"))
    m = OracleModel(os.path.join(golden_dir, "tiny_gelu_reg4.gguf"))
    a = m.forward(g["img_70x70"], classify=False)["patch_tokens"]
    b = m.forward(g["img_70x70"], classify=True)["patch_tokens"]
    assert a.shape == (25, 128) and b.shape == (29, 128)
    assert np.array_equal(a, b[4:])


def test_bgr_repack():
    x = np.arange(2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3)
    y = bgr_hwc_to_rgb_chw(x)
    assert y.shape == (3, 2, 3) and y[0, 1, 2] == x[1, 2, 2] and y[2, 0, 0] == x[0, 0, 0]


def test_oracle_thread_count_invariance(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_swiglu_reg4.npz"))
    m = OracleModel(os.path.join(golden_dir, "tiny_swiglu_reg4.gguf"))
    a = m.forward(g["img_42x42"], classify=True, nthreads=1)["logits"]
    b = m.forward(g["img_42x42"], classify=True, nthreads=4)["logits"]
    assert np.array_equal(a, b)
