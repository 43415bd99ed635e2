"""Parity under TRAINED-MODEL statistics ("next round" item 1).

Every other full-size check runs on i.i.d. N(0, 0.02) weights: pre-softmax scores of O(1), no outlier channels, no attention sinks.
The HIP path rounds q, k, v and the un-normalised probabilities to f16 where ggml keeps f32 (/root/reference/dinov2.cpp:527-536), and
the score error of that rounding scales with |score|; a residual stream with 100 x outlier channels is where f16 activations at the
weight matmuls lose most.  `synth.write_synthetic_gguf(trained_like=True)` builds checkpoints with those statistics (peaky heads with
|score| 30 - 60, outlier channels at > 100 x the median, registers as attention sinks, trained-scale head); these tests first ASSERT that the
regime is reached (tests/trained_stats.py, from the oracle's hidden states) and then hold the HIP path -- full depth, 518 x 518 -- to the
oracle in its ggml-default mode, in all four ggml-uncertain switch modes, and to exact arithmetic (oracle_forward_exact), logits and
tokens, absolute and relative.  Numbers go to gpurun_out/parity_r06.json (copied to profiles/r06_parity.json).
"""
import json
import os

import numpy as np
import pytest

from oracle.oracle import OracleModel
from tests.trained_stats import layer_stats

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_RESULTS = os.path.join(ROOT, "gpurun_out", "parity_r06.json")
_SWITCHES = {"ggml_default": dict(), "act_round_0": dict(act_round=0), "no_gelu_lut": dict(gelu_f16_lut=False),
             "act_round_0_no_gelu_lut": dict(act_round=0, gelu_f16_lut=False)}


def _record(name, **vals):
    try:
        os.makedirs(os.path.dirname(_RESULTS), exist_ok=True)
        cur = json.load(open(_RESULTS)) if os.path.exists(_RESULTS) else {}
        cur[name] = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in vals.items()}
        json.dump(cur, open(_RESULTS, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def _abs(a, b):
    return float(np.abs(a - b).max())


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.fixture(scope="module")
def trained(tmp_path_factory, pkg):
    root = tmp_path_factory.mktemp("trained_like")
    made = {}

    def get(model, layers=None):
        if (model, layers) not in made:
            path = str(root / f"{model}_{layers}.gguf")
            pkg.synth.write_synthetic_gguf(path, model, registers=4, num_classes=1000, seed=42, head_std=0.12, trained_like=True, layers=layers)
            made[(model, layers)] = path
        return made[(model, layers)]

    return get


def test_trained_like_regime_is_reached(pkg, trained):
    """The synthetic checkpoint really has the statistics the parity tests below are about (ViT-L/14 @518, three of its layers):
    several heads with max|score| >= 30 in every layer looked at, softmax mass on the registers >= 0.9 for at least one head, outlier
    channels at >= 50 x the median from the layer behind the one that creates them, and logits of trained-model size."""
    path = trained("large")
    img = pkg.synth.synthetic_images(1, 518, 518, seed=42)[0]
    L = 24
    lo = pkg.synth.trained_outlier_layer(L)
    stats = layer_stats(path, img, layers=[0, lo + 1, L - 1])
    rec = {}
    for st in stats:
        ms = np.array(st["max_abs_score"])
        rec[f"layer{st['layer']}"] = dict(heads_ge_30=int((ms >= 30).sum()), max_abs_score=float(ms.max()), median_head_max=float(np.median(ms)),
                                          sink_mass_max=float(max(st["sink_mass"])), outlier_ratio=st["outlier_ratio"], max_abs_x=st["max_abs_x"],
                                          mean_over_std=st["mean_over_std"])
        assert (ms >= 30).sum() >= 2 and ms.max() <= 120, st["layer"]
        assert max(st["sink_mass"]) >= 0.9, st["layer"]
    assert stats[0]["outlier_ratio"] < 3 and stats[1]["outlier_ratio"] >= 50 and stats[2]["outlier_ratio"] >= 50
    big = float(np.abs(OracleModel(path).forward(img, classify=True)["logits"]).max())
    rec["max_abs_logit"] = big
    _record("trained_like_regime_vit_l", **rec)
    assert 8.0 <= big <= 30.0


@pytest.mark.parametrize("fold", [-1, 1])
def test_trained_like_vit_l_full_depth(api, pkg, trained, fold):
    """ViT-L/14 + 4 registers, all 24 layers, f16, 518 x 518, batch 2 (image 1 checked): HIP vs the oracle in its ggml-default mode and
    in every ggml-uncertain switch mode, and vs exact arithmetic next to the oracle's own distance to it.  With separate LayerNorm launches
    (fold = -1: f16(LN(x)) rounded exactly where ggml rounds it) and with the LayerNorm folded into the neighbouring GEMM epilogues (fold = 1:
    the activation is rounded BEFORE the normalisation -- the outlier channels of this checkpoint are what that could hurt)."""
    path = trained("large")
    imgs = pkg.synth.synthetic_images(2, 518, 518, seed=42)
    got = api.Session(api.Model(path, classify=True, ln_fold=fold)).predict(imgs, classify=True, topk=5)
    lg, tk = got["logits"][1], got["patch_tokens"][1]
    assert np.isfinite(lg).all() and np.isfinite(tk).all()
    ex = OracleModel(path).forward_exact(imgs[1], classify=True)
    big, bigt = float(np.abs(ex["logits"]).max()), float(np.abs(ex["patch_tokens"]).max())
    rec = {"max_abs_logit_exact": big, "max_abs_token_exact": bigt, "hip_vs_exact_abs": _abs(lg, ex["logits"]),
           "hip_vs_exact_tokens_abs": _abs(tk, ex["patch_tokens"])}
    worst, worst_t = 0.0, 0.0
    for name, kw in _SWITCHES.items():
        o = OracleModel(path, **kw).forward(imgs[1], classify=True)
        rec[f"hip_vs_{name}_abs"] = _abs(lg, o["logits"])
        rec[f"hip_vs_{name}_rel"] = _rel(lg, o["logits"])
        rec[f"hip_vs_{name}_tokens_abs"] = _abs(tk, o["patch_tokens"])
        rec[f"hip_vs_{name}_tokens_rel"] = _rel(tk, o["patch_tokens"])
        rec[f"{name}_vs_exact_abs"] = _abs(o["logits"], ex["logits"])
        rec[f"{name}_vs_exact_tokens_abs"] = _abs(o["patch_tokens"], ex["patch_tokens"])
        worst = max(worst, rec[f"{name}_vs_exact_abs"])
        worst_t = max(worst_t, rec[f"{name}_vs_exact_tokens_abs"])
        if name == "ggml_default":
            top_ref = list(np.argsort(-o["probs"], kind="stable")[:5])
            rec["max_abs_dprob"] = _abs(got["probs"][1], o["probs"])
    emu = OracleModel(path, attn_round=1).forward(imgs[1], classify=True)
    rec["attn_round_emulation_vs_exact_abs"] = _abs(emu["logits"], ex["logits"])
    rec["hip_vs_attn_round_emulation_abs"] = _abs(lg, emu["logits"])
    rec["hip_over_worst_ggml_style"] = rec["hip_vs_exact_abs"] / worst
    rec["hip_tokens_over_worst_ggml_style"] = rec["hip_vs_exact_tokens_abs"] / worst_t
    rec["within_bound_absolute_1e-3"] = bool(rec["hip_vs_ggml_default_abs"] <= 1e-3)
    _record("trained_like_vit_l_f16" + ("_ln_fold" if fold > 0 else ""), **rec)
    # the stated contract (relative to the largest logit / token value), against every switch mode and against exact arithmetic
    for name in _SWITCHES:
        assert rec[f"hip_vs_{name}_rel"] <= 1e-3, (name, rec)
        assert rec[f"hip_vs_{name}_tokens_rel"] <= 5e-3, (name, rec)
    assert rec["hip_vs_exact_abs"] <= 1e-3 * max(1.0, big), rec
    # How good an approximation of the model is the HIP path next to a ggml-style implementation?  On i.i.d. weights: the same (1.0 - 1.15,
    # test_distance_to_exact_arithmetic).  HERE the f16 attention operands show: measured 8.5e-3 from exact against 6.7e-3 for the
    # ggml-default oracle (1.26 x; tokens 1.14 x), and the oracle's own emulation of that rounding (attn_round = 1) sits at 8.1e-3 --
    # with |score| of 30 - 60 the 2^-11 relative rounding of q and k is a visible, but not dominant, share.  Still 0.67 of the stated bound.
    assert rec["hip_over_worst_ggml_style"] <= 1.5, rec
    assert rec["hip_tokens_over_worst_ggml_style"] <= 1.5, rec
    assert list(got["topk_ids"][1]) == top_ref


@pytest.mark.parametrize("fold", [-1, 1])
def test_trained_like_hidden_states_per_layer(api, pkg, trained, fold):
    """Where along the depth does the HIP path leave the oracle?  Residual stream after layers 1, 6, 12, 18, 24 of the trained-like ViT-L
    (dinov2_hip_debug_hidden vs the oracle's hidden states), relative to the largest entry of the ORDINARY channels (the outlier channels
    are 100 x larger and compared separately, relative to themselves)."""
    path = trained("large")
    img = pkg.synth.synthetic_images(1, 518, 518, seed=7)
    sess = api.Session(api.Model(path, classify=True, ln_fold=fold))
    hid = OracleModel(path).forward(img[0], classify=False, hidden=True)["hidden"]
    out_ch = pkg.synth.trained_outlier_channels(1024)
    ordinary = np.setdiff1d(np.arange(1024), out_ch)
    rec = {}
    for layer in (1, 6, 12, 18, 24):
        h = sess.debug_hidden(img, layer)[0]
        d = np.abs(h - hid[layer])
        rec[f"layer{layer}_ordinary_rel"] = float(d[:, ordinary].max() / np.abs(hid[layer][:, ordinary]).max())
        rec[f"layer{layer}_outlier_rel"] = float(d[:, out_ch].max() / np.abs(hid[layer][:, out_ch]).max())
    _record("trained_like_vit_l_hidden" + ("_ln_fold" if fold > 0 else ""), **rec)
    for k, v in rec.items():
        assert v <= 5e-3, (k, rec)


@pytest.mark.parametrize("fold", [-1, 1])
def test_trained_like_vit_g_bf16_full_depth(api, pkg, trained, fold):
    """ViT-g/14 SwiGLU, 40 layers, bf16 compute (BASELINE configs[3]'s dtype) on the trained-like statistics, batch 2; the same model in f16
    next to it.  Bounds: the full-depth ViT-g ones of tests/test_gpu_configs.py (bf16 2e-2 / 4e-2, f16 2e-3 / 5e-3)."""
    path = trained("giant")
    imgs = pkg.synth.synthetic_images(2, 518, 518, seed=64)
    ora = OracleModel(path)
    exp = ora.forward(imgs[1], classify=True)
    ex = ora.forward_exact(imgs[1], classify=True)
    rec = {"max_abs_logit": float(np.abs(exp["logits"]).max()), "ggml_default_vs_exact_abs": _abs(exp["logits"], ex["logits"])}
    for name, dt, lb, tb in (("bf16", api.BF16, 2e-2, 4e-2), ("f16", api.F16, 2e-3, 5e-3)):
        got = api.Session(api.Model(path, dtype=dt, classify=True, ln_fold=fold)).predict(imgs, classify=True, want=("logits", "probs", "patch_tokens"))
        rec[f"{name}_abs_dlogit"] = _abs(got["logits"][1], exp["logits"])
        rec[f"{name}_rel_dlogit"] = _rel(got["logits"][1], exp["logits"])
        rec[f"{name}_rel_dtoken"] = _rel(got["patch_tokens"][1], exp["patch_tokens"])
        rec[f"{name}_vs_exact_abs"] = _abs(got["logits"][1], ex["logits"])
        assert np.isfinite(got["logits"]).all()
        _record("trained_like_vit_g" + ("_ln_fold" if fold > 0 else ""), **rec)
        assert rec[f"{name}_rel_dlogit"] <= lb, rec
        assert rec[f"{name}_rel_dtoken"] <= tb, rec


@pytest.mark.parametrize("model,registers,size,batch", [("small", 0, 224, 1), ("base", 4, 518, 1)])
def test_trained_like_small_configs(api, pkg, tmp_path, model, registers, size, batch):
    """BASELINE configs[0] / [1] on the trained-like statistics: ViT-S/14 without registers at 224 x 224 (the reference's CPU-runnable case)
    and ViT-B/14 + 4 registers at 518 x 518, batch 1, f16 -- HIP vs the ggml-default oracle and vs exact arithmetic."""
    path = str(tmp_path / f"{model}.gguf")
    pkg.synth.write_synthetic_gguf(path, model, registers=registers, num_classes=1000, seed=42, head_std=0.12, trained_like=True)
    imgs = pkg.synth.synthetic_images(batch, size, size, seed=11)
    got = api.Session(api.Model(path, classify=True)).predict(imgs, classify=True, topk=5)
    ora = OracleModel(path)
    exp = ora.forward(imgs[0], classify=True)
    ex = ora.forward_exact(imgs[0], classify=True)
    rec = {"max_abs_logit": float(np.abs(exp["logits"]).max()), "hip_vs_ggml_default_abs": _abs(got["logits"][0], exp["logits"]),
           "hip_vs_ggml_default_rel": _rel(got["logits"][0], exp["logits"]), "hip_vs_ggml_default_tokens_rel": _rel(got["patch_tokens"][0], exp["patch_tokens"]),
           "hip_vs_exact_abs": _abs(got["logits"][0], ex["logits"]), "ggml_default_vs_exact_abs": _abs(exp["logits"], ex["logits"])}
    _record(f"trained_like_vit_{model[0]}_f16_{size}", **rec)
    assert rec["hip_vs_ggml_default_rel"] <= 1e-3, rec
    assert rec["hip_vs_ggml_default_tokens_rel"] <= 5e-3, rec
    assert rec["hip_vs_exact_abs"] <= 1.5 * rec["ggml_default_vs_exact_abs"] + 1e-4, rec
    assert list(got["topk_ids"][0]) == list(np.argsort(-exp["probs"], kind="stable")[:5])


@pytest.mark.parametrize("wtype", ["q8_0", "q4_0"])
def test_trained_like_quantised_vit_l(api, pkg, tmp_path, wtype):
    """BASELINE configs[4] on the trained-like statistics: ViT-L/14 q8_0 / q4_0 GGUF, dequantised at load into the f16 MFMA path.  Against the
    dequantised-weights contract the stated 1e-3; against the ggml-mode oracle (activations quantised to Q8_0 blocks) the band the reference's
    own arithmetic leaves (test_ggml_quantised_arithmetic_is_its_own_noise_floor: it moves this much under one ulp of input)."""
    path = str(tmp_path / f"large_{wtype}.gguf")
    pkg.synth.write_synthetic_gguf(path, "large", registers=4, num_classes=1000, seed=42, head_std=0.12, trained_like=True, wtype=wtype)
    imgs = pkg.synth.synthetic_images(1, 518, 518, seed=5)
    got = api.Session(api.Model(path, classify=True)).predict(imgs, classify=True, want=("logits", "probs", "patch_tokens"))
    deq = OracleModel(path, quant_mode="dequant").forward(imgs[0], classify=True)
    ggml = OracleModel(path, quant_mode="ggml").forward(imgs[0], classify=True)
    rec = {"max_abs_logit": float(np.abs(deq["logits"]).max()), "hip_vs_dequantised_contract_rel": _rel(got["logits"][0], deq["logits"]),
           "hip_vs_ggml_mode_rel": _rel(got["logits"][0], ggml["logits"]), "ggml_mode_vs_dequantised_contract_rel": _rel(ggml["logits"], deq["logits"]),
           "hip_vs_dequantised_contract_tokens_rel": _rel(got["patch_tokens"][0], deq["patch_tokens"])}
    _record(f"trained_like_vit_l_{wtype}", **rec)
    assert np.isfinite(got["logits"]).all()
    assert rec["hip_vs_dequantised_contract_rel"] <= 1e-3, rec
    assert rec["hip_vs_dequantised_contract_tokens_rel"] <= 5e-3, rec
    assert rec["hip_vs_ggml_mode_rel"] <= 2e-2, rec
