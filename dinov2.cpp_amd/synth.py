"""Seeded synthetic DINOv2 checkpoints in the reference's GGUF schema.

There is no network and no pretrained checkpoint on the build or GPU box, so benchmarks and
full-size parity tests run on random-init weights of the exact architecture, written in the exact
schema /root/reference/scripts/dinov2-to-gguf.py:49-166 produces (tensor names, dtypes, fused QKV,
KV order) so that the loader sees what it would see for a converted HF checkpoint.

Model family (HF configs; /root/reference/README.md model table): hd = 64, patch = 14, img_size = 518.
"""
from __future__ import annotations

import numpy as np

from . import gguf_writer as gw

#            hidden layers heads ffn_hidden swiglu
CONFIGS = {
    "tiny":  dict(hidden=128, layers=2, heads=2, ffn=512, swiglu=False, patch=14, img_size=70),
    "tiny-swiglu": dict(hidden=128, layers=2, heads=2, ffn=256, swiglu=True, patch=14, img_size=70),
    "small": dict(hidden=384, layers=12, heads=6, ffn=1536, swiglu=False, patch=14, img_size=518),
    "base":  dict(hidden=768, layers=12, heads=12, ffn=3072, swiglu=False, patch=14, img_size=518),
    "large": dict(hidden=1024, layers=24, heads=16, ffn=4096, swiglu=False, patch=14, img_size=518),
    "giant": dict(hidden=1536, layers=40, heads=24, ffn=4096, swiglu=True, patch=14, img_size=518),
}


# ---- "trained-like" statistics (write_synthetic_gguf(trained_like=True)) ----
TRAINED_HEAD_SCALES = (1.0, 1.5, 2.0, 2.5, 3.0, 3.5)  # q and k rows of head hd in layer i scaled by [(hd + i) % 6]: scores x 1 .. x 12
TRAINED_OUTLIER_BIAS = (90.0, -70.0, 110.0, -80.0)    # fc2.bias of the outlier channels in the layer that creates them (layer_scale2 = 1 there)
TRAINED_OUTLIER_ROW_GAIN = 20.0                        # that layer's fc2 rows of the outlier channels: the token-dependent part
TRAINED_SINK_VALUE = 6.0                               # register-token embedding on the sink channels (other entries ~ N(0, 0.5))
TRAINED_SINK_KEY = 0.35                                # key weight from a sink channel onto the head's sink direction
TRAINED_SINK_QUERY = 6.0                               # query bias along that direction


def trained_outlier_channels(H: int) -> list:
    return [7, H // 3 + 5, H - 11] + ([H // 2 + 3] if H >= 1024 else [])


def trained_sink_channels(H: int) -> list:
    return [H // 8 + 2 * j + 1 for j in range(8)]


def trained_outlier_layer(L: int) -> int:
    return max(0, min(L - 2, L // 6))


def flops_per_image(cfg: dict, height: int, width: int, registers: int, num_classes: int) -> float:
    """Algorithmic FLOPs of one forward; padding FLOPs do not count."""
    H, L, F, p = cfg["hidden"], cfg["layers"], cfg["ffn"], cfg["patch"]
    P = (height // p) * (width // p)
    T = 1 + registers + P
    ffn = (2 * T * H * 2 * F + 2 * T * F * H) if cfg["swiglu"] else (4 * T * H * F)
    per_layer = 2 * T * H * 3 * H + 4 * T * T * H + 2 * T * H * H + ffn
    return float(L * per_layer + 2 * P * 3 * p * p * H + 2 * 2 * H * num_classes)


def write_synthetic_gguf(path: str, model: str | dict = "large", *, registers: int = 4, num_classes: int = 1000,
                         seed: int = 42, wtype: str = "f16", layers: int | None = None, head_std: float = 0.02,
                         trained_like: bool = False) -> dict:
    """Write a seeded random DINOv2 GGUF.  Returns the hparams dict.

    wtype: storage type of the 2-D `*.weight` matrices ("f16", "f32", "q4_0", "q4_1", "q5_0", "q5_1",
    "q8_0"), mirroring what /root/reference/quantize.cpp produces (conv kernel and 1-D tensors keep
    their dtypes, dinov2.cpp:227-236).
    head_std: standard deviation of the classifier weights.  0.02 gives max|logit| ~ 2-3 over 1000 classes; trained
    ImageNet heads produce |logit| of 10-20, which 0.12 reproduces (used by the absolute-error parity test).
    trained_like: the statistics of a TRAINED DINOv2 that i.i.d. N(0, 0.02) weights lack and that
    decide whether f16 attention operands / f16 activations are good enough:
      * peaky attention: the q and k rows of every head are scaled by a per-head factor from TRAINED_HEAD_SCALES, so that the
        pre-softmax |score| of several heads reaches 30 - 60 (ggml keeps q, k and the scores in f32, dinov2.cpp:527-536);
      * residual outlier channels: from layer TRAINED_OUTLIER_LAYER(L) on, the channels of trained_outlier_channels(H) sit at
        ~ 100 x the median |x| (a constant part through `fc2.bias` and a token-dependent part through that layer's fc2 rows),
        and the LayerNorm weights of those channels are small, as in the released checkpoints;
      * attention sinks: the register tokens carry a large component on TRAINED_SINK_CHANNELS channels that every second head's
        keys read out along the direction its query bias points in: all queries of those heads put most of their softmax mass
        on the registers.
    Use with head_std = 0.12 for trained-scale logits.  tests/test_gpu_trained_like.py asserts that the regime is reached.
    """
    cfg = dict(CONFIGS[model]) if isinstance(model, str) else dict(model)
    if layers is not None:
        cfg["layers"] = layers
    H, L, nh, F, p = cfg["hidden"], cfg["layers"], cfg["heads"], cfg["ffn"], cfg["patch"]
    M = cfg["img_size"] // p
    rng = np.random.default_rng(seed)
    gt = gw.NAME_TYPE[wtype]

    def normal(shape, std):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    w = gw.GGUFWriter(arch="dinov2")
    if num_classes > 0:
        for i in range(num_classes):
            w.add_string(str(i), f"class_{i}")
    w.add_uint32("hidden_size", H)
    w.add_uint32("num_hidden_layers", L)
    w.add_uint32("num_attention_heads", nh)
    w.add_uint32("num_classes", num_classes if num_classes > 0 else 1000)
    w.add_uint32("patch_size", p)
    w.add_uint32("img_size", cfg["img_size"])
    w.add_uint32("ftype", {"f32": 0, "f16": 1}.get(wtype, gt))
    w.add_uint32("num_register_tokens", registers)

    def mat(name, shape, std):
        a = normal(shape, std)
        if gt in (gw.GGML_F32, gw.GGML_F16):
            w.add_tensor(name, a.astype(np.float16) if gt == gw.GGML_F16 else a)
        else:
            w.add_tensor(name, a, gtype=gt)

    def vec(name, shape, std, mean=0.0):
        w.add_tensor(name, (normal(shape, std) + np.float32(mean)).astype(np.float32))

    out_ch = trained_outlier_channels(H) if trained_like else []
    sink_ch = trained_sink_channels(H) if trained_like else []
    l_out = trained_outlier_layer(L)

    # once the outlier channels exist they dominate every row's variance: trained LayerNorm weights make up for it (large on the ordinary
    # channels, small on the outliers), so that the layers behind keep seeing O(1) inputs
    restore = np.float32(np.sqrt(1.0 + sum(b * b for b in TRAINED_OUTLIER_BIAS[: len(out_ch)]) / H))
    # deeper than 24 layers the ordinary channels of a random-weight residual stream keep growing (and the scores with their square: 340 at
    # layer 36 of the 40-layer ViT-g, where even the ggml-mode oracle is 21 logit units from exact arithmetic -- a chaotic system, not a
    # test): smaller LayerScale keeps the deep models in the regime of the 24-layer one
    ls_k = np.float32(min(1.0, (24.0 / L) ** 2))

    def norm_weight(name, after_outliers=False):
        g = normal((H,), 0.1) + np.float32(1.0)
        if trained_like:
            if after_outliers:
                g *= restore
            g[out_ch] = np.float32(0.04) * np.sign(g[out_ch])
        w.add_tensor(name, g.astype(np.float32))

    vec("embeddings.cls_token", (1, 1, H), 0.5)
    vec("embeddings.position_embeddings", (1, 1 + M * M, H), 0.3)
    if registers > 0:
        reg = normal((1, registers, H), 0.5)
        if trained_like:
            reg[0, :, sink_ch] = np.float32(TRAINED_SINK_VALUE)
        w.add_tensor("embeddings.register_tokens", reg)
    w.add_tensor("embeddings.patch_embeddings.projection.weight", normal((H, 3, p, p), 0.04).astype(np.float16))
    vec("embeddings.patch_embeddings.projection.bias", (1, H, 1, 1), 0.1)
    for i in range(L):
        b = f"encoder.layer.{i}."
        norm_weight(b + "norm1.weight", i > l_out)
        vec(b + "norm1.bias", (H,), 0.05)
        qkv = normal((3 * H, H), 0.02)
        qkv[: 2 * H] *= np.float32(2.0)  # peakier attention than the near-uniform default
        qkv_bias = normal((3 * H,), 0.05)
        if trained_like:
            for hd in range(nh):
                # (x sqrt(384 / H) x 0.75: the score of a random q . k grows with H; this keeps every model of the family in the same regime)
                sc = np.float32(TRAINED_HEAD_SCALES[(hd + i) % len(TRAINED_HEAD_SCALES)] * 0.75 * np.sqrt(384.0 / H))
                qkv[hd * 64:(hd + 1) * 64] *= sc
                qkv[H + hd * 64:H + (hd + 1) * 64] *= sc
                if hd % 2 == 0 and registers > 0:  # a sink head: keys read the registers' sink channels out along d, the query bias points along d
                    d = rng.standard_normal(64).astype(np.float32)
                    d /= np.linalg.norm(d)
                    qkv[H + hd * 64:H + (hd + 1) * 64][:, sink_ch] += np.float32(TRAINED_SINK_KEY) * d[:, None]
                    qkv_bias[hd * 64:(hd + 1) * 64] += np.float32(TRAINED_SINK_QUERY) * d
        if gt in (gw.GGML_F32, gw.GGML_F16):
            w.add_tensor(b + "attention.attention.qkv.weight", qkv.astype(np.float16) if gt == gw.GGML_F16 else qkv)
        else:
            w.add_tensor(b + "attention.attention.qkv.weight", qkv, gtype=gt)
        w.add_tensor(b + "attention.attention.qkv.bias", qkv_bias)
        mat(b + "attention.output.dense.weight", (H, H), 0.02)
        vec(b + "attention.output.dense.bias", (H,), 0.05)
        ls1 = normal((H,), 0.1) + np.float32(0.3)
        if trained_like:
            ls1 *= ls_k
        w.add_tensor(b + "layer_scale1.lambda1", ls1)
        norm_weight(b + "norm2.weight", i > l_out)
        vec(b + "norm2.bias", (H,), 0.05)
        fc1n, fc2n = ("mlp.weights_in", "mlp.weights_out") if cfg["swiglu"] else ("mlp.fc1", "mlp.fc2")
        mat(b + fc1n + ".weight", (2 * F if cfg["swiglu"] else F, H), 0.02)
        vec(b + fc1n + ".bias", (2 * F if cfg["swiglu"] else F,), 0.05)
        fc2 = normal((H, F), 0.02)
        fc2_bias = normal((H,), 0.05)
        ls2 = normal((H,), 0.1) + np.float32(0.3)
        if trained_like:
            ls2 *= ls_k
        if trained_like and i == l_out:  # the layer that writes the outlier channels into the residual stream
            fc2[out_ch] *= np.float32(TRAINED_OUTLIER_ROW_GAIN)
            fc2_bias[out_ch] = np.asarray(TRAINED_OUTLIER_BIAS[: len(out_ch)], np.float32)
            ls2[out_ch] = np.float32(1.0)
        if gt in (gw.GGML_F32, gw.GGML_F16):
            w.add_tensor(b + fc2n + ".weight", fc2.astype(np.float16) if gt == gw.GGML_F16 else fc2)
        else:
            w.add_tensor(b + fc2n + ".weight", fc2, gtype=gt)
        w.add_tensor(b + fc2n + ".bias", fc2_bias)
        w.add_tensor(b + "layer_scale2.lambda1", ls2)
    norm_weight("layernorm.weight", True)
    vec("layernorm.bias", (H,), 0.05)
    if num_classes > 0:
        mat("classifier.weight", (num_classes, 2 * H), head_std)
        vec("classifier.bias", (num_classes,), 0.05)
    w.write(path)
    return dict(cfg, registers=registers, num_classes=num_classes, wtype=wtype)


def synthetic_images(batch: int, height: int, width: int, seed: int = 42) -> np.ndarray:
    """Preprocessed float images at the dino_predict level: [B, 3, H, W] planar RGB, i.i.d. N(0,1)
    (ImageNet-normalised range; seed = dino_params.seed default, /root/reference/dinov2.h:58)."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((batch, 3, height, width), dtype=np.float32)
