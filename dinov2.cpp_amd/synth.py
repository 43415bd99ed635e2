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


def flops_per_image(cfg: dict, height: int, width: int, registers: int, num_classes: int) -> float:
    """Algorithmic FLOPs of one forward; padding FLOPs do not count."""
    H, L, F, p = cfg["hidden"], cfg["layers"], cfg["ffn"], cfg["patch"]
    P = (height // p) * (width // p)
    T = 1 + registers + P
    ffn = (2 * T * H * 2 * F + 2 * T * F * H) if cfg["swiglu"] else (4 * T * H * F)
    per_layer = 2 * T * H * 3 * H + 4 * T * T * H + 2 * T * H * H + ffn
    return float(L * per_layer + 2 * P * 3 * p * p * H + 2 * 2 * H * num_classes)


def write_synthetic_gguf(path: str, model: str | dict = "large", *, registers: int = 4, num_classes: int = 1000,
                         seed: int = 42, wtype: str = "f16", layers: int | None = None, head_std: float = 0.02) -> dict:
    """Write a seeded random DINOv2 GGUF.  Returns the hparams dict.

    wtype: storage type of the 2-D `*.weight` matrices ("f16", "f32", "q4_0", "q4_1", "q5_0", "q5_1",
    "q8_0"), mirroring what /root/reference/quantize.cpp produces (conv kernel and 1-D tensors keep
    their dtypes, dinov2.cpp:227-236).
    head_std: standard deviation of the classifier weights.  0.02 gives max|logit| ~ 2-3 over 1000 classes; trained
    ImageNet heads produce |logit| of 10-20, which 0.12 reproduces (used by the absolute-error parity test).
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

    vec("embeddings.cls_token", (1, 1, H), 0.5)
    vec("embeddings.position_embeddings", (1, 1 + M * M, H), 0.3)
    if registers > 0:
        vec("embeddings.register_tokens", (1, registers, H), 0.5)
    w.add_tensor("embeddings.patch_embeddings.projection.weight", normal((H, 3, p, p), 0.04).astype(np.float16))
    vec("embeddings.patch_embeddings.projection.bias", (1, H, 1, 1), 0.1)
    for i in range(L):
        b = f"encoder.layer.{i}."
        vec(b + "norm1.weight", (H,), 0.1, 1.0)
        vec(b + "norm1.bias", (H,), 0.05)
        qkv = normal((3 * H, H), 0.02)
        qkv[: 2 * H] *= np.float32(2.0)  # peakier attention than the near-uniform default
        if gt in (gw.GGML_F32, gw.GGML_F16):
            w.add_tensor(b + "attention.attention.qkv.weight", qkv.astype(np.float16) if gt == gw.GGML_F16 else qkv)
        else:
            w.add_tensor(b + "attention.attention.qkv.weight", qkv, gtype=gt)
        vec(b + "attention.attention.qkv.bias", (3 * H,), 0.05)
        mat(b + "attention.output.dense.weight", (H, H), 0.02)
        vec(b + "attention.output.dense.bias", (H,), 0.05)
        vec(b + "layer_scale1.lambda1", (H,), 0.1, 0.3)
        vec(b + "norm2.weight", (H,), 0.1, 1.0)
        vec(b + "norm2.bias", (H,), 0.05)
        if cfg["swiglu"]:
            mat(b + "mlp.weights_in.weight", (2 * F, H), 0.02)
            vec(b + "mlp.weights_in.bias", (2 * F,), 0.05)
            mat(b + "mlp.weights_out.weight", (H, F), 0.02)
            vec(b + "mlp.weights_out.bias", (H,), 0.05)
        else:
            mat(b + "mlp.fc1.weight", (F, H), 0.02)
            vec(b + "mlp.fc1.bias", (F,), 0.05)
            mat(b + "mlp.fc2.weight", (H, F), 0.02)
            vec(b + "mlp.fc2.bias", (H,), 0.05)
        vec(b + "layer_scale2.lambda1", (H,), 0.1, 0.3)
    vec("layernorm.weight", (H,), 0.1, 1.0)
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
