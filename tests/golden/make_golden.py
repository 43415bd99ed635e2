#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the AUTHORING container only; needs transformers).

The reference (lavaman131/dinov2.cpp) holds no tests or golden vectors and cannot be built here
(ggml submodule empty, OpenCV absent), so the oracle is pinned against an independent implementation
of the same maths: HuggingFace `Dinov2[WithRegisters]ForImageClassification` built from config with
seeded random weights.  Known HF<->reference deltas are neutralised here, not in the oracle:

  * GELU: HF config hidden_act="gelu_pytorch_tanh" (reference = ggml tanh GELU, dinov2.cpp:567)
  * pos-embed interpolation: bicubic, align_corners=False, NO antialias == cv::resize(INTER_CUBIC)
    (dinov2.cpp:195-210); HF's with-registers model uses antialias=True, patched out below
  * classifier head: recomputed OUTSIDE HF by the reference's formula -- sum over patch tokens
    INCLUDING registers, divided by the constant (img_size/patch)^2 (dinov2.cpp:772-776, 794-803)
  * weights are rounded to f16 first (the converter stores >=2-D weights as F16,
    scripts/dinov2-to-gguf.py:157-159) so HF and the GGUF hold identical values

Outputs (tests/golden/): <name>.gguf in the converter's schema (written with the repo's own GGUF
writer; the `gguf` package is absent), <name>.npz with inputs and expected tensors.
Nothing from transformers or /root/reference travels to the GPU box -- only these data files.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
gw = pkg.gguf_writer

from transformers import (Dinov2Config, Dinov2ForImageClassification, Dinov2WithRegistersConfig,  # noqa: E402
                          Dinov2WithRegistersForImageClassification)
from transformers.models.dinov2 import modeling_dinov2 as md  # noqa: E402
from transformers.models.dinov2_with_registers import modeling_dinov2_with_registers as mr  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NUM_CLASSES = 10
IMG_SIZE, PATCH = 70, 14


def _interp_no_antialias(self, embeddings, height, width):
    """cv::resize(INTER_CUBIC) equivalent: bicubic, half-pixel centres, no antialias; identity on equal COUNT."""
    num_patches = embeddings.shape[1] - 1
    num_positions = self.position_embeddings.shape[1] - 1
    nh, nw = height // self.config.patch_size, width // self.config.patch_size
    if nh * nw == num_positions:  # dinov2.cpp:176-179 early return compares patch counts only
        return self.position_embeddings
    cls_pos = self.position_embeddings[:, :1]
    patch_pos = self.position_embeddings[:, 1:]
    dim = embeddings.shape[-1]
    s = int(num_positions ** 0.5)
    patch_pos = patch_pos.reshape(1, s, s, dim).permute(0, 3, 1, 2)
    patch_pos = torch.nn.functional.interpolate(patch_pos.float(), size=(nh, nw), mode="bicubic", align_corners=False)
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pos, patch_pos), dim=1)


md.Dinov2Embeddings.interpolate_pos_encoding = _interp_no_antialias
mr.Dinov2WithRegistersEmbeddings.interpolate_pos_encoding = _interp_no_antialias


def build(name, registers, swiglu, seed):
    torch.manual_seed(seed)
    common = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, mlp_ratio=3 if swiglu else 2,
                  hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6, image_size=IMG_SIZE, patch_size=PATCH,
                  use_swiglu_ffn=swiglu, num_labels=NUM_CLASSES, layerscale_value=1.0, hidden_dropout_prob=0.0,
                  attention_probs_dropout_prob=0.0, drop_path_rate=0.0)
    if registers:
        model = Dinov2WithRegistersForImageClassification(Dinov2WithRegistersConfig(num_register_tokens=registers, **common))
    else:
        model = Dinov2ForImageClassification(Dinov2Config(**common))
    model.eval()
    # non-trivial, seeded values for EVERY parameter (default init has zero biases / unit LN, which hides bugs)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for pn, p in model.named_parameters():
            if pn.endswith("lambda1"):
                p.copy_(0.3 + 0.2 * torch.randn(p.shape, generator=g))
            elif "norm" in pn and pn.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif p.ndim >= 2 and "embeddings" not in pn or "projection.weight" in pn:
                std = 0.08 if ("query" in pn or "key" in pn) else 0.05
                p.copy_((std * torch.randn(p.shape, generator=g)).half().float())  # f16-representable
            else:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
    return model


def to_gguf(model, path, registers):
    """Same tensor naming / dtypes / KV order as /root/reference/scripts/dinov2-to-gguf.py:49-166."""
    cfg = model.config
    sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    w = gw.GGUFWriter(arch="dinov2")
    for i in range(NUM_CLASSES):
        w.add_string(str(i), f"label_{i}")
    w.add_uint32("hidden_size", cfg.hidden_size)
    w.add_uint32("num_hidden_layers", cfg.num_hidden_layers)
    w.add_uint32("num_attention_heads", cfg.num_attention_heads)
    w.add_uint32("num_classes", NUM_CLASSES)
    w.add_uint32("patch_size", cfg.patch_size)
    w.add_uint32("img_size", cfg.image_size)
    w.add_uint32("ftype", 1)
    w.add_uint32("num_register_tokens", registers)
    pre = "dinov2_with_registers." if registers else "dinov2."
    L = cfg.num_hidden_layers

    def put(name, arr):
        arr = np.ascontiguousarray(arr)
        if arr.ndim >= 2 and not any(s in name for s in ("position_embeddings", "cls_token", "register_tokens", "bias")):
            w.add_tensor(name, arr.astype(np.float16))
        else:
            w.add_tensor(name, arr.astype(np.float32))

    put("embeddings.cls_token", sd[pre + "embeddings.cls_token"])
    put("embeddings.position_embeddings", sd[pre + "embeddings.position_embeddings"])
    if registers:
        put("embeddings.register_tokens", sd[pre + "embeddings.register_tokens"])
    put("embeddings.patch_embeddings.projection.weight", sd[pre + "embeddings.patch_embeddings.projection.weight"])
    put("embeddings.patch_embeddings.projection.bias",
        sd[pre + "embeddings.patch_embeddings.projection.bias"].reshape(1, -1, 1, 1))
    for i in range(L):
        b = f"encoder.layer.{i}."
        a = pre + b + "attention.attention."
        put(b + "norm1.weight", sd[pre + b + "norm1.weight"])
        put(b + "norm1.bias", sd[pre + b + "norm1.bias"])
        put(b + "attention.attention.qkv.weight",
            np.concatenate([sd[a + "query.weight"], sd[a + "key.weight"], sd[a + "value.weight"]], axis=0))
        put(b + "attention.attention.qkv.bias",
            np.concatenate([sd[a + "query.bias"], sd[a + "key.bias"], sd[a + "value.bias"]], axis=0))
        for n in ("attention.output.dense.weight", "attention.output.dense.bias", "layer_scale1.lambda1",
                  "norm2.weight", "norm2.bias"):
            put(b + n, sd[pre + b + n])
        names = ("mlp.weights_in", "mlp.weights_out") if cfg.use_swiglu_ffn else ("mlp.fc1", "mlp.fc2")
        for n in names:
            put(b + n + ".weight", sd[pre + b + n + ".weight"])
            put(b + n + ".bias", sd[pre + b + n + ".bias"])
        put(b + "layer_scale2.lambda1", sd[pre + b + "layer_scale2.lambda1"])
    put("layernorm.weight", sd[pre + "layernorm.weight"])
    put("layernorm.bias", sd[pre + "layernorm.bias"])
    put("classifier.weight", sd["classifier.weight"])
    put("classifier.bias", sd["classifier.bias"])
    w.write(path)


def expected(model, img, registers):
    """HF forward (f32) + the reference's head formula."""
    with torch.no_grad():
        base = getattr(model, "dinov2_with_registers", None) or model.dinov2
        out = base(torch.from_numpy(img)[None], output_hidden_states=True)
        hidden = torch.stack([h[0] for h in out.hidden_states]).numpy()          # [L+1, T, H]
        seq = out.last_hidden_state[0]                                           # final LN output [T, H]
        M = IMG_SIZE // PATCH
        cls = seq[0]
        pooled = seq[1:].double().sum(0).float() * (1.0 / float(M * M))           # registers INCLUDED, const divisor
        feat = torch.cat([cls, pooled])
        logits = model.classifier.weight @ feat + model.classifier.bias
        probs = torch.softmax(logits, -1)
        # HF-semantics head for reference (mean over patch tokens only)
        hf_logits = model.classifier(torch.cat([cls, seq[1 + registers:].mean(0)])[None])[0]
    return dict(hidden=hidden.astype(np.float32), final=seq.numpy().astype(np.float32),
                logits=logits.numpy().astype(np.float32), probs=probs.numpy().astype(np.float32),
                hf_logits=hf_logits.numpy().astype(np.float32))


def main():
    manifest = {}
    sizes = [(70, 70), (56, 84), (42, 42)]  # (H, W): identity pos-embed, non-square upsample, downsample
    for name, registers, swiglu, seed in [("tiny_gelu_noreg", 0, False, 101), ("tiny_gelu_reg4", 4, False, 102),
                                          ("tiny_swiglu_reg4", 4, True, 103)]:
        model = build(name, registers, swiglu, seed)
        to_gguf(model, os.path.join(OUT, name + ".gguf"), registers)
        rng = np.random.default_rng(seed)
        blob = {}
        for (hh, ww) in sizes:
            img = rng.standard_normal((3, hh, ww)).astype(np.float32)
            exp = expected(model, img, registers)
            key = f"{hh}x{ww}"
            blob[f"img_{key}"] = img
            for k, v in exp.items():
                blob[f"{k}_{key}"] = v
        # interpolated pos-embed vectors (torch bicubic == cv::resize INTER_CUBIC up to rounding)
        base = getattr(model, "dinov2_with_registers", None) or model.dinov2
        emb = base.embeddings
        for (hh, ww) in sizes:
            dummy = torch.zeros(1, 1 + (hh // PATCH) * (ww // PATCH), 128)
            with torch.no_grad():
                blob[f"pos_{hh}x{ww}"] = emb.interpolate_pos_encoding(dummy, hh, ww)[0].numpy().astype(np.float32)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **blob)
        manifest[name] = dict(registers=registers, swiglu=swiglu, seed=seed, sizes=[f"{a}x{b}" for a, b in sizes],
                              hidden=128, layers=2, heads=2, num_classes=NUM_CLASSES, img_size=IMG_SIZE, patch=PATCH)
        print("wrote", name)
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
