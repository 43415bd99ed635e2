"""HuggingFace DINOv2 checkpoint -> GGUF, without `transformers` or `gguf`.

Counterpart of /root/reference/scripts/dinov2-to-gguf.py: same file schema -- labels as string KVs keyed by the
class index (:192-194), then the uint32 hparams hidden_size, num_hidden_layers, num_attention_heads, num_classes, patch_size,
img_size, ftype, num_register_tokens (:48-57, 132); tensor names = the HF state-dict key with its first component
("dinov2." / "dinov2_with_registers.") stripped (:167-170); `embeddings.mask_token`, `norm_pre*` and the separate q/k/v
projections are dropped (:173-176) and replaced by fused `...attention.attention.qkv.{weight,bias}` = concat(q, k, v) along
rows (:85-118); 1-D tensors and position_embeddings / cls_token / register_tokens stay F32, everything else is stored as F16
(:145-153); the conv bias is reshaped to [1, C, 1, 1] (:158-159).  Input: a checkpoint directory with `config.json` and
`model.safetensors` (what `save_pretrained` / the Hub hold) instead of a Hub name -- there is no network here.

    python -m dinov2_cpp_amd.convert <checkpoint_dir> [out.gguf]
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

from . import gguf_writer as gw

ARCH = "dinov2"
_F32_NAMES = {"embeddings.position_embeddings", "embeddings.cls_token", "embeddings.register_tokens"}


def _strip(name: str) -> str:
    return ".".join(name.split(".")[1:]) if name.startswith(ARCH) else name


def _skip(name: str) -> bool:
    return name == "embeddings.mask_token" or name.startswith("norm_pre") or "attention.attention" in name


def convert_state_dict(sd: dict, config: dict, out_path: str) -> dict:
    """sd: HF state dict (name -> numpy array, any float dtype); config: the checkpoint's config.json as a dict."""
    id2label = {int(k): v for k, v in (config.get("id2label") or {}).items()} if "classifier.weight" in sd else {}
    w = gw.GGUFWriter(arch=ARCH)
    for k in sorted(id2label):
        w.add_string(str(k), id2label[k])
    names = {_strip(k): k for k in sd}
    regs = int(sd[names["embeddings.register_tokens"]].shape[1]) if "embeddings.register_tokens" in names else 0
    L = int(config["num_hidden_layers"])
    for key, val in (("hidden_size", config["hidden_size"]), ("num_hidden_layers", L),
                     ("num_attention_heads", config["num_attention_heads"]), ("num_classes", len(id2label)),
                     ("patch_size", config["patch_size"]), ("img_size", config["image_size"]), ("ftype", 1),
                     ("num_register_tokens", regs)):
        w.add_uint32(key, int(val))

    def put(name, arr):
        arr = np.asarray(arr, dtype=np.float32)
        if name == "embeddings.patch_embeddings.projection.bias":
            arr = arr.reshape(1, -1, 1, 1)
            w.add_tensor(name, np.ascontiguousarray(arr, dtype=np.float32))
        elif arr.ndim == 1 or name in _F32_NAMES:
            w.add_tensor(name, np.ascontiguousarray(arr, dtype=np.float32))
        else:
            w.add_tensor(name, np.ascontiguousarray(arr).astype(np.float16))

    for short, full in names.items():
        if not _skip(short):
            put(short, sd[full])
    for i in range(L):
        base = f"encoder.layer.{i}.attention.attention"
        for part in ("weight", "bias"):
            q, k, v = (sd[names[f"{base}.{p}.{part}"]] for p in ("query", "key", "value"))
            put(f"{base}.qkv.{part}", np.concatenate([np.asarray(q, np.float32), np.asarray(k, np.float32), np.asarray(v, np.float32)], 0))
    w.write(out_path)
    return {"num_register_tokens": regs, "num_classes": len(id2label), "tensors": len(names)}


def convert_checkpoint(ckpt_dir: str, out_path: str) -> dict:
    from safetensors.numpy import load_file
    config = json.load(open(os.path.join(ckpt_dir, "config.json")))
    sd = load_file(os.path.join(ckpt_dir, "model.safetensors"))
    return convert_state_dict(sd, config, out_path)


def main(argv=None) -> int:
    argv = sys.argv if argv is None else argv
    if len(argv) < 2:
        print(f"usage: {argv[0]} checkpoint_dir [ggml-model.gguf]", file=sys.stderr)
        return 1
    out = argv[2] if len(argv) > 2 else "./ggml-model.gguf"
    info = convert_checkpoint(argv[1], out)
    print(f"Done. Output file: {out} ({info})")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
