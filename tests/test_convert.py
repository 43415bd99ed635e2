"""HF checkpoint directory (config.json + model.safetensors) -> GGUF in the reference converter's schema."""
import json
import os
from importlib import import_module

import numpy as np

from __graft_entry__ import PKG_NAME
from oracle import gguf_np
from oracle.oracle import OracleModel


def _hf_state_dict(rng, H, L, heads, patch, img, regs, classes, swiglu):
    pre = "dinov2_with_registers." if regs else "dinov2."
    P = (img // patch) ** 2
    sd = {pre + "embeddings.cls_token": rng.standard_normal((1, 1, H)), pre + "embeddings.mask_token": rng.standard_normal((1, H)),
          pre + "embeddings.position_embeddings": rng.standard_normal((1, P + 1, H)) * 0.1,
          pre + "embeddings.patch_embeddings.projection.weight": rng.standard_normal((H, 3, patch, patch)) * 0.05,
          pre + "embeddings.patch_embeddings.projection.bias": rng.standard_normal(H) * 0.1,
          pre + "layernorm.weight": 1 + 0.1 * rng.standard_normal(H), pre + "layernorm.bias": 0.1 * rng.standard_normal(H),
          "classifier.weight": rng.standard_normal((classes, 2 * H)) * 0.05, "classifier.bias": rng.standard_normal(classes) * 0.1}
    if regs:
        sd[pre + "embeddings.register_tokens"] = rng.standard_normal((1, regs, H))
    F = 2 * H
    for i in range(L):
        b = f"{pre}encoder.layer.{i}."
        for n in ("query", "key", "value"):
            sd[b + f"attention.attention.{n}.weight"] = rng.standard_normal((H, H)) * 0.06
            sd[b + f"attention.attention.{n}.bias"] = rng.standard_normal(H) * 0.1
        sd[b + "attention.output.dense.weight"] = rng.standard_normal((H, H)) * 0.05
        sd[b + "attention.output.dense.bias"] = rng.standard_normal(H) * 0.1
        for n in ("norm1", "norm2"):
            sd[b + n + ".weight"] = 1 + 0.1 * rng.standard_normal(H)
            sd[b + n + ".bias"] = 0.1 * rng.standard_normal(H)
        sd[b + "layer_scale1.lambda1"] = 0.3 + 0.1 * rng.standard_normal(H)
        sd[b + "layer_scale2.lambda1"] = 0.3 + 0.1 * rng.standard_normal(H)
        if swiglu:
            sd[b + "mlp.weights_in.weight"] = rng.standard_normal((2 * F, H)) * 0.05
            sd[b + "mlp.weights_in.bias"] = rng.standard_normal(2 * F) * 0.1
            sd[b + "mlp.weights_out.weight"] = rng.standard_normal((H, F)) * 0.05
            sd[b + "mlp.weights_out.bias"] = rng.standard_normal(H) * 0.1
        else:
            sd[b + "mlp.fc1.weight"] = rng.standard_normal((F, H)) * 0.05
            sd[b + "mlp.fc1.bias"] = rng.standard_normal(F) * 0.1
            sd[b + "mlp.fc2.weight"] = rng.standard_normal((H, F)) * 0.05
            sd[b + "mlp.fc2.bias"] = rng.standard_normal(H) * 0.1
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in sd.items()}


def test_checkpoint_dir_to_gguf(tmp_path):
    from safetensors.numpy import save_file
    conv = import_module(PKG_NAME + ".convert")
    H, L, heads, patch, img, regs, classes = 128, 2, 2, 14, 70, 4, 7
    sd = _hf_state_dict(np.random.default_rng(3), H, L, heads, patch, img, regs, classes, swiglu=False)
    ck = tmp_path / "ckpt"
    ck.mkdir()
    save_file(sd, str(ck / "model.safetensors"))
    json.dump({"hidden_size": H, "num_hidden_layers": L, "num_attention_heads": heads, "patch_size": patch, "image_size": img,
               "id2label": {str(i): f"class {i}" for i in range(classes)}}, open(ck / "config.json", "w"))
    out = str(tmp_path / "ggml-model.gguf")
    assert conv.main(["convert", str(ck), out]) == 0
    g = gguf_np.GGUFFile(out)
    kv, t = g.kv, g.tensors
    assert kv["general.architecture"] == "dinov2" and kv["3"] == "class 3"
    assert (kv["hidden_size"], kv["num_hidden_layers"], kv["num_attention_heads"], kv["num_classes"], kv["patch_size"],
            kv["img_size"], kv["ftype"], kv["num_register_tokens"]) == (H, L, heads, classes, patch, img, 1, regs)
    assert "embeddings.mask_token" not in t and not any(".query." in n or ".key." in n or ".value." in n for n in t)
    pre = "dinov2_with_registers."
    qkv = t["encoder.layer.1.attention.attention.qkv.weight"]
    assert qkv.gtype == 1 and qkv.shape == (3 * H, H)  # F16, rows = [q; k; v]
    exp = np.concatenate([sd[pre + f"encoder.layer.1.attention.attention.{n}.weight"] for n in ("query", "key", "value")], 0)
    assert np.array_equal(qkv.to_f32(), exp.astype(np.float16).astype(np.float32))
    assert t["embeddings.position_embeddings"].gtype == 0 and t["encoder.layer.0.norm1.weight"].gtype == 0
    assert t["embeddings.patch_embeddings.projection.bias"].shape == (1, H, 1, 1)
    assert t["classifier.weight"].gtype == 1
    # the file is a loadable model: the oracle runs it
    r = OracleModel(out).forward(np.random.default_rng(0).standard_normal((3, img, img)).astype(np.float32), classify=True)
    assert r["logits"].shape == (classes,) and np.isfinite(r["logits"]).all()


def test_converter_equals_golden_generator_on_a_real_hf_state_dict(tmp_path):
    """Same tensors, byte for byte, as tests/golden/make_golden.py::to_gguf (the path that pins the oracle to HuggingFace)
    when fed a live `Dinov2WithRegistersForImageClassification.state_dict()` -- checks the key handling against real HF
    names.  Skipped where transformers is absent (it never travels to the GPU box)."""
    import importlib.util
    import pytest
    pytest.importorskip("transformers")
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    conv = import_module(PKG_NAME + ".convert")
    for regs, swiglu in ((4, False), (0, True)):
        model = mg.build("t", regs, swiglu, seed=11)
        a, b = str(tmp_path / f"a{regs}.gguf"), str(tmp_path / f"b{regs}.gguf")
        mg.to_gguf(model, a, regs)
        sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
        cfg = model.config.to_dict()
        conv.convert_state_dict(sd, cfg, b)
        ga, gb = gguf_np.GGUFFile(a), gguf_np.GGUFFile(b)
        assert set(ga.tensors) == set(gb.tensors)
        for n, t in ga.tensors.items():
            u = gb.tensors[n]
            assert (t.gtype, t.ne) == (u.gtype, u.ne), n
            assert np.array_equal(t.raw, u.raw), n
        for k in ("hidden_size", "num_hidden_layers", "num_attention_heads", "num_classes", "patch_size", "img_size", "ftype",
                  "num_register_tokens"):
            assert ga.kv[k] == gb.kv[k], k
