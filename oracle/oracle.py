"""ctypes front-end of the CPU oracle (oracle/dinov2_oracle.c).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product (dinov2.cpp_amd/), which must fail loudly without its HIP
library rather than fall back to anything here.

Mirrors the reference call sequence dino_model_load -> dino_predict
(/root/reference/dinov2.cpp:239-352, 900-999) for ONE image at a time (the reference is batch 1).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import gguf_np as G

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_fp = C.POINTER(C.c_float)


class _Layer(C.Structure):
    _fields_ = [(n, _fp) for n in ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "o_w", "o_b", "ls1", "norm2_w",
                                   "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2", "qkv_m", "o_m", "fc1_m", "fc2_m")]


class _Model(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("hidden", "layers", "heads", "registers", "patch", "img_size",
                                          "num_classes", "ffn_hidden", "swiglu")]
                + [("eps", C.c_float)]
                + [(n, C.c_int32) for n in ("act_round", "conv_round", "gelu_f16_lut", "pool_const_divisor",
                                            "pool_includes_registers", "attn_round")]
                + [(n, _fp) for n in ("patch_w", "patch_b", "cls", "pos", "reg", "ln_w", "ln_b", "head_w", "head_b")]
                + [("layer", C.POINTER(_Layer)), ("head_m", _fp)])


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libdinov2_oracle.so")
    src = os.path.join(_HERE, "dinov2_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdinov2_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libdinov2_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.oracle_forward.restype = C.c_int
        _LIB.oracle_forward.argtypes = [C.POINTER(_Model), _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp,
                                        C.c_int]
        _dp = C.POINTER(C.c_double)
        _LIB.oracle_forward_exact.restype = C.c_int
        _LIB.oracle_forward_exact.argtypes = [C.POINTER(_Model), _fp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int]
        _LIB.oracle_interpolate_pos_embed.restype = None
        _LIB.oracle_interpolate_pos_embed.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]
    return _LIB


def _p(a):
    return a.ctypes.data_as(_fp) if a is not None else _fp()


class OracleModel:
    """dino_model counterpart: hparams + name->f32 tensor map + numerics switches.

    quant_mode (only matters for quantised 2-D weights):
      "ggml"    activations quantised to q8_0 blocks (Q4_0 / Q5_0 / Q8_0 weights) or q8_1 blocks (Q4_1 / Q5_1 weights: the block sum
                s = f16(d * sum q) multiplies the weight block's minimum), weights dequantised exactly (ggml CPU semantics)
      "dequant" weights dequantised then ROUNDED TO F16, activations rounded to f16 -- the contract of a
                dequant-on-load f16 MFMA path
    """

    def __init__(self, path: str, *, quant_mode: str = "ggml", attn_round: int = 0, act_round: int | None = None,
                 gelu_f16_lut: bool = True, pool_const_divisor: bool = True, pool_includes_registers: bool = True):
        f = G.GGUFFile(path)
        self.gguf = f
        kv = f.kv
        H = self.hidden = f.u32("hidden_size")
        L = self.layers = f.u32("num_hidden_layers")
        self.heads = f.u32("num_attention_heads")
        self.patch = f.u32("patch_size")
        self.img_size = f.u32("img_size")
        self.registers = f.u32("num_register_tokens") if "num_register_tokens" in kv else 0
        self.ftype = f.u32("ftype")
        t = f.tensors
        self.has_head = "classifier.weight" in t
        self.num_classes = t["classifier.weight"].ne[1] if self.has_head else 0
        self.labels = [kv.get(str(i), "") for i in range(self.num_classes)]
        # the reference picks SwiGLU by num_hidden_layers == 40 (dinov2.cpp:740); equivalent on real
        # checkpoints and usable on tiny fixtures: presence of mlp.weights_in
        self.swiglu = "encoder.layer.0.mlp.weights_in.weight" in t
        wtypes = {t[f"encoder.layer.{i}.attention.attention.qkv.weight"].gtype for i in range(L)}
        assert len(wtypes) == 1
        wt = wtypes.pop()
        self.wtype = wt
        quant = wt not in (G.GGML_F32, G.GGML_F16, G.GGML_BF16)
        if act_round is None:
            act_round = 0 if wt == G.GGML_F32 else 2 if wt == G.GGML_BF16 else 1 if not quant else \
                ((4 if wt in (G.GGML_Q4_1, G.GGML_Q5_1) else 3) if quant_mode == "ggml" else 1)
        self._keep = []

        def w2d(name):
            a = np.ascontiguousarray(t[name].to_f32().reshape(t[name].ne[1], t[name].ne[0]))
            if quant and quant_mode == "dequant" and t[name].gtype not in (G.GGML_F32, G.GGML_F16):
                a = a.astype(np.float16).astype(np.float32)
            self._keep.append(a)
            return a

        def v(name):
            a = np.ascontiguousarray(t[name].to_f32().reshape(-1))
            self._keep.append(a)
            return a

        def wmin(name):  # block minima of a Q4_1 / Q5_1 weight for the Q8_1 dot product (act_round 4), else NULL
            if act_round != 4 or t[name].gtype not in (G.GGML_Q4_1, G.GGML_Q5_1):
                return _fp()
            a = G.block_mins(t[name].raw, t[name].gtype, t[name].ne[1])
            self._keep.append(a)
            return _p(a)

        self.ffn_hidden = (t["encoder.layer.0.mlp.weights_out.weight"].ne[0] if self.swiglu
                           else t["encoder.layer.0.mlp.fc1.weight"].ne[1])
        layers = (_Layer * L)()
        for i in range(L):
            b = f"encoder.layer.{i}."
            fc1, fc2 = ("mlp.weights_in", "mlp.weights_out") if self.swiglu else ("mlp.fc1", "mlp.fc2")
            ly = layers[i]
            ly.norm1_w, ly.norm1_b = _p(v(b + "norm1.weight")), _p(v(b + "norm1.bias"))
            ly.qkv_w, ly.qkv_b = _p(w2d(b + "attention.attention.qkv.weight")), _p(v(b + "attention.attention.qkv.bias"))
            ly.o_w, ly.o_b = _p(w2d(b + "attention.output.dense.weight")), _p(v(b + "attention.output.dense.bias"))
            ly.ls1 = _p(v(b + "layer_scale1.lambda1"))
            ly.norm2_w, ly.norm2_b = _p(v(b + "norm2.weight")), _p(v(b + "norm2.bias"))
            ly.fc1_w, ly.fc1_b = _p(w2d(b + fc1 + ".weight")), _p(v(b + fc1 + ".bias"))
            ly.fc2_w, ly.fc2_b = _p(w2d(b + fc2 + ".weight")), _p(v(b + fc2 + ".bias"))
            ly.ls2 = _p(v(b + "layer_scale2.lambda1"))
            ly.qkv_m, ly.o_m = wmin(b + "attention.attention.qkv.weight"), wmin(b + "attention.output.dense.weight")
            ly.fc1_m, ly.fc2_m = wmin(b + fc1 + ".weight"), wmin(b + fc2 + ".weight")
        self._layers = layers
        m = self.c = _Model()
        m.hidden, m.layers, m.heads, m.registers = H, L, self.heads, self.registers
        m.patch, m.img_size, m.num_classes = self.patch, self.img_size, self.num_classes
        m.ffn_hidden, m.swiglu, m.eps = self.ffn_hidden, int(self.swiglu), 1e-6
        pw = t["embeddings.patch_embeddings.projection.weight"]
        m.act_round = act_round
        m.conv_round = {G.GGML_F16: 1, G.GGML_BF16: 2}.get(pw.gtype, 0)
        m.gelu_f16_lut = int(gelu_f16_lut)
        m.pool_const_divisor, m.pool_includes_registers = int(pool_const_divisor), int(pool_includes_registers)
        m.attn_round = attn_round
        self.pos = v("embeddings.position_embeddings")
        m.patch_w = _p(v("embeddings.patch_embeddings.projection.weight"))
        m.patch_b = _p(v("embeddings.patch_embeddings.projection.bias"))
        m.cls, m.pos = _p(v("embeddings.cls_token")), _p(self.pos)
        m.reg = _p(v("embeddings.register_tokens")) if self.registers else _fp()
        m.ln_w, m.ln_b = _p(v("layernorm.weight")), _p(v("layernorm.bias"))
        if self.has_head:
            m.head_w, m.head_b = _p(w2d("classifier.weight")), _p(v("classifier.bias"))
            m.head_m = wmin("classifier.weight")
        m.layer = layers

    def set(self, **kw):
        for k, val in kw.items():
            setattr(self.c, k, int(val))
        return self

    def tokens(self, h, w):
        return 1 + self.registers + (h // self.patch) * (w // self.patch)

    def forward(self, img_chw: np.ndarray, classify: bool = False, hidden: bool = False, nthreads: int = 0) -> dict:
        """img_chw: planar RGB f32 [3, H, W] (the "input" tensor of dinov2.cpp:629-631)."""
        if nthreads <= 0:  # OpenMP's default team = every CPU the host reports; containers are often given far fewer (256
            # threads measured 20-40x slower than 16 on the GPU box), so default to a modest team unless OMP_NUM_THREADS says otherwise
            nthreads = int(os.environ.get("OMP_NUM_THREADS", 0)) or min(16, os.cpu_count() or 1)
        img = np.ascontiguousarray(img_chw, dtype=np.float32)
        assert img.ndim == 3 and img.shape[0] == 3
        _, hh, ww = img.shape
        H, R = self.hidden, self.registers
        P = (hh // self.patch) * (ww // self.patch)
        T = 1 + R + P
        if classify and not self.has_head:
            raise ValueError("model has no classifier head")
        out = {"cls": np.empty(H, np.float32), "patch_tokens": np.empty((P + (R if classify else 0), H), np.float32)}
        if classify:
            out["logits"] = np.empty(self.num_classes, np.float32)
            out["probs"] = np.empty(self.num_classes, np.float32)
        if hidden:
            out["hidden"] = np.empty((self.layers + 1, T, H), np.float32)
        rc = _lib().oracle_forward(C.byref(self.c), _p(img), hh, ww, int(classify), _p(out["cls"]),
                                   _p(out["patch_tokens"]), _p(out.get("logits")), _p(out.get("probs")),
                                   _p(out.get("hidden")), nthreads)
        if rc != 0:
            raise RuntimeError(f"oracle_forward failed: {rc}")
        return out

    def forward_exact(self, img_chw: np.ndarray, classify: bool = False, nthreads: int = 0) -> dict:
        """The same graph on the same stored weights in DOUBLE with no intermediate rounding (oracle_forward_exact): the value
        that ggml's f32/f16 arithmetic and the HIP path's f16-MFMA arithmetic both approximate.  float64 outputs."""
        if nthreads <= 0:
            nthreads = int(os.environ.get("OMP_NUM_THREADS", 0)) or min(16, os.cpu_count() or 1)
        img = np.ascontiguousarray(img_chw, dtype=np.float32)
        assert img.ndim == 3 and img.shape[0] == 3
        _, hh, ww = img.shape
        H, R = self.hidden, self.registers
        P = (hh // self.patch) * (ww // self.patch)
        if classify and not self.has_head:
            raise ValueError("model has no classifier head")
        out = {"cls": np.empty(H, np.float64), "patch_tokens": np.empty((P + (R if classify else 0), H), np.float64)}
        if classify:
            out["logits"] = np.empty(self.num_classes, np.float64)
            out["probs"] = np.empty(self.num_classes, np.float64)
        dp = C.POINTER(C.c_double)

        def d(a):
            return a.ctypes.data_as(dp) if a is not None else dp()

        rc = _lib().oracle_forward_exact(C.byref(self.c), _p(img), hh, ww, int(classify), d(out["cls"]), d(out["patch_tokens"]),
                                         d(out.get("logits")), d(out.get("probs")), nthreads)
        if rc != 0:
            raise RuntimeError(f"oracle_forward_exact failed: {rc}")
        return out

    def interpolate_pos_embed(self, h_new: int, w_new: int) -> np.ndarray:
        M = self.img_size // self.patch
        out = np.empty((1 + h_new * w_new, self.hidden), np.float32)
        _lib().oracle_interpolate_pos_embed(_p(self.pos), M, self.hidden, h_new, w_new, _p(out))
        return out


def gelu_table() -> np.ndarray:
    """ggml's f16 GELU table (65 536 f16 bit patterns indexed by the input's bit pattern), built the way ggml.c builds it."""
    out = np.zeros(65536, np.uint16)
    _lib().oracle_gelu_table.argtypes = [C.c_void_p]
    _lib().oracle_gelu_table.restype = None
    _lib().oracle_gelu_table(out.ctypes.data_as(C.c_void_p))
    return out


def bgr_hwc_to_rgb_chw(img_hwc: np.ndarray) -> np.ndarray:
    """The repack dino_predict does before upload (dinov2.cpp:914-931): BGR interleaved -> RGB planar."""
    return np.ascontiguousarray(img_hwc[:, :, ::-1].transpose(2, 0, 1), dtype=np.float32)
