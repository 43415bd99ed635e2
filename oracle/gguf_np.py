"""GGUF v3 reader and ggml quant-block DEcoders in numpy (byte / integer work).

TEST INFRASTRUCTURE ONLY (see oracle/README.md): imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  The shipped product has its own C++ GGUF reader + HIP dequant kernels
(dinov2.cpp_amd/csrc/gguf_reader.cpp, kernels_misc.hip) and never imports this file; keeping the two
readers independent is what lets one check the other.

Follows:
  * the reader-side expectations of /root/reference/dinov2.cpp:263-348 (gguf_init_from_file +
    get_val_u32/get_val_str) and the schema written by /root/reference/scripts/dinov2-to-gguf.py:49-166.
  * ggml's block formats for Q4_0/Q4_1/Q5_0/Q5_1/Q8_0 (ggml-org/ggml ggml-quants.c `dequantize_row_*`;
    un-vendored submodule, SHA unpinned, so the published algorithm is restated; type ids per
    /root/reference/README.md:342-346).
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

GGUF_MAGIC = b"GGUF"
GGUF_VERSION = 3
DEFAULT_ALIGNMENT = 32

# gguf value types
T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 = range(13)
_SCALAR_FMT = {T_U8: "<B", T_I8: "<b", T_U16: "<H", T_I16: "<h", T_U32: "<I", T_I32: "<i",
               T_F32: "<f", T_BOOL: "<?", T_U64: "<Q", T_I64: "<q", T_F64: "<d"}

# ggml tensor types
GGML_F32, GGML_F16, GGML_Q4_0, GGML_Q4_1, GGML_Q5_0, GGML_Q5_1, GGML_Q8_0, GGML_BF16 = 0, 1, 2, 3, 6, 7, 8, 30
QK = 32
# type -> (block elements, block bytes)
TYPE_LAYOUT = {GGML_F32: (1, 4), GGML_F16: (1, 2), GGML_BF16: (1, 2), GGML_Q4_0: (QK, 18),
               GGML_Q4_1: (QK, 20), GGML_Q5_0: (QK, 22), GGML_Q5_1: (QK, 24), GGML_Q8_0: (QK, 34)}
TYPE_NAME = {GGML_F32: "f32", GGML_F16: "f16", GGML_BF16: "bf16", GGML_Q4_0: "q4_0", GGML_Q4_1: "q4_1",
             GGML_Q5_0: "q5_0", GGML_Q5_1: "q5_1", GGML_Q8_0: "q8_0"}
NAME_TYPE = {v: k for k, v in TYPE_NAME.items()}


# ----------------------------------------------------------------------------------------
# quant block decoders (rows = ne0/32 consecutive blocks)
# ----------------------------------------------------------------------------------------
def dequantize(raw: np.ndarray, gtype: int, shape) -> np.ndarray:
    """uint8 block bytes -> float32 array of `shape` (numpy order, last dim = ne0)."""
    be, bb = TYPE_LAYOUT[gtype]
    blk = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1, bb)
    nb = blk.shape[0]
    d = blk[:, 0:2].copy().view(np.float16).astype(np.float32).reshape(nb, 1)
    if gtype == GGML_Q8_0:
        w = blk[:, 2:].copy().view(np.int8).astype(np.float32) * d
    elif gtype == GGML_Q4_0:
        qs = blk[:, 2:]
        lo = (qs & 0xF).astype(np.int32) - 8
        hi = (qs >> 4).astype(np.int32) - 8
        w = np.concatenate([lo, hi], axis=1).astype(np.float32) * d
    elif gtype == GGML_Q4_1:
        m = blk[:, 2:4].copy().view(np.float16).astype(np.float32).reshape(nb, 1)
        qs = blk[:, 4:]
        w = np.concatenate([qs & 0xF, qs >> 4], axis=1).astype(np.float32) * d + m
    elif gtype in (GGML_Q5_0, GGML_Q5_1):
        o = 2 if gtype == GGML_Q5_0 else 4
        qh = blk[:, o:o + 4].copy().view(np.uint32).reshape(nb)
        qs = blk[:, o + 4:]
        j = np.arange(16, dtype=np.uint32)
        h0 = (((qh[:, None] >> j) & 1) << 4).astype(np.uint8)
        h1 = (((qh[:, None] >> (j + 16)) & 1) << 4).astype(np.uint8)
        q0 = (qs & 0xF) | h0
        q1 = (qs >> 4) | h1
        q = np.concatenate([q0, q1], axis=1).astype(np.int32)
        if gtype == GGML_Q5_0:
            w = (q - 16).astype(np.float32) * d
        else:
            m = blk[:, 2:4].copy().view(np.float16).astype(np.float32).reshape(nb, 1)
            w = q.astype(np.float32) * d + m
    else:
        raise ValueError(f"not a quantised type: {gtype}")
    return w.astype(np.float32).reshape(shape)


def block_mins(raw: np.ndarray, gtype: int, rows: int) -> np.ndarray:
    """The per-block minimum `m` of a Q4_1 / Q5_1 tensor (block layout d f16 | m f16 | ...): float32 [rows, blocks per row]."""
    assert gtype in (GGML_Q4_1, GGML_Q5_1)
    _, bb = TYPE_LAYOUT[gtype]
    blk = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1, bb)
    return np.ascontiguousarray(blk[:, 2:4].copy().view(np.float16).astype(np.float32).reshape(rows, -1))


# ----------------------------------------------------------------------------------------
# reader
# ----------------------------------------------------------------------------------------
class GGUFTensor:
    __slots__ = ("name", "ne", "gtype", "raw")

    def __init__(self, name, ne, gtype, raw):
        self.name, self.ne, self.gtype, self.raw = name, ne, gtype, raw

    @property
    def shape(self):  # numpy order
        return tuple(reversed(self.ne))

    def to_f32(self) -> np.ndarray:
        if self.gtype == GGML_F32:
            return self.raw.view(np.float32).reshape(self.shape)
        if self.gtype == GGML_F16:
            return self.raw.view(np.float16).astype(np.float32).reshape(self.shape)
        if self.gtype == GGML_BF16:
            return (self.raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(self.shape)
        return dequantize(self.raw, self.gtype, self.shape)


class GGUFFile:
    def __init__(self, path: str):
        buf = np.fromfile(path, dtype=np.uint8)
        mv = memoryview(buf)
        pos = 0

        def rd(fmt):
            nonlocal pos
            v = struct.unpack_from(fmt, mv, pos)
            pos += struct.calcsize(fmt)
            return v if len(v) > 1 else v[0]

        def rstr():
            nonlocal pos
            n = rd("<Q")
            s = bytes(mv[pos:pos + n])
            pos += n
            return s.decode("utf-8")

        def rval(t):
            if t == T_STR:
                return rstr()
            if t == T_ARR:
                et = rd("<I")
                n = rd("<Q")
                return [rval(et) for _ in range(n)]
            return rd(_SCALAR_FMT[t])

        if bytes(mv[0:4]) != GGUF_MAGIC:
            raise ValueError("not a GGUF file")
        pos = 4
        self.version = rd("<I")
        if self.version not in (2, 3):
            raise ValueError(f"unsupported GGUF version {self.version}")
        n_tensors, n_kv = rd("<Q"), rd("<Q")
        self.kv: "OrderedDict[str, object]" = OrderedDict()
        for _ in range(n_kv):
            k = rstr()
            t = rd("<I")
            self.kv[k] = rval(t)
        infos = []
        for _ in range(n_tensors):
            name = rstr()
            nd = rd("<I")
            ne = tuple(struct.unpack_from(f"<{nd}Q", mv, pos))
            pos += 8 * nd
            gtype, off = rd("<I"), rd("<Q")
            infos.append((name, ne, gtype, off))
        a = int(self.kv.get("general.alignment", DEFAULT_ALIGNMENT))
        data0 = (pos + a - 1) // a * a
        self.tensors: "OrderedDict[str, GGUFTensor]" = OrderedDict()
        for name, ne, gtype, off in infos:
            be, bb = TYPE_LAYOUT[gtype]
            n = int(np.prod(ne)) if len(ne) else 1
            nbytes = n // be * bb
            self.tensors[name] = GGUFTensor(name, ne, gtype, buf[data0 + off: data0 + off + nbytes])

    def u32(self, key):
        return int(self.kv[key])
