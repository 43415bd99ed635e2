"""GGUF v3 writer + ggml block quantisers (numpy).  Host-side tooling of the product.

Counterpart of the reference's offline producers: the HF->GGUF converter
(/root/reference/scripts/dinov2-to-gguf.py:49-166, gguf-py 0.14.0 GGUFWriter pinned in
/root/reference/uv.lock:161-162) and the quantiser (/root/reference/dinov2.cpp:355-453 calling
ggml_quantize_chunk).  Used to write synthetic / fixture models in the exact on-disk schema the
loader (csrc/gguf_reader.cpp) accepts: key order, tensor dtypes (>=2-D weights F16, everything else
F32), numpy-shape -> reversed ne, 32-byte alignment.

Quantisers restate ggml-quants.c `quantize_row_{q4_0,q4_1,q5_0,q5_1,q8_0}_ref` (ggml-org/ggml,
un-vendored submodule of the reference, SHA unpinned).
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
# quant block encoders (rows = ne0/32 consecutive blocks)
# ----------------------------------------------------------------------------------------
def _f16(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16)


def quantize(x: np.ndarray, gtype: int) -> np.ndarray:
    """float32 array (last dim multiple of 32) -> uint8 array [..., nblocks*block_bytes]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    assert x.shape[-1] % QK == 0, "quantised rows must be a multiple of 32 wide"
    lead = x.shape[:-1]
    xb = x.reshape(-1, QK)  # [nb, 32]
    nb = xb.shape[0]
    lo, hi = xb[:, :16], xb[:, 16:]
    if gtype == GGML_Q8_0:
        amax = np.abs(xb).max(axis=1)
        d = (amax / np.float32(127.0)).astype(np.float32)
        idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), np.float32(0)).astype(np.float32)
        v = xb * idv[:, None]
        q = np.sign(v) * np.floor(np.abs(v) + np.float32(0.5))  # roundf: half away from zero
        out = np.zeros((nb, 34), np.uint8)
        out[:, 0:2] = _f16(d).view(np.uint8).reshape(nb, 2)
        out[:, 2:] = q.astype(np.int8).view(np.uint8)
    elif gtype in (GGML_Q4_0, GGML_Q5_0):
        idx = np.abs(xb).argmax(axis=1)
        mx = xb[np.arange(nb), idx]  # signed value with the largest magnitude
        div = np.float32(-8.0 if gtype == GGML_Q4_0 else -16.0)
        d = (mx / div).astype(np.float32)
        idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), np.float32(0)).astype(np.float32)
        off = np.float32(8.5 if gtype == GGML_Q4_0 else 16.5)
        top = 15 if gtype == GGML_Q4_0 else 31
        q0 = np.minimum(top, (lo * idv[:, None] + off).astype(np.int8).astype(np.int32)).astype(np.uint8)
        q1 = np.minimum(top, (hi * idv[:, None] + off).astype(np.int8).astype(np.int32)).astype(np.uint8)
        if gtype == GGML_Q4_0:
            out = np.zeros((nb, 18), np.uint8)
            out[:, 0:2] = _f16(d).view(np.uint8).reshape(nb, 2)
            out[:, 2:] = q0 | (q1 << 4)
        else:
            out = np.zeros((nb, 22), np.uint8)
            out[:, 0:2] = _f16(d).view(np.uint8).reshape(nb, 2)
            qh = np.zeros(nb, np.uint32)
            for j in range(16):
                qh |= ((q0[:, j].astype(np.uint32) & 0x10) >> 4) << j
                qh |= ((q1[:, j].astype(np.uint32) & 0x10) >> 4) << (j + 16)
            out[:, 2:6] = qh.view(np.uint8).reshape(nb, 4)
            out[:, 6:] = (q0 & 0xF) | ((q1 & 0xF) << 4)
    elif gtype in (GGML_Q4_1, GGML_Q5_1):
        mn, mx = xb.min(axis=1), xb.max(axis=1)
        lv = np.float32(15.0 if gtype == GGML_Q4_1 else 31.0)
        d = ((mx - mn) / lv).astype(np.float32)
        idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), np.float32(0)).astype(np.float32)
        x0 = (lo - mn[:, None]) * idv[:, None] + np.float32(0.5)
        x1 = (hi - mn[:, None]) * idv[:, None] + np.float32(0.5)
        if gtype == GGML_Q4_1:
            q0 = np.minimum(15, x0.astype(np.int8).astype(np.int32)).astype(np.uint8)
            q1 = np.minimum(15, x1.astype(np.int8).astype(np.int32)).astype(np.uint8)
            out = np.zeros((nb, 20), np.uint8)
            out[:, 0:2] = _f16(d).view(np.uint8).reshape(nb, 2)
            out[:, 2:4] = _f16(mn).view(np.uint8).reshape(nb, 2)
            out[:, 4:] = q0 | (q1 << 4)
        else:
            q0 = x0.astype(np.uint8)
            q1 = x1.astype(np.uint8)
            out = np.zeros((nb, 24), np.uint8)
            out[:, 0:2] = _f16(d).view(np.uint8).reshape(nb, 2)
            out[:, 2:4] = _f16(mn).view(np.uint8).reshape(nb, 2)
            qh = np.zeros(nb, np.uint32)
            for j in range(16):
                qh |= ((q0[:, j].astype(np.uint32) & 0x10) >> 4) << j
                qh |= ((q1[:, j].astype(np.uint32) & 0x10) >> 4) << (j + 16)
            out[:, 4:8] = qh.view(np.uint8).reshape(nb, 4)
            out[:, 8:] = (q0 & 0xF) | ((q1 & 0xF) << 4)
    else:
        raise ValueError(f"not a quantised type: {gtype}")
    return out.reshape(*lead, -1)


# ----------------------------------------------------------------------------------------
# writer
# ----------------------------------------------------------------------------------------
def _wstr(s: str | bytes) -> bytes:
    b = s.encode("utf-8") if isinstance(s, str) else s
    return struct.pack("<Q", len(b)) + b


class GGUFWriter:
    """Minimal GGUF v3 writer (subset of gguf-py 0.14 that dinov2-to-gguf.py:49-166 uses)."""

    def __init__(self, arch: str = "dinov2", alignment: int = DEFAULT_ALIGNMENT):
        self.kvs: list[tuple[str, int, object]] = []
        self.tensors: list[tuple[str, tuple, int, bytes]] = []
        self.alignment = alignment
        self.add_string("general.architecture", arch)

    def add_string(self, key, val):
        self.kvs.append((key, T_STR, val))

    def add_uint32(self, key, val):
        self.kvs.append((key, T_U32, int(val)))

    def add_float32(self, key, val):
        self.kvs.append((key, T_F32, float(val)))

    def add_tensor(self, name: str, arr: np.ndarray, gtype: int | None = None):
        """`arr` in numpy (row-major) shape; ne = reversed(shape), as gguf-py does."""
        if gtype is None:
            gtype = {np.dtype(np.float32): GGML_F32, np.dtype(np.float16): GGML_F16}[arr.dtype]
        if gtype in (GGML_F32, GGML_F16):
            data = np.ascontiguousarray(arr, dtype=np.float32 if gtype == GGML_F32 else np.float16).tobytes()
        elif gtype == GGML_BF16:
            u = np.ascontiguousarray(arr, dtype=np.float32).view(np.uint32)
            u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
            data = u.tobytes()
        else:
            data = quantize(np.asarray(arr, dtype=np.float32), gtype).tobytes()
        self.tensors.append((name, tuple(arr.shape), gtype, data))

    def add_raw_tensor(self, name, shape, gtype, data: bytes):
        self.tensors.append((name, tuple(shape), gtype, bytes(data)))

    def write(self, path: str):
        a = self.alignment
        out = bytearray()
        out += GGUF_MAGIC + struct.pack("<IQQ", GGUF_VERSION, len(self.tensors), len(self.kvs))
        for key, typ, val in self.kvs:
            out += _wstr(key) + struct.pack("<I", typ)
            out += _wstr(val) if typ == T_STR else struct.pack(_SCALAR_FMT[typ], val)
        off = 0
        offsets = []
        for name, shape, gtype, data in self.tensors:
            ne = tuple(reversed(shape))
            out += _wstr(name) + struct.pack("<I", len(ne)) + struct.pack(f"<{len(ne)}Q", *ne)
            out += struct.pack("<IQ", gtype, off)
            offsets.append(off)
            off += (len(data) + a - 1) // a * a
        out += b"\0" * ((-len(out)) % a)
        for (name, shape, gtype, data), o in zip(self.tensors, offsets):
            out += data + b"\0" * ((-len(data)) % a)
        with open(path, "wb") as f:
            f.write(out)


