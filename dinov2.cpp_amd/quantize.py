"""GGUF -> quantised GGUF re-writer: counterpart of the reference's `quantize` tool.

Mirrors dino_model_quantize / do_quantize (/root/reference/dinov2.cpp:227-236, 355-453; CLI /root/reference/quantize.cpp:24-36,
usage `./quantize model-f16.gguf model-q4_0.gguf 2`): every tensor whose name matches `.*weight` AND is 2-D is re-encoded to
ggml type `itype` (2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1, 8 q8_0 -- /root/reference/README.md:342-346); everything else (the 4-D conv
kernel, 1-D tensors, cls/pos/register embeddings) is copied as is; all KVs are copied and `ftype` is overwritten with `itype`
(dinov2.cpp:377).  Offline CPU tool (numpy); the loader dequantises the result on the device.

    python -m ... quantize.py in.gguf out.gguf 8
"""
from __future__ import annotations

import re
import struct
import sys

import numpy as np

from . import gguf_writer as gw

PATTERN = re.compile(r".*weight")  # dinov2.h:18
_ITYPES = {2: "q4_0", 3: "q4_1", 6: "q5_0", 7: "q5_1", 8: "q8_0"}
_FMT = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<?", 10: "<Q", 11: "<q", 12: "<d"}


def _read(path):
    """Minimal GGUF v2/v3 parse: ordered KVs [(key, type, value)], tensors [(name, ne, gtype, raw bytes)]."""
    buf = open(path, "rb").read()
    if buf[:4] != b"GGUF":
        raise ValueError(f"{path}: not a GGUF file")
    pos = 4
    version, = struct.unpack_from("<I", buf, pos); pos += 4
    n_tensors, n_kv = struct.unpack_from("<QQ", buf, pos); pos += 16

    def rstr():
        nonlocal pos
        n, = struct.unpack_from("<Q", buf, pos); pos += 8
        s = buf[pos:pos + n]; pos += n
        return s.decode("utf-8")

    kvs = []
    for _ in range(n_kv):
        k = rstr()
        t, = struct.unpack_from("<I", buf, pos); pos += 4
        if t == 8:
            v = rstr()
        elif t in _FMT:
            v, = struct.unpack_from(_FMT[t], buf, pos); pos += struct.calcsize(_FMT[t])
        else:
            raise ValueError(f"unsupported KV type {t} for key {k}")
        kvs.append((k, t, v))
    infos = []
    for _ in range(n_tensors):
        name = rstr()
        nd, = struct.unpack_from("<I", buf, pos); pos += 4
        ne = struct.unpack_from(f"<{nd}Q", buf, pos); pos += 8 * nd
        gtype, off = struct.unpack_from("<IQ", buf, pos); pos += 12
        infos.append((name, ne, gtype, off))
    align = next((v for k, t, v in kvs if k == "general.alignment"), gw.DEFAULT_ALIGNMENT)
    data0 = (pos + align - 1) // align * align
    tensors = []
    for name, ne, gtype, off in infos:
        be, bb = gw.TYPE_LAYOUT[gtype]
        n = int(np.prod(ne))
        tensors.append((name, ne, gtype, buf[data0 + off: data0 + off + n // be * bb]))
    return version, kvs, tensors


def do_quantize(name: str, ne) -> bool:
    """dinov2.cpp:227-236: name matches `.*weight` and the tensor is 2-D (ggml_n_dims ignores trailing 1s)."""
    nd = len(ne)
    while nd > 1 and ne[nd - 1] == 1:
        nd -= 1
    return bool(PATTERN.fullmatch(name)) and nd == 2


def dino_model_quantize(fname_inp: str, fname_out: str, itype: int) -> bool:
    if itype not in _ITYPES:
        print(f"dino_model_quantize: invalid quantization type {itype}", file=sys.stderr)  # dinov2.cpp:365-373
        return False
    _, kvs, tensors = _read(fname_inp)
    # the output keeps the input's general.alignment (the KV is copied through below): tensors must be laid out with it
    w = gw.GGUFWriter(arch=next((v for k, t, v in kvs if k == "general.architecture"), "dinov2"),
                      alignment=int(next((v for k, t, v in kvs if k == "general.alignment"), gw.DEFAULT_ALIGNMENT)))
    for k, t, v in kvs:
        if k == "general.architecture":
            continue
        if k == "ftype":
            v = itype
        w.kvs.append((k, t, v))
    total_in = total_out = 0
    for name, ne, gtype, raw in tensors:
        shape = tuple(reversed(ne))
        total_in += len(raw)
        if do_quantize(name, ne):
            if gtype == gw.GGML_F32:
                a = np.frombuffer(raw, np.float32)
            elif gtype == gw.GGML_F16:
                a = np.frombuffer(raw, np.float16).astype(np.float32)
            else:
                raise ValueError(f"unsupported tensor type {gtype} for '{name}'")  # dinov2.cpp:425
            if not np.isfinite(a).all():  # (same rule as csrc/quantize.cpp: a NaN / Inf weight has no block encoding)
                raise ValueError(f"non-finite value in tensor '{name}'")
            q = gw.quantize(a.reshape(-1, ne[0]), itype).tobytes()
            w.add_raw_tensor(name, shape, itype, q)
            total_out += len(q)
        else:
            w.add_raw_tensor(name, shape, gtype, raw)
            total_out += len(raw)
    w.write(fname_out)
    print(f"dino_model_quantize: model size = {total_in / 1048576:8.2f} MB -> quant size = {total_out / 1048576:8.2f} MB "
          f"({_ITYPES[itype]})")
    return True


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) != 4:
        print(f"usage: {argv[0]} /path/to/model-f16.gguf /path/to/model-quant.gguf type\n"
              "  type = 2 - q4_0\n  type = 3 - q4_1\n  type = 6 - q5_0\n  type = 7 - q5_1\n  type = 8 - q8_0", file=sys.stderr)
        return 1
    return 0 if dino_model_quantize(argv[1], argv[2], int(argv[3])) else 1


if __name__ == "__main__":
    raise SystemExit(main())
