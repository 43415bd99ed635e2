"""CPU suite, part 2: GGUF writer <-> oracle reader round trips, quant block known-answers, the C-ABI library
(loads without a GPU, exports every declared symbol, loader error paths that never reach the device)."""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np
from importlib import import_module
import pytest

from oracle import gguf_np as G
from __graft_entry__ import PKG_NAME

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_writer_reader_roundtrip_schema(pkg, tmp_path):
    """Synthetic checkpoint in the converter's schema (dinov2-to-gguf.py:49-166): KV order, dtypes, reversed ne."""
    p = str(tmp_path / "t.gguf")
    hp = pkg.synth.write_synthetic_gguf(p, "tiny", registers=4, num_classes=5, seed=1)
    f = G.GGUFFile(p)
    keys = list(f.kv)
    assert keys[0] == "general.architecture" and f.kv[keys[0]] == "dinov2"
    assert keys[1:6] == ["0", "1", "2", "3", "4"]  # id2label strings come before the u32 hparams
    assert keys[6:] == ["hidden_size", "num_hidden_layers", "num_attention_heads", "num_classes", "patch_size",
                        "img_size", "ftype", "num_register_tokens"]
    assert f.u32("hidden_size") == 128 and f.u32("ftype") == 1 and hp["registers"] == 4
    t = f.tensors
    assert t["embeddings.patch_embeddings.projection.weight"].ne == (14, 14, 3, 128)
    assert t["embeddings.patch_embeddings.projection.weight"].gtype == G.GGML_F16
    assert t["embeddings.patch_embeddings.projection.bias"].ne == (1, 1, 128, 1)
    assert t["embeddings.position_embeddings"].ne == (128, 26, 1) and t["embeddings.position_embeddings"].gtype == G.GGML_F32
    assert t["encoder.layer.0.attention.attention.qkv.weight"].ne == (128, 384)
    assert t["encoder.layer.1.mlp.fc2.weight"].ne == (512, 128)
    assert t["classifier.weight"].ne == (256, 5)
    assert len(t) == 3 + 2 + 2 * 14 + 2 + 2  # 14 tensors per layer; ViT-S no-reg classifier: 2 + 2 + 12*14 + 4 = 176


def test_tensor_data_alignment(pkg, tmp_path):
    """Tensor data starts on a 32-byte boundary and every tensor offset is 32-aligned (general.alignment default)."""
    p = str(tmp_path / "t.gguf")
    pkg.synth.write_synthetic_gguf(p, "tiny", registers=0, num_classes=3, seed=2)
    raw = open(p, "rb").read()
    assert raw[:4] == b"GGUF" and struct.unpack_from("<I", raw, 4)[0] == 3
    f = G.GGUFFile(p)
    whole = np.frombuffer(raw, np.uint8)
    for t in f.tensors.values():
        n = t.raw.size
        # locate the tensor bytes in the file image: offset must be a multiple of 32
        off = t.raw.__array_interface__["data"][0] - t.raw.base.__array_interface__["data"][0] if t.raw.base is not None else 0
        assert off % 32 == 0 and np.array_equal(whole[off:off + n], t.raw)
    w = f.tensors["encoder.layer.0.mlp.fc1.weight"].to_f32()
    assert w.shape == (512, 128) and np.isfinite(w).all() and 0.015 < w.std() < 0.025


def test_alignment_rule_is_one_for_loader_and_quantiser(api, pkg, golden_dir, tmp_path):
    """general.alignment must be a power of two in [1, 2^20] for BOTH native GGUF parsers (the model loader and dinov2_hip_quantize share
    gguf_alignment_ok): 48 and 2^21 are refused by both with a format error, before any device is touched; 64 loads."""
    gw = pkg.gguf_writer
    src = G.GGUFFile(os.path.join(golden_dir, "tiny_gelu_noreg.gguf"))
    w = gw.GGUFWriter(arch="dinov2", alignment=64)
    for k, v in src.kv.items():
        if isinstance(v, (int, np.integer)) and not isinstance(v, bool) and k != "general.alignment":
            w.add_uint32(k, int(v))
    w.add_uint32("general.alignment", 64)
    for name, t in src.tensors.items():
        w.add_raw_tensor(name, tuple(reversed(t.ne)), t.gtype, t.raw.tobytes())
    good = str(tmp_path / "align64.gguf")
    w.write(good)
    blob = open(good, "rb").read()
    key = struct.pack("<Q", len("general.alignment")) + b"general.alignment" + struct.pack("<I", 4)
    at = blob.index(key) + len(key)
    assert struct.unpack_from("<I", blob, at)[0] == 64
    err = C.create_string_buffer(256)
    assert api.lib().dinov2_hip_quantize(good.encode(), str(tmp_path / "ok.gguf").encode(), 8, err, 256) == 0, err.value
    for bad in (48, 1 << 21):
        p = tmp_path / f"align{bad}.gguf"
        p.write_bytes(blob[:at] + struct.pack("<I", bad) + blob[at + 4:])
        assert api.lib().dinov2_hip_quantize(str(p).encode(), str(tmp_path / "o.gguf").encode(), 8, err, 256) == 2, bad
        with pytest.raises(api.DinoError) as e:
            api.Model(str(p))
        assert e.value.status == 2 and "alignment" in str(e.value), (bad, str(e.value))


@pytest.mark.parametrize("tname,bb", [("q4_0", 18), ("q4_1", 20), ("q5_0", 22), ("q5_1", 24), ("q8_0", 34)])
def test_quant_roundtrip(pkg, tname, bb):
    gw = pkg.gguf_writer
    gt = gw.NAME_TYPE[tname]
    rng = np.random.default_rng(3)
    x = rng.standard_normal((7, 96)).astype(np.float32)
    q = gw.quantize(x, gt)
    assert q.shape == (7, 3 * bb) and q.dtype == np.uint8
    y = G.dequantize(q, gt, x.shape)
    step = {"q4_0": 1 / 8, "q4_1": 1 / 15, "q5_0": 1 / 16, "q5_1": 1 / 31, "q8_0": 1 / 127}[tname]
    span = np.abs(x).reshape(7, 3, 32).max(-1, keepdims=True).repeat(32, -1).reshape(7, 96) * (2 if tname[-1] == "1" else 1)
    assert (np.abs(x - y) <= span * step * 1.01 + 1e-3).all()  # within one quantisation step per block
    assert np.array_equal(gw.quantize(y, gt), q) or tname in ("q4_1", "q5_1")  # idempotent for the symmetric formats


def test_quant_known_answer_blocks():
    """Hand-computed bytes for one block of each format (SURVEY.md section 8(c) block table)."""
    one = np.float16(1.0).tobytes()          # d = 1.0
    half = np.float16(0.5).tobytes()         # m = 0.5
    # Q8_0: d=1, qs = -16..15  -> w = qs
    blk = one + bytes((v & 0xFF) for v in range(-16, 16))
    np.testing.assert_array_equal(G.dequantize(np.frombuffer(blk, np.uint8), G.GGML_Q8_0, (32,)), np.arange(-16, 16))
    # Q4_0: qs[j] = j | ((15-j) << 4): w[j] = j - 8, w[j+16] = 7 - j
    qs = bytes(j | ((15 - j) << 4) for j in range(16))
    exp = np.concatenate([np.arange(16) - 8, 7 - np.arange(16)]).astype(np.float32)
    np.testing.assert_array_equal(G.dequantize(np.frombuffer(one + qs, np.uint8), G.GGML_Q4_0, (32,)), exp)
    # Q4_1: w = nib * d + m
    np.testing.assert_array_equal(G.dequantize(np.frombuffer(one + half + qs, np.uint8), G.GGML_Q4_1, (32,)), exp + 8.5)
    # Q5_0: fifth bits: bit j for w[j], bit j+16 for w[j+16]; set them for even j only
    qh = sum((1 << j) | (1 << (j + 16)) for j in range(0, 16, 2))
    hi = np.where(np.arange(16) % 2 == 0, 16, 0)
    exp5 = np.concatenate([np.arange(16) + hi, (15 - np.arange(16)) + hi]).astype(np.float32)
    np.testing.assert_array_equal(G.dequantize(np.frombuffer(one + struct.pack("<I", qh) + qs, np.uint8), G.GGML_Q5_0, (32,)), exp5 - 16)
    np.testing.assert_array_equal(
        G.dequantize(np.frombuffer(one + half + struct.pack("<I", qh) + qs, np.uint8), G.GGML_Q5_1, (32,)), exp5 + 0.5)


def test_quantised_synthetic_gguf_readable(pkg, tmp_path):
    p = str(tmp_path / "q.gguf")
    pkg.synth.write_synthetic_gguf(p, "tiny", registers=4, num_classes=4, seed=4, wtype="q5_1")
    f = G.GGUFFile(p)
    assert f.u32("ftype") == 7
    assert f.tensors["encoder.layer.0.attention.attention.qkv.weight"].gtype == G.GGML_Q5_1
    assert f.tensors["embeddings.patch_embeddings.projection.weight"].gtype == G.GGML_F16  # 4-D stays F16
    assert f.tensors["encoder.layer.0.norm1.weight"].gtype == G.GGML_F32


# ---------------------------------------------------------------------------------------------- C-ABI library
def _declared_symbols():
    names = set()
    for h in ("dinov2_hip.h", "dinov2_hip_ops.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(dinov2_hip_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol(api):
    lib = api.lib()  # loads without a GPU
    assert lib.dinov2_hip_abi_version() == 1
    declared = _declared_symbols()
    assert len(declared) >= 20
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.LIB_PATH], text=True)
    exported = set(re.findall(r" T (dinov2_hip_[a-z0-9_]+)", out))
    assert declared <= exported, f"declared but not exported: {sorted(declared - exported)}"
    assert exported <= declared, f"exported but not declared in include/: {sorted(exported - declared)}"


def test_library_has_gfx950_code_object(api):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", api.LIB_PATH], capture_output=True, text=True)
    assert "gfx950" in out.stdout + out.stderr


def test_no_oracle_or_torch_symbols_in_product(api):
    """The product must not link the oracle, torch or any BLAS: hand-written kernels + the HIP runtime only."""
    out = subprocess.check_output(["readelf", "-d", api.LIB_PATH], text=True)
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert any("amdhip64" in n for n in needed)
    assert not any(s in n for n in needed for s in ("oracle", "torch", "blas", "ggml", "opencv", "MIOpen"))


def test_load_errors_before_touching_the_device(api, tmp_path):
    with pytest.raises(api.DinoError) as e:
        api.Model("/nonexistent/model.gguf")
    assert e.value.status == 1  # ERR_IO (reference: returns false + stderr, dinov2.cpp:269-272)
    bad = tmp_path / "bad.gguf"
    bad.write_bytes(b"NOTGGUF" + b"\0" * 64)
    with pytest.raises(api.DinoError) as e:
        api.Model(str(bad))
    assert e.value.status == 2  # ERR_FORMAT
    trunc = tmp_path / "trunc.gguf"
    trunc.write_bytes(b"GGUF" + struct.pack("<IQQ", 3, 5, 0))
    with pytest.raises(api.DinoError) as e:
        api.Model(str(trunc))
    assert e.value.status == 2


def test_missing_kv_is_a_status_not_an_assert(api, pkg, tmp_path):
    """The reference asserts on a missing hparam key (dinov2.cpp:58); the C-ABI returns ERR_FORMAT with the key name."""
    w = pkg.gguf_writer.GGUFWriter()
    w.add_uint32("hidden_size", 128)
    w.add_tensor("x", np.zeros(4, np.float32))
    p = tmp_path / "nokeys.gguf"
    w.write(str(p))
    with pytest.raises(api.DinoError) as e:
        api.Model(str(p))
    assert e.value.status == 2 and "num_hidden_layers" in str(e.value)


def test_product_fails_loudly_without_library(api, monkeypatch):
    monkeypatch.setattr(api, "_lib", None)
    monkeypatch.setattr(api, "LIB_PATH", "/nonexistent/libdinov2_hip.so")
    with pytest.raises(api.HipLibraryMissing):
        api.lib()


def test_flops_formula_matches_baseline(pkg):
    """SURVEY.md section 8(d) / BASELINE.md section 4 per-image algorithmic GFLOP."""
    s = pkg.synth
    assert abs(s.flops_per_image(s.CONFIGS["large"], 518, 518, 4, 1000) / 1e9 - 1017.1) < 0.2
    assert abs(s.flops_per_image(s.CONFIGS["base"], 518, 518, 4, 1000) / 1e9 - 304.2) < 0.2
    assert abs(s.flops_per_image(s.CONFIGS["giant"], 518, 518, 4, 1000) / 1e9 - 3578.4) < 0.5
    assert abs(s.flops_per_image(s.CONFIGS["small"], 224, 224, 0, 1000) / 1e9 - 12.2) < 0.1


def _build_compat_smoke(tmp_path):
    exe = str(tmp_path / "compat_smoke")
    libdir = os.path.join(ROOT, "dinov2.cpp_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "compat_smoke.cpp"),
                           "-o", exe, "-L" + libdir, "-ldinov2_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_compat_shim_compiles_with_plain_gxx(tmp_path):
    """include/dinov2_compat.hpp (reference-shaped dino_model_load / dino_predict) builds with g++ and links the C-ABI;
    load failure reports like the reference (message on stderr, false)."""
    exe = _build_compat_smoke(tmp_path)
    r = subprocess.run([exe, "/nonexistent.gguf"], capture_output=True, text=True)
    assert r.returncode == 1 and "failed to open" in r.stderr


def _build_compat_opencv(tmp_path):
    exe = str(tmp_path / "compat_cv")
    libdir = os.path.join(ROOT, "dinov2.cpp_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "cpp", "opencv_stub"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "compat_opencv_smoke.cpp"), "-o", exe,
                           "-L" + libdir, "-ldinov2_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_compat_opencv_branch_compiles(tmp_path):
    """DINOV2_WITH_OPENCV + DINOV2_COMPAT_GGML_NAMES: the reference's `inference` main, call for call (cv::Mat / cv::Size
    overloads of dinov2.h:93-112, dino_output::patch_tokens as a cv::Mat, the ggml_* lines of inference.cpp:60-73), compiles
    warning-free against a minimal stand-in for the few cv:: types involved (this image has no OpenCV) and fails like the
    reference on a missing model file."""
    exe = _build_compat_opencv(tmp_path)
    r = subprocess.run([exe, "-m", "/nonexistent.gguf"], capture_output=True, text=True)
    assert r.returncode == 1 and "failed to open" in r.stderr and "failed to load model from" in r.stderr


@pytest.mark.gpu
def test_cpp_compat_opencv_branch_runs(tmp_path, golden_dir):
    exe = _build_compat_opencv(tmp_path)
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    r = subprocess.run([exe, "-m", gguf], capture_output=True, text=True)  # features: 90x123 -> 98x126 -> 7 x 9 patches
    assert r.returncode == 0 and "patch_tokens: 63 x 128 type 5 (input 98 x 126)" in r.stdout, r.stdout + r.stderr
    assert "qntvr                  = 0" in r.stdout and "graph computation took" in r.stderr
    r = subprocess.run([exe, "-m", gguf, "-c", "-k", "3"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.count(" > label_") == 3, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_compat_shim_runs(tmp_path, golden_dir):
    exe = _build_compat_smoke(tmp_path)
    r = subprocess.run([exe, os.path.join(golden_dir, "tiny_gelu_reg4.gguf"), "classify"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.count(" > label_") == 3 and "hidden_size            = 128" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe, os.path.join(golden_dir, "tiny_gelu_reg4.gguf")], capture_output=True, text=True)
    assert r.returncode == 0 and "patch_tokens: 30 x 128" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("itype,tname", [(2, "q4_0"), (3, "q4_1"), (6, "q5_0"), (7, "q5_1"), (8, "q8_0")])
def test_quantize_tool(pkg, tmp_path, itype, tname):
    """`quantize` counterpart (dinov2.cpp:355-453): only 2-D `*.weight` tensors change type, the conv kernel / 1-D /
    embedding tensors are copied, ftype KV is overwritten, and the blocks equal a direct quantisation of the f16 values."""
    from importlib import import_module
    from __graft_entry__ import PKG_NAME
    Q = import_module(PKG_NAME + ".quantize")
    src, dst = str(tmp_path / "f16.gguf"), str(tmp_path / f"{tname}.gguf")
    pkg.synth.write_synthetic_gguf(src, "tiny", registers=4, num_classes=8, seed=9)
    assert Q.dino_model_quantize(src, dst, itype)
    a, b = G.GGUFFile(src), G.GGUFFile(dst)
    assert list(a.kv) == list(b.kv) and b.u32("ftype") == itype and a.u32("ftype") == 1
    assert list(a.tensors) == list(b.tensors)
    for name, ta in a.tensors.items():
        tb = b.tensors[name]
        assert ta.ne == tb.ne
        if name.endswith("weight") and len([d for d in ta.ne if d > 1]) == 2 and "patch_embeddings" not in name \
                and "norm" not in name:
            assert tb.gtype == itype, name
            assert np.array_equal(tb.raw, pkg.gguf_writer.quantize(ta.to_f32(), itype).reshape(-1)), name
        else:
            assert tb.gtype == ta.gtype and np.array_equal(ta.raw, tb.raw), name
    assert not Q.dino_model_quantize(src, dst, 5)  # invalid type id -> False, like dinov2.cpp:365-373


def test_loader_survives_corrupted_files(api, golden_dir, tmp_path):
    """Truncations at every structural boundary and a few hundred random byte/field corruptions of a valid fixture: the
    loader must come back with a status (never crash, hang or allocate absurd amounts) -- the reference asserts or
    segfaults on such files (dinov2.cpp:58, gguf_init_from_file).  Runs without a GPU: parse errors are reported before
    the device is touched; files that still parse end in ERR_HIP here (no device) or load fine on a GPU box."""
    good = open(os.path.join(golden_dir, "tiny_gelu_reg4.gguf"), "rb").read()
    rng = np.random.default_rng(2024)
    cases = []
    for cut in list(range(0, 64)) + list(range(64, 4096, 37)) + [len(good) // 2, len(good) - 1]:
        cases.append(good[:cut])
    for _ in range(150):  # single-byte flips in the header / KV / tensor-info region (first 16 KiB)
        b = bytearray(good)
        pos = int(rng.integers(4, min(len(good), 16384)))
        b[pos] ^= int(rng.integers(1, 256))
        cases.append(bytes(b))
    for off in (8, 16):  # absurd tensor / KV counts
        for val in (2**63, 2**40, 10**7):
            b = bytearray(good)
            b[off:off + 8] = struct.pack("<Q", val)
            cases.append(bytes(b))
    p = tmp_path / "fuzz.gguf"
    statuses = set()
    for data in cases:
        p.write_bytes(data)
        try:
            m = api.Model(str(p), classify=True)
            del m
            statuses.add(0)
        except api.DinoError as e:
            assert e.status in (1, 2, 3, 4, 5, 6), e
            assert str(e)  # every failure carries a message
            statuses.add(e.status)
    assert 2 in statuses  # format errors were exercised


@pytest.mark.parametrize("itype", [2, 3, 6, 7, 8])
def test_native_quantize_equals_python_tool(api, pkg, golden_dir, tmp_path, itype):
    """dinov2_hip_quantize (C++, behind the C-ABI and the shim's dino_model_quantize) writes the same bytes as the numpy tool
    whose quantisers the oracle's dequantisers are pinned against -- for an f16 fixture and for a model holding f32 weights."""
    import ctypes as C
    q = import_module(PKG_NAME + ".quantize")
    srcs = [os.path.join(golden_dir, "tiny_swiglu_reg4.gguf")]
    f32 = str(tmp_path / "f32.gguf")
    w = pkg.gguf_writer.GGUFWriter()
    rng = np.random.default_rng(itype)
    w.add_uint32("ftype", 0)
    w.add_string("note", "f32 weights")
    w.add_tensor("a.weight", rng.standard_normal((48, 64)).astype(np.float32) * 3)
    w.add_tensor("a.bias", rng.standard_normal(48).astype(np.float32))
    w.add_tensor("embeddings.patch_embeddings.projection.weight", rng.standard_normal((8, 3, 14, 14)).astype(np.float16))
    w.add_tensor("z.weight", np.zeros((4, 32), np.float32))  # all-zero blocks: d = 0 path
    w.write(f32)
    srcs.append(f32)
    for k, src in enumerate(srcs):
        a, b = str(tmp_path / f"py{k}.gguf"), str(tmp_path / f"cc{k}.gguf")
        assert q.dino_model_quantize(src, a, itype)
        err = C.create_string_buffer(256)
        fn = api.lib().dinov2_hip_quantize
        fn.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_size_t]
        assert fn(src.encode(), b.encode(), itype, err, 256) == 0, err.value
        assert open(a, "rb").read() == open(b, "rb").read()
    err = C.create_string_buffer(256)
    assert api.lib().dinov2_hip_quantize(srcs[0].encode(), str(tmp_path / "x.gguf").encode(), 5, err, 256) == 4  # invalid type
    assert api.lib().dinov2_hip_quantize(b"/nonexistent.gguf", str(tmp_path / "x.gguf").encode(), 8, err, 256) == 1


@pytest.mark.parametrize("H", [8, 33, 384, 1024])
def test_pca_ritz_step_matches_eigh(pkg, H):
    """The host half of dinov2_hip_pca3 (CholeskyQR of the block + 8 x 8 Rayleigh-Ritz by Jacobi, csrc/model.cpp pca_ritz): a
    block spanning the top-8 eigenspace of a symmetric matrix, arbitrarily mixed, must give back its three leading
    eigenvectors and eigenvalues -- unit length, sign convention 'largest loading positive'.  No device call."""
    api = import_module(pkg.__name__ + ".api")
    rng = np.random.default_rng(H)
    q = np.linalg.qr(rng.standard_normal((H, H)))[0]
    w = 100.0 * 0.8 ** np.arange(H)
    cov = (q * w) @ q.T
    y = q[:, :8] @ (rng.standard_normal((8, 8)) + 3 * np.eye(8))          # same span, not orthonormal
    gram = y.T @ y
    r = np.linalg.cholesky(gram).T                                          # gram = R^T R, R upper with positive diagonal
    ynext = cov @ (y @ np.linalg.inv(r))
    evals, comp = np.empty(3), np.empty((3, H))
    args = [np.ascontiguousarray(a, dtype=np.float64) for a in (y, ynext, gram)]
    assert api.lib().dinov2_hip_op_pca_ritz(*(a.ctypes.data for a in args), H, evals.ctypes.data, comp.ctypes.data) == 0
    np.testing.assert_allclose(evals, w[:3], rtol=1e-10)
    np.testing.assert_allclose(np.linalg.norm(comp, axis=1), 1.0, atol=1e-12)
    for k in range(3):
        assert abs(float(comp[k] @ q[:, k])) >= 1 - 1e-10
        assert comp[k][np.abs(comp[k]).argmax()] > 0
    # a rank-deficient block (two equal columns) stays finite: the dependent direction is dropped
    y2 = y.copy(); y2[:, 7] = y2[:, 6]
    g2 = y2.T @ y2
    a2 = [np.ascontiguousarray(a, dtype=np.float64) for a in (y2, ynext, g2)]
    assert api.lib().dinov2_hip_op_pca_ritz(*(a.ctypes.data for a in a2), H, evals.ctypes.data, comp.ctypes.data) == 0
    assert np.isfinite(evals).all() and np.isfinite(comp).all()
    assert api.lib().dinov2_hip_op_pca_ritz(*(a.ctypes.data for a in args), 4, evals.ctypes.data, comp.ctypes.data) != 0


def _patch_tensor_offset(blob: bytes, tensor: str, new_off: int) -> bytes:
    """Overwrite the u64 data offset in the tensor-info record of `tensor` (name, u32 n_dims, n_dims x u64, u32 type, u64 off)."""
    key = struct.pack("<Q", len(tensor)) + tensor.encode()
    at = blob.index(key) + len(key)
    nd, = struct.unpack_from("<I", blob, at)
    at += 4 + 8 * nd + 4
    return blob[:at] + struct.pack("<Q", new_off) + blob[at + 8:]


@pytest.mark.parametrize("off", [2 ** 64 - 64, 2 ** 63, 2 ** 40])
def test_crafted_tensor_offset_is_rejected(api, golden_dir, tmp_path, off):
    """A tensor offset chosen so that data0 + offset + nbytes wraps around 2^64 (or simply lies far outside the file) must be a
    FORMAT error in the loader and in the quantiser, before anything dereferences it (advisor finding, round 1)."""
    blob = open(os.path.join(golden_dir, "tiny_gelu_noreg.gguf"), "rb").read()
    evil = tmp_path / "evil.gguf"
    evil.write_bytes(_patch_tensor_offset(blob, "embeddings.cls_token", off))
    with pytest.raises(api.DinoError) as e:
        api.Model(str(evil))
    assert e.value.status == 2 and "cls_token" in str(e.value)
    err = C.create_string_buffer(256)
    assert api.lib().dinov2_hip_quantize(str(evil).encode(), str(tmp_path / "o.gguf").encode(), 8, err, 256) == 2
    # an element count whose product wraps is refused as well
    key = struct.pack("<Q", len("embeddings.cls_token")) + b"embeddings.cls_token"
    at = blob.index(key) + len(key) + 4
    huge = blob[:at] + struct.pack("<Q", 2 ** 62) + struct.pack("<Q", 8) + blob[at + 16:]
    evil.write_bytes(huge)
    with pytest.raises(api.DinoError) as e:
        api.Model(str(evil))
    assert e.value.status == 2


@pytest.mark.parametrize("itype", [2, 8])
def test_quantize_keeps_a_non_default_alignment(api, pkg, golden_dir, tmp_path, itype):
    """A GGUF written with general.alignment = 64: both quantisers lay the output out with 64 too (the KV is copied through),
    write identical bytes, and every tensor reads back as the quantised original (advisor finding, round 1: the output used
    to be 32-aligned under a KV that said 64 -> silently corrupted tensors)."""
    from oracle import gguf_np as G
    Q = import_module(PKG_NAME + ".quantize")
    gw = pkg.gguf_writer
    _, kvs, tensors = Q._read(os.path.join(golden_dir, "tiny_gelu_reg4.gguf"))
    w = gw.GGUFWriter(arch="dinov2", alignment=64)
    for k, t, v in kvs:
        if k != "general.architecture":
            w.kvs.append((k, t, v))
    w.add_uint32("general.alignment", 64)
    for name, ne, gtype, raw in tensors:
        w.add_raw_tensor(name, tuple(reversed(ne)), gtype, raw)
    src = str(tmp_path / "al64.gguf")
    w.write(src)
    a, b = str(tmp_path / "py.gguf"), str(tmp_path / "cc.gguf")
    assert Q.dino_model_quantize(src, a, itype)
    err = C.create_string_buffer(256)
    assert api.lib().dinov2_hip_quantize(src.encode(), b.encode(), itype, err, 256) == 0, err.value
    assert open(a, "rb").read() == open(b, "rb").read()
    fa, fs = G.GGUFFile(a), G.GGUFFile(src)
    assert fa.kv["general.alignment"] == 64
    for name, ts in fs.tensors.items():
        ta = fa.tensors[name]
        if Q.do_quantize(name, ts.ne):
            assert ta.gtype == itype
            assert np.array_equal(ta.raw, gw.quantize(ts.to_f32(), itype).reshape(-1)), name
        else:
            assert np.array_equal(ta.to_f32(), ts.to_f32()), name


def test_gguf_reader_and_quantiser_survive_mutants_under_sanitizers(golden_dir, tmp_path):
    """tests/cpp/gguf_fuzz.cpp: the host-side GGUF code (csrc/gguf_reader.cpp, csrc/quantize.cpp) built with AddressSanitizer + UndefinedBehaviorSanitizer,
    fed 3 000 mutants per fixture (byte flips, 4- / 8-byte fields overwritten with extreme values, truncations); every mutant is opened, every
    tensor's first and last data byte is touched the way the loader does, every eighth mutant goes through the quantiser.  A file may be refused --
    with a message -- never read out of bounds.  (Round 6 found one thing that way: NaN weights reached the quantiser's float -> int casts; both
    quantisers now refuse non-finite weights, as ggml's row validation does.)  The sanitizer runtimes come with ROCm's clang; skipped without them."""
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        pytest.skip("ROCm clang not available")
    src = os.path.join(ROOT, "dinov2.cpp_amd", "csrc")
    exe = str(tmp_path / "gguf_fuzz")
    r = subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        os.path.join(ROOT, "tests", "cpp", "gguf_fuzz.cpp"), os.path.join(src, "gguf_reader.cpp"), os.path.join(src, "quantize.cpp"), "-o", exe],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    for seed, name in enumerate(("tiny_gelu_reg4.gguf", "tiny_swiglu_reg4.gguf")):
        r = subprocess.run([exe, os.path.join(golden_dir, name), str(seed + 1), "3000", str(tmp_path)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        last = r.stdout.strip().splitlines()[-1]
        assert last.startswith("mutants 3000") and "refused" in last, last
        opened, refused = int(last.split("opened ")[1].split(",")[0]), int(last.split("refused ")[1].split(",")[0])
        assert opened > 500 and refused > 500, last  # both outcomes were exercised


def test_quantisers_refuse_non_finite_weights(api, pkg, golden_dir, tmp_path):
    """A NaN / Inf weight has no block encoding (and means a damaged file): the native quantiser and the Python tool both refuse it, with the tensor's name."""
    pyq = import_module(PKG_NAME + ".quantize")
    path = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    good = bytearray(open(path, "rb").read())
    t = G.GGUFFile(path).tensors["encoder.layer.0.mlp.fc1.weight"]
    assert t.gtype in (G.GGML_F16, G.GGML_F32)
    pos = bytes(good).find(t.raw.tobytes()[:64])
    assert pos > 0
    nan = struct.pack("<H", 0x7E00) if t.gtype == G.GGML_F16 else struct.pack("<f", float("nan"))
    good[pos:pos + len(nan)] = nan
    bad = tmp_path / "nan.gguf"
    bad.write_bytes(bytes(good))
    err = C.create_string_buffer(256)
    assert api.lib().dinov2_hip_quantize(str(bad).encode(), str(tmp_path / "out.gguf").encode(), 8, err, 256) == 2
    assert b"non-finite" in err.value and b"fc1.weight" in err.value
    with pytest.raises(ValueError, match="non-finite"):
        pyq.dino_model_quantize(str(bad), str(tmp_path / "out2.gguf"), 8)
