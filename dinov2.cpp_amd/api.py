"""ctypes binding of libdinov2_hip.so (include/dinov2_hip.h) + a mirror of the reference's host API.

The product path has NO fallback: if the HIP library is missing or fails to load, importing a model raises
(`HipLibraryMissing`) -- nothing here ever computes on the CPU.

Mirror of /root/reference/dinov2.h (same names, argument meaning and error behaviour where sane):
  dino_params            dinov2.h:57-68     (seed, topk, enable_flash_attn, n_threads, classify, model, ...)
  dino_hparams           dinov2.h:25-47
  dino_model_load(...)   dinov2.h:98-99     -> returns (ok: bool, model)   [reference: bool + out-param]
  dino_predict(...)      dinov2.h:111-112   -> dino_output | None          [reference: unique_ptr, {} on failure]
  dino_output            dinov2.h:85-88     preds / patch_tokens
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DINOV2_HIP_LIB") or os.path.join(_HERE, "libdinov2_hip.so")  # env override: tuning variants only

F16, BF16 = 0, 1
BGR_HWC, RGB_CHW = 0, 1
CLASSIFY = 1

STATUS = {0: "OK", 1: "ERR_IO", 2: "ERR_FORMAT", 3: "ERR_UNSUPPORTED", 4: "ERR_INVALID", 5: "ERR_HIP", 6: "ERR_NO_HEAD"}


class HipLibraryMissing(RuntimeError):
    pass


class DinoError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"{STATUS.get(status, status)}: {msg}")
        self.status = status


class LoadOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("compute_dtype", C.c_int32), ("classify", C.c_int32),
                ("skip_tensor_data", C.c_int32), ("quirk_pool_const_divisor", C.c_int32),
                ("quirk_pool_includes_registers", C.c_int32), ("batch_invariant", C.c_int32), ("ln_fold", C.c_int32), ("reserved", C.c_int32 * 8)]


class HParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("hidden_size", "num_hidden_layers", "num_attention_heads", "num_classes",
                                          "num_register_tokens", "patch_size", "img_size", "ftype")] + \
               [("eps", C.c_float)] + \
               [(n, C.c_uint32) for n in ("ffn_hidden", "swiglu", "has_classifier", "weight_type", "compute_dtype")]


class Input(C.Structure):
    _fields_ = [("data", C.c_void_p), ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("layout", C.c_int32), ("on_device", C.c_int32)]


class Output(C.Structure):
    _fields_ = [("cls", C.c_void_p), ("patch_tokens", C.c_void_p), ("logits", C.c_void_p), ("probs", C.c_void_p),
                ("topk_ids", C.c_void_p), ("topk_probs", C.c_void_p), ("topk", C.c_int32), ("on_device", C.c_int32)]


class GroupOpts(C.Structure):
    _fields_ = [("load", LoadOpts), ("n_devices", C.c_int32), ("devices", C.POINTER(C.c_int32)), ("broadcast", C.c_int32),
                ("streams_per_device", C.c_int32), ("reserved", C.c_int32 * 7)]


_lib = None


def set_tuning(key, value):
    """Testing aid (include/dinov2_hip_ops.h): flip one of the library's tuning switches inside this process; 0 = its own choice."""
    rc = lib().dinov2_hip_op_set_tuning(key.encode(), int(value))
    if rc != 0:
        raise ValueError(f"unknown tuning key {key!r}")


def reset_tuning(key):
    """Back to the value the environment gave the switch when the library first read it (0 if none)."""
    set_tuning(key, -1)


def build_id() -> str:
    """The commit libdinov2_hip.so was built from (dinov2_hip_build_id)."""
    return lib().dinov2_hip_build_id().decode()


def gemm_plan(dtype, epilogue, M, N, K):
    """The kernel plan launch_gemm picks for a shape (text; no device needed)."""
    buf = C.create_string_buffer(256)
    rc = lib().dinov2_hip_op_gemm_plan(int(dtype), int(epilogue), int(M), int(N), int(K), buf, 256)
    if rc != 0:
        raise ValueError(f"launch_gemm refuses dtype={dtype} epilogue={epilogue} M={M} N={N} K={K}")
    return buf.value.decode()


def lib():
    """Load libdinov2_hip.so; raise loudly if it is not built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    try:  # share ONE HIP runtime with torch when torch is in the process (same SONAME libamdhip64.so.7)
        import sys
        if "torch" in sys.modules:
            import torch  # noqa: F401
    except Exception:
        pass
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryMissing(f"cannot load {LIB_PATH}: {e}") from e
    vp, i32, u32, sz, cp = C.c_void_p, C.c_int32, C.c_uint32, C.c_size_t, C.c_char_p
    L.dinov2_hip_abi_version.restype = C.c_int
    L.dinov2_hip_default_load_opts.argtypes = [C.POINTER(LoadOpts)]
    L.dinov2_hip_default_load_opts.restype = None
    L.dinov2_hip_model_load.argtypes = [cp, C.POINTER(LoadOpts), C.POINTER(vp), cp, sz]
    L.dinov2_hip_model_free.argtypes = [vp]
    L.dinov2_hip_model_free.restype = None
    L.dinov2_hip_model_hparams.argtypes = [vp, C.POINTER(HParams)]
    L.dinov2_hip_model_label.argtypes = [vp, i32]
    L.dinov2_hip_model_label.restype = cp
    L.dinov2_hip_model_arena.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.dinov2_hip_session_create.argtypes = [vp, vp, C.POINTER(vp), cp, sz]
    L.dinov2_hip_session_free.argtypes = [vp]
    L.dinov2_hip_session_free.restype = None
    L.dinov2_hip_workspace_bytes.argtypes = [vp, i32, i32, i32]
    L.dinov2_hip_workspace_bytes.restype = sz
    L.dinov2_hip_session_sync.argtypes = [vp]
    L.dinov2_hip_session_stream.argtypes = [vp]
    L.dinov2_hip_session_stream.restype = vp
    L.dinov2_hip_predict.argtypes = [vp, C.POINTER(Input), C.POINTER(Output), u32, cp, sz]
    L.dinov2_hip_default_group_opts.argtypes = [C.POINTER(GroupOpts)]
    L.dinov2_hip_default_group_opts.restype = None
    L.dinov2_hip_group_create.argtypes = [cp, C.POINTER(GroupOpts), C.POINTER(vp), cp, sz]
    L.dinov2_hip_group_free.argtypes = [vp]
    L.dinov2_hip_group_free.restype = None
    L.dinov2_hip_group_size.argtypes = [vp]
    L.dinov2_hip_group_model.argtypes = [vp, i32]
    L.dinov2_hip_group_model.restype = vp
    L.dinov2_hip_group_broadcast_ms.argtypes = [vp]
    L.dinov2_hip_group_broadcast_ms.restype = C.c_double
    L.dinov2_hip_group_describe.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.dinov2_hip_group_predict.argtypes = [vp, C.POINTER(Input), C.POINTER(Output), u32, cp, sz]
    L.dinov2_hip_group_submit.argtypes = [vp, C.POINTER(Input), C.POINTER(Output), u32, C.POINTER(C.c_int64), cp, sz]
    L.dinov2_hip_group_wait.argtypes = [vp, C.c_int64, cp, sz]
    L.dinov2_hip_fetch.argtypes = [vp, C.POINTER(Output), cp, sz]
    L.dinov2_hip_host_alloc.argtypes = [sz]
    L.dinov2_hip_host_alloc.restype = vp
    L.dinov2_hip_host_free.argtypes = [vp]
    L.dinov2_hip_host_free.restype = None
    L.dinov2_hip_interpolate_pos_embed.argtypes = [vp, i32, i32, vp]
    L.dinov2_hip_preprocess_size.argtypes = [i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.dinov2_hip_preprocess.argtypes = [i32, vp, i32, i32, i32, vp]
    L.dinov2_hip_session_profile.argtypes = [vp, i32]
    L.dinov2_hip_session_profile_read.argtypes = [vp, i32, C.POINTER(cp), C.POINTER(C.c_float), C.POINTER(i32)]
    L.dinov2_hip_debug_hidden.argtypes = [vp, C.POINTER(Input), i32, vp, cp, sz]
    L.dinov2_hip_pca3.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, cp, sz]
    # diagnostic ops (include/dinov2_hip_ops.h)
    fp = C.POINTER(C.c_float)
    L.dinov2_hip_op_gemm.argtypes = [i32, i32, fp, fp, fp, fp, C.c_int64, fp, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                                     C.c_float]
    L.dinov2_hip_build_id.restype = C.c_char_p
    L.dinov2_hip_op_gemm_resid_ln.argtypes = [i32, fp, fp, fp, fp, fp, fp, fp, fp, i32, i32, i32]
    L.dinov2_hip_op_gemm_ln_consumer.argtypes = [i32, i32, fp, fp, fp, fp, fp, C.c_float, fp, i32, i32, i32, i32, i32, C.c_float]
    L.dinov2_hip_op_ln_prepare.argtypes = [i32, fp, fp, fp, fp, i32, i32]
    L.dinov2_hip_op_im2col.argtypes = [i32, fp, fp, i32, i32, i32, i32, i32, i32]
    L.dinov2_hip_op_ln_fold_vectors.argtypes = [i32, fp, fp, fp, fp, fp, fp, i32, i32]
    L.dinov2_hip_op_attention.argtypes = [i32, fp, fp, i32, i32, i32, i32]
    L.dinov2_hip_op_layernorm.argtypes = [i32, fp, fp, fp, fp, i32, i32, C.c_float]
    L.dinov2_hip_op_convert_weight.argtypes = [i32, vp, C.c_uint64, u32, fp, i32, i32, i32, i32]
    L.dinov2_hip_op_pca_ritz.argtypes = [vp, vp, vp, i32, vp, vp]
    L.dinov2_hip_op_clock_probe.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.dinov2_hip_op_clock_slots.argtypes = [C.POINTER(C.c_uint64)]
    L.dinov2_hip_op_probe_tr16.argtypes = [C.POINTER(C.c_int16)]
    L.dinov2_hip_op_preprocess_u8.argtypes = [i32, vp, i32, i32, i32, i32, vp]
    L.dinov2_hip_op_gemm_bench.argtypes = [i32] * 6
    L.dinov2_hip_op_gemm_bench.restype = C.c_float
    L.dinov2_hip_op_attention_bench.argtypes = [i32] * 6
    L.dinov2_hip_op_attention_bench.restype = C.c_float
    L.dinov2_hip_op_set_tuning.argtypes = [cp, i32]
    L.dinov2_hip_op_get_tuning.argtypes = [cp]
    L.dinov2_hip_op_gemm_plan.argtypes = [i32, i32, i32, i32, i32, C.c_char_p, i32]
    _lib = L
    return L


def _errbuf():
    return C.create_string_buffer(512)


U8_BGR_HWC = 2


def preprocess_size(mode: int, h: int, w: int, patch: int = 14) -> tuple[int, int]:
    oh, ow = C.c_int32(), C.c_int32()
    if lib().dinov2_hip_preprocess_size(mode, h, w, patch, C.byref(oh), C.byref(ow)) != 0:
        raise DinoError(4, "preprocess_size")
    return oh.value, ow.value


def dino_preprocess(img_bgr_u8: np.ndarray, patch: int = 14) -> np.ndarray:
    """dino_preprocess (dinov2.cpp:135-156) on the host: [h, w, 3] uint8 BGR -> f32 BGR [(h/p+1)*p, (w/p+1)*p, 3]."""
    return _preprocess(0, img_bgr_u8, patch)


def dino_classify_preprocess(img_bgr_u8: np.ndarray, patch: int = 14) -> np.ndarray:
    """dino_classify_preprocess (dinov2.cpp:106-132): resize to 256x256 ignoring aspect, centre-crop 224, normalise."""
    return _preprocess(1, img_bgr_u8, patch)


def _preprocess(mode, img, patch):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 3 and img.shape[2] == 3
    oh, ow = preprocess_size(mode, img.shape[0], img.shape[1], patch)
    out = np.empty((oh, ow, 3), np.float32)
    if lib().dinov2_hip_preprocess(mode, img.ctypes.data, img.shape[0], img.shape[1], patch, out.ctypes.data) != 0:
        raise DinoError(4, "preprocess")
    return out


class Model:
    """dino_model counterpart (owning handle)."""

    def __init__(self, path: str, *, device: int = 0, dtype: int = F16, classify: bool = True,
                 skip_tensor_data: bool = False, pool_const_divisor: bool = True, pool_includes_registers: bool = True,
                 batch_invariant: bool = True, ln_fold: int = 0):
        """ln_fold: 0 = the library's choice, 1 = LayerNorm folded into the neighbouring GEMMs, -1 = separate LayerNorm launches."""
        L = lib()
        o = LoadOpts()
        L.dinov2_hip_default_load_opts(C.byref(o))
        o.device, o.compute_dtype, o.classify = device, dtype, int(classify)
        o.skip_tensor_data = int(skip_tensor_data)
        o.quirk_pool_const_divisor, o.quirk_pool_includes_registers = int(pool_const_divisor), int(pool_includes_registers)
        o.batch_invariant = int(batch_invariant)
        o.ln_fold = int(ln_fold)
        h = C.c_void_p()
        err = _errbuf()
        rc = L.dinov2_hip_model_load(path.encode(), C.byref(o), C.byref(h), err, len(err))
        if rc != 0:
            raise DinoError(rc, err.value.decode(errors="replace"))
        self._h = h
        hp = HParams()
        L.dinov2_hip_model_hparams(h, C.byref(hp))
        self.hparams = hp
        self.device = device

    def label(self, i: int) -> str | None:
        s = lib().dinov2_hip_model_label(self._h, i)
        return s.decode() if s is not None else None

    def arena(self) -> tuple[int, int]:
        p, n = C.c_void_p(), C.c_size_t()
        lib().dinov2_hip_model_arena(self._h, C.byref(p), C.byref(n))
        return int(p.value or 0), int(n.value)

    def workspace_bytes(self, batch, h, w) -> int:
        return int(lib().dinov2_hip_workspace_bytes(self._h, batch, h, w))

    def interpolate_pos_embed(self, h_new, w_new) -> np.ndarray:
        out = np.empty((1 + h_new * w_new, self.hparams.hidden_size), np.float32)
        rc = lib().dinov2_hip_interpolate_pos_embed(self._h, h_new, w_new, out.ctypes.data)
        if rc != 0:
            raise DinoError(rc, "interpolate_pos_embed")
        return out

    def tokens(self, h, w):
        p = self.hparams.patch_size
        return 1 + self.hparams.num_register_tokens + (h // p) * (w // p)

    def close(self):
        if getattr(self, "_h", None):
            lib().dinov2_hip_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _alloc_outputs(hp, B, hh, ww, layout, classify, topk, want):
    """Host output arrays + the filled Output struct for a predict of B images (shared by Session and Group)."""
    Hd, R, ps = hp.hidden_size, hp.num_register_tokens, hp.patch_size
    nh, nw = preprocess_size(1 if classify else 0, hh, ww, ps) if layout == U8_BGR_HWC else (hh, ww)
    P = (nh // ps) * (nw // ps)
    out = {}
    o = Output()
    if "cls" in want:
        out["cls"] = np.empty((B, Hd), np.float32)
        o.cls = out["cls"].ctypes.data
    if "patch_tokens" in want:
        out["patch_tokens"] = np.empty((B, P + (R if classify else 0), Hd), np.float32)
        o.patch_tokens = out["patch_tokens"].ctypes.data
    if classify:
        Cn = hp.num_classes
        if "logits" in want:
            out["logits"] = np.empty((B, Cn), np.float32)
            o.logits = out["logits"].ctypes.data
        if "probs" in want:
            out["probs"] = np.empty((B, Cn), np.float32)
            o.probs = out["probs"].ctypes.data
        if topk > 0:
            out["topk_ids"] = np.empty((B, topk), np.int32)
            out["topk_probs"] = np.empty((B, topk), np.float32)
            o.topk_ids, o.topk_probs, o.topk = out["topk_ids"].ctypes.data, out["topk_probs"].ctypes.data, topk
    return out, o


def pinned_empty(shape, dtype=np.float32) -> np.ndarray:
    """numpy array over page-locked host memory (dinov2_hip_host_alloc): host <-> device copies of such buffers run at the full
    PCIe rate.  The memory is released when the array (and every view of it) is garbage-collected."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = lib().dinov2_hip_host_alloc(max(n, 1))
    if not ptr:
        raise MemoryError(f"dinov2_hip_host_alloc({n}) failed")
    buf = (C.c_char * max(n, 1)).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    import weakref
    weakref.finalize(buf, lib().dinov2_hip_host_free, ptr)
    return arr


class Group:
    """dinov2_hip_group: N devices behind one handle -- host threads + sessions per device inside the library (two lanes per
    device by default: one lane's PCIe copies run under the other's kernels), the global batch split contiguously, outputs
    landing at the shard offsets of the caller's arrays."""

    def __init__(self, path: str, devices=None, *, dtype: int = F16, classify: bool = True, broadcast: bool = True,
                 batch_invariant: bool = True, streams_per_device: int = 2):
        L = lib()
        o = GroupOpts()
        L.dinov2_hip_default_group_opts(C.byref(o))
        o.load.compute_dtype, o.load.classify = dtype, int(classify)
        o.load.batch_invariant = int(batch_invariant)
        o.broadcast = int(broadcast)
        o.streams_per_device = int(streams_per_device)
        if devices is not None:
            self._devs = (C.c_int32 * len(devices))(*devices)
            o.n_devices, o.devices = len(devices), self._devs
        h = C.c_void_p()
        err = _errbuf()
        rc = L.dinov2_hip_group_create(path.encode(), C.byref(o), C.byref(h), err, len(err))
        if rc != 0:
            raise DinoError(rc, err.value.decode(errors="replace"))
        self._h = h
        self.size = int(L.dinov2_hip_group_size(h))
        self.hparams = HParams()
        L.dinov2_hip_model_hparams(L.dinov2_hip_group_model(h, 0), C.byref(self.hparams))
        self.broadcast_ms = float(L.dinov2_hip_group_broadcast_ms(h))
        buf = C.create_string_buffer(8192)
        self.topology = buf.value.decode().splitlines() if L.dinov2_hip_group_describe(h, buf, len(buf)) == 0 else []

    def predict(self, images: np.ndarray, *, classify: bool = False, layout: int = RGB_CHW, topk: int = 0,
                want=("cls", "patch_tokens", "logits", "probs")) -> dict:
        img = np.ascontiguousarray(images, dtype=np.uint8 if layout == U8_BGR_HWC else np.float32)
        if img.ndim == 3:  # one image -> a batch of one, like Session.predict
            img = img[None]
        if img.ndim != 4 or (img.shape[1] if layout == RGB_CHW else img.shape[3]) != 3:
            raise ValueError(f"expected [B, 3, H, W] (RGB_CHW) or [B, H, W, 3] images, got shape {img.shape}")
        B = img.shape[0]
        hh, ww = (img.shape[2], img.shape[3]) if layout == RGB_CHW else (img.shape[1], img.shape[2])
        out, o = _alloc_outputs(self.hparams, B, hh, ww, layout, classify, topk, want)
        i = Input(img.ctypes.data, B, hh, ww, layout, 0)
        err = _errbuf()
        rc = lib().dinov2_hip_group_predict(self._h, C.byref(i), C.byref(o), CLASSIFY if classify else 0, err, len(err))
        if rc != 0:
            raise DinoError(rc, err.value.decode(errors="replace"))
        return out

    def submit(self, images: np.ndarray, *, classify: bool = False, layout: int = RGB_CHW, topk: int = 0,
               want=("cls", "patch_tokens", "logits", "probs")):
        """First half of predict (dinov2_hip_group_submit): returns a pending-job handle at once; up to `streams_per_device` jobs
        may be in flight.  `images` must not be modified until wait() has returned for the handle."""
        img = np.ascontiguousarray(images, dtype=np.uint8 if layout == U8_BGR_HWC else np.float32)
        if img.ndim == 3:
            img = img[None]
        if img.ndim != 4 or (img.shape[1] if layout == RGB_CHW else img.shape[3]) != 3:
            raise ValueError(f"expected [B, 3, H, W] (RGB_CHW) or [B, H, W, 3] images, got shape {img.shape}")
        B = img.shape[0]
        hh, ww = (img.shape[2], img.shape[3]) if layout == RGB_CHW else (img.shape[1], img.shape[2])
        out, o = _alloc_outputs(self.hparams, B, hh, ww, layout, classify, topk, want)
        i = Input(img.ctypes.data, B, hh, ww, layout, 0)
        t = C.c_int64(-1)
        err = _errbuf()
        rc = lib().dinov2_hip_group_submit(self._h, C.byref(i), C.byref(o), CLASSIFY if classify else 0, C.byref(t), err, len(err))
        if rc != 0:
            raise DinoError(rc, err.value.decode(errors="replace"))
        return (t.value, img, out)  # the handle keeps the input and output arrays alive

    def wait(self, handle) -> dict:
        """Second half: blocks until the job's results are in the arrays; handles are waited for in submission order."""
        err = _errbuf()
        rc = lib().dinov2_hip_group_wait(self._h, handle[0], err, len(err))
        if rc != 0:
            raise DinoError(rc, err.value.decode(errors="replace"))
        return handle[2]

    def close(self):
        if getattr(self, "_h", None):
            lib().dinov2_hip_group_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Session:
    """ggml_gallocr_t counterpart: stream + workspace, reusable across predicts."""

    def __init__(self, model: Model, stream: int | None = None):
        self.model = model
        h = C.c_void_p()
        err = _errbuf()
        rc = lib().dinov2_hip_session_create(model._h, C.c_void_p(stream) if stream else None, C.byref(h), err, len(err))
        if rc != 0:
            raise DinoError(rc, err.value.decode(errors="replace"))
        self._h = h

    def predict(self, images: np.ndarray, *, classify: bool = False, layout: int = RGB_CHW, topk: int = 0,
                want=("cls", "patch_tokens", "logits", "probs")) -> dict:
        """images: f32 host array [B,3,H,W] (RGB_CHW) or [B,H,W,3] (BGR_HWC), or RAW uint8 [B,h,w,3] BGR with
        layout=U8_BGR_HWC (preprocessed on the device).  Returns host numpy outputs."""
        img = np.ascontiguousarray(images, dtype=np.uint8 if layout == U8_BGR_HWC else np.float32)
        if img.ndim == 3:
            img = img[None]
        B = img.shape[0]
        hh, ww = (img.shape[2], img.shape[3]) if layout == RGB_CHW else (img.shape[1], img.shape[2])
        out, o = _alloc_outputs(self.model.hparams, B, hh, ww, layout, classify, topk, want)
        i = Input(img.ctypes.data, B, hh, ww, layout, 0)
        err = _errbuf()
        rc = lib().dinov2_hip_predict(self._h, C.byref(i), C.byref(o), CLASSIFY if classify else 0, err, len(err))
        if rc != 0:
            raise DinoError(rc, err.value.decode(errors="replace"))
        return out

    def predict_device(self, img_ptr: int, B: int, hh: int, ww: int, *, classify: bool, layout: int = RGB_CHW,
                       logits_ptr: int = 0, probs_ptr: int = 0, cls_ptr: int = 0, patch_ptr: int = 0):
        """Asynchronous predict on device-resident input/outputs (raw device pointers)."""
        i = Input(img_ptr, B, hh, ww, layout, 1)
        o = Output(cls_ptr or None, patch_ptr or None, logits_ptr or None, probs_ptr or None, None, None, 0, 1)
        err = _errbuf()
        rc = lib().dinov2_hip_predict(self._h, C.byref(i), C.byref(o), CLASSIFY if classify else 0, err, len(err))
        if rc != 0:
            raise DinoError(rc, err.value.decode(errors="replace"))

    def debug_hidden(self, images: np.ndarray, layer: int, layout: int = RGB_CHW) -> np.ndarray:
        img = np.ascontiguousarray(images, dtype=np.float32)
        if img.ndim == 3:
            img = img[None]
        B = img.shape[0]
        hh, ww = (img.shape[2], img.shape[3]) if layout == RGB_CHW else (img.shape[1], img.shape[2])
        T = self.model.tokens(hh, ww)
        out = np.empty((B, T, self.model.hparams.hidden_size), np.float32)
        i = Input(img.ctypes.data, B, hh, ww, layout, 0)
        err = _errbuf()
        rc = lib().dinov2_hip_debug_hidden(self._h, C.byref(i), layer, out.ctypes.data, err, len(err))
        if rc != 0:
            raise DinoError(rc, err.value.decode(errors="replace"))
        return out

    def pca3(self, tokens: np.ndarray | None = None, shape: tuple[int, int] | None = None):
        """Top-3 PCA of a [P, H] token matrix (cv::PCA(DATA_AS_ROW, 3) + project, inference.cpp:76-81), on the device: means,
        covariance (one MFMA GEMM), block iteration for the eigenvectors, projection.  `tokens=None` works on the patch tokens
        the last predict() left on the device (image 0; `shape` = their (P, H)) without moving them.
        Returns (components [3, H], mean [H], projection [P, 3])."""
        if tokens is None:
            P, H = shape
            ptr = None
        else:
            x = np.ascontiguousarray(tokens, dtype=np.float32)
            if x.ndim != 2:
                raise ValueError("pca3: tokens must be [P, H]")
            P, H = x.shape
            ptr = x.ctypes.data
        comp, mean, proj = np.empty((3, H), np.float32), np.empty(H, np.float32), np.empty((P, 3), np.float32)
        err = _errbuf()
        rc = lib().dinov2_hip_pca3(self._h, ptr, P, H, 0, comp.ctypes.data, mean.ctypes.data, proj.ctypes.data, err, len(err))
        if rc != 0:
            raise DinoError(rc, err.value.decode(errors="replace"))
        return comp, mean, proj

    def sync(self):
        lib().dinov2_hip_session_sync(self._h)

    @property
    def stream(self) -> int:
        return int(lib().dinov2_hip_session_stream(self._h) or 0)

    def profile(self, enable: bool):
        lib().dinov2_hip_session_profile(self._h, int(enable))

    def profile_read(self) -> dict:
        n = 32
        names = (C.c_char_p * n)()
        ms = (C.c_float * n)()
        cnt = (C.c_int32 * n)()
        k = lib().dinov2_hip_session_profile_read(self._h, n, names, ms, cnt)
        return {names[i].decode(): (float(ms[i]), int(cnt[i])) for i in range(k)}

    def close(self):
        if getattr(self, "_h", None):
            lib().dinov2_hip_session_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# Mirror of the reference's host API (dinov2.h)
# ------------------------------------------------------------------------------------------------
@dataclass
class dino_params:  # dinov2.h:57-68
    seed: int = 42
    topk: int = 5
    enable_flash_attn: bool = False  # accepted and ignored: attention here is always fused and exact-masked
    camera_id: int = 0
    n_threads: int = 4               # meaningless on the GPU path; kept for signature parity
    classify: bool = False
    model: str = "../ggml-model-f16.gguf"
    fname_inp: str = "../assets/tench.jpg"
    image_out: str = "pca_visual.jpg"
    eps: float = 1e-6


@dataclass
class dino_output:  # dinov2.h:85-88
    preds: list | None = None            # top-k class ids (the reference stores uint32(prob) == 0, dinov2.cpp:975)
    probs: np.ndarray | None = None      # top-k probabilities
    patch_tokens: np.ndarray | None = None  # [P, H] f32, row = y*w0 + x  (cv::Mat of dinov2.cpp:979-992)


@dataclass
class dino_model:  # dinov2.h:49-55
    hparams: HParams | None = None
    handle: Model | None = None
    session: Session | None = None
    id2label: dict = field(default_factory=dict)


def dino_model_load(img_size, fname: str, model: dino_model, params: dino_params, *, device: int = 0,
                    dtype: int = F16) -> bool:
    """dinov2.h:98-99.  `img_size` is unused, as in the reference (dinov2.cpp:309-324).  Returns False and prints to
    stderr when the file cannot be opened (dinov2.cpp:269-272); other errors raise DinoError instead of asserting."""
    import sys
    try:
        model.handle = Model(fname, device=device, dtype=dtype, classify=params.classify)
    except DinoError as e:
        if e.status == 1:
            print(f"dino_model_load: failed to open '{fname}'", file=sys.stderr)
            return False
        raise
    model.hparams = model.handle.hparams
    model.session = Session(model.handle)
    if params.classify and model.hparams.has_classifier:
        model.id2label = {i: model.handle.label(i) for i in range(model.hparams.num_classes)}
    return True


def dino_predict(model: dino_model, img: np.ndarray, params: dino_params, allocr: Session | None = None):
    """dinov2.h:111-112.  `img`: preprocessed CV_32FC3-compatible array [H, W, 3], BGR interleaved (what
    dino_preprocess returns).  `allocr` plays the role of the reusable ggml_gallocr_t.  Prints the top-k lines the
    reference prints (dinov2.cpp:972-974) and returns a dino_output (None on failure, like the empty unique_ptr)."""
    sess = allocr or model.session
    try:
        r = sess.predict(img[None], classify=params.classify, layout=BGR_HWC, topk=params.topk if params.classify else 0,
                         want=("probs",) if params.classify else ("patch_tokens",))
    except DinoError as e:
        import sys
        print(f"dino_predict: {e}", file=sys.stderr)
        return None
    out = dino_output()
    if params.classify:
        ids, pr = r["topk_ids"][0], r["topk_probs"][0]
        for i, p in zip(ids, pr):
            if i >= 0:
                print(f" > {model.id2label.get(int(i), str(int(i)))} : {p:.2f}")
        out.preds, out.probs = [int(i) for i in ids if i >= 0], pr
    else:
        out.patch_tokens = r["patch_tokens"][0]
    return out
