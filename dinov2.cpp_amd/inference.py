"""`inference` CLI counterpart: the caller of the hot path, /root/reference/inference.cpp:24-104.

Same flags (dino_params_parse, /root/reference/dinov2.cpp:865-898; the reference's `-o` bug that overwrites the input path
is NOT reproduced), same stderr/stdout lines (`main: graph computation took N ms` is what scripts/benchmark.sh:73-77 scrapes),
same flow: imread -> dino_model_load -> preprocess -> timed dino_predict -> top-k lines, or 3-component PCA of the patch
tokens -> min-max to 0..255 -> reshape to the patch grid -> nearest-neighbour resize to the preprocessed size -> image file.
Image decode/encode uses PIL instead of OpenCV; preprocessing runs on the device (raw 8-bit input).

    python -m dinov2_cpp_amd.inference -m model.gguf -i image.jpg [-c] [-k 5] [-o pca_visual.jpg]
"""
from __future__ import annotations

import sys
import time

import numpy as np

from . import api


def pca_visual(patch_tokens: np.ndarray, rows: int, cols: int, out_h: int, out_w: int, session=None) -> np.ndarray:
    """cv::PCA(DATA_AS_ROW, 3) + project + normalize(0, 255, NORM_MINMAX, CV_8U) + reshape + resize(INTER_NEAREST)
    (inference.cpp:76-92).  Returns uint8 [out_h, out_w, 3].  Eigenvector signs are a free choice in any PCA; here each
    component is oriented so that its largest-magnitude loading is positive.  With a `session` the means and the covariance
    are computed on the device (dinov2_hip_pca3); without one this is the host tool's numpy path.  With a session,
    `patch_tokens` may also be the hidden size H alone: the PCA then runs on the patch tokens the session's last predict()
    left on the device."""
    if session is not None:
        if isinstance(patch_tokens, (int, np.integer)):
            _, _, proj = session.pca3(None, (rows * cols, int(patch_tokens)))
        else:
            _, _, proj = session.pca3(patch_tokens)
        return _render(proj, rows, cols, out_h, out_w)
    try:  # a 2 170 x 1 024 problem spread over hundreds of BLAS threads is slower than over eight
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=8):
            return _pca_visual(patch_tokens, rows, cols, out_h, out_w)
    except ImportError:
        return _pca_visual(patch_tokens, rows, cols, out_h, out_w)


def _pca_visual(patch_tokens, rows, cols, out_h, out_w):
    x = patch_tokens.astype(np.float32)
    mean = x.mean(0, keepdims=True, dtype=np.float64).astype(np.float32)
    xc = x - mean
    cov = (xc.T @ xc).astype(np.float64) / x.shape[0]  # one sgemm; the full eigendecomposition below it was 10x its cost
    comp = None
    if cov.shape[0] > 64:
        try:  # three leading eigenpairs only (Lanczos): ~10 ms at H = 1024 where numpy's full eigh takes ~400 ms
            from scipy.sparse.linalg import eigsh
            w, v = eigsh(cov, k=3, which="LA", v0=np.ones(cov.shape[0]), tol=1e-7)
            comp = v[:, np.argsort(-w)].T.copy()
        except Exception:
            comp = None
    if comp is None:
        w, v = np.linalg.eigh(cov)
        comp = v[:, ::-1][:, :3].T.copy()  # top-3, rows = components
    for c in comp:
        if c[np.abs(c).argmax()] < 0:
            c *= -1
    proj = xc @ comp.T.astype(np.float32)                        # [P, 3]; same dtype on both sides keeps this a BLAS call
    return _render(proj, rows, cols, out_h, out_w)


def _render(proj, rows, cols, out_h, out_w):
    proj = np.asarray(proj, np.float32)
    lo, hi = proj.min(), proj.max()                                     # all in f32, like the C++ program
    norm = np.zeros_like(proj) if hi == lo else (proj - lo) * (np.float32(255.0) / (hi - lo))
    img = np.rint(norm).clip(0, 255).astype(np.uint8).reshape(rows, cols, 3)
    # cv::resize(INTER_NEAREST): source index = min(floor(dst * ifx), src - 1) with ifx = 1 / (dst_size / src_size)
    yy = np.minimum(np.floor(np.arange(out_h) * (1.0 / (out_h / rows))).astype(np.int64), rows - 1)
    xx = np.minimum(np.floor(np.arange(out_w) * (1.0 / (out_w / cols))).astype(np.int64), cols - 1)
    return img[yy][:, xx]


def _usage(prog, p):
    e = sys.stderr
    print(f"usage: {prog} [options]\n\noptions:", file=e)
    print("  -h, --help              show this help message and exit", file=e)
    print(f"  -m FNAME, --model       model path (default: {p.model})", file=e)
    print(f"  -i FNAME, --inp         input file (default: {p.fname_inp})", file=e)
    print(f"  -o FNAME, --out         output file for backbone PCA features (default: {p.image_out})", file=e)
    print(f"  -k N, --topk            top k classes to print (default: {p.topk})", file=e)
    print(f"  -t N, --threads         number of threads to use during computation (default: {p.n_threads})", file=e)
    print(f"  -c, --classify          whether to classify the image or get backbone PCA features (default: {int(p.classify)})", file=e)
    print(f"  -fa, --flash_attn          whether to enable flash_attn, less accurate (default: {int(p.enable_flash_attn)})", file=e)
    print("", file=e)


def dino_params_parse(argv, p: api.dino_params) -> bool:
    i = 1
    while i < len(argv):
        a = argv[i]
        if a in ("-s", "--seed"):
            i += 1; p.seed = int(argv[i])
        elif a in ("-m", "--model"):
            i += 1; p.model = argv[i]
        elif a in ("-i", "--inp"):
            i += 1; p.fname_inp = argv[i]
        elif a in ("-o", "--out"):
            i += 1; p.image_out = argv[i]
        elif a in ("-t", "--threads"):
            i += 1; p.n_threads = int(argv[i])
        elif a in ("-k", "--topk"):
            i += 1; p.topk = int(argv[i])
        elif a in ("-cid", "--camera_id"):
            i += 1; p.camera_id = int(argv[i])
        elif a in ("-fa", "--flash_attn"):
            p.enable_flash_attn = True
        elif a in ("-c", "--classify"):
            p.classify = True
        elif a in ("-h", "--help"):
            _usage(argv[0], p); raise SystemExit(0)
        else:
            print(f"error: unknown argument: {a}", file=sys.stderr)
            _usage(argv[0], p); raise SystemExit(0)
        i += 1
    return True


def main(argv=None) -> int:
    from PIL import Image
    argv = sys.argv if argv is None else argv
    p = api.dino_params()
    dino_params_parse(argv, p)
    print(f"main: seed = {p.seed}", file=sys.stderr)
    try:
        rgb = np.asarray(Image.open(p.fname_inp).convert("RGB"))
    except Exception:
        print(f"main: failed to load image from '{p.fname_inp}'", file=sys.stderr)
        return 1
    img = np.ascontiguousarray(rgb[:, :, ::-1])  # cv::imread gives BGR
    print(f"main: loaded image '{p.fname_inp}' ({img.shape[0]} x {img.shape[1]})", file=sys.stderr)
    model = api.dino_model()
    if not api.dino_model_load(img.shape[:2], p.model, model, p):
        print(f"main: failed to load model from '{p.model}'", file=sys.stderr)
        return 1
    hp = model.hparams
    for k in ("hidden_size", "num_hidden_layers", "num_register_tokens", "num_attention_heads", "patch_size", "img_size", "ftype"):
        print(f"dino_model_load: {k:<22} = {getattr(hp, k)}")
    oh, ow = api.preprocess_size(1 if p.classify else 0, img.shape[0], img.shape[1], hp.patch_size)
    print(f"main: preprocessed image ({oh} x {ow})", file=sys.stderr)
    sess = model.session
    sess.sync()
    t0 = time.perf_counter()
    r = sess.predict(img[None], classify=p.classify, layout=api.U8_BGR_HWC, topk=p.topk if p.classify else 0,
                     want=("probs",) if p.classify else ())  # features: the patch tokens stay on the device for the PCA
    sess.sync()
    print(f"main: graph computation took {int(round((time.perf_counter() - t0) * 1e3))} ms", file=sys.stderr)
    if p.classify:
        print("", file=sys.stderr)
        for i, pr in zip(r["topk_ids"][0], r["topk_probs"][0]):
            if i >= 0:
                print(f" > {model.id2label.get(int(i), str(int(i)))} : {pr:.2f}")
        return 0
    rows, cols = oh // hp.patch_size, ow // hp.patch_size
    vis = pca_visual(hp.hidden_size, rows, cols, oh, ow, session=sess)
    try:
        Image.fromarray(np.ascontiguousarray(vis[:, :, ::-1])).save(p.image_out)  # stored BGR like the cv::Mat -> RGB file
        print(f"main: Saved image to: {p.image_out}", file=sys.stderr)
    except Exception:
        print(f"main: failed to save image to '{p.image_out}'", file=sys.stderr)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
