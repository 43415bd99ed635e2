"""the `inference` caller around the hot path (PCA visualisation + CLI), /root/reference/inference.cpp."""
import os
import sys
from importlib import import_module

import numpy as np
import pytest

from __graft_entry__ import PKG_NAME


def test_pca_visual_matches_numpy_svd():
    inf = import_module(PKG_NAME + ".inference")
    rng = np.random.default_rng(0)
    base = rng.standard_normal((12, 3)) @ rng.standard_normal((3, 32)) * 3 + rng.standard_normal((12, 32)) * 0.01
    vis = inf.pca_visual(base.astype(np.float32), 3, 4, 42, 56)
    assert vis.shape == (42, 56, 3) and vis.dtype == np.uint8 and vis.min() == 0 and vis.max() == 255
    xc = base - base.mean(0)
    u, s, vt = np.linalg.svd(xc, full_matrices=False)
    proj = xc @ vt[:3].T
    got = vis[::14, ::14].reshape(12, 3).astype(np.float64)  # one sample per patch (nearest resize by 14)
    for c in range(3):  # same components up to sign and the global min-max affine map
        assert abs(np.corrcoef(got[:, c], proj[:, c])[0, 1]) > 0.999


def test_pca_visual_lanczos_path_matches_svd():
    """H > 64 takes the three-eigenpair Lanczos solver: same components (up to sign) as a full SVD."""
    inf = import_module(PKG_NAME + ".inference")
    rng = np.random.default_rng(3)
    P, H = 20 * 31, 256
    base = rng.standard_normal((P, 3)) * np.array([9.0, 5.0, 2.5]) @ rng.standard_normal((3, H)) + rng.standard_normal((P, H)) * 0.05
    vis = inf.pca_visual(base.astype(np.float32), 20, 31, 20 * 14, 31 * 14)
    xc = base - base.mean(0)
    proj = xc @ np.linalg.svd(xc, full_matrices=False)[2][:3].T
    got = vis[::14, ::14].reshape(P, 3).astype(np.float64)
    for c in range(3):
        assert abs(np.corrcoef(got[:, c], proj[:, c])[0, 1]) > 0.999


def test_cli_flag_parsing(api):
    inf = import_module(PKG_NAME + ".inference")
    p = api.dino_params()
    inf.dino_params_parse(["inference", "-m", "a.gguf", "-i", "b.jpg", "-o", "c.png", "-k", "3", "-c", "-fa", "-t", "8"], p)
    assert (p.model, p.fname_inp, p.image_out, p.topk, p.classify, p.enable_flash_attn, p.n_threads) == \
           ("a.gguf", "b.jpg", "c.png", 3, True, True, 8)  # -o sets the OUTPUT (the reference overwrites fname_inp, dinov2.cpp:875)
    with pytest.raises(SystemExit):
        inf.dino_params_parse(["inference", "--bogus"], api.dino_params())


@pytest.mark.gpu
def test_cli_end_to_end(golden_dir, tmp_path, capsys):
    from PIL import Image
    inf = import_module(PKG_NAME + ".inference")
    src, out = str(tmp_path / "in.png"), str(tmp_path / "pca.png")
    rng = np.random.default_rng(1)
    Image.fromarray(rng.integers(0, 256, (60, 75, 3), dtype=np.uint8)).save(src)
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    assert inf.main(["inference", "-m", gguf, "-i", src, "-c", "-k", "3"]) == 0
    cap = capsys.readouterr()
    assert cap.out.count(" > label_") == 3 and "graph computation took" in cap.err and "preprocessed image (224 x 224)" in cap.err
    assert inf.main(["inference", "-m", gguf, "-i", src, "-o", out]) == 0
    cap = capsys.readouterr()
    assert "Saved image to" in cap.err and "preprocessed image (70 x 84)" in cap.err
    assert np.asarray(Image.open(out)).shape == (70, 84, 3)
    assert inf.main(["inference", "-m", gguf, "-i", "/nonexistent.jpg"]) == 1


def _build_cpp_inference(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "inference")
    libdir = os.path.join(root, "dinov2.cpp_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "examples"),
                           os.path.join(root, "examples", "inference.cpp"), "-o", exe, os.path.join(libdir, "libdinov2_hip.so"),
                           f"-Wl,-rpath,{libdir}"])
    return exe


def test_cpp_inference_builds_and_reports_errors(tmp_path):
    """examples/inference.cpp (the reference's `inference` flow on the C++ shim, JPEG / PPM through examples/jpeg_codec.hpp instead of OpenCV codecs) builds with plain
    g++; usage and failure paths behave like the reference (unknown flag -> usage + exit 0; unreadable image -> message + 1)."""
    import subprocess
    exe = _build_cpp_inference(tmp_path)
    r = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 0 and "unknown argument" in r.stderr and "usage:" in r.stderr
    r = subprocess.run([exe, "-i", "/nonexistent.ppm"], capture_output=True, text=True)
    assert r.returncode == 1 and "failed to load image" in r.stderr


@pytest.mark.gpu
def test_cpp_inference_end_to_end(golden_dir, tmp_path):
    """Same image through the C++ program and the Python CLI: identical top-k lines; the PCA map has the preprocessed size."""
    import subprocess
    exe = _build_cpp_inference(tmp_path)
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (60, 75, 3), dtype=np.uint8)
    src, out = str(tmp_path / "in.ppm"), str(tmp_path / "pca.ppm")
    with open(src, "wb") as f:
        f.write(b"P6\n# a comment\n75 60\n255\n" + rgb.tobytes())
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    r = subprocess.run([exe, "-m", gguf, "-i", src, "-c", "-k", "3"], capture_output=True, text=True)
    assert r.returncode == 0 and "graph computation took" in r.stderr and "preprocessed image (224 x 224)" in r.stderr, r.stderr
    cpp_lines = [ln for ln in r.stdout.splitlines() if ln.startswith(" > ")]
    assert len(cpp_lines) == 3
    from PIL import Image
    png = str(tmp_path / "in.png")
    Image.fromarray(rgb).save(png)
    py = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, '.'); from __graft_entry__ import load_package, PKG_NAME; "
                         "load_package(); from importlib import import_module; "
                         f"sys.exit(import_module(PKG_NAME + '.inference').main(['inference', '-m', r'{gguf}', '-i', r'{png}', '-c', '-k', '3']))"],
                        capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert [ln for ln in py.stdout.splitlines() if ln.startswith(" > ")] == cpp_lines, py.stdout + py.stderr
    r = subprocess.run([exe, "-m", gguf, "-i", src, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0 and "Saved image to" in r.stderr and "preprocessed image (70 x 84)" in r.stderr, r.stderr
    hdr = open(out, "rb").read(15)
    assert hdr.startswith(b"P6\n84 70\n255\n")
    # the Python CLI renders the same map (both go through dinov2_hip_pca3).  The two programs hand the image over differently
    # (u8 BGR to the device-side preprocessing vs the shim's host preprocessing to f32), so allow one grey level on a few cells
    pyout = str(tmp_path / "pca_py.png")
    py = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, '.'); from __graft_entry__ import load_package, PKG_NAME; "
                         "load_package(); from importlib import import_module; "
                         f"sys.exit(import_module(PKG_NAME + '.inference').main(['inference', '-m', r'{gguf}', '-i', r'{png}', '-o', r'{pyout}']))"],
                        capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert py.returncode == 0, py.stderr
    cpp_img = np.frombuffer(open(out, "rb").read()[len(b"P6\n84 70\n255\n"):], np.uint8).reshape(70, 84, 3)
    py_img = np.asarray(Image.open(pyout).convert("RGB"))
    d = np.abs(py_img.astype(np.int32) - cpp_img.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() <= 0.05, (int(d.max()), int((d > 0).sum()), np.argwhere(d > 0)[:8].tolist())


@pytest.mark.gpu
def test_cpp_realtime_loop(golden_dir, tmp_path):
    """examples/realtime.cpp: the reference's realtime loop (realtime.cpp:56-108) without camera and window -- 854 x 480 frames ->
    raw-u8 predict (tokens stay on the device) -> dinov2_hip_pca3(tokens = NULL) -> min-max -> 35 x 62 map -> nearest resize ->
    hconcat.  Builds with plain g++ against the shim; the right half of the written frame is the PCA map the Python helper
    renders from the same frame's tokens (one grey level of slack: f16-rounded covariance), the left half is the input frame."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "realtime")
    libdir = os.path.join(root, "dinov2.cpp_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "examples"), os.path.join(root, "examples", "realtime.cpp"),
                           "-o", exe, "-L" + libdir, "-ldinov2_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    gguf = os.path.join(golden_dir, "tiny_gelu_reg4.gguf")
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:480, 0:854]
    rgb = np.stack([(xx * 255 // 854), (yy * 255 // 480), ((xx // 61 + yy // 48) % 2) * 200], -1).astype(np.uint8)
    rgb = np.clip(rgb.astype(np.int32) + rng.integers(-8, 9, rgb.shape), 0, 255).astype(np.uint8)
    src, out = str(tmp_path / "frame.ppm"), str(tmp_path / "combined.ppm")
    with open(src, "wb") as f:
        f.write(b"P6\n854 480\n255\n" + rgb.tobytes())
    r = subprocess.run([exe, "-m", gguf, "-n", "4", "-i", src, "-o", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "REALTIME_OK" in r.stdout and r.stderr.count("graph computation took") == 4, r.stdout + r.stderr
    assert "2170 patches" in r.stdout
    raw = open(out, "rb").read()
    hdr = b"P6\n1708 480\n255\n"
    assert raw.startswith(hdr)
    comb = np.frombuffer(raw[len(hdr):], np.uint8).reshape(480, 1708, 3)
    assert np.array_equal(comb[:, :854], rgb)
    # the same frame through the Python API: features of the raw frame, PCA map at the frame size
    from __graft_entry__ import load_package, PKG_NAME
    load_package()
    from importlib import import_module
    api = import_module(PKG_NAME + ".api")
    inf = import_module(PKG_NAME + ".inference")
    sess = api.Session(api.Model(gguf, classify=False))
    bgr = np.ascontiguousarray(rgb[:, :, ::-1])
    tok = sess.predict(bgr[None], classify=False, layout=api.U8_BGR_HWC, want=("patch_tokens",))["patch_tokens"][0]
    assert tok.shape[0] == 35 * 62
    ref = inf.pca_visual(tok, 35, 62, 480, 854, session=sess)  # BGR map, like the C++ program's before its PPM swap
    got_bgr = comb[:, 854:, ::-1]
    d = np.abs(ref.astype(np.int32) - got_bgr.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() <= 0.05, (int(d.max()), float((d > 0).mean()))
    # synthetic frames (no -i): the loop runs and reports its rates
    r = subprocess.run([exe, "-m", gguf, "-n", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "frames/s" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_pca_visual_device_matches_host(golden_dir):
    """pca_visual with a session (device covariance) against the numpy path on structured tokens: same picture up to one grey
    level (the covariance is accumulated from f16-rounded centred tokens)."""
    from __graft_entry__ import load_package, PKG_NAME
    load_package()
    from importlib import import_module
    api, inf = import_module(PKG_NAME + ".api"), import_module(PKG_NAME + ".inference")
    sess = api.Session(api.Model(os.path.join(golden_dir, "tiny_gelu_reg4.gguf"), classify=False))
    rng = np.random.default_rng(5)
    rows, cols, H = 16, 20, 384
    yy, xx = np.mgrid[0:rows, 0:cols]
    lat = np.stack([np.sin(yy / 3.0), np.cos(xx / 4.0), (yy + xx) / 20.0], -1).reshape(-1, 3) * np.array([8.0, 5.0, 3.0])
    tok = (lat @ np.linalg.qr(rng.standard_normal((H, 3)))[0].T + 0.05 * rng.standard_normal((rows * cols, H))).astype(np.float32)
    a = inf.pca_visual(tok, rows, cols, rows * 14, cols * 14, session=sess).astype(np.int32)
    b = inf.pca_visual(tok, rows, cols, rows * 14, cols * 14).astype(np.int32)
    assert a.shape == (rows * 14, cols * 14, 3) and np.abs(a - b).max() <= 1


@pytest.mark.gpu
def test_reference_benchmark_script_commands_run_unchanged(pkg, golden_dir, tmp_path):
    """/root/reference/scripts/benchmark.sh:55-100, the commands exactly as the script issues them, in the script's directory layout:
    `cd build/`, `./bin/quantize ../ggml-model.gguf ../ggml-model-quant.gguf <id>`, `./bin/inference -c -m <model> -i ../assets/tench.jpg -t N`
    with stderr through the script's own sed expression -- against `make -C dinov2.cpp_amd examples` binaries, a synthetic ViT-S/14 GGUF in
    place of the converted checkpoint (no network) and tests/golden/tench.jpg = the reference's assets/tench.jpg (progressive JPEG, decoded by
    examples/jpeg_codec.hpp)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "dinov2.cpp_amd"), "examples"], stdout=subprocess.DEVNULL)
    work = tmp_path / "repo"
    (work / "build" / "bin").mkdir(parents=True)
    (work / "assets").mkdir()
    for exe in ("inference", "quantize"):
        shutil.copy(os.path.join(root, "build", "bin", exe), work / "build" / "bin" / exe)
    shutil.copy(os.path.join(golden_dir, "tench.jpg"), work / "assets" / "tench.jpg")
    pkg.synth.write_synthetic_gguf(str(work / "ggml-model.gguf"), "small", registers=0, num_classes=1000, seed=5)
    sed = r"s/.*main: graph computation took ([0-9]+) ms.*/\1/p"
    cwd = str(work / "build")

    def graph_ms(model):
        out = subprocess.run(f"./bin/inference -c -m {model} -i ../assets/tench.jpg -t 12 2>&1 | sed -En '{sed}'", shell=True, cwd=cwd,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.strip().isdigit(), (out.stdout, out.stderr)
        return int(out.stdout)

    assert 0 <= graph_ms("../ggml-model.gguf") < 5000
    full = subprocess.run("./bin/inference -c -m ../ggml-model.gguf -i ../assets/tench.jpg -t 12", shell=True, cwd=cwd, capture_output=True, text=True, timeout=300)
    assert "loaded image '../assets/tench.jpg' (408 x 612)" in full.stderr and "preprocessed image (224 x 224)" in full.stderr, full.stderr
    top = [ln for ln in full.stdout.splitlines() if ln.startswith(" > ")]
    assert len(top) == 5
    for q in (2, 8):  # q4_0, q8_0: benchmark.sh:60
        r = subprocess.run(f"./bin/quantize ../ggml-model.gguf ../ggml-model-quant.gguf {q}", shell=True, cwd=cwd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert 0 <= graph_ms("../ggml-model-quant.gguf") < 5000
    # the feature path writes the PCA picture as a JPEG (the reference's default output name) that a stock decoder reads
    r = subprocess.run("./bin/inference -m ../ggml-model.gguf -i ../assets/tench.jpg", shell=True, cwd=cwd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Saved image to: pca_visual.jpg" in r.stderr, r.stderr
    from PIL import Image
    assert np.asarray(Image.open(os.path.join(cwd, "pca_visual.jpg"))).shape == (420, 616, 3)
