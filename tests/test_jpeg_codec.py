"""examples/jpeg_codec.hpp -- the JPEG decoder / encoder the example programs use in place of cv::imread / cv::imwrite
(/root/reference/inference.cpp:36, :95) -- against PIL (libjpeg-turbo, the same decoder family cv::imread uses): decoded bytes must be
IDENTICAL, for the reference's own default input (tests/golden/tench.jpg: progressive, 4:4:4) and for generated files of every layout the
decoder claims (baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0, grey, odd sizes, restart intervals).  CPU only."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    exe = str(tmp_path_factory.mktemp("jpeg") / "jpeg_tool")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "cpp", "jpeg_tool.cpp"), "-o", exe], check=True)
    return exe


def _picture(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 120 * np.sin(xx / 9.0 + yy / 23.0), 127 + 120 * np.cos(yy / 7.0), (xx * 5 + yy * 3) % 256], -1)
    base[h // 3: h // 3 + 9, w // 4: w // 4 + 17] = [255, 0, 0]  # a hard edge: chroma upsampling and ringing show here
    return np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)


def _decode(tool, path, tmp):
    out = str(tmp / (os.path.basename(path) + ".ppm"))
    subprocess.run([tool, "decode", path, out], check=True)
    return np.asarray(PIL.open(out))


def test_decodes_the_reference_input_like_libjpeg(tool, tmp_path):
    path = os.path.join(ROOT, "tests", "golden", "tench.jpg")
    got = _decode(tool, path, tmp_path)
    exp = np.asarray(PIL.open(path).convert("RGB"))
    assert got.shape == exp.shape == (408, 612, 3)
    assert np.array_equal(got, exp)


_CASES = [dict(size=(64, 96), subsampling=0, quality=90), dict(size=(57, 83), subsampling=0, quality=75),
          dict(size=(64, 96), subsampling=2, quality=92), dict(size=(57, 83), subsampling=2, quality=60), dict(size=(33, 35), subsampling=2, quality=95),
          dict(size=(71, 130), subsampling=1, quality=85), dict(size=(16, 17), subsampling=1, quality=50),
          dict(size=(120, 200), subsampling=2, quality=88, progressive=True), dict(size=(41, 67), subsampling=0, quality=97, progressive=True),
          dict(size=(90, 123), subsampling=1, quality=80, progressive=True),
          dict(size=(100, 150), subsampling=2, quality=90, restart_marker_blocks=3), dict(size=(100, 150), subsampling=0, quality=90, restart_marker_rows=1),
          dict(size=(77, 91), subsampling=2, quality=85, progressive=True, restart_marker_blocks=5),
          dict(size=(50, 70), grey=True, quality=90), dict(size=(51, 69), grey=True, quality=80, progressive=True),
          dict(size=(8, 8), subsampling=2, quality=90), dict(size=(1, 1), subsampling=0, quality=90), dict(size=(300, 7), subsampling=2, quality=30, optimize=True)]


@pytest.mark.parametrize("case", _CASES, ids=lambda c: "-".join(f"{k}={v}" for k, v in c.items()))
def test_decoder_matches_pil_byte_for_byte(tool, tmp_path, case):
    kw = dict(case)
    h, w = kw.pop("size")
    grey = kw.pop("grey", False)
    img = _picture(h, w, h * 1000 + w)
    pil = PIL.fromarray(img[:, :, 0] if grey else img)
    path = str(tmp_path / "case.jpg")
    pil.save(path, "JPEG", **kw)
    got = _decode(tool, path, tmp_path)
    exp = np.asarray(PIL.open(path).convert("RGB"))
    assert got.shape == exp.shape
    assert np.array_equal(got, exp), (np.abs(got.astype(int) - exp.astype(int)).max(), int((got != exp).sum()))


def test_rejects_what_it_cannot_decode(tool, tmp_path):
    bad = str(tmp_path / "bad.jpg")
    open(bad, "wb").write(b"\xff\xd8\xff\xe0\x00\x10JFIF\x00" + b"\x00" * 40)
    assert subprocess.run([tool, "decode", bad, str(tmp_path / "o.ppm")], capture_output=True).returncode == 1
    data = open(os.path.join(ROOT, "tests", "golden", "tench.jpg"), "rb").read()
    open(bad, "wb").write(data[: len(data) // 3])  # truncated: must not crash (missing scans decode as zero coefficients or fail cleanly)
    assert subprocess.run([tool, "decode", bad, str(tmp_path / "o.ppm")], capture_output=True).returncode in (0, 1)


@pytest.mark.parametrize("size", [(64, 96), (57, 83), (9, 200)])
def test_encoder_writes_a_jpeg_that_libjpeg_decodes_to_the_picture(tool, tmp_path, size):
    img = _picture(size[0], size[1], 5)
    src, jpg = str(tmp_path / "src.ppm"), str(tmp_path / "out.jpg")
    PIL.fromarray(img).save(src)
    subprocess.run([tool, "encode", src, jpg], check=True)
    dec = np.asarray(PIL.open(jpg).convert("RGB")).astype(np.float64)
    assert dec.shape == img.shape
    mse = ((dec - img) ** 2).mean()
    assert 10 * np.log10(255.0 ** 2 / mse) > 34.0  # quality 95, 4:4:4, noisy picture
    assert np.array_equal(_decode(tool, jpg, tmp_path), np.asarray(PIL.open(jpg).convert("RGB")))  # and the decoder reads its own encoder


def test_decoder_survives_corrupt_input_under_sanitizers(tmp_path):
    """Byte flips and truncations of valid files (the reference's progressive tench.jpg, a baseline 4:2:0 file, a file with restart markers),
    2 000 mutations each, decoded in a build with AddressSanitizer + UndefinedBehaviorSanitizer: the decoder may refuse a file, it must not
    read or write out of bounds.  (Round 6 found one that way: an over-subscribed Huffman table indexed past the 9-bit lookup table.)"""
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    exe = str(tmp_path / "jpeg_tool_san")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        os.path.join(ROOT, "tests", "cpp", "jpeg_tool.cpp"), "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    img = _picture(61, 83, 9)
    files = [os.path.join(ROOT, "tests", "golden", "tench.jpg")]
    for i, kw in enumerate((dict(quality=80, subsampling=2), dict(quality=80, subsampling=1, restart_marker_blocks=2), dict(quality=85, subsampling=2, progressive=True))):
        path = str(tmp_path / f"f{i}.jpg")
        PIL.fromarray(img).save(path, "JPEG", **kw)
        files.append(path)
    for seed, path in enumerate(files):
        r = subprocess.run([exe, "fuzz", path, f"{seed + 1} 2000"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (path, r.stderr[-2000:])
        assert "decoded" in r.stdout
