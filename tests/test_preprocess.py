"""dino_preprocess / dino_classify_preprocess without OpenCV.
CPU: the C-ABI host implementation vs the numpy oracle (float64) vs torch bicubic (independent implementation).
GPU (tests/test_gpu_configs.py): the device preprocessing kernel vs the same numpy oracle, alone and followed by the forward."""
import os

import numpy as np
import pytest

from oracle import preprocess_np as PP


def _images():
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:408, 0:612]
    smooth = np.stack([(xx * 255 // 611), (yy * 255 // 407), ((xx + yy) % 256)], -1).astype(np.uint8)  # tench.jpg-sized
    return {"noise_300x200": rng.integers(0, 256, (300, 200, 3), dtype=np.uint8), "smooth_408x612": smooth,
            "tiny_20x33": rng.integers(0, 256, (20, 33, 3), dtype=np.uint8),
            "multiple_of_14": rng.integers(0, 256, (518, 518, 3), dtype=np.uint8)}


@pytest.mark.parametrize("name", list(_images()))
def test_host_preprocess_matches_oracle(api, name):
    img = _images()[name]
    for mode, fn in ((0, api.dino_preprocess), (1, api.dino_classify_preprocess)):
        got = fn(img)
        exp = PP.preprocess(mode, img)
        assert got.shape == exp.shape == (*PP.preprocess_size(mode, img.shape[0], img.shape[1], 14), 3)
        # f32 source coordinates and weights (as in cv::resize) vs the float64 oracle: <= ~1e-4 on white-noise images
        assert np.abs(got - exp).max() < 3e-4, (name, mode)


def test_resize_quirks(api):
    """The feature resize ALWAYS grows by one patch, even from a multiple of 14 (518 -> 532, dinov2.cpp:140-141; SURVEY
    fact 5); classify always yields 224x224 whatever the aspect ratio (dinov2.cpp:111-119)."""
    assert api.preprocess_size(0, 518, 518) == (532, 532)
    assert api.preprocess_size(0, 408, 612) == (420, 616)  # tench.jpg -> the 616x420 pca_visual.jpg of the reference
    assert api.preprocess_size(1, 408, 612) == (224, 224)
    assert api.dino_classify_preprocess(_images()["noise_300x200"]).shape == (224, 224, 3)


def test_bgr_mean_std_indexing(api):
    """Channel i of the BGR image is normalised with mean[2 - i], std[2 - i] (dinov2.cpp:124-127)."""
    img = np.zeros((28, 28, 3), np.uint8)
    img[..., 0] = 255  # pure blue
    out = api.dino_preprocess(img)
    c = out[20, 20]
    np.testing.assert_allclose(c, [(1 - 0.406) / 0.225, (0 - 0.456) / 0.224, (0 - 0.485) / 0.229], rtol=1e-5)


def test_bicubic_equals_torch(api):
    """cv::INTER_CUBIC restatement == torch bicubic (align_corners=False, no antialias), up- and down-scaling."""
    torch = pytest.importorskip("torch")
    img = _images()["noise_300x200"]
    got = api.dino_preprocess(img)
    x = torch.from_numpy(img.astype(np.float32) / 255.0).permute(2, 0, 1)[None]
    y = torch.nn.functional.interpolate(x, size=got.shape[:2], mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
    exp = (y - np.array([0.406, 0.456, 0.485], np.float32)) / np.array([0.225, 0.224, 0.229], np.float32)
    assert np.abs(got - exp).max() < 3e-4
    got = api.dino_classify_preprocess(img)  # 300x200 -> 256x256 (down in y, up in x) -> crop
    y = torch.nn.functional.interpolate(x, size=(256, 256), mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
    exp = ((y - np.array([0.406, 0.456, 0.485], np.float32)) / np.array([0.225, 0.224, 0.229], np.float32))[16:240, 16:240]
    assert np.abs(got - exp).max() < 3e-4


# The -m gpu checks of the DEVICE preprocessing kernel (kernel output and kernel -> forward, both against the oracle) live in
# tests/test_gpu_configs.py: test_preprocess_kernel_vs_oracle, test_device_preprocess_then_forward_vs_oracle.


def test_host_preprocess_under_sanitizers(tmp_path):
    """tests/cpp/preprocess_san.cpp: csrc/preprocess.cpp built with AddressSanitizer + UndefinedBehaviorSanitizer, both modes, 52 image sizes (1 x 1,
    smaller than a patch, odd, 3 x 2000, random) x two patch sizes, the output buffer sized exactly as dinov2_hip_preprocess_size promises."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        pytest.skip("ROCm clang not available")
    exe = str(tmp_path / "pp_san")
    r = subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        os.path.join(root, "tests", "cpp", "preprocess_san.cpp"), os.path.join(root, "dinov2.cpp_amd", "csrc", "preprocess.cpp"), "-o", exe],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("preprocessed 2"), r.stdout
