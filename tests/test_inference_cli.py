"""the `inference` caller around the hot path (PCA visualisation + CLI), /root/reference/inference.cpp."""
import os
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
