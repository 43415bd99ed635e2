"""Statistics of a checkpoint's forward that decide whether f16 operands are good enough (test helper; uses the CPU oracle).

From the oracle's per-layer hidden states (f32, the ggml contract) this recomputes, in float64 numpy, what the attention of every
layer sees -- LayerNorm 1, q and k of every head, the pre-softmax scores `q . k / 8` (/root/reference/dinov2.cpp:479-536) -- and
reports per layer:
  max_abs_score[head]   largest |pre-softmax score| of the head
  sink_mass[head]       mean over queries of the softmax mass on the register tokens
  outlier_ratio         largest per-channel median |x| / median over channels of that median (residual stream entering the layer)
  mean_over_std         largest |row mean| / row std of the residual stream (what a LayerNorm folded into the next GEMM is sensitive to)
`python -m tests.trained_stats [model] [size]` prints them for a synthetic trained-like checkpoint (calibration of synth.py's constants).
"""
import sys

import numpy as np

from oracle.oracle import OracleModel


def layer_stats(path, img_chw, layers=None):
    ora = OracleModel(path)
    t = ora.gguf.tensors
    H, nh, R = ora.hidden, ora.heads, ora.registers
    hid = ora.forward(img_chw, classify=False, hidden=True)["hidden"].astype(np.float64)  # [L + 1, T, H]: input of layer l at index l
    out = []
    for l in (range(ora.layers) if layers is None else layers):
        x = hid[l]
        mu = x.mean(-1, keepdims=True)
        sd = x.std(-1, keepdims=True)
        g = t[f"encoder.layer.{l}.norm1.weight"].to_f32().reshape(-1).astype(np.float64)
        b = t[f"encoder.layer.{l}.norm1.bias"].to_f32().reshape(-1).astype(np.float64)
        ln = (x - mu) / np.sqrt(sd * sd + 1e-6) * g + b
        W = t[f"encoder.layer.{l}.attention.attention.qkv.weight"].to_f32().reshape(3 * H, H).astype(np.float64)
        bq = t[f"encoder.layer.{l}.attention.attention.qkv.bias"].to_f32().reshape(-1).astype(np.float64)
        qk = ln @ W[: 2 * H].T + bq[: 2 * H]
        mx, sink = [], []
        for h in range(nh):
            q = qk[:, h * 64:(h + 1) * 64]
            k = qk[:, H + h * 64:H + (h + 1) * 64]
            s = q @ k.T / 8.0
            mx.append(float(np.abs(s).max()))
            p = np.exp(s - s.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            sink.append(float(p[:, 1:1 + R].sum(-1).mean()) if R else 0.0)
        med = np.median(np.abs(x), axis=0)
        out.append(dict(layer=l, max_abs_score=mx, sink_mass=sink, outlier_ratio=float(med.max() / np.median(med)),
                        max_abs_x=float(np.abs(x).max()), mean_over_std=float((np.abs(mu) / sd).max())))
    return out


if __name__ == "__main__":
    import os
    import tempfile

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import dinov2_cpp_amd as pkg

    model = sys.argv[1] if len(sys.argv) > 1 else "small"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 224
    path = os.path.join(tempfile.gettempdir(), f"tl_{model}.gguf")
    pkg.synth.write_synthetic_gguf(path, model, registers=4, num_classes=1000, seed=42, head_std=0.12, trained_like=True)
    img = pkg.synth.synthetic_images(1, size, size, seed=42)[0]
    for st in layer_stats(path, img):
        ms = np.array(st["max_abs_score"])
        sk = np.array(st["sink_mass"])
        print(f"layer {st['layer']:2d}: max|score| per head min {ms.min():6.1f} median {np.median(ms):6.1f} max {ms.max():6.1f}  heads>=30: {(ms >= 30).sum():2d}  "
              f"sink mass max {sk.max():.2f} median {np.median(sk):.2f}  outlier ratio {st['outlier_ratio']:6.1f}  max|x| {st['max_abs_x']:7.1f}  |mean|/std {st['mean_over_std']:.3f}")
    lg = OracleModel(path).forward(img, classify=True)["logits"]
    print("max|logit|", float(np.abs(lg).max()))
