"""numpy restatement of the reference's host preprocessing (TEST INFRASTRUCTURE ONLY, see oracle/README.md).

Follows /root/reference/dinov2.cpp:106-132 (dino_classify_preprocess) and :135-156 (dino_preprocess):
convertTo(CV_32FC3, 1/255) -> cv::resize(INTER_CUBIC) -> (centre crop 224) -> (c - mean[2-i]) / std[2-i] on B,G,R.
cv::resize(INTER_CUBIC) on CV_32F = separable cubic convolution, A = -0.75, source coordinate (d + 0.5) * src/dst - 0.5,
taps floor-1..floor+2 clamped to the border, no antialiasing (OpenCV documentation; un-verifiable offline, no cv2 here --
SURVEY.md appendix C checked that this formula equals torch bicubic with align_corners=False).
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], np.float64)  # RGB order, dinov2.h:16
STD = np.array([0.229, 0.224, 0.225], np.float64)


def _axis(src, dst):
    d = np.arange(dst, dtype=np.float64)
    f = (d + 0.5) * (src / dst) - 0.5
    s = np.floor(f)
    t = f - s
    A = -0.75
    w = np.stack([((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A,
                  ((A + 2) * t - (A + 3)) * t * t + 1,
                  ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1], axis=1)
    w = np.concatenate([w, 1 - w.sum(1, keepdims=True)], axis=1)
    idx = np.clip(s[:, None].astype(np.int64) + np.arange(-1, 3)[None, :], 0, src - 1)
    return idx, w


def resize_cubic(img, oh, ow):
    """img [h, w, c] float64 -> [oh, ow, c]"""
    h, w, _ = img.shape
    ix, wx = _axis(w, ow)
    iy, wy = _axis(h, oh)
    tmp = (img[:, ix, :] * wx[None, :, :, None]).sum(2)      # horizontal pass  [h, ow, c]
    return (tmp[iy, :, :] * wy[:, :, None, None]).sum(1)     # vertical pass    [oh, ow, c]


def preprocess_size(mode, h, w, patch):
    return (224, 224) if mode == 1 else ((h // patch + 1) * patch, (w // patch + 1) * patch)


def preprocess(mode, bgr_u8, patch=14):
    """bgr_u8 [h, w, 3] uint8 -> f32 [oh, ow, 3] (BGR interleaved, normalised), computed in float64."""
    x = bgr_u8.astype(np.float64) / 255.0
    h, w, _ = x.shape
    if mode == 1:
        y = resize_cubic(x, 256, 256)[16:240, 16:240]
    else:
        oh, ow = preprocess_size(0, h, w, patch)
        y = resize_cubic(x, oh, ow)
    return ((y - MEAN[::-1]) / STD[::-1]).astype(np.float32)
