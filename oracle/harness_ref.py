"""CPU oracle of the evaluation harness (test.py:17-40, 110-124) -- TEST INFRASTRUCTURE, numpy only.
Pinned by tests/golden/harness.npz: outputs of the reference's own ``correct_bayer_channels`` /
``auto_correct_rb`` (their source text executed from test.py by oracle/make_golden.py) and of the
numpy expression test.py:118 uses for the uint8 conversion.  PSNR follows scikit-image's published
definition for uint8 inputs (data_range = 255); scikit-image itself is absent offline."""
import numpy as np


def to_uint8_hwc(pred_chw: np.ndarray) -> np.ndarray:
    """test.py:117-118 for one image: clamp, transpose to HWC, * 255, truncate."""
    return (np.clip(pred_chw, 0.0, 1.0).astype(np.float32).transpose(1, 2, 0) * 255).astype(np.uint8)


def correct_bayer_channels(rgb, pattern="RGGB"):   # test.py:17-29
    pattern = pattern.upper()
    if pattern == "BGGR":
        return rgb[..., [2, 1, 0]]
    if pattern == "GBRG":
        return rgb[..., [1, 0, 2]]
    if pattern == "GRBG":
        return rgb[..., [0, 2, 1]]
    return rgb


def auto_correct_rb(rgb):   # test.py:31-40
    return rgb[..., [2, 1, 0]] if rgb[..., 0].mean() < rgb[..., 2].mean() else rgb


def psnr_u8(a: np.ndarray, b: np.ndarray) -> float:
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float(10.0 * np.log10(255.0 ** 2 / mse)) if mse > 0 else float("inf")


def pack_raw(raw_u16: np.ndarray, black_levels, white_level: int, ratio: float) -> np.ndarray:
    """correctdataloader.py:58-72 + :86 + :103 for one uint16 frame [2h, 2w] -> float32 [4, h, w] in the loader's
    channel order.  The arithmetic is done in float64 and rounded once: under NumPy >= 2 the reference's
    ``float32_array - np.int64_scalar`` promotes to float64 (pinned by tests/golden/harness.npz)."""
    black = float(np.min(np.asarray(black_levels)))
    im = (raw_u16.astype(np.float64) - black) / (float(white_level) - black)
    im = np.clip(im, 0.0, 1.0)
    out = np.stack((im[0::2, 0::2], im[0::2, 1::2], im[1::2, 1::2], im[1::2, 0::2]), axis=0)
    return np.minimum(out * float(ratio), 1.0).astype(np.float32)
