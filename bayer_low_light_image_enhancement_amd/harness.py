"""Evaluation-harness mirror of the reference's test loop (test.py:17-40 and 106-143).

``test.py`` post-processes every prediction on the host: ``clamp(0,1) * 255`` truncated to uint8,
Bayer-order channel fix-ups, PSNR / SSIM from scikit-image, JPEG + CSV.  Here the conversion and
the PSNR reduction run on the device with exact integer arithmetic (``csrc/rf_harness.hip``); the
channel fix-ups keep the reference's function names and semantics.  SSIM is restated on the host
from scikit-image's definition (7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance,
channel-wise mean) -- scikit-image is not installable here, so SSIM parity is unpinned.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np
import torch

from . import _lib


def correct_bayer_channels(rgb, pattern: str = "RGGB"):
    """Channel order fix-up per Bayer pattern (test.py:17-29); works on numpy arrays and tensors (HWC)."""
    pattern = pattern.upper()
    if pattern == "BGGR":
        rgb = rgb[..., [2, 1, 0]]
    elif pattern == "GBRG":
        rgb = rgb[..., [1, 0, 2]]
    elif pattern == "GRBG":
        rgb = rgb[..., [0, 2, 1]]
    return rgb


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def to_uint8_hwc(pred: torch.Tensor) -> torch.Tensor:
    """``(clamp(pred,0,1)[b].cpu().numpy().transpose(1,2,0) * 255).astype(np.uint8)`` for every image
    (test.py:117-118): float32 ``[B,C,H,W]`` on the device -> uint8 ``[B,H,W,C]`` on the device."""
    if pred.device.type != "cuda" or pred.dtype != torch.float32 or pred.dim() != 4:
        raise RuntimeError("to_uint8_hwc expects a float32 [B,C,H,W] ROCm tensor")
    pred = pred.contiguous()
    b, c, h, w = pred.shape
    out = torch.empty((b, h, w, c), dtype=torch.uint8, device=pred.device)
    with torch.cuda.device(pred.device):
        _lib.check(_lib.load().rf_to_uint8_hwc(C.c_void_p(pred.data_ptr()), C.c_void_p(out.data_ptr()), b, c, h, w, _stream(pred)),
                   "rf_to_uint8_hwc")
    return out


def channel_means_u8(img: torch.Tensor) -> torch.Tensor:
    """Exact per-channel means of uint8 ``[B,H,W,C]`` images (float64 ``[B,C]`` on the host)."""
    img = img.contiguous()
    b, h, w, c = img.shape
    sums = torch.empty((b, c), dtype=torch.int64, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(_lib.load().rf_u8_channel_sums(C.c_void_p(img.data_ptr()), C.c_void_p(sums.data_ptr()), b, c, h * w, _stream(img)),
                   "rf_u8_channel_sums")
    return sums.cpu().double() / float(h * w)


def auto_correct_rb(rgb: torch.Tensor) -> torch.Tensor:
    """Swap R and B when the red channel is darker than the blue one (test.py:31-40), per image."""
    means = channel_means_u8(rgb)
    out = rgb.clone()
    for i in range(rgb.shape[0]):
        if means[i, 0] < means[i, 2]:
            out[i] = rgb[i][..., [2, 1, 0]]
    return out


def psnr_u8(a: torch.Tensor, b: torch.Tensor) -> np.ndarray:
    """``skimage.metrics.peak_signal_noise_ratio`` on uint8 images (data_range 255, test.py:123):
    ``10 log10(255^2 / mean((a-b)^2))`` per image, the squared error summed exactly in uint64."""
    if a.shape != b.shape or a.dtype != torch.uint8 or b.dtype != torch.uint8:
        raise RuntimeError("psnr_u8 expects two uint8 tensors of the same shape")
    a, b = a.contiguous(), b.contiguous()
    n = a[0].numel()
    sse = torch.empty((a.shape[0],), dtype=torch.int64, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().rf_u8_sse(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(sse.data_ptr()),
                                         a.shape[0], n, _stream(a)), "rf_u8_sse")
    mse = sse.cpu().numpy().astype(np.float64) / float(n)
    with np.errstate(divide="ignore"):
        return 10.0 * np.log10(255.0 ** 2 / mse)


def ssim_u8(a: np.ndarray, b: np.ndarray) -> float:
    """``structural_similarity(a, b, channel_axis=-1)`` restated from scikit-image's definition for one
    uint8 HWC image pair (host, float64).  PARITY UNPINNED: scikit-image is absent offline."""
    from scipy.ndimage import uniform_filter

    win, k1, k2, rng = 7, 0.01, 0.03, 255.0
    npx = win * win
    cov_norm = npx / (npx - 1.0)
    c1, c2 = (k1 * rng) ** 2, (k2 * rng) ** 2
    pad = (win - 1) // 2
    vals = []
    for ch in range(a.shape[-1]):
        x, y = a[..., ch].astype(np.float64), b[..., ch].astype(np.float64)
        ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
        uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
        vals.append(s[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))


def evaluate(pred: torch.Tensor, gt: torch.Tensor, bayer_pattern: str = "RGGB", with_ssim: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """The body of test.py's loop (test.py:110-124) for a batch: uint8 conversion, Bayer fix-ups on both
    images, PSNR (and optionally SSIM).  ``pred``/``gt`` are float32 ``[B,3,H,W]`` device tensors."""
    p8 = auto_correct_rb(correct_bayer_channels(to_uint8_hwc(torch.clamp(pred, 0, 1)), bayer_pattern).contiguous())
    g8 = auto_correct_rb(correct_bayer_channels(to_uint8_hwc(gt), bayer_pattern).contiguous())
    psnr = psnr_u8(p8, g8)
    ssim = np.array([ssim_u8(p8[i].cpu().numpy(), g8[i].cpu().numpy()) for i in range(p8.shape[0])]) if with_ssim else np.array([])
    return psnr, ssim


_SID_MODES = {"loader": 0, "rgbg": 0, "rggb": 1, "unshuffle": 1, "mosaic": 2}


def pack_raw(raw: torch.Tensor, black_level: int, white_level: int, ratio: float = 1.0, layout: str = "loader") -> torch.Tensor:
    """SID front-end on the device (correctdataloader.py:58-72 ``pack_raw``, :86 ``* ratio``, :103 ``min(., 1)``).

    ``raw``: uint16 (or int16-viewed) Bayer frames ``[B, 2h, 2w]`` on the GPU; ``black_level`` =
    ``min(raw.black_level_per_channel)``.  ``layout``: ``"loader"`` = the reference loader's channel order
    (sites (0,0) (0,1) (1,1) (1,0)), ``"rggb"`` = ``pixel_unshuffle`` order, ``"mosaic"`` = normalised
    ``[B,1,2h,2w]`` mosaic (the input of ``RawFormer.forward``)."""
    if not raw.is_cuda:
        raise RuntimeError("pack_raw: the SID front-end runs on the ROCm device only (no CPU fallback)")
    if raw.dtype not in (torch.uint16, torch.int16):
        raise TypeError(f"pack_raw: expected a uint16 Bayer frame, got {raw.dtype}")
    if raw.dim() == 2:
        raw = raw.unsqueeze(0)
    raw = raw.contiguous()
    B, H2, W2 = raw.shape
    if H2 % 2 or W2 % 8:
        raise ValueError(f"pack_raw: mosaic {H2}x{W2} must have an even height and a width that is a multiple of 8")
    mode = _SID_MODES[layout]
    h, w = H2 // 2, W2 // 2
    out = torch.empty((B, 1, H2, W2) if mode == 2 else (B, 4, h, w), dtype=torch.float32, device=raw.device)
    with torch.cuda.device(raw.device):
        _lib.check(_lib.load().rf_sid_pack(C.c_void_p(raw.data_ptr()), C.c_void_p(out.data_ptr()), B, h, w, int(black_level),
                                           int(white_level), float(ratio), mode, _stream(raw)), "rf_sid_pack")
    return out
