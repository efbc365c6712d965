"""Deterministic synthetic data for the RawFormer hot path.

Nothing here comes from the reference: the reference ships neither weights nor
datasets, so parity and throughput are measured on

* parameters regenerated from ``(seed, parameter name, shape)`` by a
  counter-based hash, so this container, the GPU box and every rank produce the
  same floats without shipping a checkpoint, and
* synthetic low-light Bayer mosaics: a smooth scene under
  RGGB colour-filter gains, scaled into the low-light range, with
  signal-dependent noise, amplified and clamped to [0, 1].

Only numpy is used; values are float32.
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wraps modulo 2**64)."""
    z = (z + _GOLD) & _MASK
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def uniform01(seed: int, name: str, count: int, offset: int = 0) -> np.ndarray:
    """``count`` float32 values in [0, 1): value i depends only on (seed, name, offset + i)."""
    with np.errstate(over="ignore"):
        key = _mix64(np.array([(seed * 0x2545F4914F6CDD1D + _fnv1a64(name)) & 0xFFFFFFFFFFFFFFFF],
                              dtype=np.uint64))[0]
        idx = np.arange(offset, offset + count, dtype=np.uint64)
        h = _mix64(key + idx * _GOLD)
    return ((h >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def uniform(seed: int, name: str, shape, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return (lo + (hi - lo) * u.astype(np.float64)).astype(np.float32).reshape(shape)


def param_values(seed: int, name: str, shape) -> np.ndarray:
    """Deterministic value for one parameter of a RawFormer state_dict.

    The rule is keyed on the *trailing* part of the name so both key layouts
     get sensible magnitudes: conv weights are uniform
    with variance 1/fan_in, norm scales are near one, everything else small.
    """
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("r_w", "g_w", "b_w", "filt", "y_weights"):
        raise ValueError(f"{name} is a fixed buffer, not a generated parameter")
    if leaf == "wb_gains":          # TrueColorRawFormer front end (BayerTORGBColorMultiLvl.py:78-83): camera-like values
        return uniform(seed, name, shape, 0.5, 2.0)
    if leaf == "color_matrix":      # [3, 4] = 3x3 matrix near identity | bias column
        return (np.eye(3, 4, dtype=np.float32) + uniform(seed, name, shape, -0.1, 0.1)).astype(np.float32)
    if leaf == "gamma_param":
        return uniform(seed, name, shape, 1.5, 2.5)
    if leaf in ("alpha", "beta", "gamma"):
        return uniform(seed, name, shape, 0.5, 1.5)
    if leaf == "running_var":  # BatchNorm statistics must stay positive
        return uniform(seed, name, shape, 0.5, 1.5)
    if leaf in ("temperature", "scale", "log_temperature"):
        return uniform(seed, name, shape, 0.5, 2.0)
    if len(shape) == 1:
        if leaf == "weight":  # LayerNorm scale
            return uniform(seed, name, shape, 0.8, 1.2)
        return uniform(seed, name, shape, -0.1, 0.1)  # biases
    if len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        if ".up" in "." + name and shape[2] == 2:  # ConvTranspose2d weight [Cin, Cout, 2, 2]
            fan_in = shape[0]
        bound = float(np.sqrt(3.0 / fan_in))
        return uniform(seed, name, shape, -bound, bound)
    return uniform(seed, name, shape, -0.1, 0.1)


def fill_state_dict(state_dict, seed: int):
    """Overwrite every generated parameter of ``state_dict`` in place (torch tensors)."""
    import torch

    for name, t in state_dict.items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf in ("r_w", "g_w", "b_w", "filt", "y_weights") or not t.dtype.is_floating_point:
            continue
        v = param_values(seed, name, tuple(t.shape))
        with torch.no_grad():
            t.copy_(torch.from_numpy(v).reshape(t.shape))
    return state_dict


def random_mosaic(seed: int, batch: int, height: int, width: int) -> np.ndarray:
    """Uniform-random mosaic ``[B, 1, height, width]`` in [0, 1) (BASELINE config 1)."""
    return uniform01(seed, "mosaic.uniform", batch * height * width).reshape(batch, 1, height, width)


def bayer_mosaic(seed: int, batch: int, height: int, width: int) -> np.ndarray:
    """Synthetic low-light RGGB mosaic ``[B, 1, height, width]`` float32 in [0, 1].

    Image ``b`` uses seed ``seed + b`` so a rank can generate only its shard.
    """
    out = np.empty((batch, 1, height, width), dtype=np.float32)
    yy = (np.arange(height, dtype=np.float64) / height)[:, None]
    xx = (np.arange(width, dtype=np.float64) / width)[None, :]
    gains = np.empty((2, 2), dtype=np.float64)
    gains[0, 0], gains[0, 1], gains[1, 0], gains[1, 1] = 0.5, 1.0, 1.0, 0.6  # R G / G B
    cfa = np.tile(gains, (height // 2 + 1, width // 2 + 1))[:height, :width]
    for b in range(batch):
        s = seed + b
        prm = uniform01(s, "bayer.scene", 8 * 5).astype(np.float64).reshape(8, 5)
        scene = np.zeros((height, width), dtype=np.float64)
        for fx, fy, ph, amp, _ in prm:
            scene += (0.25 + amp) * np.cos(2 * np.pi * ((1 + 6 * fx) * xx + (1 + 6 * fy) * yy + ph))
        scene = (scene - scene.min()) / max(scene.max() - scene.min(), 1e-12)
        level = 0.02 + 0.08 * float(prm[0, 4])
        clean = scene * cfa * level
        # signal-dependent noise from two hashed uniforms (Box-Muller), var = a*x + b
        u1 = uniform01(s, "bayer.noise.u1", height * width).astype(np.float64).reshape(height, width)
        u2 = uniform01(s, "bayer.noise.u2", height * width).astype(np.float64).reshape(height, width)
        gauss = np.sqrt(-2.0 * np.log(np.maximum(u1, 2.0 ** -24))) * np.cos(2 * np.pi * u2)
        noisy = clean + gauss * np.sqrt(1e-3 * clean + 1e-5) * 0.1
        out[b, 0] = np.clip(noisy * (0.5 / level), 0.0, 1.0).astype(np.float32)
    return out


def smooth_rgb(seed: int, batch: int, height: int, width: int) -> np.ndarray:
    """Seeded smooth RGB ground truth ``[B, 3, height, width]`` in [0, 1] for PSNR-vs-GT checks."""
    out = np.empty((batch, 3, height, width), dtype=np.float32)
    yy = (np.arange(height, dtype=np.float64) / height)[:, None]
    xx = (np.arange(width, dtype=np.float64) / width)[None, :]
    for b in range(batch):
        prm = uniform01(seed + b, "gt.scene", 3 * 4 * 3).astype(np.float64).reshape(3, 4, 3)
        for c in range(3):
            img = np.zeros((height, width), dtype=np.float64)
            for fx, fy, ph in prm[c]:
                img += np.cos(2 * np.pi * ((1 + 3 * fx) * xx + (1 + 3 * fy) * yy + ph))
            out[b, c] = (0.5 + img / 8.0).astype(np.float32)
    return np.clip(out, 0.0, 1.0)
