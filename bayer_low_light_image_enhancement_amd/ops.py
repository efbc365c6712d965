"""Operator-level Python surface over the C ABI (same kernels the whole forward uses).

Function names and argument meaning follow the reference's Python
(``downshuffle``, ``dwt_init`` / ``iwt_init``, ``CustomDWT`` / ``CustomIDWT``, ``HaarDWT``,
``LayerNorm``, ``Attention`` ...), so the parity tests read like calls into the reference.
Tensors must be float32, contiguous, on a ROCm device; torch is only the allocator and the
stream provider here.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib


def _chk(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if t.device.type != "cuda":
        raise RuntimeError(f"{name}: tensor is on {t.device}; the RawFormer HIP path has no CPU implementation")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _scratch(nbytes: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=like.device)


def downshuffle(x: torch.Tensor, r: int = 2) -> torch.Tensor:
    """``downshuffle(var, 2)`` (RawFomer_WFB_FFAB/model.py:287-298): [B,C,2h,2w] -> [B,4C,h,w]."""
    if r != 2:
        raise ValueError("only r=2 is used by RawFormer")
    x = _chk(x, "x")
    b, c, h2, w2 = x.shape
    if h2 % 2 or w2 % 2:
        raise RuntimeError(f"downshuffle: spatial size {h2}x{w2} not divisible by 2")
    out = torch.empty((b, 4 * c, h2 // 2, w2 // 2), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().rf_pixel_unshuffle2(_ptr(x), _ptr(out), b, c, h2 // 2, w2 // 2, _stream(x)), "rf_pixel_unshuffle2")
    return out


def pixel_shuffle(x: torch.Tensor, r: int = 2) -> torch.Tensor:
    """``nn.PixelShuffle(2)`` (RawFomer_WFB_FFAB/model.py:471,507): [B,4C,h,w] -> [B,C,2h,2w]."""
    if r != 2:
        raise ValueError("only r=2 is used by RawFormer")
    x = _chk(x, "x")
    b, c4, h, w = x.shape
    if c4 % 4:
        raise RuntimeError(f"pixel_shuffle: {c4} channels not divisible by 4")
    out = torch.empty((b, c4 // 4, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().rf_pixel_shuffle2(_ptr(x), _ptr(out), b, c4 // 4, h, w, _stream(x)), "rf_pixel_shuffle2")
    return out


def dwt_init(x: torch.Tensor) -> torch.Tensor:
    """``dwt_init`` / ``DWT`` (RawFomer_WFB_FFAB/blocks.py:102-115): [B,C,2h,2w] -> [4B,C,h,w]."""
    x = _chk(x, "x")
    b, c, h2, w2 = x.shape
    if h2 % 2 or w2 % 2:
        raise RuntimeError(f"dwt_init: spatial size {h2}x{w2} not divisible by 2")
    out = torch.empty((4 * b, c, h2 // 2, w2 // 2), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().rf_dwt_haar(_ptr(x), _ptr(out), b, c, h2 // 2, w2 // 2, _stream(x)), "rf_dwt_haar")
    return out


def iwt_init(x: torch.Tensor) -> torch.Tensor:
    """``iwt_init`` / ``IWT`` (RawFomer_WFB_FFAB/blocks.py:119-136): [4B,C,h,w] -> [B,C,2h,2w]."""
    x = _chk(x, "x")
    b4, c, h, w = x.shape
    if b4 % 4:
        raise RuntimeError(f"iwt_init: batch {b4} not divisible by 4")
    out = torch.empty((b4 // 4, c, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().rf_idwt_haar(_ptr(x), _ptr(out), b4 // 4, c, h, w, _stream(x)), "rf_idwt_haar")
    return out


def _k16(kernel: Sequence[Sequence[float]]):
    flat = [float(v) for row in kernel for v in row]
    if len(flat) != 16:
        raise ValueError("kernel must be 4x4")
    return (C.c_float * 16)(*flat)


DEFAULT_KERNEL = ((1, 1, 1, 1), (1, -1, 1, 1), (1, 1, -1, 1), (1, 1, 1, -1))  # README.md:98-103


def custom_dwt(x: torch.Tensor, kernel=DEFAULT_KERNEL, norm: bool = True) -> torch.Tensor:
    """``CustomDWT(kernel, norm=norm)(x)`` (README.md:92-117): [B,C,2h,2w] -> [B,4C,h,w]."""
    x = _chk(x, "x")
    b, c, h2, w2 = x.shape
    if h2 % 2 or w2 % 2:
        raise RuntimeError(f"CustomDWT: spatial size {h2}x{w2} not divisible by 2")
    out = torch.empty((b, 4 * c, h2 // 2, w2 // 2), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().rf_dwt_custom(_ptr(x), _ptr(out), _k16(kernel), int(norm), b, c, h2 // 2, w2 // 2, _stream(x)),
                   "rf_dwt_custom")
    return out


def custom_idwt(x: torch.Tensor, kernel=DEFAULT_KERNEL, norm: bool = True) -> torch.Tensor:
    """``CustomIDWT(kernel, norm=norm)(x)`` (README.md:120-144): [B,4C,h,w] -> [B,C,2h,2w]."""
    x = _chk(x, "x")
    b, c4, h, w = x.shape
    if c4 % 4:
        raise RuntimeError(f"CustomIDWT: {c4} channels not divisible by 4")
    out = torch.empty((b, c4 // 4, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().rf_idwt_custom(_ptr(x), _ptr(out), _k16(kernel), int(norm), b, c4 // 4, h, w, _stream(x)),
                   "rf_idwt_custom")
    return out


def haar_dwt(x: torch.Tensor) -> Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
    """``HaarDWT()(x)`` (FrequencyawareLumaChromaAttentionRAWFormer.py:39-73): LL, (LH, HL, HH)."""
    x = _chk(x, "x")
    b, c, h, w = x.shape
    out = torch.empty((4, b, c, (h + 1) // 2, (w + 1) // 2), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().rf_haar_dwt(_ptr(x), _ptr(out), b, c, h, w, _stream(x)), "rf_haar_dwt")
    return out[0], (out[1], out[2], out[3])


def layernorm2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    """``LayerNorm(dim)(x)`` on NCHW (FrequencyawareLumaChromaAttentionRAWFormer.py:180-187);
    ``bias=None`` is ``BiasFree_LayerNorm`` (RawFomer_WFB_FFAB/model.py:89-103)."""
    x, weight = _chk(x, "x"), _chk(weight, "weight")
    bias = None if bias is None else _chk(bias, "bias")
    b, c, h, w = x.shape
    if weight.numel() != c:
        raise RuntimeError(f"LayerNorm: weight has {weight.numel()} elements, input has {c} channels")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().rf_layernorm2d(_ptr(x), _ptr(out), _ptr(weight), _ptr(bias), float(eps), b, c, h, w, _stream(x)),
                   "rf_layernorm2d")
    return out


def conv1x1(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
            x2: Optional[torch.Tensor] = None, ln_weight: Optional[torch.Tensor] = None,
            ln_bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``nn.Conv2d(Cin, Cout, 1)``; optional fused LayerNorm prologue, ``torch.cat([x, x2], 1)``
    input and residual add."""
    x, weight = _chk(x, "x"), _chk(weight, "weight")
    b, c1, h, w = x.shape
    c2 = 0
    if x2 is not None:
        x2 = _chk(x2, "x2")
        c2 = x2.shape[1]
    cout = weight.shape[0]
    if weight.numel() != cout * (c1 + c2):
        raise RuntimeError(f"conv1x1: weight {tuple(weight.shape)} does not match {c1 + c2} input channels")
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_conv1x1_scratch_bytes(c1 + c2, cout, C.byref(sz)), "rf_conv1x1_scratch_bytes")
    scratch = _scratch(sz.value, x)
    out = torch.empty((b, cout, h, w), dtype=x.dtype, device=x.device)
    args = [None if t is None else _chk(t, "arg") for t in (bias, ln_weight, ln_bias, residual)]
    with torch.cuda.device(x.device):
        _lib.check(lib.rf_conv1x1(_ptr(x), _ptr(x2), _ptr(out), _ptr(weight), _ptr(args[0]), _ptr(args[1]), _ptr(args[2]),
                                  _ptr(args[3]), _ptr(scratch), b, c1, c2, cout, h, w, _stream(x)), "rf_conv1x1")
    return out


def dwconv3x3(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, gelu: bool = False) -> torch.Tensor:
    """``nn.Conv2d(C, C, 3, padding=1, groups=C)`` (+ exact GELU)."""
    x, weight = _chk(x, "x"), _chk(weight, "weight")
    bias = None if bias is None else _chk(bias, "bias")
    b, c, h, w = x.shape
    if weight.numel() != c * 9:
        raise RuntimeError(f"dwconv3x3: weight {tuple(weight.shape)} does not match {c} channels")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().rf_dwconv3x3(_ptr(x), _ptr(out), _ptr(weight), _ptr(bias), int(gelu), b, c, h, w, _stream(x)),
                   "rf_dwconv3x3")
    return out


def conv3x3(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act: str = "none",
            store: str = "plain") -> torch.Tensor:
    """``nn.Conv2d(Cin, Cout, 3, padding=1)``; ``act='lrelu'`` adds LeakyReLU(0.2), ``'relu'`` ReLU;
    ``store='unshuffle'`` = Downsample (a8), ``'shuffle'`` = conv_out + PixelShuffle (a10)."""
    x, weight = _chk(x, "x"), _chk(weight, "weight")
    bias = None if bias is None else _chk(bias, "bias")
    b, cin, h, w = x.shape
    cout = weight.shape[0]
    if tuple(weight.shape[1:]) != (cin, 3, 3):
        raise RuntimeError(f"conv3x3: weight {tuple(weight.shape)} does not match {cin} input channels")
    mode = {"plain": 0, "unshuffle": 1, "shuffle": 2}[store]
    shape = {0: (b, cout, h, w), 1: (b, cout * 4, h // 2, w // 2), 2: (b, cout // 4, 2 * h, 2 * w)}[mode]
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_conv3x3_scratch_bytes(cin, cout, C.byref(sz)), "rf_conv3x3_scratch_bytes")
    scratch = _scratch(sz.value, x)
    out = torch.empty(shape, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.rf_conv3x3(_ptr(x), _ptr(out), _ptr(weight), _ptr(bias), _ptr(scratch), {"none": 0, "lrelu": 1, "relu": 2}[act], mode,
                                  b, cin, cout, h, w, _stream(x)), "rf_conv3x3")
    return out


def conv_transpose2x2(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``nn.ConvTranspose2d(Cin, Cout, 2, stride=2)`` (RawFomer_WFB_FFAB/model.py:461)."""
    x, weight = _chk(x, "x"), _chk(weight, "weight")
    bias = None if bias is None else _chk(bias, "bias")
    b, cin, h, w = x.shape
    if weight.shape[0] != cin or tuple(weight.shape[2:]) != (2, 2):
        raise RuntimeError(f"ConvTranspose2d: weight {tuple(weight.shape)} does not match {cin} input channels")
    cout = weight.shape[1]
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_convT2x2_scratch_bytes(cin, cout, C.byref(sz)), "rf_convT2x2_scratch_bytes")
    scratch = _scratch(sz.value, x)
    out = torch.empty((b, cout, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.rf_convT2x2(_ptr(x), _ptr(out), _ptr(weight), _ptr(bias), _ptr(scratch), b, cin, cout, h, w, _stream(x)),
                   "rf_convT2x2")
    return out


def channel_attention(x: torch.Tensor, qkv_w, qkv_b, dw_w, dw_b, temperature, proj_w, proj_b, heads: int) -> torch.Tensor:
    """``Attention(dim, heads, bias=True)(x)`` (FrequencyawareLumaChromaAttentionRAWFormer.py:212-235)."""
    x = _chk(x, "x")
    ts = [None if t is None else _chk(t, "param") for t in (qkv_w, qkv_b, dw_w, dw_b, temperature, proj_w, proj_b)]
    b, c, h, w = x.shape
    if c % heads:
        raise RuntimeError(f"Attention: {c} channels not divisible by {heads} heads")
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_chan_attn_scratch_bytes(b, c, heads, h, w, C.byref(sz)), "rf_chan_attn_scratch_bytes")
    scratch = _scratch(sz.value, x)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.rf_chan_attn(_ptr(x), _ptr(out), *[_ptr(t) for t in ts], _ptr(scratch), b, c, heads, h, w, _stream(x)),
                   "rf_chan_attn")
    return out


def flca_guidance(x4: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """Guidance planes ``[y_low, y_high, cr, cb]`` an FLCA block sees at feature size ``size``
    (``BayerLumaChroma`` + ``HaarDWT`` + bilinear ``F.interpolate``,
    FrequencyawareLumaChromaAttentionRAWFormer.py:79-97,138-149)."""
    x4 = _chk(x4, "x4")
    b, c, h, w = x4.shape
    if c != 4:
        raise RuntimeError("flca_guidance expects packed RGGB [B,4,H,W]")
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_guidance_scratch_bytes(b, h, w, C.byref(sz)), "rf_guidance_scratch_bytes")
    scratch = _scratch(sz.value, x4)
    out = torch.empty((b, 4, size[0], size[1]), dtype=x4.dtype, device=x4.device)
    with torch.cuda.device(x4.device):
        _lib.check(lib.rf_flca_guidance(_ptr(x4), _ptr(out), _ptr(scratch), b, h, w, size[0], size[1], _stream(x4)), "rf_flca_guidance")
    return out


def bayer_luma_chroma(x4: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``BayerLumaChroma()(x4)`` -> ``(y, cr, cb)``, each ``[B,1,H,W]``
    (FrequencyawareLumaChromaAttentionRAWFormer.py:79-97): the base planes the guidance kernels keep at the head of their
    scratch area (include/rawformer_hip.h, rf_flca_guidance)."""
    x4 = _chk(x4, "x4")
    b, c, h, w = x4.shape
    if c != 4:
        raise RuntimeError("bayer_luma_chroma expects packed RGGB [B,4,H,W]")
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_guidance_scratch_bytes(b, h, w, C.byref(sz)), "rf_guidance_scratch_bytes")
    scratch = _scratch(sz.value, x4)
    out = torch.empty((b, 4, h, w), dtype=x4.dtype, device=x4.device)
    with torch.cuda.device(x4.device):
        _lib.check(lib.rf_flca_guidance(_ptr(x4), _ptr(out), _ptr(scratch), b, h, w, h, w, _stream(x4)), "rf_flca_guidance")
    planes = scratch[: 3 * b * h * w * 4].view(torch.float32).reshape(3, b, 1, h, w)
    return planes[0].clone(), planes[1].clone(), planes[2].clone()


def conv_ffn(x: torch.Tensor, pw1_w, pw1_b, dw_w, dw_b, pw2_w, pw2_b) -> torch.Tensor:
    """``conv_ffn(dim, expansion)(x)`` = 1x1 -> depthwise 3x3 -> GELU -> 1x1
    (FrequencyawareLumaChromaAttentionRAWFormer.py:190-209, RawFomer_WFB_FFAB/model.py:319-336), as the three kernels
    the forward launches for it at U-Net levels 1-3 (level 0 runs it inside the fused FFN kernel together with
    LayerNorm and the residual: ``transformer_block``)."""
    return conv1x1(dwconv3x3(conv1x1(x, pw1_w, pw1_b), dw_w, dw_b, gelu=True), pw2_w, pw2_b)


_TB_KEYS = ("norm1.body.weight", "norm1.body.bias", "attn.temperature", "attn.qkv.weight", "attn.qkv.bias",
            "attn.qkv_dwconv.weight", "attn.qkv_dwconv.bias", "attn.project_out.weight", "attn.project_out.bias",
            "norm2.body.weight", "norm2.body.bias", "ffn.pointwise1.weight", "ffn.pointwise1.bias",
            "ffn.depthwise.weight", "ffn.depthwise.bias", "ffn.pointwise2.weight", "ffn.pointwise2.bias")


def transformer_block(x: torch.Tensor, params, heads: int, ffn_expansion_factor: int = 2, prefix: str = "") -> torch.Tensor:
    """``TransformerBlock(dim, heads, ffn_expansion_factor, bias=True)(x)``
    (FrequencyawareLumaChromaAttentionRAWFormer.py:238-254).  ``params`` maps the block's
    state_dict keys (``norm1.body.weight`` ...) to device tensors.  Same schedule as the whole
    forward, i.e. the fused gfx950 kernels where the shape allows."""
    x = _chk(x, "x")
    b, c, h, w = x.shape
    ts = [_chk(params[prefix + k], k) for k in _TB_KEYS]
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_transformer_block_scratch_bytes(b, c, heads, ffn_expansion_factor, h, w, C.byref(sz)),
               "rf_transformer_block_scratch_bytes")
    scratch = _scratch(sz.value, x)
    out = torch.empty_like(x)
    ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    with torch.cuda.device(x.device):
        _lib.check(lib.rf_transformer_block(_ptr(x), _ptr(out), ptrs, _ptr(scratch), b, c, heads, ffn_expansion_factor, h, w,
                                            _stream(x)), "rf_transformer_block")
    return out


_FLCA_KEYS = ("alpha", "beta", "gamma", "low_attn.0.weight", "high_attn.0.weight", "chroma_attn.0.weight",
              "se.1.weight", "se.1.bias", "se.3.weight", "se.3.bias")


def flca(feat: torch.Tensor, x4: torch.Tensor, params, prefix: str = "") -> torch.Tensor:
    """``FLCA(channels)(feat, *BayerLumaChroma()(x4))`` (FrequencyawareLumaChromaAttentionRAWFormer.py:103-162):
    guidance from the packed RGGB frame ``x4``, spatial gates, squeeze-excite."""
    feat = _chk(feat, "feat")
    b, c, h, w = feat.shape
    guide = flca_guidance(x4, (h, w))
    ts = [_chk(params[prefix + k].reshape(-1) if params[prefix + k].dim() == 0 else params[prefix + k], k) for k in _FLCA_KEYS]
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_flca_scratch_bytes(b, c, h, w, C.byref(sz)), "rf_flca_scratch_bytes")
    scratch = _scratch(sz.value, feat)
    out = torch.empty_like(feat)
    ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    with torch.cuda.device(feat.device):
        _lib.check(lib.rf_flca(_ptr(feat), _ptr(guide), _ptr(out), ptrs, _ptr(scratch), b, c, h, w, _stream(feat)), "rf_flca")
    return out


def token_attention(qkv: torch.Tensor, heads: int, scale: Optional[float] = None) -> torch.Tensor:
    """``softmax(q k^T * scale) v`` per (image, head) with tokens = pixels (Attenblock.py:212-217), flash style:
    ``qkv`` is ``[B, 3*heads*d, H, W]`` (q, k, v thirds, channel = head*d + i), the result ``[B, heads*d, H, W]``."""
    qkv = _chk(qkv, "qkv")
    b, c3, h, w = qkv.shape
    if c3 % (3 * heads):
        raise RuntimeError(f"token_attention: {c3} channels are not 3 x {heads} heads x d")
    inner = c3 // 3
    d, n = inner // heads, h * w
    out = torch.empty((b, inner, h, w), dtype=qkv.dtype, device=qkv.device)
    q = qkv.data_ptr()
    with torch.cuda.device(qkv.device):
        _lib.check(_lib.load().rf_token_attn(C.c_void_p(q), C.c_void_p(q + 4 * inner * n), C.c_void_p(q + 8 * inner * n), _ptr(out),
                                             3 * inner * n, inner * n, b, heads, d, n, float(d ** -0.5 if scale is None else scale),
                                             _stream(qkv)), "rf_token_attn")
    return out


def luma_film(qkv: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, luma: Optional[torch.Tensor] = None,
              alpha: Optional[torch.Tensor] = None) -> torch.Tensor:
    """FiLM ``gamma * t + beta`` on the q, k, v thirds of ``qkv`` and the query bias
    ``alpha * centered(avg_pool3(1 - luma))`` (Attenblock.py:193-210)."""
    qkv, gamma, beta = _chk(qkv, "qkv"), _chk(gamma, "gamma"), _chk(beta, "beta")
    b, c3, h, w = qkv.shape
    inner = c3 // 3
    if tuple(gamma.shape) != (b, inner, h, w) or tuple(beta.shape) != (b, inner, h, w):
        raise RuntimeError(f"luma_film: gamma/beta {tuple(gamma.shape)} do not match {(b, inner, h, w)}")
    if luma is not None:
        luma = _chk(luma, "luma")
        if tuple(luma.shape) != (b, 1, h, w):
            raise RuntimeError(f"luma_film: luma {tuple(luma.shape)} must be {(b, 1, h, w)}")
        alpha = _chk(alpha, "alpha")
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_luma_film_scratch_bytes(b, h, w, C.byref(sz)), "rf_luma_film_scratch_bytes")
    scratch = _scratch(sz.value, qkv)
    out = torch.empty_like(qkv)
    with torch.cuda.device(qkv.device):
        _lib.check(lib.rf_luma_film(_ptr(qkv), _ptr(gamma), _ptr(beta), inner * h * w, _ptr(luma), _ptr(alpha), _ptr(out), _ptr(scratch),
                                    b, inner, h, w, _stream(qkv)), "rf_luma_film")
    return out


def luminance_aware_mhsa(x: torch.Tensor, luma: torch.Tensor, params, heads: int = 8, prefix: str = "") -> torch.Tensor:
    """``LuminanceAwareMHSA.forward(x, luma)`` (Attenblock.py:161-220) from its state_dict ``params``
    (``to_qkv``, ``proj``, ``luma_cond.net.{0,2}``, ``luma_cond.{gamma,beta}``, ``alpha``): 1x1 GEMM, two 3x3
    convs + ReLU, two 1x1 GEMMs, FiLM + luma bias, flash token attention, 1x1 GEMM."""
    p = lambda k: params[prefix + k]  # noqa: E731
    qkv = conv1x1(x, p("to_qkv.weight"), params.get(prefix + "to_qkv.bias"))
    hcond = conv3x3(luma, p("luma_cond.net.0.weight"), p("luma_cond.net.0.bias"), act="relu")
    hcond = conv3x3(hcond, p("luma_cond.net.2.weight"), p("luma_cond.net.2.bias"), act="relu")
    gamma = conv1x1(hcond, p("luma_cond.gamma.weight"), p("luma_cond.gamma.bias"))
    beta = conv1x1(hcond, p("luma_cond.beta.weight"), p("luma_cond.beta.bias"))
    alpha = params.get(prefix + "alpha")
    qkv = luma_film(qkv, gamma, beta, luma if alpha is not None else None, None if alpha is None else alpha.reshape(1))
    out = token_attention(qkv, heads)
    return conv1x1(out, p("proj.weight"), params.get(prefix + "proj.bias"))


def dwgate3x3(x: torch.Tensor, wa: torch.Tensor, ba: Optional[torch.Tensor], wb: torch.Tensor, bb: Optional[torch.Tensor]) -> torch.Tensor:
    """``gelu(g) * a + gelu(a) * g`` with ``a = dw3x3(x; wa, ba)``, ``g = dw3x3(x; wb, bb)`` in one pass
    (middle of the WFB ``FeedForward``, RawFomer_WFB_FFAB/model.py:55-59 with the rep-conv branch fused)."""
    x, wa, wb = _chk(x, "x"), _chk(wa, "wa"), _chk(wb, "wb")
    b, c, h, w = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().rf_dwgate3x3(_ptr(x), _ptr(out), _ptr(wa), _ptr(None if ba is None else _chk(ba, "ba")), _ptr(wb),
                                            _ptr(None if bb is None else _chk(bb, "bb")), b, c, h, w, _stream(x)), "rf_dwgate3x3")
    return out


def dwconv5x5(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``nn.Conv2d(C, C, 5, padding=2, groups=C)`` (Illumination_Estimator.depth_conv, model.py:182-183)."""
    x, weight = _chk(x, "x"), _chk(weight, "weight")
    b, c, h, w = x.shape
    if tuple(weight.shape) != (c, 1, 5, 5):
        raise RuntimeError(f"dwconv5x5: weight {tuple(weight.shape)} does not match {c} channels")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().rf_dwconv5x5(_ptr(x), _ptr(out), _ptr(weight), _ptr(None if bias is None else _chk(bias, "bias")),
                                            b, c, h, w, _stream(x)), "rf_dwconv5x5")
    return out


def fuse_rep_convs(params, prefix: str = "", eps: float = 1e-5):
    """``FeedForward.fuse()`` (RawFomer_WFB_FFAB/model.py:27-40, 66-87) on a state_dict: the depthwise 3x3 + BN,
    the depthwise 1x1 + BN and the identity as ONE depthwise 3x3 weight and bias (weight preparation, host side)."""
    def bn_fold(name):
        s = params[prefix + name + ".bn.weight"] / torch.sqrt(params[prefix + name + ".bn.running_var"] + eps)
        wgt = params[prefix + name + ".c.weight"] * s[:, None, None, None]
        return wgt, params[prefix + name + ".bn.bias"] - params[prefix + name + ".bn.running_mean"] * s
    w1, b1 = bn_fold("rep_conv1")
    w2, b2 = bn_fold("rep_conv2")
    ident = torch.zeros_like(w1)
    ident[:, :, 1, 1] = 1.0
    return (w1 + torch.nn.functional.pad(w2, [1, 1, 1, 1]) + ident).contiguous(), (b1 + b2).contiguous()


def wfb_feed_forward(x: torch.Tensor, params, prefix: str = "") -> torch.Tensor:
    """Eval-mode ``FeedForward.forward`` (RawFomer_WFB_FFAB/model.py:42-62): 1x1, fused gate pass, 1x1 + identity."""
    p = lambda k: params.get(prefix + k)  # noqa: E731
    wa, ba = fuse_rep_convs(params, prefix)
    hid = conv1x1(x, p("project_in.weight"), p("project_in.bias"))
    gated = dwgate3x3(hid, wa, ba, p("dwconv.weight"), p("dwconv.bias"))
    return conv1x1(gated, p("project_out.weight"), p("project_out.bias"), residual=x)


def illumination_estimator(img: torch.Tensor, params, prefix: str = "") -> Tuple[torch.Tensor, torch.Tensor]:
    """``Illumination_Estimator.forward`` (RawFomer_WFB_FFAB/model.py:186-200) -> (illu_fea, illu_map).  The channel
    mean that the reference concatenates folds into conv1's weights: W [img ; mean] = (W_img + w_mean / C) img."""
    img = _chk(img, "img")
    b, c, h, w = img.shape
    w1 = params[prefix + "conv1.weight"]
    if w1.shape[1] != c + 1:
        raise RuntimeError(f"illumination_estimator: conv1 expects {w1.shape[1]} = C + 1 channels, image has {c}")
    pad = (-c) % 4                                        # the GEMM kernels take channel counts in multiples of 4
    wf = torch.cat([w1[:, :c] + w1[:, c:] / c, w1.new_zeros(w1.shape[0], pad, 1, 1)], dim=1).contiguous()
    xin = torch.cat([img, img.new_zeros(b, pad, h, w)], dim=1) if pad else img
    x1 = conv1x1(xin, wf, params.get(prefix + "conv1.bias"))
    fea = dwconv5x5(x1, params[prefix + "depth_conv.weight"], params.get(prefix + "depth_conv.bias"))
    return fea, conv1x1(fea, params[prefix + "conv2.weight"], params.get(prefix + "conv2.bias"))


def upsample_cat_reduce(x: torch.Tensor, skip: torch.Tensor, up_w: torch.Tensor, up_b: Optional[torch.Tensor],
                        cr_w: torch.Tensor, cr_b: Optional[torch.Tensor]) -> torch.Tensor:
    """Decoder step ``channel_reduce(cat[up(x), skip])`` (RawFomer_WFB_FFAB/model.py:494-503) as one kernel on composed
    weights: ``x`` [B,2C,h,w], ``skip`` [B,C,2h,2w] -> [B,C,2h,2w]."""
    x, skip, up_w, cr_w = _chk(x, "x"), _chk(skip, "skip"), _chk(up_w, "up_w"), _chk(cr_w, "cr_w")
    b, c2, h, w = x.shape
    c = c2 // 2
    if tuple(skip.shape) != (b, c, 2 * h, 2 * w) or tuple(up_w.shape) != (c2, c, 2, 2) or tuple(cr_w.shape) != (c, c2, 1, 1):
        raise RuntimeError("upsample_cat_reduce: shapes do not describe ConvTranspose2d(2C,C,2,2) + cat + Conv2d(2C,C,1)")
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_upcat_scratch_bytes(c, C.byref(sz)), "rf_upcat_scratch_bytes")
    scratch = _scratch(sz.value, x)
    out = torch.empty_like(skip)
    with torch.cuda.device(x.device):
        _lib.check(lib.rf_upcat(_ptr(x), _ptr(skip), _ptr(out), _ptr(up_w), _ptr(None if up_b is None else _chk(up_b, "up_b")), _ptr(cr_w),
                                _ptr(None if cr_b is None else _chk(cr_b, "cr_b")), _ptr(scratch), b, c, h, w, _stream(x)), "rf_upcat")
    return out


def atten_transformer_block(x: torch.Tensor, luma: torch.Tensor, params, heads: int = 8, prefix: str = "") -> torch.Tensor:
    """``Attenblock.TransformerBlock.forward(x, luma=luma)`` (Attenblock.py:225-236): ``x + attn(LN1(x))`` then
    ``x + ffn(LN2(x))`` with ``LuminanceAwareMHSA`` and ``ConvFFN``.  Both LayerNorms ride in the prologue of the 1x1
    GEMM that consumes them, both residuals in the epilogue of the 1x1 GEMM that produces the branch."""
    p = lambda k: params.get(prefix + k)  # noqa: E731
    a = prefix + "attn."
    qkv = conv1x1(x, params[a + "to_qkv.weight"], params.get(a + "to_qkv.bias"), ln_weight=p("norm1.body.weight"), ln_bias=p("norm1.body.bias"))
    hcond = conv3x3(luma, params[a + "luma_cond.net.0.weight"], params[a + "luma_cond.net.0.bias"], act="relu")
    hcond = conv3x3(hcond, params[a + "luma_cond.net.2.weight"], params[a + "luma_cond.net.2.bias"], act="relu")
    gamma = conv1x1(hcond, params[a + "luma_cond.gamma.weight"], params[a + "luma_cond.gamma.bias"])
    beta = conv1x1(hcond, params[a + "luma_cond.beta.weight"], params[a + "luma_cond.beta.bias"])
    alpha = params.get(a + "alpha")
    qkv = luma_film(qkv, gamma, beta, luma if alpha is not None else None, None if alpha is None else alpha.reshape(1))
    x1 = conv1x1(token_attention(qkv, heads), params[a + "proj.weight"], params.get(a + "proj.bias"), residual=x)
    f = prefix + "ffn."
    hid = conv1x1(x1, params[f + "pointwise1.weight"], params.get(f + "pointwise1.bias"), ln_weight=p("norm2.body.weight"), ln_bias=p("norm2.body.bias"))
    hid = dwconv3x3(hid, params[f + "depthwise.weight"], params.get(f + "depthwise.bias"), gelu=True)
    return conv1x1(hid, params[f + "pointwise2.weight"], params.get(f + "pointwise2.bias"), residual=x1)


def bayer_luma(mosaic: torch.Tensor, pattern: str = "rggb") -> torch.Tensor:
    """``BayerLuma(pattern)(mosaic)`` (Attenblock.py:79-138): ``[B,1,H,W]`` mosaic -> normalised luma ``[B,1,H,W]``."""
    mosaic = _chk(mosaic, "mosaic")
    b, c, h, w = mosaic.shape
    if c != 1:
        raise RuntimeError("bayer_luma expects a one-channel mosaic")
    pat = {"rggb": 0, "bggr": 1, "grbg": 2, "gbrg": 3}[pattern.lower()]
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_bayer_luma_scratch_bytes(b, h, w, C.byref(sz)), "rf_bayer_luma_scratch_bytes")
    scratch = _scratch(sz.value, mosaic)
    out = torch.empty_like(mosaic)
    with torch.cuda.device(mosaic.device):
        _lib.check(lib.rf_bayer_luma(_ptr(mosaic), _ptr(out), _ptr(scratch), b, h, w, pat, _stream(mosaic)), "rf_bayer_luma")
    return out


# ------------------------------------------------------------------------------------------ FFAB / FEB (f2)
_FEB_KEYS = ("fpre.weight", "fpre.bias", "process1.0.weight", "process1.0.bias", "process1.2.weight", "process1.2.bias",
             "process2.0.weight", "process2.0.bias", "process2.2.weight", "process2.2.bias")
_PB_KEYS = tuple("frequency_process." + k for k in _FEB_KEYS) + ("cat.weight", "cat.bias")
_FFAB_KEYS = (("conv0.0.weight", "conv0.0.bias") + tuple("conv0.1." + k for k in _PB_KEYS)
              + tuple(f"conv{i}." + k for i in (1, 2, 3) for k in _PB_KEYS)
              + tuple(f"{n}.0." + k for n in ("conv4",) for k in _PB_KEYS) + ("conv4.1.weight", "conv4.1.bias")
              + tuple("conv5.0." + k for k in _PB_KEYS) + ("conv5.1.weight", "conv5.1.bias")
              + tuple("convout.0." + k for k in _PB_KEYS) + ("convout.1.weight", "convout.1.bias"))


def rfft2_polar(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """``f = torch.fft.rfft2(x, norm='ortho'); (f.abs() + 1e-6, f.angle())`` (FEB, RawFomer_WFB_FFAB/blocks.py:27-29)."""
    x = _chk(x, "x")
    b, c, h, w = x.shape
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_rfft2_polar_scratch_bytes(b * c, h, w, C.byref(sz)), "rf_rfft2_polar_scratch_bytes")
    scratch = _scratch(sz.value, x)
    mag = torch.empty((b, c, h, w // 2 + 1), dtype=x.dtype, device=x.device)
    pha = torch.empty_like(mag)
    with torch.cuda.device(x.device):
        _lib.check(lib.rf_rfft2_polar(_ptr(x), _ptr(mag), _ptr(pha), _ptr(scratch), b * c, h, w, _stream(x)), "rf_rfft2_polar")
    return mag, pha


def polar_irfft2(mag: torch.Tensor, pha: torch.Tensor, width: int) -> torch.Tensor:
    """``torch.fft.irfft2(torch.complex(mag * cos(pha), mag * sin(pha)), s=(h, width), norm='ortho')`` (blocks.py:32-35)."""
    mag, pha = _chk(mag, "mag"), _chk(pha, "pha")
    b, c, h, wf = mag.shape
    if wf != width // 2 + 1:
        raise RuntimeError(f"half spectrum of width {wf} does not belong to an output width of {width}")
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_rfft2_polar_scratch_bytes(b * c, h, width, C.byref(sz)), "rf_rfft2_polar_scratch_bytes")
    scratch = _scratch(sz.value, mag)
    out = torch.empty((b, c, h, width), dtype=mag.dtype, device=mag.device)
    with torch.cuda.device(mag.device):
        _lib.check(lib.rf_polar_irfft2(_ptr(mag), _ptr(pha), _ptr(out), _ptr(scratch), b * c, h, width, _stream(mag)), "rf_polar_irfft2")
    return out


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def feb(x: torch.Tensor, params, prefix: str = "") -> torch.Tensor:
    """``FEB(nc)(x)`` (RawFomer_WFB_FFAB/blocks.py:11-39)."""
    x = _chk(x, "x")
    b, c, h, w = x.shape
    ts = [_chk(params[prefix + k], k) for k in _FEB_KEYS]
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_feb_scratch_bytes(b, c, h, w, C.byref(sz)), "rf_feb_scratch_bytes")
    scratch = _scratch(sz.value, x)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.rf_feb(_ptr(x), _ptr(out), _ptr_array(ts), _ptr(scratch), b, c, h, w, _stream(x)), "rf_feb")
    return out


def ffab(x: torch.Tensor, params, prefix: str = "", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``FFAB(nc)(x)`` (RawFomer_WFB_FFAB/blocks.py:59-92): seven ProcessBlocks (FEB + 1x1 + residual), dense concatenations.
    ``out``: an existing contiguous tensor (or leading slice of one) of x's shape to write into -- lets a caller place the result
    where the next operator reads it instead of concatenating afterwards."""
    x = _chk(x, "x")
    b, c, h, w = x.shape
    ts = [_chk(params[prefix + k], k) for k in _FFAB_KEYS]
    lib = _lib.load()
    sz = C.c_size_t()
    _lib.check(lib.rf_ffab_scratch_bytes(b, c, h, w, C.byref(sz)), "rf_ffab_scratch_bytes")
    scratch = _scratch(sz.value, x)
    if out is None:
        out = torch.empty_like(x)
    elif tuple(out.shape) != tuple(x.shape) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != x.device:
        raise RuntimeError(f"ffab: `out` must be a contiguous float32 tensor of shape {tuple(x.shape)} on {x.device}")
    with torch.cuda.device(x.device):
        _lib.check(lib.rf_ffab(_ptr(x), _ptr(out), _ptr_array(ts), _ptr(scratch), b, c, h, w, _stream(x)), "rf_ffab")
    return out


def wmb_ll_branch(x: torch.Tensor, params, prefix: str = "", high=None) -> torch.Tensor:
    """The wavelet branch of ``WMB.forward`` (RawFomer_WFB_FFAB/model.py:215-243) with the Mamba module ``mb`` (``mamba_ssm``:
    absent offline, parity unpinned) replaced by ``high`` (a callable on the ``[3B,C,h/2,w/2]`` high bands; identity when
    ``None``):  ``t = 2 LN(x) - 1;  LL, hi = DWT(t);  LL = FFAB(illu(LL)[0]);  t + clamp((IWT(cat(LL, high(hi))) + 1) / 2, 0, 1)``.
    ``2 LN(x) - 1`` is the LayerNorm kernel with weight ``2 w`` and bias ``2 b - 1``."""
    x = _chk(x, "x")
    n = x.shape[0]
    t = layernorm2d(x, 2.0 * params[prefix + "norm1.body.weight"], 2.0 * params[prefix + "norm1.body.bias"] - 1.0)
    d = dwt_init(t)
    fea, _ = illumination_estimator(d[:n], params, prefix + "illu.")      # d[:n] = the LL band: a contiguous leading slice, no copy
    if high is not None:
        d[n:].copy_(high(d[n:]))                                         # the caller's module (Mamba in the reference) on the high bands
    ffab(fea, params, prefix + "ffab.", out=d[:n])                        # LL band replaced in place: IWT reads [LL ; high] as it lies
    y = iwt_init(d)
    out = torch.empty_like(t)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().rf_affine_clamp_add(_ptr(y), _ptr(t), _ptr(out), t.numel(), 0.5, 0.5, 0.0, 1.0, _stream(x)), "rf_affine_clamp_add")
    return out
