"""CPU oracle for the RawFormer inference hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a from-the-math restatement (plain torch CPU ops, float32) of the
functions SURVEY.md section 8(a) lists.  It exists so the HIP path can be
checked on the GPU box, where ``/root/reference`` does not exist.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it; the product package never does.

Pinning: every function here is compared, in this container, with the
reference's own Python classes run on the same inputs
(``oracle/make_golden.py``), and the reference's outputs are committed as
fixtures under ``tests/golden/`` (``tests/test_oracle_golden.py`` re-checks the
restatement against them everywhere).  The reference has no golden vectors or
unit tests of its own for this path.

Parameter names follow the reference's ``state_dict`` layout of
``FrequencyawareLumaChromaAttentionRAWFormer.RawFormer`` /
``RawFomer_WFB_FFAB/model.py``.  All citations are
relative to the reference tree.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- a1 / a10
def pixel_unshuffle2(x: Tensor) -> Tensor:
    """Bayer pack, ``out[b, 4c+2i+j, y, x] = in[b, c, 2y+i, 2x+j]``.

    Follows ``downshuffle(var, 2)``: RawFomer_WFB_FFAB/model.py:287-298
    (same function: FrequencyawareLumaChromaAttentionRAWFormer.py:18-33).
    """
    b, c, h, w = x.shape
    t = x.reshape(b, c, h // 2, 2, w // 2, 2)
    return t.permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, h // 2, w // 2).contiguous()


def pixel_shuffle2(x: Tensor) -> Tensor:
    """``out[b, c, 2y+i, 2x+j] = in[b, 4c+2i+j, y, x]`` (``nn.PixelShuffle(2)``,
    RawFomer_WFB_FFAB/model.py:471,507)."""
    b, c4, h, w = x.shape
    c = c4 // 4
    t = x.reshape(b, c, 2, 2, h, w)
    return t.permute(0, 1, 4, 2, 5, 3).reshape(b, c, h * 2, w * 2).contiguous()


# --------------------------------------------------------------------------- a4
def layernorm2d(x: Tensor, weight: Tensor, bias: Optional[Tensor], eps: float = 1e-5) -> Tensor:
    """Per-pixel LayerNorm over channels of an NCHW tensor, biased variance.

    With ``bias``: FrequencyawareLumaChromaAttentionRAWFormer.py:180-187 (``nn.LayerNorm``),
    RawFomer_WFB_FFAB/model.py:106-120.  ``bias=None`` is the BiasFree form, which does
    NOT subtract the mean from the numerator: RawFomer_WFB_FFAB/model.py:89-103.
    """
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
    w = weight.view(1, -1, 1, 1)
    if bias is None:
        return x / torch.sqrt(var + eps) * w
    return (x - mu) / torch.sqrt(var + eps) * w + bias.view(1, -1, 1, 1)


# --------------------------------------------------------------------------- a5
def channel_attention(x: Tensor, qkv_w: Tensor, qkv_b: Optional[Tensor], dw_w: Tensor,
                      dw_b: Optional[Tensor], temperature: Tensor, proj_w: Tensor,
                      proj_b: Optional[Tensor], heads: int) -> Tensor:
    """Transposed (channel x channel) multi-head self-attention.

    FrequencyawareLumaChromaAttentionRAWFormer.py:212-235, RawFomer_WFB_FFAB/model.py:338-370,
    model.py:56-79.  ``temperature`` holds one value per head in any shape
    (``[heads,1,1]`` or ``[1,heads,1,1]``).
    """
    b, c, h, w = x.shape
    qkv = F.conv2d(x, qkv_w, qkv_b)
    qkv = F.conv2d(qkv, dw_w, dw_b, padding=1, groups=qkv.shape[1])
    q, k, v = qkv.reshape(b, 3, heads, c // heads, h * w).unbind(dim=1)
    qn = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    kn = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    attn = torch.einsum("bhin,bhjn->bhij", qn, kn) * temperature.reshape(1, heads, 1, 1)
    attn = torch.softmax(attn, dim=-1)
    out = torch.einsum("bhij,bhjn->bhin", attn, v).reshape(b, c, h, w)
    return F.conv2d(out, proj_w, proj_b)


# --------------------------------------------------------------------------- a6
def conv_ffn(x: Tensor, pw1_w: Tensor, pw1_b: Tensor, dw_w: Tensor, dw_b: Tensor,
             pw2_w: Tensor, pw2_b: Tensor) -> Tensor:
    """1x1 -> depthwise 3x3 -> exact (erf) GELU -> 1x1.
    FrequencyawareLumaChromaAttentionRAWFormer.py:190-209, RawFomer_WFB_FFAB/model.py:319-336,
    model.py:41-54."""
    t = F.conv2d(x, pw1_w, pw1_b)
    t = F.conv2d(t, dw_w, dw_b, padding=1, groups=t.shape[1])
    t = 0.5 * t * (1.0 + torch.erf(t * (1.0 / math.sqrt(2.0))))
    return F.conv2d(t, pw2_w, pw2_b)


# --------------------------------------------------------------------------- a11 - a14
def dwt_init(x: Tensor) -> Tensor:
    """2x2 Haar analysis, sub-bands LL,HL,LH,HH stacked on the BATCH axis.
    RawFomer_WFB_FFAB/blocks.py:102-115."""
    a = x[:, :, 0::2, 0::2] / 2  # (row 0, col 0)
    b = x[:, :, 1::2, 0::2] / 2  # (row 1, col 0)
    c = x[:, :, 0::2, 1::2] / 2  # (row 0, col 1)
    d = x[:, :, 1::2, 1::2] / 2  # (row 1, col 1)
    return torch.cat((a + b + c + d, -a - b + c + d, -a + b - c + d, a - b - c + d), dim=0)


def iwt_init(x: Tensor) -> Tensor:
    """Inverse of :func:`dwt_init`.  RawFomer_WFB_FFAB/blocks.py:119-136."""
    n = x.shape[0] // 4
    ll, hl, lh, hh = (x[i * n:(i + 1) * n] / 2 for i in range(4))
    out = x.new_zeros((n, x.shape[1], x.shape[2] * 2, x.shape[3] * 2), dtype=torch.float32)
    out[:, :, 0::2, 0::2] = ll - hl - lh + hh
    out[:, :, 1::2, 0::2] = ll - hl + lh - hh
    out[:, :, 0::2, 1::2] = ll + hl - lh - hh
    out[:, :, 1::2, 1::2] = ll + hl + lh + hh
    return out


DEFAULT_CUSTOM_KERNEL = ((1.0, 1.0, 1.0, 1.0), (1.0, -1.0, 1.0, 1.0),
                         (1.0, 1.0, -1.0, 1.0), (1.0, 1.0, 1.0, -1.0))  # README.md:98-103


def custom_dwt(x: Tensor, kernel: Sequence[Sequence[float]] = DEFAULT_CUSTOM_KERNEL,
               norm: bool = True) -> Tensor:
    """``y[b, s*C+ch, y, x] = sum_{i,j} K[s, 2i+j] * x[b, ch, 2y+i, 2x+j]`` (K/2 if ``norm``).
    ``CustomDWT``: README.md:92-117 (sub-band-major channel order, README.md:116)."""
    k = torch.tensor(kernel, dtype=torch.float32)
    if norm:
        k = k / 2.0
    bsz, c, h, w = x.shape
    taps = torch.stack((x[:, :, 0::2, 0::2], x[:, :, 0::2, 1::2],
                        x[:, :, 1::2, 0::2], x[:, :, 1::2, 1::2]), dim=1)  # [B, 4(tap), C, h/2, w/2]
    out = torch.einsum("st,btchw->bschw", k, taps)
    return out.reshape(bsz, 4 * c, h // 2, w // 2)


def custom_idwt(x: Tensor, kernel: Sequence[Sequence[float]] = DEFAULT_CUSTOM_KERNEL,
                norm: bool = True) -> Tensor:
    """Transposed conv with the same K: ``out[b, ch, 2y+i, 2x+j] = sum_s K[s, 2i+j] * x[b, s*C+ch, y, x]``.
    ``CustomIDWT``: README.md:120-144."""
    k = torch.tensor(kernel, dtype=torch.float32)
    if norm:
        k = k / 2.0
    bsz, c4, h, w = x.shape
    c = c4 // 4
    bands = x.reshape(bsz, 4, c, h, w)
    taps = torch.einsum("st,bschw->btchw", k, bands)  # [B, tap, C, h, w]
    out = x.new_zeros((bsz, c, 2 * h, 2 * w))
    out[:, :, 0::2, 0::2] = taps[:, 0]
    out[:, :, 0::2, 1::2] = taps[:, 1]
    out[:, :, 1::2, 0::2] = taps[:, 2]
    out[:, :, 1::2, 1::2] = taps[:, 3]
    return out


def haar_dwt(x: Tensor) -> Tuple[Tensor, Tuple[Tensor, Tensor, Tensor]]:
    """Orthonormal 2x2 Haar (all taps +-1/2), reflect-pad right/bottom when odd.
    Returns ``LL, (LH, HL, HH)`` with LH = low rows x high cols, HL = high rows x low cols.
    ``HaarDWT``: FrequencyawareLumaChromaAttentionRAWFormer.py:39-73, UnetLummaDWT.py:9-43."""
    h, w = x.shape[-2:]
    if (h & 1) or (w & 1):
        x = F.pad(x, (0, w & 1, 0, h & 1), mode="reflect")
    a = x[:, :, 0::2, 0::2]
    b = x[:, :, 0::2, 1::2]
    c = x[:, :, 1::2, 0::2]
    d = x[:, :, 1::2, 1::2]
    return (a + b + c + d) * 0.5, ((a - b + c - d) * 0.5, (a + b - c - d) * 0.5, (a - b - c + d) * 0.5)


# --------------------------------------------------------------------------- a15
def bayer_luma_chroma(x4: Tensor, eps: float = 1e-6) -> Tuple[Tensor, Tensor, Tensor]:
    """Luma / chroma guidance from packed RGGB planes (R, G1, G2, B).
    ``BayerLumaChroma``: FrequencyawareLumaChromaAttentionRAWFormer.py:79-97."""
    r = x4[:, 0:1]
    g = 0.5 * (x4[:, 1:2] + x4[:, 2:3])
    b = x4[:, 3:4]
    y = 0.299 * r + 0.587 * g + 0.114 * b
    y = y / y.amax(dim=(2, 3), keepdim=True).clamp_min(eps)
    return y, r - y, b - y


def bilinear_resize(x: Tensor, size: Tuple[int, int]) -> Tensor:
    """``F.interpolate(mode='bilinear', align_corners=False)`` written out:
    ``src = (dst + 0.5) * in/out - 0.5`` clamped at 0, neighbour index clamped at in-1."""
    def axis(n_in: int, n_out: int):
        scale = n_in / n_out
        src = ((torch.arange(n_out, dtype=torch.float32) + 0.5) * scale - 0.5).clamp_min(0.0)
        i0 = src.floor().to(torch.int64).clamp_max(n_in - 1)
        i1 = (i0 + 1).clamp_max(n_in - 1)
        lam = src - i0.to(torch.float32)
        return i0, i1, lam

    y0, y1, ly = axis(x.shape[-2], size[0])
    x0, x1, lx = axis(x.shape[-1], size[1])
    top = x[:, :, y0][:, :, :, x0] * (1 - lx) + x[:, :, y0][:, :, :, x1] * lx
    bot = x[:, :, y1][:, :, :, x0] * (1 - lx) + x[:, :, y1][:, :, :, x1] * lx
    return top * (1 - ly).view(1, 1, -1, 1) + bot * ly.view(1, 1, -1, 1)


def flca_guidance(y: Tensor, cr: Tensor, cb: Tensor, size: Tuple[int, int], eps: float = 1e-8) -> Tensor:
    """The four guidance planes an FLCA block sees at feature size ``size``:
    ``[y_low, y_high, cr, cb]`` (FrequencyawareLumaChromaAttentionRAWFormer.py:138-149)."""
    ll, (lh, hl, hh) = haar_dwt(y)
    mag = torch.sqrt(lh * lh + hl * hl + hh * hh + eps)
    return torch.cat([bilinear_resize(ll, size), bilinear_resize(mag, size),
                      bilinear_resize(cr, size), bilinear_resize(cb, size)], dim=1)


def flca(feat: Tensor, y: Tensor, cr: Tensor, cb: Tensor, p: Dict[str, Tensor], pre: str) -> Tensor:
    """Frequency-aware luma-chroma attention: FrequencyawareLumaChromaAttentionRAWFormer.py:103-162."""
    g = flca_guidance(y, cr, cb, feat.shape[-2:])
    a_low = torch.sigmoid(F.conv2d(g[:, 0:1], p[pre + "low_attn.0.weight"], padding=1))
    a_high = torch.tanh(F.conv2d(g[:, 1:2], p[pre + "high_attn.0.weight"], padding=1))
    a_chr = torch.sigmoid(F.conv2d(g[:, 2:4], p[pre + "chroma_attn.0.weight"], padding=1))
    x = feat * (1 + p[pre + "alpha"] * a_low + p[pre + "beta"] * a_high + p[pre + "gamma"] * a_chr)
    pooled = x.mean(dim=(2, 3), keepdim=True)
    hid = torch.relu(F.conv2d(pooled, p[pre + "se.1.weight"], p[pre + "se.1.bias"]))
    ch = torch.sigmoid(F.conv2d(hid, p[pre + "se.3.weight"], p[pre + "se.3.bias"]))
    return x * ch


# --------------------------------------------------------------------------- a7 / a3 / a8 / a9 / a2
@dataclass(frozen=True)
class RawFormerConfig:
    """Which of the reference's wirings is restated.

    ``variant='flca'``: FrequencyawareLumaChromaAttentionRAWFormer.py:284-370 (the only whole
    model in the reference that runs; no clamps).  ``variant='plain'``: the same U-Net with the
    conv branch of RawFomer_WFB_FFAB/model.py:393-412 (``branch_lrelu=True``) or model.py:94-108
    (``branch_lrelu=False``); ``clamp_io`` adds the clamps of RawFomer_WFB_FFAB/model.py:475,508.
    """
    dim: int = 32
    heads: Tuple[int, int, int, int] = (8, 8, 8, 8)
    variant: str = "flca"
    branch_lrelu: bool = True
    clamp_io: bool = False


def transformer_block(x: Tensor, p: Dict[str, Tensor], pre: str, heads: int) -> Tensor:
    """``x + attn(LN(x))`` then ``x + ffn(LN(x))``.
    FrequencyawareLumaChromaAttentionRAWFormer.py:238-254, model.py:81-92."""
    a = layernorm2d(x, p[pre + "norm1.body.weight"], p[pre + "norm1.body.bias"])
    x = x + channel_attention(a, p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"],
                              p[pre + "attn.qkv_dwconv.weight"], p[pre + "attn.qkv_dwconv.bias"],
                              p[pre + "attn.temperature"], p[pre + "attn.project_out.weight"],
                              p[pre + "attn.project_out.bias"], heads)
    f = layernorm2d(x, p[pre + "norm2.body.weight"], p[pre + "norm2.body.bias"])
    return x + conv_ffn(f, p[pre + "ffn.pointwise1.weight"], p[pre + "ffn.pointwise1.bias"],
                        p[pre + "ffn.depthwise.weight"], p[pre + "ffn.depthwise.bias"],
                        p[pre + "ffn.pointwise2.weight"], p[pre + "ffn.pointwise2.bias"])


def conv_transformer(x: Tensor, p: Dict[str, Tensor], pre: str, heads: int, cfg: RawFormerConfig,
                     guide: Optional[Tuple[Tensor, Tensor, Tensor]] = None) -> Tensor:
    """One U-Net stage: ``lrelu(conv3x3(conv1x1(cat[branch(x), transformer(x)])))``.
    FrequencyawareLumaChromaAttentionRAWFormer.py:257-278 (FLCA branch),
    RawFomer_WFB_FFAB/model.py:393-412 / model.py:94-108 (conv branch)."""
    if cfg.variant == "flca":
        branch = flca(x, guide[0], guide[1], guide[2], p, pre + "FLCA.")
    else:
        branch = F.conv2d(x, p[pre + "conv.weight"], p[pre + "conv.bias"], padding=1)
        if cfg.branch_lrelu:
            branch = F.leaky_relu(branch, 0.2)
    trans = transformer_block(x, p, pre + "Transformer.", heads)
    t = F.conv2d(torch.cat([branch, trans], dim=1), p[pre + "channel_reduce.weight"],
                 p[pre + "channel_reduce.bias"])
    return F.leaky_relu(F.conv2d(t, p[pre + "Conv_out.weight"], p[pre + "Conv_out.bias"], padding=1), 0.2)


def downsample(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """3x3 conv C -> C/2 then Bayer-style pack.  RawFomer_WFB_FFAB/model.py:300-307 (no bias),
    model.py:21-30 (bias)."""
    return pixel_unshuffle2(F.conv2d(x, w, b, padding=1))


def conv_transpose2x2(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """``out[b,o,2y+i,2x+j] = bias[o] + sum_k x[b,k,y,x] * W[k,o,i,j]``
    (``nn.ConvTranspose2d(2C, C, 2, stride=2)``, RawFomer_WFB_FFAB/model.py:461,464,467)."""
    t = torch.einsum("bkyx,koij->boyixj", x, w)
    bsz, o, h, _, wd, _ = t.shape
    t = t.reshape(bsz, o, 2 * h, 2 * wd)
    return t if b is None else t + b.view(1, -1, 1, 1)


def rawformer_forward(p: Dict[str, Tensor], x: Tensor, cfg: RawFormerConfig, packed: bool = False) -> Tensor:
    """Whole forward, mosaic ``[B,1,2H,2W]`` (or packed ``[B,4,H,W]`` when ``packed``) to
    ``[B,3,2H,2W]``.  Wiring: RawFomer_WFB_FFAB/model.py:473-508 ==
    FrequencyawareLumaChromaAttentionRAWFormer.py:330-370."""
    if cfg.clamp_io:
        x = x.clamp(0.0, 1.0)
    x4 = x if packed else pixel_unshuffle2(x)
    guide = bayer_luma_chroma(x4) if cfg.variant == "flca" else None
    h = cfg.heads
    t = F.conv2d(x4, p["embedding.weight"], p["embedding.bias"], padding=1)
    e1 = conv_transformer(t, p, "conv_tran1.", h[0], cfg, guide)
    e2 = conv_transformer(downsample(e1, p["down1.body.0.weight"], p.get("down1.body.0.bias")),
                          p, "conv_tran2.", h[1], cfg, guide)
    e3 = conv_transformer(downsample(e2, p["down2.body.0.weight"], p.get("down2.body.0.bias")),
                          p, "conv_tran3.", h[2], cfg, guide)
    e4 = conv_transformer(downsample(e3, p["down3.body.0.weight"], p.get("down3.body.0.bias")),
                          p, "conv_tran4.", h[3], cfg, guide)

    def up(t_in, skip, i):
        u = conv_transpose2x2(t_in, p[f"up{i}.weight"], p[f"up{i}.bias"])
        return F.conv2d(torch.cat([u, skip], dim=1), p[f"channel_reduce{i}.weight"],
                        p[f"channel_reduce{i}.bias"])

    d3 = conv_transformer(up(e4, e3, 1), p, "conv_tran5.", h[2], cfg, guide)
    d2 = conv_transformer(up(d3, e2, 2), p, "conv_tran6.", h[1], cfg, guide)
    d1 = conv_transformer(up(d2, e1, 3), p, "conv_tran7.", h[0], cfg, guide)
    out = F.leaky_relu(F.conv2d(d1, p["conv_out.weight"], p["conv_out.bias"], padding=1), 0.2)
    out = pixel_shuffle2(out)
    return out.clamp(0.0, 1.0) if cfg.clamp_io else out


# --------------------------------------------------------------------------- parameter shapes
def param_shapes(cfg: RawFormerConfig, inp_channels: int = 1, out_channels: int = 3,
                 ffn_expansion_factor: int = 2) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of the learnable parameters (fixed buffers excluded), in the
    reference's state_dict order (FrequencyawareLumaChromaAttentionRAWFormer.py:297-328)."""
    d = cfg.dim
    s: Dict[str, Tuple[int, ...]] = {}
    s["embedding.weight"] = (d, inp_channels * 4, 3, 3)
    s["embedding.bias"] = (d,)

    def stage(i: int, c: int, nh: int):
        pre = f"conv_tran{i}."
        if cfg.variant == "flca":
            f = pre + "FLCA."
            hid = max(8, c // 8)
            s[f + "alpha"] = ()
            s[f + "beta"] = ()
            s[f + "gamma"] = ()
            s[f + "low_attn.0.weight"] = (c, 1, 3, 3)
            s[f + "high_attn.0.weight"] = (c, 1, 3, 3)
            s[f + "chroma_attn.0.weight"] = (c, 2, 3, 3)
            s[f + "se.1.weight"] = (hid, c, 1, 1)
            s[f + "se.1.bias"] = (hid,)
            s[f + "se.3.weight"] = (c, hid, 1, 1)
            s[f + "se.3.bias"] = (c,)
        else:
            s[pre + "conv.weight"] = (c, c, 3, 3)
            s[pre + "conv.bias"] = (c,)
        t = pre + "Transformer."
        hc = c * ffn_expansion_factor
        s[t + "norm1.body.weight"] = (c,)
        s[t + "norm1.body.bias"] = (c,)
        s[t + "attn.temperature"] = (nh, 1, 1)
        s[t + "attn.qkv.weight"] = (3 * c, c, 1, 1)
        s[t + "attn.qkv.bias"] = (3 * c,)
        s[t + "attn.qkv_dwconv.weight"] = (3 * c, 1, 3, 3)
        s[t + "attn.qkv_dwconv.bias"] = (3 * c,)
        s[t + "attn.project_out.weight"] = (c, c, 1, 1)
        s[t + "attn.project_out.bias"] = (c,)
        s[t + "norm2.body.weight"] = (c,)
        s[t + "norm2.body.bias"] = (c,)
        s[t + "ffn.pointwise1.weight"] = (hc, c, 1, 1)
        s[t + "ffn.pointwise1.bias"] = (hc,)
        s[t + "ffn.depthwise.weight"] = (hc, 1, 3, 3)
        s[t + "ffn.depthwise.bias"] = (hc,)
        s[t + "ffn.pointwise2.weight"] = (c, hc, 1, 1)
        s[t + "ffn.pointwise2.bias"] = (c,)
        s[pre + "channel_reduce.weight"] = (c, 2 * c, 1, 1)
        s[pre + "channel_reduce.bias"] = (c,)
        s[pre + "Conv_out.weight"] = (c, c, 3, 3)
        s[pre + "Conv_out.bias"] = (c,)

    for i in range(1, 4):
        c = d * 2 ** (i - 1)
        stage(i, c, cfg.heads[i - 1])
        s[f"down{i}.body.0.weight"] = (c // 2, c, 3, 3)
    stage(4, d * 8, cfg.heads[3])
    for i, lvl in ((1, 2), (2, 1), (3, 0)):
        c = d * 2 ** lvl
        s[f"up{i}.weight"] = (2 * c, c, 2, 2)
        s[f"up{i}.bias"] = (c,)
        s[f"channel_reduce{i}.weight"] = (c, 2 * c, 1, 1)
        s[f"channel_reduce{i}.bias"] = (c,)
        stage(4 + i, c, cfg.heads[lvl])
    s["conv_out.weight"] = (out_channels * 4, d, 3, 3)
    s["conv_out.bias"] = (out_channels * 4,)
    return s


# ------------------------------------------------------------------------------------------
# a16: luminance-aware token attention (Attenblock.py:143-220).  Tokens are pixels; d = C / heads.
# ------------------------------------------------------------------------------------------
def token_attention(qkv: Tensor, heads: int) -> Tensor:
    """Attenblock.py:190-191, 212-217: softmax(q k^T d^-1/2) v per head, [B,3*inner,h,w] -> [B,inner,h,w]."""
    b, c3, h, w = qkv.shape
    inner = c3 // 3
    d = inner // heads
    q, k, v = (t.reshape(b, heads, d, h * w).transpose(2, 3) for t in qkv.chunk(3, dim=1))     # b h N d
    attn = torch.softmax(torch.matmul(q, k.transpose(2, 3)) * (d ** -0.5), dim=-1)
    return torch.matmul(attn, v).transpose(2, 3).reshape(b, inner, h, w)


def luma_film(qkv: Tensor, gamma: Tensor, beta: Tensor, luma: Optional[Tensor], alpha: Optional[Tensor]) -> Tensor:
    """Attenblock.py:193-210: FiLM on q, k, v; centred, 3x3-average-pooled (1 - luma) scaled by alpha added to q."""
    q, k, v = qkv.chunk(3, dim=1)
    q, k, v = gamma * q + beta, gamma * k + beta, gamma * v + beta
    if luma is not None:
        inv = F.avg_pool2d(1.0 - luma, 3, padding=1, stride=1)
        inv = inv - inv.mean(dim=(2, 3), keepdim=True)
        q = q + alpha * inv
    return torch.cat((q, k, v), dim=1)


def luminance_aware_mhsa(x: Tensor, luma: Tensor, p: Dict[str, Tensor], pre: str, heads: int) -> Tensor:
    """LuminanceAwareMHSA.forward(x, luma) (Attenblock.py:177-220) incl. LumaCond (143-159)."""
    qkv = F.conv2d(x, p[pre + "to_qkv.weight"], p.get(pre + "to_qkv.bias"))
    hc = F.relu(F.conv2d(luma, p[pre + "luma_cond.net.0.weight"], p[pre + "luma_cond.net.0.bias"], padding=1))
    hc = F.relu(F.conv2d(hc, p[pre + "luma_cond.net.2.weight"], p[pre + "luma_cond.net.2.bias"], padding=1))
    gamma = F.conv2d(hc, p[pre + "luma_cond.gamma.weight"], p[pre + "luma_cond.gamma.bias"])
    beta = F.conv2d(hc, p[pre + "luma_cond.beta.weight"], p[pre + "luma_cond.beta.bias"])
    alpha = p.get(pre + "alpha")
    qkv = luma_film(qkv, gamma, beta, luma if alpha is not None else None, alpha)
    return F.conv2d(token_attention(qkv, heads), p[pre + "proj.weight"], p.get(pre + "proj.bias"))


# ------------------------------------------------------------------------------------------
# a17: WFB extras without Mamba (RawFomer_WFB_FFAB/model.py:17-87, 174-200), eval mode
# ------------------------------------------------------------------------------------------
def _conv_bn(x: Tensor, p: Dict[str, Tensor], pre: str, pad: int) -> Tensor:
    """Conv2d_BN (model.py:17-25) in eval mode: depthwise conv without bias, then BatchNorm2d on running stats."""
    c = x.shape[1]
    y = F.conv2d(x, p[pre + "c.weight"], None, padding=pad, groups=c)
    return F.batch_norm(y, p[pre + "bn.running_mean"], p[pre + "bn.running_var"], p[pre + "bn.weight"], p[pre + "bn.bias"],
                        training=False, eps=1e-5)


def wfb_feed_forward(x: Tensor, p: Dict[str, Tensor], pre: str) -> Tensor:
    """FeedForward.forward (model.py:53-62), un-fused."""
    hid = F.conv2d(x, p[pre + "project_in.weight"], p.get(pre + "project_in.bias"))
    x1 = hid + _conv_bn(hid, p, pre + "rep_conv1.", 1) + _conv_bn(hid, p, pre + "rep_conv2.", 0)
    x2 = F.conv2d(hid, p[pre + "dwconv.weight"], p.get(pre + "dwconv.bias"), padding=1, groups=hid.shape[1])
    g = F.gelu(x2) * x1 + F.gelu(x1) * x2
    return F.conv2d(g, p[pre + "project_out.weight"], p.get(pre + "project_out.bias")) + x


def illumination_estimator(img: Tensor, p: Dict[str, Tensor], pre: str) -> Tuple[Tensor, Tensor]:
    """Illumination_Estimator.forward (model.py:186-200)."""
    inp = torch.cat([img, img.mean(dim=1, keepdim=True)], dim=1)
    x1 = F.conv2d(inp, p[pre + "conv1.weight"], p[pre + "conv1.bias"])
    fea = F.conv2d(x1, p[pre + "depth_conv.weight"], p[pre + "depth_conv.bias"], padding=2, groups=x1.shape[1])
    return fea, F.conv2d(fea, p[pre + "conv2.weight"], p[pre + "conv2.bias"])


def atten_transformer_block(x: Tensor, luma: Tensor, p: Dict[str, Tensor], pre: str, heads: int) -> Tensor:
    """Attenblock.TransformerBlock.forward(x, luma=luma) (Attenblock.py:225-236); LayerNorm = nn.LayerNorm over channels
    (Attenblock.py:37-45), ConvFFN = 1x1 -> dw3x3 -> GELU -> 1x1 (Attenblock.py:47-66)."""
    x = x + luminance_aware_mhsa(layernorm2d(x, p[pre + "norm1.body.weight"], p[pre + "norm1.body.bias"]), luma, p, pre + "attn.", heads)
    y = layernorm2d(x, p[pre + "norm2.body.weight"], p[pre + "norm2.body.bias"])
    return x + conv_ffn(y, p[pre + "ffn.pointwise1.weight"], p[pre + "ffn.pointwise1.bias"], p[pre + "ffn.depthwise.weight"],
                        p[pre + "ffn.depthwise.bias"], p[pre + "ffn.pointwise2.weight"], p[pre + "ffn.pointwise2.bias"])


def bayer_luma(mosaic: Tensor, pattern: str = "rggb") -> Tensor:
    """BayerLuma.forward (Attenblock.py:127-138) with the mask kernels of ``_create_kernel`` (:92-125)."""
    k = {c: torch.zeros(1, 1, 3, 3) for c in "rgb"}
    pos = {"rggb": ("r", "g", "g", "b"), "bggr": ("b", "g", "g", "r"), "grbg": ("g", "r", "b", "g"), "gbrg": ("g", "b", "r", "g")}[pattern.lower()]
    for (i, j), c in zip(((0, 0), (0, 1), (1, 0), (1, 1)), pos):
        k[c][0, 0, i, j] = 0.5 if c == "g" else 1.0
    rgb = torch.cat([F.conv2d(mosaic, k[c], padding=1) for c in "rgb"], dim=1)
    luma = torch.sum(rgb * torch.tensor([0.299, 0.587, 0.114]).view(1, 3, 1, 1), dim=1, keepdim=True)
    lo, hi = luma.amin(dim=(2, 3), keepdim=True), luma.amax(dim=(2, 3), keepdim=True)
    return (luma - lo) / (hi - lo + 1e-6)


# ------------------------------------------------------------------------------------------
# f2: FFAB / FEB (RawFomer_WFB_FFAB/blocks.py:11-92) and the Mamba-free part of WMB (model.py:203-245)
# ------------------------------------------------------------------------------------------
def feb(x: Tensor, p: Dict[str, Tensor], pre: str, exact_symmetric_bins: bool = False) -> Tensor:
    """FEB.forward (blocks.py:23-39): clamp, 1x1, rfft2 (ortho), |.|+1e-6 and angle through two 1x1 MLPs
    (LeakyReLU 0.1), clamp of the magnitude to [0, 1e4], polar -> cartesian, irfft2 (ortho), + clamped input, clamp.

    ``exact_symmetric_bins``: the four bins (0 | h/2, 0 | w/2) of a real 2-D transform are real by symmetry.  The reference's
    FFT (pocketfft) returns an exact +0 imaginary part there for power-of-two sizes, but rounding noise of either sign
    (~1e-8) for other sizes, so that ``angle`` of a negative real bin is +pi or -pi at random -- and the phase feeds a 1x1 MLP,
    which is not 2 pi periodic.  With the flag the imaginary parts are set to +0 (what exact arithmetic gives, and what the
    HIP transform does); for power-of-two sizes it changes nothing."""
    h, w = x.shape[-2:]
    x = x.clamp(-10.0, 10.0)
    f = torch.fft.rfft2(F.conv2d(x, p[pre + "fpre.weight"], p[pre + "fpre.bias"]), norm="ortho")
    if exact_symmetric_bins:
        f = f.clone()
        for yy in {0, h // 2} if h % 2 == 0 else {0}:
            for xx in {0, w // 2}:
                f[..., yy, xx] = torch.complex(f[..., yy, xx].real, torch.zeros_like(f[..., yy, xx].real))
    mag, pha = f.abs() + 1e-6, torch.angle(f)

    def mlp(t, name):
        t = F.leaky_relu(F.conv2d(t, p[pre + name + ".0.weight"], p[pre + name + ".0.bias"]), 0.1)
        return F.conv2d(t, p[pre + name + ".2.weight"], p[pre + name + ".2.bias"])

    mag = mlp(mag, "process1").clamp(0.0, 1e4)
    pha = mlp(pha, "process2")
    out = torch.fft.irfft2(torch.complex(mag * torch.cos(pha), mag * torch.sin(pha)), s=(h, w), norm="ortho")
    return (out + x).clamp(-10.0, 10.0)


def process_block(x: Tensor, p: Dict[str, Tensor], pre: str) -> Tensor:
    """ProcessBlock.forward (blocks.py:48-55): ``cat(FEB(x)) + x`` with ``cat`` a 1x1 conv."""
    return F.conv2d(feb(x, p, pre + "frequency_process."), p[pre + "cat.weight"], p[pre + "cat.bias"]) + x


def ffab(x: Tensor, p: Dict[str, Tensor], pre: str) -> Tensor:
    """FFAB.forward (blocks.py:83-92): seven ProcessBlocks with dense concatenations."""
    x = process_block(F.conv2d(x, p[pre + "conv0.0.weight"], p[pre + "conv0.0.bias"]), p, pre + "conv0.1.")
    x1 = process_block(x, p, pre + "conv1.")
    x2 = process_block(x1, p, pre + "conv2.")
    x3 = process_block(x2, p, pre + "conv3.")

    def tail(a, b, name):
        t = process_block(torch.cat((a, b), dim=1), p, pre + name + ".0.")
        return F.conv2d(t, p[pre + name + ".1.weight"], p[pre + name + ".1.bias"])

    x4 = tail(x2, x3, "conv4")
    x5 = tail(x1, x4, "conv5")
    return tail(x, x5, "convout")


def wmb_ll_branch(x: Tensor, p: Dict[str, Tensor], pre: str, high=None) -> Tensor:
    """WMB.forward (RawFomer_WFB_FFAB/model.py:215-245) up to and including the residual of the wavelet branch, with the
    Mamba module ``mb`` (absent offline: ``mamba_ssm``, parity unpinned) replaced by ``high`` (a callable on the
    ``[3B,C,h/2,w/2]`` high bands; identity when ``None``):
    ``t = 2 LN(x) - 1;  LL, high = DWT(t);  LL = FFAB(illu(LL));  out = t + clamp((IWT(cat(LL, high(high))) + 1) / 2, 0, 1)``."""
    n = x.shape[0]
    t = 2.0 * layernorm2d(x, p[pre + "norm1.body.weight"], p[pre + "norm1.body.bias"]) - 1.0
    d = dwt_init(t)
    ll, hi = d[:n], d[n:]
    ll, _ = illumination_estimator(ll, p, pre + "illu.")
    ll = ffab(ll, p, pre + "ffab.")
    if high is not None:
        hi = high(hi)
    return t + ((iwt_init(torch.cat((ll, hi), dim=0)) + 1.0) / 2.0).clamp(0.0, 1.0)


# ------------------------------------------------------------------------------------------
# f4: TrueColorRawFormer (BayerTORGBColorMultiLvl.py:387-462), the reference's latest whole model: the same U-Net with
# a learned Bayer front end, a colour-aware multi-level FLCA branch, exp(log_temperature) attention and a colour head.
# ------------------------------------------------------------------------------------------
def enhanced_bayer_processor(x4: Tensor, p: Dict[str, Tensor], pre: str, eps: float = 1e-6):
    """EnhancedBayerProcessor.forward (BayerTORGBColorMultiLvl.py:100-134) -> y, cr, cb, refined_rgb."""
    wb = x4 * (F.softplus(p[pre + "wb_gains"]) + 1e-6).view(1, 4, 1, 1)
    r, g, b = wb[:, 0:1], 0.5 * (wb[:, 1:2] + wb[:, 2:3]), wb[:, 3:4]
    rgb = torch.cat([r, g, b], dim=1)
    m = p[pre + "color_matrix"]
    lin = torch.einsum("bchw,oc->bohw", rgb, m[:, :3]) + m[:, 3].view(1, 3, 1, 1)
    y = (lin * torch.tensor([0.2126, 0.7152, 0.0722]).view(1, 3, 1, 1)).sum(dim=1, keepdim=True)
    y = y / y.amax(dim=(2, 3), keepdim=True).clamp_min(eps)
    c = F.relu(F.conv2d(torch.cat([r, g, b, y], dim=1), p[pre + "chroma_extractor.0.weight"], p[pre + "chroma_extractor.0.bias"], padding=1))
    c = torch.tanh(F.conv2d(c, p[pre + "chroma_extractor.2.weight"], p[pre + "chroma_extractor.2.bias"], padding=1))
    d = F.gelu(F.conv2d(lin, p[pre + "demosaic_refine.0.weight"], p[pre + "demosaic_refine.0.bias"], padding=1))
    d = F.conv2d(d, p[pre + "demosaic_refine.2.weight"], p[pre + "demosaic_refine.2.bias"], padding=1)
    return y, c[:, 0:1], c[:, 1:2], lin + d


def enhanced_flca_guidance(y: Tensor, cr: Tensor, cb: Tensor, rgb: Tensor, size: Tuple[int, int], levels: int = 2, eps: float = 1e-8) -> Tensor:
    """Guidance planes of EnhancedFLCA at feature size ``size`` (BayerTORGBColorMultiLvl.py:236-275):
    ``[y, cr, cb, R, G, y_low, y_high]`` -- y_low = LL of the deepest pyramid level, y_high = mean of the resized
    high-band magnitudes of all levels."""
    highs, cur = [], y
    for _ in range(levels):
        ll, (lh, hl, hh) = haar_dwt(cur)
        highs.append(torch.sqrt(lh * lh + hl * hl + hh * hh + eps))
        cur = ll
    hf = [bilinear_resize(t, size) for t in highs]
    y_high = torch.stack(hf, dim=0).mean(dim=0) if len(hf) > 1 else hf[0]
    rs = bilinear_resize(rgb, size)
    return torch.cat([bilinear_resize(y, size), bilinear_resize(cr, size), bilinear_resize(cb, size), rs[:, 0:1], rs[:, 1:2],
                      bilinear_resize(cur, size), y_high], dim=1)


def enhanced_flca(feat: Tensor, guide, p: Dict[str, Tensor], pre: str, levels: int = 2) -> Tensor:
    """EnhancedFLCA.forward (BayerTORGBColorMultiLvl.py:249-293)."""
    g = enhanced_flca_guidance(*guide, feat.shape[-2:], levels)
    color = torch.sigmoid(F.conv2d(g[:, 0:5], p[pre + "color_attention.0.weight"], p[pre + "color_attention.0.bias"], padding=1))
    low = torch.sigmoid(F.conv2d(g[:, 5:6], p[pre + "low_attn.0.weight"], p[pre + "low_attn.0.bias"], padding=1))
    high = torch.tanh(F.conv2d(g[:, 6:7], p[pre + "high_attn.0.weight"], p[pre + "high_attn.0.bias"], padding=1))
    x = feat * (1.0 + color + torch.tanh(low + high))
    r = F.conv2d(F.relu(F.conv2d(x, p[pre + "res_proj.0.weight"], p[pre + "res_proj.0.bias"])), p[pre + "res_proj.2.weight"], p[pre + "res_proj.2.bias"])
    x = x + torch.tanh(r) * 0.2
    hid = torch.relu(F.conv2d(x.mean(dim=(2, 3), keepdim=True), p[pre + "se.1.weight"], p[pre + "se.1.bias"]))
    return x * torch.sigmoid(F.conv2d(hid, p[pre + "se.3.weight"], p[pre + "se.3.bias"]))


def camera_color_correction(x: Tensor, p: Dict[str, Tensor], pre: str) -> Tensor:
    """CameraAwareColorCorrection.forward (BayerTORGBColorMultiLvl.py:160-176)."""
    gamma = F.softplus(p[pre + "gamma_param"]) + 1e-6
    x = torch.pow(x.clamp(0.0, 1.0), 1.0 / gamma)
    x = F.conv2d(F.relu(F.conv2d(x, p[pre + "color_transform.0.weight"], p[pre + "color_transform.0.bias"])),
                 p[pre + "color_transform.2.weight"], p[pre + "color_transform.2.bias"])
    out = []
    for i in range(x.shape[1]):
        ch = x[:, i:i + 1]
        mod = torch.sigmoid(F.conv2d(F.relu(F.conv2d(ch, p[pre + "tone_curve.0.weight"], p[pre + "tone_curve.0.bias"])),
                                     p[pre + "tone_curve.2.weight"], p[pre + "tone_curve.2.bias"]))
        out.append((ch * (0.8 + 0.4 * mod)).clamp(0.0, 1.0))
    return torch.cat(out, dim=1).clamp(0.0, 1.0)


def truecolor_stage(x: Tensor, guide, p: Dict[str, Tensor], pre: str, heads: int, levels: int = 2) -> Tensor:
    """EnhancedConv_Transformer.forward (BayerTORGBColorMultiLvl.py:371-377)."""
    tp = dict(p)
    tp[pre + "Transformer.attn.temperature"] = p[pre + "Transformer.attn.log_temperature"].exp()     # Attention, :344
    t = F.conv2d(torch.cat([enhanced_flca(x, guide, p, pre + "FLCA.", levels), transformer_block(x, tp, pre + "Transformer.", heads)], dim=1),
                 p[pre + "channel_reduce.weight"], p[pre + "channel_reduce.bias"])
    return F.leaky_relu(F.conv2d(t, p[pre + "Conv_out.weight"], p[pre + "Conv_out.bias"], padding=1), 0.2)


def truecolor_forward(p: Dict[str, Tensor], x: Tensor, dim: int, heads=(8, 8, 8, 8), levels: int = 2) -> Tensor:
    """TrueColorRawFormer.forward (BayerTORGBColorMultiLvl.py:421-462); mosaic sizes divisible by 16 (no reflect padding)."""
    x4 = pixel_unshuffle2(x)
    guide = enhanced_bayer_processor(x4, p, "bayer_processor.")
    t = F.conv2d(x4, p["embedding.weight"], p["embedding.bias"], padding=1)
    e1 = truecolor_stage(t, guide, p, "conv_tran1.", heads[0], levels)
    e2 = truecolor_stage(downsample(e1, p["down1.body.0.weight"]), guide, p, "conv_tran2.", heads[1], levels)
    e3 = truecolor_stage(downsample(e2, p["down2.body.0.weight"]), guide, p, "conv_tran3.", heads[2], levels)
    e4 = truecolor_stage(downsample(e3, p["down3.body.0.weight"]), guide, p, "conv_tran4.", heads[3], levels)

    def up(t_in, skip, i):
        u = conv_transpose2x2(t_in, p[f"up{i}.weight"], p[f"up{i}.bias"])
        return F.conv2d(torch.cat([u, skip], dim=1), p[f"channel_reduce{i}.weight"], p[f"channel_reduce{i}.bias"])

    d3 = truecolor_stage(up(e4, e3, 1), guide, p, "conv_tran5.", heads[2], levels)
    d2 = truecolor_stage(up(d3, e2, 2), guide, p, "conv_tran6.", heads[1], levels)
    d1 = truecolor_stage(up(d2, e1, 3), guide, p, "conv_tran7.", heads[0], levels)
    out = pixel_shuffle2(F.relu(F.conv2d(d1, p["conv_out.weight"], p["conv_out.bias"], padding=1)))
    return camera_color_correction(out, p, "color_correction.")
