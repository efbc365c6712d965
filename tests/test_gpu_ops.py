"""GPU parity, operator level: every C-ABI operator (through ctypes -> librawformer_hip.so)
against (i) the reference's outputs in tests/golden/per_op.npz and (ii) the CPU oracle on
extra shapes (ragged widths, tiny and large tiles).

Tolerances (float32 path; the f32 MFMA is an exact fmaf chain, so differences are
reassociation only):  data-movement ops and dwt_init/iwt_init bit-exact; everything else
max-abs <= 2e-5 on O(1) activations (observed ~1e-6).
"""
import numpy as np
import pytest
import torch

import cases
from cases import golden, rnd, params
from oracle import rawformer_ref as R

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    from bayer_low_light_image_enhancement_amd import ops as o
    return o


@pytest.fixture(scope="module")
def g():
    return golden("per_op")


def close(a, b, tol=TOL):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, f"shape {a.shape} vs {b.shape}"
    err = float(np.max(np.abs(a - b))) if a.size else 0.0
    assert err <= tol, f"max-abs {err:.3e} > {tol}"


def exact(a, b):
    b = torch.from_numpy(np.asarray(b)) if not torch.is_tensor(b) else b
    assert torch.equal(a.cpu(), b.cpu())


def dev(d, device):
    return {k: v.to(device) for k, v in d.items()}


def test_library_is_the_hip_one(ops, device):
    import ctypes
    from bayer_low_light_image_enhancement_amd import _lib
    assert isinstance(_lib.load(), ctypes.CDLL) and _lib.load().rf_version() >= 1


def test_cpu_tensor_fails_loudly(ops):
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        ops.dwt_init(torch.zeros(1, 1, 4, 4))


def test_pixel_shuffles(ops, g, device):
    exact(ops.downshuffle(rnd("x.shuffle", (2, 3, 8, 12)).to(device)), g["downshuffle"])
    exact(ops.pixel_shuffle(rnd("x.pixelshuffle", (2, 12, 6, 10)).to(device)), g["pixelshuffle"])
    for shp in ((1, 1, 32, 64), (2, 5, 6, 10), (1, 2, 512, 512)):     # vector and scalar paths
        x = rnd("x.ps", shp)
        exact(ops.downshuffle(x.to(device)), R.pixel_unshuffle2(x))
        exact(ops.pixel_shuffle(ops.downshuffle(x.to(device))), x)


def test_dwt_haar_bit_exact(ops, g, device):
    exact(ops.dwt_init(rnd("x.dwt", (2, 5, 12, 20)).to(device)), g["dwt_init"])
    exact(ops.iwt_init(rnd("x.iwt", (8, 5, 6, 10)).to(device)), g["iwt_init"])
    for shp in ((1, 3, 16, 16), (2, 32, 64, 128), (1, 2, 10, 14)):
        x = rnd("x.dwt2", shp)
        d = ops.dwt_init(x.to(device))
        exact(d, R.dwt_init(x))
        exact(ops.iwt_init(d), R.iwt_init(R.dwt_init(x)))
        close(ops.iwt_init(d), x, 1e-6)                                # round trip


def test_dwt_custom(ops, g, device):
    xd = rnd("x.dwt", (2, 5, 12, 20)).to(device)
    k2 = g["custom_kernel_rand"].tolist()
    y = ops.custom_dwt(xd)
    close(y, g["custom_dwt_default"], 1e-6)
    close(ops.custom_idwt(y), g["custom_idwt_default"], 1e-6)
    y2 = ops.custom_dwt(xd, k2, norm=False)
    close(y2, g["custom_dwt_rand_nonorm"], 1e-6)
    close(ops.custom_idwt(y2, k2, norm=False), g["custom_idwt_rand_nonorm"], 2e-6)
    x = rnd("x.cdwt", (2, 8, 64, 96))
    close(ops.custom_dwt(x.to(device), k2, True), R.custom_dwt(x, k2, True), 1e-6)
    close(ops.custom_idwt(x.to(device), k2, True), R.custom_idwt(x, k2, True), 1e-6)


def test_haar_dwt(ops, g, device):
    for tag, shp in (("even", (2, 3, 12, 20)), ("odd", (2, 3, 17, 19))):
        ll, (lh, hl, hh) = ops.haar_dwt(rnd("x.haar." + tag, shp).to(device))
        close(torch.stack([ll, lh, hl, hh]), g[f"haar_{tag}"], 1e-6)
    x = rnd("x.haar.big", (1, 2, 64, 128))
    ll, (lh, hl, hh) = ops.haar_dwt(x.to(device))
    oll, (olh, ohl, ohh) = R.haar_dwt(x)
    close(torch.stack([ll, lh, hl, hh]), torch.stack([oll, olh, ohl, ohh]), 1e-6)


def test_layernorm(ops, g, device):
    for c, hw in cases.LN_CASES:
        p = dev(params({"body.weight": (c,), "body.bias": (c,)}), device)
        close(ops.layernorm2d(rnd(f"x.ln{c}", (2, c) + hw, -2, 3).to(device), p["body.weight"], p["body.bias"]),
              g[f"layernorm_c{c}"])
    xl = rnd("x.ln.wfb", (2, 60, 32), -2, 3)
    x4 = xl.reshape(2, 6, 10, 32).permute(0, 3, 1, 2).contiguous().to(device)
    p = dev(params({"weight": (32,), "bias": (32,)}), device)
    close(ops.layernorm2d(x4, p["weight"], p["bias"]).permute(0, 2, 3, 1).reshape(2, 60, 32), g["wfb_withbias_ln"])
    close(ops.layernorm2d(x4, p["weight"], None).permute(0, 2, 3, 1).reshape(2, 60, 32), g["wfb_biasfree_ln"])
    x = rnd("x.ln.odd", (1, 24, 5, 7), -3, 3)    # P = 35: scalar path
    w, b = rnd("ln.w", (24,), 0.5, 1.5), rnd("ln.b", (24,))
    close(ops.layernorm2d(x.to(device), w.to(device), b.to(device)), R.layernorm2d(x, w, b))


@pytest.mark.parametrize("c1,c2,cout,hw", [(32, 0, 96, (16, 16)), (64, 0, 32, (8, 24)), (16, 16, 16, (16, 8)),
                                           (48, 48, 48, (8, 8)), (256, 0, 768, (4, 4)), (24, 0, 80, (5, 7)),
                                           (128, 128, 128, (16, 16)), (32, 0, 64, (64, 64)),
                                           # LayerNorm + 1x1 on conv1x1_b3_ln_kernel: K = 64 (a K block split over two waves), 128, 256;
                                           # 2 / 3 / 4 output tiles per chunk; ragged last pixel tile; persistent workgroups (320 tiles)
                                           (64, 0, 192, (24, 40)), (64, 0, 128, (9, 20)), (128, 0, 384, (160, 128)),
                                           (128, 0, 256, (12, 20)), (256, 0, 512, (40, 48))])
def test_conv1x1(ops, device, c1, c2, cout, hw):
    import torch.nn.functional as F
    x = rnd("c1.x", (2, c1) + hw)
    x2 = rnd("c1.x2", (2, c2) + hw) if c2 else None
    k = c1 + c2
    w = rnd("c1.w", (cout, k, 1, 1), -1, 1) / np.sqrt(k)
    b = rnd("c1.b", (cout,))
    res = rnd("c1.res", (2, cout) + hw)
    xin = x if x2 is None else torch.cat([x, x2], 1)
    ref = F.conv2d(xin, w, b)
    close(ops.conv1x1(x.to(device), w.to(device), b.to(device), x2=None if x2 is None else x2.to(device)), ref)
    close(ops.conv1x1(x.to(device), w.to(device), None, x2=None if x2 is None else x2.to(device), residual=res.to(device)),
          F.conv2d(xin, w) + res)
    if c2 == 0:
        lw, lb = rnd("c1.lw", (c1,), 0.5, 1.5), rnd("c1.lb", (c1,))
        xs = x * 2.0 + 0.7
        close(ops.conv1x1(xs.to(device), w.to(device), b.to(device), ln_weight=lw.to(device), ln_bias=lb.to(device)),
              F.conv2d(R.layernorm2d(xs, lw, lb), w, b))
        close(ops.conv1x1(xs.to(device), w.to(device), b.to(device), ln_weight=lw.to(device)),
              F.conv2d(R.layernorm2d(xs, lw, None), w, b))


@pytest.mark.parametrize("c,hw", [(96, (16, 16)), (7, (5, 9)), (64, (64, 64)), (3, (33, 130))])
def test_dwconv3x3(ops, device, c, hw):
    import torch.nn.functional as F
    x = rnd("dw.x", (2, c) + hw)
    w, b = rnd("dw.w", (c, 1, 3, 3)) / 3, rnd("dw.b", (c,))
    ref = F.conv2d(x, w, b, padding=1, groups=c)
    close(ops.dwconv3x3(x.to(device), w.to(device), b.to(device)), ref)
    close(ops.dwconv3x3(x.to(device), w.to(device), b.to(device), gelu=True), F.gelu(ref))


@pytest.mark.parametrize("cin,cout,hw", [(4, 32, (16, 16)), (32, 32, (16, 24)), (32, 16, (64, 64)), (64, 32, (32, 32)),
                                         (16, 12, (16, 16)), (48, 24, (8, 8)), (128, 128, (8, 16)), (8, 8, (5, 7)),
                                         (256, 256, (4, 4)), (24, 40, (10, 18)), (32, 32, (72, 136))])
def test_conv3x3(ops, device, cin, cout, hw):
    import torch.nn.functional as F
    x = rnd("c3.x", (2, cin) + hw)
    w = rnd("c3.w", (cout, cin, 3, 3)) / np.sqrt(cin * 9 / 3)
    b = rnd("c3.b", (cout,))
    ref = F.conv2d(x, w, b, padding=1)
    close(ops.conv3x3(x.to(device), w.to(device), b.to(device)), ref)
    close(ops.conv3x3(x.to(device), w.to(device), None, act="lrelu"), F.leaky_relu(F.conv2d(x, w, padding=1), 0.2))
    if hw[0] % 2 == 0 and hw[1] % 2 == 0:
        close(ops.conv3x3(x.to(device), w.to(device), None, store="unshuffle"), R.downsample(x, w))
    if cout % 4 == 0:
        close(ops.conv3x3(x.to(device), w.to(device), b.to(device), act="lrelu", store="shuffle"),
              R.pixel_shuffle2(F.leaky_relu(ref, 0.2)))


def test_downsample_and_convtranspose_golden(ops, g, device):
    c = 32
    x = rnd("x.root", (2, c, 16, 16)).to(device)
    p = dev(params({"body.0.weight": (c // 2, c, 3, 3)}), device)
    close(ops.conv3x3(x, p["body.0.weight"], None, store="unshuffle"), g["downsample"])
    p = dev(params({"net.0.weight": (c // 2, c, 3, 3), "net.0.bias": (c // 2,)}), device)
    close(ops.conv3x3(x, p["net.0.weight"], p["net.0.bias"], store="unshuffle"), g["root_downsample"])
    p = dev(params({"up1.weight": (c, c // 2, 2, 2), "up1.bias": (c // 2,)}), device)
    close(ops.conv_transpose2x2(x, p["up1.weight"], p["up1.bias"]), g["convtranspose"])


@pytest.mark.parametrize("cin,cout,hw", [(64, 32, (8, 8)), (16, 8, (5, 7)), (256, 128, (4, 6)), (32, 16, (32, 64))])
def test_conv_transpose(ops, device, cin, cout, hw):
    x = rnd("ct.x", (2, cin) + hw)
    w, b = rnd("ct.w", (cin, cout, 2, 2)) / np.sqrt(cin), rnd("ct.b", (cout,))
    close(ops.conv_transpose2x2(x.to(device), w.to(device), b.to(device)), R.conv_transpose2x2(x, w, b))


def test_channel_attention_golden(ops, g, device):
    for c, hw in cases.ATTN_CASES:
        x = rnd(f"x.attn{c}", (2, c) + hw).to(device)
        p = dev(params(cases.attention_spec(c)), device)
        out = ops.channel_attention(x, p["qkv.weight"], p["qkv.bias"], p["qkv_dwconv.weight"], p["qkv_dwconv.bias"],
                                    p["temperature"], p["project_out.weight"], p["project_out.bias"], 8)
        close(out, g[f"attention_c{c}"])
    c = 32   # root model.py flavour: scale [1,8,1,1]
    p = dev(params({"scale": (1, 8, 1, 1), "qkv.0.weight": (3 * c, c, 1, 1), "qkv.0.bias": (3 * c,),
                    "qkv.1.weight": (3 * c, 1, 3, 3), "qkv.1.bias": (3 * c,), "proj.weight": (c, c, 1, 1),
                    "proj.bias": (c,)}), device)
    out = ops.channel_attention(rnd("x.root", (2, c, 16, 16)).to(device), p["qkv.0.weight"], p["qkv.0.bias"], p["qkv.1.weight"],
                                p["qkv.1.bias"], p["scale"], p["proj.weight"], p["proj.bias"], 8)
    close(out, g["root_attention"])


@pytest.mark.parametrize("c,heads,hw", [(32, 8, (64, 64)), (64, 8, (24, 40)), (128, 8, (16, 16)), (256, 8, (8, 8)),
                                        (48, 8, (16, 16)), (96, 8, (10, 14)), (192, 8, (8, 8)), (384, 8, (4, 4)),
                                        (512, 8, (4, 4)), (32, 4, (5, 7)), (16, 1, (16, 16))])
def test_channel_attention_oracle(ops, device, c, heads, hw):
    x = rnd("ca.x", (2, c) + hw)
    p = params({"temperature": (heads, 1, 1), "qkv.weight": (3 * c, c, 1, 1), "qkv.bias": (3 * c,),
                "qkv_dwconv.weight": (3 * c, 1, 3, 3), "qkv_dwconv.bias": (3 * c,),
                "project_out.weight": (c, c, 1, 1), "project_out.bias": (c,)})
    ref = R.channel_attention(x, p["qkv.weight"], p["qkv.bias"], p["qkv_dwconv.weight"], p["qkv_dwconv.bias"],
                              p["temperature"], p["project_out.weight"], p["project_out.bias"], heads)
    d = dev(p, device)
    out = ops.channel_attention(x.to(device), d["qkv.weight"], d["qkv.bias"], d["qkv_dwconv.weight"], d["qkv_dwconv.bias"],
                                d["temperature"], d["project_out.weight"], d["project_out.bias"], heads)
    close(out, ref)


def test_flca_guidance(ops, device):
    x4 = rnd("x.packed", (2, 4, 32, 48), 0, 1)
    y, cr, cb = R.bayer_luma_chroma(x4)
    for size in ((32, 48), (16, 24), (8, 12), (4, 6)):
        close(ops.flca_guidance(x4.to(device), size), R.flca_guidance(y, cr, cb, size), 2e-6)


@pytest.mark.parametrize("c,hw", cases.ATTN_CASES)
def test_transformer_block_golden(ops, g, device, c, hw):
    """a7: the reference's TransformerBlock outputs; c = 32 runs the fused attention-front / FFN kernels."""
    p = dev(params(cases.transformer_spec(c)), device)
    out = ops.transformer_block(rnd(f"x.attn{c}", (2, c) + hw).to(device), p, heads=8)
    close(out, g[f"transformer_c{c}"])


@pytest.mark.parametrize("c,heads,hw", [(32, 8, (64, 64)), (32, 8, (20, 72)), (32, 4, (8, 132)), (64, 8, (32, 32)), (32, 2, (12, 8))])
def test_transformer_block_oracle(ops, device, c, heads, hw):
    """Fused kernels on partial tiles (heights / widths that are not multiples of the 4x64 tile)."""
    spec = cases.transformer_spec(c)
    spec["attn.temperature"] = (heads, 1, 1)
    p = params(spec)
    x = rnd("tb.x", (2, c) + hw)
    ref = R.transformer_block(x, p, "", heads)
    close(ops.transformer_block(x.to(device), dev(p, device), heads=heads), ref)


def test_flca_golden(ops, g, device):
    x4 = rnd("x.packed", (2, 4, 32, 48), 0, 1).to(device)
    for c, hw in cases.FLCA_CASES:
        p = dev(params(cases.flca_spec(c)), device)
        out = ops.flca(rnd(f"x.flca{c}", (2, c) + hw).to(device), x4, p)
        close(out, g[f"flca_c{c}"])


@pytest.mark.parametrize("c,hw", [(32, (10, 266)), (16, (7, 34))])
def test_flca_even_width_not_multiple_of_four(ops, device, c, hw):
    """w % 4 == 2 (level 3 of a 2848 x 4256 frame is 266 wide): the two-pixel vector path of the spatial gate and its pooled sums."""
    x4 = rnd("x.packed.w2", (1, 4, 2 * hw[0], 2 * hw[1]), 0, 1)
    feat = rnd(f"x.flca.w2.{c}", (1, c) + hw)
    p = params(cases.flca_spec(c))
    ref = R.flca(feat, *R.bayer_luma_chroma(x4), p, "")
    close(ops.flca(feat.to(device), x4.to(device), dev(p, device)), ref, 2e-5)


@pytest.mark.gpu
def test_upsample_cat_reduce_matches_two_step_reference(device):
    """Decoder step on composed weights vs ConvTranspose2d -> cat -> Conv2d (model.py:494-503) in torch fp32 on the CPU."""
    from bayer_low_light_image_enhancement_amd import ops
    for c, b, h, w in ((32, 2, 16, 24), (48, 1, 8, 12), (128, 1, 8, 8), (24, 1, 5, 4)):
        x = rnd(f"upcat.x{c}", (b, 2 * c, h, w))
        skip = rnd(f"upcat.s{c}", (b, c, 2 * h, 2 * w))
        p = params({"up.weight": (2 * c, c, 2, 2), "up.bias": (c,), "cr.weight": (c, 2 * c, 1, 1), "cr.bias": (c,)})
        up = torch.nn.functional.conv_transpose2d(x, p["up.weight"], p["up.bias"], stride=2)
        ref = torch.nn.functional.conv2d(torch.cat([up, skip], 1), p["cr.weight"], p["cr.bias"])
        got = ops.upsample_cat_reduce(x.to(device), skip.to(device), *(p[k].to(device) for k in ("up.weight", "up.bias", "cr.weight", "cr.bias")))
        assert float((got.cpu() - ref).abs().max()) < 2e-5, c


# ---- round-2 additions: goldens that only the CPU oracle used to see, operator-level cover of attn_mid<128>, the
# ---- diagnostic twin library (fused vs op-by-op), rejected head layouts
def test_conv_ffn_golden(ops, g, device):
    """a6: the reference's conv_ffn outputs through the three kernels the forward launches at levels 1-3."""
    for c, hw in cases.ATTN_CASES:
        p = dev(params(cases.ffn_spec(c)), device)
        out = ops.conv_ffn(rnd(f"x.attn{c}", (2, c) + hw).to(device), p["pointwise1.weight"], p["pointwise1.bias"],
                           p["depthwise.weight"], p["depthwise.bias"], p["pointwise2.weight"], p["pointwise2.bias"])
        close(out, g[f"conv_ffn_c{c}"])


def test_luma_chroma_golden(ops, g, device):
    """a15: BayerLumaChroma (y normalised by its per-image maximum, cr, cb) against the reference's module."""
    y, cr, cb = ops.bayer_luma_chroma(rnd("x.packed", (2, 4, 32, 48), 0, 1).to(device))
    close(torch.cat([y, cr, cb], 1), g["luma_chroma"], 1e-6)


@pytest.mark.parametrize("c,heads,hw", [(128, 8, (20, 136)), (128, 8, (8, 8)), (64, 8, (36, 72)), (64, 4, (6, 132))])
def test_transformer_block_oracle_mid_levels(ops, device, c, heads, hw):
    """attn_mid_kernel<64|128> on multi-tile shapes that are not multiples of the 4x64 tile (levels 1-2 of the forward)."""
    spec = cases.transformer_spec(c)
    spec["attn.temperature"] = (heads, 1, 1)
    p = params(spec)
    x = rnd("tb.mid.x", (1, c) + hw)
    ref = R.transformer_block(x, p, "", heads)
    close(ops.transformer_block(x.to(device), dev(p, device), heads=heads), ref)


@pytest.mark.parametrize("shape", [(2, 32, 256, 512), (8, 32, 128, 256)])
def test_transformer_block_persistent_workgroups(ops, device, shape):
    """Level-0 fused kernels with SEVERAL tiles per persistent workgroup and both co-resident workgroups of every CU busy
    (512 workgroups): the small cases above give each workgroup at most one tile of the fused FFN."""
    c, heads = 32, 8
    p = params(cases.transformer_spec(c))
    x = rnd("tb.big.x", shape)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    ref = R.transformer_block(x, p, "", heads)
    close(ops.transformer_block(x.to(device), dev(p, device), heads=heads), ref)


def test_transformer_block_ffn_expansion_4(ops, device):
    """expansion 4 takes the op-by-op FFN whose hidden tensor is 4C wide (scratch sized for it)."""
    c, heads, hw = 32, 8, (16, 24)
    spec = cases.transformer_spec(c)
    spec.update({"ffn.pointwise1.weight": (4 * c, c, 1, 1), "ffn.pointwise1.bias": (4 * c,), "ffn.depthwise.weight": (4 * c, 1, 3, 3),
                 "ffn.depthwise.bias": (4 * c,), "ffn.pointwise2.weight": (c, 4 * c, 1, 1)})
    p = params(spec)
    x = rnd("tb.e4.x", (2, c) + hw)
    close(ops.transformer_block(x.to(device), dev(p, device), heads=heads, ffn_expansion_factor=4), R.transformer_block(x, p, "", heads))


def test_unsupported_head_layout_is_an_error_not_a_wrong_answer(ops, device):
    c, heads = 80, 2          # head size 40: the query tile of channels 32..47 straddles both heads = 5 key tiles
    p = dev(params(cases.attention_spec(c)), device)
    with pytest.raises(RuntimeError, match="key tiles"):
        ops.channel_attention(rnd("x.bad", (1, c, 8, 8)).to(device), p["qkv.weight"], p["qkv.bias"], p["qkv_dwconv.weight"],
                              p["qkv_dwconv.bias"], p["temperature"][:2].contiguous(), p["project_out.weight"], p["project_out.bias"], heads)


@pytest.mark.gpu
def test_eight_wave_ffn_at_64_channels_through_the_diagnostic_twin(device):
    """ffn_fused8_kernel<64> is built but not dispatched by the shipped library (it ties the op-by-op chain); the diagnostic twin
    selects it with RF_FFN8_64=1: ragged tiles and several tiles per workgroup against the oracle."""
    import os
    import subprocess
    import sys
    from bayer_low_light_image_enhancement_amd import build
    diag = build.build_diag_library()
    code = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import cases
from cases import rnd, params
from bayer_low_light_image_enhancement_amd import ops
from oracle import rawformer_ref as R
dev = torch.device("cuda:0")
for heads, shape in ((8, (1, 64, 36, 72)), (4, (2, 64, 6, 132)), (8, (2, 64, 128, 256))):
    spec = cases.transformer_spec(64); spec["attn.temperature"] = (heads, 1, 1)
    p = params(spec)
    x = rnd("tb.ffn8.x", shape)
    ref = R.transformer_block(x, p, "", heads)
    out = ops.transformer_block(x.to(dev), {k: v.to(dev) for k, v in p.items()}, heads=heads).cpu()
    err = float((out - ref).abs().max())
    print(shape, err)
    assert err <= 2e-5, (shape, err)
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RF_LIB_PATH=diag, RF_FFN8_64="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_fused_kernels_agree_with_the_op_by_op_schedule(device, tmp_path):
    """The diagnostic twin library (build.py --diag: -DRF_DIAG adds the RF_NO_FUSE / RF_NO_UPCAT switches the shipped
    library does not have) runs the same block op by op; both schedules must agree to reassociation error."""
    import os
    import subprocess
    import sys
    from bayer_low_light_image_enhancement_amd import build
    diag = build.build_diag_library()
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import cases
from cases import rnd, params
from bayer_low_light_image_enhancement_amd import ops, RawFormer, synth
dev = torch.device("cuda:0")
outs = {}
for c, hw in ((32, (20, 72)), (64, (12, 68)), (128, (8, 64))):
    p = {k: v.to(dev) for k, v in params(cases.transformer_spec(c)).items()}
    outs[f"tb{c}"] = ops.transformer_block(rnd("tb.x", (2, c) + hw).to(dev), p, heads=8).cpu().numpy()
m = RawFormer(dim=16)
m.load_state_dict({**m.state_dict(), **cases.model_state(16, 21)}, strict=True)
m = m.to(dev).eval()
with torch.no_grad():
    outs["model"] = m(torch.from_numpy(synth.bayer_mosaic(21, 2, 64, 64)).to(dev)).cpu().numpy()
np.savez(sys.argv[2], **outs)
'''
    res = {}
    for tag, env in (("fused", {}), ("plain", {"RF_LIB_PATH": diag, "RF_NO_FUSE": "1", "RF_NO_UPCAT": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        e = dict(os.environ)
        e.pop("RF_LIB_PATH", None)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code, cases.REPO, out], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[tag] = np.load(out)
    for k in res["fused"].files:
        close(res["fused"][k], res["plain"][k], 1e-5)
        assert not np.array_equal(res["fused"][k], res["plain"][k]) or k == "model", f"{k}: the switch changed nothing"


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["flca", "plain"])
def test_composed_stage_tail_agrees_with_the_two_gemm_form(device, tmp_path, variant):
    """run_stage composes pointwise2 into channel_reduce (one bf16x3 GEMM over [branch ; x1 ; hidden] with [Wa' | Wb | Wb W2]) wherever
    the FFN runs op by op; the diagnostic twin's RF_NO_COMPOSE=1 runs the two GEMMs.  dim 32 on a 256 x 128 mosaic: levels 1-3
    (C = 64 / 128 / 256) are composed, level 0 runs the fused FFN kernel.  Both forms against the oracle, and against each other."""
    import os
    import subprocess
    import sys
    from bayer_low_light_image_enhancement_amd import build, synth
    diag = build.build_diag_library()
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import cases
from bayer_low_light_image_enhancement_amd import RawFormer, synth
dev = torch.device("cuda:0")
kw = dict(variant=sys.argv[3]) if sys.argv[3] == "flca" else dict(variant="plain", branch_lrelu=True)
m = RawFormer(dim=32, **kw)
m.load_state_dict({**m.state_dict(), **cases.model_state(32, 77, sys.argv[3])}, strict=True)
m = m.to(dev).eval()
with torch.no_grad():
    y = m(torch.from_numpy(synth.bayer_mosaic(77, 2, 256, 128)).to(dev)).cpu().numpy()
np.save(sys.argv[2], y)
'''
    res = {}
    for tag, env in (("composed", {"RF_LIB_PATH": diag}), ("two", {"RF_LIB_PATH": diag, "RF_NO_COMPOSE": "1"})):
        out = str(tmp_path / f"{tag}.npy")
        r = subprocess.run([sys.executable, "-c", code, cases.REPO, out, variant], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[tag] = np.load(out)
    sd = cases.model_state(32, 77, variant)
    cfg = R.RawFormerConfig(dim=32, variant=variant, branch_lrelu=True)
    with torch.no_grad():
        ref = R.rawformer_forward(sd, torch.from_numpy(synth.bayer_mosaic(77, 2, 256, 128)), cfg).numpy()
    close(res["composed"], ref, 2e-5)
    close(res["two"], ref, 2e-5)
    close(res["composed"], res["two"], 1e-5)
    assert not np.array_equal(res["composed"], res["two"]), "the switch changed nothing"
