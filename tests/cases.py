"""Shared description of the golden per-operator cases.

``oracle/make_golden.py`` ran the REFERENCE's classes on these inputs/parameters (all
regenerated from ``synth`` by name) and stored the reference outputs in
``tests/golden/per_op.npz``.  The CPU tests check the oracle against those outputs; the GPU
tests check the HIP path against both.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from bayer_low_light_image_enhancement_amd import synth  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")
SEED = 11


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rnd(name, shape, lo=-1.0, hi=1.0, seed=SEED):
    return torch.from_numpy(synth.uniform(seed, name, shape, lo, hi))


def params(spec, seed=SEED, prefix=""):
    """spec: {name: shape}; values exactly as ``synth.fill_state_dict`` gave the reference module."""
    return {prefix + k: torch.from_numpy(synth.param_values(seed, k, s)).reshape(s) for k, s in spec.items()}


def attention_spec(c):
    return {"temperature": (8, 1, 1), "qkv.weight": (3 * c, c, 1, 1), "qkv.bias": (3 * c,),
            "qkv_dwconv.weight": (3 * c, 1, 3, 3), "qkv_dwconv.bias": (3 * c,),
            "project_out.weight": (c, c, 1, 1), "project_out.bias": (c,)}


def ffn_spec(c):
    return {"pointwise1.weight": (2 * c, c, 1, 1), "pointwise1.bias": (2 * c,),
            "depthwise.weight": (2 * c, 1, 3, 3), "depthwise.bias": (2 * c,),
            "pointwise2.weight": (c, 2 * c, 1, 1), "pointwise2.bias": (c,)}


def transformer_spec(c):
    s = {"norm1.body.weight": (c,), "norm1.body.bias": (c,)}
    s.update({"attn." + k: v for k, v in attention_spec(c).items()})
    s.update({"norm2.body.weight": (c,), "norm2.body.bias": (c,)})
    s.update({"ffn." + k: v for k, v in ffn_spec(c).items()})
    return s


def flca_spec(c):
    hid = max(8, c // 8)
    return {"alpha": (), "beta": (), "gamma": (), "low_attn.0.weight": (c, 1, 3, 3),
            "high_attn.0.weight": (c, 1, 3, 3), "chroma_attn.0.weight": (c, 2, 3, 3),
            "se.1.weight": (hid, c, 1, 1), "se.1.bias": (hid,), "se.3.weight": (c, hid, 1, 1), "se.3.bias": (c,)}


def conv_transformer_flca_spec(c):
    s = {"FLCA." + k: v for k, v in flca_spec(c).items()}
    s.update({"Transformer." + k: v for k, v in transformer_spec(c).items()})
    s.update({"channel_reduce.weight": (c, 2 * c, 1, 1), "channel_reduce.bias": (c,),
              "Conv_out.weight": (c, c, 3, 3), "Conv_out.bias": (c,)})
    return s


ATTN_CASES = ((16, (16, 24)), (32, (16, 16)), (48, (8, 12)))
FLCA_CASES = ((16, (32, 48)), (32, (16, 24)), (64, (8, 12)), (128, (4, 6)))
LN_CASES = ((16, (16, 24)), (48, (8, 8)))

MODEL_CASES = (("d16_b2_32x32", 16, 2, 32, 32, 21), ("d16_b1_32x48", 16, 1, 32, 48, 22),
               ("d32_b2_64x64", 32, 2, 64, 64, 23), ("d48_b1_32x32", 48, 1, 32, 32, 24))


def model_state(dim, seed, variant="flca"):
    """Deterministic canonical state_dict for a whole model (what make_golden gave the reference)."""
    from oracle import rawformer_ref as R
    cfg = R.RawFormerConfig(dim=dim, variant=variant)
    shapes = R.param_shapes(cfg)
    return {k: torch.from_numpy(synth.param_values(seed, k, s)).reshape(s) for k, s in shapes.items()}


# a16: Attenblock.LuminanceAwareMHSA cases of tests/golden/attenblock.npz: (tag, dim, heads, B, h, w)
ATTEN_CASES = (("d4", 32, 8, 2, 16, 16), ("d6", 48, 8, 1, 12, 20), ("d16", 64, 4, 1, 8, 24), ("d32", 64, 2, 1, 16, 16))


def atten_spec(dim, heads):
    inner = heads * (dim // heads)
    hid = max(16, inner // 2)
    return {"alpha": (), "to_qkv.weight": (3 * inner, dim, 1, 1), "to_qkv.bias": (3 * inner,),
            "proj.weight": (dim, inner, 1, 1), "proj.bias": (dim,),
            "luma_cond.net.0.weight": (hid, 1, 3, 3), "luma_cond.net.0.bias": (hid,),
            "luma_cond.net.2.weight": (hid, hid, 3, 3), "luma_cond.net.2.bias": (hid,),
            "luma_cond.gamma.weight": (inner, hid, 1, 1), "luma_cond.gamma.bias": (inner,),
            "luma_cond.beta.weight": (inner, hid, 1, 1), "luma_cond.beta.bias": (inner,)}


def atten_inputs(tag, dim, heads, b, h, w):
    x = rnd(f"atten.{tag}.x", (b, dim, h, w), seed=41)
    luma = rnd(f"atten.{tag}.luma", (b, 1, h, w), 0.0, 1.0, seed=42)
    return x, luma, params(atten_spec(dim, heads), seed=700 + dim + heads)


# a17: WFB extras of tests/golden/wfb_extras.npz
WFB_FF_CASES = (("ff32", 32, 2.0, 2, 16, 24), ("ff48", 48, 2.5, 1, 10, 14))     # tag, dim, expansion, B, h, w
WFB_IE_CASES = (("ie32", 32, 2, 16, 24), ("ie40", 40, 1, 9, 14))                # tag, middle channels, B, h, w


def wfb_ff_spec(dim, fac):
    hid = int(dim * fac)
    s = {"project_in.weight": (hid, dim, 1, 1), "project_in.bias": (hid,), "dwconv.weight": (hid, 1, 3, 3), "dwconv.bias": (hid,),
         "project_out.weight": (dim, hid, 1, 1), "project_out.bias": (dim,)}
    for name, k in (("rep_conv1", 3), ("rep_conv2", 1)):
        s[f"{name}.c.weight"] = (hid, 1, k, k)
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"{name}.bn.{leaf}"] = (hid,)
    return s


def wfb_ie_spec(mid):
    return {"conv1.weight": (mid, 4, 1, 1), "conv1.bias": (mid,), "depth_conv.weight": (mid, 1, 5, 5), "depth_conv.bias": (mid,),
            "conv2.weight": (3, mid, 1, 1), "conv2.bias": (3,)}


def atten_tb_spec(dim, heads):
    s = {"norm1.body.weight": (dim,), "norm1.body.bias": (dim,), "norm2.body.weight": (dim,), "norm2.body.bias": (dim,)}
    s.update({"attn." + k: v for k, v in atten_spec(dim, heads).items()})
    s.update({"ffn." + k: v for k, v in ffn_spec(dim).items()})
    return s


def atten_tb_inputs(tag, dim, heads, b, h, w):
    x = rnd(f"atten.tb.{tag}.x", (b, dim, h, w), seed=43)
    luma = rnd(f"atten.tb.{tag}.luma", (b, 1, h, w), 0.0, 1.0, seed=44)
    return x, luma, params(atten_tb_spec(dim, heads), seed=750 + dim)
