#!/usr/bin/env python3
"""Diagnostic (GPU box): where does the fused attention differ -- the v path or the Gram statistics?
Runs the level-0 TransformerBlock with the FFN's pw2 zeroed (block = x + attention) and with temperature 0 (softmax uniform)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import cases
from cases import rnd, params
from bayer_low_light_image_enhancement_amd import ops
from oracle import rawformer_ref as R
dev = torch.device("cuda:0")
c, heads = 32, 8
base = params(cases.transformer_spec(c))
x = rnd("tb.dbg.x", (2, c, 64, 64))
torch.set_num_threads(16)
for tag, edit in (("full", {}), ("ffn-off", {"ffn.pointwise2.weight": 0, "ffn.pointwise2.bias": 0}),
                  ("ffn-off,temp0", {"ffn.pointwise2.weight": 0, "ffn.pointwise2.bias": 0, "attn.temperature": 0})):
    p = {k: v.clone() for k, v in base.items()}
    for k, s in edit.items(): p[k] = p[k] * s
    ref = R.transformer_block(x, p, "", heads)
    out = ops.transformer_block(x.to(dev), {k: v.to(dev) for k, v in p.items()}, heads=heads).cpu()
    d = (out - ref).abs()
    print(tag, "max err", float(d.max()), "mean", float(d.mean()), "per image", [float(d[i].max()) for i in range(2)],
          "rows 0-3 / 4-7 / 60-63:", float(d[:, :, 0:4].max()), float(d[:, :, 4:8].max()), float(d[:, :, 60:64].max()))
