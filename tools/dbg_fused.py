#!/usr/bin/env python3
"""Diagnostic (GPU box): fused level-0 TransformerBlock on a shape that gives every persistent workgroup several tiles."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import cases
from cases import rnd, params
from bayer_low_light_image_enhancement_amd import ops
from oracle import rawformer_ref as R
dev = torch.device("cuda:0")
c, heads = 32, 8
p = params(cases.transformer_spec(c))
for shp in ((2, c, 64, 64), (8, c, 128, 256), (8, c, 512, 512)):
    x = rnd("tb.big.x", shp)
    torch.set_num_threads(16)
    ref = R.transformer_block(x, p, "", heads)
    out = ops.transformer_block(x.to(dev), {k: v.to(dev) for k, v in p.items()}, heads=heads).cpu()
    d = (out - ref).abs()
    print(shp, "max err", float(d.max()), "mean", float(d.mean()))
    if float(d.max()) > 1e-4:
        bad = (d > 1e-4)
        print("  bad fraction", float(bad.float().mean()), "rows with errors:", sorted(set(torch.nonzero(bad)[:, 2].tolist()))[:40])
        print("  cols:", sorted(set(torch.nonzero(bad)[:, 3].tolist()))[:40], "images", sorted(set(torch.nonzero(bad)[:, 0].tolist())))
