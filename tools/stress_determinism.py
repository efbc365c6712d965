#!/usr/bin/env python3
"""Diagnostic (GPU box): run-to-run bit stability of the shipped kernels under load -- the level-0 TransformerBlock at the cfg2
shape and the whole cfg2 forward, N repetitions each, every output compared bitwise with the first one."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import cases
from cases import rnd, params
from bayer_low_light_image_enhancement_amd import ops, RawFormer, synth
from oracle import rawformer_ref as R
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
c, heads = 32, 8
p = {k: v.to(dev) for k, v in params(cases.transformer_spec(c)).items()}
for shp in ((8, c, 512, 512), (2, c, 512, 512), (3, c, 512, 512)):
    x = rnd("tb.big.x", shp).to(dev)
    first = ops.transformer_block(x, p, heads=heads)
    bad = 0
    for i in range(N):
        out = ops.transformer_block(x, p, heads=heads)
        if not torch.equal(out, first):
            bad += 1
    print("transformer_block", shp, f"{bad} of {N} repetitions differ from the first", flush=True)
dim = 32
shapes = R.param_shapes(R.RawFormerConfig(dim=dim))
sd = {k: torch.from_numpy(synth.param_values(5, k, s)).reshape(s) for k, s in shapes.items()}
m = RawFormer(dim=dim)
m.load_state_dict({**m.state_dict(), **sd}, strict=True)
m = m.to(dev).eval()
x = torch.from_numpy(synth.bayer_mosaic(2, 8, 1024, 1024)).to(dev)
with torch.no_grad():
    first = m(x)
    bad = 0
    for i in range(N // 3):
        if not torch.equal(m(x), first):
            bad += 1
print("RawFormer-S cfg2 forward", f"{bad} of {N // 3} repetitions differ from the first", flush=True)
