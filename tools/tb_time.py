#!/usr/bin/env python3
"""Diagnostic (GPU box): time of the level-0 TransformerBlock kernels at the cfg2 shape for the library in RF_LIB_PATH,
and the block's error against the oracle at that size."""
import ctypes, json, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import cases
from cases import rnd, params
from bayer_low_light_image_enhancement_amd import _lib, ops
dev = torch.device("cuda:0")
c, heads = 32, 8
p = params(cases.transformer_spec(c))
pd = {k: v.to(dev) for k, v in p.items()}
x = rnd("tb.big.x", (8, c, 512, 512))
xd = x.to(dev)
lib = _lib.load()
for _ in range(3): out = ops.transformer_block(xd, pd, heads=heads)
torch.cuda.synchronize()
lib.rf_profile_begin()
for _ in range(10): out = ops.transformer_block(xd, pd, heads=heads)
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
_lib.check(lib.rf_profile_end(buf, len(buf)), "rf_profile_end")
for r in json.loads(buf.value.decode()):
    if "fused" in r["kernel"] or "attn_front" in r["kernel"]:
        print(os.environ.get("RF_LIB_PATH", "main"), r["kernel"], round(r["ms"] / r["launches"] * 1e3, 1), "us")
if "--check" in sys.argv:
    from oracle import rawformer_ref as R
    torch.set_num_threads(16)
    ref = R.transformer_block(x, p, "", heads)
    print("  max err", float((out.cpu() - ref).abs().max()))
