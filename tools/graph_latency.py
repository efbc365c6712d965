#!/usr/bin/env python3
"""Single-frame latency of RawFormer-S, eager launches vs one captured HIP graph (GPU box only)."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayer_low_light_image_enhancement_amd import RawFormer, synth
from oracle.rawformer_ref import RawFormerConfig, param_shapes

dev = torch.device("cuda:0")
shapes = param_shapes(RawFormerConfig(dim=32))
sd = {k: torch.from_numpy(synth.param_values(132, k, s)).reshape(s) for k, s in shapes.items()}
m = RawFormer(dim=32); m.load_state_dict(sd, strict=False); m = m.to(dev).eval()
for size in (256, 512, 1024):
    x = torch.from_numpy(synth.bayer_mosaic(2, 1, size, size)).to(dev)
    with torch.no_grad():
        for _ in range(3): y = m(x)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): y = m(x)
        torch.cuda.synchronize(); eager = (time.perf_counter() - t) / 50
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): m(x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            yg = m(x)
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(yg, y)
        t = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); graph = (time.perf_counter() - t) / 50
    print(f"mosaic {size}x{size} B=1: eager {eager*1e3:.3f} ms, graph {graph*1e3:.3f} ms")
