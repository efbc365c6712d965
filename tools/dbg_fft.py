#!/usr/bin/env python3
"""Diagnostic (GPU box): rfft2_polar against torch.fft on the CPU; where do large FEB differences come from?"""
import os, sys, math
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from cases import rnd
from bayer_low_light_image_enhancement_amd import ops
dev = torch.device("cuda:0")
x = rnd("fft.tall.x", (1, 1, 24, 8), seed=71)
mag, pha = ops.rfft2_polar(x.to(dev))
f = torch.fft.rfft2(x, norm="ortho")
for y in (0, 12):
    for xx in (0, 4):
        print("bin", y, xx, "mine pha", float(pha[0, 0, y, xx]), "ref pha", float(torch.angle(f)[0, 0, y, xx]), "ref", complex(f[0, 0, y, xx]))
for shape in ((2, 32, 128, 128), (1, 16, 36, 52)):
    x = rnd("feb.big.x", shape, -2.0, 2.0, seed=63)
    mag, pha = ops.rfft2_polar(x.to(dev))
    f = torch.fft.rfft2(x, norm="ortho")
    rm, rp = f.abs() + 1e-6, torch.angle(f)
    dm = (mag.cpu() - rm).abs().max()
    dp = (pha.cpu() - rp).abs()
    flips = dp > 3.0
    print(shape, "mag err", float(dm), "phase: flips", int(flips.sum()), "max |dpha| elsewhere", float(dp[~flips].max()),
          "cplx err", float((torch.polar(mag.cpu(), pha.cpu()) - torch.polar(rm, rp)).abs().max()))
    if flips.any():
        idx = torch.nonzero(flips)[:5]
        for i in idx:
            print("   flip at", i.tolist(), "ref value", complex(f[tuple(i.tolist())]))
