#!/usr/bin/env python3
"""Kernel micro-benchmarks at the shapes RawFormer-S/B use in BASELINE configs 2-3 (GPU box only).

Each case runs through the C ABI (ops.*) with the library's own HIP-event bracket
(rf_profile_begin/end), so the figure is the kernel's time on its launch stream, without the
weight-repack helper that the operator-level entry points run first.

usage: python tools/kbench.py [conv1x1] [conv3x3] [dw] [attn] [flca] [dwt] [--dim 32] [--batch 8] [--size 512]
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bayer_low_light_image_enhancement_amd import _lib, ops  # noqa: E402


def timed(fn, iters=10, warm=2):
    lib = _lib.load()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    lib.rf_profile_begin()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(lib.rf_profile_end(buf, len(buf)), "rf_profile_end")
    return [r for r in json.loads(buf.value.decode()) if not r["kernel"].startswith("pack")]


def report(tag, recs):
    for r in recs:
        us = r["ms"] / r["launches"] * 1e3
        tf = r["flops"] / max(r["ms"], 1e-9) / 1e9
        gb = r["bytes"] / max(r["ms"], 1e-9) / 1e6
        print(f"{tag:44s} {r['kernel']:28s} {us:9.1f} us  {tf:7.2f} TF/s  {gb:8.1f} GB/s", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["conv1x1", "conv3x3", "dw", "attn", "flca", "dwt"])
    ap.add_argument("--dim", type=int, default=32)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, S, d = a.batch, a.size, a.dim
    r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    for lvl in range(4):
        C, h = d << lvl, S >> lvl
        x = r(B, C, h, h)
        if "conv1x1" in a.what:
            lw, lb = r(C), r(C)
            report(f"L{lvl} qkv   {C}->{3*C} LN        {h}x{h}", timed(lambda: ops.conv1x1(x, r(3 * C, C, 1, 1), r(3 * C), ln_weight=lw, ln_bias=lb)))
            report(f"L{lvl} pw1   {C}->{2*C} LN        {h}x{h}", timed(lambda: ops.conv1x1(x, r(2 * C, C, 1, 1), r(2 * C), ln_weight=lw, ln_bias=lb)))
            report(f"L{lvl} proj  {C}->{C} +res       {h}x{h}", timed(lambda: ops.conv1x1(x, r(C, C, 1, 1), r(C), residual=x)))
            x2 = r(B, 2 * C, h, h)
            report(f"L{lvl} pw2   {2*C}->{C} +res      {h}x{h}", timed(lambda: ops.conv1x1(x2, r(C, 2 * C, 1, 1), r(C), residual=x)))
            report(f"L{lvl} cat   {C}+{C}->{C}         {h}x{h}", timed(lambda: ops.conv1x1(x, r(C, 2 * C, 1, 1), r(C), x2=x)))
            if lvl < 3:
                xs = r(B, 2 * C, h // 2, h // 2)
                report(f"L{lvl} convT {2*C}->{C} x4        {h//2}x{h//2}", timed(lambda: ops.conv_transpose2x2(xs, r(2 * C, C, 2, 2), r(C))))
        if "conv3x3" in a.what:
            report(f"L{lvl} conv3 {C}->{C} lrelu      {h}x{h}", timed(lambda: ops.conv3x3(x, r(C, C, 3, 3), r(C), act="lrelu")))
            if lvl < 3:
                report(f"L{lvl} down  {C}->{C//2} unshuf   {h}x{h}", timed(lambda: ops.conv3x3(x, r(C // 2, C, 3, 3), None, store="unshuffle")))
            if lvl == 0:
                report(f"L0 embed 4->{C}             {h}x{h}", timed(lambda: ops.conv3x3(r(B, 4, h, h), r(C, 4, 3, 3), r(C))))
                report(f"L0 out   {C}->12 shuffle     {h}x{h}", timed(lambda: ops.conv3x3(x, r(12, C, 3, 3), r(12), act="lrelu", store="shuffle")))
        if "dw" in a.what:
            x3 = r(B, 3 * C, h, h)
            report(f"L{lvl} dw    {3*C}               {h}x{h}", timed(lambda: ops.dwconv3x3(x3, r(3 * C, 1, 3, 3), r(3 * C))))
            x2 = r(B, 2 * C, h, h)
            report(f"L{lvl} dw+gelu {2*C}             {h}x{h}", timed(lambda: ops.dwconv3x3(x2, r(2 * C, 1, 3, 3), r(2 * C), gelu=True)))
        if "attn" in a.what:
            report(f"L{lvl} chan_attn C={C}           {h}x{h}", timed(lambda: ops.channel_attention(
                x, r(3 * C, C, 1, 1), r(3 * C), r(3 * C, 1, 3, 3), r(3 * C), r(8, 1, 1), r(C, C, 1, 1), r(C), 8)))
    if "c3probe" in a.what:   # conv3x3 at controlled (Cin, Cout, size, batch): separates chunk count from image size
        for ci, co, sz, bb in ((32, 32, 512, 8), (64, 32, 512, 8), (32, 32, 256, 32), (32, 32, 1024, 2), (64, 32, 256, 8),
                               (16, 32, 512, 8), (32, 64, 512, 8), (64, 64, 512, 4)):
            xx = r(bb, ci, sz, sz)
            report(f"c3 {ci}->{co} {sz}x{sz} B={bb}", timed(lambda: ops.conv3x3(xx, r(co, ci, 3, 3), r(co), act="lrelu")))
            del xx
    if "dwt" in a.what:
        x = r(B, d, S, S)
        report("dwt_init", timed(lambda: ops.dwt_init(x)))
        report("iwt_init", timed(lambda: ops.iwt_init(x.reshape(4 * B, d // 4, S, S))))
        report("CustomDWT", timed(lambda: ops.custom_dwt(x)))
        report("CustomIDWT", timed(lambda: ops.custom_idwt(x)))
        report("downshuffle", timed(lambda: ops.downshuffle(x)))
        report("pixel_shuffle", timed(lambda: ops.pixel_shuffle(x)))
        report("layernorm2d", timed(lambda: ops.layernorm2d(x, r(d), r(d))))


if __name__ == "__main__":
    main()
