#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fused kernels (needs the RF_STAMP build of the library,
RF_LIB_PATH=.../librawformer_hip_stamp.so).  Not part of the product."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cases
from bayer_low_light_image_enhancement_amd import RawFormer, synth, _lib

lib = _lib.load()
lib.rf_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
dev = torch.device("cuda:0")
m = RawFormer(dim=32); m.load_state_dict(cases.model_state(32, 132), strict=False); m = m.to(dev).eval()
x = torch.from_numpy(synth.bayer_mosaic(2, 8, 1024, 1024)).to(dev)
buf = (ctypes.c_ulonglong * 8)()
with torch.no_grad():
    m(x); torch.cuda.synchronize(); lib.rf_debug_stamps(buf)
    m(x); torch.cuda.synchronize(); lib.rf_debug_stamps(buf)
v = list(buf)
tot = sum(v[:5])
names = ["barrier wait", "load x + LN", "phase A (1x1 MFMA -> LDS)", "phase B (stencil + MFMA)", "tail (v store / epilogue)"]
print("both fused kernels, all launches of one forward (wave-cycles summed over waves):")
for n, c in zip(names, v[:5]):
    print(f"  {n:32s} {c/1e6:10.1f} Mcycles  {100*c/tot:5.1f} %")
