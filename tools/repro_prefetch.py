#!/usr/bin/env python3
"""Diagnostic (GPU box): the level-0 fused TransformerBlock of diagnostic library builds against the shipped library.

    python tools/repro_prefetch.py pf1 pf2 ...        # names of tools/_diag/lib_<name>.so (tools/variant.py)

Each library runs in its own process (RF_LIB_PATH is read at import); outputs go to /tmp.  For every shape the variant's
output is compared with the shipped library's (which tests/ pin against the reference); wrong pixels are attributed to
4x64 tiles, to the tile's position inside its workgroup's run of tiles and to the lane group that loaded them.
"""
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = ((8, 32, 128, 256), (1, 32, 512, 512), (2, 32, 512, 512), (3, 32, 512, 512), (8, 32, 512, 512))
TMP = "/tmp/rf_repro"


def worker(name):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch
    import cases
    from cases import rnd, params
    from bayer_low_light_image_enhancement_amd import ops
    dev = torch.device("cuda:0")
    c, heads = 32, 8
    p = {k: v.to(dev) for k, v in params(cases.transformer_spec(c)).items()}
    os.makedirs(TMP, exist_ok=True)
    for shp in SHAPES:
        x = rnd("tb.big.x", shp).to(dev)
        outs = [ops.transformer_block(x, p, heads=heads).cpu().numpy() for _ in range(2)]
        print(name, shp, "run-to-run identical:", bool((outs[0] == outs[1]).all()), flush=True)
        np.save(os.path.join(TMP, f"{name}_{'x'.join(map(str, shp))}.npy"), outs[0])
    # the depthwise-convolved v that attn_front leaves in the block's workspace (bufB of rf_block.hip), on a workspace pre-filled
    # with a sentinel: a tile that was never written shows the sentinel, a tile written to the wrong place shows another tile's data
    import ctypes as C
    from bayer_low_light_image_enhancement_amd import _lib
    lib = _lib.load()
    B, h, w = 2, 512, 512
    x = rnd("tb.big.x", (B, c, h, w)).to(dev)
    ts = [ops._chk(p[k], k) for k in ops._TB_KEYS]
    sz = C.c_size_t()
    lib.rf_transformer_block_scratch_bytes(B, c, heads, 2, h, w, C.byref(sz))
    scratch = torch.full((sz.value // 4,), 777.0, dtype=torch.float32, device=dev)
    out = torch.empty_like(x)
    ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    _lib.check(lib.rf_transformer_block(x.data_ptr(), out.data_ptr(), ptrs, scratch.data_ptr(), B, c, heads, 2, h, w, 0), "tb")
    torch.cuda.synchronize()
    al = lambda n: (n + 63) // 64 * 64
    P = h * w
    voff = 17920 + al(B * 96 * P)
    v = scratch[voff: voff + B * c * P].cpu().numpy().reshape(B, c, h, w)
    np.save(os.path.join(TMP, f"{name}_v.npy"), v)
    raw = C.CDLL(_lib.LIB_PATH)
    if hasattr(raw, "rf_debug_repro"):          # -DRF_REPRO_CHECK builds: prefetched registers vs a fresh load of the same addresses
        buf = (C.c_uint * (4 + 64 * 8))()
        raw.rf_debug_repro(buf, len(buf))
        rec = np.frombuffer(buf, dtype=np.uint32)
        print(f"{name}: register-content check (all shapes so far): {rec[1]} wave-tiles checked, {rec[2]} lanes whose b3 pieces changed between the split and the v part (load 241: checksums bp0 then/now, bp1 then/now), {rec[0]} mismatches (load 255 = the carried geometry differs: "
              f"fresh goff / carried goff / fresh valid,lds_off / carried valid,lds_off in the register/fresh/goff/valid fields); first records:")
        f = lambda u: float(np.array([u], dtype=np.uint32).view(np.float32)[0])
        print('   lanes x pixel groups whose normalised pieces differ from a fresh recomputation (load 224 + g):', int(rec[3]))
        for i in range(min(int(rec[0]) + int(rec[2]) + int(rec[3]), 64)):
            r = rec[4 + 8 * i: 12 + 8 * i]
            hw = int(r[7])
            print(f"   slab {r[0] & 0xffff} image {r[0] >> 16} tile {r[1] & 0xffff} (#{r[1] >> 16} of its workgroup) thread {r[2] & 0xffff} (wave {(r[2] & 0xffff) >> 6} lane {r[2] & 63}) load {r[2] >> 16}: "
                  f"register {f(r[3]):+.6f} fresh {f(r[4]):+.6f} [{r[3]:08x} {r[4]:08x} {r[5]:08x} {r[6]:08x}] goff {r[5]} valid {r[6] & 1} lds_off {(int(r[6]) >> 8) - 1}; "
                  f"r7 {f(r[7]):+.6f} r5 {f(r[5]):+.6f} r6 {f(r[6]):+.6f} HW_ID wave {hw & 15} simd {(hw >> 4) & 3} pipe {(hw >> 6) & 3} cu {(hw >> 8) & 15} sh {(hw >> 12) & 1} se {(hw >> 13) & 7}")
    ns = 256
    poff = 17920 + 2 * al(B * 96 * P) + al(B * c * P)
    np.save(os.path.join(TMP, f"{name}_partial.npy"), scratch[poff: poff + B * ns * 2 * 16 * 66].cpu().numpy().reshape(B, ns, 2, 16, 66))


def compare(name):
    for shp in SHAPES:
        tag = "x".join(map(str, shp))
        ref = np.load(os.path.join(TMP, f"pf0_{tag}.npy"))
        out = np.load(os.path.join(TMP, f"{name}_{tag}.npy"))
        d = np.abs(out - ref)
        print(f"{name} {shp}: max |diff| vs shipped {d.max():.3e} mean {d.mean():.3e}")
        if d.max() <= 1e-4:
            continue
        bad = (d > 1e-4).any(axis=1)                     # [B, h, w]
        B, h, w = bad.shape
        tiles = bad.reshape(B, h // 4, 4, w // 64, 64)
        tbad = tiles.any(axis=(2, 4))                    # [B, tiles_y, tiles_x]
        tiles_y = h // 4
        ids = [(b, tx * tiles_y + ty) for b, ty, tx in zip(*np.nonzero(tbad))]
        print(f"   wrong pixels {bad.mean():.4%}; wrong tiles {tbad.sum()} of {tbad.size}; per image {tbad.sum(axis=(1, 2)).tolist()}")
        pos = np.bincount([t % 4 for _, t in ids], minlength=4)
        print("   position of the wrong tiles inside their workgroup's 4 tiles:", pos.tolist())
        # inside wrong tiles: which rows / 16-pixel column groups
        sub = np.zeros((4, 16), dtype=np.int64)
        for b, ty, tx in zip(*np.nonzero(tbad)):
            t = tiles[b, ty, :, tx, :]                   # [4, 64]
            sub += t.reshape(4, 16, 4).any(axis=2)
        print("   wrong 4-pixel groups by (row in tile, group): ")
        for r in range(4):
            print("     ", sub[r].tolist())
        ch = (d > 1e-4).sum(axis=(0, 2, 3))
        print("   wrong values per channel:", ch.tolist())
        print("   first wrong tiles (image, tile id, slab):", [(b, t, t // 4) for b, t in ids[:24]])


def compare_v(name):
    ref = np.load(os.path.join(TMP, "pf0_v.npy"))
    v = np.load(os.path.join(TMP, f"{name}_v.npy"))
    B, c, h, w = v.shape
    tiles_y = h // 4
    d = np.abs(v - ref) > 1e-5
    td = d.reshape(B, c, tiles_y, 4, w // 64, 64)
    tbad = td.any(axis=(1, 3, 5))                               # [B, tiles_y, tiles_x]
    print(f"{name} v buffer (2, 32, 512, 512): wrong tiles {tbad.sum()} per image {tbad.sum(axis=(1, 2)).tolist()}; sentinel values left {int((v == 777.0).sum())}")
    vt = v.reshape(B, c, tiles_y, 4, w // 64, 64)
    rt = ref.reshape(B, c, tiles_y, 4, w // 64, 64)
    n = 0
    for b, ty, tx in zip(*np.nonzero(tbad)):
        t = vt[b, :, ty, :, tx, :]
        frac = float(td[b, :, ty, :, tx, :].mean())
        chans = np.nonzero(td[b, :, ty, :, tx, :].any(axis=(1, 2)))[0].tolist()
        rows = np.nonzero(td[b, :, ty, :, tx, :].any(axis=(0, 2)))[0].tolist()
        # does the wrong content equal some OTHER tile of the right answer (a misdirected store)?
        hit = None
        wrongc = chans[0]
        cand = np.nonzero((np.abs(rt[:, wrongc] - t[wrongc][None, None, :, None, :]) < 1e-6).all(axis=(2, 4)))
        if len(cand[0]):
            hit = [(int(bb), int(xx) * tiles_y + int(yy)) for bb, yy, xx in zip(*cand)][:3]
        print(f"   image {b} tile {tx * tiles_y + ty} (slab {(tx * tiles_y + ty) // 4}, pos {(tx * tiles_y + ty) % 4}): {frac:.1%} of values wrong, channels {chans[:6]}{'...' if len(chans) > 6 else ''} ({len(chans)}), rows {rows}; "
              f"sentinel {int((t == 777.0).sum())}; equals correct tile(s) {hit}")
        if n < 8:
            for r in range(4):
                print("        row", r, "".join("X" if q else "." for q in td[b, :, ty, r, tx, :].any(axis=0)))
        n += 1
        if n >= 40:
            break
    pr, pp = np.load(os.path.join(TMP, "pf0_partial.npy")), np.load(os.path.join(TMP, f"{name}_partial.npy"))
    dp = np.abs(pr - pp) > 1e-3 * (np.abs(pr) + 1.0)
    print(f"   Gram partials: wrong slabs {np.nonzero(dp.any(axis=(2, 3, 4)))}")


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(sys.argv[2])
        sys.exit(0)
    names = ["pf0"] + [n for n in sys.argv[1:] if n != "pf0"]
    for n in names:
        env = dict(os.environ)
        if n != "pf0":
            env["RF_LIB_PATH"] = os.path.join(REPO, "tools", "_diag", f"lib_{n}.so")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", n], env=env)
        if r.returncode != 0:
            print(n, "worker failed", r.returncode)
            sys.exit(1)
    for n in names[1:]:
        compare(n)
        compare_v(n)
