#!/usr/bin/env python3
"""RawFormer inference throughput on MI355X: megapixels/s, RAW -> sRGB.

A step = one ``RawFormer.forward`` (HIP path) over one batch of synthetic low-light Bayer
frames already resident in HBM.  The default workload is BASELINE.json configs[1]:
RawFormer-S (dim 32), batch 8 of packed 4x512x512 (mosaic 1x1024x1024) per GPU.

Multi-GPU (one process per GPU, RCCL):
  * ``python bench.py --gpus N`` with no WORLD_SIZE in the environment starts the N ranks itself
    (a ``torch.distributed.run`` child; this parent never touches the GPU), relays rank 0's JSON line
    and exits non-zero when fewer than N ranks took part.  Launched BY ``torch.distributed.run`` it is
    one of the ranks; WORLD_SIZE must then equal ``--gpus``.
  * cfg1/cfg2/cfg3/frame1: every rank runs the per-GPU batch on its own images -- the path shards along
    the batch with no data-path collective (weak scaling; only the timing barrier and the MAX over
    ranks use RCCL).
  * cfg5 (BASELINE configs[4]): one TRAINING step per step -- forward, L1 loss, backward (explicit adjoint schedule,
    rf_train_step), bucketed all-reduce of the flat 13.4 MB gradient buffer over RCCL INSIDE the timed region, AdamW on the flat
    buffers; 4 images per GPU (weak scaling: 8 GPUs = the batch of 32 the config names).
  * cfg4 (RawFormer-L, one SID Sony frame 1x4x1424x2128, BASELINE configs[3]): N = 1 runs the whole frame; N > 1 cuts it into N
    EXACT row shards (tiling.forward_full_frame_exact / rf_set_shard): 80 packed rows of recomputed context per side, the Gram /
    squeeze-excite / luma-max statistics all-reduced inside the forward, one all-gather of the interior strips -- the stitched
    frame equals the whole-frame forward, the only mode that meets the north star's "within 1e-3 PSNR of the reference"
    (strong scaling; all collectives timed).  `cfg4x` is kept as an alias.
  * cfg4t: the same frame in N INDEPENDENT overlapping tiles (multiples of 64 mosaic px), rank r runs tile r and ONE all-gather
    of the tile outputs stitches the sRGB frame -- cheaper (no statistics collectives, 1.2-1.3 x the ideal pixels) but every
    tile is its own "image" for the attention / squeeze-excite / luma statistics: 22.5 dB from the untiled reference with
    these weights (tests/golden/tiling_psnr.json).

Rank 0 prints ONE JSON line.  Besides the driver's fields it carries
  roofline      the dominant kernel (largest share of the forward), timed live with HIP events
                on the launch stream (rf_profile_begin/end) over the same K steps in a second,
                untimed pass, priced against its algorithmic FLOPs / bytes;
  cpu_baseline  the CPU oracle (oracle/rawformer_ref.py, torch CPU ops, all host cores) on a
                bounded sample of the same workload, rank 0 at N = 1 only;
  kernels       the per-kernel-class breakdown of that profiled pass (share of forward time);
  dwt_roofline  the stand-alone DWT / IDWT kernels at the stage-0 activation size.

``--dry-run`` (CPU, gloo, a stand-in forward) exercises the launcher, rendezvous, sharding, the
tiled all-gather and the timing reduction without a GPU; its line says ``"dry_run": true`` and is
not a measurement.
"""
from __future__ import annotations

import argparse
import ctypes
import glob
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* = f32 vector rate
PEAK_BF16_MFMA_TFLOPS = 2500.0 # dense bf16 MFMA (the bf16x3 GEMM kernels issue 6 bf16 products per f32 product)
PEAK_HBM_GBS = 8000.0          # HBM3E spec

WORKLOADS = {
    # name: (dim, batch per GPU, mosaic H, mosaic W, description)
    "cfg2": (32, 8, 1024, 1024, "RawFormer-S(FLCA) dim=32, batch=8 of packed 4x512x512 synthetic Bayer per GPU"),
    "cfg1": (32, 1, 256, 256, "RawFormer-S(FLCA) dim=32, one packed 4x128x128 frame"),
    "cfg3": (48, 8, 1024, 1024, "RawFormer-B(FLCA) dim=48, batch=8 of packed 4x512x512 synthetic Bayer per GPU"),
    "cfg4": (64, 1, 2848, 4256, "RawFormer-L(FLCA) dim=64, one SID Sony full frame, packed 4x1424x2128"),
    "cfg4x": (64, 1, 2848, 4256, "RawFormer-L(FLCA) dim=64, one SID Sony full frame, packed 4x1424x2128 (exact row shards)"),
    "cfg4t": (64, 1, 2848, 4256, "RawFormer-L(FLCA) dim=64, one SID Sony full frame, packed 4x1424x2128 (independent tiles)"),
    "frame1": (32, 1, 1024, 1024, "RawFormer-S(FLCA) dim=32, ONE packed 4x512x512 frame (test.py:72 batch_size=1)"),
    "cfg5": (32, 4, 1024, 1024, "RawFormer-S(FLCA) dim=32 TRAINING step (forward + L1 loss + backward + gradient all-reduce + AdamW), "
                                "batch 4 of packed 4x512x512 per GPU"),
}
TILE_GRIDS = {1: (1, 1), 2: (1, 2), 3: (1, 3), 4: (2, 2), 5: (1, 5), 6: (2, 3), 7: (1, 7), 8: (2, 4)}
TILE_OVERLAP = 64   # mosaic px of context per interior tile edge (tests/golden/tiling_psnr.json quantifies 32/64/128)
TILE_ALIGN = 64     # tile sizes are multiples of 64 mosaic px: level 3 stays on the 16-byte vector paths


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# --------------------------------------------------------------------------------------------- launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv) -> int:
    """Parent of an N-rank run.  Nothing here initialises HIP (no torch.cuda call, no library load): the ranks
    are CHILD processes of ``torch.distributed.run``, never a re-exec of a process that has touched the GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    log(f"starting {n} ranks: {' '.join(cmd[1:8])} bench.py {' '.join(argv)}")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                line = json.loads(ln)
            except ValueError:
                continue
    if proc.returncode != 0:
        sys.stdout.write(proc.stdout)
        log(f"rank launch failed with exit code {proc.returncode}")
        return proc.returncode or 1
    if line is None or line.get("n_gpus") != n:
        sys.stdout.write(proc.stdout)
        log(f"expected a result from {n} ranks, got {None if line is None else line.get('n_gpus')}")
        return 3
    print(json.dumps(line), flush=True)
    return 0


# --------------------------------------------------------------------------------------------- measurement helpers
def profile_pass(fn, steps):
    """Run ``fn`` ``steps`` times with every kernel launch bracketed by HIP events."""
    import torch
    from bayer_low_light_image_enhancement_amd import _lib

    lib = _lib.load()
    torch.cuda.synchronize()
    _lib.check(lib.rf_profile_begin(), "rf_profile_begin")
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(lib.rf_profile_end(buf, len(buf)), "rf_profile_end")
    return json.loads(buf.value.decode())


def dry_profile_pass(fn, steps):
    """``--dry-run`` stand-in of profile_pass: same control flow (the step is re-run ``steps`` times on the ranks that
    profile), one synthetic record instead of HIP-event timings."""
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ms = (time.perf_counter() - t0) * 1e3
    return [{"kernel": "dry_run_step", "launches": steps, "ms": ms, "flops": 0.0, "bytes": 0.0}]


def issued_over_algorithmic(kernel: str) -> float:
    """MFMA flops the kernel ISSUES per algorithmic flop.  The 3x3 convolutions use the Winograd F(4,3) form along x:
    6 products per 4 outputs x 3 taps instead of 12; the bf16x3 GEMM issues six bf16 products per
    f32 product (three-piece split of both operands)."""
    if kernel.startswith("conv3x3_kernel"):
        return 0.5
    return 6.0 if kernel.startswith("conv1x1_b3_") else 1.0


def pipe_peak(kernel: str) -> float:
    return PEAK_BF16_MFMA_TFLOPS if kernel.startswith("conv1x1_b3_") else PEAK_F32_MFMA_TFLOPS


def roofline_of(rec):
    """Both floors of a kernel class and the one that binds.  ``frac_hbm`` = algorithmic bytes / time / 8 TB/s; ``frac_mfma`` =
    the MFMA flops the kernel's algorithm ISSUES / time / the peak of the pipe it issues them on (f32: 157.3 TFLOP/s; the bf16x3
    GEMMs: 2500 TFLOP/s with six products per f32 product; Winograd F(4,3): half the algorithmic flops).  ``bound`` is the
    larger floor -- i.e. the ridge of the pipe actually used decides (52 flop/B for bf16x3, 19.7 for f32) -- ``frac`` its
    fraction, never above 1; ``achieved`` / ``peak`` are in that bound's unit and price ALGORITHMIC work."""
    ms = rec["ms"] / rec["launches"]
    flops, byts = rec["flops"] / rec["launches"], rec["bytes"] / rec["launches"]
    ratio, pipe = issued_over_algorithmic(rec["kernel"]), pipe_peak(rec["kernel"])
    t = ms * 1e-3
    frac_hbm = byts / t / (PEAK_HBM_GBS * 1e9)
    frac_mfma = flops * ratio / t / (pipe * 1e12)
    common = {"kernel": rec["kernel"], "frac_hbm": round(frac_hbm, 4), "frac_mfma": round(frac_mfma, 4), "traffic": None,
              "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": None,
              "floor_us": round(max(byts / (PEAK_HBM_GBS * 1e9), flops * ratio / (pipe * 1e12)) * 1e6, 2)}
    if frac_mfma >= frac_hbm:
        ach = flops / t / 1e12
        return dict(common, bound="mfma", achieved=round(ach, 3), peak=round(pipe / ratio, 1), unit="TFLOP/s", frac=round(frac_mfma, 4),
                    mfma_pipe_peak=pipe, mfma_flops_issued_over_algorithmic=ratio, issued_tflops=round(ach * ratio, 3))
    return dict(common, bound="hbm", achieved=round(byts / t / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(frac_hbm, 4))


def _latest_profile(suffix: str, workload: str):
    """Newest committed ``profiles/*_<suffix>_<workload>.json`` (cfg2 also answers to the un-suffixed name of rounds 1-2)."""
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"*_{suffix}_{workload}.json")))
    if not files and workload == "cfg2":
        files = sorted(glob.glob(os.path.join(REPO, "profiles", f"*_{suffix}.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))
    except Exception:
        return None


def _pmc_record(kernels: dict, key: str, field: str, weight: str):
    """``field`` of the kernel ``key``; a bracket key that names a kernel family by its first template argument only (the
    library's ``gram2_kernel<1>`` covers rocprof's ``gram2_kernel<1, 6, 2, false>`` ...) is the launch-weighted mean of the family."""
    if key in kernels:
        return kernels[key].get(field)
    if key.endswith(">"):
        fam = [v for k, v in kernels.items() if k.startswith(key[:-1] + ",")]
        tot = sum(v.get(weight, 0) for v in fam)
        if fam and tot:
            val = sum(v[field] * v.get(weight, 0) for v in fam if v.get(field) is not None) / tot
            return round(val, 4) if val < 10 else int(val)
    return None


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of ``kernel`` from the committed rocprofv3 PMC passes of this same command and workload
    (profiles/*_traffic_<workload>.json, made by tools/traffic.py: FETCH_SIZE and WRITE_SIZE in separate passes, gfx950
    corrections applied).  PMC counters cannot be read from inside the process; None when no file covers the workload or the
    kernel is not in it."""
    d = _latest_profile("traffic", workload)
    rec = _pmc_record((d or {}).get("kernels", {}), kernel, "hbm_bytes_per_launch", "launches_sampled")
    return rec


def pmc_mfma_util(kernel, workload):
    """Matrix-pipe busy fraction of ``kernel`` from the committed SQ-counter pass (profiles/*_mfma_util_<workload>.json,
    made by tools/mfma_util.py from ``rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE``)."""
    d = _latest_profile("mfma_util", workload)
    return _pmc_record((d or {}).get("kernels", {}), kernel, "mfma_busy_frac", "dispatches")


def host_cores() -> int:
    """Usable host cores: affinity mask, capped by the cgroup CPU quota of the box."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    cores = min(cores, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except Exception:
            continue
    return cores


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(dim, hm, wm, sd):
    """Oracle forward on the host cores, bounded to roughly 10-30 s of CPU work."""
    import torch
    from bayer_low_light_image_enhancement_amd import synth
    from oracle import rawformer_ref as R

    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = R.RawFormerConfig(dim=dim)
    sd_cpu = {k: v.detach().cpu() for k, v in sd.items()}
    base = {"unit": "MP/s", "cores": cores, "kind": "port", "cpu": cpu_model()}
    if hm * wm > 4 << 20:   # full SID frame: a 1024x1024 crop of it bounds the sample (the oracle is per-pixel work)
        hm, wm = 1024, 1024
    x1 = torch.from_numpy(synth.bayer_mosaic(900, 1, hm, wm))
    with torch.no_grad():
        t0 = time.perf_counter()
        R.rawformer_forward(sd_cpu, x1, cfg)          # warm-up, also sizes the sample
        t_one = time.perf_counter() - t0
        log(f"cpu_baseline: warm-up frame took {t_one:.2f} s on {cores} threads")
        if t_one > 15.0:   # slow host: the warm-up frame itself is the bounded sample
            return dict(base, value=round(hm * wm / 1e6 / t_one, 4),
                        sample=f"1 frame of 1x{hm}x{wm} mosaic, torch CPU fp32 oracle, single cold run, {cores} threads")
        nimg = int(max(1, min(8, 12.0 // max(t_one, 1e-3))))
        x = torch.from_numpy(synth.bayer_mosaic(901, nimg, hm, wm))
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            R.rawformer_forward(sd_cpu, x, cfg)
            times.append(time.perf_counter() - t0)
    best = min(times)
    return dict(base, value=round(nimg * hm * wm / 1e6 / best, 4),
                sample=f"{nimg} frame(s) of 1x{hm}x{wm} mosaic, torch CPU fp32 oracle, best of 2 after warm-up, {cores} threads")


def dwt_microbench(device):
    """Stand-alone DWT / IDWT (a11-a13) at the cfg2 stage-0 activation size 8x32x512x512."""
    import torch
    from bayer_low_light_image_enhancement_amd import ops

    x = torch.rand(8, 32, 512, 512, device=device)
    res = {}
    for name, fn, arg in (("dwt_init", ops.dwt_init, x), ("iwt_init", ops.iwt_init, x.reshape(32, 8, 512, 512)),
                          ("CustomDWT", ops.custom_dwt, x), ("CustomIDWT", ops.custom_idwt, x.reshape(8, 32, 512, 512))):
        for _ in range(3):
            fn(arg)
        recs = profile_pass(lambda: fn(arg), 20)
        rec = recs[0]
        ms = rec["ms"] / rec["launches"]
        gbs = rec["bytes"] / rec["launches"] / (ms * 1e-3) / 1e9
        res[name] = {"kernel": rec["kernel"], "avg_launch_us": round(ms * 1e3, 2), "achieved": round(gbs, 1),
                     "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "elements": x.numel(), "bytes_per_element": 8}
    return res


def synthetic_state(model, seed):
    """Seeded weights for every parameter the library's registry (rf_param_info) declares, by name and shape; the
    module's constant buffers (luma weights, Haar filters) keep the values its constructor gave them."""
    import torch
    from bayer_low_light_image_enhancement_amd import synth

    sd = model.state_dict()
    for k, p in model.named_parameters():
        sd[k] = torch.from_numpy(synth.param_values(seed, k, tuple(p.shape))).reshape(p.shape)
    return sd


# --------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="CPU + gloo + stand-in forward: launcher / sharding / collective rehearsal")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))      # BEFORE any torch.cuda / HIP call in this process
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher provides WORLD_SIZE={world}; refusing to report a "
                         f"run of {world} rank(s) as {args.gpus}")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist

    if args.dry_run:
        device = torch.device("cpu")
        torch.set_num_threads(2)
    else:
        if torch.cuda.device_count() <= local:
            raise SystemExit(f"bench.py: rank {rank} needs GPU {local} but only {torch.cuda.device_count()} device(s) are visible "
                             "(the RawFormer HIP path has no CPU fallback)")
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ   # under torchrun the RCCL path is taken even for one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.dry_run:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: {dist.get_world_size()} ranks joined, expected {args.gpus}")

    from bayer_low_light_image_enhancement_amd import synth, tiling

    dim, batch, hm, wm, desc = WORKLOADS[args.workload]
    tiled = args.workload == "cfg4t" and world > 1
    exact = args.workload in ("cfg4", "cfg4x") and world > 1
    if args.dry_run:
        if args.workload in ("cfg4", "cfg4x", "cfg4t"):
            hm, wm = 2848 // 4 // 16 * 16, 4256 // 4 // 16 * 16     # a quarter-size frame keeps the rehearsal fast
        else:
            hm, wm = hm // 8, wm // 8
        sd = {}

        def forward(t):   # stand-in of the right shape; NOT the model (no GPU here)
            return t.repeat(1, 3, 1, 1) * 0.5

        class _StandIn:   # rehearses the collectives of RawFormer.forward_window (one sum, one max) around the stand-in forward
            def forward_window(self, win, y_lo, y_hi, total_rows, group=None):
                stat = torch.ones(64)
                dist.all_reduce(stat, group=group)
                dist.all_reduce(stat, op=dist.ReduceOp.MAX, group=group)
                return forward(win)
        model = _StandIn()
    else:
        from bayer_low_light_image_enhancement_amd import RawFormer
        model = RawFormer(dim=dim)
        sd = synthetic_state(model, 100 + dim)
        model.load_state_dict(sd, strict=True)
        model = model.to(device).eval()

        def forward(t):
            return model(t)

    trainer = None
    if args.workload == "cfg5":
        seed0 = 2 + rank * batch
        x = torch.from_numpy(synth.bayer_mosaic(seed0, batch, hm, wm)).to(device)
        gt = torch.from_numpy(synth.smooth_rgb(seed0, batch, hm, wm)).to(device)
        if args.dry_run:
            from bayer_low_light_image_enhancement_amd.train import OverlappedReducer
            flat = torch.zeros(2_477_528)
            reducer = OverlappedReducer(flat)
            bounds = [flat.numel() * k // 12 for k in range(13)]

            def step():          # rehearses the gradient all-reduce as the training step drives it: ranges announced from the end
                reducer.begin()  # of the flat buffer towards its start, buckets started as soon as they are complete
                for k in range(11, -1, -1):
                    reducer.ready(bounds[k], bounds[k + 1] - bounds[k])
                reducer.finish()
                return flat
        else:
            from bayer_low_light_image_enhancement_amd.train import Trainer
            model.train()
            trainer = Trainer(model, lr=1e-4, weight_decay=1e-2, decoupled=True, loss="l1")

            def step():
                return trainer.step(x, gt)
    elif tiled:
        # strong scaling: ONE frame (same seed on every rank), N tiles, one per rank, one all-gather
        x = torch.from_numpy(synth.bayer_mosaic(10, 1, hm, wm)).to(device)
        tiles = tiling.plan_tiles(hm, wm, TILE_GRIDS[world], overlap=TILE_OVERLAP, align=TILE_ALIGN)
        assert len(tiles) == world

        def step():
            with torch.no_grad():
                return tiling.forward_full_frame_sharded(forward, x, tiles)
    elif exact:
        x = torch.from_numpy(synth.bayer_mosaic(10, 1, hm, wm)).to(device)
        shards = tiling.plan_row_shards(hm // 2, world)

        def step():
            with torch.no_grad():
                return tiling.forward_full_frame_exact(model, x)
    else:
        # this rank's images: seeds are disjoint across ranks
        seed0 = 10 if args.workload in ("cfg4", "cfg4x", "cfg4t") else 2 + rank * batch
        x = torch.from_numpy(synth.bayer_mosaic(seed0, batch, hm, wm)).to(device)

        def step():
            with torch.no_grad():
                return forward(x)

    def fence():
        if not args.dry_run:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        if not args.dry_run:
            torch.cuda.synchronize()

    log(f"rank {rank}/{world}: model and {batch} frame(s) of {hm}x{wm} resident on {device}" + (f", {len(tiles)} tiles" if tiled else ""))
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()

    mp_per_step = (1 if (tiled or exact) else world) * batch * hm * wm / 1e6
    model_name = {32: "S", 48: "B", 64: "L"}[dim]
    line = {
        "metric": f"megapixels/sec RawFormer-{model_name} RAW->sRGB" if args.workload != "cfg2" else "megapixels/sec RawFormer-S 512x512 RAW->sRGB",
        "value": round(mp_per_step * args.steps / elapsed, 3),
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong" if args.workload in ("cfg4", "cfg4x", "cfg4t") else "weak",
        "vs_baseline": None,
        "dtype": "f32",     # storage, accumulation and results; the K >= 128 1x1 GEMMs contract three-piece bf16 splits of the f32
                            # operands on the bf16 matrix pipe (same error as the f32 MFMA chain: profiles/r02_ubench_bf16x3.txt)
        "data": "synthetic",
        "config": {"workload": desc, "name": args.workload, "frames_per_gpu": batch, "mosaic": [hm, wm], "packed": [4, hm // 2, wm // 2],
                   "variant": "flca", "weights": "synthetic (seeded, random-init scale)",
                   "parallelism": (f"tile-sharded x{world} (grid {TILE_GRIDS[world][0]}x{TILE_GRIDS[world][1]}, overlap {TILE_OVERLAP}, "
                                   f"all-gather of tile outputs inside the timed region)") if tiled else
                                  (f"exact row shards x{world} (window {shards[0].rows} of {hm // 2} packed rows, statistics all-reduced inside "
                                   f"the forward, all-gather of the interior strips; all inside the timed region)") if exact else f"batch-sharded x{world}",
                   "device": "cpu (dry run)" if args.dry_run else torch.cuda.get_device_name(local)},
    }
    if args.dry_run:
        line["dry_run"] = True
        line["data"] = "synthetic; stand-in forward on CPU (launcher rehearsal, not a measurement)"
    log(f"timed region done: {elapsed / args.steps * 1e3:.3f} ms/step")
    if args.workload == "cfg5":
        line["metric"] = "megapixels/sec RawFormer-S 512x512 training step (AdamW + L1), data-parallel"
        line["config"]["parallelism"] = (f"data-parallel x{world}: flat-gradient all-reduce in >= 4 MiB buckets, each started from inside the backward pass as "
                                         "soon as its gradients are final (rf_set_grad_ready), inside the timed region")
        line["config"]["optimizer"] = "AdamW lr 1e-4 weight_decay 1e-2; L1 loss"
        if trainer is not None:
            line["config"]["final_loss"] = round(float(out), 6)
    # The profiled pass re-runs ``step``.  When the step holds collectives (tile all-gather, the all-reduces inside
    # forward_window, the gradient buckets) EVERY rank must run it or rank 0's collectives have no partner and the job
    # hangs until the watchdog fires; only rank 0 keeps the records.  A collective-free step is profiled on rank 0 alone.
    step_has_collectives = world > 1 and (tiled or exact or args.workload == "cfg5")
    do_profile = not args.no_profile and (rank == 0 or step_has_collectives)
    recs = None
    if do_profile:
        prof_step = step if (tiled or exact or args.workload == "cfg5") else (lambda: forward(x))
        with torch.no_grad():
            recs = (dry_profile_pass if args.dry_run else profile_pass)(prof_step, args.steps)
        log(f"rank {rank}/{world}: profiled pass ran" + (" (step holds collectives: every rank takes part)" if step_has_collectives else ""))
        if step_has_collectives:
            fence()
    if rank == 0 and recs:
        total = sum(r["ms"] for r in recs)
        recs.sort(key=lambda r: -r["ms"])
        top = recs[0]
        rl = roofline_of(top)
        rl["launches_per_step"] = top["launches"] // args.steps
        rl["traffic"] = pmc_traffic(top["kernel"], args.workload)
        rl["mfma_util"] = pmc_mfma_util(top["kernel"], args.workload)
        rl["algorithmic_bytes_per_launch"] = round(top["bytes"] / top["launches"])
        rl["share_of_forward"] = round(top["ms"] / total, 4)
        line["roofline"] = rl
        rows, floor_ms, alg_bytes, pmc_bytes, pmc_cover_ms = [], 0.0, 0.0, 0.0, 0.0
        for r in recs:
            priced = bool(r["flops"] or r["bytes"])
            rf = roofline_of(r) if priced else None
            per_step = r["launches"] / args.steps
            traffic = pmc_traffic(r["kernel"], args.workload)
            rows.append({"kernel": r["kernel"], "launches_per_step": r["launches"] // args.steps,
                         "ms_per_step": round(r["ms"] / args.steps, 4), "share": round(r["ms"] / total, 4),
                         "tflops": round(r["flops"] / max(r["ms"], 1e-9) / 1e9, 2), "gbs": round(r["bytes"] / max(r["ms"], 1e-9) / 1e6, 1),
                         "bound": rf["bound"] if rf else None, "frac": rf["frac"] if rf else None,
                         "frac_hbm": rf["frac_hbm"] if rf else None, "frac_mfma": rf["frac_mfma"] if rf else None,
                         "traffic": traffic, "mfma_util": pmc_mfma_util(r["kernel"], args.workload)})
            if rf:
                floor_ms += rf["floor_us"] * 1e-3 * per_step
                alg_bytes += r["bytes"] / args.steps
            if traffic is not None:
                pmc_bytes += traffic * per_step
                pmc_cover_ms += r["ms"] / args.steps
        line["kernels"] = rows
        line["profiled_ms_per_step"] = round(total / args.steps, 4)
        # the step as a whole: every kernel at its own binding roof (sum of floors) against what was measured, and the HBM bytes
        # actually moved (PMC files of this workload, when committed) against the algorithmic bytes
        line["forward_roofline"] = {
            "sum_of_kernel_floors_ms": round(floor_ms, 4), "measured_ms_per_step": line["ms_per_step"],
            "frac_of_floors": round(floor_ms / line["ms_per_step"], 4),
            "algorithmic_gb_per_step": round(alg_bytes / 1e9, 3),
            "pmc_hbm_gb_per_step": round(pmc_bytes / 1e9, 3) if pmc_bytes else None,
            "pmc_covers_share_of_step": round(pmc_cover_ms / (total / args.steps), 4) if pmc_bytes else None,
            "hbm_time_of_pmc_bytes_ms": round(pmc_bytes / (PEAK_HBM_GBS * 1e9) * 1e3, 4) if pmc_bytes else None,
            "frac_hbm_whole_step": round(pmc_bytes / (PEAK_HBM_GBS * 1e9) * 1e3 / line["ms_per_step"], 4) if pmc_bytes else None}
        log("profiled pass done")
        if args.workload == "cfg2" and not args.dry_run:
            line["dwt_roofline"] = dwt_microbench(device)
            log("dwt microbench done")
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.dry_run:
        line["cpu_baseline"] = cpu_baseline(dim, hm, wm, sd)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
