#!/usr/bin/env python3
"""RawFormer-S inference throughput on MI355X: megapixels/s, RAW -> sRGB.

A step = one ``RawFormer.forward`` (HIP path) over one batch of synthetic low-light Bayer
frames already resident in HBM.  The N = 1 workload is BASELINE.json configs[1]:
RawFormer-S (dim 32), batch 8 of packed 4x512x512 (mosaic 1x1024x1024).  With ``--gpus N``
(launched by torch.distributed.run, one rank per GPU) every rank runs the same per-GPU batch
on its own images: the path shards along the batch with no data-path collective (weak
scaling); the only collectives are the timing barrier and the MAX over ranks.

Rank 0 prints ONE JSON line.  Besides the driver's fields it carries
  roofline      the dominant kernel (largest share of the forward), timed live with HIP events
                on the launch stream (rf_profile_begin/end) over the same K steps in a second,
                untimed pass, priced against its algorithmic FLOPs;
  cpu_baseline  the CPU oracle (oracle/rawformer_ref.py, torch CPU ops, all host cores) on a
                bounded sample of the same workload, rank 0 at N = 1 only;
  kernels       the per-kernel-class breakdown of that profiled pass (share of forward time);
  dwt_roofline  the stand-alone DWT / IDWT kernels at the stage-0 activation size.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* = f32 vector rate
PEAK_HBM_GBS = 8000.0          # HBM3E spec

WORKLOADS = {
    # name: (dim, batch per GPU, mosaic H, mosaic W, description)
    "cfg2": (32, 8, 1024, 1024, "RawFormer-S(FLCA) dim=32, batch=8 of packed 4x512x512 synthetic Bayer per GPU"),
    "cfg1": (32, 1, 256, 256, "RawFormer-S(FLCA) dim=32, one packed 4x128x128 frame"),
    "cfg3": (48, 8, 1024, 1024, "RawFormer-B(FLCA) dim=48, batch=8 of packed 4x512x512 synthetic Bayer per GPU"),
}


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


def roofline_of(rec):
    ms = rec["ms"] / rec["launches"]
    flops, byts = rec["flops"] / rec["launches"], rec["bytes"] / rec["launches"]
    intensity = flops / max(byts, 1.0)
    if intensity >= PEAK_F32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
        ach = flops / (ms * 1e-3) / 1e12
        return {"kernel": rec["kernel"], "bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": None}
    ach = byts / (ms * 1e-3) / 1e9
    return {"kernel": rec["kernel"], "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None, "avg_launch_us": round(ms * 1e3, 2),
            "launches_per_step": None}


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of ``kernel`` from the committed rocprofv3 PMC passes of this same command
    (profiles/*_traffic.json, made by tools/traffic.py: FETCH_SIZE and WRITE_SIZE in separate passes, gfx950
    corrections applied).  PMC counters cannot be read from inside the process; None when the workload is not
    the profiled one or the kernel is not in the file."""
    if workload != "cfg2":
        return None
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_traffic.json")))
    if not files:
        return None
    try:
        rec = json.load(open(files[-1]))["kernels"].get(kernel)
        return rec["hbm_bytes_per_launch"] if rec else None
    except Exception:
        return None


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


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(dim, hm, wm, sd):
    """Oracle forward on the host cores, bounded to roughly 10-30 s of CPU work."""
    import torch
    from bayer_low_light_image_enhancement_amd import synth
    from oracle import rawformer_ref as R

    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = R.RawFormerConfig(dim=dim)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    x1 = torch.from_numpy(synth.bayer_mosaic(900, 1, hm, wm))
    with torch.no_grad():
        t0 = time.perf_counter()
        R.rawformer_forward(sd_cpu, x1, cfg)          # warm-up, also sizes the sample
        t_one = time.perf_counter() - t0
        log(f"cpu_baseline: warm-up frame took {t_one:.2f} s on {cores} threads")
        if t_one > 15.0:   # slow host: the warm-up frame itself is the bounded sample
            return {"value": round(hm * wm / 1e6 / t_one, 4), "unit": "MP/s", "cores": cores, "kind": "port",
                    "sample": f"1 frame of 1x{hm}x{wm} mosaic, torch CPU fp32 oracle, single cold run, {cores} threads"}
        nimg = int(max(1, min(8, 12.0 // max(t_one, 1e-3))))
        x = torch.from_numpy(synth.bayer_mosaic(901, nimg, hm, wm))
        times = []
        for _ in range(2):
            t0 = time.perf_counter()
            R.rawformer_forward(sd_cpu, x, cfg)
            times.append(time.perf_counter() - t0)
    best = min(times)
    return {"value": round(nimg * hm * wm / 1e6 / best, 4), "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": f"{nimg} frame(s) of 1x{hm}x{wm} mosaic, torch CPU fp32 oracle, best of 2 after warm-up, {cores} threads"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the RawFormer HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ   # under torchrun the RCCL path is taken even for one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=device)

    from bayer_low_light_image_enhancement_amd import RawFormer, synth
    from oracle.rawformer_ref import RawFormerConfig, param_shapes   # shapes only; the oracle is not run here

    dim, batch, hm, wm, desc = WORKLOADS[args.workload]
    shapes = param_shapes(RawFormerConfig(dim=dim))
    sd = {k: torch.from_numpy(synth.param_values(100 + dim, k, s)).reshape(s) for k, s in shapes.items()}
    model = RawFormer(dim=dim)
    model.load_state_dict(sd, strict=False)
    model = model.to(device).eval()
    # this rank's images: seeds are disjoint across ranks
    x = torch.from_numpy(synth.bayer_mosaic(2 + rank * batch, batch, hm, wm)).to(device)

    def step():
        with torch.no_grad():
            return model(x)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: model and {batch} frame(s) of {hm}x{wm} resident on {device}")
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

    mp_per_step = world * batch * hm * wm / 1e6
    line = {
        "metric": "megapixels/sec RawFormer-S 512x512 RAW->sRGB",
        "value": round(mp_per_step * args.steps / elapsed, 3),
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": desc, "frames_per_gpu": batch, "mosaic": [hm, wm], "packed": [4, hm // 2, wm // 2],
                   "variant": "flca", "weights": "synthetic (seeded, random-init scale)", "parallelism": f"batch-sharded x{world}",
                   "device": torch.cuda.get_device_name(local)},
    }
    log(f"timed region done: {elapsed / args.steps * 1e3:.3f} ms/step")
    if rank == 0 and not args.no_profile:
        recs = profile_pass(step, args.steps)
        total = sum(r["ms"] for r in recs)
        recs.sort(key=lambda r: -r["ms"])
        top = recs[0]
        rl = roofline_of(top)
        rl["launches_per_step"] = top["launches"] // args.steps
        rl["traffic"] = pmc_traffic(top["kernel"], args.workload)
        rl["algorithmic_bytes_per_launch"] = round(top["bytes"] / top["launches"])
        rl["share_of_forward"] = round(top["ms"] / total, 4)
        if top["kernel"].startswith("conv3x3_kernel"):
            # `achieved` prices the ALGORITHMIC flops (18 Cin Cout per pixel); the kernel uses the Winograd
            # F(4,3) form along x and issues half of them as MFMA work, so the matrix pipe itself runs at frac / 2 of peak
            rl["mfma_flops_issued_over_algorithmic"] = 0.5
            rl["frac_of_peak_issued"] = round(rl["frac"] * 0.5, 4)
        line["roofline"] = rl
        line["kernels"] = [{"kernel": r["kernel"], "launches_per_step": r["launches"] // args.steps,
                            "ms_per_step": round(r["ms"] / args.steps, 4), "share": round(r["ms"] / total, 4),
                            "tflops": round(r["flops"] / max(r["ms"], 1e-9) / 1e9, 2),
                            "gbs": round(r["bytes"] / max(r["ms"], 1e-9) / 1e6, 1)} for r in recs]
        line["profiled_ms_per_step"] = round(total / args.steps, 4)
        log("profiled pass done")
        line["dwt_roofline"] = dwt_microbench(device)
        log("dwt microbench done")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(dim, hm, wm, sd)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
