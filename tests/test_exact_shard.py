"""Exact spatially-sharded whole frame: planner logic on the CPU, and on the GPU box two ranks (two
processes on the one GPU, gloo all-reduce of the device buffers) against the whole-frame forward."""
import os
import subprocess
import sys
import tempfile

import pytest

from bayer_low_light_image_enhancement_amd import tiling

HERE = os.path.dirname(os.path.abspath(__file__))


def test_row_shards_partition_the_frame_with_full_context():
    for rows, world, halo in ((1424, 8, 80), (1424, 2, 80), (256, 2, 80), (64, 8, 80), (712, 4, 40), (128, 3, 16)):
        shards = tiling.plan_row_shards(rows, world, halo)
        assert len(shards) == world and len({s.rows for s in shards}) == 1          # equal windows: equal slab grids
        pos = 0
        for s in shards:
            assert s.dst == pos and s.y_lo % 8 == 0 and s.y_hi % 8 == 0 and s.start % 8 == 0 and s.rows % 8 == 0
            assert s.start + s.y_lo == s.dst and 0 <= s.start and s.start + s.rows <= rows and s.y_hi <= s.rows
            # context: `halo` rows beyond the interior, or the frame border
            assert s.y_lo >= halo or s.start == 0
            assert s.rows - s.y_hi >= halo or s.start + s.rows == rows
            pos += s.y_hi - s.y_lo
        assert pos == rows


def test_row_shards_reject_bad_sizes():
    with pytest.raises(ValueError):
        tiling.plan_row_shards(100, 2)          # not a multiple of 8
    with pytest.raises(ValueError):
        tiling.plan_row_shards(16, 4)           # fewer than 8 rows per shard


def _run_two_ranks(variant, rows, cols, dim, halo):
    import json
    world = 2
    with tempfile.TemporaryDirectory() as d:
        rdv = os.path.join(d, "rdv")
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "shard_worker.py"), str(r), str(world), rdv, str(rows), str(cols),
                                   str(dim), variant, str(halo)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                 for r in range(world)]
        outs = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=300)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            outs.append(o)
    recs = []
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        recs.append(json.loads([ln for ln in o.splitlines() if ln.startswith("{")][-1]))
    return recs


@pytest.mark.gpu
@pytest.mark.parametrize("variant,rows,cols,dim,halo", [("flca", 384, 64, 16, tiling.HALO_ROWS), ("plain", 352, 40, 16, tiling.HALO_ROWS),
                                                         ("flca", 512, 128, 32, tiling.HALO_ROWS), ("flca", 384, 64, 16, 40)])
def test_two_rank_exact_shard_matches_the_oracle_whole_frame(variant, rows, cols, dim, halo):
    """The stitched frame against the CPU oracle's WHOLE-frame forward (pinned to the reference; the reference itself has no
    tiling: test.py:72,116), tolerance of the whole-model tests; and against the HIP whole-frame forward to fp32 summation
    order.  With two ranks every window touches a frame border, so it holds 2 * halo rows of one-sided context: the case
    halo = 40 is the minimal legal context (80 rows >= the receptive field of 77)."""
    recs = _run_two_ranks(variant, rows, cols, dim, halo)
    r0 = recs[0]
    print(r0)
    assert r0["err_vs_oracle"] is not None and r0["err_vs_oracle"] <= 5e-5 * max(r0["scale"], 1.0), r0
    for r in recs:
        assert r["err_vs_hip_whole"] <= 5e-5 * max(r["scale"], 1.0), r            # measured 4e-6 .. 8e-6
        assert r["err_independent_tiles"] > 20 * r["err_vs_hip_whole"], r          # the statistics, not the context, make it exact


@pytest.mark.gpu
def test_too_little_context_is_visible():
    """8 rows of halo = 16 rows of one-sided context, far inside the receptive field: the frame must NOT match (guards the
    comparison itself against passing vacuously)."""
    recs = _run_two_ranks("flca", 384, 64, 16, 8)
    assert max(r["err_vs_hip_whole"] for r in recs) > 1e-4, recs


@pytest.mark.gpu
def test_single_rank_on_rccl_takes_the_nccl_branch_of_the_callback():
    """Two ranks cannot share the one GPU of the test box under RCCL, so the two-rank tests above run their collectives over
    gloo.  One rank CAN initialise the ``nccl`` backend: every statistics all-reduce of the sharded forward (and the final
    all-gather) then goes through RCCL on the launch stream -- with a world of one they are identities, which is exactly what
    the result must show: the sharded call equals the whole-frame forward bit for bit, and the oracle within tolerance."""
    import json
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, RF_SHARD_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
        p = subprocess.run([sys.executable, os.path.join(HERE, "shard_worker.py"), "0", "1", os.path.join(d, "rdv"), "128", "64", "16", "flca",
                            str(tiling.HALO_ROWS)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["err_vs_hip_whole"] == 0.0, r
    assert r["err_vs_oracle"] <= 5e-5 * max(r["scale"], 1.0), r
