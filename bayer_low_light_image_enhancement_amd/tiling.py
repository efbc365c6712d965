"""Image-parallel inference: batch sharding and full-frame tiling across ranks.

The reference has no tiling or multi-GPU inference at all (test.py:72 runs whole frames with
batch_size=1; training uses nn.DataParallel, train.py:108-111).  RawFormer is per-image
everywhere (attention, squeeze-excite and the luma maximum reduce over ONE image), so

* a batch shards across ranks with no communication (``shard_batch``), and
* a full frame can be cut into overlapping tiles that are each an independent "image"
  (``plan_tiles`` / ``forward_tiled``).  Independent tiles change the per-image statistics, so
  the parity oracle for tiled mode is *the same forward run on the same tiles*
  (SURVEY.md section 8e), not the whole-frame forward.

``forward_full_frame_exact`` is the exact alternative: row shards with ``HALO_ROWS`` rows of recomputed context whose global
statistics are all-reduced inside the forward (``RawFormer.forward_window``, C ABI ``rf_set_shard``), so the stitched frame
equals the whole-frame forward up to summation order.

``forward_full_frame_sharded`` runs rank r's tiles on rank r and stitches the sRGB frame on
every rank with ONE all-gather of fixed-size tile outputs (RCCL over xGMI with the ``nccl``
backend; ``gloo`` in the CPU tests).  Nothing here touches the HIP library: the per-tile
forward is any callable ``[b,1,h,w] -> [b,3,h,w]`` (normally ``RawFormer.forward``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Sequence, Tuple

import torch

ALIGN = 16  # mosaic sizes must be multiples of 16 (Bayer pack + three 2x down-samplings)


@dataclass(frozen=True)
class Tile:
    """A tile in MOSAIC coordinates.  ``src`` is what the model sees (multiple of 16, inside the
    frame), ``dst`` the part of the frame it is responsible for, ``crop`` = dst relative to src."""
    src: Tuple[int, int, int, int]   # y0, x0, h, w
    dst: Tuple[int, int, int, int]
    crop: Tuple[int, int, int, int]


def _cuts(n: int, parts: int) -> List[int]:
    """Frame split points on even (Bayer-phase preserving) coordinates."""
    pts = [int(round(i * n / parts / 2.0)) * 2 for i in range(parts + 1)]
    pts[0], pts[-1] = 0, n
    return pts


def _span(lo: int, hi: int, n: int, overlap: int, align: int = ALIGN) -> Tuple[int, int]:
    """Source interval covering [lo, hi) plus ``overlap`` on each side, length a multiple of
    ``align`` (of ALIGN when the aligned length would not fit the frame), start even, clipped to [0, n)."""
    a = max(0, lo - overlap)
    a -= a % 2
    b = min(n, hi + overlap)
    length = -(-(b - a) // align) * align
    if length > n:
        length = -(-(b - a) // ALIGN) * ALIGN
    if length > n:
        raise ValueError(f"frame side {n} is not a multiple of {ALIGN} and too small to tile")
    if a + length > n:            # grow to the left instead of past the border
        a = n - length
        a -= a % 2
    return a, length


def plan_tiles(height: int, width: int, grid: Tuple[int, int], overlap: int = 32, align: int = ALIGN) -> List[Tile]:
    """Cut a ``height x width`` mosaic into ``grid = (rows, cols)`` tiles with ``overlap`` mosaic
    pixels of context on interior edges.  Works for sizes such as 2848x4256 (SID Sony) whose
    eighth/sixteenth parts are not integers: tile sources are grown to a multiple of ``align``
    (16 is what the model needs; 64 keeps every U-Net level of a tile a multiple of 4 pixels wide,
    i.e. on the 16-byte vector paths of the kernels)."""
    if height % 2 or width % 2:
        raise ValueError("mosaic size must be even")
    if align % ALIGN:
        raise ValueError(f"align must be a multiple of {ALIGN}")
    ys, xs = _cuts(height, grid[0]), _cuts(width, grid[1])
    tiles = []
    for i in range(grid[0]):
        for j in range(grid[1]):
            y0, hh = _span(ys[i], ys[i + 1], height, overlap, align)
            x0, ww = _span(xs[j], xs[j + 1], width, overlap, align)
            dst = (ys[i], xs[j], ys[i + 1] - ys[i], xs[j + 1] - xs[j])
            tiles.append(Tile((y0, x0, hh, ww), dst, (dst[0] - y0, dst[1] - x0, dst[2], dst[3])))
    return tiles


def shard_batch(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced share of ``n_items`` images for ``rank`` (may be empty)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def forward_tiled(forward: Callable[[torch.Tensor], torch.Tensor], x: torch.Tensor, tiles: Sequence[Tile],
                  out_channels: int = 3) -> torch.Tensor:
    """Single-process tiled forward: every tile is an independent image."""
    b, _, h, w = x.shape
    out = x.new_empty((b, out_channels, h, w))
    for t in tiles:
        y0, x0, hh, ww = t.src
        o = forward(x[:, :, y0:y0 + hh, x0:x0 + ww].contiguous())
        cy, cx, ch, cw = t.crop
        out[:, :, t.dst[0]:t.dst[0] + ch, t.dst[1]:t.dst[1] + cw] = o[:, :, cy:cy + ch, cx:cx + cw]
    return out


def forward_full_frame_sharded(forward: Callable[[torch.Tensor], torch.Tensor], x: torch.Tensor,
                               tiles: Sequence[Tile], group=None, out_channels: int = 3) -> torch.Tensor:
    """Tile-parallel full-frame forward.  Every rank holds the whole mosaic ``x`` (it is 1/3 the
    size of the output), runs tiles ``rank, rank + world, ...`` and receives all tile outputs with
    one all-gather; the stitched frame is returned on every rank.

    Tile outputs are padded to the largest tile so the collective has a fixed shape
    (config 4: 8 tiles of 3 x ~1.5k x ~1.1k fp32, about 18 MB per rank).
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b, _, h, w = x.shape
    per_rank = -(-len(tiles) // world)
    mh = max(t.src[2] for t in tiles)
    mw = max(t.src[3] for t in tiles)
    mine = x.new_zeros((per_rank, b, out_channels, mh, mw))
    for slot in range(per_rank):
        idx = rank + slot * world
        if idx < len(tiles):
            y0, x0, hh, ww = tiles[idx].src
            mine[slot, :, :, :hh, :ww] = forward(x[:, :, y0:y0 + hh, x0:x0 + ww].contiguous())
    gathered = x.new_empty((world,) + tuple(mine.shape))
    dist.all_gather_into_tensor(gathered.view(world * per_rank, b, out_channels, mh, mw), mine, group=group)
    out = x.new_empty((b, out_channels, h, w))
    for idx, t in enumerate(tiles):
        r, slot = idx % world, idx // world
        cy, cx, ch, cw = t.crop
        out[:, :, t.dst[0]:t.dst[0] + ch, t.dst[1]:t.dst[1] + cw] = gathered[r, slot, :, :, cy:cy + ch, cx:cx + cw]
    return out


# ---- exact row sharding ---------------------------------------------------------------------------------------------
# Receptive field of RawFormer along one axis, in packed (level-0) rows per side, when the global statistics are exact:
# embedding 3x3 (1) + per stage {qkv depthwise 3x3, FFN depthwise 3x3, Conv_out 3x3} = 3 rows of its level + each
# down-sampling 3x3 (1 row of its level): encoder 5 + 4*2 + 4*4 + 3*8, decoder 3*4 + 3*2 + 3, output 3x3 1 -> 75 rows for the
# convolution chain.  FLCA's gates sit in parallel with the TransformerBlock (3x3 of their level on guidance planes that are
# Haar bands of the luma, bilinearly resampled to the stage: up to 2 more level-0 rows at level 0); interval propagation
# through the whole U-Net from 8-aligned cuts gives 77.  HALO_ROWS = 80 = the next multiple of 8: a margin of 3 rows.
HALO_ROWS = 80


@dataclass(frozen=True)
class RowShard:
    """One rank's share of a frame, in PACKED rows (mosaic rows / 2)."""
    start: int      # first row of the window the rank runs
    rows: int       # window height (equal on all ranks)
    y_lo: int       # interior rows [y_lo, y_hi) inside the window ...
    y_hi: int
    dst: int        # ... which are rows [dst, dst + y_hi - y_lo) of the frame


def plan_row_shards(packed_rows: int, world: int, halo: int = HALO_ROWS) -> List[RowShard]:
    """Cut ``packed_rows`` (a multiple of 8: every U-Net level then has whole rows per shard) into ``world`` interiors on
    8-row boundaries, each inside an equally tall window that extends ``halo`` rows beyond the interior or to the frame
    border.  Equal windows give every rank the same slab grid, which is what lets the partial-sum buffers be all-reduced
    element-wise."""
    if packed_rows % 8 or halo % 8 or halo < 0:
        raise ValueError("packed rows and halo must be multiples of 8")
    units = packed_rows // 8
    if world < 1 or units < world:
        raise ValueError(f"{packed_rows} packed rows cannot be cut into {world} shards of at least 8 rows")
    base, extra = divmod(units, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + 8 * (base + (1 if r < extra else 0)))
    rows = min(packed_rows, 8 * (base + (1 if extra else 0)) + 2 * halo)
    shards = []
    for r in range(world):
        r0, r1 = bounds[r], bounds[r + 1]
        start = min(max(r0 - halo, 0), packed_rows - rows)
        shards.append(RowShard(start, rows, r0 - start, r1 - start, r0))
    return shards


def forward_full_frame_exact(model, x: torch.Tensor, group=None, halo: int = HALO_ROWS) -> torch.Tensor:
    """Row-sharded full-frame forward whose result equals the whole-frame forward (up to fp32 summation order): rank r runs
    ``model.forward_window`` on its window of the mosaic ``x`` ``[B,1,h,w]`` (h divisible by 16), the statistics are
    all-reduced inside the forward, and one all-gather of the interior strips stitches the sRGB frame on every rank."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b, _, h, w = x.shape
    if h % 16:
        raise ValueError(f"mosaic height {h} must be divisible by 16")
    shards = plan_row_shards(h // 2, world, halo)
    me = shards[rank]
    win = x[:, :, 2 * me.start: 2 * (me.start + me.rows), :].contiguous()
    o = model.forward_window(win, me.y_lo, me.y_hi, h // 2, group=group)
    mh = max(s.y_hi - s.y_lo for s in shards)
    mine = o.new_zeros((b, o.shape[1], 2 * mh, w))
    mine[:, :, : 2 * (me.y_hi - me.y_lo)] = o[:, :, 2 * me.y_lo: 2 * me.y_hi]
    gathered = o.new_empty((world,) + tuple(mine.shape))
    dist.all_gather_into_tensor(gathered.view(world * b, o.shape[1], 2 * mh, w), mine, group=group)
    out = o.new_empty((b, o.shape[1], h, w))
    for r, s in enumerate(shards):
        n = 2 * (s.y_hi - s.y_lo)
        out[:, :, 2 * s.dst: 2 * s.dst + n] = gathered[r, :, :, :n]
    return out
