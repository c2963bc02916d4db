"""Multi-GPU frame rendering: ray tiles sharded across ranks, one gather per frame.

Rays are independent, so there is no data-path collective: each rank renders the tiles t with
``t % world == rank`` (interleaved, because human-hit rays cluster in the image centre and contiguous row
blocks would be unbalanced) into a local [n_local, C] buffer; rank 0 receives one padded buffer per rank through
``torch.distributed.gather`` (RCCL over xGMI with backend "nccl", gloo in the CPU tests) and permutes them into
the frame.  The reference itself never shards renders (single device, sequential `rays_per_batch` loop).
"""
import os
import time

import torch
import torch.distributed as dist

# tile of the frame renderers' sharding (render_frame_sharded): the local rays of a rank are rendered as ONE ray list whatever
# the tile is, so the tile only sets how finely the ranks interleave across the image -- 1024 rays is less than an image row of
# the human configurations, so every rank's rays cross the body (hit rays cluster in the image centre)
FRAME_TILE = int(os.environ.get("NEUMAN_FRAME_TILE", "1024"))
# run the collective path on an initialised process group of ONE rank as well (what a one-GPU box can execute of the N > 1 path)
FORCE_COLLECTIVE = os.environ.get("NEUMAN_FORCE_COLLECTIVE", "0") == "1"


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


_INDEX_CACHE = {}


def tile_ray_indices(total_rays, tile, rank, world, device='cpu'):
    """Global ray indices owned by `rank`: tiles rank, rank+world, ... of `tile` consecutive rays each.  Cached: the frame
    assembly on rank 0 asks for every rank's list once per frame, and the mask below costs a host sync on a GPU."""
    key = (int(total_rays), int(tile), int(rank), int(world), str(device))
    hit = _INDEX_CACHE.get(key)
    if hit is not None:
        return hit
    n_tiles = (total_rays + tile - 1) // tile
    # (a rank beyond the last tile owns nothing: torch.arange refuses start > end -- a frame of fewer tiles than ranks, found by the world-8 test)
    tiles = torch.arange(rank, n_tiles, world, device=device) if rank < n_tiles else torch.zeros(0, dtype=torch.int64, device=device)
    idx = (tiles[:, None] * tile + torch.arange(tile, device=device)[None, :]).reshape(-1)
    idx = idx[idx < total_rays]
    if len(_INDEX_CACHE) > 256:
        _INDEX_CACHE.clear()
    _INDEX_CACHE[key] = idx
    return idx


def balanced_tile(total_rays, world, max_tile=8192):
    """Largest tile <= max_tile that gives every rank the same number of tiles (n_tiles a multiple of `world`): with a fixed
    8192 an 800x800 frame is 79 tiles, 10 / 9 per rank at world 8 -- 2.4 % of the frame time spent waiting for the ranks that
    got 10; 80 tiles of 8000 are 10 each.  Only the last tile can be short, by less than n_tiles rays."""
    per_round = world * max_tile
    n_tiles = world * ((total_rays + per_round - 1) // per_round)
    tile = max(1, (total_rays + n_tiles - 1) // n_tiles)
    while tile > 1 and ((total_rays + tile - 1) // tile) % world:      # (rounding up can lose a tile on tiny frames)
        tile -= 1
    return tile


def max_local_rays(total_rays, tile, world):
    n_tiles = (total_rays + tile - 1) // tile
    return ((n_tiles + world - 1) // world) * tile


def frame_source_rows(total_rays, tile, world, device='cpu'):
    """For every global ray, its row in the concatenation of the ranks' padded shards ([world * cap, C], cap =
    max_local_rays): the whole frame assembly on the destination rank is ONE index_select with this list.  Cached."""
    key = ("rows", int(total_rays), int(tile), int(world), str(device))
    hit = _INDEX_CACHE.get(key)
    if hit is not None:
        return hit
    cap = max_local_rays(total_rays, tile, world)
    rows = torch.empty(total_rays, dtype=torch.int64, device=device)
    for r in range(world):
        idx = tile_ray_indices(total_rays, tile, r, world, device=device)
        rows[idx] = r * cap + torch.arange(idx.shape[0], device=device)
    _INDEX_CACHE[key] = rows
    return rows


def gather_frame(local, local_idx, total_rays, tile, dst=0, force_collective=False):
    """Assemble [total_rays, C] on rank `dst` from every rank's (local values, global indices).

    local [n_local, C] float32; returns the frame on `dst`, None elsewhere.  One collective per frame; the payload
    is padded to the largest shard so all ranks send equal sizes (a requirement of gather on RCCL); the receive buffers
    are slices of one [world * cap, C] tensor, which one index_select turns into the frame.  `force_collective` runs the
    collective path at world size 1 too (an initialised process group of one rank: what the single-GPU test box can execute).
    """
    rank, world = rank_world()
    if world == 1 and not (force_collective and dist.is_available() and dist.is_initialized()):
        out = torch.empty((total_rays, local.shape[1]), device=local.device, dtype=local.dtype)
        out[local_idx] = local
        return out
    cap = max_local_rays(total_rays, tile, world)
    home = local.device
    if local.is_cuda and dist.get_backend() == "gloo":          # ranks sharing a GPU in the tests: gloo moves host tensors
        local = local.cpu()
    send = torch.zeros((cap, local.shape[1]), device=local.device, dtype=local.dtype)
    send[:local.shape[0]] = local
    big = torch.empty((world, cap, local.shape[1]), device=local.device, dtype=local.dtype) if rank == dst else None
    dist.gather(send, gather_list=list(big.unbind(0)) if rank == dst else None, dst=dst)
    if rank != dst:
        return None
    rows = frame_source_rows(total_rays, tile, world, device=home)
    return big.reshape(world * cap, local.shape[1]).to(home).index_select(0, rows)


def render_sharded(render_rays_fn, origins, dirs, tile=8192, dst=0, force_collective=False):
    """Render a frame's rays across all ranks.

    render_rays_fn(o [n,3], d [n,3]) -> [n, C] on the same device.  `origins`/`dirs` are the full frame's rays
    (every rank holds them; they are tiny next to the compute).  Returns [total, C] on rank dst, None elsewhere.
    """
    rank, world = rank_world()
    total = origins.shape[0]
    idx = tile_ray_indices(total, tile, rank, world, device=origins.device)
    local = render_rays_fn(origins[idx].contiguous(), dirs[idx].contiguous())
    return gather_frame(local, idx, total, tile, dst, force_collective)


# The frame renderers of render_utils shard only when asked to: under a process group they become COLLECTIVES (every rank must make the
# call, ranks other than 0 get None back), which a caller that renders its validation frames on rank 0 alone -- the usual pattern beside
# data-parallel training (bkg_trainer / human_trainer under a group) -- must not get by surprise.  NEUMAN_SHARD_FRAMES=1 or
# set_frame_sharding(True) turns it on (bench.py and tools/bench_configs.py do; a torchrun render driver must too -- left off under a group of
# more than one rank, every rank renders the whole frame and sharding_active() says so once, on stderr).
SHARD_FRAMES = os.environ.get("NEUMAN_SHARD_FRAMES", "0") == "1"
# per-frame timing of render_frame_sharded (HIP events on the current stream, read by frame_stats(); no host synchronisation in the call)
FRAME_STATS = os.environ.get("NEUMAN_FRAME_STATS", "0") == "1"


def set_frame_sharding(on=True, stats=None):
    """Opt the four frame renderers in to (out of) sharding under the initialised process group; `stats`: also time every frame."""
    global SHARD_FRAMES, FRAME_STATS
    SHARD_FRAMES = bool(on)
    if stats is not None:
        FRAME_STATS = bool(stats)


def sharding_active():
    """True when the frame renderers shard: asked for (SHARD_FRAMES) and an initialised process group of more than one rank (or of one,
    with FORCE_COLLECTIVE)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    if not SHARD_FRAMES:
        global _WARNED_UNSHARDED
        if not _WARNED_UNSHARDED and dist.get_world_size() > 1:
            _WARNED_UNSHARDED = True
            import sys
            print(f"neuman_hip.parallel: a process group of {dist.get_world_size()} ranks is up but frame sharding is off: every rank renders the "
                  "WHOLE frame (and gets it back).  Render drivers set NEUMAN_SHARD_FRAMES=1 or parallel.set_frame_sharding(True); trainers that "
                  "render validation frames on one rank leave it off.", file=sys.stderr, flush=True)
        return False
    return dist.get_world_size() > 1 or FORCE_COLLECTIVE


LAST_FRAME_STATS = {}
_WARNED_UNSHARDED = False


def frame_stats():
    """LAST_FRAME_STATS with the timings resolved to milliseconds (waits for the frame's last event when they are HIP events)."""
    st = dict(LAST_FRAME_STATS)
    ev = st.pop("_events", None)
    if ev is not None:
        ev[2].synchronize()
        st["render_ms"], st["gather_ms"] = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    return st


def render_frame_sharded(rays_fn, origins, dirs, max_tile=None, dst=0):
    """What render_vanilla / render_smpl_nerf / render_hybrid_nerf / render_hybrid_nerf_multi_persons call under a process group when
    sharding is on (reference render_test_views.py:70-90, render_gathering.py:186-200 render one frame on one device; here the frame's
    rays are split over the ranks).  A COLLECTIVE: every rank of the group must call it with the same frame.

    rays_fn(o [n,3], d [n,3]) -> tuple of per-ray tensors ([n] or [n,C]) on the rays' device.  Every rank holds the frame's rays,
    the weights and the posed meshes; it renders the rays of its interleaved tiles as ONE list (so the renderers' own batching, hit
    compaction and C calls see a smaller frame, nothing else), the columns travel as one [n_local, sum C] buffer through ONE gather,
    and rank `dst` gets the tuple for the whole frame; the other ranks get None.  Rays are independent: the frame is bit-identical
    to the unsharded one (tests/test_parallel_gpu.py).  A rank without rays (fewer tiles than ranks) renders one ray for the column
    layout and sends none.  LAST_FRAME_STATS: this rank's tile / ray counts, and with FRAME_STATS its timings (frame_stats())."""
    rank, world = rank_world()
    total = origins.shape[0]
    tile = balanced_tile(total, world, max_tile or FRAME_TILE)
    idx = tile_ray_indices(total, tile, rank, world, device=origins.device)
    timed = FRAME_STATS
    cuda = origins.is_cuda
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if (timed and cuda) else None
    t = [0.0, 0.0, 0.0]

    def mark(k):
        if ev is not None:
            ev[k].record()
        elif timed:
            t[k] = time.perf_counter()
    mark(0)
    n_local = int(idx.shape[0])
    take = idx if n_local else torch.zeros(1, dtype=idx.dtype, device=idx.device)
    outs = rays_fn(origins[take].contiguous(), dirs[take].contiguous())
    cols = [x.reshape(x.shape[0], -1).to(torch.float32) for x in outs]
    local = torch.cat(cols, 1) if len(cols) > 1 else cols[0]
    if not n_local:
        local = local[:0]
    mark(1)
    frame = gather_frame(local, idx, total, tile, dst, force_collective=FORCE_COLLECTIVE)
    mark(2)
    LAST_FRAME_STATS.clear()
    LAST_FRAME_STATS.update(rank=rank, world=world, tile=tile, tiles=(n_local + tile - 1) // tile, rays=n_local)
    if ev is not None:
        LAST_FRAME_STATS["_events"] = ev
    elif timed:
        LAST_FRAME_STATS.update(render_ms=(t[1] - t[0]) * 1e3, gather_ms=(t[2] - t[1]) * 1e3)
    if frame is None:
        return None
    res, c0 = [], 0
    for x, c in zip(outs, cols):
        w = c.shape[1]
        part = frame[:, c0:c0 + w]
        res.append(part.reshape(total, *x.shape[1:]))
        c0 += w
    return tuple(res)
