"""Multi-rank ray-tile sharding + frame gather, on CPU with gloo (world_size 1, 2, 3, 4 and 8: the node's real size)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neuman_hip import parallel


def test_tile_assignment_is_a_partition():
    for total, tile, world in [(4096, 256, 2), (1000, 64, 4), (640000, 8192, 8), (5, 8, 2)]:
        seen = torch.cat([parallel.tile_ray_indices(total, tile, r, world) for r in range(world)])
        assert seen.numel() == total and torch.equal(torch.sort(seen)[0], torch.arange(total))
        sizes = [parallel.tile_ray_indices(total, tile, r, world).numel() for r in range(world)]
        assert max(sizes) <= parallel.max_local_rays(total, tile, world)
        assert max(sizes) - min(sizes) <= tile


def test_balanced_tile_gives_every_rank_the_same_tile_count():
    for total, world, cap in [(640000, 8, 8192), (640000, 4, 8192), (640000, 2, 8192), (640000, 1, 8192), (262144, 8, 8192),
                              (2073600, 8, 8192), (1000, 4, 64), (7, 8, 8192)]:
        tile = parallel.balanced_tile(total, world, cap)
        assert 1 <= tile <= cap
        n_tiles = (total + tile - 1) // tile
        sizes = [parallel.tile_ray_indices(total, tile, r, world).numel() for r in range(world)]
        assert sum(sizes) == total
        if total >= world * world:
            assert n_tiles % world == 0, (total, world, tile, n_tiles)
            assert max(sizes) - min(sizes) < n_tiles          # only the last tile is short
    assert parallel.balanced_tile(640000, 8, 8192) == 8000


def _fake_render(o, d):
    # deterministic per-ray "pixel": depends only on the ray, like the real renderer
    return torch.stack([o[:, 0] + d[:, 1], o[:, 1] * d[:, 2], d[:, 0], (o * d).sum(-1)], 1)


def _worker(rank, world, port, total, tile, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    o = torch.randn(total, 3, generator=g)
    d = torch.randn(total, 3, generator=g)
    frame = parallel.render_sharded(_fake_render, o, d, tile=tile, dst=0)
    if rank == 0:
        q.put(frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total,tile", [(2, 1000, 64), (4, 777, 32), (8, 5000, 128)])
def test_sharded_render_matches_single_rank(world, total, tile):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, tile, q)) for r in range(world)]
    for p in procs:
        p.start()
    frame = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    o = torch.randn(total, 3, generator=g)
    d = torch.randn(total, 3, generator=g)
    np.testing.assert_array_equal(frame, _fake_render(o, d).numpy())


def _fake_frame(o, d):
    # what the renderers hand to render_frame_sharded: a tuple of per-ray tensors of different widths (rgb, depth, acc, a per-ray flag)
    return (torch.stack([o[:, 0] + d[:, 1], o[:, 1] * d[:, 2], d[:, 0]], 1), (o * d).sum(-1), o[:, 2:3] - d[:, 2:3], (o[:, 0] > 0).float())


def _frame_worker(rank, world, port, total, max_tile, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1)
    o = torch.randn(total, 3, generator=g)
    d = torch.randn(total, 3, generator=g)
    out = parallel.render_frame_sharded(_fake_frame, o, d, max_tile=max_tile, dst=0)
    st = dict(parallel.LAST_FRAME_STATS)
    assert st['rank'] == rank and st['world'] == world and (st['rays'] > 0 or total < world)
    if rank == 0:
        q.put([x.numpy() for x in out] + [st['tile']])
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total,max_tile", [(2, 1003, 64), (3, 500, 1024), (3, 2, 64),      # (3, 2): a rank without rays
                                                  (8, 6401, 100), (8, 5, 64)])                    # the node's size: 8 padded shards; three ranks without rays
def test_frame_drivers_shard_and_assemble_tuples(world, total, max_tile):
    """parallel.render_frame_sharded (what render_vanilla / render_smpl_nerf / render_hybrid_nerf / render_hybrid_nerf_multi_persons call
    under a process group): interleaved tiles, the tuple's columns through one gather, the frame on rank 0 only, bit-identical"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_frame_worker, args=(r, world, port, total, max_tile, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(1)
    o = torch.randn(total, 3, generator=g)
    d = torch.randn(total, 3, generator=g)
    want = _fake_frame(o, d)
    assert 1 <= got[-1] <= max_tile
    assert len(got) - 1 == len(want)
    for a, b in zip(got[:-1], want):
        assert a.shape == tuple(b.shape)
        np.testing.assert_array_equal(a, b.numpy())


def test_world_size_one_needs_no_process_group():
    o, d = torch.randn(300, 3), torch.randn(300, 3)
    frame = parallel.render_sharded(_fake_render, o, d, tile=64)
    assert torch.equal(frame, _fake_render(o, d))


@pytest.mark.parametrize("n", [2, 8])
def test_bench_launches_itself_at_n_gt_1(n):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment re-executes under torch.distributed.run, all ranks reach
    init_process_group and assemble a frame through parallel.gather_frame (gloo + host tensors: what runs without GPUs); N = 8 is
    the driver's own launch at the node's size."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--backend", "gloo", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["launch_check"] and line["world"] == n and line["backend"] == "gloo" and line["frame_assembled"]
    assert len(line["tiles_per_rank"]) == n and sum(line["tiles_per_rank"]) * line["tile"] >= 1000 and max(line["tiles_per_rank"]) - min(line["tiles_per_rank"]) <= 1


def test_bench_refuses_more_gpus_than_visible_with_a_json_line():
    """asked for more devices than the box has: ONE JSON error line and exit code 2, not an AssertionError"""
    import json
    import os
    import subprocess
    import sys
    import torch
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    n = (torch.cuda.device_count() if torch.cuda.is_available() else 0) + 7
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 2
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["value"] is None and line["n_gpus"] == n and "visible" in line["error"]
