"""CPU, gloo, world 2 and 8: the collectives of neuman_hip/dp.py (data-parallel training of the background NeRF; reference train.py:26-28) on a toy
model -- the flat gradient buffer's ONE all_reduce gives the full-batch gradient when every rank's loss is its share of the global mean, the
small all_gather carries counts and maxima, broadcast_parameters makes the ranks equal, shard_batch takes rank-strided rays."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
from neuman_hip import dp  # noqa: E402


def _model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(5, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))


def _batch():
    g = torch.Generator().manual_seed(4)
    return {'origin': torch.randn(37, 5, generator=g), 'color': torch.rand(37, 3, generator=g), 'note': 'kept', 'scalar': torch.tensor(2.0)}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _model(10 + rank)                                        # ranks start different on purpose
    dp.broadcast_parameters([net])
    sync = dp.GradSync(net.parameters(), n_extra=2)
    batch = _batch()
    mine = dp.shard_batch(batch, rank, world)
    assert mine['note'] == 'kept' and mine['scalar'].shape == () and mine['origin'].shape[0] == len(range(rank, 37, world))
    stats = dp.all_gather_floats([float(mine['color'].numel()), mine['color'].max()])
    n_global = float(stats[:, 0].sum())
    assert n_global == 37 * 3 and float(stats[:, 1].max()) == float(batch['color'].max())
    outs = []
    for it in range(2):                                            # second round: zero_grad's set_to_none detaches the views, reduce() must cope
        if it == 0:
            sync.zero()
        else:
            for p in net.parameters():
                p.grad = None
        loss = ((net(mine['origin']) - mine['color']) ** 2).sum() / n_global
        loss.backward()
        vals = sync.reduce(extra=[loss.detach(), 1.0])
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync.views))
        outs.append(([p.grad.clone().numpy() for p in net.parameters()], vals))
    q.put((rank, outs, [p.detach().numpy() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_flat_gradient_all_reduce_gives_the_full_batch_gradient(world):
    """world 8: the node's real size -- eight normaliser contributions in the all_gather, ranks with 5 and with 4 of the 37 rays"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    net = _model(10)                                               # rank 0's weights are everybody's
    for other in got[1:]:
        for a, b in zip(got[0][2], other[2]):
            np.testing.assert_array_equal(a, b)
    for a, b in zip(got[0][2], net.parameters()):
        np.testing.assert_array_equal(a, b.detach().numpy())
    batch = _batch()
    loss = torch.nn.functional.mse_loss(net(batch['origin']), batch['color'])
    loss.backward()
    for rank in range(world):
        for grads, vals in got[rank][1]:
            assert vals[1] == world and abs(vals[0] - float(loss)) < 1e-6
            for g, p in zip(grads, net.parameters()):
                np.testing.assert_allclose(g, p.grad.numpy(), rtol=1e-5, atol=1e-7)


def test_single_process_without_a_group_is_the_identity():
    net = _model(1)
    sync = dp.GradSync(net.parameters(), n_extra=1)
    x = torch.randn(8, 5)
    net(x).sum().backward()
    before = [p.grad.clone() for p in net.parameters()]
    vals = sync.reduce(extra=[3.0])
    assert vals == [3.0] and all(torch.equal(a, p.grad) for a, p in zip(before, net.parameters()))
    assert dp.rank_world() == (0, 1) and dp.shard_batch({'origin': x}, 0, 1)['origin'] is x
    assert dp.all_gather_floats([1.0, torch.tensor(2.0)]).tolist() == [[1.0, 2.0]]


def _optim_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _model(3)
    optim = torch.optim.Adam(net.parameters(), lr=1e-3 * (rank + 1))
    for _ in range(rank + 1):                                      # ranks "resumed" from different checkpoints: other moments, other step counts
        optim.zero_grad()
        net(torch.full((4, 5), float(rank + 1))).sum().backward()
        optim.step()
    dp.broadcast_parameters([net])
    dp.broadcast_optimizer_state(optim)
    st = optim.state_dict()
    q.put((rank, [p.detach().numpy() for p in net.parameters()],
           {k: {n: (v.numpy() if torch.is_tensor(v) else v) for n, v in e.items()} for k, e in st['state'].items()}, st['param_groups'][0]['lr']))
    dist.barrier()
    dist.destroy_process_group()


def test_optimizer_state_follows_rank_0_after_a_resume():
    """broadcast_parameters after resume() made the weights equal but left Adam's moments and step counts per rank"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_optim_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for other in got[1:]:
        assert other[3] == got[0][3] == 1e-3
        for a, b in zip(got[0][1], other[1]):
            np.testing.assert_array_equal(a, b)
        assert other[2].keys() == got[0][2].keys()
        for k in got[0][2]:
            for n in got[0][2][k]:
                np.testing.assert_array_equal(np.asarray(got[0][2][k][n]), np.asarray(other[2][k][n]))
            assert float(np.asarray(other[2][k]['step'])) == 1.0
