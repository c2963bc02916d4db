"""Data-parallel training check (tests/test_hip_dp_train.py; run as one process per rank with RANK / WORLD_SIZE / MASTER_* in the environment):

    dist_train_check.py gloo   WORLD_SIZE ranks sharing GPU 0, gradients reduced over gloo (what a one-GPU box can run of N > 1)
    dist_train_check.py nccl   RANK=0 WORLD_SIZE=1: the RCCL all_reduce / all_gather / broadcast on a group of one rank

Every rank builds the same synthetic scene and the same nets, and runs (a) the single-process trainer on the FULL batches -- the
reference semantics of nn.DataParallel, train.py:26-28: one step on the whole batch -- and (b) the data-parallel trainer on its
rank-strided share of the same batches.  Rank 0 prints one JSON line: the deviation of the reduced gradient of the first iteration from the
full-batch gradient (relative to each tensor's largest entry), the two loss curves over 20 steps, the final weights' deviation."""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "ml-neuman_amd"))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))


def main():
    backend = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from neuman_hip import bkg_trainer, dp, ray_batches, synthetic, train
    import test_hip_bkg_trainer as T
    G = types.SimpleNamespace(io=__import__('neuman_hip.data_io', fromlist=['x']), rb=ray_batches)
    store = T._scene_store(G)
    opt = T.trainer_opt(penalize_empty_space=0.1, rays_per_batch=512, lrate_decay=250, perturb=0.0, out=None)

    def nets():
        c, f = synthetic.make_joiner(0).to(dev).train(), synthetic.make_joiner(1).to(dev).train()
        return c, f, torch.optim.Adam([{"params": c.parameters(), "lr": opt.learning_rate}, {"params": f.parameters(), "lr": opt.learning_rate}])

    # (a) the whole batch on one process (every rank computes it: the comparison needs no communication)
    c1, f1, o1 = nets()
    full = ray_batches.BackgroundRayBatcher(opt, store, draws='device', seed=3)
    tr1 = bkg_trainer.BackgroundNeRFTrainer(opt, c1, o1, fine_net=f1, batches=full, data_parallel=False)
    o1.zero_grad()
    b0 = full()
    terms = tr1.loss_func(b0)
    sum(terms).backward()
    g_full = [p.grad.clone() for p in list(c1.parameters()) + list(f1.parameters())]
    full_terms = [float(t.detach()) for t in terms]
    o1.zero_grad()
    full.gen.manual_seed(3)                                          # rewind: the loop below starts from the same first batch
    curve1 = []
    tr1.iteration = 0
    for _ in range(steps):
        curve1.append(tr1.train_batch(full())['total_loss'])
        tr1.iteration += 1

    # (b) data parallel
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    c2, f2, o2 = nets()
    if rank != 0:                                                    # the constructor's broadcast must repair this
        with torch.no_grad():
            for p in c2.parameters():
                p.add_(0.01)
    mine = ray_batches.BackgroundRayBatcher(opt, store, draws='device', seed=3, rank=rank, world=world)
    tr2 = bkg_trainer.BackgroundNeRFTrainer(opt, c2, o2, fine_net=f2, batches=mine, data_parallel=True)
    assert all(torch.equal(a, b.to(dev)) for a, b in zip(c2.parameters(), synthetic.make_joiner(0).parameters()))     # rank 0's weights everywhere
    tr2.sync.zero()
    lb = mine()
    assert lb['origin'].shape[0] == len(range(rank, 512, world))
    assert torch.equal(lb['origin'], b0['origin'][rank::world]) and torch.equal(lb['color'], b0['color'][rank::world])
    terms2, dead = tr2.loss_func_dp(lb)
    sum(terms2).backward()
    vals = tr2.sync.reduce(extra=[t.detach() for t in terms2])
    g_dp = [p.grad.clone() for p in list(c2.parameters()) + list(f2.parameters())]
    worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(g_dp, g_full))
    mine.gen.manual_seed(3)
    curve2 = []
    for _ in range(steps):
        curve2.append(tr2.train_batch(mine())['total_loss'])
        tr2.iteration += 1
    wdev = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(list(c2.parameters()) + list(f2.parameters()),
                                                                                              list(c1.parameters()) + list(f1.parameters())))
    # ranks hold identical weights after the loop
    flat = torch.cat([p.detach().reshape(-1) for p in list(c2.parameters()) + list(f2.parameters())])
    both = dp.all_gather_floats([flat.double().sum(), flat.double().abs().sum()])
    same = bool((both == both[0:1]).all())
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        tr2.save_model(os.path.join(td, f'ck{rank}.pth.tar'))
        wrote = os.path.exists(os.path.join(td, f'ck{rank}.pth.tar'))
        keys = sorted(torch.load(os.path.join(td, f'ck{rank}.pth.tar'), map_location='cpu', weights_only=False)['coarse_model_state_dict'])[:1] if wrote else []
    if rank == 0:
        print(json.dumps({"backend": backend, "world": world, "dead": bool(dead), "first_iteration_gradient_dev": worst,
                          "first_iteration_terms_full": full_terms, "first_iteration_terms_dp": vals,
                          "curve_full": curve1, "curve_dp": curve2, "final_weight_dev": wdev, "ranks_hold_equal_weights": same,
                          "checkpoint_written_by_rank0": wrote, "checkpoint_first_key": keys, "gemm": train.GEMM_PRECISION}), flush=True)
    else:
        assert not wrote
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
