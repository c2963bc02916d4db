"""Executed by tests/test_parallel_gpu.py (one process per rank): the reference-named FRAME DRIVERS under a process group.

    dist_drivers_check.py nccl   (RANK=0 WORLD_SIZE=1, NEUMAN_FORCE_COLLECTIVE=1)  RCCL gather on a group of one rank
    dist_drivers_check.py gloo   (RANK=r WORLD_SIZE=2)  two processes share the box's one GPU; each renders the rays of its
                                                        interleaved tiles through the same driver call, gloo assembles the frame

render_vanilla, render_smpl_nerf, render_hybrid_nerf (BASELINE config 4: render_test_views.py:74) and
render_hybrid_nerf_multi_persons (config 5: render_gathering.py:191) are each called twice with the same arguments: before any
process group exists (the unsharded frame) and under the group.  Rank 0 compares the two bit for bit and prints one JSON object;
the other ranks must get None.
"""
import json
import os
import sys
import types

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from neuman_hip import parallel, render_utils, synthetic  # noqa: E402


def main():
    backend = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    coarse, fine = synthetic.make_joiner(0).to(dev), synthetic.make_joiner(1).to(dev)
    humans = [synthetic.make_joiner(2 + k, 'rotate').to(dev) for k in range(3)]
    net = types.SimpleNamespace(coarse_bkg_net=coarse, fine_bkg_net=fine, coarse_human_net=humans[0], parameters=coarse.parameters)
    models = [types.SimpleNamespace(coarse_human_net=h, parameters=h.parameters) for h in humans]
    verts_c, faces = synthetic.capsule_mesh(20, 24)
    posed, T = synthetic.twist_transforms(verts_c)
    W, H = 96, 64
    cap = synthetic.SimpleCapture(W, H, fx=1.2 * W, c2w=synthetic.spherical_c2w(20., -5., 3.0), near=0.0, far=3.14)
    shifts = [np.array([dx, 0, 0.1 * k], np.float32) for k, dx in enumerate((-0.7, 0.0, 0.7))]
    posed_l = [(posed + s).astype(np.float32) for s in shifts]
    T_l = []
    for s in shifts:
        t = T.copy()
        t[:, :3, 3] += s
        T_l.append(t)

    calls = {
        "render_vanilla": lambda: render_utils.render_vanilla(coarse, cap, fine, samples_per_ray=32, importance_samples_per_ray=32, return_depth=True),
        "render_smpl_nerf": lambda: render_utils.render_smpl_nerf(net, cap, posed, faces, T, samples_per_ray=32, return_depth=True, return_mask=True),
        "render_hybrid_nerf": lambda: render_utils.render_hybrid_nerf(net, cap, posed, faces, T, samples_per_ray=32, importance_samples_per_ray=32,
                                                                      return_depth=True),
        "render_hybrid_nerf_multi_persons": lambda: render_utils.render_hybrid_nerf_multi_persons(
            net, cap, models, posed_l, [faces] * 3, T_l, samples_per_ray=24, importance_samples_per_ray=16, return_depth=True),
    }
    whole = {k: fn() for k, fn in calls.items()}                  # unsharded, before any process group exists
    assert not parallel.sharding_active()
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    assert not parallel.sharding_active()                         # a process group alone does not turn the renderers into collectives
    parallel.set_frame_sharding(True)
    assert parallel.sharding_active(), "set NEUMAN_FORCE_COLLECTIVE=1 for a group of one rank"
    report = {"backend": backend, "world": world, "rays": W * H}
    for k, fn in calls.items():
        out = fn()
        stats = dict(parallel.LAST_FRAME_STATS)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rays": stats["rays"], "tiles": stats["tiles"]})
        if rank == 0:
            same = all(np.array_equal(a, b) for a, b in zip(out, whole[k]))
            finite = all(np.isfinite(a).all() for a in out)
            report[k] = {"bit_identical": bool(same), "finite": bool(finite), "outputs": len(out), "tile": stats["tile"],
                         "rays_per_rank": [p["rays"] for p in per_rank], "hit_fraction": float((whole["render_smpl_nerf"][2] > 0).mean())}
        else:
            assert out is None
    dist.barrier()
    if rank == 0:
        print(json.dumps(report), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
