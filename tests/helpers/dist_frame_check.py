"""Executed by tests/test_parallel_gpu.py (one process per rank): the multi-GPU frame path with REAL device-rendered tiles.

    dist_frame_check.py nccl   (RANK=0 WORLD_SIZE=1)  the RCCL gather collective on an initialised group of one rank, device tensors
    dist_frame_check.py gloo   (RANK=r WORLD_SIZE=2)  two processes share the box's one GPU, render their own tiles on it and
                                                      assemble the frame through gloo (host tensors)

Either way rank 0 compares the assembled frame bit for bit with the frame rendered unsharded in one piece and prints
one JSON object.
"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from neuman_hip import parallel, ray_utils, render_utils, synthetic  # noqa: E402


def main():
    backend = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    coarse, fine = synthetic.make_joiner(0).to(dev), synthetic.make_joiner(1).to(dev)
    cap = synthetic.SimpleCapture(200, 120)
    o, d = ray_utils.shot_all_rays_dev(cap, dev)
    total, tile = o.shape[0], 1024                              # 24 tiles (the last one ragged: 24000 = 23 * 1024 + 448)

    def render(oo, dd):
        rgb, depth = render_utils.render_vanilla_rays(coarse, fine, oo, dd, 0.0, 3.14, 32, 32, True)
        return torch.cat([rgb, depth[:, None]], 1)

    with torch.no_grad():
        whole = render(o, d)                                     # unsharded, before any process group exists
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            frame = parallel.render_sharded(render, o, d, tile=tile, force_collective=True)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            idx = parallel.tile_ray_indices(total, tile, rank, world, device=dev)
            local = render(o[idx].contiguous(), d[idx].contiguous()).cpu()
            frame = parallel.gather_frame(local, idx.cpu(), total, tile)
        dist.barrier()
        if rank == 0:
            frame = frame.to(dev)
            print(json.dumps({"backend": backend, "world": world, "rays": total, "tiles": (total + tile - 1) // tile,
                              "bit_identical": bool(torch.equal(frame, whole)), "finite": bool(torch.isfinite(frame).all())}), flush=True)
        else:
            assert frame is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
