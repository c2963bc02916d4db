"""What ONE rank of an N-GPU render of the C2 frame does, measured on one GPU (a number the first real 8-GPU run can be
compared with; nothing here is a multi-GPU measurement).  For N in 1, 2, 4, 8: rank 0's interleaved tiles of the 800 x 800 frame
(parallel.balanced_tile(total, N, FRAME_TILE): 640000 / N rays), rendered as render_frame_sharded renders them -- ONE ray list through
render_vanilla_rays -- timed with HIP events over `--frames` frames; plus the frame assembly at world 1 on an RCCL group of one rank
(gather + index_select of the [rays, 4] buffer).  Prints one JSON line per N and a prediction line: frame time at N = per-rank render time
+ assembly, speed-up over N = 1."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-neuman_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from neuman_hip import parallel, ray_utils, render_utils, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
coarse, fine = synthetic.make_joiner(0).to(dev), synthetic.make_joiner(1).to(dev)
cap = synthetic.SimpleCapture(800, 800)
o, d = ray_utils.shot_all_rays_dev(cap, dev)
total = o.shape[0]
S, NI = 128, 128
res = {}
with torch.no_grad():
    for world in (1, 2, 4, 8):
        tile = parallel.balanced_tile(total, world, parallel.FRAME_TILE)
        idx = parallel.tile_ray_indices(total, tile, 0, world, device=dev)
        oo, dd = o[idx].contiguous(), d[idx].contiguous()
        render_utils.render_vanilla_rays(coarse, fine, oo, dd, cap.near['bkg'], cap.far['bkg'], S, NI, True)
        torch.cuda.synchronize()
        ms = []
        for _ in range(args.frames):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            oo, dd = o[idx].contiguous(), d[idx].contiguous()             # (the per-frame gather of the rank's rays is part of the sharded path)
            rgb, depth = render_utils.render_vanilla_rays(coarse, fine, oo, dd, cap.near['bkg'], cap.far['bkg'], S, NI, True)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        res[world] = {"world": world, "tile": tile, "rays_of_rank0": int(idx.shape[0]), "render_ms": sorted(ms)[len(ms) // 2], "all_ms": ms}
        print(json.dumps(res[world]), flush=True)
    # frame assembly on a group of one rank: the RCCL gather + index_select of a [rays, 4] buffer of the N = 8 shard size and of the whole frame
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    asm = {}
    for n in (total // 8, total):
        buf = torch.rand((n, 4), device=dev)
        ii = torch.arange(n, device=dev)
        parallel.gather_frame(buf, ii, n, parallel.FRAME_TILE, force_collective=True)
        torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            parallel.gather_frame(buf, ii, n, parallel.FRAME_TILE, force_collective=True)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        asm[n] = sorted(ms)[len(ms) // 2]
    dist.destroy_process_group()
base = res[1]["render_ms"]
pred = {str(w): {"frame_ms": res[w]["render_ms"] + asm[total], "speedup": base / (res[w]["render_ms"] + asm[total])} for w in (2, 4, 8)}
print(json.dumps({"assembly_ms_group_of_one": {"rays_80000": asm[total // 8], "rays_640000": asm[total]},
                  "predicted": pred, "note": "prediction = rank 0's measured render time of its N-th of the frame + the whole-frame assembly measured on a group of one "
                  "rank (an upper bound on what rank 0 does after the payloads arrived; wire time of 7 x 1.28 MB over xGMI ~ 10-70 us is not in it); measured on ONE "
                  "GPU -- ranks on one node share power and fabric limits this cannot show"}), flush=True)
