"""Throughput of the other BASELINE configs on synthetic scenes: C3 canonical / posed human, C4-like hybrid (render_test_views.py:74),
C5-like three-actor composite (render_gathering.py:191).  Rays resident in HBM, one warm-up frame, median of 3 (max over ranks).
One JSON line per config.

    python tools/bench_configs.py [--small]                     one MI355X
    python tools/bench_configs.py --gpus N                      N ranks, one per GPU, RCCL (self-launches under torch.distributed.run):
                                                                the frames go through the drivers' own sharding (render_utils._frame ->
                                                                parallel.render_frame_sharded: interleaved ray tiles, ONE gather per frame)
    python tools/bench_configs.py --gpus 2 --share-gpu          the same with every rank on device 0 and gloo assembly: what a one-GPU
                                                                box can execute of the N > 1 path (timings then only show it runs)
    python tools/bench_configs.py --imbalance [--small]         no timing: per-rank hit-ray counts of the interleaved tiles at world
                                                                2 / 4 / 8 for the C4 / C5 cameras (one device computes all ranks' lists)

Every line of a sharded run carries the per-rank evidence: tiles, rays, hit rays (rays x actors), local render ms, gather ms.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def self_launch(args):
    n = args.gpus
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not args.share_gpu and have < n:
        print(json.dumps({"error": f"--gpus {n} but {have} HIP device(s) visible (use --share-gpu to run the ranks on one)", "n_gpus": n,
                          "n_gpus_visible": have}), flush=True)
        raise SystemExit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(cmd[0], cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--imbalance", action="store_true")
    ap.add_argument("--only", default="", help="comma list of C3,C3p,C4,C5")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from neuman_hip import parallel, ray_utils, render_utils, synthetic
    parallel.set_frame_sharding(True, stats=world > 1)              # the drivers shard only when asked to (a collective per frame)
    small = args.small
    only = set(x for x in args.only.split(",") if x)
    coarse, fine, human = (synthetic.make_joiner(0).to(dev), synthetic.make_joiner(1).to(dev), synthetic.make_joiner(2, 'rotate').to(dev))
    verts_c, faces = synthetic.capsule_mesh() if not small else synthetic.capsule_mesh(20, 24)
    posed, T = synthetic.twist_transforms(verts_c)

    def rays(cap):
        coords = np.argwhere(np.ones(cap.shape))[:, ::-1]
        o, d = ray_utils.shot_rays(cap, coords)
        return torch.from_numpy(o).to(dev, torch.float32).contiguous(), torch.from_numpy(d).to(dev, torch.float32).contiguous()

    def hits_of(o, d, clouds, idx=None):
        """hit rays x actors among rays `idx` (all rays: None)"""
        n = 0
        for v in clouds:
            near, far = ray_utils.geometry_guided_near_far(o if idx is None else o[idx].contiguous(), d if idx is None else d[idx].contiguous(), v, 0.2)
            n += int((near < far).sum().item())
        return n

    def frame(fn, o, d):
        """the drivers' path: sharded under a process group (frame on rank 0), plain otherwise"""
        return render_utils._frame(fn, o, d)

    def report(name, cap, fn, o, d, clouds, extra=None):
        ts, rs, gs = [], [], []
        for it in range(4):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            frame(fn, o, d)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            if it:                                                # (the first frame is the warm-up)
                ts.append(time.perf_counter() - t0)
                fs = parallel.frame_stats()
                rs.append(fs.get("render_ms"))
                gs.append(fs.get("gather_ms"))
        dt = sorted(ts)[len(ts) // 2]
        total = cap.shape[0] * cap.shape[1]
        line = {"config": name, "rays": total, "n_gpus": world, "ms_per_frame": dt * 1e3, "rays_per_s": total / dt}
        line.update(extra or {})
        if world > 1:
            st = parallel.frame_stats()
            idx = parallel.tile_ray_indices(total, st["tile"], rank, world, device=dev)
            mine = {"rank": rank, "tiles": st["tiles"], "rays": st["rays"], "hit_rays": hits_of(o, d, clouds, idx),
                    "render_ms": sorted(rs)[len(rs) // 2], "gather_ms": sorted(gs)[len(gs) // 2]}
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
            hr = [p["hit_rays"] for p in per_rank]
            line.update({"backend": dist.get_backend(), "share_gpu": bool(args.share_gpu), "tile_rays": st["tile"], "per_rank": per_rank,
                         "hit_ray_imbalance": (max(hr) / (sum(hr) / world) - 1.0) if sum(hr) else 0.0,
                         "render_ms_imbalance": max(p["render_ms"] for p in per_rank) / (sum(p["render_ms"] for p in per_rank) / world) - 1.0})
        if rank == 0:
            print(json.dumps(line), flush=True)

    def imbalance(name, cap, o, d, clouds):
        total = cap.shape[0] * cap.shape[1]
        masks = []
        for v in clouds:
            near, far = ray_utils.geometry_guided_near_far(o, d, v, 0.2)
            masks.append((near < far))
        hit = torch.stack(masks, 0).sum(0)                        # actors hit per ray
        out = {"config": name, "rays": total, "hit_rays_x_actors": int(hit.sum().item()), "hit_fraction": float((hit > 0).float().mean().item())}
        for w in (2, 4, 8):
            for label, mt in (("frame_tile", parallel.FRAME_TILE), ("tile_8192", 8192)):
                tile = parallel.balanced_tile(total, w, mt)
                per = [int(hit[parallel.tile_ray_indices(total, tile, r, w, device=dev)].sum().item()) for r in range(w)]
                rays_per = [int(parallel.tile_ray_indices(total, tile, r, w, device=dev).numel()) for r in range(w)]
                out[f"world{w}_{label}"] = {"tile": tile, "hit_rays_per_rank": per, "rays_per_rank": rays_per,
                                            "hit_ray_imbalance": max(per) / (sum(per) / w) - 1.0 if sum(per) else 0.0}
        print(json.dumps(out), flush=True)

    with torch.no_grad():
        v_dev = torch.from_numpy(posed).to(dev)
        mesh = ray_utils.mesh_to_device(posed, faces, T, dev)
        if rank == 0 and not args.imbalance:
            print(json.dumps({'mesh_tree': mesh.info()}), flush=True)
        res = 128 if small else 512
        cap = synthetic.SimpleCapture(res, res, fx=1.6 * res, c2w=synthetic.spherical_c2w(40., 0., 3.0))
        o, d = rays(cap)
        cloud = torch.from_numpy(synthetic.human_vertex_cloud(0)).to(dev)
        if args.imbalance:
            imbalance("C3 posed human 512x512", cap, o, d, [v_dev])
        else:
            # ---- C3: canonical 360 view of the human net, 512x512, 128 samples, hit rays only
            if not only or "C3" in only:
                near, far = ray_utils.geometry_guided_near_far(o, d, cloud, 0.2)
                report("C3 canonical human 512x512x128", cap, lambda oo, dd: render_utils.render_smpl_nerf_rays(human, oo, dd, cloud, None, 128, True, True, 0.2, 1.0),
                       o, d, [cloud], {"hit_fraction": float((near < far).float().mean())})
            # ---- C3 posed: same camera on the twisted capsule mesh, with the obs->canonical warp
            if not only or "C3p" in only:
                near, far = ray_utils.geometry_guided_near_far(o, d, v_dev, 0.2)
                report("C3 posed human (warp) 512x512x128", cap,
                       lambda oo, dd: render_utils.render_smpl_nerf_rays(human, oo, dd, v_dev, mesh, 128, True, False, 0.2, 1.0), o, d, [v_dev],
                       {"hit_fraction": float((near < far).float().mean())})
        # ---- C4-like: 1280x720 hybrid, bkg 128+128, human 128
        w, h = (320, 180) if small else (1280, 720)
        cap = synthetic.SimpleCapture(w, h, fx=1.2 * w, c2w=synthetic.spherical_c2w(20., -5., 3.0), near=0.0, far=3.14)
        o, d = rays(cap)
        if args.imbalance:
            imbalance("C4-like hybrid 1280x720", cap, o, d, [v_dev])
        elif not only or "C4" in only:
            report("C4-like hybrid 1280x720, bkg 128+128, human 128", cap,
                   lambda oo, dd: render_utils.render_hybrid_rays(coarse, fine, human, oo, dd, 0.0, 3.14, v_dev, mesh, 128, 128, True, 0.2)[:2], o, d, [v_dev])
        # ---- C5-like: 1920x1080, 192 + 128, three actors x 192
        w, h = (480, 270) if small else (1920, 1080)
        cap = synthetic.SimpleCapture(w, h, fx=1.2 * w, c2w=synthetic.spherical_c2w(20., -5., 3.5), near=0.0, far=3.14)
        o, d = rays(cap)
        vs, ms = [], []
        for k, dx in enumerate((-0.7, 0.0, 0.7)):
            p2 = (posed + np.array([dx, 0, 0.1 * k], np.float32)).astype(np.float32)
            T2 = T.copy()
            T2[:, :3, 3] += np.array([dx, 0, 0.1 * k])
            vs.append(torch.from_numpy(p2).to(dev))
            ms.append(ray_utils.mesh_to_device(p2, faces, T2, dev))
        if args.imbalance:
            imbalance("C5-like 3 actors 1920x1080", cap, o, d, vs)
        elif not only or "C5" in only:
            report("C5-like 3 actors 1920x1080, bkg 192+128, 3 x 192", cap,
                   lambda oo, dd: render_utils.render_multi_rays(coarse, fine, [human] * 3, oo, dd, 0.0, 3.14, vs, ms, 192, 128, True, 0.2), o, d, vs)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
