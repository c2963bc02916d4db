"""Training iterations/s of the background NeRF under data parallelism (neuman_hip/dp.py; reference train.py:26-28: nn.DataParallel around
both nets), one process per GPU over RCCL:

    python tools/train_dp_bench.py --gpus N [--rays 4096] [--steps 20] [--warmup 3] [--weak] [--share-gpu]

`--gpus N` (N > 1) without WORLD_SIZE in the environment launches itself under torch.distributed.run like bench.py.  The batch of `--rays`
rays is the GLOBAL batch (the reference's DataParallel splits one batch over the GPUs: "strong"); `--weak` gives every rank `--rays` rays.
`--share-gpu`: N ranks on device 0 with gloo (what a one-GPU box can execute of the N > 1 path).  `--dist`: the RCCL collectives on a group
of ONE rank.  Rank 0 prints ONE JSON line (bench.py's keys).  The timed region is K iterations -- batch draw, both networks forward and
backward, the gradient all_reduce, Adam -- between barriers, the maximum over the ranks."""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def self_launch(args):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    if not args.share_gpu:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            print(json.dumps({"error": f"--gpus {args.gpus} but {have} HIP device(s) visible (--share-gpu runs the ranks on one)", "value": None}), flush=True)
            raise SystemExit(2)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(cmd[0], cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--weak", action="store_true")
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--dist", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    grouped = world > 1 or args.dist
    if grouped:
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from neuman_hip import bkg_trainer, ray_utils, synthetic, train
    S, NI = 128, 128
    cap = synthetic.SimpleCapture(800, 800)
    o_all, d_all = ray_utils.shot_all_rays_dev(cap, dev)
    n_global = args.rays * (world if args.weak else 1)
    g = torch.Generator(device='cpu').manual_seed(0)                 # the same draws on every rank
    target = synthetic.make_joiner(7).to(dev).eval()                 # colours a field can fit: another net's rendering of the same rays

    def batch():
        idx = torch.randint(0, o_all.shape[0], (n_global,), generator=g)[rank::world].to(dev)
        o, d = o_all[idx].contiguous(), d_all[idx].contiguous()
        n = o.shape[0]
        with torch.no_grad():
            from neuman_hip import render_utils
            color = render_utils.render_vanilla_rays(target, None, o, d, cap.near['bkg'], cap.far['bkg'], 32, 0, True)[0]
        return {'origin': o, 'direction': d, 'near': torch.full((n, 1), float(cap.near['bkg']), device=dev),
                'far': torch.full((n, 1), float(cap.far['bkg']), device=dev), 'color': color}
    opt = types.SimpleNamespace(samples_per_ray=S, importance_samples_per_ray=NI, perturb=1.0, raw_noise_std=0.0, white_bkg=True, margin=0.9,
                                penalize_empty_space=0.0, empty_space_loss_fn='mse', delay_iters=0, lrate_decay=250, learning_rate=5e-4,
                                ablate_nerft=False, rays_per_batch=n_global, max_iter=0, valid_iter=0, out=None, resume=False, load_weights=False)
    coarse, fine = synthetic.make_joiner(0).to(dev).train(), synthetic.make_joiner(1).to(dev).train()
    optim = torch.optim.Adam([{"params": coarse.parameters(), "lr": opt.learning_rate}, {"params": fine.parameters(), "lr": opt.learning_rate}])
    tr = bkg_trainer.BackgroundNeRFTrainer(opt, coarse, optim, fine_net=fine, batches=batch, data_parallel=grouped)

    def sync():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()
    losses = []
    for _ in range(args.warmup):
        losses.append(tr.train_batch(batch())['total_loss'])
        tr.iteration += 1
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(tr.train_batch(batch())['total_loss'])
        tr.iteration += 1
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if grouped:
        tt = t.cpu() if args.share_gpu else t
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    if rank == 0:
        evals = n_global * (S + S + NI)
        print(json.dumps({
            "metric": "training iterations/s of the background NeRF (trainers/vanilla_nerf_trainer.py:45-96 + backward + Adam), data parallel over rays",
            "value": args.steps / dt, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": f"training products {train.GEMM_PRECISION}, fp16 storage of activations / dZ from {train.STORE16_MIN_ROWS} samples per call: {train.STORE16}",
            "data": "synthetic",
            "config": {"workload": f"{n_global} rays per iteration over {world} rank(s), 128 coarse + 256 fine evaluations per ray, both 8x256 nets, photometric loss, Adam",
                       "rays_per_rank": len(range(rank, n_global, world)), "parallelism": f"dp{world}: one all_reduce of 2 x 595,844 float32 gradients per iteration"
                                                                                        + (", ranks share one GPU, gloo" if args.share_gpu else ", RCCL")},
            "rays_per_s": n_global * args.steps / dt, "mlp_tflops_fwd_bwd": evals * 1186816 * 3 * args.steps / dt / 1e12,
            "loss_first": losses[0], "loss_last": losses[-1],
            "hardware_note": "N > 1 across devices is unmeasured on hardware: the build box has one GPU" if world > 1 and args.share_gpu else None}), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
