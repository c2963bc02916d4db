#!/usr/bin/env python
"""bench.py -- rays/s of the NeuMan ray-march hot path on MI355X (BASELINE.json metric, config 2).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches itself: it checks that N devices are
visible (one JSON error line and exit code 2 otherwise) and re-executes under torch.distributed.run with one rank per GPU.

One "step" = one 800x800 frame of the background NeRF: 128 coarse + 128 importance samples per ray (the reference
evaluates the 8x256 MLP 128 + 256 = 384 times per ray, render_utils.py:108-161), synthetic-dense weights
(SURVEY 8d), rays already resident in HBM when the timed region starts.  With N > 1 the frame's ray tiles are
sharded across the ranks (no data-path collective) and assembled on rank 0 by one RCCL gather per frame, which is
inside the timed region; total work is fixed, so `scaling` is "strong".

Precision (`--precision`, default "mixed" = the package default): the coarse pass, whose compositing weights place the
importance samples, runs split-fp16 x3 (float32-class sigma: the inverse CDF amplifies a coarse-pass error by 1 / pdf);
the fine pass, whose output is composited into the frame, runs the 16-bit fixed-point limbs on the i8 MFMA.  Sample
positions are bit-identical to the all-fp16x3 path and every pixel stays within 1e-4 of the oracle on identical samples
(tests/test_hip_render.py); the bench line also reports the all-fp16x3, all-bf16x3 and all-i8x3 frame rates measured in
the same run (`other_precisions`).

Parity (`parity_vs_oracle`): the CPU oracle renders the first 4096 rays of this very frame for the CPU baseline; its pixels,
coarse weights and fine sample positions are kept and the device renders the same rays in the timed configuration.  Reported: RGB
L-inf, PSNR, rays off by more than 1e-4, and the attribution of that deviation (oracle/attribution.py): (a) the device's shading
pass on the ORACLE's sample positions and (a') the oracle's on the DEVICE's, both <= 1e-4 on every ray; (b) the coarse weights;
(c) every ray beyond 1e-4 among the 6 % most displaced and a first-order bound with a measured Lipschitz constant on every ray;
(d) the count of rays beyond 1e-4 of the ARBITER -- the reference's own render_vanilla run in float64 on these rays (tests/golden/arbiter.npz) --
against the count the reference's own float32 run leaves (+ a quarter): the inverse-CDF step of ray_utils.py:164-194 is ill conditioned.
`parity_vs_reference` holds the same comparison for whole frames of two workloads whose coarse and fine network agree; on the
well-conditioned one (`unconditional`) EVERY ray of the 800x800 frame is within 1e-4 of the reference's float64 frame.

The coarse pass evaluates the density head only (the reference computes the coarse colours, composites them and
discards the result, render_utils.py:139-141): sigma is bit-identical, 17 % of that pass's MACs are not issued.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel -- the fine launch (163.84 M of the frame's
245.76 M evaluations): algorithmic FLOPs (1,186,816 per MLP evaluation) / its launch time measured with HIP
events on the launch stream, against the 2.5 PFLOP/s dense bf16 MFMA peak; `roofline_coarse` is the same for the
coarse launch.  `cpu_baseline` times the CPU restatement of the reference (the numpy port under oracle/, kind "port") on
a bounded prefix of the same frame, on this box's host cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOP_PER_EVAL = 1186816            # 593,408 MAC per sample evaluation
PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA, MI355X_MICROARCH.md
W, H, S, NI = 800, 800, 128, 128
EVALS_PER_RAY = S + (S + NI)
DTYPES = {
    "mixed": "coarse (sampling) pass: split-fp16 hi+lo MFMA x3, f32 accumulate; fine (shading) pass: per-row-scaled int16 as two "
             "int8 limbs on the i8 MFMA x3, exact int32 accumulate, encodings on split bf16",
    "fp16x3": "fp16x3 (split-fp16 hi+lo MFMA x3, f32 accumulate, power-of-two operand scalings)",
    "bf16x3": "bf16x3 (split-bf16 hi+lo MFMA x3, f32 accumulate)",
    "i8x3": "i8x3 (per-row-scaled int16 as 2 int8 limbs, i8 MFMA x3, exact int32 accumulate; encodings on bf16x3)",
    "bf16": "bf16 (f32 accumulate)", "fp32": "f32"}
PMC_SUMMARIES = ("profiles/r06_bench_pmc_summary.json", "profiles/r05_bench_pmc_summary.json", "profiles/r04_bench_pmc_summary.json", "profiles/r03_bench_pmc_summary.json", "profiles/r02_bench_pmc_summary.json", "profiles/r01_bench_pmc_summary.json")
KERNEL_OF = {"fp16x3": "nerf_mlp_kernel<4, false>", "bf16x3": "nerf_mlp_kernel<1, false>", "i8x3": "nerf_mlp_i8s_kernel", "bf16": "nerf_mlp_kernel<2, false>",
             "fp32": "nerf_mlp_ref_kernel"}


def cpu_baseline(max_rays=4096):
    """CPU restatement of the reference renderer (oracle/: numpy + torch's CPU BLAS for the dense layers, the call the
    reference itself makes) on the first `max_rays` rays of the same frame with the same weights and sampling.  Reported,
    never the thing shipped.  Returns (baseline dict, oracle outputs of that slice) -- the pixels and sample positions are
    what `parity_vs_oracle` scores the device against."""
    import numpy as np
    from oracle import attribution, compositing, nerf_mlp, ray_ops
    from oracle.nerf_mlp import JoinerSpec
    from neuman_hip import synthetic
    from threadpoolctl import threadpool_limits
    cap = synthetic.SimpleCapture(W, H)
    nets = [(synthetic.state_numpy(synthetic.make_joiner(seed)), JoinerSpec()) for seed in (0, 1)]
    origins, dirs = ray_ops.shot_all_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, cap.shape)

    def run(n_rays, keep=None):
        """reference render_utils.py:108-161 on rays [0, n_rays) in batches of 2048 (oracle.render.render_vanilla's loop,
        spelled out so that the fine sample positions can be kept)"""
        t0 = time.perf_counter()
        for i in range(0, n_rays, 2048):
            j = min(i + 2048, n_rays)
            o, d = origins[i:j].astype(np.float32), dirs[i:j].astype(np.float32)
            near = np.full((j - i, 1), cap.near['bkg'], np.float32)
            far = np.full((j - i, 1), cap.far['bkg'], np.float32)
            pts, dd, z = ray_ops.ray_to_samples(o, d, near, far, S)
            out = nerf_mlp.joiner_forward(*nets[0], pts, dd)
            _, _, _, w, _ = compositing.raw2outputs(out, z, d)
            pts, dd, zf = ray_ops.ray_to_importance_samples(o, d, z, w, NI)
            out = nerf_mlp.joiner_forward(*nets[1], pts, dd)
            rgb, _, _, _, depth = compositing.raw2outputs(out, zf, d)
            if keep is not None:
                keep.append((rgb, depth, zf, w))
        return time.perf_counter() - t0

    # BLAS on every hardware thread of a big host is slower than on a subset: pick the fastest thread count on a short
    # probe, then time the bounded sample with it (cores = the threads actually used)
    ncpu = os.cpu_count() or 1
    t_before = torch.get_num_threads()
    best = (None, 0.0)
    for threads in sorted({min(ncpu, t) for t in (8, 16, 32, 64, 128, ncpu)}):
        torch.set_num_threads(threads)
        with threadpool_limits(limits=threads):
            run(256)
            rate = 1024 / run(1024)
        if rate > best[1]:
            best = (threads, rate)
    torch.set_num_threads(best[0])
    keep = []
    with threadpool_limits(limits=best[0]):
        dt = run(max_rays, keep)
    pvr = None
    try:                                                          # measured in the build container by tools/port_vs_reference.py
        for name in ("r06_port_vs_reference.json", "r02_port_vs_reference.json"):
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                with open(path) as f:
                    pvr = json.load(f)
                pvr["source"] = "profiles/" + name
                break
    except Exception:
        pass
    base = {"value": max_rays / dt, "unit": "rays/s", "cores": best[0], "kind": "port",
            "sample": f"first {max_rays} rays of the 800x800 frame, 128+128 samples/ray, rays_per_batch=2048, {dt:.1f} s "
                      f"with {best[0]} BLAS threads (best of a 8..{ncpu} probe) on a {ncpu}-thread host",
            "port_vs_reference_speed": pvr}
    o32, d32 = origins[:max_rays].astype(np.float32), dirs[:max_rays].astype(np.float32)

    def fine_pass_on(z_dev):
        """the oracle's fine network + compositing on given sample positions (rows [0, len(z_dev)) of the slice) -> rgb"""
        n = z_dev.shape[0]
        torch.set_num_threads(best[0])
        pts = (o32[:n, None, :] + d32[:n, None, :] * z_dev[..., None]).astype(np.float32)
        with threadpool_limits(limits=best[0]):
            out = nerf_mlp.joiner_forward(*nets[1], pts, np.broadcast_to(d32[:n, None, :], pts.shape))
        torch.set_num_threads(t_before)
        return compositing.raw2outputs(out, z_dev, d32[:n])[0]

    oracle = {"rgb": np.concatenate([k[0] for k in keep]), "depth": np.concatenate([k[1] for k in keep]),
              "z_fine": np.concatenate([k[2] for k in keep]), "w_coarse": np.concatenate([k[3] for k in keep]), "fine_pass_on": fine_pass_on,
              "checker": attribution}                              # (the checker's entry points travel with its outputs: nothing else of bench.py imports oracle/)
    torch.set_num_threads(t_before)
    return base, oracle


def parity_vs_oracle(oracle, coarse, fine, origins, dirs, precision, full=True):
    """The device's rendering of the oracle's rays, scored against the oracle's pixels with the deviation attributed
    (oracle/attribution.py, statements (a)-(d): both conditional parities on every ray, the coarse weights, the displacement rank and
    first-order bound, the count against 1.5 x the oracle-vs-reference floor of these very rays).  `oracle` is cpu_baseline()'s second
    result: the checker's outputs.  `full=False` (the other precisions): counts and PSNR only."""
    import numpy as np
    from neuman_hip import render_utils
    attribution = oracle["checker"]
    n = oracle["rgb"].shape[0]
    o, d = origins[:n].contiguous(), dirs[:n].contiguous()
    z_ora = torch.as_tensor(oracle["z_fine"]).to(o.device)
    rgb, z, w, rgb_on = attribution.device_two_pass(render_utils, coarse, fine, o, d, 0.0, 3.14, S, NI, z_ora, precision=precision)
    err = np.abs(rgb - oracle["rgb"]).max(-1)
    mse = float(np.mean((rgb.astype(np.float64) - oracle["rgb"]) ** 2))
    out = {"rays": int(n), "rgb_linf": float(err.max()), "psnr_db": float(10 * np.log10(1.0 / max(mse, 1e-30))), "rays_gt_1e-4": int((err > 1e-4).sum())}
    out["vs_reference_f64"] = attribution.against_arbiter(rgb, attribution.load_arbiter("bench"))[0]["vs_reference_f64"]
    if full:
        rep, fails = attribution.two_pass(rgb, z, w, rgb_on, oracle["rgb"], oracle["z_fine"], oracle["w_coarse"], oracle["fine_pass_on"],
                                          arbiter=attribution.load_arbiter("bench"), max_forward_rays=2048)
        out["attribution"] = rep
        out["attribution_statements_violated"] = fails
    return out


def parity_vs_reference(checker, dev, origins, dirs, precision):
    """The device against the ARBITER: frames THE REFERENCE ITSELF rendered on these rays in float64, beside the reference's own float32
    frames (tests/golden/arbiter*.npz, made by tests/golden/make_golden_f64.py from the unmodified reference; `checker` = oracle.attribution,
    which only loads and counts).  Two workloads whose coarse and fine network agree, rendered as the timed frame is (whole 800x800 frame,
    128 + 128, the timed precision):
      unconditional  synthetic 'fog' preset (density positive everywhere: the inverse CDF is well conditioned on every ray) -- EVERY ray of the
                     frame within 1e-4 of the reference's float64 frame (all 640 000 when tests/golden/arbiter_full.npz is present, else the
                     64 000 of rows 5::10); violated -> `every_ray_within_1e-4` false
      opaque00       surfaces (rows 5::10: 64 000 rays): the count beyond 1e-4 beside the reference's own float32 count"""
    import numpy as np
    from neuman_hip import render_utils, synthetic
    out = {"what": "device frames vs the reference's own render_vanilla run in float64 on identical rays and weights (tests/golden/make_golden_f64.py), beside "
                   "the reference's float32 run vs the same; RGB L-inf over f32 pixels before any quantisation"}
    for name, preset in (("unconditional", "fog"), ("opaque00", "opaque")):
        net = synthetic.make_joiner(0, preset=preset).to(dev)
        net.precision = precision
        with torch.no_grad():
            rgb = render_utils.render_vanilla_rays(net, net, origins, dirs, 0.0, 3.14, S, NI, True)[0].cpu().numpy()
        arb = checker.load_arbiter("wc_fog00" if preset == "fog" else "wc_opaque00")
        rows = rgb.reshape(H, W, 3)[arb["rows"]].reshape(-1, 3)
        rep = checker.against_arbiter(rows, arb)[0]
        rep["workload"] = f"synthetic.make_joiner(0, preset='{preset}') as coarse and fine net, 800x800, 128 + 128, rows 5::10 ({rows.shape[0]} rays)"
        if preset == "fog":
            rep["every_ray_within_1e-4"] = bool(rep["vs_reference_f64"]["rays_gt_1e-4"] == 0)
            if os.path.exists(checker.ARBITER_FULL):
                full = checker.load_arbiter_full()
                assert full["name"] == "fog00", full["name"]
                rep_full = checker.against_arbiter(rgb, full)[0]
                rep_full["workload"] = "the same, EVERY ray of the 800x800 frame (640000 rays)"
                rep_full["every_ray_within_1e-4"] = bool(rep_full["vs_reference_f64"]["rays_gt_1e-4"] == 0)
                rep = {"whole_frame": rep_full, "rows_5_10": rep, "every_ray_within_1e-4": rep_full["every_ray_within_1e-4"] and rep["every_ray_within_1e-4"]}
        out[name] = rep
    return out


def kernel_of(which, precision):
    """the kernel a pass's MLP launch runs: the density-only fp16x3 launch of the sampling pass has its own (csrc/mlp_f16t.hip)"""
    if which == "coarse" and precision == "fp16x3" and os.environ.get("NEUMAN_SIGMA_KERNEL") != "w":
        return "nerf_sigma_f16t_kernel"
    return KERNEL_OF[precision]


def pmc_traffic_per_launch(kernel, launch):
    """HBM bytes of one MLP launch from the committed PMC passes of this same command (FETCH_SIZE and WRITE_SIZE are
    collected in separate rocprofv3 runs, so they cannot be measured inside this process): FETCH_SIZE*2 (gfx950 reports
    half the bytes of wide streaming reads, MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes, of dispatch `launch` of
    `kernel` in a --steps 1 --warmup 0 --timed-only run (exactly one coarse and one fine launch).  None when absent."""
    for rel in PMC_SUMMARIES:                                     # newest first; the file used is named in the line (`traffic_source`)
        path = os.path.join(ROOT, rel)
        try:
            with open(path) as f:
                s = json.load(f)
            key = kernel.split("<")[0]
            fetch = [x["FETCH_SIZE"] for x in s["fetch"] if x["kernel"].split("<")[0] == key]
            write = [x["WRITE_SIZE"] for x in s["write"] if x["kernel"].split("<")[0] == key]
            return (2 * fetch[launch] + write[launch]) * 1024.0, rel
        except Exception:
            continue
    return None, None


def train_iteration(dev, origins, dirs, cap, rays=2048, iters=5):
    """median wall time of one training iteration: rays_per_batch random rays of the frame, 128 coarse + 256 fine samples through both
    networks forward and backward, photometric loss, Adam step (tools/train_step_bench.py is the stand-alone form)"""
    import torch.nn.functional as F
    from neuman_hip import ray_utils, render_utils, synthetic, train
    coarse, fine = synthetic.make_joiner(0).to(dev).train(), synthetic.make_joiner(1).to(dev).train()
    optim = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
    g = torch.Generator(device='cpu').manual_seed(0)
    target = synthetic.make_joiner(7).to(dev).eval()                       # the colours to fit: another net's rendering of the same rays (not noise)
    near = torch.full((rays,), float(cap.near['bkg']), device=dev)
    far = torch.full((rays,), float(cap.far['bkg']), device=dev)
    losses = []
    pool = []                                                              # the batches of every iteration below, drawn and rendered before the clock starts
    with torch.no_grad():
        for _ in range(2 * (2 + iters)):
            idx = torch.randint(0, origins.shape[0], (rays,), generator=g).to(dev)
            o, d = origins[idx].contiguous(), dirs[idx].contiguous()
            pool.append((o, d, render_utils.render_vanilla_rays(target, None, o, d, cap.near['bkg'], cap.far['bkg'], 32, 0, True)[0]))

    def step():
        o, d, color = pool[len(losses) % len(pool)]
        optim.zero_grad()
        pts, _, z = ray_utils.sample_z(o, d, near, far, S, want_points=True)
        out = coarse(pts, d[:, None, :].expand(pts.shape))
        rgb, _, _, w, _ = render_utils.raw2outputs(out, z, d, white_bkg=True)
        loss = F.mse_loss(rgb, color)
        with torch.no_grad():
            zf = ray_utils.importance_z(z, w.detach(), NI)
        ptsf = o[:, None, :] + d[:, None, :] * zf[..., None]
        outf = fine(ptsf, d[:, None, :].expand(ptsf.shape))
        loss = loss + F.mse_loss(render_utils.raw2outputs(outf, zf, d, white_bkg=True)[0], color)
        loss.backward()
        optim.step()
        losses.append(loss.detach())
        return loss

    def timed():
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            loss = step()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3, float(loss.detach())
    ms, _ = timed()
    keep = train.STORE16
    train.STORE16 = False                                           # the float32 copies of round 4, same kernels otherwise: the split, reported beside
    try:
        ms32, _ = timed()
    finally:
        train.STORE16 = keep
    evals = rays * (S + S + NI)
    return {"rays_per_batch": rays, "samples": [S, S + NI], "ms_per_iteration": ms, "iterations_per_s": 1e3 / ms,
            "mlp_tflops_fwd_bwd": evals * FLOP_PER_EVAL * 3 / ms / 1e9, "gemm_precision": train.GEMM_PRECISION,
            "fp16_storage": bool(keep), "ms_per_iteration_float32_storage": ms32,
            "loss_first": float(losses[0]), "loss_last": float(losses[-1]), "iterations_run": len(losses),
            "target": "the rays' colours rendered by a second net (seed 7, 32 samples): a field the trained nets approach", "what": "forward with saved activations (one kernel), the whole backward-data pass (one kernel), batched weight-gradient products, "
            "differentiable compositing, Adam (csrc/mlp.hip, mlp_bwd.hip, train.hip); what the step keeps between its passes is fp16 (activations x 32, dZ x a "
            "measured power of two) and the weight-gradient products are single fp16 MFMAs with float32 accumulation; ms_per_iteration_float32_storage = "
            "the same step with float32 copies (NEUMAN_TRAIN_STORE16=0, round 4's arithmetic: parameter gradients 6e-6 from the reference's autograd at the "
            "small golden; fp16 storage: 2e-5 .. 9e-5 at 131k / 262k evaluations, tests/test_hip_train16.py)"}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started by hand: become `torch.distributed.run --nproc-per-node N bench.py ...`.
    Returns only on error (after printing one JSON line)."""
    import socket
    n = args.gpus
    if args.backend == "nccl":
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(json.dumps({"error": f"--gpus {n} but {have} HIP device(s) visible", "n_gpus": n, "n_gpus_visible": have,
                              "metric": "rays_per_sec", "value": None}), flush=True)
            raise SystemExit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(cmd[0], cmd, env)


def launch_check(args, rank, world):
    """--launch-check: what a box without N GPUs can execute of the N > 1 start-up -- process group up (backend as asked), one
    frame-assembly gather of host tensors through the product's gather_frame, one JSON line from rank 0."""
    from neuman_hip import parallel
    total, tile = 1000, parallel.balanced_tile(1000, world, 64)
    idx = parallel.tile_ray_indices(total, tile, rank, world)
    frame = parallel.gather_frame(idx.to(torch.float32)[:, None] * 2.0, idx, total, tile)
    if rank == 0:
        ok = bool(torch.equal(frame[:, 0], torch.arange(total, dtype=torch.float32) * 2.0))
        print(json.dumps({"launch_check": True, "world": dist.get_world_size(), "backend": dist.get_backend(), "frame_assembled": ok, "tile": tile,
                          "tiles_per_rank": [int(parallel.tile_ray_indices(total, tile, r, world).shape[0] + tile - 1) // tile for r in range(world)]}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", default="mixed", choices=list(DTYPES))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-precisions", action="store_true",
                    help="skip the all-bf16x3 / all-i8x3 frame rates reported beside the headline")
    ap.add_argument("--dist", action="store_true", help="initialise the RCCL process group and run the gather collective at world size 1 too "
                    "(what a single-GPU box can execute of the N > 1 path; tests/test_parallel_gpu.py)")
    ap.add_argument("--timed-only", action="store_true", help="profiling runs: nothing but the warm-up and the timed steps "
                    "(no quality check, other precisions or CPU baseline), so every MLP launch rocprofv3 sees is a timed one")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (nccl = RCCL; gloo only with --launch-check)")
    ap.add_argument("--launch-check", action="store_true", help="bring the process group up, assemble one dummy frame through the gather "
                    "and exit (no device needed with --backend gloo: the CPU-side test of the self-launch)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(json.dumps({"error": f"--gpus {args.gpus} but WORLD_SIZE={world}", "n_gpus": args.gpus, "value": None}), flush=True)
        raise SystemExit(2)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if args.launch_check:
        if args.backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(args.backend, rank=rank, world_size=world,
                                **({"device_id": torch.device("cuda", local)} if args.backend == "nccl" else {}))
        return launch_check(args, rank, world)
    if args.backend != "nccl":
        raise SystemExit("--backend gloo is for --launch-check only: the timed path runs on HIP devices over RCCL")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists for the hot path)")
    torch.cuda.set_device(local)
    if world > 1 or args.dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from neuman_hip import parallel, ray_utils, render_utils, synthetic
    if args.dist:
        parallel.FORCE_COLLECTIVE = True                                   # the gather of a group of ONE rank is run too
    sharded = world > 1 or args.dist
    parallel.set_frame_sharding(sharded, stats=sharded)                    # the frame renderers' own sharding: the SAME tiling code as the drivers'
    dev = torch.device("cuda", local)
    coarse = synthetic.make_joiner(0).to(dev)
    fine = synthetic.make_joiner(1).to(dev)
    coarse.precision = fine.precision = args.precision
    cap = synthetic.SimpleCapture(W, H)
    origins, dirs = ray_utils.shot_all_rays_dev(cap, dev)                  # a1 on the device
    total = origins.shape[0]
    TILE = parallel.balanced_tile(total, world, parallel.FRAME_TILE)       # (what render_frame_sharded picks: reported, not used here)
    n_local = int(parallel.tile_ray_indices(total, TILE, rank, world, device=dev).shape[0])

    # HIP events around every MLP launch (same stream the kernel is launched on: torch's current stream)
    mlp_events = {"coarse": [], "fine": []}
    for name, net in (("coarse", coarse), ("fine", fine)):
        inner = net.forward_rays

        def timed(o, d, z, precision=None, sigma_scale=1.0, role=None, sigma_only=False, _inner=inner, _log=mlp_events[name]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = _inner(o, d, z, precision=precision, sigma_scale=sigma_scale, role=role, sigma_only=sigma_only)
            e1.record()
            _log.append((e0, e1, z.numel()))
            return out
        net.forward_rays = timed

    gather_events = []

    def step():
        """one frame exactly as render_vanilla renders it (render_utils.py:781-797): the frame's rays through render_utils._frame -- this
        device alone at N = 1; under the process group the rays of this rank's interleaved tiles (parallel.FRAME_TILE) and ONE gather
        assembling (rgb, depth) on rank 0 (parallel.render_frame_sharded, the code path of all four drivers)"""
        frame = render_utils._frame(lambda oo, dd: render_utils.render_vanilla_rays(coarse, fine, oo, dd, cap.near['bkg'], cap.far['bkg'], S, NI, True),
                                    origins, dirs)
        if sharded:
            gather_events.append(parallel.LAST_FRAME_STATS.get("_events"))
        return frame

    def sync():
        if world > 1 or args.dist:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        for log in mlp_events.values():
            log.clear()
        gather_events.clear()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1 or args.dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()

    # per rank: [fine evaluations, fine launch ms (sum), coarse evaluations, coarse ms, launches of each, rays, frame-assembly ms (sum)]
    def log_sums(which):
        log = mlp_events[which]
        return float(sum(n for _, _, n in log)), float(sum(e0.elapsed_time(e1) for e0, e1, _ in log))
    mine = torch.tensor([*log_sums("fine"), *log_sums("coarse"), float(len(mlp_events["fine"])), float(n_local),
                         float(sum(ev[1].elapsed_time(ev[2]) for ev in gather_events if ev is not None))], device=dev, dtype=torch.float64)
    per_rank = [mine]
    if world > 1 or args.dist:
        per_rank = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    per_rank = [r.tolist() for r in per_rank]
    slowest = max(range(world), key=lambda r: per_rank[r][1])               # the rank whose fine launches took longest bounds the frame

    def roofline(which, precision, launch_index):
        col = 0 if which == "fine" else 2
        density_only = which == "coarse" and precision in ("fp16x3", "bf16x3", "bf16")
        evals, ms = per_rank[slowest][col], per_rank[slowest][col + 1]
        n_launch = int(per_rank[slowest][4])
        kernel = kernel_of(which, precision)
        traffic, traffic_source = pmc_traffic_per_launch(kernel, launch_index) if world == 1 else (None, None)
        achieved = evals * FLOP_PER_EVAL / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # what the matrix pipe really executes: MFMA ops per algorithmic FLOP x the share of the layers evaluated, against the
        # dense peak of the MFMA type in use (MI355X_MICROARCH.md: bf16 / fp16 2.5 PFLOP/s; i8 ~2x the bf16 rate)
        issued = {"fp16x3": 3.0, "bf16x3": 3.0, "i8x3": 3.0, "bf16": 1.0}.get(precision, 0.0) * ((593408 - 102144) / 593408 if density_only else 1.0)
        hw_peak = 5000.0 if precision == "i8x3" else PEAK_BF16_TFLOPS
        mfma_type = {"i8x3": "i8 (v_mfma_i32_32x32x32_i8; 256 + 32 encoding inputs on bf16)", "fp16x3": "fp16 (v_mfma_f32_32x32x16_f16)"}.get(precision, "bf16")
        hardware = {"mfma_type": mfma_type, "mfma_ops_per_algorithmic_flop": issued, "rate": achieved * issued, "peak": hw_peak,
                    "unit": "Tops/s", "frac": achieved * issued / hw_peak} if issued else None
        share = (593408 - 102144) / 593408 if density_only else 1.0
        return {"bound": "mfma", "kernel": kernel, "hardware": hardware,
                "issued": None if not density_only else {
                    "what": "the same launch priced on the MACs it actually issues (491,264 of the 593,408 per evaluation: the density head only); "
                            "`achieved` / `frac` above price the reference's full evaluation, which this launch replaces",
                    "achieved": achieved * share, "frac": achieved * share / PEAK_BF16_TFLOPS, "unit": "TFLOP/s"},
                "launch": f"{which} pass, {int(evals) // max(1, n_launch)} evaluations per launch" + (f" on rank {slowest}, the slowest of {world}" if world > 1 else ""),
                "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_source if traffic_source else ("not collected at N > 1: PMC passes are single-process runs" if world > 1 else None),
                "traffic_unit": "bytes of HBM traffic per launch (FETCH_SIZE*2 + WRITE_SIZE) REPLAYED from the committed rocprofv3 --pmc passes of this "
                                "command named in traffic_source (counters need their own runs; not measured in this process); algorithmic: "
                                "16 B/evaluation out + 4 B/evaluation z in",
                "launches": n_launch, "avg_launch_ms": ms / max(1, n_launch),
                "note": "algorithmic FLOPs = 1,186,816 per MLP evaluation (what the reference performs); fp16x3, bf16x3 and i8x3 all issue 3 MFMAs "
                        "per algorithmic one (i8 at twice the bf16 rate), so hardware MFMA work is 3x the algorithmic figure"
                        + ("; this launch evaluates the density head only (nm_mlp_sigma_rays): the reference composites the coarse "
                           "colours and discards them (render_utils.py:139-141), so feature/views/rgb layers -- 102,144 of the 593,408 "
                           "MACs per evaluation -- are not issued; sigma is bit-identical" if density_only else "")}

    p_coarse = "fp16x3" if args.precision == "mixed" else args.precision
    p_fine = "i8x3" if args.precision == "mixed" else args.precision
    same_kernel = kernel_of("coarse", p_coarse) == kernel_of("fine", p_fine)  # then the fine launch is that kernel's second dispatch

    if rank == 0:
        rl_fine, rl_coarse = roofline("fine", p_fine, 1 if same_kernel else 0), roofline("coarse", p_coarse, 0)
        for net in (coarse, fine):
            net.__dict__.pop('forward_rays', None)              # drop the event-recording wrappers

        others, parity, base, pvr = None, None, None, None
        extras = args.precision != "fp32" and not args.timed_only
        if extras and world == 1 and not args.no_other_precisions:
            others = {}
            with torch.no_grad():
                for p in ("fp16x3", "bf16x3", "i8x3"):
                    if p == args.precision:
                        continue
                    coarse.precision = fine.precision = p
                    step()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        step()
                    torch.cuda.synchronize()
                    dtp = time.perf_counter() - t1
                    others[p] = {"value": total * args.steps / dtp, "unit": "rays/s", "ms_per_step": dtp / args.steps * 1e3}
            others["note"] = ("same frame and timing brackets, both passes in the named precision; fp16x3 = float32-class arithmetic everywhere, "
                              "bf16x3 = round 1's parity mode, i8x3 = the fast mode (its coarse-pass error moves importance samples: not "
                              "parity grade end to end); their parity against the oracle is under parity_vs_oracle.other_precisions; never `value`")
            coarse.precision = fine.precision = args.precision
        if extras and world == 1 and not args.no_cpu_baseline:
            base, oracle = cpu_baseline()
            with torch.no_grad():
                parity = parity_vs_oracle(oracle, coarse, fine, origins, dirs, args.precision)
                parity["what"] = (f"first {parity['rays']} rays of the timed frame, device ({args.precision}) vs the CPU oracle (float32 restatement of "
                                  "the reference, pinned on the reference's own outputs): f32 pixels before any quantisation; `attribution` = statements "
                                  "(a)-(d) of oracle/attribution.py, `attribution_statements_violated` must be empty; (d) is scored against the reference's own "
                                  "float64 frame of these rays (tests/golden/arbiter.npz)")
                if others is not None:
                    parity["other_precisions"] = {}
                    for p in ("fp16x3", "bf16x3", "i8x3"):
                        if p != args.precision:
                            parity["other_precisions"][p] = parity_vs_oracle(oracle, coarse, fine, origins, dirs, p, full=False)
            with torch.no_grad():
                pvr = parity_vs_reference(oracle["checker"], dev, origins, dirs, args.precision)
            pvr["timed_workload_first_4096_rays"] = {"device_vs_reference_f64": parity["attribution"]["d_device_vs_reference_f64"],
                                                     "reference_f32_vs_reference_f64": parity["attribution"]["d_reference_f32_vs_reference_f64"],
                                                     "oracle_vs_reference_f64": parity["attribution"]["d_oracle_vs_reference_f64"],
                                                     "allowed_rays_gt_1e-4": parity["attribution"]["d_allowed_rays_gt_1e-4"],
                                                     "note": "the timed workload's fine net is an independent random field (seed 1 against seed 0): ill conditioned; statement (d) of "
                                                             "oracle/attribution.py holds the device to the reference's own float32 count + a quarter"}
            parity["note"] = ("the inverse CDF of ray_utils.py:164-194 turns a coarse-weight difference d into a position difference d / pdf, and "
                              "the synthetic workload's fine net (an independent random field) turns that into colour: the reference's OWN float32 frame "
                              "is beyond 1e-4 of its float64 evaluation on attribution.d_reference_f32_vs_reference_f64 of these rays; "
                              "conditional on the sample positions -- either side's -- every ray is within 1e-4")
        workloads = None
        if extras and world == 1 and not args.no_other_precisions:
            # early ray termination (north star) on the workload that can show it: the 'opaque' preset (surface-like sigma; the
            # same field for the coarse and the fine net, as a trained pair agrees on where the surfaces are).  Same frame, same
            # sample counts; eps = 0 is the reference's semantics (every sample evaluated), eps = 1e-4 drops rays whose
            # transmittance fell below it at a 32-sample chunk boundary.  Never `value`.
            net = synthetic.make_joiner(1, preset='opaque').to(dev)
            net.precision = args.precision
            res = {}
            with torch.no_grad():
                for eps in (0.0, 1e-4):
                    render_utils.TERMINATION_EPS = eps
                    try:
                        tr = {}
                        rgb_e, _ = render_utils.render_vanilla_rays(net, net, origins, dirs, 0.0, 3.14, S, NI, True, trace=tr)
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        for _ in range(args.steps):
                            render_utils.render_vanilla_rays(net, net, origins, dirs, 0.0, 3.14, S, NI, True)
                        torch.cuda.synchronize()
                        dte = (time.perf_counter() - t1) / args.steps
                    finally:
                        render_utils.TERMINATION_EPS = 0.0
                    res[eps] = (rgb_e, dte, tr.get('march', [None])[0], tr.get('march_coarse', [None])[0])
            st = res[1e-4][2]
            workloads = {"opaque_preset_early_termination": {
                "what": "800x800, 128 + 128 samples, synthetic.make_joiner(1, preset='opaque') as coarse and fine net; both passes marched front to "
                        "back in chunks with ballot / prefix-sum compaction of the live rays between chunks (nm_mlp_sigma_ray_chunk / "
                        "nm_mlp_forward_ray_chunk, nm_transmittance_chunk, nm_compact_hits): the fine pass cut at eps, the coarse pass at transmittance 4e-13 "
                        "(render_utils.TERMINATION_COARSE: under half a float32 ulp of sample_pdf's 1e-5, so the importance samples are bit-identical); one host read per chunk (the live count: no launch once nobody is live, chunk halved "
                        "to 16 samples while rays are being cut)",
                "eps": 1e-4, "rays_per_s_every_sample": total / res[0.0][1], "rays_per_s_terminated": total / res[1e-4][1],
                "speedup": res[0.0][1] / res[1e-4][1], "fine_evaluations_done": st['evaluated'] / st['total'] if st else None,
                "coarse_evaluations_done": (lambda c_: c_['evaluated'] / c_['total'] if c_ else None)(res[1e-4][3]),
                "rgb_linf_vs_every_sample": float((res[0.0][0] - res[1e-4][0]).abs().max())}}
            # one iteration of the background trainer (trainers/vanilla_nerf_trainer.py:45-96 + backward + Adam) at the
            # reference's batch, on the training slice's default arithmetic.  Never `value`.
            workloads["background_trainer_iteration"] = train_iteration(dev, origins, dirs, cap)
        line = {
            "metric": "rays_per_sec (800x800 frame, 128 samples/ray coarse + 128 importance, NeuMan background NeRF)",
            "value": total * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": DTYPES[args.precision], "data": "synthetic",
            "config": {"workload": "BASELINE config 2: background NeRF (models/vanilla.py 8x256, posenc), 800x800 = 640000 rays, "
                                   "128 coarse + 256 fine MLP evaluations per ray, synthetic-dense weights (seeds 0/1), near 0 far 3.14",
                       "rays_per_frame": total, "mlp_evals_per_ray": EVALS_PER_RAY,
                       "parallelism": f"ray-tile sharding x{world}, 1 gather/frame", "tile_rays": TILE, "precision": args.precision},
            "multi_gpu": {"rccl_world": dist.get_world_size() if (world > 1 or args.dist) else 1,
                          "backend": dist.get_backend() if (world > 1 or args.dist) else None,
                          "tiles_per_rank": [int(-(-r[5] // TILE)) for r in per_rank], "rays_per_rank": [int(r[5]) for r in per_rank],
                          "fine_launch_ms_per_rank": [r[1] / max(1.0, r[4]) for r in per_rank],
                          "coarse_launch_ms_per_rank": [r[3] / max(1.0, r[4]) for r in per_rank],
                          "frame_assembly_ms_per_rank": [r[6] / max(1, args.steps) for r in per_rank],
                          "frame_assembly": "render_utils._frame -> parallel.render_frame_sharded (the drivers' own path): one dist.gather per frame to rank 0 + one "
                                            "index_select; at N = 1 without a process group the frame's ray list is rendered in place, nothing to assemble",
                          "omitted_at_n_gt_1": None if world == 1 else ["cpu_baseline (rank 0 at N = 1 only)", "parity_vs_oracle (scored in the N = 1 run: the "
                                               "ranks run the same kernels on disjoint rays)", "roofline.traffic (PMC passes are single-process)"]},
            "parity_vs_oracle": parity,
            "parity_vs_reference": pvr,
            "other_precisions": others,
            "other_workloads": workloads,
            "roofline": rl_fine,
            "roofline_coarse": rl_coarse,
        }
        if base is not None:
            line["cpu_baseline"] = base
        print(json.dumps(line), flush=True)
    if world > 1 or args.dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
