"""The background trainer's loop end to end on a synthetic scene resident in HBM (neuman_hip/ray_batches.py + bkg_trainer.py):
`captures` frames of width x height, `rays` rays per batch drawn on the device from all of them, S coarse + S + NI fine samples.
Prints one JSON line: batch assembly time, iteration time, iterations/s.
    python tools/train_loop_bench.py [captures] [width] [height] [rays] [S] [NI]"""
import json
import os
import sys
import time
import types

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neuman_hip import bkg_trainer, data_io, ray_batches, synthetic  # noqa: E402

C, W, H, R, S, NI = (int(a) for a in (sys.argv[1:] + [None] * 6)[:6] if a is not None) if len(sys.argv) == 7 else (100, 1280, 720, 4096, 128, 128)
rng = np.random.default_rng(0)
caps = []
yy, xx = np.mgrid[0:H, 0:W]
for i in range(C):
    ang = 0.02 * (i - C / 2)
    q = np.array([np.cos(ang / 2), 0.0, np.sin(ang / 2), 0.0], np.float32)
    cap = data_io.Capture(f'/nowhere/images/{i:05d}.png', data_io.PinholeCamera(W, H, 1.25 * W, 1.25 * W, W / 2, H / 2),
                          data_io.CameraPose(np.array([0.01 * i, 0.0, 0.0], np.float32), q), frame_id={'frame_id': i, 'total_frames': C})
    cap.image = np.stack([(xx * 255 // W), (yy * 255 // H), np.full_like(xx, (2 * i) % 256)], -1).astype(np.uint8)
    cap.mask = (((yy - H / 2) / (H / 3)) ** 2 + ((xx - W / 2 - i) / (W / 10)) ** 2 < 1).astype(np.uint8)
    cap.depth_map = rng.uniform(0.5, 3.0, size=(H, W)).astype(np.float32)
    cap.near, cap.far = {'bkg': 0.0}, {'bkg': 3.14}
    caps.append(cap)
t0 = time.perf_counter()
store = ray_batches.FrameStore(caps, 'cuda', dilation=30)
torch.cuda.synchronize()
t_store = time.perf_counter() - t0
opt = types.SimpleNamespace(samples_per_ray=S, importance_samples_per_ray=NI, perturb=1.0, raw_noise_std=0.0, white_bkg=True, margin=0.8,
                            penalize_empty_space=0.1, empty_space_loss_fn='mse', delay_iters=0, lrate_decay=250, learning_rate=5e-4, ablate_nerft=False,
                            rays_per_batch=R, max_iter=10, valid_iter=0, out=None)
coarse, fine = synthetic.make_joiner(0).cuda().train(), synthetic.make_joiner(1).cuda().train()
optim = torch.optim.Adam([{"params": coarse.parameters(), "lr": 5e-4}, {"params": fine.parameters(), "lr": 5e-4}])
batches = ray_batches.BackgroundRayBatcher(opt, store, draws='device', seed=0)
batches.next_batch()                                             # builds the candidate pool once
torch.cuda.synchronize()
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    b = batches.next_batch()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
batch_ms = sorted(ts)[len(ts) // 2] * 1e3
tr = bkg_trainer.BackgroundNeRFTrainer(opt, coarse, optim, fine_net=fine, batches=batches)
for _ in range(2):
    tr.train_batch(batches())
torch.cuda.synchronize()
ts = []
for _ in range(8):
    t0 = time.perf_counter()
    rep = tr.train_batch(batches())
    tr.iteration += 1
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
it_ms = sorted(ts)[len(ts) // 2] * 1e3
print(json.dumps({"captures": C, "image": [W, H], "rays_per_batch": R, "samples": [S, S + NI],
                  "scene_resident_bytes": int(sum(t.numel() * t.element_size() for t in (store.images, store.masks, store.border, store.depth))),
                  "frame_store_build_s": t_store, "batch_assembly_ms": batch_ms, "iteration_ms": it_ms, "iterations_per_s": 1e3 / it_ms,
                  "batch_share_of_iteration": batch_ms / it_ms, "loss": rep['total_loss'],
                  "note": "batch = draws on the device from all captures, colour/depth gathers, one nm_shot_rays_cams launch; the reference builds "
                          "the same batch on the host per iteration (np.argwhere over every capture's mask, datasets/background_rays.py:47-101)"}))
