"""Time the product shapes of one training pass (n = 524288 rows: the fine pass of 2048 rays x 256 samples) on the three GEMM entry
points: which shapes are where the iteration's time goes.  python tools/gemm_shapes.py [n]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ml-neuman_amd"))
import torch  # noqa: E402
from neuman_hip import _lib, train  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)   # noqa: E731
ws = torch.empty(64 << 20, device=dev)
cases = [
    ("forward 256x256 (+bias, relu)", (0, 0, n, 256, 256), dict(flags=train.BIAS | train.RELU)),
    ("forward K=64 first layer", (0, 0, n, 256, 64), dict(flags=train.BIAS | train.RELU)),
    ("forward N=4 head (K=256)", (0, 0, n, 4, 256), dict(flags=train.BIAS)),
    ("forward N=128 views (K=256)", (0, 0, n, 128, 256), {}),
    ("backward-data 256x256 + mask", (0, 1, n, 256, 256), dict(flags=train.MASK)),
    ("backward-data K=4 (from d_raw)", (0, 1, n, 256, 4), dict(flags=train.MASK)),
    ("backward-weights 256x256", (1, 1, 256, 256, n), {}),
    ("backward-weights M=4 (d_raw)", (1, 1, 4, 256, n), {}),
    ("backward-weights K=64 (first layer)", (1, 1, 256, 64, n), {}),
]
for name, (akm, bkm, M, N, K), kw in cases:
    A = rnd(K, M) if akm else rnd(M, K)
    B = rnd(K, N) if bkm else rnd(N, K)
    C = torch.empty((M, N), device=dev)
    bias = rnd(N) if kw.get('flags', 0) & train.BIAS else None
    mask = rnd(M, N) if kw.get('flags', 0) & train.MASK else None
    row = [f"{name:38s}"]
    for prec in ('f32', 'bf16x3', 'fp16x3'):
        def run():
            train._gemm(akm, bkm, M, N, K, A, A.shape[1], B, B.shape[1], C, N, bias=bias, mask=mask, ldmask=N, flags=kw.get('flags', 0), ws=ws, precision=prec)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        byts = 4 * (M * K + N * K + M * N * (2 if mask is not None else 1))
        row.append(f"{prec} {ms * 1e3:7.0f} us {2 * M * N * K / ms / 1e9:6.0f} TF/s {byts / ms / 1e9:5.2f} TB/s")
    print(" | ".join(row))
