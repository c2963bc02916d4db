"""Timing of the merge + composite tail at the C4 / C5 list sizes: nm_merge_composite_lists (one kernel, merged list in LDS) against
nm_merge_sorted list by list + nm_composite (the merged list through HBM).  One JSON line per size."""
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import torch  # noqa: E402
from neuman_hip import render_utils as R  # noqa: E402


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rays, sizes in ((166000, (256, 128)), (1 << 19, (320, 192, 192, 192))):
    g = torch.Generator(device='cuda').manual_seed(0)
    zs = [torch.sort(torch.rand((rays, S), device='cuda', generator=g) * 3 + 0.2 * i, dim=1)[0].contiguous() for i, S in enumerate(sizes)]
    raws = [torch.randn((rays, S, 4), device='cuda', generator=g).contiguous() for S in sizes]
    d = torch.nn.functional.normalize(torch.randn((rays, 3), device='cuda', generator=g), dim=-1).contiguous()

    def old():
        z, raw = zs[0], raws[0]
        for a, b in zip(zs[1:], raws[1:]):
            z, raw = R.merge_sorted(z, raw, a, b)
        return R.raw2outputs(raw, z, d, want_weights=False)

    def new():
        return R.merge_composite_lists(zs, raws, d, True)
    a, b = old(), new()
    same = torch.equal(a[0], b[0]) and torch.equal(a[4], b[1]) and torch.equal(a[2], b[2])
    nbytes = rays * sum(sizes) * 20
    t_old, t_new = t(old), t(new)
    print(json.dumps({"rays": rays, "lists": sizes, "bit_identical": same, "merge_sorted_x%d_plus_composite_ms" % (len(sizes) - 1): t_old, "merge_composite_lists_ms": t_new,
                      "algorithmic_GBps": nbytes / t_new / 1e6}), flush=True)
