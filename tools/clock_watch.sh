#!/bin/bash
# Sample GPU clock / power while a sustained MLP load runs: tools/clock_watch.sh <precision>
# (evidence for DESIGN.md section 6: is the k-loop clock- / power-limited?)
prec=${1:-bf16x3}
python - "$prec" <<'PY' &
import sys, os, time
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import torch
from neuman_hip import synthetic
prec = sys.argv[1]
n = 128 * 256 * 40
j = synthetic.make_joiner(0).cuda()
g = torch.Generator(device='cuda').manual_seed(0)
pts = (torch.rand((n, 3), device='cuda', generator=g) * 3 - 1.5).contiguous()
dirs = torch.nn.functional.normalize(torch.randn((n, 3), device='cuda', generator=g), dim=-1).contiguous()
t0 = time.time()
it = 0
while time.time() - t0 < 12:
    for _ in range(50):
        j(pts, dirs, precision=prec)
    torch.cuda.synchronize()
    it += 50
dt = time.time() - t0
print(f"{prec}: {it * n * 1186816 / dt / 1e12:.1f} TFLOP/s algorithmic sustained over {dt:.1f} s")
PY
pid=$!
sleep 6
for i in 1 2 3; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr -s ' ' | head -6
  sleep 1.5
done
wait $pid
