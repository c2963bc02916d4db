#!/bin/bash
# Evidence for the kernels beside the headline path (run through gpurun):  tools/profile_extras.sh r01
#   gpurun_out/<tag>/warp_pmc.json           search_kernel counters on the C3-posed workload (tools/warp_probe.py), --pmc passes with
#                                            --kernel-trace only
#   gpurun_out/<tag>/warp_probe.json         its timing line
#   gpurun_out/<tag>/train_kernel_stats.csv  rocprofv3 --kernel-trace --stats of tools/train_step_bench.py
#   gpurun_out/<tag>/train_step.jsonl        training iteration times (2048 and 4096 rays, f32; 2048 rays, bf16x3)
set -u
TAG=${1:-r00}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd $R
python tools/warp_probe.py > $OUT/warp_probe.json 2>/dev/null
rm -rf /tmp/px_$TAG && mkdir -p /tmp/px_$TAG
i=0
for c in "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_BRANCH GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/px_$TAG/pmc$i -o w -- python tools/warp_probe.py > /dev/null 2>&1
done
python - <<PY
import csv, json
out = {}
for i in (1, 2):
    rows = [r for r in csv.DictReader(open('/tmp/px_$TAG/pmc%d/w_counter_collection.csv' % i)) if 'search_kernel' in r['Kernel_Name']]
    last = rows[-1]['Dispatch_Id']
    for r in rows:
        if r['Dispatch_Id'] == last:
            out[r['Counter_Name']] = float(r['Counter_Value'])
            out.setdefault('duration_ms_pass%d' % i, (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e6)
cyc = out['GRBM_GUI_ACTIVE'] / 8
out['derived'] = {'cycles_per_xcd': cyc, 'clock_ghz': cyc / out['duration_ms_pass2'] / 1e6,
                  'valu_busy_frac': out['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc), 'valu_instructions_per_wave': out['SQ_INSTS_VALU'] / out['SQ_WAVES']}
out['note'] = ("search_kernel<true>, last dispatch of tools/warp_probe.py (7,340,288 samples, SMPL-size mesh) under rocprofv3 --pmc <one group per "
               "pass> --kernel-trace; SQ_ACTIVE_INST_VALU counts quad-cycles: x4 / (1024 SIMDs x cycles) = fraction of VALU issue slots busy")
json.dump(out, open('$OUT/warp_pmc.json', 'w'), indent=1)
print(json.dumps(out['derived']))
PY
(python tools/train_step_bench.py; python tools/train_step_bench.py 4096; NEUMAN_TRAIN_GEMM=bf16x3 python tools/train_step_bench.py) 2>/dev/null | grep '^{' > $OUT/train_step.jsonl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px_$TAG/train -o t -- python tools/train_step_bench.py > /dev/null 2>&1
cp /tmp/px_$TAG/train/t_kernel_stats.csv $OUT/train_kernel_stats.csv
cat $OUT/train_step.jsonl | cut -c 1-160
