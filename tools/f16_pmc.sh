#!/bin/bash
# rocprofv3 counters of the coarse (density-only fp16x3) launch: cycles (GRBM_GUI_ACTIVE / 8 per XCD), MFMA pipe busy, duration -> effective clock
# usage (through gpurun): tools/f16_pmc.sh w t f16nowread ...   (w: nerf_mlp_kernel; t: nerf_sigma_f16t_kernel; anything else: that variant library with =t)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
for k in "$@"; do
  rm -rf /tmp/pmc_$k
  lib=""; sk=$k
  if [ "$k" != w ] && [ "$k" != t ]; then lib=$R/ml-neuman_amd/lib/exp/libneuman_hip_$k.so; sk=t; fi
  for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
    n=$(echo $c | cut -d" " -f1)
    ( cd $R; NEUMAN_HIP_LIB=$lib NEUMAN_SIGMA_KERNEL=$sk timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$k/$n -o p -- python tools/coarse_time.py > /dev/null 2>&1 )
  done
  python - <<PY
import csv, json, glob
out = {"kernel": "$k"}
for f in glob.glob('/tmp/pmc_$k/*/p_counter_collection.csv'):
    rows = [r for r in csv.DictReader(open(f)) if 'nerf_mlp_kernel' in r['Kernel_Name'] or 'nerf_sigma_f16t' in r['Kernel_Name']]
    if not rows: continue
    big = max(rows, key=lambda r: float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    for r in rows:
        if r['Dispatch_Id'] == big['Dispatch_Id']:
            out[r['Counter_Name']] = float(r['Counter_Value'])
            out['ms_' + f.split('/')[-2]] = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e6
if 'GRBM_GUI_ACTIVE' in out:
    cyc = out['GRBM_GUI_ACTIVE'] / 8
    out['cycles_per_xcd'] = cyc
    out['clock_ghz'] = cyc / out['ms_GRBM_GUI_ACTIVE'] / 1e6
    out['mfma_busy_frac'] = out['SQ_VALU_MFMA_BUSY_CYCLES'] / (256 * 4 * cyc)
print(json.dumps(out))
PY
done
