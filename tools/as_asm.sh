#!/bin/bash
# ISA of the activation-stationary i8 kernel -> /tmp/asm/i8s.s, with its spill counts: tools/as_asm.sh [-DFLAG ..]
cd /root/repo/ml-neuman_amd && mkdir -p /tmp/asm && python - "$@" <<'PY' 2>&1 | grep -v "warning\|bsteps\|\^\|generated"
import build as B, subprocess, os, sys
subprocess.run([B.HIPCC]+B.FLAGS+sys.argv[1:]+['-S','--cuda-device-only','-o','/tmp/asm/i8s.s',os.path.join(B.CSRC,'mlp_i8s.hip')],check=True)
PY
grep "vgpr_spill_count\|codeLenInByte" /tmp/asm/i8s.s
awk '/^\.LBB/{lab=$1} /scratch_/{c[lab]++} /v_mfma/{m[lab]++} END{for(l in c) if (m[l]>0) print l, c[l], "mfma", m[l]+0}' /tmp/asm/i8s.s | sort -t_ -k2 -n | awk '{printf "%s | ", $0} END{print ""}'
