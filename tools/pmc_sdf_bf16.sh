#!/bin/bash
# PMC passes over a 4M-point nerfart_sdf_fwd (bf16x3) micro-benchmark; summaries -> gpurun_out/pmc_*.txt
# usage (GPU box): bash tools/pmc_sdf_bf16.sh [precision]
set -u
REPO=$(pwd)
PREC=${1:-bf16x3}
mkdir -p $REPO/gpurun_out
cat > /tmp/mb.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from nerfart_amd import scene, hip
m, rk, fn = scene.build_model("VolSDF", device="cuda", precision="$PREC")
blob, _ = m.packed()
pts = torch.rand(4*1024*1024, 3, device="cuda") * 4 - 2
for _ in range(3): hip.sdf_fwd(blob, pts, 3.0, precision=hip.PRECISIONS["$PREC"])
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $REPO/gpurun_out/pmc_list.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc$i -o r -- python /tmp/mb.py > /tmp/pmc$i.log 2>&1
  db=$(find /tmp/pmc$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $REPO/tools/prof_summary.py $db $REPO/gpurun_out/pmc_${PREC}_$i.txt "pmc set $i: $set" > /dev/null; else tail -5 /tmp/pmc$i.log > $REPO/gpurun_out/pmc_${PREC}_$i.txt; fi
done
