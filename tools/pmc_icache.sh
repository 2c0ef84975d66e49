#!/bin/bash
# Instruction-cache counters over a 4M-point nerfart_sdf_fwd (bf16x3): is the 86 KB straight-line kernel fetch-bound?
# usage (GPU box): bash tools/pmc_icache.sh   -> gpurun_out/pmc_icache_*.txt
set -u
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cat > /tmp/mb.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from nerfart_amd import scene, hip
m, rk, fn = scene.build_model("VolSDF", device="cuda", precision="bf16x3")
blob, _ = m.packed()
pts = torch.rand(4*1024*1024, 3, device="cuda") * 4 - 2
for _ in range(3): hip.sdf_fwd(blob, pts, 3.0, precision=1)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmci$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmci$i -o r -- python /tmp/mb.py > /tmp/pmci$i.log 2>&1
  db=$(find /tmp/pmci$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $REPO/tools/prof_summary.py $db $REPO/gpurun_out/pmc_icache_$i.txt "icache pmc set $i: $set" > /dev/null; else tail -5 /tmp/pmci$i.log > $REPO/gpurun_out/pmc_icache_$i.txt; fi
done
