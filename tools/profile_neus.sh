#!/bin/bash
# Run on the GPU box:  bash tools/profile_neus.sh <tag>
#   cfg 4 (tools/bench_neus.py: NeuS 480x270, 64 + 64 spp) timed plainly, under rocprofv3 --kernel-trace --stats, and one PMC pass
#   (matrix-core busy + clocks) -> gpurun_out/<tag>_neus_line.json, <tag>_neus_kernel_stats.txt, <tag>_neus_pmc_set1.txt, <tag>_neus_pmc_summary.json
set -u
REPO=$(pwd); TAG=$1; OUT=$REPO/gpurun_out; mkdir -p $OUT
python tools/bench_neus.py --steps 3 > $OUT/${TAG}_neus_line.json 2> $OUT/${TAG}_neus.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_neus && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_neus -o r -- python $REPO/tools/bench_neus.py --steps 3 > /tmp/kt_neus.log 2>&1
db=$(find /tmp/kt_neus -name "*.db" | head -1)
[ -n "$db" ] && python $REPO/tools/prof_summary.py $db $OUT/${TAG}_neus_kernel_stats.txt "$TAG: python tools/bench_neus.py --steps 3 (1x MI355X) under rocprofv3 --kernel-trace --stats" > /dev/null
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmn$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmn$i -o r -- python $REPO/tools/bench_neus.py --steps 1 > /tmp/pmn$i.log 2>&1
  db=$(find /tmp/pmn$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $REPO/tools/prof_summary.py $db $OUT/${TAG}_neus_pmc_set$i.txt "$TAG neus pmc pass $i ($set): python tools/bench_neus.py --steps 1" > /dev/null; else tail -5 /tmp/pmn$i.log > $OUT/${TAG}_neus_pmc_set$i.txt; fi
done
python $REPO/tools/pmc_summary.py $OUT/${TAG}_neus_pmc_summary.json $OUT/${TAG}_neus_pmc_set*.txt
cat $OUT/${TAG}_neus_line.json
