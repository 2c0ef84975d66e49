#!/bin/bash
# Run on the GPU box:  bash tools/profile_train.sh <tag>
#   fine-tune step (tools/bench_train.py: cfg 3 at 480x270, native CLIP / style / VGG kernels + native pass 2) timed plainly, then
#   under rocprofv3 --kernel-trace --stats -> gpurun_out/<tag>_train_step.json, gpurun_out/<tag>_train_kernel_stats.txt
set -u
REPO=$(pwd); TAG=$1; OUT=$REPO/gpurun_out; mkdir -p $OUT
python tools/bench_train.py --steps 2 > $OUT/${TAG}_train_step.json 2> $OUT/${TAG}_train.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_train && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_train -o r -- python $REPO/tools/bench_train.py --steps 1 > /tmp/kt_train.log 2>&1
db=$(find /tmp/kt_train -name "*.db" | head -1)
[ -n "$db" ] && python $REPO/tools/prof_summary.py $db $OUT/${TAG}_train_kernel_stats.txt "$TAG: python tools/bench_train.py --steps 1 (1x MI355X) under rocprofv3 --kernel-trace --stats" > /dev/null
tail -2 $OUT/${TAG}_train_step.json
# PMC passes over the same step (separate runs: FETCH_SIZE and WRITE_SIZE do not fit one pass) -> <tag>_train_pmc_summary.json
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmt$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmt$i -o r -- python $REPO/tools/bench_train.py --steps 1 > /tmp/pmt$i.log 2>&1
  db=$(find /tmp/pmt$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $REPO/tools/prof_summary.py $db $OUT/${TAG}_train_pmc_set$i.txt "$TAG train pmc pass $i ($set): python tools/bench_train.py --steps 1" > /dev/null; else tail -5 /tmp/pmt$i.log > $OUT/${TAG}_train_pmc_set$i.txt; fi
done
python $REPO/tools/pmc_summary.py $OUT/${TAG}_train_pmc_summary.json $OUT/${TAG}_train_pmc_set*.txt
