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
