#!/bin/bash
# Run on the GPU box:  bash tools/profile_clip.sh <tag>
#   B = 16 forward + backward of the native CLIP image encoder: plain timing, rocprofv3 kernel trace, one PMC pass
#   (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE)  -> gpurun_out/<tag>_clip_*.{json,txt}
set -u
REPO=$(pwd); TAG=$1; OUT=$REPO/gpurun_out; mkdir -p $OUT
python tools/bench_clip.py 16 50 > $OUT/${TAG}_clip_b16.json 2> $OUT/${TAG}_clip.err
python tools/bench_clip.py 64 20 >> $OUT/${TAG}_clip_b16.json 2>> $OUT/${TAG}_clip.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_clip && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_clip -o r -- python $REPO/tools/bench_clip.py 16 20 > /tmp/kt_clip.log 2>&1
db=$(find /tmp/kt_clip -name "*.db" | head -1)
[ -n "$db" ] && python $REPO/tools/prof_summary.py $db $OUT/${TAG}_clip_kernel_stats.txt "$TAG: python tools/bench_clip.py 16 20 (B = 16 fwd + bwd x 23 calls, 1x MI355X) under rocprofv3 --kernel-trace --stats" > /dev/null
rm -rf /tmp/pm_clip && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pm_clip -o r -- python $REPO/tools/bench_clip.py 16 20 > /tmp/pm_clip.log 2>&1
db=$(find /tmp/pm_clip -name "*.db" | head -1)
if [ -n "$db" ]; then python $REPO/tools/prof_summary.py $db $OUT/${TAG}_clip_pmc.txt "$TAG pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE): python tools/bench_clip.py 16 20" > /dev/null; else tail -5 /tmp/pm_clip.log > $OUT/${TAG}_clip_pmc.txt; fi
cat $OUT/${TAG}_clip_b16.json
