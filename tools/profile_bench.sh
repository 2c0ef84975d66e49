#!/bin/bash
# Run on the GPU box:  bash tools/profile_bench.sh <tag> [bench args...]
#   1. plain bench line                        -> gpurun_out/<tag>_bench_line.json
#   2. rocprofv3 --kernel-trace --stats        -> gpurun_out/<tag>_bench_kernel_stats.txt (+ the bench line under rocprof)
#   3. rocprofv3 --pmc passes (separate runs)  -> gpurun_out/<tag>_pmc_<set>.txt, gpurun_out/<tag>_pmc_summary.json
# Only text summaries leave the box (the .db files stay in /tmp).
set -u
REPO=$(pwd)
TAG=$1; shift
ARGS="$@"
OUT=$REPO/gpurun_out
mkdir -p $OUT
python bench.py $ARGS > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $REPO/bench.py $ARGS --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_line_under_rocprof.json 2> /tmp/kt.err
db=$(find /tmp/kt -name "*.db" | head -1)
[ -n "$db" ] && python $REPO/tools/prof_summary.py $db $OUT/${TAG}_bench_kernel_stats.txt "$TAG: python bench.py $ARGS (1x MI355X) under rocprofv3 --kernel-trace --stats" > /dev/null
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf /tmp/pm$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm$i -o r -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary ${PMC_ARGS:-} > /tmp/pm$i.log 2>&1
  db=$(find /tmp/pm$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $REPO/tools/prof_summary.py $db $OUT/${TAG}_pmc_set$i.txt "$TAG pmc pass $i ($set): python bench.py --steps 1 --warmup 1 ${PMC_ARGS:-}" > /dev/null; else tail -5 /tmp/pm$i.log > $OUT/${TAG}_pmc_set$i.txt; fi
done
python $REPO/tools/pmc_summary.py $OUT/${TAG}_pmc_summary.json $OUT/${TAG}_pmc_set*.txt
