#!/bin/bash
# Run on the GPU box: bash tools/power_probe_precision.sh <tag>
#   socket power and shader clock (rocm-smi, 5 Hz) while bench.py renders frames in the shipped mixed mode, at pure bf16x3 (3 MFMAs per product) and at
#   the 2-MFMA measurement variant (fp16x2): does the matrix work saved show up as time at the SAME power (the frame is power capped), as lower
#   power, or as a higher clock?  -> gpurun_out/<tag>_power_<precision>.json lines
set -u
REPO=$(pwd); TAG=$1; OUT=$REPO/gpurun_out; mkdir -p $OUT
for prec in ${PRECS:-mixed bf16x3 fp16x2}; do
  ( for i in $(seq 1 80); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.2; done ) > $OUT/${TAG}_smi_$prec.txt &
  SMI=$!
  python bench.py --precision $prec --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_$prec.json 2> /dev/null
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  python - <<PY
import json,re
d=json.loads(open("$OUT/${TAG}_bench_$prec.json").read().strip().splitlines()[-1])
txt=open("$OUT/${TAG}_smi_$prec.txt").read()
pw=[float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", txt)]
ck=[int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)]
busy=[p for p in pw if p > 600]
ckb=[c for c in ck if c > 1000]
print(json.dumps({"precision":"$prec","rays_per_s":d["value"],"ms_per_step":d["ms_per_step"],"k2_ms_per_step":d["config"]["mlp_kernel_ms_per_step"]["k_sdf_only"],
  "power_w_max":max(pw) if pw else None,"power_w_median_while_rendering":sorted(busy)[len(busy)//2] if busy else None,
  "sclk_mhz_median_while_rendering":sorted(ckb)[len(ckb)//2] if ckb else None,"sclk_mhz_min_max":[min(ck),max(ck)] if ck else None,"samples":len(pw)}))
PY
done
