#!/bin/bash
# Run on the GPU box: bash tools/power_probe.sh <tag>
#   socket power and shader clock (rocm-smi, 5 Hz) while bench.py renders frames with the default 8-wave K2 and with NERFART_K2=w32:
#   is the sustained K2 rate set by the power cap (clock below the 2.4 GHz nominal) rather than by issue slots?
set -u
REPO=$(pwd); TAG=$1; OUT=$REPO/gpurun_out; mkdir -p $OUT
for var in v1 w32; do
  ( for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.2; done ) > $OUT/${TAG}_smi_$var.txt &
  SMI=$!
  NERFART_K2=$var python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_$var.json 2> /dev/null
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  python - <<PY
import json,re
d=json.loads(open("$OUT/${TAG}_bench_$var.json").read().strip().splitlines()[-1])
pw=[float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", open("$OUT/${TAG}_smi_$var.txt").read())]
ck=[int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", open("$OUT/${TAG}_smi_$var.txt").read())]
print(json.dumps({"k2":"$var","rays_per_s":d["value"],"ms_per_step":d["ms_per_step"],"k2_ms_per_step":d["config"]["mlp_kernel_ms_per_step"]["k_sdf_only"],
  "power_w_max":max(pw) if pw else None,"power_w_median":sorted(pw)[len(pw)//2] if pw else None,"sclk_mhz_min_max":[min(ck),max(ck)] if ck else None,"samples":len(pw)}))
PY
done
head -3 $OUT/${TAG}_smi_v1.txt
