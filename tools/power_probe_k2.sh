#!/bin/bash
# Run on the GPU box: bash tools/power_probe_k2.sh <tag>
#   socket power and shader clock (rocm-smi, 5 Hz) while ONE kernel runs in a loop for ~6 s: K2 (sdf only) on 4 M random points at C-ABI precision 1
#   (split-bf16, 3 MFMAs per product), 4 (2 MFMAs) and 5 (1 MFMA, compensated one-term weights) - is a kernel at the package power cap (the split-bf16
#   kernels: DESIGN.md 4.1b) or at its issue port?  -> gpurun_out/<tag>_power_k2.json lines
set -u
REPO=$(pwd); TAG=$1; OUT=$REPO/gpurun_out; mkdir -p $OUT
: > $OUT/${TAG}_power_k2.json
for prec in 1 4 5; do
  ( for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.2; done ) > $OUT/${TAG}_smi_k2_$prec.txt &
  SMI=$!
  python - <<PY > $OUT/${TAG}_k2_$prec.txt
import sys, time, torch
sys.path.insert(0, "$REPO")
from nerfart_amd import scene, hip
m, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device="cuda:0", precision="bf16x3")
if $prec == 1: blob = m.packed()[0]
elif $prec == 4: blob = m.set_sampler_precision("fp16x2").packed_sampler()[0]
else: blob = m.set_sampler_precision("fp16x1c", guard=0.005, late_round=3).packed_sampler()[0]
x = ((torch.rand(1 << 22, 3, generator=torch.Generator().manual_seed(0)) * 2 - 1) * 1.5).cuda()
hip.sdf_fwd(blob, x, 3.0, precision=$prec); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(20): hip.sdf_fwd(blob, x, 3.0, precision=$prec)
    torch.cuda.synchronize(); n += 20
print((time.perf_counter() - t0) / n * 1e3)
PY
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  python - <<PY >> $OUT/${TAG}_power_k2.json
import json,re
txt=open("$OUT/${TAG}_smi_k2_$prec.txt").read()
pw=[float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", txt)]
ck=[int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)]
busy=sorted(p for p in pw if p > 500); ckb=sorted(c for c, p in zip(ck, pw) if p > 500) if len(ck) == len(pw) else sorted(c for c in ck if c > 1000)
print(json.dumps({"k2_precision": $prec, "ms_per_4M_points_sustained": round(float(open("$OUT/${TAG}_k2_$prec.txt").read().split()[-1]), 3),
  "power_w_median_while_running": busy[len(busy)//2] if busy else None, "power_w_max": max(pw) if pw else None,
  "sclk_mhz_median_while_running": ckb[len(ckb)//2] if ckb else None, "samples": len(pw)}))
PY
done
cat $OUT/${TAG}_power_k2.json
