#!/bin/bash
# Run on the GPU box: bash tools/power_probe_ubench.sh <tag>
#   tools/_build/ubench_power (MFMA-only / + fragment reads / + VALU streams, 6 s each) with the socket power sampled at ~5 Hz beside it
set -u
REPO=$(pwd); TAG=$1; OUT=$REPO/gpurun_out; mkdir -p $OUT
( while true; do echo "$(date +%s.%N) $(rocm-smi --showpower 2>/dev/null | grep -o 'Power (W): [0-9.]*' | head -1)"; sleep 0.15; done ) > $OUT/${TAG}_ubench_smi.txt &
SMI=$!
( while IFS= read -r line; do echo "$(date +%s.%N) $line"; done < <(tools/_build/ubench_power 6) ) > $OUT/${TAG}_ubench_power.txt
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
python - <<PY
import re, json
ev=[(float(l.split()[0]), l.split(" ",1)[1].strip()) for l in open("$OUT/${TAG}_ubench_power.txt") if l.strip()]
pw=[(float(l.split()[0]), float(re.search(r"Power \(W\): ([0-9.]+)", l).group(1))) for l in open("$OUT/${TAG}_ubench_smi.txt") if "Power" in l]
prev=ev[0][0]
for t, line in ev[1:]:
    d=json.loads(line)
    # power samples of the second half of this mode's run
    s=[p for (tp,p) in pw if prev + (t-prev)/2 <= tp <= t]
    d["power_w_median_second_half"]=sorted(s)[len(s)//2] if s else None
    d["power_w_max"]=max(s) if s else None
    print(json.dumps(d))
    prev=t
PY
