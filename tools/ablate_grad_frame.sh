#!/bin/bash
# What the softplus' scratch round trip of k_sdf_grad_bf16 costs IN THE FRAME and in the NeuS frame (VERDICT r05 next 3 ii): the real library against the
# timing-only variants of tools/ablate_grad.py (results wrong by construction), same box, same call.   python tools/ablate_grad.py build   first (here).
out=${1:-gpurun_out/ablate_grad_frame.log}
: > "$out"
for v in full no_scratch no_scratch_stores no_scratch_loads; do
  lib="$PWD/nerfart_amd/csrc/libnerfart_hip.so"; [ "$v" != full ] && lib="$PWD/gpurun_ablate/libgrad_$v.so"
  for rep in 1 2; do
    echo "== $v (rep $rep): VolSDF frame (bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2)" >> "$out"
    NERFART_HIP_LIB=$lib python bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'ms_per_step': d['ms_per_step'], 'mlp_kernel_ms_per_step': d['config']['mlp_kernel_ms_per_step']}))" >> "$out"
    echo "== $v (rep $rep): NeuS frame (tools/bench_neus.py --steps 4)" >> "$out"
    NERFART_HIP_LIB=$lib python tools/bench_neus.py --steps 4 2>/dev/null | tail -1 >> "$out"
  done
done
cat "$out"
