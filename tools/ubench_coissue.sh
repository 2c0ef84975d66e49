#!/bin/bash
# GPU box: run the co-issue micro-benchmark (built here by: hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench_coissue tools/ubench_coissue.hip)
mkdir -p gpurun_out
timeout 300 tools/_build/ubench_coissue > gpurun_out/${1:-r02s}_ubench_coissue.txt 2>&1
echo "rc $?"
tail -5 gpurun_out/${1:-r02s}_ubench_coissue.txt
