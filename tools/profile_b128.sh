#!/bin/bash
# GPU box: rocprofv3 kernel stats of BASELINE configs[1] (B = 128, landmarks only) -> gpurun_out/b128/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b128; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o k -- python "$R/bench.py" --batch 128 --lmk-only --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$O/log.txt" 2>&1
echo rc=$?
find "$R/gpurun_out/b128" -name "k_kernel_trace.csv" -delete
tail -1 "$O/log.txt" | cut -c1-200
