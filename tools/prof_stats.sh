#!/bin/bash
# usage (on the GPU box via gpurun): bash tools/prof_stats.sh <tag> [bench args]
# runs bench.py plain, then under rocprofv3 --kernel-trace --stats, and prints a per-kernel table
tag=$1; shift
R=$GRAFT_REPO_ROOT
python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.err; echo "bench rc=$?"; cat $R/gpurun_out/bench_$tag.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$tag.log 2>&1; echo "prof rc=$?"
ls $R/gpurun_out/prof_$tag
python - <<PY
import csv, glob, collections
fs = glob.glob('$R/gpurun_out/prof_$tag/*kernel_stats.csv')
if fs:
    print(open(fs[0]).read()[:6000])
PY
