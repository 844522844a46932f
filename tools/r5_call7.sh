#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c7; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 800 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | grep -v "input bound" | tail -25 ) > $O/pytest_gpu.txt
timeout 400 bash tools/ab_env.sh SYN_RM_PAIR34=0 SYN_RM_PAIR34=0 > /dev/null 2>&1; cp $R/gpurun_out/ab_env.txt $O/
timeout 200 bash tools/ab_env.sh --batch 2307 SYN_RM_PAIR34=0 > /dev/null 2>&1; cp $R/gpurun_out/ab_env.txt $O/ab_env_2307.txt
cat $O/pytest_gpu.txt $O/ab_env.txt $O/ab_env_2307.txt
