#!/bin/bash
# round 5, GPU call 3: full GPU suite; L2 warm-up of the weight runs (lb4 chain, lb chain, head) against builds without it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c3; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 700 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | grep -v "input bound" | tail -30 ) > $O/pytest_gpu.txt
timeout 600 bash tools/ab_perlaunch.sh lb4nt lbnt headnt lb4nt lbnt headnt > /dev/null 2>&1; cp $R/gpurun_out/ab_perlaunch.txt $O/
cat $O/pytest_gpu.txt $O/ab_perlaunch.txt
