#!/bin/bash
# round 5, GPU call 4: full GPU suite (features.5 + 6 in one launch, pruned kernels), pair against two launches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c4; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 800 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | grep -v "input bound" | tail -30 ) > $O/pytest_gpu.txt
timeout 400 bash tools/ab_env.sh SYN_RM_PAIR56=0 r5svc1 SYN_RM_PAIR56=0 > /dev/null 2>&1; cp $R/gpurun_out/ab_env.txt $O/
cat $O/pytest_gpu.txt $O/ab_env.txt
