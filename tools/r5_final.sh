#!/bin/bash
# round 5, final GPU call: the whole GPU suite (no -x), then the measurement bundle (tools/round_profile.sh r5)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "input bound" | tail -15 ) > $O/pytest_gpu_final.txt
cat $O/pytest_gpu_final.txt
timeout 1500 bash tools/round_profile.sh r5 2>&1 | tail -60
