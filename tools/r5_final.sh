#!/bin/bash
# round 5, final GPU call: the whole GPU suite (no -x), the smoke entry, then the measurement bundle (tools/round_profile.sh r5)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "input bound" | tail -6 ) > $O/pytest_gpu_final.txt
cat $O/pytest_gpu_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
timeout 1500 bash tools/round_profile.sh r5 2>&1 | grep -E "rc=|^\{" | cut -c1-200
