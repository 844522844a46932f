#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c6; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -k "pipelined_slices" 2>&1 | grep -v "input bound" | tail -25 ) > $O/pytest_pipe.txt
for i in 1 2; do for p in "1 10" "0 10" "1 30"; do set -- $p; printf "pipe=%s smax=%s " $1 $2; for b in 128 1 32 96 256 512; do SYN_LB4_PIPE=$1 SYN_LB4_PIPE_SMAX=$2 timeout 120 python bench.py --lmk-only --batch $b --steps 200 --warmup 20 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d %.4f' % (d['config']['global_batch'], d['ms_per_step']), end='  ')"; done; echo; done; done > $O/b128.txt 2>&1
cat $O/pytest_pipe.txt $O/b128.txt
