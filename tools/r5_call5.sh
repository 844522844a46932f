#!/bin/bash
# round 5, GPU call 5: pipelined slices of features.15-17 (small batches): bit test first (bounded), then the suite, then B = 128 / 1 / 256 step times against SYN_LB4_PIPE=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c5; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -k "pipelined_slices" 2>&1 | grep -v "input bound" | tail -25 ) > $O/pytest_pipe.txt
( timeout 800 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | grep -v "input bound" | tail -25 ) > $O/pytest_gpu.txt
for i in 1 2; do for p in 1 0; do printf "pipe=%s " $p; for b in 128 1 32 256 512; do SYN_LB4_PIPE=$p timeout 120 python bench.py --lmk-only --batch $b --steps 200 --warmup 20 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d %.4f' % (d['config']['global_batch'], d['ms_per_step']), end='  ')"; done; echo; done; done > $O/b128.txt 2>&1
cat $O/pytest_pipe.txt $O/pytest_gpu.txt $O/b128.txt
