#!/bin/bash
# round 5, GPU call 2: full GPU suite on the new default library (stem with four service waves, features.5/6 with two service waves per unit,
# features.7 in the small chain, landmarks + pose in one launch), per-launch A/B against the old forms, timing-only ablations of the lb4 chain
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c2; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 700 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | grep -v "input bound" | tail -40 ) > $O/pytest_gpu.txt
timeout 600 bash tools/ab_perlaunch.sh stemsv1 r5svc1 lb4abl1 lb4abl3 lb4abl4 lb4abl8 lb4abl16 lb4abl36 lb4abl63 > /dev/null 2>&1; cp $R/gpurun_out/ab_perlaunch.txt $O/
for i in 1 2; do for b in 128 1; do timeout 120 python bench.py --lmk-only --batch $b --steps 200 --warmup 20 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d %.4f' % (d['config']['global_batch'], d['ms_per_step']), end='  ')"; done; echo; done > $O/b128.txt 2>&1
cat $O/pytest_gpu.txt $O/ab_perlaunch.txt $O/b128.txt
