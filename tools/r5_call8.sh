#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c8; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 800 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | grep -v "input bound" | tail -12 ) > $O/pytest_gpu.txt
timeout 400 bash tools/ab_env.sh lbdw1 lbdw1 > /dev/null 2>&1; cp $R/gpurun_out/ab_env.txt $O/
for i in 1 2; do for v in base lbdw1; do lib=$R/synergynet_amd/libsynergy_hip.so; [ $v = lbdw1 ] && lib=$R/synergynet_amd/libsynergy_hip_lbdw1.so; printf "%-6s " $v; for b in 128 1 256; do SYNERGY_HIP_LIB=$lib timeout 120 python bench.py --lmk-only --batch $b --steps 200 --warmup 20 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d %.4f' % (d['config']['global_batch'], d['ms_per_step']), end='  ')"; done; echo; done; done > $O/b128.txt 2>&1
cat $O/pytest_gpu.txt $O/ab_env.txt $O/b128.txt
