#!/bin/bash
# GPU box: B = 128 / 1 / 256 landmarks-only step time for variant libraries, interleaved.  usage: tools/b128ab.sh "<tag> <tag>"  (base = in-tree)
R=$GRAFT_REPO_ROOT
for i in 1 2; do for t in $1; do
  lib=$R/synergynet_amd/libsynergy_hip_$t.so; [ $t = base ] && lib=$R/synergynet_amd/libsynergy_hip.so
  for rm in 2047 1023; do
    printf "%-6s rm=%s " $t $rm
    for b in 128 1 8 31; do SYNERGY_HIP_LIB=$lib SYNERGY_HIP_EARLY_RM=$rm python $R/bench.py --lmk-only --batch $b --steps 200 --warmup 20 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d %.4f' % (d['config']['global_batch'], d['ms_per_step']), end='  ')"; done; echo
  done
done; done
