#!/bin/bash
# B = 128 landmarks-only step (BASELINE configs[1]), one stream: ms/step of the default library and of variant libraries (tags), alternating
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/b128.txt; : > $out
for rep in 1 2; do for tag in "" "$@"; do
  if [ -n "$tag" ]; then export SYNERGY_HIP_LIB=$R/synergynet_amd/libsynergy_hip_$tag.so; else unset SYNERGY_HIP_LIB; fi
  echo -n "${tag:-default}: " >> $out
  python $R/bench.py --lmk-only --batch 128 --steps 300 --warmup 30 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $out
done; done
cat $out
