#!/bin/bash
# per-launch times of the B = 1024 backbone (bench.py roofline.per_launch, HIP events in the library) for the default library and variant libraries (tags)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/ab_perlaunch.txt; : > $out
for tag in "" "$@" ""; do
  if [ -n "$tag" ]; then export SYNERGY_HIP_LIB=$R/synergynet_amd/libsynergy_hip_$tag.so; else unset SYNERGY_HIP_LIB; fi
  echo "== ${tag:-default}" >> $out
  python $R/bench.py --steps 100 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], 'backbone', r['backbone']['ms'], ' '.join('%s:%.1f' % (p['feature'], p['ms']*1e3) for p in r['per_launch']))" >> $out
done
cat $out
