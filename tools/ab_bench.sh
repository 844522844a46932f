#!/bin/bash
# GPU box: A/B of two builds of the library on the SAME box (boxes differ by +-3 %): per-launch times of
# synergynet_amd/libsynergy_hip_prev.so (built from an older commit: tools/build_prev.sh <commit>) vs the current one, interleaved.
for i in 1 2 3; do
  for lib in libsynergy_hip_prev.so libsynergy_hip.so; do
    SYNERGY_HIP_LIB=$GRAFT_REPO_ROOT/synergynet_amd/$lib python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib'.ljust(24), d['ms_per_step'], d['roofline']['backbone']['ms'], ' '.join(f\"{p['feature']}:{p['ms']*1e3:.0f}\" for p in d['roofline']['per_launch']))"
  done
done
