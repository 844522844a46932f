#!/bin/bash
# Build synergynet_amd/libsynergy_hip_prev.so from the csrc/ of another commit (default HEAD~1) for tools/ab_bench.sh.
set -e
c=${1:-HEAD~1}
d=$(mktemp -d)
git archive $c synergynet_amd/csrc include | tar -x -C $d
srcs=$(python - <<PY
from synergynet_amd.build import SOURCES
print(' '.join('$d/synergynet_amd/csrc/' + s for s in SOURCES))
PY
)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -I$d/include -o synergynet_amd/libsynergy_hip_prev.so $srcs
rm -rf $d
echo built synergynet_amd/libsynergy_hip_prev.so from $c
