#!/bin/bash
# Build synergynet_amd/libsynergy_hip_<tag>.so = the current objects with ONE source replaced by a variant file (A/B runs on one box
# through SYNERGY_HIP_LIB, see tools/ab_bench.sh).  usage: tools/build_variant.sh <tag> <source name in csrc, e.g. fused_block_rm.hip> <variant file> [extra hipcc flags]
set -e
tag=$1; name=$2; var=$3; shift 3
R=$(cd $(dirname $0)/.. && pwd)
python -c "from synergynet_amd.build import build_library; build_library()" >/dev/null
mkdir -p $R/synergynet_amd/_obj/variants
o=$R/synergynet_amd/_obj/variants/${name%.hip}_$tag.o      # variant objects never share a directory with the default build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I$R/synergynet_amd/csrc -c $var -o $o "$@"
objs=$(python - <<PY
import os
from synergynet_amd.build import SOURCES, OBJ
print(' '.join(os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES if s != '$name'))
PY
)
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/synergynet_amd/libsynergy_hip_$tag.so $objs $o
echo built synergynet_amd/libsynergy_hip_$tag.so
