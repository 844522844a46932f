#!/bin/bash
# phase profile (s_memtime) of conv_lt_kernel: one ResNet-50 forward at B = 512 with the LT_PROF variant library; build it first (here, no GPU needed):
#   tools/build_variant.sh ltprof resnet_kernels.hip synergynet_amd/csrc/resnet_kernels.hip -DLT_PROF=1
R=$GRAFT_REPO_ROOT
export SYNERGY_HIP_LIB=$R/synergynet_amd/libsynergy_hip_ltprof.so
python $R/bench.py --arch resnet50 --batch 512 --steps 1 --warmup 0 --prewarm 0 --no-cpu-baseline --no-extras --overlap 0 2>&1 | grep "^lt<" | sort | uniq -c | sort -rn | head -60 > $R/gpurun_out/lt_prof.txt
cat $R/gpurun_out/lt_prof.txt
