#!/bin/bash
# phase profile (s_memtime) of conv_lt_kernel: one ResNet-50 forward at B = 512 with the LT_PROF variant library
R=$GRAFT_REPO_ROOT
export SYNERGY_HIP_LIB=$R/synergynet_amd/libsynergy_hip_ltprof.so
python $R/bench.py --arch resnet50 --batch 512 --steps 1 --warmup 0 --prewarm 0 --no-cpu-baseline --no-extras --overlap 0 2>&1 | grep "^lt<" | sort | uniq -c | sort -rn | head -60 > $R/gpurun_out/lt_prof.txt
cat $R/gpurun_out/lt_prof.txt
