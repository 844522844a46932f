#!/bin/bash
# GPU box: reconstruction-alone time of several library builds, interleaved over rounds (process-to-process noise on one box is ~5 %).
# usage: tools/ab_recon.sh "<tag> <tag> ..." [rounds] [B]    ("base" = the in-tree libsynergy_hip.so; SYN_* env passes through)
tags=$1; rounds=${2:-4}; B=${3:-1024}
for i in $(seq $rounds); do
  for t in $tags; do
    lib=$GRAFT_REPO_ROOT/synergynet_amd/libsynergy_hip_$t.so; [ $t = base ] && lib=$GRAFT_REPO_ROOT/synergynet_amd/libsynergy_hip.so
    printf "%-8s " $t; SYNERGY_HIP_LIB=$lib python $GRAFT_REPO_ROOT/tools/time_recon.py $B 400 2>/dev/null | tail -1
  done
done
