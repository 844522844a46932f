#!/bin/bash
# GPU box: per-launch times of several library builds, interleaved (boxes differ by +-3 %, runs on one box do not).
# usage: tools/ab_variants.sh "<tag> <tag> ..." [bench args]     ("base" = the in-tree libsynergy_hip.so)
tags=$1; shift
for i in 1 2; do
  for t in $tags; do
    lib=$GRAFT_REPO_ROOT/synergynet_amd/libsynergy_hip_$t.so; [ $t = base ] && lib=$GRAFT_REPO_ROOT/synergynet_amd/libsynergy_hip.so
    printf "%-8s " $t; SYNERGY_HIP_LIB=$lib python $GRAFT_REPO_ROOT/tools/perlaunch.py "$@"
  done
done
