#!/bin/bash
# per-launch durations of one ResNet-50 forward (B = 512) for the default library and each variant library given (tags): A/B of conv_lt_kernel
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/ab_lt.txt; : > $out
for tag in "" "$@"; do
  unset SYNERGY_HIP_LIB SYNERGY_HIP_RESNET_GEMM SYNERGY_HIP_TEST_KNOBS SYNERGY_HIP_RESNET_FUSE
  case "$tag" in
    "") ;;
    env:*) export "${tag#env:}" ;;
    *) export SYNERGY_HIP_LIB=$R/synergynet_amd/libsynergy_hip_$tag.so ;;
  esac
  bash $R/tools/resnet_layers.sh > /dev/null 2>&1
  echo "== ${tag:-default}" >> $out
  python - >> $out <<PY
import re
rows=[l.split() for l in open('$R/gpurun_out/resnet_layers.txt')]
us=[float(r[-2]) for r in rows]
names=[r[0] for r in rows]
print('total %.0f us; conv_lt sum %.0f' % (sum(us), sum(u for n,u in zip(names,us) if 'conv_lt' in n)))
print(' '.join('%.0f' % u for u in us))
print(' '.join(n[:12] for n in names[:12]))
PY
done
cat $out
