#!/bin/bash
# per-launch times of the B = 1024 backbone for a list of variants, interleaved default first and last (one box).  A variant is
#   <tag>            -> SYNERGY_HIP_LIB = synergynet_amd/libsynergy_hip_<tag>.so (tools/build_variant.sh)
#   <NAME>=<VALUE>   -> that environment variable set, default library
# usage: bash tools/ab_env.sh [--batch N] <variant> ...      output: gpurun_out/ab_env.txt
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}; out=$R/gpurun_out/ab_env.txt; mkdir -p $R/gpurun_out; : > $out
extra=""; if [ "$1" = "--batch" ]; then extra="--batch $2"; shift 2; fi
for v in "" "$@" ""; do
  unset SYNERGY_HIP_LIB; envs=""
  case "$v" in "") ;; *=*) envs="$v";; *) export SYNERGY_HIP_LIB=$R/synergynet_amd/libsynergy_hip_$v.so;; esac
  echo "== ${v:-default}" >> $out
  env $envs python $R/bench.py --steps 100 --no-cpu-baseline --no-extras $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], 'backbone', r['backbone']['ms'], ' '.join('%s:%.1f' % (p['feature'], p['ms']*1e3) for p in r['per_launch']))" >> $out
done
cat $out
