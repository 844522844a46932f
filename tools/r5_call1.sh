#!/bin/bash
# round 5, GPU call 1: full GPU suite on the new default library, the r5-prep variants (never run before), per-launch A/B of the two large-batch changes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c1; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest_gpu.txt
( SYN_SMALL_F7=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "small_batch_chain or ragged_batches_match or across_their_batch_thresholds" 2>&1 | tail -8 ) > $O/pytest_f7.txt
( SYNERGY_HIP_LIB=$R/synergynet_amd/libsynergy_hip_split.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "small_batch_chain or ragged_batches_match" 2>&1 | tail -8 ) > $O/pytest_split.txt
( SYN_SMALL_F7=1 SYNERGY_HIP_LIB=$R/synergynet_amd/libsynergy_hip_split.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "small_batch_chain or ragged_batches_match" 2>&1 | tail -8 ) > $O/pytest_split_f7.txt
timeout 400 bash tools/ab_perlaunch.sh lb4v1 rmmed3 > /dev/null 2>&1; cp $R/gpurun_out/ab_perlaunch.txt $O/
# B = 128 / 1 landmarks-only, one stream: base | SMALL_F7 | split | split + SMALL_F7, interleaved twice
for i in 1 2; do for v in base f7 split splitf7; do
  lib=$R/synergynet_amd/libsynergy_hip.so; f7=0
  case $v in f7) f7=1;; split) lib=$R/synergynet_amd/libsynergy_hip_split.so;; splitf7) lib=$R/synergynet_amd/libsynergy_hip_split.so; f7=1;; esac
  printf "%-8s " $v
  for b in 128 1; do SYN_SMALL_F7=$f7 SYNERGY_HIP_LIB=$lib timeout 120 python bench.py --lmk-only --batch $b --steps 200 --warmup 20 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d %.4f' % (d['config']['global_batch'], d['ms_per_step']), end='  ')"; done; echo
done; done > $O/b128ab.txt 2>&1
cat $O/pytest_gpu.txt $O/pytest_f7.txt $O/pytest_split.txt $O/pytest_split_f7.txt $O/ab_perlaunch.txt $O/b128ab.txt
