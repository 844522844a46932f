#!/bin/bash
# GPU box: the small-batch chain (features.8-14, one face per workgroup) with four against eight streams per face: parity test once, then the
# landmarks-only step time at a few batch sizes, interleaved.  usage: bash tools/small_ns_ab.sh
R=$GRAFT_REPO_ROOT
echo "test NS=8: $(cd $R && SYN_SMALL_NS=8 timeout 250 python -m pytest tests/test_gpu_parity.py -q -x -k 'small_batch_chain or ragged_batches_match or (across_their_batch_thresholds and (128 or 224))' < /dev/null 2>&1 | tail -1)"
for rep in 1 2; do for ns in 4 8; do
  printf "NS=%s " $ns
  for b in 128 1 8 64 256; do SYN_SMALL_NS=$ns python $R/bench.py --lmk-only --batch $b --steps 200 --warmup 20 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d %.4f' % (d['config']['global_batch'], d['ms_per_step']), end='  ')"; done; echo
done; done
