#!/bin/bash
# GPU box: row-band variants at small batches.  Each argument is "<stem_band>,<batch>[,<rm_band2>,<rm_band3>]" (SYNERGY_HIP_TEST_KNOBS names) (stem: 10 U + NBD
# or 0 = the tiled stem; features.2 / 3 bands: 1 / 0); prints the per-launch microseconds (tools/perlaunch.py), interleaved twice.
# TEST=1: also run the batch-threshold parity tests under each setting.
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/band_ab.txt; : > $out
for rep in 1 2; do for c in "$@"; do
  IFS=, read st bs b2 b3 <<< "$c"; bs=${bs:-128}
  export SYNERGY_HIP_TEST_KNOBS=stem_band=$st,rm_band2=${b2:-1},rm_band3=${b3:-1}
  if [ "$TEST" = 1 ] && [ $rep = 1 ]; then
    echo "== $c test: $(cd $R && timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k 'across_their_batch_thresholds or ragged_batches_match or every_feature or u8_ingest' < /dev/null 2>&1 | tail -1)" >> $out
  fi
  echo "== $c B=$bs: $(timeout 120 python $R/tools/perlaunch.py --lmk-only --batch $bs --steps 200 --warmup 20 --overlap 0 < /dev/null 2>&1 | tail -1)" >> $out
done; done
cat $out
