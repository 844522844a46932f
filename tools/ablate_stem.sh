for a in 0 16; do
  export SYNERGY_HIP_TEST_KNOBS=ablate_stem=$a
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl$a -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "ablate=$a $(grep stem_block1 /tmp/abl$a/*kernel_stats.csv | sed -e 's/.*)",//' | cut -d, -f1-3)"
done
