for a in 0 8; do
  export SYN_ABLATE_HEAD=$a
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ablh$a -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "ablate=$a $(grep head_kernel /tmp/ablh$a/*kernel_stats.csv | sed -e 's/.*)",//' | cut -d, -f1-3)"
done
