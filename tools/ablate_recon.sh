for a in 2 10 11; do
  export SYN_ABLATE_RECON=$a
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ablr$a -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "ablate=$a $(grep 'recon_kernel(' /tmp/ablr$a/*kernel_stats.csv | sed -e 's/.*)",//' | cut -d, -f1-6)"
done
