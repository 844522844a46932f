#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/det_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python $R/tools/bench_detector.py > $O/run.log 2>&1; echo "rc=$?"
rm -f $O/k_kernel_trace.csv
python - <<PY
import csv
for r in list(csv.DictReader(open('$O/k_kernel_stats.csv')))[:8]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f} pct={r['Percentage']}")
PY
