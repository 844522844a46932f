#!/bin/bash
# rocprofv3 kernel stats of the mesh-consumer pipeline (tools/bench_render.py 64) -> gpurun_out/render_prof
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/render_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python $R/tools/bench_render.py 64 > $O/run.log 2>&1; echo "rc=$?"
rm -f $O/k_kernel_trace.csv
python - <<PY
import csv
for r in list(csv.DictReader(open('$O/k_kernel_stats.csv')))[:10]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
