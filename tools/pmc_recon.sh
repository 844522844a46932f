#!/bin/bash
# GPU box: HBM bytes of the reconstruction kernel for the packed and the pitched output (FETCH_SIZE / WRITE_SIZE in passes of their own)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in packed ""; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pr; timeout 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pr -o k -- python $R/tools/time_recon.py 1024 5 $mode > /tmp/pr.log 2>&1
    python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('/tmp/pr/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'recon_f16_kernel<4' in r['Kernel_Name']: agg[r['Kernel_Name'][:60] + ' ' + r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items(): print('${mode:-pitched}', k, 'launches', len(v), 'avg', sum(v) / len(v))
PY
  done
done
