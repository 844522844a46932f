#!/bin/bash
# usage: bash tools/pmc_pass.sh <tag> "<counter list>"   (PMC pass only: no --stats / traces besides kernel-trace)
tag=$1; ctrs=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag -o $tag -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1; echo "pmc rc=$?"
ls $R/gpurun_out/pmc_$tag
python - <<PY
import csv, glob, collections
fs = glob.glob('$R/gpurun_out/pmc_$tag/*counter_collection.csv')
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:110]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
names = sorted({c for k in agg for c in agg[k]})
print('kernel | ' + ' | '.join(names))
for k in agg:
    print(k, '|', ' | '.join(f"{agg[k][c]/max(cnt[(k,c)],1):.4g}" for c in names))
PY
