#!/bin/bash
# Round-end measurement bundle on the GPU box: bench JSON, rocprofv3 kernel stats, HBM traffic PMC passes.
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round_$tag; mkdir -p $O
python $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o k -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o k -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1; echo "write rc=$?"
rm -f $O/*/k_kernel_trace.csv
python - <<PY
import csv, glob, collections, json
out = {}
for name in ('fetch', 'write'):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob('$O/%s/*counter_collection.csv' % name):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']; agg[k] += float(r['Counter_Value']); cnt[k] += 1
    out[name] = {k: agg[k] / cnt[k] for k in agg}
    for k in sorted(agg, key=lambda k: -agg[k])[:20]:
        print(name, f'{agg[k]/cnt[k]:12.1f} KB/launch x{cnt[k]:3d}', k[:110])
json.dump(out, open('$O/traffic_raw.json', 'w'), indent=1)
PY
