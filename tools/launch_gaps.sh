#!/bin/bash
# GPU box: idle time between consecutive kernels of the single-stream step (rocprofv3 kernel trace timestamps).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gaps; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --overlap 0 > $O/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob('$O/**/k_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'syn::' in r['Kernel_Name']]
# last full step: find the last 'stem' kernel start
idx = [i for i, r in enumerate(rows) if 'stem' in r['Kernel_Name']]
a = idx[-2]; b = idx[-1]
seq = rows[a:b]
tot_k = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seq)
span = int(rows[b]['Start_Timestamp']) - int(seq[0]['Start_Timestamp'])
print('kernels in step:', len(seq), 'sum of kernel durations %.1f us' % (tot_k / 1e3), 'step span %.1f us' % (span / 1e3), 'idle %.1f us' % ((span - tot_k) / 1e3))
for x, y in zip(seq, seq[1:] + [rows[b]]):
    print('%-60s dur %7.1f us  gap after %6.1f us' % (x['Kernel_Name'][10:70], (int(x['End_Timestamp']) - int(x['Start_Timestamp'])) / 1e3, (int(y['Start_Timestamp']) - int(x['End_Timestamp'])) / 1e3))
PY
rm -rf $O/*/ 2>/dev/null
