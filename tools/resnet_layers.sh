#!/bin/bash
# GPU box: per-dispatch kernel durations of one ResNet-50 forward (B = 512) in launch order -> gpurun_out/resnet_layers.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rn_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o k -- python $R/bench.py --arch resnet50 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --overlap 0 > $O/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob('$O/**/k_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# last forward: find the last stem kernel
idx = [i for i, r in enumerate(rows) if 'resnet_stem' in r['Kernel_Name']]
start = idx[-1]
out = []
for r in rows[start:start + 60]:
    name = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    short = name.split('(')[0].replace('void syn::', '').replace('syn::', '')[:40]
    out.append('%-42s grid %-8s %8.1f us' % (short, r.get('Grid_Size', r.get('Grid_Size_X', '?')), d))
    if 'recon' in name or 'pool_fc' in name: break
open('$R/gpurun_out/resnet_layers.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $O
