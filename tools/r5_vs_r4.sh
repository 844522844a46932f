#!/bin/bash
# one box: round-4 tree (git worktree _r4 at 9ac575e, its own library) against this tree -- headline, per-launch backbone, one stream, B = 128 / 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_vs_r4.txt; cd $R; : > $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d.get('roofline') or {}
pl=r.get('per_launch') or []
print(d['value'], d['ms_per_step'], 'backbone', (r.get('backbone') or {}).get('ms'), 'frac', r.get('frac'), ' '.join('%s:%.1f' % (p['feature'], p['ms']*1e3) for p in pl))"; }
for i in 1 2; do for t in r5 r4; do
  d=$R; [ $t = r4 ] && d=$R/_r4
  echo "== $t two replicas:  $(cd $d && python bench.py --steps 100 --no-cpu-baseline --no-extras 2>/dev/null | line)" >> $O
  echo "== $t one stream:    $(cd $d && python bench.py --steps 100 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | line | cut -d' ' -f1-2)" >> $O
  echo "== $t B=128 lmk-only: $(cd $d && python bench.py --lmk-only --batch 128 --steps 200 --warmup 20 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | line | cut -d' ' -f1-2)" >> $O
  echo "== $t B=1 lmk-only:   $(cd $d && python bench.py --lmk-only --batch 1 --steps 200 --warmup 20 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | line | cut -d' ' -f1-2)" >> $O
  echo "== $t resnet50 B=512: $(cd $d && python bench.py --arch resnet50 --batch 512 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | line | cut -d' ' -f1-4)" >> $O
done; done
cat $O
