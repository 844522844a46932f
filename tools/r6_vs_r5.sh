#!/bin/bash
# one box: round-5 tree (git worktree _r5 at 472c784, its own library: `git worktree add _r5 472c784 && (cd _r5 && python -c "import __graft_entry__ as g; g.build()")`
# in the authoring container) against this tree -- headline, per-launch backbone, one stream, B = 128 / 1, ResNet-50 step and backbone forward
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_vs_r5.txt; cd $R; : > $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d.get('roofline') or {}
pl=r.get('per_launch') or []
print(d['value'], d['ms_per_step'], 'backbone', (r.get('backbone') or {}).get('ms'), 'frac', r.get('frac'), ' '.join('%s:%.1f' % (p['feature'], p['ms']*1e3) for p in pl))"; }
for i in 1 2; do for t in r6 r5; do
  d=$R; [ $t = r5 ] && d=$R/_r5
  echo "== $t two replicas:  $(cd $d && python bench.py --steps 100 --no-cpu-baseline --no-extras 2>/dev/null | line)" >> $O
  echo "== $t one stream:    $(cd $d && python bench.py --steps 100 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | line | cut -d' ' -f1-2)" >> $O
  echo "== $t B=128 lmk-only: $(cd $d && python bench.py --lmk-only --batch 128 --steps 200 --warmup 20 --overlap 0 --no-cpu-baseline --no-extras 2>/dev/null | line | cut -d' ' -f1-2)" >> $O
  echo "== $t resnet50 B=512 step: $(cd $d && python bench.py --arch resnet50 --batch 512 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | line | cut -d' ' -f1-2)" >> $O
  echo "== $t resnet50 B=512 forward: $(cd $d && python tools/time_resnet.py 512 2>/dev/null | tail -1)" >> $O
done; done
cat $O
