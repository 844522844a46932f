"""GPU box: one line per run -- step time, backbone time and the per-launch microseconds of bench.py's roofline block.
usage: python tools/perlaunch.py [bench args]      (env knobs such as SYNERGY_HIP_EARLY_RM pass through)"""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(R, 'bench.py'), '--no-cpu-baseline', '--no-extras'] + sys.argv[1:], capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
    r = d['roofline']
    print(f"{d['ms_per_step']:.4f} ms/step  backbone {r['backbone']['ms']:.4f}  frac {r['frac']:.3f} | " +
          ' '.join(f"{p['feature']}:{p['ms'] * 1e3:.0f}" for p in r['per_launch']))
except Exception as e:
    print('bench failed:', e, out.stderr[-800:])
