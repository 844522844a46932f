"""Pretty-print a bench.py JSON line from stdin: headline + per-launch table."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print(f"{d['value']:.0f} faces/s  {d['ms_per_step']:.3f} ms/step  family {r['achieved']} TF ({r['frac']*100:.1f}%)  backbone {r.get('backbone')}")
for x in r.get('per_launch', []):
    print(f"  f{x['feature']:<3d} {x['ms']*1e3:7.1f} us  {x['tflops']:6.1f} TF")
