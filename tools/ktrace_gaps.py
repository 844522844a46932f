"""GPU box helper: from a rocprofv3 --kernel-trace directory, per run of consecutive forwards (split where stem kernels change type): wall per
forward = (end of last kernel - start of first) / forwards, sum of kernel durations per forward, and the idle gaps between kernels."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
for tag in ('StemRmCfg<4, 2, true>', 'StemRmCfg<4, 2, false>'):
    idx = [i for i, r in enumerate(rows) if tag in r[2]]
    if len(idx) < 6:
        continue
    idx = idx[3:]                                    # skip warm-up
    a, b = idx[0], idx[-1]
    n = len(idx) - 1
    seg = rows[a:b]
    wall = (rows[b][0] - rows[a][0]) / n
    busy = sum(e - s for s, e, _ in seg) / n
    gaps = sum(max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)) / n
    print(tag, 'forwards', n, 'wall/fwd %.1f us' % (wall / 1e3), 'kernel time/fwd %.1f us' % (busy / 1e3), 'gaps/fwd %.1f us' % (gaps / 1e3), 'kernels/fwd %.1f' % (len(seg) / n))
    import collections, re
    per = collections.defaultdict(list)
    for s_, e_, nm in seg:
        per[re.sub(r'\(.*', '', nm)[:70]].append((e_ - s_) / 1e3)
    for k_, v_ in per.items():
        print('     %-72s n=%3d avg %.1f us' % (k_, len(v_), sum(v_) / len(v_)))
