"""GPU box helper: print (calls, average us, short name) of every kernel of a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, re, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        print('%5s %9.1f  %s' % (r['Calls'], float(r['AverageNs']) / 1e3, re.sub(r'\(.*', '', r['Name'])[:110]))
