"""Copy a tools_round.sh result directory into profiles/<name>/ and derive profiles/traffic_r1.json.
usage: python tools/make_profiles.py gpurun_out/round_<tag> r1"""
import csv, glob, json, os, shutil, sys
src, name = sys.argv[1], sys.argv[2]
dst = os.path.join('profiles', name)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, 'stats', 'k_kernel_stats.csv'), os.path.join(dst, 'kernel_stats_b1024.csv'))
shutil.copy(os.path.join(src, 'stats', 'k_agent_info.csv'), os.path.join(dst, 'agent_info.csv'))
shutil.copy(os.path.join(src, 'bench.json'), os.path.join(dst, 'bench_b1024.json'))
raw = {}
counts = {}
for kind in ('fetch', 'write'):
    agg, cnt = {}, {}
    for f in glob.glob(os.path.join(src, kind, '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            agg[k] = agg.get(k, 0.0) + float(r['Counter_Value'])
            cnt[k] = cnt.get(k, 0) + 1
    raw[kind] = {k: agg[k] / cnt[k] for k in agg}
    counts[kind] = cnt
json.dump(raw, open(os.path.join(dst, 'hbm_pmc_raw_b1024.json'), 'w'), indent=1)
n_fwd = min(c for k, c in counts['fetch'].items() if 'stem_block1' in k)          # one stem launch per forward
fam = [k for k in raw['fetch'] if 'fused_block' in k]
rd = sum(raw['fetch'][k] * 2 * 1024 * counts['fetch'][k] / n_fwd for k in fam)
wr = sum(raw['write'][k] * 1024 * counts['write'][k] / n_fwd for k in fam)
n_launch = sum(counts['fetch'][k] for k in fam) / n_fwd
out = dict(source='rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) of bench.py B=1024; '
                  'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); KB -> bytes',
           fused_block_launches_per_forward=n_launch, fused_block_bytes_per_forward=dict(read=rd, write=wr),
           fused_block_bytes_per_launch=round((rd + wr) / n_launch),
           all_kernels_bytes_per_launch={k: dict(read=raw['fetch'][k] * 2 * 1024, write=raw['write'].get(k, 0) * 1024) for k in raw['fetch']})
json.dump(out, open('profiles/traffic_r1.json', 'w'), indent=1)
print('launches/forward', n_launch, 'read MB', rd / 1e6, 'write MB', wr / 1e6)
d = json.load(open(os.path.join(dst, 'bench_b1024.json')))
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d.get('cpu_baseline'))
