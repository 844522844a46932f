"""Copy a tools/round_profile.sh result directory into profiles/<name>/ and derive profiles/traffic_<name>.json.
usage: python tools/make_profiles.py gpurun_out/round_<tag> r2"""
import csv, glob, json, os, shutil, subprocess, sys, time
src, name = sys.argv[1], sys.argv[2]
dst = os.path.join('profiles', name)
os.makedirs(dst, exist_ok=True)


def cp(a, b):
    if os.path.isfile(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))


cp('stats/k_kernel_stats.csv', 'kernel_stats_b1024.csv')
cp('stats1/k_kernel_stats.csv', 'kernel_stats_b1024_one_stream.csv')
cp('stats_r50/k_kernel_stats.csv', 'resnet50_kernel_stats_b512.csv')
cp('stats_r50_1/k_kernel_stats.csv', 'resnet50_kernel_stats_b512_one_stream.csv')
cp('stats_b128/k_kernel_stats.csv', 'kernel_stats_b128_lmk_only.csv')
cp('stats/k_agent_info.csv', 'agent_info.csv')
cp('bench.json', 'bench_b1024.json')
cp('bench_force_dist.json', 'bench_force_dist_b1024.json')
cp('stage_profile_b1024.txt', 'stage_profile_b1024.txt')
cp('pmc_raw.json', 'pmc_raw_b1024.json')
if os.path.isfile(os.path.join(src, 'bench_force_dist.log')):      # RCCL banner + the broadcast line, without the hundreds of topology warnings
    keep = [l for l in open(os.path.join(src, 'bench_force_dist.log'), errors='replace')
            if ('RCCL version' in l or 'NCCL INFO' in l and ('Bootstrap' in l or 'Using network' in l or 'comm 0x' in l or 'Init COMPLETE' in l or 'Broadcast' in l)
                or 'RCCL broadcast' in l or 'Librccl' in l)]
    open(os.path.join(dst, 'bench_force_dist_b1024.log'), 'w').writelines(keep[:60])

raw = json.load(open(os.path.join(src, 'pmc_raw.json')))
fetch, write = raw['fetch'], raw['write']
launches = raw['fetch_launches']
n_fwd = min(c for k, c in launches.items() if 'stem' in k and 'kernel' in k)          # one stem launch per forward
fam = [k for k in fetch if 'fused_block' in k or 'fused_chain' in k or 'fused_pair' in k]      # (chains: several blocks per launch)
rd = sum(fetch[k]['FETCH_SIZE'] * 2 * 1024 * launches[k] / n_fwd for k in fam)
n_fwd_w = min(c for k, c in raw['write_launches'].items() if 'stem' in k and 'kernel' in k)      # (the passes run for a time, not a count: each has its own number of forwards)
wr = sum(write[k]['WRITE_SIZE'] * 1024 * raw['write_launches'][k] / n_fwd_w for k in fam)
n_launch = sum(launches[k] for k in fam) / n_fwd


def ratio(a, b):
    return round(a / b, 4) if b else None


per_kernel = {}
busy_sum = act_sum = 0.0
for k in fetch:
    m, v, l, w = raw['mfma'].get(k, {}), raw['valu'].get(k, {}), raw['lds'].get(k, {}), raw['wait'].get(k, {})
    simd_cycles = m.get('GRBM_GUI_ACTIVE', 0) * 128          # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs; 1024 SIMDs
    per_kernel[k] = dict(
        read_bytes=fetch[k]['FETCH_SIZE'] * 2 * 1024, write_bytes=write.get(k, {}).get('WRITE_SIZE', 0) * 1024,
        mfma_pipe_busy=ratio(m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), simd_cycles),
        valu_insts=v.get('SQ_INSTS_VALU'), mfma_mops_f16=v.get('SQ_INSTS_VALU_MFMA_MOPS_F16'),
        valu_active_of_wave_cycles=ratio(v.get('SQ_ACTIVE_INST_VALU', 0), v.get('SQ_WAVE_CYCLES', 0)),
        lds_insts=l.get('SQ_INSTS_LDS'), lds_bank_conflict_of_active=ratio(l.get('SQ_LDS_BANK_CONFLICT', 0), l.get('SQ_LDS_IDX_ACTIVE', 0)),
        lds_array_busy=ratio(l.get('SQ_LDS_IDX_ACTIVE', 0), m.get('GRBM_GUI_ACTIVE', 0) * 32),      # 256 LDS arrays, counter summed over XCDs
        wait_any_of_wave_cycles=ratio(w.get('SQ_WAIT_ANY', 0), v.get('SQ_WAVE_CYCLES', 0)),
        wait_inst_any_of_wave_cycles=ratio(w.get('SQ_WAIT_INST_ANY', 0), v.get('SQ_WAVE_CYCLES', 0)),
        wait_inst_lds_of_wave_cycles=ratio(w.get('SQ_WAIT_INST_LDS', 0), v.get('SQ_WAVE_CYCLES', 0)))
    if k in fam:
        busy_sum += m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) * raw['mfma_launches'][k]
        act_sum += simd_cycles * raw['mfma_launches'][k]
def git(*a):
    try:
        return subprocess.run(['git'] + list(a), capture_output=True, text=True).stdout.strip()
    except OSError:
        return ''


# provenance (bench.py prints it as roofline.counters_source): the commit the bundle was taken at -- recorded by round_profile.sh's
# caller in <src>/commit.txt, else HEAD now, marked dirty when the kernels differ from it -- and the box
commit = open(os.path.join(src, 'commit.txt')).read().strip() if os.path.isfile(os.path.join(src, 'commit.txt')) else git('rev-parse', '--short=12', 'HEAD')
if git('status', '--porcelain', '--', 'synergynet_amd/csrc', 'bench.py'):
    commit += '+dirty'
box = None
if os.path.isfile(os.path.join(dst, 'agent_info.csv')):
    rows = [r for r in csv.DictReader(open(os.path.join(dst, 'agent_info.csv'))) if r.get('Agent_Type') == 'GPU']
    if rows:
        box = '%s (%s CUs, %s MHz max)' % (rows[0].get('Name'), rows[0].get('Cu_Count'), rows[0].get('Max_Engine_Clk_Fcompute'))
out = dict(commit=commit, box=box, collected=time.strftime('%Y-%m-%d', time.gmtime(os.path.getmtime(os.path.join(src, 'pmc_raw.json')))),
           source='rocprofv3 --kernel-trace --pmc <one counter group per run> of bench.py B=1024 --overlap 0 (tools/round_profile.sh): '
                  'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads), KB -> bytes; '
                  'mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 128): the counter sums the per-SIMD busy cycles '
                  'of all 1024 SIMDs and GRBM_GUI_ACTIVE arrives summed over the 8 XCDs',
           fused_block_launches_per_forward=n_launch, fused_block_bytes_per_forward=dict(read=rd, write=wr),
           fused_block_bytes_per_launch=round((rd + wr) / n_launch), fused_block_mfma_pipe_busy=ratio(busy_sum, act_sum),
           per_kernel=per_kernel)
json.dump(out, open(os.path.join('profiles', f'traffic_{name}.json'), 'w'), indent=1)
print('launches/forward', n_launch, 'read MB', rd / 1e6, 'write MB', wr / 1e6, 'family MFMA pipe busy', out['fused_block_mfma_pipe_busy'])
d = json.load(open(os.path.join(dst, 'bench_b1024.json')))
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], {k: v for k, v in d.get('cpu_baseline', {}).items() if k in ('value', 'cores', 'kind')})

# ResNet-50 counters (tools/pmc_resnet.sh prints per-kernel averages to gpurun_out/pmc_resnet.txt): a compact table next to the bundle
pmc_txt = os.path.join(os.path.dirname(src.rstrip('/')), 'pmc_resnet.txt')
if os.path.isfile(pmc_txt):
    import ast
    import re
    rows = {}
    for line in open(pmc_txt):
        m = re.match(r'(mfma|wait|valu) (.*?) (\{.*\}) launches (\d+)', line.strip())
        if m:
            k = re.sub(r'\(.*', '', m.group(2).replace('void syn::', '').replace('syn::', ''))
            rows.setdefault(k, {}).update(ast.literal_eval(m.group(3)))
    tab = ['# ResNet-50 B = 512, per-launch averages of the PMC passes of tools/pmc_resnet.sh (one pass per counter group, kernel-trace only).',
           '# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES)   (matrix pipe busy share of the SIMD cycles of busy CUs)',
           '# wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES',
           '%-36s %9s %9s %12s %12s' % ('kernel', 'mfma_busy', 'wait', 'insts_valu', 'lds_idx_act')]
    for k, v in rows.items():
        if 'SQ_BUSY_CU_CYCLES' in v and 'SQ_WAVE_CYCLES' in v and 'SQ_INSTS_VALU' in v:
            tab.append('%-36s %9.3f %9.3f %12d %12d' % (k[:36], v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * v['SQ_BUSY_CU_CYCLES']),
                                                       v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'], v['SQ_INSTS_VALU'], v['SQ_LDS_IDX_ACTIVE']))
    open(os.path.join(dst, 'resnet50_pmc_b512.txt'), 'w').write('\n'.join(tab) + '\n')
    print('resnet50_pmc_b512.txt:', len(tab) - 4, 'kernels')
