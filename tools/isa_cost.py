"""CPU-only (hipcc cross-compiles): instruction mix and an ISSUE-TIME estimate of every loop of a kernel.

    python tools/isa_cost.py fused_block_rm.hip "RmCfg<24, 144, 24, 30, 1, 1, true, 3, 2" [-DSYN_X=1 ...]

Why it exists: these kernels are bound by instruction issue (matrix and vector time of the waves of a SIMD ADD), GPU minutes are
scarce, and the cost table measured in round 4 (tools/ubench/valu_issue.hip; SIMD cycles per wave64 instruction with 2-4 waves per SIMD)
lets a variant be priced before it is run:
    v_fma_f32 / v_fmac / v_mul / v_add  2.7      v_pk_fma / v_pk_mul / v_pk_add  4.8 (two lanes' worth)
    every other VALU (dpp moves, med3, cvt_pk, fma_mix, perm, mov)  4.0
    v_mfma 32x32x16  32      v_mfma 16x16x32  16
    ds_* / buffer / global  4 issue slots (their latency is not priced)      s_*  1
The estimate is per wave and per trip of the loop; it ignores stalls, so it is a lower bound that ranks variants, not a prediction."""
import collections
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_scan import compile_to_asm  # noqa: E402

FAST = ('v_fma_f32', 'v_fmac_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_mac_f32')


def classify(op):
    if op.startswith('v_mfma'):
        return ('mfma32', 32.0) if '32x32' in op else ('mfma16', 16.0)
    if op.startswith('v_pk_'):
        return ('pk', 4.8)
    if op.startswith(FAST) and 'dpp' not in op:
        return ('fma', 2.7)
    if op.startswith('v_'):
        return ('valu', 4.0)
    if op.startswith('ds_'):
        return ('lds', 4.0)
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
        return ('vmem', 4.0)
    if op.startswith('s_'):
        return ('salu', 1.0)
    return ('other', 1.0)


def kernels(text):
    funcs = re.split(r'\n(?=_ZN3syn[^\n]*:\s*(?:;.*)?\n)', text)
    for fn in funcs:
        m = re.match(r'(_ZN3syn\S+):', fn)
        if m:
            name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
            yield re.sub(r'\(.*', '', name), fn.split('\n')


def loops_of(lines):
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r'(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    out = set()
    for i, l in enumerate(lines):
        mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            out.add((labels[mm.group(1)], i))
    return sorted(out)


def cost(body, detail=False):
    hist, cyc, ops = collections.Counter(), collections.Counter(), collections.Counter()
    for l in body:
        mm = re.match(r'\s+([a-z]\w+)', l)
        if not mm or l.lstrip().startswith(('.', ';')):
            continue
        op = mm.group(1)
        dpp = ' row_' in l or 'wave_sh' in l or 'quad_perm' in l
        k, c = classify(op + ('_dpp' if dpp else ''))
        hist[k] += 1
        cyc[k] += c
        ops[op + ('(dpp)' if dpp else '') + (' clamp' if ' clamp' in l else '')] += 1
    return hist, cyc, ops


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    defs = [a for a in sys.argv[2:] if a.startswith('-D')]
    want = [a for a in sys.argv[2:] if not a.startswith('-')]
    detail = '--ops' in sys.argv
    text = compile_to_asm(sys.argv[1], defs)
    for short, lines in kernels(text):
        if want and not any(w in short for w in want):
            continue
        print('==', short[:150])
        for a, b in loops_of(lines):
            hist, cyc, ops = cost(lines[a:b + 1])
            if hist['mfma32'] + hist['mfma16'] < 3:
                continue
            tot = sum(cyc.values())
            print('  loop @%d-%d: %4d instr, est %6.0f cyc | ' % (a, b, sum(hist.values()), tot) +
                  ' '.join('%s %d (%.0f)' % (k, hist[k], cyc[k]) for k in ('mfma32', 'mfma16', 'fma', 'pk', 'valu', 'lds', 'vmem', 'salu') if hist[k]))
            if detail:
                print('     ' + ', '.join('%s %d' % kv for kv in ops.most_common(28)))


if __name__ == '__main__':
    main()
