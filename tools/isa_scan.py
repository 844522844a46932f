"""CPU-only (hipcc cross-compiles): the load / wait / matrix-instruction SEQUENCE of every loop of a kernel, one line per loop:

    python tools/isa_scan.py resnet_kernels.hip "conv_lt_kernel<4>" "conv_h2s_kernel<2, 4, 2>"
    python tools/isa_scan.py fused_block_lb.hip fused_chain_lb_kernel

  L = global / buffer load, S = store, | = s_barrier, wN = s_waitcnt vmcnt(N), Mk = k matrix instructions in a row.

What to look for: `L w0` right behind each other inside a loop -- something consumes a loaded register at once (a
select that implements zero padding, a BN shift loaded at the top of its tile) and, vector memory retiring in order, drains every load in
flight behind it; `w0` at the top of a loop whose steady state would allow `w8` -- the loop is entered with other loads in flight than
the back edge leaves (peel the first iteration), or its body has a conditional fetch (the wait-count bookkeeping gives up at joins)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def loop_sequences(text, want):
    """{demangled kernel name: [(first line, last line, n_mfma, [tokens]) per loop with >= 4 matrix instructions]}"""
    out = {}
    funcs = re.split(r'\n(?=_ZN3syn[^\n]*:\s*(?:;.*)?\n)', text)
    for fn in funcs:
        m = re.match(r'(_ZN3syn\S+):', fn)
        if not m:
            continue
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        short = re.sub(r'\(.*', '', name)
        if want and not any(w in short for w in want):
            continue
        lines = fn.split('\n')
        labels = {}
        for i, l in enumerate(lines):
            mm = re.match(r'(\.LBB\d+_\d+):', l)
            if mm:
                labels[mm.group(1)] = i
        loops = set()
        for i, l in enumerate(lines):
            mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                loops.add((labels[mm.group(1)], i))
        res = []
        for a, b in sorted(loops):
            body = lines[a:b + 1]
            nm = sum('v_mfma' in l for l in body)
            if nm < 4:
                continue
            seq = []
            for l in body:
                if re.search(r'\b(global_load|buffer_load|scratch_load)', l):
                    seq.append('L')
                elif re.search(r'\b(global_store|buffer_store|scratch_store)', l):
                    seq.append('S')
                elif 's_barrier' in l:
                    seq.append('|')
                else:
                    mm = re.search(r's_waitcnt.*vmcnt\((\d+)\)', l)
                    if mm:
                        seq.append('w' + mm.group(1))
                    elif 'v_mfma' in l:
                        if seq and seq[-1].startswith('M'):
                            seq[-1] = 'M%d' % (int(seq[-1][1:]) + 1)
                        else:
                            seq.append('M1')
            res.append((a, b, nm, seq))
        out[short] = res
    return out


def adjacency(text, want):
    """Per kernel: how the matrix instructions sit among the vector instructions -- MFMAs followed directly by another MFMA, the MFMA bursts
    (runs of back-to-back MFMAs: count, mean, max), the histogram of VALU instructions between two consecutive MFMAs (0 = back to back,
    1-3 = in a shadow a 16-cycle MFMA can cover, >= 4 = a vector phase), and the VALU total.  (the evidence next to a timing.)"""
    out = {}
    for fn in re.split(r'\n(?=_ZN3syn[^\n]*:\s*(?:;.*)?\n)', text):
        m = re.match(r'(_ZN3syn\S+):', fn)
        if not m:
            continue
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        short = re.sub(r'\(.*', '', name)
        if want and not any(w in short for w in want):
            continue
        kinds = []
        for l in fn.split('\n'):
            t = l.strip().split(' ')[0] if l.strip() else ''
            if t.startswith('v_mfma'):
                kinds.append('M')
            elif t.startswith('v_') and not t.startswith('v_accvgpr'):
                kinds.append('V')
            elif t.startswith(('ds_', 'buffer_', 'global_', 's_barrier', 's_waitcnt', 's_cbranch', 's_branch')):
                kinds.append('o')
        gaps, bursts, run, since = [], [], 0, None
        for k in kinds:
            if k == 'M':
                if since is not None:
                    gaps.append(since)
                if since == 0 or since is None:
                    run += 1
                else:
                    if run:
                        bursts.append(run)
                    run = 1
                since = 0
            elif k == 'V' and since is not None:
                since += 1
        if run:
            bursts.append(run)
        n_m = kinds.count('M')
        hist = {'0': sum(g == 0 for g in gaps), '1-3': sum(1 <= g <= 3 for g in gaps), '4-8': sum(4 <= g <= 8 for g in gaps), '>8': sum(g > 8 for g in gaps)}
        out[short] = dict(mfma=n_m, valu=kinds.count('V'), back_to_back=hist['0'], gaps=hist, bursts=len(bursts),
                          burst_mean=round(sum(bursts) / max(1, len(bursts)), 1), burst_max=max(bursts) if bursts else 0)
    return out


def scan(text, want):
    for short, loops in loop_sequences(text, want).items():
        print('==', short[:140])
        for a, b, nm, seq in loops:
            print('  loop @%d-%d, %d mfma: %s' % (a, b, nm, ' '.join(seq)))


def compile_to_asm(src, defines=()):
    path = src if os.path.isfile(src) else os.path.join(ROOT, 'synergynet_amd', 'csrc', src)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'k.s')
        r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-w',
                            '-I' + os.path.join(ROOT, 'synergynet_amd', 'csrc'), '-I' + os.path.join(ROOT, 'include'), '-o', out, path] + list(defines),
                           capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-3000:])
        return open(out).read()


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    args = [a for a in sys.argv[2:] if a != '--adjacency']
    text = compile_to_asm(sys.argv[1], [a for a in args if a.startswith('-D')])
    if '--adjacency' in sys.argv:
        for k, v in adjacency(text, [a for a in args if not a.startswith('-D')]).items():
            print(k[:100], v)
        return
    scan(text, [a for a in args if not a.startswith('-D')])


if __name__ == '__main__':
    main()
