"""CPU-only (hipcc cross-compiles): per-kernel register / spill / scratch report of every HIP source, and the list of
`s_waitcnt vmcnt(0)` that sit inside loops of the hot kernels.  Why it exists: vector memory retires in order on gfx950, so
a scratch reload -- or any load the compiler cannot count precisely -- placed after a burst of stores waits for the stores'
acknowledgements; round 1 lost a third of the reconstruction kernel to exactly that.

  python tools/isa_lint.py            # table; exit code 1 if a product kernel spills or touches scratch
"""
import os
import re
import subprocess
import sys
import shutil
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from synergynet_amd.build import CSRC, SOURCES  # noqa: E402


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for src in SOURCES:
            if src == 'synergy_abi.hip':
                continue
            out = os.path.join(td, src + '.s')
            r = subprocess.run([shutil.which('hipcc') or '/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-w',
                                '-I' + os.path.join(ROOT, 'include'), '-o', out, os.path.join(CSRC, src)],
                               capture_output=True, text=True)
            if r.returncode:
                print(src, 'FAILED TO COMPILE\n', r.stderr[-2000:])
                bad += 1
                continue
            text = open(out).read()
            print(f'== {src}')
            for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n'
                                 r'\s+\.vgpr_spill_count:\s+(\d+)', text):
                name, scratch, vgpr, spill = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))
                dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
                short = re.sub(r'\(.*', '', dem)[:110]
                flag = ''
                if scratch or spill:
                    # PROF = second template argument of the fused-block kernels, third of the reconstruction kernel
                    is_prof = bool(re.search(r'(fused_block_\w+<.*>, true(, \d+)?(, (true|false))?>$)|(recon_f16_kernel<\d+, (true|false), true>$)', short))
                    flag = '  <-- spills (profiling instantiation)' if is_prof else '  <-- SPILLS / SCRATCH'
                    bad += 0 if is_prof else 1
                print(f'   {vgpr:4d} vgprs {spill:4d} spilled {scratch:5d} B scratch  {short}{flag}')
    print('product kernels with spills or scratch:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
