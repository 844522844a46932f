"""CPU-only guard (hipcc cross-compiles gfx950): no kernel of the library may spill registers or use scratch memory.

Vector memory retires in order on this ISA, so a scratch reload that lands after a burst of stores waits for the stores'
acknowledgements -- one spilled address per store instruction cost the reconstruction kernel a third of its time in round 1
(DESIGN.md 5.2).  tools/isa_lint.py prints the per-kernel table; this test fails on any spill / scratch use outside the
profiling instantiations."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.isfile('/opt/rocm/bin/hipcc'), reason='hipcc not available')
def test_no_kernel_spills_or_uses_scratch(capsys):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import isa_lint
    rc = isa_lint.main()
    out = capsys.readouterr().out
    assert rc == 0, out[-4000:]
    assert 'recon_f16_kernel<4, true, false, false>' in out and 'recon_f16_kernel<4, false, false, true>' in out and 'fused_block_early_kernel' in out      # the table really covers the hot kernels
