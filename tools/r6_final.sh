#!/bin/bash
# round 6, final GPU call: the CPU suite and the whole GPU suite (no -x) FIRST -- a bundle is only cut on a green tree (VERDICT r5 #2c) -- then the
# smoke entry and the measurement bundle (tools/round_profile.sh r6 -> gpurun_out/round_r6, copied into profiles/r6 by tools/make_profiles.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m "not gpu" -q --tb=short -p no:cacheprovider 2>&1 | tail -4 ) > $O/pytest_cpu_final.txt
cat $O/pytest_cpu_final.txt
grep -q " passed" $O/pytest_cpu_final.txt && ! grep -q "failed\|error" $O/pytest_cpu_final.txt || { echo "CPU suite not green: no bundle"; exit 1; }
( timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "input bound" | tail -6 ) > $O/pytest_gpu_final.txt
cat $O/pytest_gpu_final.txt
grep -q " passed" $O/pytest_gpu_final.txt && ! grep -q "failed\|error" $O/pytest_gpu_final.txt || { echo "GPU suite not green: no bundle"; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
timeout 1800 bash tools/round_profile.sh r6 2>&1 | grep -E "rc=|^\{" | cut -c1-200
bash tools/pmc_resnet.sh > $O/round_r6/resnet50_pmc_b512.txt 2>&1; tail -12 $O/round_r6/resnet50_pmc_b512.txt
