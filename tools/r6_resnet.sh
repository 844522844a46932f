#!/bin/bash
# GPU box: ResNet-50 parity tests + backbone time at B = 512 (round 6: pair activation format)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_resnet; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests -m gpu -q --tb=line -k "resnet" 2>&1 | tail -15 ) > $O/pytest.txt; cat $O/pytest.txt
for i in 1 2 3; do python tools/time_resnet.py 512; done 2>&1 | grep backbone | tee $O/time.txt
