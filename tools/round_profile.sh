#!/bin/bash
# GPU box: the measurement bundle of a round.  usage: bash tools/round_profile.sh <tag>   (writes gpurun_out/round_<tag>/)
#   bench JSON (default run), bench with the RCCL leg forced on one rank, rocprofv3 kernel stats (two-stream default and
#   --overlap 0, ResNet-50), PMC passes (each in its own run, kernel-trace only): HBM FETCH/WRITE, MFMA-pipe busy, VALU / LDS /
#   wait counters, and the per-stage s_memtime tables of the fused blocks.
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round_$tag; mkdir -p $O
python $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
SYN_BENCH_FORCE_DIST=1 NCCL_DEBUG=INFO python $R/bench.py --no-cpu-baseline --no-extras > $O/bench_force_dist.json 2> $O/bench_force_dist.log; echo "force-dist rc=$?"
python $R/tools/stage_profile.py 1024 > $O/stage_profile_b1024.txt 2>&1; echo "stage rc=$?"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- $B > $O/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o k -- $B --overlap 0 > $O/stats1.log 2>&1; echo "stats(one stream) rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_r50 -o k -- $B --arch resnet50 --batch 512 > $O/stats_r50.log 2>&1; echo "stats(resnet50) rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_r50_1 -o k -- $B --arch resnet50 --batch 512 --overlap 0 > $O/stats_r50_1.log 2>&1; echo "stats(resnet50, one stream) rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b128 -o k -- $B --lmk-only --batch 128 --steps 20 --overlap 0 > $O/stats_b128.log 2>&1; echo "stats(b128 lmk-only) rc=$?"
P="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --overlap 0"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
            "valu:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES" \
            "lds:SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" \
            "wait:SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc_$name -o k -- $P > $O/pmc_$name.log 2>&1; echo "pmc $name rc=$?"
done
rm -f $O/*/k_kernel_trace.csv $O/*/*/k_kernel_trace.csv
python - <<PY
import csv, glob, collections, json
out = {}
for name in ('fetch', 'write', 'mfma', 'valu', 'lds', 'wait'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob('$O/pmc_%s/**/*counter_collection.csv' % name, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']; agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    out[name] = {k: {c: agg[k][c] / cnt[(k, c)] for c in agg[k]} for k in agg}
    out[name + '_launches'] = {k: max(cnt[(k, c)] for c in agg[k]) for k in agg}
json.dump(out, open('$O/pmc_raw.json', 'w'), indent=1)
for name in ('mfma', 'valu', 'lds', 'wait'):
    for k, v in out[name].items():
        print(name, k[:90], {c: round(x) for c, x in v.items()})
PY
