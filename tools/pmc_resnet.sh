R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_r50; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
P="python $R/bench.py --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras --overlap 0 --arch resnet50 --batch 512"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o k -- $P > $O/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/wait -o k -- $P > $O/wait.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/valu -o k -- $P > $O/valu.log 2>&1
python - <<PY
import csv, glob, collections
for name in ('mfma','wait','valu'):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob('$O/%s/**/*counter_collection.csv'%name, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:60]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
    for k,v in agg.items():
        if 'conv' in k or 'stem' in k or 'pool' in k: print(name,k,{c:round(x/cnt[(k,c)]) for c,x in v.items()}, 'launches', max(cnt[(k,c)] for c in v))
PY
rm -rf $O/*/k_kernel_trace.csv
