# PMC counters of gemm_dense_kernel at one shape: bash tools/pmc_dense.sh [M N K epi two reps]   (default: 6000 12288 4096 0 1 6)
cd /tmp && export TMPDIR=/tmp
ARGS="${@:-6000 12288 4096 0 1 6}"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r2d
mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc$i -o p -f csv -- python $R/tools/dense_one.py $ARGS > $OUT/pmc$i.log 2>&1
done
python3 - <<'PY'
import csv,glob,os,collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r2d'
for f in sorted(glob.glob(out+'/pmc*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        if 'gemm_dense' not in r['Kernel_Name']: continue
        a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    for k,(n,v) in agg.items(): print(f"  {k:32s} n={n} avg={v/n:.4g}")
for f in sorted(glob.glob(out+'/pmc1/**/*kernel_trace.csv', recursive=True)):
    t=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'gemm_dense' in r['Kernel_Name']]
    print("duration avg us", sum(t)/len(t), "min", min(t))
PY
