# PMC counters of the many-row attention on tools/attn_mid.py's shape: bash tools/pmc_attn_mid.sh <outdir-name> [attn_mid args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
shift
mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc$i -o p -f csv -- python $R/tools/attn_mid.py "$@" > $OUT/pmc$i.log 2>&1
done
OUTDIR=$OUT python3 - <<'PY'
import csv,glob,os,collections
out=os.environ['OUTDIR']
for f in sorted(glob.glob(out+'/pmc*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(lambda: [0,0.0]))
    for r in csv.DictReader(open(f)):
        if 'attn_' not in r['Kernel_Name'] or 'combine' in r['Kernel_Name']: continue
        key=(r['Kernel_Name'][:70], r.get('Grid_Size','?'))
        a=agg[key][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    for key,d in agg.items():
        print(key)
        for k,(n,v) in d.items(): print(f"  {k:32s} n={n} avg={v/n:.4g}")
for f in sorted(glob.glob(out+'/pmc1/**/*kernel_trace.csv', recursive=True)):
    d=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        if 'attn_' in r['Kernel_Name']:
            k=(r['Kernel_Name'][:70], r.get('Grid_Size','?')); d[k][0]+=1; d[k][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    for k,(n,t) in d.items(): print("duration", k, n, f"{t/n:.1f} us avg")
PY
