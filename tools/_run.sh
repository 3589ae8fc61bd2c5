cd /tmp && export TMPDIR=/tmp
out=/tmp/pmc_asm
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out -o asm -- python3 $GRAFT_REPO_ROOT/tools/ns_assemble_time.py C3 > $out/log.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES --kernel-trace --output-format csv -d ${out}2 -o asm -- python3 $GRAFT_REPO_ROOT/tools/ns_assemble_time.py C3 > ${out}2/log.txt 2>&1
tail -1 ${out}2/log.txt
python3 - <<'PY'
import csv, glob, collections
for d in ('/tmp/pmc_asm','/tmp/pmc_asm2'):
    fs=glob.glob(d+'/**/*counter_collection.csv', recursive=True)
    if not fs: print('no counters in', d); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name']
        if 's6_assemble2' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
    for k in acc:
        for c,v in acc[k].items(): print("   %-24s %14.0f per dispatch (%d)"%(c, v/n[(k,c)], n[(k,c)]))
PY
