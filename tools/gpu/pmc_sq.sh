cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -f csv -d $GRAFT_REPO_ROOT/$O/p1 -o a -- python $GRAFT_REPO_ROOT/tools/one_fwd.py > $GRAFT_REPO_ROOT/$O/p1.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA -f csv -d $GRAFT_REPO_ROOT/$O/p2 -o b -- python $GRAFT_REPO_ROOT/tools/one_fwd.py > $GRAFT_REPO_ROOT/$O/p2.log 2>&1 )
python - <<'PY'
import csv, glob, re, collections
def rd(d):
    f=glob.glob(f"gpurun_out/pmc_sq/{d}/**/*counter_collection.csv", recursive=True)[0]
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m=re.search(r"mkws::(\w+)(<[^>]*>)?", r["Kernel_Name"])
        if m: per[m.group(1)+(m.group(2) or "").replace(" ","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per
a,b=rd("p1"),rd("p2")
for k in sorted(a):
    if not any(t in k for t in ("chain_kernel","pw_gemm_kernel<1,4","mid_kernel","front_kernel<3,2","stem_block","frontend")): continue
    c={n:sum(v[-max(1,len(v)//4):])/max(1,len(v)//4) for n,v in a[k].items()}; c.update({n:sum(v[-max(1,len(v)//4):])/max(1,len(v)//4) for n,v in b.get(k,{}).items()})
    wc=c.get("SQ_WAVE_CYCLES",1)
    print(f"{k[:44]:44s} wait_any {c.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst {c.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active {c.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} valu {c.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} lds {c.get('SQ_ACTIVE_INST_LDS',0)/wc:.2f} vmem {c.get('SQ_ACTIVE_INST_VMEM',0)/wc:.2f} | waitlds {c.get('SQ_WAIT_INST_LDS',0)/wc:.2f} bankconf/ldsinst {c.get('SQ_LDS_BANK_CONFLICT',0)/max(1,c.get('SQ_INSTS_LDS',1)):.2f} valu_insts {c.get('SQ_INSTS_VALU',0):.3g} lds_insts {c.get('SQ_INSTS_LDS',0):.3g} mfma_busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3g} wavecyc {wc:.3g}")
PY
