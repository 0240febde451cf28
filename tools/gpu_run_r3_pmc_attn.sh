cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in 1 0; do
(cd /tmp && AITK_ATTN_FWD64=$v AITK_PMC_F8=0 AITK_PMC_M=4608 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INST_CYCLES_VMEM -d "$GRAFT_REPO_ROOT/gpurun_out/r3_pmc_attn/v$v" -o v$v --output-format csv -- python "$GRAFT_REPO_ROOT/tools/gpu_pmc_target.py" > /dev/null 2>&1)
echo "variant fwd64=$v rc=$?"
python - <<PY
import csv,glob
acc={}
for f in glob.glob("gpurun_out/r3_pmc_attn/v$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_fwd" in r["Kernel_Name"]:
            a=acc.setdefault(r["Counter_Name"],[0.0,0]); a[0]+=float(r["Counter_Value"]); a[1]+=1
m={k:v[0]/v[1] for k,v in acc.items()}
print({k:round(v) for k,v in m.items()})
if "SQ_WAVE_CYCLES" in m:
    wc=m["SQ_WAVE_CYCLES"]; print("wait_any %.3f wait_inst %.3f active %.3f mfma_busy/(32*busy) %.3f" % (m["SQ_WAIT_ANY"]/wc, m["SQ_WAIT_INST_ANY"]/wc, m["SQ_ACTIVE_INST_ANY"]/wc, m["SQ_VALU_MFMA_BUSY_CYCLES"]/(32*m["SQ_BUSY_CYCLES"])))
PY
done
rm -rf gpurun_out/r3_pmc_attn
