cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bp/$n -- python $R/tools/prof_bp.py > /dev/null 2>&1
done
cd $R && python tools/summarize_prof.py bpx gpurun_out/pmc_bp 2>&1 | tail -5
find gpurun_out/pmc_bp -name "*kernel_trace.csv" -delete
python - <<'PY'
import json
d=json.load(open("profiles/bpx_pmc.json"))
for k,v in d.get("bp_beam",{}).items(): print(k, v)
PY
