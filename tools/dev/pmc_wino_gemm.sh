#!/bin/bash
# dev: SQ counters of rpn_wino_gemm_kernel at one layer shape (tools/dev/wino_kernels_time.py Cin Cout H W), per launch
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
R=$(pwd); export TMPDIR=/tmp
out=$R/gpurun_out/pmc_wino; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && REP=3 timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/s$i -o r -- python3 $R/tools/dev/wino_kernels_time.py "$@" > $out/s$i.log 2>&1) || echo "set $i failed"
done
python3 - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join("gpurun_out/pmc_wino", "**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "rpn_wino_gemm_kernel" in n:
            a = acc[(n.split("(")[0][:48], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()): print("%-50s %-28s %14.0f per launch over %d" % (k, c, v / n, n))
PY
find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete
