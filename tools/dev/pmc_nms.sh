#!/bin/bash
# dev: PMC counters of nms_kernel (tiles-only or full build via FRCNN_HIP_LIB)
cd /root/repo; export TMPDIR=/tmp
out=gpurun_out/pmc_nms; mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  BENCH_BOXES=build_dbg/bench_boxes_v.npy timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/s$i -o r -- python tools/dev/nms_time.py > $out/s$i.log 2>&1 || echo "set $i failed"
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_nms/s*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("nms_kernel") or "nms_kernel" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items(): print(k, "%.0f" % (v / n), "per launch over", n)
PY
find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete
