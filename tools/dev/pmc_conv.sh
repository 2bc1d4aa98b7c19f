#!/bin/bash
# dev: PMC counters of the bf16 conv kernels inside the FPN bf16 training step
cd /root/repo; export TMPDIR=/tmp
out=gpurun_out/pmc_conv; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/s$i -o r -- python bench.py --config fpn --amp bf16 --steps 4 --warmup 2 --no-also --no-cpu-baseline --no-kernel-events > $out/s$i.log 2>&1 || echo "set $i failed"
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_conv/s*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            for k in ("rpn_conv3x3_head_kernel", "rpn_conv3x3_wgrad_kernel", "rpn_conv3x3_bwd_data_kernel"):
                if k in n:
                    a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for (k, c), (v, n) in sorted(acc.items()): print(k, c, "%.0f" % (v / n), "per launch over", n)
PY
find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete
