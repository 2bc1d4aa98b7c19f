#!/bin/bash
# SQ counters of the fp32 conv kernels (separate from any trace), V and F shapes -> gpurun_out/conv_pmc_{V,F}.csv
cd /tmp && export TMPDIR=/tmp
for s in V F; do
  rm -rf /tmp/pmc_$s
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
     --output-format csv -d /tmp/pmc_$s -- python3 $GRAFT_REPO_ROOT/tools/dev/conv_f32_run.py $s 3 > /tmp/pmc_$s.log 2>&1
  f=$(find /tmp/pmc_$s -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$s" <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/conv_pmc_$s.txt 2>&1
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "rpn_conv" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%d mean %.4g" % (c, len(v), sum(v) / len(v)))
PY
done
