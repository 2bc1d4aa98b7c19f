#!/bin/bash
# HBM traffic of the hot-path kernels from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), --kernel-trace only, no --stats / sys-trace.
# Usage (on the GPU box, from the repo root):  bash tools/pmc_traffic.sh gpurun_out/pmc_r01
set -e
OUT=${1:-gpurun_out/pmc}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$ROOT/$OUT/fetch" -- python3 "$ROOT/tools/hotpath_bench.py" --iters 20 --regime trained > "$ROOT/$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$ROOT/$OUT/write" -- python3 "$ROOT/tools/hotpath_bench.py" --iters 20 --regime trained > "$ROOT/$OUT/write.log" 2>&1
cd "$ROOT"
python3 tools/pmc_traffic.py "$OUT" > "$OUT/traffic.json"
cat "$OUT/traffic.json"
