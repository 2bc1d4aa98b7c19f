#!/bin/bash
# HBM traffic of the hot-path kernels from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), --kernel-trace only, no --stats / sys-trace.
# The profiled command is bench.py ITSELF (its own frames and shapes), few steps, no CPU baseline.
# Usage (on the GPU box, from the repo root):  bash tools/pmc_traffic.sh gpurun_out/pmc_r02_vgg vgg
#                                              bash tools/pmc_traffic.sh gpurun_out/pmc_r02_fpn fpn
#                                              bash tools/pmc_traffic.sh gpurun_out/pmc_r02_fpn_bf16 fpn_bf16   (= --config fpn --amp bf16)
set -e
OUT=${1:-gpurun_out/pmc}
CFG=${2:-vgg}
shift 2 || true
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BCFG=$CFG; EXTRA=""
if [ "$CFG" = "fpn_bf16" ]; then BCFG=fpn; EXTRA="--amp bf16"; fi
ARGS="--config $BCFG $EXTRA --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events --no-also $*"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$ROOT/$OUT/fetch" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$ROOT/$OUT/write" -- python3 "$ROOT/bench.py" $ARGS > "$ROOT/$OUT/write.log" 2>&1
cd "$ROOT"
python3 tools/pmc_traffic.py "$OUT" "$CFG" > "$OUT/traffic.json"
cat "$OUT/traffic.json"
