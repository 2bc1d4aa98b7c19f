#!/bin/bash
# One round's tracked evidence (run on the GPU box from the repo root):  bash tools/profile_round.sh r02
#   rocprofv3 --kernel-trace --stats of bench.py (VGG; FPN fp32; FPN bf16) -> per-step kernel summaries (tools/profile_summary.py)
#   PMC FETCH_SIZE / WRITE_SIZE passes over the same commands (tools/pmc_traffic.sh)
# Everything lands under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -e
TAG=${1:-rXX}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
prof() {   # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_$name" -- python3 "$R/bench.py" "$@" --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-events --no-also > "$R/gpurun_out/${TAG}_prof_$name.log" 2>&1
    python3 "$R/tools/profile_summary.py" $(ls "$R"/gpurun_out/${TAG}_prof_$name/*/*_kernel_trace.csv | head -1) --steps 10 > "$R/gpurun_out/${TAG}_${name}_kernel_summary.csv"
    cp $(ls "$R"/gpurun_out/${TAG}_prof_$name/*/*_kernel_stats.csv | head -1) "$R/gpurun_out/${TAG}_${name}_rocprofv3_kernel_stats.csv"
    grep '^{"metric"' "$R/gpurun_out/${TAG}_prof_$name.log" | tail -1 > "$R/gpurun_out/${TAG}_${name}_bench_under_rocprof.json"
    rm -rf "$R/gpurun_out/${TAG}_prof_$name"          # the raw trace (tens of MB) stays on the box: gpurun copies back at most 64 MiB
    echo "$name done"
}
prof vgg
prof fpn --config fpn
prof fpn_bf16 --config fpn --amp bf16
cd "$R"
bash tools/pmc_traffic.sh gpurun_out/${TAG}_pmc_vgg vgg > /dev/null
bash tools/pmc_traffic.sh gpurun_out/${TAG}_pmc_fpn fpn > /dev/null
bash tools/pmc_traffic.sh gpurun_out/${TAG}_pmc_fpn_bf16 fpn_bf16 > /dev/null
for c in vgg fpn fpn_bf16; do cp gpurun_out/${TAG}_pmc_$c/traffic.json gpurun_out/${TAG}_pmc_traffic_$c.json; rm -rf gpurun_out/${TAG}_pmc_$c; done
echo "pmc done"
# the plain bench lines of the same session (live HIP-event roofline + CPU baseline), reading this session's PMC traffic
mkdir -p profiles
for c in vgg fpn fpn_bf16; do cp gpurun_out/${TAG}_pmc_traffic_$c.json profiles/${TAG}_pmc_traffic_$c.json; done
python3 bench.py --no-also > gpurun_out/${TAG}_bench_vgg.json 2> gpurun_out/${TAG}_bench_vgg.err
cp bench_detail.json gpurun_out/${TAG}_bench_detail_vgg.json
python3 bench.py --config fpn > gpurun_out/${TAG}_bench_fpn.json 2> gpurun_out/${TAG}_bench_fpn.err
cp bench_detail.json gpurun_out/${TAG}_bench_detail_fpn.json
python3 bench.py --config fpn --amp bf16 > gpurun_out/${TAG}_bench_fpn_bf16.json 2> gpurun_out/${TAG}_bench_fpn_bf16.err
cp bench_detail.json gpurun_out/${TAG}_bench_detail_fpn_bf16.json
python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err     # the driver's command: headline + the two FPN figures in `also`
echo "bench done"
