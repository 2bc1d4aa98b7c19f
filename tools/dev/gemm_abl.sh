# developer: the stage's kernels under ablation builds (build_dbg/<name>): bash tools/dev/gemm_abl.sh name... -- Cin Cout H W ...
names=(); while [ "$1" != "--" ]; do names+=("$1"); shift; done; shift
for v in "${names[@]}"; do
  echo "== $v"
  if [ $v = base ]; then L=""; else L="build_dbg/$v/libfrcnn_hip.so"; fi
  FRCNN_HIP_LIB=$L timeout -k 10 120 python tools/dev/wino_kernels_time.py "$@" 2>&1 | grep -E "fwd|->" || exit 1
done
