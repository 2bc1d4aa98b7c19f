# developer: tools/dev/c3_time.py under variant builds of csrc/conv_c3.hip (build_dbg/<name>): bash tools/dev/c3_variants.sh name...
for v in "$@"; do
  echo "== $v"
  if [ $v = base ]; then L=""; else L="build_dbg/$v/libfrcnn_hip.so"; fi
  FRCNN_HIP_LIB=$L timeout -k 10 120 python tools/dev/c3_time.py 2>&1 | tail -2 || exit 1
done
