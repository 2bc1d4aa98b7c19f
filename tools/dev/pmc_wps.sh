cd $GRAFT_REPO_ROOT; R=$(pwd); export TMPDIR=/tmp
for v in default wps1; do
  out=$R/gpurun_out/pmc_w_$v; rm -rf $out; mkdir -p $out
  if [ $v != default ]; then export FRCNN_HIP_LIB=$R/build_dbg/$v/libfrcnn_hip.so; else unset FRCNN_HIP_LIB; fi
  python3 tools/dev/wino_kernels_time.py 256 256 150 250 2>&1 | grep -E "fwd|bwd|wgrad"
  (cd /tmp && REP=3 timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $out -o r -- python3 $R/tools/dev/wino_kernels_time.py 256 256 150 250 > $out.log 2>&1)
  python3 - $out <<'PY'
import csv, glob, collections, os, sys
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "rpn_wino_gemm_kernel<false" in n:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
d = {k: v[0] / v[1] for k, v in acc.items()}
print(sys.argv[1].split("_")[-1], {k: int(v) for k, v in d.items()}, "MFMA busy per SIMD / kernel cycles = %.3f" % (d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (d["GRBM_GUI_ACTIVE"] / 8)), "wave lifetime share = %.3f" % (d["SQ_WAVE_CYCLES"] * 4 / (d["GRBM_GUI_ACTIVE"] / 8) / (1024 * (2 if "default" in sys.argv[1] else 1))))
PY
  rm -rf $out
done
