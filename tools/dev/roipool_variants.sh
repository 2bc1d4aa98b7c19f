#!/bin/bash
# tools/dev/roipool_variants.sh name...  ->  roipool_time.py under the default build and each build_dbg/<name> variant
cd /root/repo
python tools/dev/roipool_time.py 2>&1 | grep -v amdgpu.ids
FRCNN_ROI_BWD_SHARED=1 python tools/dev/roipool_time.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do FRCNN_HIP_LIB=build_dbg/$v/libfrcnn_hip.so python tools/dev/roipool_time.py 2>&1 | grep -v amdgpu.ids; done
