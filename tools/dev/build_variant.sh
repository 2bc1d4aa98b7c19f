#!/bin/bash
# tools/dev/build_variant.sh FILE NAME -DFLAG...  ->  build_dbg/NAME/libfrcnn_hip.so  (csrc/FILE.hip rebuilt with the flags, the other objects reused;
# load it with FRCNN_HIP_LIB=build_dbg/NAME/libfrcnn_hip.so; build_dbg/ is git-ignored but travels to the GPU box)
set -e
cd /root/repo/faster_rcnn_pytorch_amd/csrc
f=$1; name=$2; shift; shift
mkdir -p /root/repo/build_dbg/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math "$@" -c $f.hip -o /root/repo/build_dbg/$name/$f.o
objs=$(ls ../lib/obj/*.o | grep -v "/$f.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build_dbg/$name/libfrcnn_hip.so $objs /root/repo/build_dbg/$name/$f.o
