#!/bin/bash
# Development: build the instrumented library (make PROF=1: cycle stamps in the kernels, WENET_RX_PROFILE=...) beside the product, into
# gpurun_out-independent tools/prof_build/ (git-ignored; travels to the GPU box).  Use: WENET_RX_LIB=tools/prof_build/libwenet_rx.so python tools/gpu_oct_prof.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/tools/prof_build
rm -rf $D; mkdir -p $D/wenet_amd/csrc $D/include
cp -r $ROOT/wenet_amd/csrc/*.h $ROOT/wenet_amd/csrc/*.hip $ROOT/wenet_amd/csrc/*.inc $ROOT/wenet_amd/csrc/*.cpp $ROOT/wenet_amd/csrc/Makefile $ROOT/wenet_amd/csrc/tables $D/wenet_amd/csrc/
cp $ROOT/include/*.h $D/include/; cp $ROOT/wenet_amd/codeid.py $D/wenet_amd/
make -s -j8 -C $D/wenet_amd/csrc PROF=1 EXTRA="$PROF_EXTRA" ../libwenet_rx.so
mv $D/wenet_amd/libwenet_rx.so $D/libwenet_rx.so
rm -rf $D/wenet_amd $D/include
ls -la $D
