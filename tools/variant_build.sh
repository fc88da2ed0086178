#!/bin/bash
# Development: build a VARIANT of the library beside the product -- tools/variants/<name>/libwenet_rx.so (git-ignored; travels to the GPU box) -- from the
# current tree with extra compiler flags for the named translation units (the others are linked from the product's objects).
#   tools/variant_build.sh <name> "<flags>" [unit ...]        e.g.  tools/variant_build.sh phi0b128 "-DWR_PHI0_FORM=2" ldpc_kernel
# Use: WENET_RX_LIB=tools/variants/<name>/libwenet_rx.so python bench.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; shift; shift
UNITS=${@:-ldpc_kernel}
D=$ROOT/tools/variants/$NAME
rm -rf $D; mkdir -p $D/wenet_amd/csrc $D/include
cp -r $ROOT/wenet_amd/csrc/*.h $ROOT/wenet_amd/csrc/*.hip $ROOT/wenet_amd/csrc/*.inc $ROOT/wenet_amd/csrc/*.cpp $ROOT/wenet_amd/csrc/*.o $ROOT/wenet_amd/csrc/Makefile $ROOT/wenet_amd/csrc/tables $D/wenet_amd/csrc/
cp $ROOT/include/*.h $D/include/
cp $ROOT/wenet_amd/codeid.py $D/wenet_amd/
touch $D/wenet_amd/csrc/*.o
for u in $UNITS; do rm -f $D/wenet_amd/csrc/$u.o; done
make -s -j8 -C $D/wenet_amd/csrc EXTRA="$FLAGS" ../libwenet_rx.so 2>&1 | grep -E "error|Error" || true
mv $D/wenet_amd/libwenet_rx.so $D/libwenet_rx.so
rm -rf $D/wenet_amd $D/include
ls -la $D/libwenet_rx.so
