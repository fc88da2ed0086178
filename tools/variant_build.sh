#!/bin/bash
# Development: build a variant of the library with extra compiler flags beside the product, into tools/variants/<name>.so (git-ignored; travels to the GPU box).
# usage: tools/variant_build.sh <name> "<flags>"      then: WENET_RX_LIB=tools/variants/<name>.so python ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
D=$(mktemp -d /tmp/wvar.XXXXXX)
mkdir -p $D/wenet_amd/csrc $D/include $ROOT/tools/variants
cp -r $ROOT/wenet_amd/csrc/*.h $ROOT/wenet_amd/csrc/*.hip $ROOT/wenet_amd/csrc/*.inc $ROOT/wenet_amd/csrc/*.cpp $ROOT/wenet_amd/csrc/Makefile $ROOT/wenet_amd/csrc/tables $D/wenet_amd/csrc/
cp $ROOT/include/*.h $D/include/
make -s -j8 -C $D/wenet_amd/csrc EXTRA="$*" ../libwenet_rx.so 2>&1 | grep -E "error|Error" || true
mv $D/wenet_amd/libwenet_rx.so $ROOT/tools/variants/$N.so
rm -rf $D
ls -la $ROOT/tools/variants/$N.so
