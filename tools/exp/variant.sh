#!/bin/bash
# usage: variant.sh NAME FILE "DEFS" -- tools/exp/lib_NAME.so = the experiment library with FILE.hip recompiled with extra -D flags
set -e
NAME=$1; FILE=$2; DEFS=$3
ROOT=$(cd $(dirname $0)/../.. && pwd)
SRC=$ROOT/bcnn_amd/csrc
make -C $SRC exp -j8 > /dev/null
TMP=/tmp/var_$NAME; rm -rf $TMP; mkdir -p $TMP
cp $SRC/build_exp/*.o $TMP/
(cd $SRC && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -DBCNN_HIP_EXPERIMENT $DEFS -c $FILE.hip -o $TMP/$FILE.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/exp/lib_$NAME.so $TMP/*.o -ldl
echo built tools/exp/lib_$NAME.so
