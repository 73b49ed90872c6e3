#!/bin/bash
# usage: variant_multi.sh NAME "DEFS" FILE... -- tools/exp/lib_NAME.so = the experiment library with several files recompiled with -D flags
set -e
NAME=$1; DEFS=$2; shift; shift
ROOT=$(cd $(dirname $0)/../.. && pwd)
SRC=$ROOT/bcnn_amd/csrc
make -C $SRC exp -j8 > /dev/null
TMP=/tmp/var_$NAME; rm -rf $TMP; mkdir -p $TMP
cp $SRC/build_exp/*.o $TMP/
for FILE in "$@"; do
  (cd $SRC && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -DBCNN_HIP_EXPERIMENT $DEFS -c $FILE.hip -o $TMP/$FILE.o) &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/exp/lib_$NAME.so $TMP/*.o -ldl
echo built tools/exp/lib_$NAME.so
