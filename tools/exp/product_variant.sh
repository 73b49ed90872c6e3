#!/bin/bash
# product-flavoured variant: all objects from build/, FILE recompiled with DEFS -> tools/exp/lib_NAME.so
set -e
NAME=$1; FILES=$2; DEFS=$3
ROOT=/root/repo; SRC=$ROOT/bcnn_amd/csrc
TMP=/tmp/pvar_$NAME; rm -rf $TMP; mkdir -p $TMP
cp $SRC/build/*.o $TMP/
for F in $FILES; do (cd $SRC && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm $DEFS -c $F.hip -o $TMP/$F.o); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/exp/lib_$NAME.so $TMP/*.o -ldl
echo built lib_$NAME.so
