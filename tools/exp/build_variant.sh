#!/bin/bash
# usage: build_variant.sh NAME "sed-expr" -- builds tools/exp/lib_NAME.so from csrc with conv_direct.hip patched by sed
set -e
NAME=$1; EXPR=$2
SRC=/root/repo/bcnn_amd/csrc
TMP=/tmp/exp_$NAME; rm -rf $TMP; mkdir -p $TMP
cp $SRC/*.hip $SRC/*.h $TMP/
mkdir -p $TMP/../../include 2>/dev/null || true
sed -i "s#\"../../include/bcnn_hip.h\"#\"/root/repo/include/bcnn_hip.h\"#" $TMP/common.h
sed -i -E "$EXPR" $TMP/${FILE:-conv_direct}.hip
cd $TMP
for f in runtime blas1 activation batchnorm pool depthwise conv_igemm conv_igemm_dma conv_dw_dma conv_bwd conv_direct conv gemm next; do
  if [ $f = ${FILE:-conv_direct} ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $DEFS -c $f.hip -o $f.o & else cp /root/repo/bcnn_amd/csrc/build/$f.o $f.o; fi
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/exp/lib_$NAME.so *.o
echo built tools/exp/lib_$NAME.so
