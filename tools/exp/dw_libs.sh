#!/bin/bash
# tools/prof_dw.py (stand-alone depthwise kernels, MobileNet shapes) for variant libraries: dw_libs.sh tools/exp/lib_a.so ...
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  echo "== $lib"
  BCNN_HIP_LIB=$PWD/$lib python3 tools/prof_dw.py 10 2>&1 | grep -E "^c(512|1024) |sum"
done
