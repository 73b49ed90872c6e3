#!/bin/bash
# ablations / parameter variants of the LDS-staged depthwise kernels, timed on MobileNet's layer shapes (run on the GPU box)
set -e
cd $(dirname $0)/../..
for v in "base:" "nodw:-DDWL_ABL=1" "nodx:-DDWL_ABL=2" "nobn:-DDWL_ABL=4" "none:-DDWL_ABL=7" $DW_EXTRA; do
  name=${v%%:*}; defs=${v#*:}
  bash tools/exp/variant.sh dw_$name depthwise_lds "$defs" > /dev/null
  BCNN_HIP_LIB=tools/exp/lib_dw_$name.so python3 tools/prof_dw.py ${DW_ITERS:-10} | tail -${DW_TAIL:-10}
done
