#!/bin/bash
# A/B of experiment libraries on the same box: LIBS="tools/exp/lib_a.so tools/exp/lib_b.so" [SHAPES="..."] [REPS=2] ab.sh
# prints the fused Winograd forward / dX times per library, alternating libraries so that clock drift hits both
SHAPES=${SHAPES:-"128 64 56 56 64;128 128 28 28 128;128 256 14 14 256;128 512 7 7 512"}
IFS=';' read -ra SH <<< "$SHAPES"
for shape in "${SH[@]}"; do
  echo "== $shape"
  for rep in $(seq ${REPS:-2}); do
    for lib in $LIBS; do
      printf "%-36s" $(basename $lib)
      BCNN_HIP_LIB=$PWD/$lib BCNN_HIP_WINOGRAD=0 BCNN_HIP_WINOGRAD_FUSED=1 BCNN_HIP_WINOGRAD_DW_FUSED=${DWF:-0} python3 tools/prof_layer.py $shape 3 1 1 ${ITERS:-10} 2>&1 \
        | grep "winograd\|conv_dw" | awk '{printf "%s %s  ", $1, $2} END {print ""}'
    done
  done
done
