#!/bin/bash
# A/B of conv_window.hip variants on one box: win_variants.sh name1 name2 ... (tools/exp/lib_<name>.so, built with
# tools/exp/variant.sh <name> conv_window "-DWABL_..."; "product" = the shipped library). REPS rounds, alternating.
cd ${GRAFT_REPO_ROOT:-.}
for rep in $(seq ${REPS:-2}); do
  for v in "$@"; do
    if [ "$v" = product ]; then unset BCNN_HIP_LIB; else export BCNN_HIP_LIB=$PWD/tools/exp/lib_$v.so; fi
    python3 tools/prof_window.py ${SHAPE:-} 2>&1 | tail -1
  done
done
