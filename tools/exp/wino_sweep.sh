#!/bin/bash
# Winograd F(2x2,3x3) vs the direct LDS-DMA kernels on the 3x3 / s1 shapes of the ResNet-18 step at N = 128
# (experiment build: BCNN_HIP_WINOGRAD=0 forces direct, =1 forces Winograd). Run under gpurun from the repo root.
export BCNN_HIP_LIB=$PWD/bcnn_amd/lib/libbcnn_hip_exp.so
for shape in "128 64 56 56 64" "128 128 28 28 128" "128 256 14 14 256" "128 512 7 7 512" ${EXTRA_SHAPES}; do
  for w in ${VARIANTS:-000 100 011}; do   # <three-kernel form><fused fwd/dX><fused dW>
    echo "== N C H W F = $shape  BCNN_HIP_WINOGRAD=${w:0:1} BCNN_HIP_WINOGRAD_FUSED=${w:1:1} BCNN_HIP_WINOGRAD_DW_FUSED=${w:2:1}"
    BCNN_HIP_WINOGRAD=${w:0:1} BCNN_HIP_WINOGRAD_FUSED=${w:1:1} BCNN_HIP_WINOGRAD_DW_FUSED=${w:2:1} python3 tools/prof_layer.py $shape 3 1 1 ${ITERS:-10} 2>&1 | grep -v amdgpu.ids
  done
done
