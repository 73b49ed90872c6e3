cd $GRAFT_REPO_ROOT
for t in 4 5 3 2 0; do echo "tile $t"; BCNN_HIP_IGEMM_TILE=$t python tools/prof_layer.py 128 512 7 7 512 3 1 1 8 | grep -E "conv_fwd|conv_dx"; done
for t in 4 5 3 2; do echo "s2 tile $t"; BCNN_HIP_IGEMM_TILE=$t python tools/prof_layer.py 128 256 14 14 512 3 2 1 8 | grep -E "conv_fwd|conv_dx"; done
