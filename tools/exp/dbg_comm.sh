cd $GRAFT_REPO_ROOT
gcc -std=gnu99 -O1 -DBCNN_USE_HIP -Iinclude tools/dp_train.c -o /tmp/dp_train -Lbcnn_amd/lib -lbcnn -lbcnn_hip -Wl,-rpath,$PWD/bcnn_amd/lib -lm
for i in 1 2; do /tmp/dp_train 0 1 - 6 nocomm 2>/dev/null; done
for i in 1 2; do /tmp/dp_train 0 1 - 6 2>/dev/null; done
/tmp/dp_train 0 1 - 1 nocomm 2>/dev/null; /tmp/dp_train 0 1 - 1 2>/dev/null
/tmp/dp_train 0 1 - 2 nocomm 2>/dev/null; /tmp/dp_train 0 1 - 2 2>/dev/null
