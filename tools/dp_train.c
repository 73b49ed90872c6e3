/* tools/dp_train.c -- a plain C consumer of the public API that trains data-parallel, one process per GPU, with the
 * gradient all-reduce INSIDE the library (bcnn_set_data_parallel_comm: RCCL over xGMI behind the C-ABI).
 *
 *   gcc -std=gnu99 -DBCNN_USE_HIP -Iinclude tools/dp_train.c -Lbcnn_amd/lib -lbcnn -lbcnn_hip -lm -o dp_train
 *   for r in 0 1 ... N-1:  BCNN_HIP_JOB_NONCE=$$ ./dp_train $r N /tmp/job.id 20 &   (rank r uses GPU r; the nonce marks THIS job's id file)
 *   ./dp_train 0 1 - 20 nocomm                                        (single process, no communicator)
 *
 * Every rank builds the same net from the same seed (identical initial parameters), feeds its own shard of a synthetic
 * data set, and ends with identical parameters on all ranks: the printed checksum is the same everywhere, and for
 * world = 1 it is bit-identical to the `nocomm` run (tests/test_comm_cabi.py). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <bcnn/bcnn.h>
#include <bcnn_hip.h>

#define CHECK(x) do { if ((x) != BCNN_SUCCESS) { fprintf(stderr, "failed: %s\n", #x); return 2; } } while (0)

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "Usage: %s <rank> <world> <id file | -> <steps> [nocomm]\n", argv[0]);
        return 1;
    }
    const int rank = atoi(argv[1]), world = atoi(argv[2]), steps = atoi(argv[4]);
    const char *id_path = strcmp(argv[3], "-") ? argv[3] : NULL;
    const int use_comm = !(argc > 5 && !strcmp(argv[5], "nocomm"));
    const int batch = 8, classes = 10;

    bcnn_hip_set_device(rank % bcnn_hip_device_count()); /* one device per process, set once (bcnn_cl.c:281-285) */
    bcnn_net *net = NULL;
    CHECK(bcnn_init_net(&net, BCNN_MODE_TRAIN));
    bcnn_set_log_context(net, NULL, BCNN_LOG_SILENT);
    bcnn_set_input_shape(net, 32, 32, 3, batch);
    srand(7); /* the builders draw their Xavier weights from rand(): same parameters on every rank */
    CHECK(bcnn_add_convolutional_layer(net, 64, 3, 1, 1, 1, 1, BCNN_FILLER_XAVIER, BCNN_ACT_RELU, 0, "input", "c1"));
    CHECK(bcnn_add_maxpool_layer(net, 2, 2, BCNN_PADDING_SAME, "c1", "p1"));
    CHECK(bcnn_add_convolutional_layer(net, 64, 3, 1, 1, 1, 1, BCNN_FILLER_XAVIER, BCNN_ACT_RELU, 0, "p1", "c2"));
    CHECK(bcnn_add_avgpool_layer(net, "c2", "gap"));
    CHECK(bcnn_add_fullc_layer(net, classes, BCNN_FILLER_XAVIER, BCNN_ACT_NONE, 0, "gap", "fc"));
    CHECK(bcnn_add_softmax_layer(net, "fc", "prob"));
    CHECK(bcnn_add_cost_layer(net, BCNN_LOSS_EUCLIDEAN, BCNN_METRIC_ERROR_RATE, 1.0f, "prob", "label", "cost"));
    CHECK(bcnn_compile_net(net));
    bcnn_set_sgd_optimizer(net, 0.05f, 0.9f);
    bcnn_set_weight_regularizer(net, 5e-4f);
    if (use_comm) CHECK(bcnn_set_data_parallel_comm(net, rank, world, id_path));

    bcnn_tensor *in = bcnn_get_tensor_by_name(net, "input"), *lab = bcnn_get_tensor_by_name(net, "label");
    const int in_sz = batch * 3 * 32 * 32;
    float loss = 0.f;
    for (int it = 0; it < steps; ++it) {
        srand(1000 + 97 * it + rank); /* this rank's shard of step `it` */
        for (int i = 0; i < in_sz; ++i) in->data[i] = 2.0f * rand() / RAND_MAX - 1.0f;
        memset(lab->data, 0, sizeof(float) * batch * classes);
        for (int b = 0; b < batch; ++b) lab->data[b * classes + rand() % classes] = 1.0f;
        loss = bcnn_train_on_batch(net); /* loader hook uploads, forward, backward (+ all-reduce), update */
    }
    double sum = 0.0, asum = 0.0;
    const char *names[] = {"input_w", "p1_w", "gap_w", "gap_b"};
    for (int k = 0; k < 4; ++k) {
        bcnn_tensor *t = bcnn_get_tensor_by_name(net, names[k]); /* refreshes the host copy */
        if (!t) { fprintf(stderr, "no tensor %s\n", names[k]); return 3; }
        const int sz = t->n * t->c * t->h * t->w;
        for (int i = 0; i < sz; ++i) { sum += t->data[i]; asum += t->data[i] < 0 ? -t->data[i] : t->data[i]; }
    }
    printf("rank %d/%d loss %.6f checksum %.9e %.9e\n", rank, world, loss, sum, asum);
    bcnn_end_net(&net);
    return 0;
}
