/* tests/fake_rccl/fake_rccl.c -- TEST DOUBLE for librccl.so.1 (never shipped, never on the product's library path).
 *
 * The GPU pool hands out one device, so the library's own RCCL path (bcnn_amd/csrc/comm.hip) had only ever run at world
 * size 1: the non-zero-rank branch of its rendezvous, the id-file hand-over and the multi-rank all-reduce never executed.
 * This file implements the six RCCL entry points comm.hip resolves with dlopen -- ncclGetUniqueId, ncclCommInitRank,
 * ncclCommDestroy, ncclAllReduce, ncclBroadcast, ncclGetErrorString -- for several PROCESSES SHARING ONE GPU: device
 * buffers are staged through the host and exchanged as files under a directory named by the unique id (write to a
 * temporary name, rename, poll). Sums are formed in rank order on every rank, so all ranks get bit-identical results.
 * A test puts a directory holding this library as `librccl.so.1` at the front of LD_LIBRARY_PATH of a plain C consumer.
 *
 * build: gcc -shared -fPIC -O1 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.c -o librccl.so.1 -L/opt/rocm/lib -lamdhip64 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct fake_comm { int rank, world; unsigned long seq; char dir[120]; };
typedef struct fake_comm *ncclComm_t;

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof(*id));
    const char *base = getenv("FAKE_RCCL_DIR");
    snprintf(id->internal, sizeof(id->internal), "%s/job_%ld_%ld", base ? base : "/tmp", (long)getpid(), (long)time(NULL));
    return mkdir(id->internal, 0700) == 0 ? ncclSuccess : ncclSystemError;
}

/* publish `bytes` of `data` as <dir>/<tag>.<seq>.<rank> */
static int put(const struct fake_comm *c, const char *tag, const void *data, size_t bytes) {
    char tmp[300], dst[256];
    snprintf(dst, sizeof(dst), "%s/%s.%lu.%d", c->dir, tag, c->seq, c->rank);
    snprintf(tmp, sizeof(tmp), "%s.tmp", dst);
    FILE *f = fopen(tmp, "wb");
    if (!f) return -1;
    const int ok = bytes == 0 || fwrite(data, 1, bytes, f) == bytes;
    if (fclose(f) != 0 || !ok || rename(tmp, dst) != 0) return -1;
    return 0;
}

/* wait (bounded) for rank r's record of this sequence number and read it */
static int get(const struct fake_comm *c, const char *tag, int r, void *data, size_t bytes) {
    char src[256];
    snprintf(src, sizeof(src), "%s/%s.%lu.%d", c->dir, tag, c->seq, r);
    for (int waited = 0; waited < 120000; waited += 5) {
        FILE *f = fopen(src, "rb");
        if (f) {
            const int ok = bytes == 0 || fread(data, 1, bytes, f) == bytes;
            fclose(f);
            return ok ? 0 : -1;
        }
        usleep(5000);
    }
    fprintf(stderr, "[fake rccl] rank %d timed out waiting for %s\n", c->rank, src);
    return -1;
}

static int barrier(struct fake_comm *c, const char *tag) {
    if (put(c, tag, NULL, 0) != 0) return -1;
    for (int r = 0; r < c->world; ++r)
        if (get(c, tag, r, NULL, 0) != 0) return -1;
    ++c->seq;
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank) {
    struct fake_comm *c = (struct fake_comm *)calloc(1, sizeof(*c));
    if (!c) return ncclSystemError;
    c->rank = rank; c->world = world;
    snprintf(c->dir, sizeof(c->dir), "%s", id.internal);
    if (barrier(c, "init") != 0) { free(c); return ncclSystemError; }  /* collective, like the real call */
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) { free(c); return ncclSuccess; }

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t st) {
    if (t != ncclFloat || op != ncclSum) return ncclInvalidArgument;
    const size_t bytes = count * sizeof(float);
    float *mine = (float *)malloc(bytes), *other = (float *)malloc(bytes), *total = (float *)calloc(count, sizeof(float));
    if (!mine || !other || !total) return ncclSystemError;
    /* stream order: everything queued on `st` before this call (the event wait comm.hip put there) has to be done */
    if (hipMemcpyAsync(mine, send, bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    if (put(c, "ar", mine, bytes) != 0) return ncclSystemError;
    for (int r = 0; r < c->world; ++r) {  /* rank order on every rank: identical sums everywhere */
        const float *src = mine;
        if (r != c->rank) {
            if (get(c, "ar", r, other, bytes) != 0) return ncclSystemError;
            src = other;
        }
        for (size_t i = 0; i < count; ++i) total[i] += src[i];
    }
    ++c->seq;
    if (hipMemcpyAsync(recv, total, bytes, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    free(mine); free(other); free(total);
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t st) {
    if (t != ncclFloat) return ncclInvalidArgument;
    const size_t bytes = count * sizeof(float);
    float *buf = (float *)malloc(bytes);
    if (!buf) return ncclSystemError;
    if (c->rank == root) {
        if (hipMemcpyAsync(buf, send, bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
        if (put(c, "bc", buf, bytes) != 0) return ncclSystemError;
    } else if (get(c, "bc", root, buf, bytes) != 0) return ncclSystemError;
    ++c->seq;
    if (hipMemcpyAsync(recv, buf, bytes, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    free(buf);
    return ncclSuccess;
}
