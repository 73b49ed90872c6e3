// comm.hip -- RCCL behind the C-ABI: the data-parallel exchange step of the hot path for a plain C consumer.
//
// Process model = the reference's (src/cli/bcnn_cl.c:281-285, src/bcnn_utils.c:201): ONE device per process, set once
// (bcnn_hip_set_device); N processes on a node form the job. The only collective of the path is the all-reduce
// (sum, fp32) of the flat weight-gradient arena after backward; it runs on a private HIP stream so
// that buckets reduced early overlap the rest of backward, with event ordering against the compute stream on both
// sides. xGMI is point-to-point: a few large buckets (8 MB, chosen by the caller) amortise the ring's per-link latency.
//
// RCCL is resolved with dlopen at bcnn_hip_comm_init, not linked: single-process users never load it, and inside a
// process that already carries an RCCL (PyTorch-ROCm's bundled librccl.so.1) the same library instance is reused.
// Every RCCL / HIP return is checked and fatal (print + exit), the reference's device-error convention
// (src/bcnn_utils.h:174-195).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <string>

#include "common.h"

namespace bcnn_hip {

struct CommApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    const char* (*GetErrorString)(ncclResult_t);
};

struct Comm {
    void* lib = nullptr;
    CommApi api{};
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    int rank = 0, world = 0;  // world == 0: not initialised
};
static Comm g_comm;  // per process, like the one device the process owns

#define RCCL_CHECK(expr)                                                                                        \
    do {                                                                                                        \
        ncclResult_t _r = (expr);                                                                               \
        if (_r != ncclSuccess) {                                                                                \
            fprintf(stderr, "[bcnn_hip] %s:%d: %s failed: %s\n", __FILE__, __LINE__, #expr,                     \
                    g_comm.api.GetErrorString ? g_comm.api.GetErrorString(_r) : "RCCL error");                  \
            exit(100 + (int)_r);                                                                                \
        }                                                                                                       \
    } while (0)

static void* must_sym(void* lib, const char* name) {
    void* p = dlsym(lib, name);
    if (!p) {
        fprintf(stderr, "[bcnn_hip] RCCL symbol %s not found: %s\n", name, dlerror());
        exit(1);
    }
    return p;
}

static void load_rccl() {
    if (g_comm.lib) return;
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        g_comm.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_comm.lib) break;
    }
    if (!g_comm.lib) {
        fprintf(stderr, "[bcnn_hip] cannot load RCCL (librccl.so.1): %s\n", dlerror());
        exit(1);
    }
    g_comm.api.GetUniqueId = (decltype(g_comm.api.GetUniqueId))must_sym(g_comm.lib, "ncclGetUniqueId");
    g_comm.api.CommInitRank = (decltype(g_comm.api.CommInitRank))must_sym(g_comm.lib, "ncclCommInitRank");
    g_comm.api.CommDestroy = (decltype(g_comm.api.CommDestroy))must_sym(g_comm.lib, "ncclCommDestroy");
    g_comm.api.AllReduce = (decltype(g_comm.api.AllReduce))must_sym(g_comm.lib, "ncclAllReduce");
    g_comm.api.Broadcast = (decltype(g_comm.api.Broadcast))must_sym(g_comm.lib, "ncclBroadcast");
    g_comm.api.GetErrorString = (decltype(g_comm.api.GetErrorString))must_sym(g_comm.lib, "ncclGetErrorString");
}

// Rendezvous through a file every rank can see: rank 0 publishes {magic, world, ncclUniqueId} with an atomic rename,
// the others poll for it (bounded). The path must be unique per job (stale files of an earlier job are not detected).
struct IdRecord {
    char magic[8];
    int world;
    ncclUniqueId id;
};

static void exchange_id(int rank, int world, const char* path, ncclUniqueId* id) {
    IdRecord rec;
    if (rank == 0) {
        RCCL_CHECK(g_comm.api.GetUniqueId(id));
        memcpy(rec.magic, "BCNNHIP1", 8);
        rec.world = world;
        rec.id = *id;
        const std::string tmp = std::string(path) + ".tmp";
        FILE* fp = fopen(tmp.c_str(), "wb");
        if (!fp || fwrite(&rec, sizeof(rec), 1, fp) != 1 || fclose(fp) != 0 || rename(tmp.c_str(), path) != 0) {
            fprintf(stderr, "[bcnn_hip] bcnn_hip_comm_init: cannot publish the communicator id at %s\n", path);
            exit(1);
        }
        return;
    }
    for (int tries = 0; tries < 1200; ++tries) {  // 120 s
        FILE* fp = fopen(path, "rb");
        if (fp) {
            const size_t got = fread(&rec, sizeof(rec), 1, fp);
            fclose(fp);
            if (got == 1 && memcmp(rec.magic, "BCNNHIP1", 8) == 0) {
                if (rec.world != world) {
                    fprintf(stderr, "[bcnn_hip] bcnn_hip_comm_init: %s was published for world size %d, this rank says %d\n",
                            path, rec.world, world);
                    exit(1);
                }
                *id = rec.id;
                return;
            }
        }
        usleep(100000);
    }
    fprintf(stderr, "[bcnn_hip] bcnn_hip_comm_init: rank %d timed out waiting for %s (is rank 0 running?)\n", rank, path);
    exit(1);
}

}  // namespace bcnn_hip

using namespace bcnn_hip;

extern "C" {

void bcnn_hip_comm_init(int rank, int world, const char* id_path) {
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && (!id_path || !id_path[0]))) {
        fprintf(stderr, "[bcnn_hip] bcnn_hip_comm_init: bad arguments (rank %d, world %d, id path %s)\n", rank, world,
                id_path ? id_path : "(null)");
        exit(1);
    }
    if (g_comm.world != 0) {
        fprintf(stderr, "[bcnn_hip] bcnn_hip_comm_init: a communicator already exists in this process\n");
        exit(1);
    }
    load_rccl();
    ncclUniqueId id;
    if (world == 1 && (!id_path || !id_path[0])) {
        RCCL_CHECK(g_comm.api.GetUniqueId(&id));
    } else {
        exchange_id(rank, world, id_path, &id);
    }
    RCCL_CHECK(g_comm.api.CommInitRank(&g_comm.comm, world, id, rank));  // binds to the process's current device
    HIP_CHECK(hipStreamCreateWithFlags(&g_comm.stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&g_comm.ready, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&g_comm.done, hipEventDisableTiming));
    g_comm.rank = rank;
    g_comm.world = world;
}

int bcnn_hip_comm_world(void) { return g_comm.world; }
int bcnn_hip_comm_rank(void) { return g_comm.rank; }

void bcnn_hip_allreduce_sum(float* buf_d, size_t n) {
    if (g_comm.world == 0) {
        fprintf(stderr, "[bcnn_hip] bcnn_hip_allreduce_sum: no communicator (call bcnn_hip_comm_init first)\n");
        exit(1);
    }
    if (n == 0) return;
    // the collective reads what the compute stream has produced so far ...
    HIP_CHECK(hipEventRecord(g_comm.ready, current_stream()));
    HIP_CHECK(hipStreamWaitEvent(g_comm.stream, g_comm.ready, 0));
    RCCL_CHECK(g_comm.api.AllReduce(buf_d, buf_d, n, ncclFloat, ncclSum, g_comm.comm, g_comm.stream));
}

void bcnn_hip_broadcast(float* buf_d, size_t n, int root) {
    if (g_comm.world == 0) {
        fprintf(stderr, "[bcnn_hip] bcnn_hip_broadcast: no communicator (call bcnn_hip_comm_init first)\n");
        exit(1);
    }
    if (n == 0) return;
    HIP_CHECK(hipEventRecord(g_comm.ready, current_stream()));
    HIP_CHECK(hipStreamWaitEvent(g_comm.stream, g_comm.ready, 0));
    RCCL_CHECK(g_comm.api.Broadcast(buf_d, buf_d, n, ncclFloat, root, g_comm.comm, g_comm.stream));
}

void bcnn_hip_comm_join(void) {
    if (g_comm.world == 0) return;
    // ... and whatever the compute stream does next sees every collective queued so far (no host block)
    HIP_CHECK(hipEventRecord(g_comm.done, g_comm.stream));
    HIP_CHECK(hipStreamWaitEvent(current_stream(), g_comm.done, 0));
}

void bcnn_hip_comm_destroy(void) {
    if (g_comm.world == 0) return;
    HIP_CHECK(hipStreamSynchronize(g_comm.stream));
    RCCL_CHECK(g_comm.api.CommDestroy(g_comm.comm));
    HIP_CHECK(hipEventDestroy(g_comm.ready));
    HIP_CHECK(hipEventDestroy(g_comm.done));
    HIP_CHECK(hipStreamDestroy(g_comm.stream));
    g_comm.comm = nullptr; g_comm.stream = nullptr; g_comm.ready = nullptr; g_comm.done = nullptr;
    g_comm.world = 0; g_comm.rank = 0;
}

}  // extern "C"
