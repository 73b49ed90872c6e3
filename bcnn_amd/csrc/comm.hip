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
#include <ctime>
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
    int refs = 0;             // nets (or direct callers) holding the communicator: destroyed when the last lets go
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
    // BCNN_HIP_RCCL_LIB: an explicit file instead of the system's RCCL (a site-specific build; the test double of
    // tests/fake_rccl inside a process that already holds PyTorch's librccl, where the soname would resolve to that one)
    const char* forced = getenv("BCNN_HIP_RCCL_LIB");
    const char* names[] = {forced && forced[0] ? forced : "librccl.so.1", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        g_comm.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_comm.lib) break;
        if (forced && forced[0] && n == names[0]) {
            fprintf(stderr, "[bcnn_hip] cannot load BCNN_HIP_RCCL_LIB=%s: %s\n", forced, dlerror());
            exit(1);
        }
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

// Rendezvous through a file every rank can see: rank 0 publishes {magic, world, job nonce, publish time, payload}
// with an atomic rename, the others poll for it (bounded). Three things keep a record of an EARLIER job at the same
// path from being taken for this job's (a mismatched ncclUniqueId would block ncclCommInitRank forever):
//   * rank 0 unlinks the file once ncclCommInitRank has returned -- that call is collective, every rank has read it;
//   * the launcher may export BCNN_HIP_JOB_NONCE (any string, the same on all ranks of a job, different per job):
//     it is stored in the record and a record with another nonce is ignored (the poll goes on until rank 0 of THIS
//     job renames its own record into place);
//   * a record published more than `BCNN_HIP_ID_MAX_AGE_S` (default 600) seconds before the fetch started is stale.
struct IdRecord {
    char magic[8];
    int world;
    unsigned payload_bytes;
    unsigned long long nonce;       // FNV-1a of BCNN_HIP_JOB_NONCE, 0 when the variable is unset
    long long published_unix_s;
    unsigned char payload[256];
};
static_assert(sizeof(ncclUniqueId) <= 256, "ncclUniqueId does not fit the rendezvous record");

static unsigned long long job_nonce() {
    const char* s = getenv("BCNN_HIP_JOB_NONCE");
    if (!s || !s[0]) return 0ull;
    unsigned long long h = 1469598103934665603ull;
    for (; *s; ++s) h = (h ^ (unsigned char)*s) * 1099511628211ull;
    return h ? h : 1ull;
}

static long long max_record_age_s() {
    const char* s = getenv("BCNN_HIP_ID_MAX_AGE_S");
    const long long v = s ? atoll(s) : 0;
    return v > 0 ? v : 600;
}

static int rendezvous_publish(const char* path, const void* blob, size_t n, int world) {
    if (!path || !path[0] || n > sizeof(((IdRecord*)0)->payload)) return -1;
    IdRecord rec;
    memset(&rec, 0, sizeof(rec));
    memcpy(rec.magic, "BCNNHIP2", 8);
    rec.world = world;
    rec.payload_bytes = (unsigned)n;
    rec.nonce = job_nonce();
    rec.published_unix_s = (long long)time(nullptr);
    memcpy(rec.payload, blob, n);
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long long)getpid());
    FILE* fp = fopen(tmp.c_str(), "wb");
    if (!fp) return -1;
    const bool ok = fwrite(&rec, sizeof(rec), 1, fp) == 1;
    if (fclose(fp) != 0 || !ok || rename(tmp.c_str(), path) != 0) {
        unlink(tmp.c_str());
        return -1;
    }
    return 0;
}

// 0: payload fetched; 1: timed out; 2: a record of THIS job says another world size
static int rendezvous_fetch(const char* path, void* blob, size_t n, int world, int timeout_ms, int* seen_world) {
    if (!path || !path[0]) return 1;
    const unsigned long long nonce = job_nonce();
    const long long oldest = (long long)time(nullptr) - max_record_age_s();
    for (int waited = 0;; waited += 50) {
        IdRecord rec;
        FILE* fp = fopen(path, "rb");
        if (fp) {
            const size_t got = fread(&rec, sizeof(rec), 1, fp);
            fclose(fp);
            if (got == 1 && memcmp(rec.magic, "BCNNHIP2", 8) == 0 && rec.nonce == nonce && rec.payload_bytes == n &&
                rec.published_unix_s >= oldest) {
                if (rec.world != world) {
                    if (seen_world) *seen_world = rec.world;
                    return 2;
                }
                memcpy(blob, rec.payload, n);
                return 0;
            }
        }
        if (waited >= timeout_ms) return 1;
        usleep(50000);
    }
}

static void exchange_id(int rank, int world, const char* path, ncclUniqueId* id) {
    if (rank == 0) {
        RCCL_CHECK(g_comm.api.GetUniqueId(id));
        if (rendezvous_publish(path, id, sizeof(*id), world) != 0) {
            fprintf(stderr, "[bcnn_hip] bcnn_hip_comm_init: cannot publish the communicator id at %s\n", path);
            exit(1);
        }
        return;
    }
    int seen = 0;
    const int r = rendezvous_fetch(path, id, sizeof(*id), world, 120000, &seen);
    if (r == 2) {
        fprintf(stderr, "[bcnn_hip] bcnn_hip_comm_init: %s was published for world size %d, this rank says %d\n", path,
                seen, world);
        exit(1);
    }
    if (r != 0) {
        fprintf(stderr,
                "[bcnn_hip] bcnn_hip_comm_init: rank %d timed out waiting for %s (is rank 0 running with the same "
                "BCNN_HIP_JOB_NONCE?)\n", rank, path);
        exit(1);
    }
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
    // collective: every rank has read the record by now; a later job at the same path must not find it
    if (rank == 0 && id_path && id_path[0]) unlink(id_path);
    HIP_CHECK(hipStreamCreateWithFlags(&g_comm.stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&g_comm.ready, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&g_comm.done, hipEventDisableTiming));
    g_comm.rank = rank;
    g_comm.world = world;
    g_comm.refs = 1;
}

void bcnn_hip_comm_retain(void) {
    if (g_comm.world == 0) {
        fprintf(stderr, "[bcnn_hip] bcnn_hip_comm_retain: no communicator (call bcnn_hip_comm_init first)\n");
        exit(1);
    }
    ++g_comm.refs;
}

int bcnn_hip_rendezvous_publish(const char* path, const void* blob, size_t n, int world) {
    return rendezvous_publish(path, blob, n, world);
}

int bcnn_hip_rendezvous_fetch(const char* path, void* blob, size_t n, int world, int timeout_ms) {
    return rendezvous_fetch(path, blob, n, world, timeout_ms, nullptr);
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
    if (--g_comm.refs > 0) return;  // another net of this process still trains over it
    HIP_CHECK(hipStreamSynchronize(g_comm.stream));
    RCCL_CHECK(g_comm.api.CommDestroy(g_comm.comm));
    HIP_CHECK(hipEventDestroy(g_comm.ready));
    HIP_CHECK(hipEventDestroy(g_comm.done));
    HIP_CHECK(hipStreamDestroy(g_comm.stream));
    g_comm.comm = nullptr; g_comm.stream = nullptr; g_comm.ready = nullptr; g_comm.done = nullptr;
    g_comm.world = 0; g_comm.rank = 0; g_comm.refs = 0;
}

}  // extern "C"
