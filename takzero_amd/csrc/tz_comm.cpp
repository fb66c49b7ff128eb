// tz_comm.cpp — the exchange between the self-play shards of one job, native, at the C ABI.
//
// The reference's "cluster" is N independent processes appending to the same files of one directory (README.md:130;
// selfplay/src/main.rs:332-366): it has no collective of its own.  With one process per GPU the same hand-over — every
// shard's finished targets and replays reach `learn`, a new model reaches every shard — is
//   * an all-gather of byte counts followed by an all-gather of the (padded) packed records, and
//   * a broadcast of the weights from the rank that read model_latest.ot,
// over RCCL (ncclAllGather / ncclBroadcast on the shard's GPU; xGMI inside a node).  librccl is opened with dlopen on first
// use, so that hosts which never create a communicator do not depend on it and a process that already carries an RCCL
// (PyTorch's) shares that copy.
// A second transport, "fs", moves the same bytes through files of a shared directory — the reference's own medium — for jobs
// without RCCL and for the CPU tests (world 2, no GPU): same packing, same ordering, same code above the transport.
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <ctime>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "tz_engine.h"
#include "tz_host_exchange.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

// Which RCCL.  A process may already carry one (PyTorch ships its own copy, without a SONAME, next to its own copy of the HIP
// runtime): asking the loader for "librccl.so.1" would then map a second RCCL beside it — two sets of globals and a double free
// when the process exits (measured: tools/rccl_copies_check.py).  So an RCCL that is already mapped is preferred, provided it
// is bound to the same HIP runtime as this library (its hipMalloc is ours): buffers of one runtime mean nothing to the other.
int collect_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
    const char* path = info->dlpi_name;
    if (!path) return 0;
    const char* base = strrchr(path, '/');
    base = base ? base + 1 : path;
    if (!strncmp(base, "librccl.so", 10)) static_cast<std::vector<std::string>*>(out)->push_back(path);
    return 0;
}

bool same_hip_runtime(void* handle) {
    void* theirs = dlsym(handle, "hipMalloc");
    hipError_t (*ours)(void**, size_t) = &hipMalloc;
    return theirs == reinterpret_cast<void*>(ours);
}

Rccl* rccl() {
    static Rccl r;
    if (r.handle || !r.error.empty()) return &r;
    std::vector<std::string> candidates;
    const char* forced = getenv("TZ_RCCL_LIB");   // an explicit library, e.g. /opt/rocm/lib/librccl.so.1
    if (forced && *forced) candidates.push_back(forced);
    else dl_iterate_phdr(collect_loaded_rccl, &candidates);
    const bool had_loaded = !candidates.empty();
    if (!(forced && *forced))
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) candidates.push_back(name);
    std::string rejected;
    for (const std::string& name : candidates) {
        // RTLD_LOCAL: if a second librccl is mapped later (PyTorch ships its own and loads it with `import torch`), its symbols
        // and this copy's must not interpose - that mix is what aborted at exit with a double free (tools/rccl_copies_check.py torch_after)
        void* h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) continue;
        if (same_hip_runtime(h)) {
            r.handle = h;
            break;
        }
        rejected += " " + name;
        dlclose(h);
    }
    if (!r.handle && !rejected.empty()) {
        r.error = "no RCCL bound to this library's HIP runtime (tried:" + rejected + "): the process holds two HIP runtimes" +
                  (had_loaded ? " - import torch before takzero_amd, or" : ";") + " set TZ_RCCL_LIB";
        return &r;
    }
    if (!r.handle) {
        r.error = std::string("cannot open librccl: ") + dlerror();
        return &r;
    }
#define TZ_SYM(field, sym)                                                   \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym));     \
    if (!r.field) r.error = std::string("librccl has no symbol ") + sym;
    TZ_SYM(GetUniqueId, "ncclGetUniqueId")
    TZ_SYM(CommInitRank, "ncclCommInitRank")
    TZ_SYM(CommDestroy, "ncclCommDestroy")
    TZ_SYM(AllGather, "ncclAllGather")
    TZ_SYM(Broadcast, "ncclBroadcast")
    TZ_SYM(AllReduce, "ncclAllReduce")
    TZ_SYM(GetErrorString, "ncclGetErrorString")
#undef TZ_SYM
    return &r;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Rendezvous files in a directory that outlives the job (selfplay's data directory): rank 0 replaces whatever is there and removes its
// file once everybody has used it; the other ranks take a file only if it is not older than their own start minus a slack, so what
// a crashed job left behind is not mistaken for this job's.
const time_t g_process_start = time(nullptr);
constexpr time_t STALE_SLACK_S = 300;
std::string g_published_id_path;   // rank 0: the id file to remove once its communicator is up

int publish_fresh(const std::string& path, const void* data, size_t n) {
    (void)unlink(path.c_str());
    const std::string tmp = path + ".part";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(data, 1, n, f) != n) {
        if (f) fclose(f);
        return tz_fail(TZ_EINVAL, "comm: cannot write " + tmp);
    }
    fclose(f);
    if (rename(tmp.c_str(), path.c_str())) return tz_fail(TZ_EINVAL, "comm: cannot publish " + path);
    return TZ_OK;
}

int read_fresh(const std::string& path, void* out, size_t n, double timeout_s) {
    const double t0 = now_s();
    for (;;) {
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && (size_t)st.st_size == n && st.st_mtime + STALE_SLACK_S >= g_process_start) {
            FILE* f = fopen(path.c_str(), "rb");
            const size_t got = f ? fread(out, 1, n, f) : 0;
            if (f) fclose(f);
            if (got == n) return TZ_OK;
        }
        if (now_s() - t0 > timeout_s) return tz_fail(TZ_ESTATE, "comm: rank 0 did not publish " + path + " (a file older than this process does not count)");
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
}

}  // namespace

struct tz_comm {
    int rank = 0, world = 1, device = -1;
    bool fs = false;
    // rccl
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    unsigned char* dev_send = nullptr;
    unsigned char* dev_recv = nullptr;
    size_t send_cap = 0, recv_cap = 0;
    // fs
    std::string dir;
    uint64_t nonce = 0;   // of this job (rank 0 draws it, xch-job.bin carries it): the exchange files of another job never match
    uint64_t seq = 0;
    double timeout_s = 600.0;
    uint64_t bytes_gathered = 0, collectives = 0;
    std::vector<unsigned char> last;   // payload of the last tz_comm_all_gather, until tz_comm_take
};

namespace {

#define TZ_NCCL(c, call)                                                                                              \
    do {                                                                                                              \
        ncclResult_t _r = (call);                                                                                     \
        if (_r != ncclSuccess) return tz_fail(TZ_EDEVICE, std::string(#call) + ": " + rccl()->GetErrorString(_r));     \
    } while (0)

int ensure_dev(tz_comm* c, size_t send, size_t recv) {
    TZ_HIP(hipSetDevice(c->device));
    if (send > c->send_cap) {
        if (c->dev_send) (void)hipFree(c->dev_send);
        c->send_cap = std::max(send, (size_t)1 << 16);
        TZ_HIP(hipMalloc(&c->dev_send, c->send_cap));
    }
    if (recv > c->recv_cap) {
        if (c->dev_recv) (void)hipFree(c->dev_recv);
        c->recv_cap = std::max(recv, (size_t)1 << 16);
        TZ_HIP(hipMalloc(&c->dev_recv, c->recv_cap));
    }
    return TZ_OK;
}

// fixed-size all-gather: `bytes` from every rank, rank-major into out[world * bytes]
int all_gather_fixed(tz_comm* c, const void* mine, size_t bytes, unsigned char* out) {
    if (c->world == 1 && c->fs) {
        memcpy(out, mine, bytes);
        return TZ_OK;
    }
    if (c->fs) {
        const uint64_t seq = c->seq++;
        char tag[24];
        snprintf(tag, sizeof tag, "%016llx", (unsigned long long)c->nonce);
        auto name = [&](uint64_t s, int r) { return c->dir + "/xch-" + tag + "-" + std::to_string(s) + "-" + std::to_string(r) + ".bin"; };
        const std::string tmp = name(seq, c->rank) + ".part";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(mine, 1, bytes, f) != bytes) {
            if (f) fclose(f);
            return tz_fail(TZ_EINVAL, "comm(fs): cannot write " + tmp);
        }
        fclose(f);
        if (rename(tmp.c_str(), name(seq, c->rank).c_str())) return tz_fail(TZ_EINVAL, "comm(fs): cannot publish " + tmp);
        const double t0 = now_s();
        for (int r = 0; r < c->world; r++) {
            if (r == c->rank) {
                memcpy(out + (size_t)r * bytes, mine, bytes);
                continue;
            }
            for (;;) {
                struct stat st;
                if (stat(name(seq, r).c_str(), &st) == 0 && (size_t)st.st_size == bytes) break;
                if (now_s() - t0 > c->timeout_s) return tz_fail(TZ_ESTATE, "comm(fs): rank " + std::to_string(r) + " did not arrive");
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            FILE* g = fopen(name(seq, r).c_str(), "rb");
            const size_t got = g ? fread(out + (size_t)r * bytes, 1, bytes, g) : 0;
            if (g) fclose(g);
            if (got != bytes) return tz_fail(TZ_EINVAL, "comm(fs): short read from rank " + std::to_string(r));
        }
        // every rank has published `seq`, so every rank has finished reading `seq - 1`: my file of that round can go
        if (seq >= 1) (void)unlink(name(seq - 1, c->rank).c_str());
        return TZ_OK;
    }
    int rc = ensure_dev(c, bytes, bytes * c->world);
    if (rc) return rc;
    TZ_HIP(hipMemcpyAsync(c->dev_send, mine, bytes, hipMemcpyHostToDevice, c->stream));
    TZ_NCCL(c, rccl()->AllGather(c->dev_send, c->dev_recv, bytes, ncclUint8, c->comm, c->stream));
    TZ_HIP(hipMemcpyAsync(out, c->dev_recv, bytes * c->world, hipMemcpyDeviceToHost, c->stream));
    TZ_HIP(hipStreamSynchronize(c->stream));
    return TZ_OK;
}

// variable-size all-gather: counts first, then the payloads padded to the largest
int all_gather_var(tz_comm* c, const std::vector<unsigned char>& mine, std::vector<std::vector<unsigned char>>& all) {
    std::vector<uint64_t> sizes(c->world);
    const uint64_t my = mine.size();
    int rc = all_gather_fixed(c, &my, sizeof my, reinterpret_cast<unsigned char*>(sizes.data()));
    if (rc) return rc;
    uint64_t widest = 0;
    for (auto s : sizes) widest = std::max(widest, s);
    all.assign(c->world, {});
    c->collectives++;
    if (widest == 0) return TZ_OK;
    widest = (widest + 15) / 16 * 16;
    std::vector<unsigned char> padded(widest, 0), gathered((size_t)widest * c->world);
    if (!mine.empty()) memcpy(padded.data(), mine.data(), mine.size());
    if ((rc = all_gather_fixed(c, padded.data(), widest, gathered.data()))) return rc;
    for (int r = 0; r < c->world; r++) {
        all[r].assign(gathered.begin() + (size_t)r * widest, gathered.begin() + (size_t)r * widest + sizes[r]);
        c->bytes_gathered += sizes[r];
    }
    return TZ_OK;
}

int broadcast_bytes(tz_comm* c, void* data, size_t bytes, int root) {
    if ((c->world == 1 && c->fs) || bytes == 0) return TZ_OK;
    if (c->fs) {   // through the gather: only the root contributes
        std::vector<unsigned char> mine;
        if (c->rank == root) mine.assign((unsigned char*)data, (unsigned char*)data + bytes);
        std::vector<std::vector<unsigned char>> all;
        int rc = all_gather_var(c, mine, all);
        if (rc) return rc;
        if (all[root].size() != bytes) return tz_fail(TZ_ESTATE, "comm(fs): broadcast size mismatch");
        if (c->rank != root) memcpy(data, all[root].data(), bytes);
        return TZ_OK;
    }
    int rc = ensure_dev(c, bytes, 0);
    if (rc) return rc;
    if (c->rank == root) TZ_HIP(hipMemcpyAsync(c->dev_send, data, bytes, hipMemcpyHostToDevice, c->stream));
    TZ_NCCL(c, rccl()->Broadcast(c->dev_send, c->dev_send, bytes, ncclUint8, root, c->comm, c->stream));
    if (c->rank != root) TZ_HIP(hipMemcpyAsync(data, c->dev_send, bytes, hipMemcpyDeviceToHost, c->stream));
    TZ_HIP(hipStreamSynchronize(c->stream));
    return TZ_OK;
}

}  // namespace

extern "C" {

int tz_comm_unique_id(unsigned char* id_out) {
    if (!id_out) return tz_fail(TZ_EINVAL, "tz_comm_unique_id: null argument");
    Rccl* r = rccl();
    if (!r->error.empty()) return tz_fail(TZ_EDEVICE, "tz_comm_unique_id: " + r->error);
    ncclUniqueId id;
    TZ_NCCL(nullptr, r->GetUniqueId(&id));
    static_assert(sizeof id == TZ_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof id);
    return TZ_OK;
}

int tz_comm_rendezvous_id(const char* directory, int rank, unsigned char* id_inout, double timeout_s) {
    if (!directory || !id_inout) return tz_fail(TZ_EINVAL, "tz_comm_rendezvous_id: null argument");
    const std::string path = std::string(directory) + "/rccl_id.bin";
    if (rank == 0) {
        int rc = tz_comm_unique_id(id_inout);
        if (rc) return rc;
        if ((rc = publish_fresh(path, id_inout, TZ_COMM_ID_BYTES))) return rc;   // whatever an earlier job left there is gone first
        g_published_id_path = path;
        return TZ_OK;
    }
    return read_fresh(path, id_inout, TZ_COMM_ID_BYTES, timeout_s);
}

int tz_comm_create_rccl(const unsigned char* id, int rank, int world, int device_id, tz_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return tz_fail(TZ_EINVAL, "tz_comm_create_rccl: bad argument");
    Rccl* r = rccl();
    if (!r->error.empty()) return tz_fail(TZ_EDEVICE, "tz_comm_create_rccl: " + r->error);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev)
        return tz_fail(TZ_EDEVICE, "tz_comm_create_rccl: no such HIP device (the RCCL transport needs the shard's GPU)");
    TZ_HIP(hipSetDevice(device_id));
    tz_comm* c = new tz_comm();
    c->rank = rank;
    c->world = world;
    c->device = device_id;
    ncclUniqueId nid;
    memcpy(&nid, id, sizeof nid);
    ncclResult_t res = r->CommInitRank(&c->comm, world, nid, rank);
    if (res != ncclSuccess) {
        delete c;
        return tz_fail(TZ_EDEVICE, std::string("ncclCommInitRank: ") + r->GetErrorString(res));
    }
    if (hipStreamCreate(&c->stream) != hipSuccess) {
        r->CommDestroy(c->comm);
        delete c;
        return tz_fail(TZ_EDEVICE, "tz_comm_create_rccl: hipStreamCreate failed");
    }
    if (rank == 0 && !g_published_id_path.empty()) {   // every rank has joined with it: a later job in this directory must not find it
        (void)unlink(g_published_id_path.c_str());
        g_published_id_path.clear();
    }
    *out = c;
    return TZ_OK;
}

int tz_comm_create_fs(const char* directory, int rank, int world, double timeout_s, tz_comm** out) {
    if (!directory || !out || world < 1 || rank < 0 || rank >= world) return tz_fail(TZ_EINVAL, "tz_comm_create_fs: bad argument");
    tz_comm* c = new tz_comm();
    c->rank = rank;
    c->world = world;
    c->fs = true;
    c->dir = directory;
    if (timeout_s > 0) c->timeout_s = timeout_s;
    if (world > 1) {   // the job's nonce: drawn by rank 0, published fresh (a leftover xch-job.bin is replaced / not accepted when stale)
        const std::string job = c->dir + "/xch-job.bin";
        int rc;
        if (rank == 0) {
            c->nonce = ((uint64_t)std::chrono::system_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)getpid() << 32);
            rc = publish_fresh(job, &c->nonce, sizeof c->nonce);
        } else {
            rc = read_fresh(job, &c->nonce, sizeof c->nonce, c->timeout_s);
        }
        if (rc) {
            delete c;
            return rc;
        }
    }
    *out = c;
    return TZ_OK;
}

int tz_comm_destroy(tz_comm* c) {
    if (!c) return TZ_OK;
    if (c->fs) {
        // a closing round: once it completes every rank has read my last payload, which the round then deletes; what stays
        // behind is this round's 8 bytes per rank (nobody can know that the others have read those) under this job's nonce,
        // which no later job in the directory uses
        if (c->world > 1 && c->seq >= 1) {
            uint64_t one = 1;
            std::vector<uint64_t> all(c->world);
            (void)all_gather_fixed(c, &one, sizeof one, reinterpret_cast<unsigned char*>(all.data()));
        }
    } else {
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->comm) rccl()->CommDestroy(c->comm);
        if (c->dev_send) (void)hipFree(c->dev_send);
        if (c->dev_recv) (void)hipFree(c->dev_recv);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
    return TZ_OK;
}

int tz_comm_info(tz_comm* c, int* rank_out, int* world_out, int* is_rccl_out, uint64_t* collectives_out, uint64_t* bytes_gathered_out) {
    if (!c) return tz_fail(TZ_EINVAL, "tz_comm_info: null handle");
    if (rank_out) *rank_out = c->rank;
    if (world_out) *world_out = c->world;
    if (is_rccl_out) *is_rccl_out = c->fs ? 0 : 1;
    if (collectives_out) *collectives_out = c->collectives;
    if (bytes_gathered_out) *bytes_gathered_out = c->bytes_gathered;
    return TZ_OK;
}

int tz_comm_all_gather(tz_comm* c, const void* data, uint64_t bytes, uint64_t* sizes_out, uint64_t* total_out) {
    if (!c || (!data && bytes)) return tz_fail(TZ_EINVAL, "tz_comm_all_gather: bad argument");
    std::vector<unsigned char> mine;
    if (bytes) mine.assign((const unsigned char*)data, (const unsigned char*)data + bytes);
    std::vector<std::vector<unsigned char>> all;
    int rc = all_gather_var(c, mine, all);
    if (rc) return rc;
    c->last.clear();
    for (int r = 0; r < c->world; r++) {
        if (sizes_out) sizes_out[r] = all[r].size();
        c->last.insert(c->last.end(), all[r].begin(), all[r].end());
    }
    if (total_out) *total_out = c->last.size();
    return TZ_OK;
}

int tz_comm_take(tz_comm* c, void* out, uint64_t out_cap) {
    if (!c || (!out && !c->last.empty())) return tz_fail(TZ_EINVAL, "tz_comm_take: bad argument");
    if (c->last.size() > out_cap) return tz_fail(TZ_EINVAL, "tz_comm_take: output buffer too small");
    if (!c->last.empty()) memcpy(out, c->last.data(), c->last.size());
    c->last.clear();
    return TZ_OK;
}

int tz_comm_broadcast(tz_comm* c, void* data, uint64_t bytes, int root) {
    if (!c || (!data && bytes) || root < 0 || root >= c->world) return tz_fail(TZ_EINVAL, "tz_comm_broadcast: bad argument");
    return broadcast_bytes(c, data, bytes, root);
}

int tz_comm_barrier(tz_comm* c) {
    if (!c) return tz_fail(TZ_EINVAL, "tz_comm_barrier: null handle");
    uint64_t one = 1;
    std::vector<uint64_t> all(c->world);
    return all_gather_fixed(c, &one, sizeof one, reinterpret_cast<unsigned char*>(all.data()));
}

int tz_selfplay_set_comm(tz_selfplay* sp, tz_comm* c, int writer_rank) {
    if (!sp) return tz_fail(TZ_EINVAL, "tz_selfplay_set_comm: null handle");
    if (!c) return tz_selfplay_set_exchange(sp, nullptr);
    if (writer_rank >= c->world) return tz_fail(TZ_EINVAL, "tz_selfplay_set_comm: writer rank outside the world");
    HostExchange x;
    x.rank = c->rank;
    x.world = c->world;
    x.writer = writer_rank;
    x.all_gather = [c](const std::vector<unsigned char>& mine, std::vector<std::vector<unsigned char>>& all) { return all_gather_var(c, mine, all); };
    return tz_selfplay_set_exchange(sp, &x);
}

}  // extern "C"
