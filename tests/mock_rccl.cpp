// mock_rccl — a stand-in for librccl for two-rank rehearsals on ONE GPU (test infrastructure).  RCCL refuses two ranks on one
// device ("Duplicate GPU detected"), so on a one-GPU box the RCCL side of csrc/tz_comm.cpp — the dlopen'd entry points, the
// unique-id rendezvous, the device staging buffers, the stream synchronisation, the count / pad logic at world > 1 — would never
// run with more than one rank.  This library has the entry points tz_comm binds, with RCCL's signatures and semantics on device
// buffers (read and written with hipMemcpy on the caller's stream order), and moves the bytes between the processes through
// files of a directory named after the unique id.  Selected with TZ_RCCL_LIB=<this .so> (tz_comm.cpp: rccl()).
//   hipcc -shared -fPIC tests/mock_rccl.cpp -o libmockrccl.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Mock {
    std::string dir;
    int rank = 0, n = 1;
    unsigned long long seq = 0;
};

size_t elem_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8:
        case ncclUint8: return 1;
        case ncclFloat16:
        case ncclBfloat16: return 2;
        case ncclInt32:
        case ncclUint32:
        case ncclFloat32: return 4;
        default: return 8;
    }
}

std::string name(const Mock* m, unsigned long long seq, int rank) { return m->dir + "/" + std::to_string(seq) + "-" + std::to_string(rank); }

bool publish(const Mock* m, const std::vector<unsigned char>& bytes) {
    const std::string path = name(m, m->seq, m->rank), tmp = path + ".part";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = bytes.empty() || fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    fclose(f);
    return ok && rename(tmp.c_str(), path.c_str()) == 0;
}

bool fetch(const Mock* m, int rank, size_t bytes, unsigned char* out) {
    const std::string path = name(m, m->seq, rank);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && (size_t)st.st_size == bytes) break;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    const size_t got = bytes ? fread(out, 1, bytes, f) : 0;
    fclose(f);
    return got == bytes;
}

void retire(Mock* m) {   // everybody has published round seq, so everybody has read round seq - 1
    if (m->seq >= 1) (void)unlink(name(m, m->seq - 1, m->rank).c_str());
    m->seq++;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f || fread(id->internal, 1, 16, f) != 16) {
        if (f) fclose(f);
        return ncclSystemError;
    }
    fclose(f);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    Mock* m = new Mock();
    char hex[40];
    for (int i = 0; i < 16; i++) snprintf(hex + 2 * i, 3, "%02x", (unsigned char)id.internal[i]);
    m->dir = std::string("/tmp/mockrccl-") + hex;
    m->rank = rank;
    m->n = nranks;
    (void)mkdir(m->dir.c_str(), 0700);
    *comm = reinterpret_cast<ncclComm_t>(m);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete reinterpret_cast<Mock*>(comm);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    Mock* m = reinterpret_cast<Mock*>(comm);
    const size_t bytes = sendcount * elem_size(datatype);
    std::vector<unsigned char> mine(bytes), all(bytes * m->n);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;   // what the stream wrote into sendbuff is there
    if (bytes && hipMemcpy(mine.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!publish(m, mine)) return ncclSystemError;
    for (int r = 0; r < m->n; r++)
        if (!fetch(m, r, bytes, all.data() + (size_t)r * bytes)) return ncclSystemError;
    if (bytes && hipMemcpy(recvbuff, all.data(), all.size(), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    retire(m);
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream) {
    Mock* m = reinterpret_cast<Mock*>(comm);
    const size_t bytes = count * elem_size(datatype);
    std::vector<unsigned char> data(bytes);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (m->rank == root) {
        if (bytes && hipMemcpy(data.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (!publish(m, data)) return ncclSystemError;
    } else {
        if (!publish(m, {})) return ncclSystemError;   // an empty marker: the round's arrival
    }
    for (int r = 0; r < m->n; r++) {
        std::vector<unsigned char> got(r == root ? bytes : 0);
        if (!fetch(m, r, got.size(), got.data())) return ncclSystemError;
        if (r == root) data = got;
    }
    if (bytes && hipMemcpy(recvbuff, data.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    retire(m);
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) { return ncclInvalidUsage; }   // bound, never called

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclSystemError ? "mock: file exchange failed" : "mock: error"; }

}  // extern "C"
