// How fast can all CUs pull the same buffer through L2 into registers?  The access pattern of the net kernel's weight
// stream without anything else: a workgroup of 8 waves walks a buffer of 1-KB fragments, wave w takes fragments
// 16*k + 2*w and 16*k + 2*w + 1 of step k (buffer_load_b128, 64 lanes x 16 B), DEPTH steps in flight.
//   l2_stream_bench [MB per pass = 47] [workgroups = 512] [passes = 1] [depth = 4]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512, 2) void stream_kernel(const unsigned* buf, int frags, int passes, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(buf), 0, frags * 1024, 0x00020000);
    const int steps = frags / 16;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 ring[DEPTH][2];
    for (int p = 0; p < passes; p++) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            ring[d][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (d * 16 + 2 * wave) * 1024, 0);
            ring[d][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (d * 16 + 2 * wave + 1) * 1024, 0);
        }
        for (int k = 0; k < steps; k += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                acc ^= ring[d][0];
                acc ^= ring[d][1];
                const int nk = k + d + DEPTH;   // beyond the end: the buffer descriptor returns zeros
                ring[d][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (nk * 16 + 2 * wave) * 1024, 0);
                ring[d][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (nk * 16 + 2 * wave + 1) * 1024, 0);
            }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = acc[0];
}

int main(int argc, char** argv) {
    const int mb = argc > 1 ? atoi(argv[1]) : 47, wgs = argc > 2 ? atoi(argv[2]) : 512, passes = argc > 3 ? atoi(argv[3]) : 1,
              depth = argc > 4 ? atoi(argv[4]) : 4;
    const int frags = mb * 1024 / 16 * 16;
    unsigned *buf = nullptr, *out = nullptr;
    hipMalloc(&buf, (size_t)frags * 1024);
    hipMalloc(&out, 4096 * 4);
    std::vector<unsigned> h((size_t)frags * 256);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)(i * 2654435761u);
    hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int it = 0; it < 3; it++) {
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) {
            if (depth == 2) stream_kernel<2><<<wgs, 512>>>(buf, frags, passes, out);
            else if (depth == 8) stream_kernel<8><<<wgs, 512>>>(buf, frags, passes, out);
            else stream_kernel<4><<<wgs, 512>>>(buf, frags, passes, out);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        const double bytes = (double)wgs * passes * frags * 1024.0;
        printf("%d MB x %d workgroups x %d passes, depth %d: %.3f ms per launch, %.2f TB/s into registers\n", mb, wgs, passes, depth, ms, bytes / ms / 1e9);
    }
    return 0;
}
