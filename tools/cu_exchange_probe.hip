// cu_exchange_probe.hip — what a per-layer exchange of activations between the CUs of one board group would cost (VERDICT r2 #6:
// "split a board group's 256 output channels over 2-4 CUs with a per-layer exchange through L2").  C workgroups form a group; per
// round every member stores its slice of the layer's output (SLICE bytes), releases it at agent scope, arrives at the group's
// counter, waits for the other members, and loads their slices — 41 rounds, as the 41 convolutions of net5.  Members of a group
// are either neighbours in the launch order (which the dispatcher spreads over the 8 XCDs: the exchange crosses L2s) or 8 apart
// (same XCD, one L2).  Prints microseconds per round; the net kernel's own work is not in it.
//   hipcc -O3 --offload-arch=gfx950 tools/cu_exchange_probe.hip -o /tmp/cu_exchange_probe && /tmp/cu_exchange_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                     \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

// RAW: the members share an XCD and meet in its L2 by hand — plain stores (the L1 writes through), s_waitcnt, an atomic that the L2
// executes, loads that bypass the L1 (sc0 sc1): no L2 write-back, no invalidate.  Outside the HIP memory model (there is no "XCD" scope);
// valid only because blocks b, b + 8, ... are dispatched to one XCD.  !RAW: agent-scope release / acquire as the language defines them
// (buffer_wbl2 sc1 + buffer_inv sc1 on gfx950: the 8 L2s are not coherent with each other).
template <int C, int RAW>
__global__ __launch_bounds__(256) void exchange_kernel(uint4* slices, unsigned* counters, int slice16, int rounds, int same_xcd,
                                                       unsigned long long* cycles, unsigned* sink) {
    const int b = blockIdx.x;
    int group, member;
    if (same_xcd) {   // blocks b, b + 8, b + 16 .. land on one XCD (round-robin dispatch over 8 XCDs)
        const int span = 8 * C;
        group = (b / span) * 8 + b % 8;
        member = (b % span) / 8;
    } else {
        group = b / C;
        member = b % C;
    }
    unsigned* counter = counters + 32 * group;   // one 128-byte line per group
    unsigned acc = 0;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; r++) {
        uint4* mine = slices + ((size_t)(group * 2 + (r & 1)) * C + member) * slice16;
        if constexpr (RAW != 2)
            for (int i = threadIdx.x; i < slice16; i += 256) mine[i] = make_uint4(r, member, i, acc);
        if constexpr (RAW == 2) {
            // RAW == 2: as RAW == 1, but every access carries sc0 sc1 (system scope: stores write through the L2, loads and the atomic go
            // past it), so that the members may sit on different XCDs: no dependence on where the dispatcher puts a block
            const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, slice16 * 16, 0x00020000);
            for (int i = threadIdx.x; i < slice16; i += 256) {
                typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t v = {(unsigned)r, (unsigned)member, (unsigned)i, acc};
                __builtin_amdgcn_raw_buffer_store_b128(v, mrs, i * 16, 0, 17);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(counter, 0, 128, 0x00020000);
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned want = (unsigned)C * (r + 1);
                while ((unsigned)__builtin_amdgcn_raw_buffer_load_b32(crs, 0, 0, 17) < want) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            for (int m = 1; m < C; m++) {
                uint4* theirs = slices + ((size_t)(group * 2 + (r & 1)) * C + (member + m) % C) * slice16;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(theirs, 0, slice16 * 16, 0x00020000);
                for (int i = threadIdx.x; i < slice16; i += 256) {
                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                    const u32x4_t v = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 17));
                    acc += v.x + v.z;
                    if (v.x != (unsigned)r) acc += 1u << 30;
                }
            }
        } else if constexpr (RAW == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have been acknowledged by the L2
            __syncthreads();
            const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(counter, 0, 128, 0x00020000);
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned want = (unsigned)C * (r + 1);
                while ((unsigned)__builtin_amdgcn_raw_buffer_load_b32(crs, 0, 0, 17) < want) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            for (int m = 1; m < C; m++) {
                uint4* theirs = slices + ((size_t)(group * 2 + (r & 1)) * C + (member + m) % C) * slice16;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(theirs, 0, slice16 * 16, 0x00020000);
                for (int i = threadIdx.x; i < slice16; i += 256) {
                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                    const u32x4_t v = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 17));
                    acc += v.x + v.z;
                    if (v.x != (unsigned)r) acc += 1u << 30;
                }
            }
        } else {
        __threadfence();          // release at agent scope: the slice is visible to the other CUs before the arrival is
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)C * (r + 1);
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        __threadfence();          // acquire: the loads below are not served from a stale L1 line
        for (int m = 1; m < C; m++) {
            const uint4* theirs = slices + ((size_t)(group * 2 + (r & 1)) * C + (member + m) % C) * slice16;
            for (int i = threadIdx.x; i < slice16; i += 256) {
                typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(theirs + i));
                acc += v.x + v.z;
                if (v.x != (unsigned)r) acc += 1u << 30;   // a stale slice shows in the sink
            }
        }
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) cycles[b] = t1 - t0;
    atomicAdd(sink, acc >> 30);
}

template <int C, int RAW>
int run(int groups, int slice_bytes, int same_xcd) {
    const int rounds = 41, blocks = groups * C, slice16 = slice_bytes / 16;
    uint4* slices;
    unsigned *counters, *sink;
    unsigned long long* cycles;
    CHECK(hipMalloc(&slices, (size_t)groups * 2 * C * slice_bytes));
    CHECK(hipMalloc(&counters, (size_t)groups * 128));
    CHECK(hipMalloc(&cycles, blocks * sizeof(unsigned long long)));
    CHECK(hipMalloc(&sink, 4));
    double best = 1e30, stale = 0;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipMemset(counters, 0, (size_t)groups * 128));
        CHECK(hipMemset(sink, 0, 4));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        exchange_kernel<C, RAW><<<blocks, 256>>>(slices, counters, slice16, rounds, same_xcd, cycles, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned s = 0;
        CHECK(hipMemcpy(&s, sink, 4, hipMemcpyDeviceToHost));
        stale += s;
        if (rep && ms < best) best = ms;
    }
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cycles, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    for (auto c : h) mx = c > mx ? c : mx;
    printf("%d CUs per group, %3d groups (%3d workgroups), slice %5d B, %-22s: kernel %.1f us = %.2f us per round (slowest workgroup %.2f us per round at 100 MHz ticks), stale reads %g\n",
           C, groups, blocks, slice_bytes, RAW == 2 ? (same_xcd ? "one XCD, sc0 sc1 accesses" : "C XCDs, sc0 sc1 accesses") : RAW ? "one XCD, by hand in L2" : same_xcd ? "one XCD, agent scope" : "C XCDs, agent scope", best * 1e3, best * 1e3 / rounds,
           (double)mx / 100.0 / rounds, stale);
    (void)hipFree(slices);
    (void)hipFree(counters);
    (void)hipFree(cycles);
    (void)hipFree(sink);
    return 0;
}

int main() {
    // batch 128 on one- or two-board groups: 128 or 64 groups; a slice = a member's share of 25 or 50 pixels x 256 channels of fp16
    if (run<2, 1>(128, 6400, 1) || run<2, 1>(64, 12800, 1) || run<4, 1>(64, 6400, 1) || run<4, 1>(32, 12800, 1)) return 1;
    for (int same = 1; same >= 0; same--)
        if (run<2, 2>(128, 6400, same) || run<4, 2>(64, 6400, same) || run<4, 2>(32, 12800, same)) return 1;
    for (int same = 1; same >= 0; same--) {
        if (run<2, 0>(128, 6400, same)) return 1;
        if (run<2, 0>(64, 12800, same)) return 1;
        if (run<4, 0>(64, 6400, same)) return 1;
        if (run<4, 0>(32, 12800, same)) return 1;
    }
    return 0;
}
