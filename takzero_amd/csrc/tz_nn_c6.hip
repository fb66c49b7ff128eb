// tz_nn_c6.hip — third translation unit of the network kernels: TZ_PREC_F16C6, the fused trunk + heads launch with fp16 products
// and FP6 (E2M3) block-scaled correction products.  Replaces, like net_mfma_kernel, the LibTorch op sequence of
// takzero/src/network/net5.rs:184-285, net6_simhash.rs:194-324, residual.rs:13-63 and game_repr (repr.rs:169-244); the reference
// computes it in fp32 (net5.rs:184-191,237) and the north star asks for logits within 1e-3 of that.
//
// Arithmetic.  A weight w and an activation x are carried as hi + lo: hi = fp16(.), lo = what fp16 drops (about 2^-12 of the
// value).  w*x = wh*xh + (wl*xh + wh*xl) + wl*xl: the first product runs on v_mfma_f32_16x16x32_f16 exactly as in TZ_PREC_F16; the
// correction only has to be right to a few bits, and runs on E2M3 copies of all four operands through
// v_mfma_scale_f32_16x16x128_f8f6f4 - 128 input channels per instruction in the 16 cycles the fp16 form takes for 32
// (tools/mfma_f6_probe.hip, profiles/r03_mfma_f6_probe.txt) - so 256 channels of a tap cost 8 + 4 MFMAs = 192 cycles per
// (row tile, 16 outputs) where TZ_PREC_F16C8's E4M3 corrections cost 256 and the split form 384.  E2M3 has E4M3's three mantissa
// bits but only two exponent bits; the range comes from the instruction's block scales: one E8M0 byte per lane, i.e. per (row,
// block of 32 input channels).  The products are scaled by the hardware, so the corrections add straight into the main fp32
// accumulator: no second accumulator set, no per-layer merge.  CPU study: tools/fp6_correction_study.py (1.35e-4 at trained logit
// scale against 1.22e-4 for E4M3 and 3.3e-3 for fp16 alone).
//
// What makes 8 boards per workgroup fit (TZ_PREC_F16C8 holds 4: its weight stream and fixed costs are per workgroup):
//   * xh's E2M3 copy is not stored: v_cvt_scalef32_pk32_fp6_f16 makes it in the k-loop from the four fp16 fragments of the 128
//     channels a lane has just read for the main product (one 16-cycle vector instruction per 12 MFMAs);
//   * a block of 32 is therefore "the 8 channels of lane group q in each of 4 consecutive k-chunks".  The image stores the 32
//     output channels of wave w in exactly those places (planes 4 (w / 4) .. + 3, piece w % 4), so a block is one wave's outputs
//     of one pixel and its scale is found in that wave's epilogue; the weights' input channels are permuted to match on the host;
//   * xl lives in three more planes (24 B per block: 16 + 8) and the two scale bytes of a block in a fourth of a plane;
//   * the residual connection does not read the image back: a block's input (fp32) waits in a global scratch buffer, written by
//     the epilogue that produced it and read by the epilogue that starts the block's second conv (`seeds`, one 16-B store and one
//     load per output tile and lane, coalesced; the image alone would carry it to 15 bits, which doubles the logit error).
// Image: 8 fp16 planes + 3 xl planes + scales = 11.5 planes; 5x5 with 8 boards: 158.9 KB + head scratch + compact tap table.
#define TZ_NN_SPLIT_TU 1
#define TZ_NN_C6_TU 1
#include "tz_nn.hip"

namespace {

typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// Cache policy of the seed stores and loads: streamed once each way, a conv apart, by the same lane.  The loads are sc0 sc1
// (served by L2, never by the CU's L1), the stores nt.
#ifndef TZ_C6_SEED_ST
#define TZ_C6_SEED_ST 2    // nt
#endif
#ifndef TZ_C6_SEED_LD
#define TZ_C6_SEED_LD 17   // sc0 sc1
#endif
constexpr float C6_LO = 2048.0f;   // the lo part is kept times 2^11 while it is a half: it stays a normal number wherever hi is

template <int NB, int P, bool PERM>
struct C6Geo {
    typedef RowMap<NB, P, PERM> RM;
    static constexpr int RT = RM::RT;
    static constexpr bool TT = PERM && P >= 8 && 16 % P == 0;           // compact tap table (whole 8-row runs per square)
    // The eight zero rows an off-board tap reads.  With the compact row map (5x5, 8 boards: 25 squares in 26 slots) the upper half
    // of the last row tile is padding: those rows are never written and serve as the zero rows, which is what lets twelve planes fit.
    static constexpr bool ZPAD = TT && RM::square_at(RT * RM::PPT - 1) < 0 && P == 8;
    static constexpr int ZROW = ZPAD ? RT * 16 - 8 : RT * 16, LROWS = ZPAD ? RT * 16 : RT * 16 + 8, PLANE = LROWS * LDS_ROWB;
    // Twelve planes of [row][4 pieces of 16 B] with the fp16 planes' piece rotation, so that one fragment address (row, piece) serves
    // them all, six per K-half G:  6 G + 0..3 the fp16 planes of the half's four k-chunks; 6 G + 4 ("XB"): the last 8 B of a block's
    // 24-B xl string, the block's scale dword (byte 0: scale of xh's copy, byte 1: scale of xl) and that first scale once more as the
    // fp32 number the conversion takes; 6 G + 5 ("XA"): the first 16 B of the xl string.  Everything a pass-A step reads is within
    // 64 KB of the half's first plane, everything a pass-B step reads within 64 KB of XB: one address add per step, the rest
    // immediates.  A whole dword for the scales: two 16-bit stores of two waves into one LDS dword lose one of them now and then
    // (measured, tools/c6_debug.py).
    static constexpr int NPL = 12;
    __host__ __device__ static constexpr int hi_plane(int kc) { return 6 * (kc >> 2) + (kc & 3); }   // k-chunk kc of the 256 channels
    __host__ __device__ static constexpr int xb_plane(int G) { return 6 * G + 4; }
    __host__ __device__ static constexpr int xa_plane(int G) { return 6 * G + 5; }
    static constexpr int HS = NPL * PLANE;                               // head scratch: float[2][RT * 16]
    static constexpr int TAB = HS + 2 * RT * 16 * 4;
    static constexpr int TAB_BYTES = TT ? 9 * RT * RM::PPT * 4 : 9 * RT * 64 * 2;
    static constexpr int SMEM = TAB + TAB_BYTES;
    static constexpr int SEED_BYTES = RT * 16 * 1024;                    // per workgroup: [rt][wave][j] tiles of 64 lanes x 16 B
    static_assert(PLANE < 65536 && 4 * PLANE + 16 < 65536, "16-bit tap table entries, plane offsets as ds_read immediates");
    static_assert(SMEM <= 160 * 1024, "TZ_PREC_F16C6: the image does not fit a CU");
};

// an opaque copy: what is computed from it is computed where it is used, not hoisted out of the loops as one more live register
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// E8M0 byte of the E2M3 block scale for a block whose largest magnitude is amax (tz_fp6.h: tz_e2m3_block_scale_byte)
__device__ __forceinline__ unsigned c6_scale_byte(float amax, unsigned min_byte) {
    const unsigned b = (__float_as_uint(amax * (16.0f / 15.0f)) >> 23) & 0xffu;
    return max(b, min_byte + 2u) - 2u;
}

__device__ __forceinline__ f32x4 mfma_f6(const i32x4& a0, const i32x2& a1, const i32x4& b0, const i32x2& b1, f32x4 c, int sa, int sb, bool second) {
    const i32x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], 0, 0};
    const i32x8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], 0, 0};
    // FMT 2 = E2M3 on both sides; op_sel picks the scale byte: 0 for the (wl, xh) term, 1 for the (wh, xl) term
    return second ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 1, sa, 1, sb)
                  : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, sa, 0, sb);
}

// One 3x3 conv over 256 input channels out of the image: for every tap and both K-halves G, the row tiles the tap does not skip.  A
// step is a row tile's 6 RNX MFMAs into the one accumulator: wl x xh and wh x xl (FP6-scaled), then the fp16 chunks 0..3 - the same order
// in every workgroup form; the two FP6 products sit together because every change of MFMA kind inside an accumulator's chain costs
// issue time (measured: profiles/r03_c6_kloop_ab.txt).  All of a (tap, G) group's weights are stationary together (4 RNX fp16 fragments, 2 RNX E2M3
// fragments, RNX scale dwords); the next group's replace them one by one as the group's last tile lets go of them.  A step reads the
// next step's seven operands from LDS (4 fp16 fragments, the xl string in two pieces, the scale pair) while it computes - the reads
// are pinned ahead of its MFMAs, left alone the scheduler sinks them to save registers - and converts the next step's xh copy
// between its own MFMAs.  The lane's fragment address of a (tap, tile) is one table entry from LDS (read three steps ahead) plus a
// lane constant per plane group; every plane is an immediate offset from there.
// (A form with two passes per group - the wh x xl products in a pass of their own, during which the next group's weights arrive - gave
// the same bits at the same speed with twice the code: profiles/r03_c6_kloop_ab.txt.)
template <int NB, int P, bool PERM>
struct C6Sched {
    typedef RowMap<NB, P, PERM> RM;
    static constexpr int RT = RM::RT;
    struct Item {
        int tap, G, rt, idx, na, valid;
    };
    static constexpr int count() {
        int n = 0;
        for (int t = 0; t < 9; t++) n += 2 * __builtin_popcount(RM::tap_tile_mask(t));
        return n;
    }
    static constexpr Item at(int n) {
        if (n < 0) return Item{0, 0, 0, 0, 0, 0};
        for (int tap = 0; tap < 9; tap++) {
            const unsigned m = RM::tap_tile_mask(tap);
            const int na = __builtin_popcount(m);
            for (int G = 0; G < 2; G++) {
                if (n < na) {
                    int seen = 0;
                    for (int rt = 0; rt < RT; rt++)
                        if ((m >> rt) & 1) {
                            if (seen == n) return Item{tap, G, rt, n, na, 1};
                            seen++;
                        }
                }
                n -= na;
            }
        }
        return Item{0, 0, 0, 0, 0, 0};
    }
};

template <int NB, int P, bool PERM, int RNX, typename TAB, typename WF, typename W6A, typename W6B, typename WSF>
__device__ __forceinline__ void k_loop_c6(const unsigned char* lds, int lc0, f32x4 (&acc)[RowMap<NB, P, PERM>::RT][RNX], TAB tab, WF wf, W6A w6a, W6B w6b, WSF wsf) {
    typedef C6Geo<NB, P, PERM> GEO;
    typedef C6Sched<NB, P, PERM> S;
    constexpr int N = S::count(), PLANE = GEO::PLANE;
    const int lca[2] = {lc0, lc0 + 6 * PLANE}, lcb[2] = {lc0 + 4 * PLANE, lc0 + 10 * PLANE};
    f16x8 Wf[4][RNX];
    i32x4 W6a[2][RNX];
    i32x2 W6b[2][RNX];
    int Ws[RNX];
    f16x8 ah[2][4];
    i32x2 sca[2];
    i32x4 xa[2];
    i32x2 xb[2];
    u32x6 x6[2];
    int tq[4];
    auto load_ops = [&](auto m_c) {
        constexpr int m = decltype(m_c)::value;
        constexpr auto it = S::at(m);
        constexpr int slot = m & 1;
        const int t = tq[m % 4];
        const int a0 = t + lca[it.G], ab = t + lcb[it.G];
#pragma unroll
        for (int c = 0; c < 4; c++) ah[slot][c] = *reinterpret_cast<const f16x8*>(lds + a0 + c * PLANE);
        sca[slot] = *reinterpret_cast<const i32x2*>(lds + ab + 8);
        xa[slot] = *reinterpret_cast<const i32x4*>(lds + ab + PLANE);
        xb[slot] = *reinterpret_cast<const i32x2*>(lds + ab);
    };
    auto convert = [&](int sl) {
        f16x32 src;
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int i = 0; i < 8; i++) src[8 * c + i] = ah[sl][c][i];
        x6[sl] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(src, __int_as_float(sca[sl][1]));
    };
    {
        constexpr auto i0 = S::at(0);
#pragma unroll
        for (int j = 0; j < RNX; j++) {
#pragma unroll
            for (int c = 0; c < 4; c++) Wf[c][j] = wf(i0.tap, 4 * i0.G + c, j);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                W6a[t][j] = w6a(i0.tap, i0.G, j, t);
                W6b[t][j] = w6b(i0.tap, i0.G, j, t);
            }
            Ws[j] = wsf(i0.tap, i0.G, j);
        }
        tq[0] = tab(i0.tap, i0.rt);
        if constexpr (N > 1) tq[1] = tab(S::at(1).tap, S::at(1).rt);
        if constexpr (N > 2) tq[2] = tab(S::at(2).tap, S::at(2).rt);
        load_ops(IntC<0>{});
        convert(0);
    }
    auto item = [&](auto n_c) {
        constexpr int n = decltype(n_c)::value;
        constexpr auto it = S::at(n);
        constexpr int slot = n & 1;
        constexpr bool last = it.idx == it.na - 1, more = n + 1 < N;
        constexpr auto ng = S::at(more ? n + 1 : -1);   // after the last tile of a group: the first of the next
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + 3 < N) tq[(n + 3) % 4] = tab(S::at(n + 3).tap, S::at(n + 3).rt);
        if constexpr (more) load_ops(IntC<(more ? n + 1 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        const i32x4 xh0 = {(int)x6[slot][0], (int)x6[slot][1], (int)x6[slot][2], (int)x6[slot][3]};
        const i32x2 xh1 = {(int)x6[slot][4], (int)x6[slot][5]};
#ifndef TZ_C6_ABL
#define TZ_C6_ABL 0   // development builds: 1 = no (wl, xh) products and no conversion, 2 = no (wh, xl) products
#endif
        if constexpr (!(TZ_C6_ABL & 1)) {
#pragma unroll
            for (int j = 0; j < RNX; j++) acc[it.rt][j] = mfma_f6(W6a[0][j], W6b[0][j], xh0, xh1, acc[it.rt][j], Ws[j], sca[slot][0], false);
        }
        if constexpr (last && more) {
#pragma unroll
            for (int j = 0; j < RNX; j++) {
                W6a[0][j] = w6a(ng.tap, ng.G, j, 0);
                W6b[0][j] = w6b(ng.tap, ng.G, j, 0);
            }
        }
        int wsn[RNX];
        if constexpr (last && more) {   // the scale dwords of the next group: requested before this group's last use of its own
#pragma unroll
            for (int j = 0; j < RNX; j++) wsn[j] = wsf(ng.tap, ng.G, j);
        }
        if constexpr (!(TZ_C6_ABL & 2)) {
#pragma unroll
            for (int j = 0; j < RNX; j++) acc[it.rt][j] = mfma_f6(W6a[1][j], W6b[1][j], xa[slot], xb[slot], acc[it.rt][j], Ws[j], sca[slot][0], true);
        }
        if constexpr (last && more) {
#pragma unroll
            for (int j = 0; j < RNX; j++) {
                W6a[1][j] = w6a(ng.tap, ng.G, j, 1);
                W6b[1][j] = w6b(ng.tap, ng.G, j, 1);
                Ws[j] = wsn[j];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int j = 0; j < RNX; j++) acc[it.rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[c][j], ah[slot][c], acc[it.rt][j], 0, 0, 0);
            if constexpr (last && more) {
#pragma unroll
                for (int j = 0; j < RNX; j++) Wf[c][j] = wf(ng.tap, 4 * ng.G + c, j);
            }
            if constexpr (more) {
                if (c == 1 && !(TZ_C6_ABL & 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    convert(slot ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    auto run = [&](auto self, auto n_c) -> void {
        constexpr int n = decltype(n_c)::value;
        if constexpr (n < N) {
            item(n_c);
            self(self, IntC<n + 1>{});
        }
    };
    run(run, IntC<0>{});
}

// 4x4 exchange between the register index (four row tiles) and the lane group: after it, r[s] of lane group Q holds what r[Q] of
// lane group s held (v_permlane32_swap / v_permlane16_swap, tools/mfma_f6_probe.hip)
__device__ __forceinline__ void quad_transpose(unsigned (&r)[4]) {
    u32x2 t;
    t = __builtin_amdgcn_permlane32_swap(r[0], r[2], false, false);
    r[0] = t[0];
    r[2] = t[1];
    t = __builtin_amdgcn_permlane32_swap(r[1], r[3], false, false);
    r[1] = t[0];
    r[3] = t[1];
    t = __builtin_amdgcn_permlane16_swap(r[0], r[1], false, false);
    r[0] = t[0];
    r[1] = t[1];
    t = __builtin_amdgcn_permlane16_swap(r[2], r[3], false, false);
    r[2] = t[0];
    r[3] = t[1];
}

template <int NB, int P, int RNP, bool PERM>
__global__ __launch_bounds__(512, 2) void net_c6_kernel(NetArgs a) {
    typedef _Float16 ET;
    typedef f16x8 ex8;
    typedef f16x4 ex4;
    typedef C6Geo<NB, P, PERM> GEO;
    typedef RowMap<NB, P, PERM> RM;
    constexpr int RN = 2, NW = 8, TAPS = 9, NT = 512, LAYOUT = 1;
    constexpr int NN = NB * NB, ROWS = P * NN, RT = GEO::RT, LROWS = GEO::LROWS, ZROW = GEO::ZROW, PLANE = GEO::PLANE, NPL = GEO::NPL;
    constexpr bool TT = GEO::TT;
    constexpr int LAYER_FRAGS = TAPS * 8 * 16, REC6 = 3328, LAYER_REC6 = TAPS * 2 * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* hscratch = reinterpret_cast<float*>(lds + GEO::HS);
    const int count = a.count_dev ? *a.count_dev : a.count_host;
    const int pos0 = blockIdx.x * P;
    if (pos0 >= count) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, lr = lane & 15;
    const int valid_boards = min(P, count - pos0);
    const size_t m0 = (size_t)pos0 * NN;
    const int ct0 = wave * RN;
    const int lane16 = lane * 16;
    // the workgroup's part of `seeds`: [rt][wave][j] tiles of 64 lanes x 16 B, addressed through a buffer descriptor (the tile is a
    // scalar offset, the lane one VGPR: per-tile 64-bit pointers would be hoisted out of the layer loop and cost 52 registers)
    const __amdgpu_buffer_rsrc_t seed_rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.seeds) + (size_t)blockIdx.x * GEO::SEED_BYTES, 0,
                                                                             GEO::SEED_BYTES, 0x00020000);
    auto seed_off = [&](int rt, int j) -> int { return ((rt * NW + wave) * RN + j) * 1024; };

    // ---- packed states into LDS (plane 7 is free until the first conv's epilogue), as in net_mfma_kernel
    static_assert(sizeof(tz_state) % 4 == 0 && P * sizeof(tz_state) <= (size_t)RT * 16 * LDS_ROWB, "state staging fits a plane");
    {
        constexpr int DW = sizeof(tz_state) / 4;
        uint32_t* stage = reinterpret_cast<uint32_t*>(lds + 7 * PLANE);
        for (int i = tid; i < valid_boards * DW; i += NT) {
            const int b = i / DW, d = i - b * DW, pos = pos0 + b;
            stage[i] = reinterpret_cast<const uint32_t*>(a.states + (a.game_index ? a.game_index[pos] : pos))[d];
        }
        __syncthreads();
    }
    const tz_state* staged = reinterpret_cast<const tz_state*>(lds + 7 * PLANE);
    // ---- game_repr (repr.rs:169-228) into planes 0..kc_in-1 (hi) and 8.. (lo halves, times 2^11): the first conv runs the split form
    for (int row = tid; row < LROWS; row += NT) {
        int board = 0, px = -1;
        if (row < RT * 16) RM::decode(row, board, px);
        const bool ok = px >= 0 && board < valid_boards;
        const tz_state* s = nullptr;
        int fd = 0;
        if (ok) {
            s = staged + board;
            fd = state_flat_diff<NB>(s);
        }
        float ssq = 0.0f;
        for (int c8 = 0; c8 < a.kc_in * 4; c8++) {
            ex8 v, vl;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int c = c8 * 8 + k;
                const float pv = (ok && c < a.cin_real) ? plane_value<NB>(s, px, c, fd) : 0.0f;
                ssq += pv * pv;
                v[k] = (ET)pv;
                vl[k] = (ET)((pv - (float)v[k]) * SPLIT_SCALE);
            }
            *reinterpret_cast<ex8*>(lds + LdsImg<LAYOUT>::store_addr(row, c8, PLANE)) = v;
            *reinterpret_cast<ex8*>(lds + 8 * PLANE + LdsImg<LAYOUT>::store_addr(row, c8, PLANE)) = vl;
        }
        if (row < RT * 16) hscratch[row] = ssq;
    }
    // the zero rows of the twelve planes (an off-board tap reads zeros, scale byte 0 included)
    for (int i = tid; i < NPL * 8 * 4; i += NT) {
        const int plane = i >> 5, zr = (i >> 2) & 7, pc = i & 3;
        *reinterpret_cast<uint4*>(lds + plane * PLANE + (ZROW + zr) * LDS_ROWB + pc * 16) = make_uint4(0, 0, 0, 0);
    }
    // ---- tap table
    const int tslot = TT ? lr / P : 0;
    const int lane_const = TT ? (lr % P) * LDS_ROWB + lds_piece(lr % P, q) : 0;
    int* tap_c = reinterpret_cast<int*>(lds + GEO::TAB);
    uint16_t* tap_l = reinterpret_cast<uint16_t*>(lds + GEO::TAB);
    if constexpr (TT) {
        for (int i = tid; i < TAPS * RT * RM::PPT; i += NT) {
            const int tap = i / (RT * RM::PPT), rt = (i / RM::PPT) % RT, sl = i % RM::PPT;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int sq = RM::square_at(rt * RM::PPT + sl);
            const int y = sq / NB + dy, x = sq % NB + dx;
            const bool ok = sq >= 0 && y >= 0 && y < NB && x >= 0 && x < NB;
            tap_c[i] = (ok ? RM::row_of(0, y * NB + x) : ZROW) * LDS_ROWB;
        }
    } else {
        for (int tap = wave; tap < TAPS; tap += NW) {
            int tb[RT];
            if constexpr (PERM) tap_bases_map<NB, P, true, LAYOUT>(tap, lr, q, ZROW, tb);
            else tap_bases_rc<NB, RT, TAPS, LAYOUT>(tap, lr, q, ROWS, ZROW, tb);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) tap_l[(tap * RT + rt) * 64 + lane] = (uint16_t)tb[rt];
        }
    }
    // the k-loop's view of the same table: one entry per (tap, tile), the lane's constant part apart (zero with the per-lane table)
    auto tab = [&](int tap, int rt) -> int {
        if constexpr (TT) return tap_c[(tap * RT + rt) * RM::PPT + tslot];
        else return (int)tap_l[(tap * RT + rt) * 64 + lane];
    };
    auto ta = [&](int tap, int rt) -> int {
        if constexpr (TT) return tap_c[(tap * RT + rt) * RM::PPT + tslot] + lane_const;
        else return (int)tap_l[(tap * RT + rt) * 64 + lane];
    };
    f32x4 acc[RT][RN];
    // The epilogue of a layer: ReLU, the image parts of the new activations, and the hand-over through `seeds`.
    //   store: this output (plus store_bias, the second bias of the block it is the input of) waits in `seeds` - or, after the last
    //          layer, is what the value / UBE heads read
    //   load_seed: the next conv is a block's second - its accumulator starts from what was left there
    auto epilogue = [&](bool store, const float* store_bias, bool load_seed) {
        // The lane's places in the image, recomputed here from an opaque copy of the lane number: kept as loop invariants they
        // would sit in registers through every k-loop (which has none to spare).
        int l_ = lane;
        asm volatile("" : "+v"(l_));
        const int q = l_ >> 4, lr = l_ & 15, lane16 = l_ * 16;
        // where the lane's four channels 16 (2 wave + j) + 4 q .. of an output tile sit in the fp16 planes (o = 16 j + 4 q + k)
        int obase[RN];
#pragma unroll
        for (int j = 0; j < RN; j++) obase[j] = GEO::hi_plane(4 * (wave >> 2) + 2 * j + (q >> 1)) * PLANE + lr * LDS_ROWB + lds_piece(lr, wave & 3) + (q & 1) * 8;
        // after the quad exchange the lane owns pixel lr of row tile (quad * 4 + q): where that pixel's xl string and scales go
        const int xb_base = GEO::xb_plane(wave >> 2) * PLANE + lr * LDS_ROWB + lds_piece(lr, wave & 3);
        const int xa_base = xb_base + PLANE;
        // compact row map: the upper half of the last row tile is padding and doubles as the zero rows - it is never written
        const bool pad_lane = GEO::ZPAD && lr >= 8;
        f32x4 sb[RN];
#pragma unroll
        for (int j = 0; j < RN; j++) sb[j] = store_bias ? *reinterpret_cast<const f32x4*>(store_bias + (ct0 + j) * 16 + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int quad = 0; quad < (RT + 3) / 4; quad++) {
            unsigned lp[RN][2][4];   // [j][pair][tile of the quad]: packed halves of the lo parts
            unsigned mh[4], ml[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int rt = quad * 4 + t;
                float amax_h = 0.0f, amax_l = 0.0f;
#pragma unroll
                for (int j = 0; j < RN; j++) {
                    if (rt < RT) {
                        f32x4 v;
                        ex4 hi;
                        float lo[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            v[k] = __builtin_amdgcn_fmed3f(acc[rt][j][k], 0.0f, 65504.0f);
                            hi[k] = (ET)v[k];
                            lo[k] = (v[k] - (float)hi[k]) * C6_LO;
                        }
                        amax_h = fmaxf(fmaxf(amax_h, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
                        amax_l = fmaxf(fmaxf(amax_l, fmaxf(fabsf(lo[0]), fabsf(lo[1]))), fmaxf(fabsf(lo[2]), fabsf(lo[3])));
                        if (!(rt == RT - 1 && pad_lane)) *reinterpret_cast<ex4*>(lds + obase[j] + rt * 16 * LDS_ROWB) = hi;
                        lp[j][0][t] = __builtin_bit_cast(unsigned, f16x2{(ET)lo[0], (ET)lo[1]});
                        lp[j][1][t] = __builtin_bit_cast(unsigned, f16x2{(ET)lo[2], (ET)lo[3]});
                        // aux 2 = nt: streamed once each way, no reuse
                        if (store) {
                            // The data registers of a 16-byte buffer store with a scalar offset are read well after the instruction has
                            // issued: a vector instruction that rewrites them three instructions later (what hipcc schedules when
                            // registers are this scarce) corrupts the upper half of every 8-lane group (measured, tools/c6_debug.py:
                            // 100 % of launches wrong without the pad, 0 % with it).  They stay untouched for 16 cycles.
                            u32x4 sd = __builtin_bit_cast(u32x4, v + sb[j]);
                            __builtin_amdgcn_raw_buffer_store_b128(sd, seed_rs, lane16, seed_off(rt, j), TZ_C6_SEED_ST);
                            asm volatile("s_nop 15" : "+v"(sd));
                        }
                        if (load_seed) acc[rt][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(seed_rs, lane16, seed_off(rt, j), TZ_C6_SEED_LD));
                    } else {
                        lp[j][0][t] = 0u;
                        lp[j][1][t] = 0u;
                    }
                }
                mh[t] = __float_as_uint(amax_h);
                ml[t] = __float_as_uint(amax_l);
            }
#pragma unroll
            for (int j = 0; j < RN; j++) {
                quad_transpose(lp[j][0]);
                quad_transpose(lp[j][1]);
            }
            quad_transpose(mh);
            quad_transpose(ml);
            // the lane now holds pixel lr of row tile quad * 4 + q: slot s = the channels lane group s computed, 16 j + 4 s + 2 pair ..
            const float ah_ = fmaxf(fmaxf(__uint_as_float(mh[0]), __uint_as_float(mh[1])), fmaxf(__uint_as_float(mh[2]), __uint_as_float(mh[3])));
            const float al_ = fmaxf(fmaxf(__uint_as_float(ml[0]), __uint_as_float(ml[1])), fmaxf(__uint_as_float(ml[2]), __uint_as_float(ml[3])));
            const unsigned eh = c6_scale_byte(ah_, 1u), el = c6_scale_byte(al_, 12u);
            unsigned srcw[16];
#pragma unroll
            for (int j = 0; j < RN; j++)
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    srcw[(j * 4 + s) * 2 + 0] = lp[j][0][s];
                    srcw[(j * 4 + s) * 2 + 1] = lp[j][1][s];
                }
            typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
            u32x16 sv;
#pragma unroll
            for (int i = 0; i < 16; i++) sv[i] = srcw[i];
            const u32x6 x6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, sv), __uint_as_float(el << 23));
            const int rt_mine = quad * 4 + q;
            if (rt_mine < RT && !(rt_mine == RT - 1 && pad_lane)) {
                const int ro = rt_mine * 16 * LDS_ROWB;
                *reinterpret_cast<u32x4*>(lds + xa_base + ro) = u32x4{x6[0], x6[1], x6[2], x6[3]};
                *reinterpret_cast<u32x4*>(lds + xb_base + ro) = u32x4{x6[4], x6[5], eh | ((el - 11u) << 8), eh << 23};
            }
        }
    };

    // ---- first conv (cin_pad = 32 kc_in channels), split form in two passes through the one accumulator: the correction
    // products first, scaled by 2^-11, then bias and the main product
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w_in), 0, TAPS * a.kc_in * 16 * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w_in_lo), 0, TAPS * a.kc_in * 16 * 1024, 0x00020000);
#pragma unroll
        for (int j = 0; j < RN; j++)
#pragma unroll
            for (int rt = 0; rt < RT; rt++) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        if (a.rnd_in) {   // RND input x / sum(x^2) (net5.rs:127), as in net_mfma_kernel
            for (int row = tid; row < RT * 16; row += NT) {
                int board = 0, px = -1;
                RM::decode(row, board, px);
                if (px < 0 || board >= valid_boards) continue;
                float ss = 0.0f;
                for (int sq = 0; sq < NN; sq++) ss += hscratch[RM::row_of(board, sq)];
                const tz_state* s = staged + board;
                const int fd = state_flat_diff<NB>(s);
                ET* out = reinterpret_cast<ET*>(a.rnd_in) + (size_t)(pos0 + board) * a.rnd_stride + px * a.cin_real;
                for (int c8 = 0; c8 < a.cin_real / 8; c8++) {
                    ex8 v;
#pragma unroll
                    for (int k = 0; k < 8; k++) v[k] = (ET)(plane_value<NB>(s, px, c8 * 8 + k, fd) / ss);
                    *reinterpret_cast<ex8*>(out + c8 * 8) = v;
                }
            }
        }
        for (int pass = 0; pass < 2; pass++) {
            for (int tap = 0; tap < TAPS; tap++) {
                int abase[RT];
#pragma unroll
                for (int rt = 0; rt < RT; rt++) abase[rt] = ta(tap, rt);
                for (int kc = 0; kc < a.kc_in; kc++) {
                    ex8 bh[RN], bl[RN];
#pragma unroll
                    for (int j = 0; j < RN; j++) {
                        const int fo = ((tap * a.kc_in + kc) * 16 + ct0 + j) * 1024;
                        bh[j] = __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, fo, 0));
                        bl[j] = __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(rsl, lane16, fo, 0));
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        const ex8 avh = *reinterpret_cast<const ex8*>(lds + abase[rt] + kc * PLANE);
                        if (pass == 0) {
                            const ex8 avl = *reinterpret_cast<const ex8*>(lds + abase[rt] + (8 + kc) * PLANE);
#pragma unroll
                            for (int j = 0; j < RN; j++) {
                                acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], avh, acc[rt][j], 0, 0, 0);
                                acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], avl, acc[rt][j], 0, 0, 0);
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < RN; j++) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], avh, acc[rt][j], 0, 0, 0);
                        }
                    }
                }
            }
            if (pass == 0) {
#pragma unroll
                for (int j = 0; j < RN; j++) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias_in + (ct0 + j) * 16 + q * 4);
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) acc[rt][j] = acc[rt][j] * SPLIT_INV + b4;
                }
            }
        }
        __syncthreads();
        epilogue(true, a.nlayers > 1 ? a.bias + FILTERS : nullptr, false);
    }
    // ---- residual tower
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w), 0, a.nlayers * LAYER_FRAGS * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc6 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.w8), 0, a.nlayers * LAYER_REC6 * REC6, 0x00020000);
    for (int layer = 0; layer < a.nlayers; layer++) {
        if ((layer & 1) == 0) {
#pragma unroll
            for (int j = 0; j < RN; j++) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + layer * FILTERS + (ct0 + j) * 16 + q * 4);
#pragma unroll
                for (int rt = 0; rt < RT; rt++) acc[rt][j] = b4;
            }
        }
        __syncthreads();
        {
            auto wl = [&](int tap, int kc, int j) -> ex8 {
                const int frag = layer * LAYER_FRAGS + (tap * 8 + kc) * 16 + (ct0 + j);
                return __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, frag * 1024, 0));
            };
            auto w6a = [&](int tap, int G, int j, int term) -> i32x4 {
                const int rec = layer * LAYER_REC6 + (tap * 2 + G) * 16 + (ct0 + j);
                return __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc6, lane16, rec * REC6 + term * 1536, 0));
            };
            auto w6b = [&](int tap, int G, int j, int term) -> i32x2 {
                const int rec = layer * LAYER_REC6 + (tap * 2 + G) * 16 + (ct0 + j);
                return __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(wrsrc6, opaque(lane16) >> 1, rec * REC6 + term * 1536 + 1024, 0));
            };
            auto wsf = [&](int tap, int G, int j) -> int {
                const int rec = layer * LAYER_REC6 + (tap * 2 + G) * 16 + (ct0 + j);
                return __builtin_amdgcn_raw_buffer_load_b32(wrsrc6, opaque(lane16) >> 2, rec * REC6 + 3072, 0);
            };
#ifdef TZ_C6_SKEW   // development builds: the second wave of every SIMD starts its k-loop 64 x TZ_C6_SKEW cycles late
            if (wave >= 4) __builtin_amdgcn_s_sleep(TZ_C6_SKEW);
#endif
            k_loop_c6<NB, P, PERM, RN>(lds, lane_const, acc, tab, wl, w6a, w6b, wsf);
        }
        __syncthreads();
        if ((layer & 1) == 0) epilogue(false, nullptr, true);
        else epilogue(true, layer + 2 < a.nlayers ? a.bias + (layer + 2) * FILTERS : nullptr, false);
    }
    __syncthreads();
    // ---- value / UBE heads: conv1x1(256->1) + bias, ReLU, Linear(nn->1) per board (net5.rs:89-120), on the tower's output in fp32
    // as the last epilogue left it in `seeds` (other waves' tiles: their stores were drained at the barrier above, the loads are
    // served by L2); a lane takes channels 4 lane .. of a row, i.e. lane group lane & 3 of output tile (wave lane >> 3, j (lane >> 2) & 1)
    {
        const float* hw = a.heads;
        float wv[4], wu[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            wv[k] = hw[lane * 4 + k];
            wu[k] = hw[FILTERS + lane * 4 + k];
        }
        const int htile = ((lane >> 3) * RN + ((lane >> 2) & 1)) * 1024 + (lane & 3) * 256;
        const float* lv = hw + 2 * FILTERS;
        const float* lu = lv + NN;
        const float bv = lu[NN], bu = lu[NN + 1], lbv = lu[NN + 2], lbu = lu[NN + 3];
        constexpr int HROWS = PERM ? RT * 16 : ROWS;
        constexpr int HU = 4;
        for (int row0 = wave; row0 < HROWS; row0 += NW * HU) {
            float dv[HU], du[HU];
#pragma unroll
            for (int u = 0; u < HU; u++) {
                const int row = min(row0 + NW * u, HROWS - 1);
                const f32x4 xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(seed_rs, htile + (row >> 4) * (NW * RN * 1024) + (row & 15) * 16, 0, TZ_C6_SEED_LD));
                dv[u] = 0.f;
                du[u] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    dv[u] += xv[k] * wv[k];
                    du[u] += xv[k] * wu[k];
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
                for (int u = 0; u < HU; u++) {
                    dv[u] += __shfl_xor(dv[u], d);
                    du[u] += __shfl_xor(du[u], d);
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < HU; u++) {
                    const int row = row0 + NW * u;
                    if (row < HROWS) {
                        const float av = dv[u] + bv, bb = du[u] + bu;
                        hscratch[row] = av > 0.f ? av : 0.f;
                        hscratch[RT * 16 + row] = bb > 0.f ? bb : 0.f;
                    }
                }
            }
        }
        __syncthreads();
        for (int pos = wave; pos < P; pos += NW) {
            if (pos0 + pos >= count) break;
            float sv = 0.f, su = 0.f;
            for (int px = lane; px < NN; px += 64) {
                const int hr = RM::row_of(pos, px);
                sv += hscratch[hr] * lv[px];
                su += hscratch[RT * 16 + hr] * lu[px];
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                sv += __shfl_xor(sv, d);
                su += __shfl_xor(su, d);
            }
            if (lane == 0) {
                a.value[pos0 + pos] = tanhf(sv + lbv);
                a.ube[pos0 + pos] = su + lbu;
            }
        }
    }
    // ---- policy conv (net5.rs:75-87): 16 RNP output channels per wave, fp32 out
    {
        constexpr int RNPW = RNP;
        const int ctp = wave * RNPW;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.w_pol), 0, TAPS * 8 * 8 * RNP * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs6 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.w_pol8), 0, TAPS * 2 * 8 * RNP * REC6, 0x00020000);
        f32x4 pacc[RT][RNPW];
#pragma unroll
        for (int j = 0; j < RNPW; j++) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias_pol + (ctp + j) * 16 + q * 4);
#pragma unroll
            for (int rt = 0; rt < RT; rt++) pacc[rt][j] = b4;
        }
        auto wl = [&](int tap, int kc, int j) -> ex8 {
            const int frag = (tap * 8 + kc) * (8 * RNP) + (ctp + j);
            return __builtin_bit_cast(ex8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, frag * 1024, 0));
        };
        auto w6a = [&](int tap, int G, int j, int term) -> i32x4 {
            const int rec = (tap * 2 + G) * (8 * RNP) + (ctp + j);
            return __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs6, lane16, rec * REC6 + term * 1536, 0));
        };
        auto w6b = [&](int tap, int G, int j, int term) -> i32x2 {
            const int rec = (tap * 2 + G) * (8 * RNP) + (ctp + j);
            return __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(rs6, opaque(lane16) >> 1, rec * REC6 + term * 1536 + 1024, 0));
        };
        auto wsf = [&](int tap, int G, int j) -> int {
            const int rec = (tap * 2 + G) * (8 * RNP) + (ctp + j);
            return __builtin_amdgcn_raw_buffer_load_b32(rs6, opaque(lane16) >> 2, rec * REC6 + 3072, 0);
        };
        k_loop_c6<NB, P, PERM, RNPW>(lds, lane_const, pacc, tab, wl, w6a, w6b, wsf);
#pragma unroll
        for (int j = 0; j < RNPW; j++) {
            const int cbase = (ctp + j) * 16 + q * 4;
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                int board = 0, sq = -1;
                RM::decode(rt * 16 + lr, board, sq);
                if (sq >= 0 && board < valid_boards)
                    *reinterpret_cast<f32x4*>(a.policy_out + (m0 + board * NN + sq) * a.pol_stride + cbase) = pacc[rt][j];
                // the store's data registers stay untouched for 16 cycles (see the seed stores: here it was the address arithmetic of
                // the next store that reused a dead accumulator register too early - single policy entries of a partial workgroup
                // came out 2e-6 off, tools/c6_forms_debug.py)
                asm volatile("s_nop 15" : "+v"(pacc[rt][j]));
            }
        }
    }
}

template <int NB, int P, int RNP, bool PERM>
int launch_c6(const NetArgs& a, int max_positions, hipStream_t st) {
    typedef C6Geo<NB, P, PERM> GEO;
    if (!a.seeds) return tz_fail(TZ_ESTATE, "TZ_PREC_F16C6: no seed buffer");
    auto kern = net_c6_kernel<NB, P, RNP, PERM>;
    static bool attr_done[64] = {};
    int attr_dev = 0;
    TZ_HIP(hipGetDevice(&attr_dev));
    bool& attr_set = attr_done[attr_dev & 63];
    if (!attr_set) {
        TZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEO::SMEM));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((max_positions + P - 1) / P), dim3(512), GEO::SMEM, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tz_fail(TZ_EDEVICE, std::string("net launch: ") + hipGetErrorString(e));
    return TZ_OK;
}

}  // namespace

#if !defined(TZ_C6_PART) || TZ_C6_PART == 0
// bytes of `seeds` a launch over max_positions positions of board size n may touch
size_t tz_nn_c6_seed_bytes(int n, int max_positions) {
    // the largest per-position share over the workgroup forms below (row tiles x 16 KB per workgroup)
    const size_t per_pos = n == 6 ? (size_t)5 * 16384 / 2 : (size_t)2 * 16384;   // 6x6: 2 boards on 5 row tiles; 5x5: 1 board on 2
    return per_pos * (size_t)(max_positions + 8) + 16 * 16384;
}
#endif

// The workgroup forms are spread over three translation units that compile side by side (TZ_C6_PART: tz_nn_c6.hip = 0, tz_nn_c6b.hip = 1,
// tz_nn_c6c.hip = 2): each unrolled k-loop is minutes of compile time.
#ifndef TZ_C6_PART
#define TZ_C6_PART 0
#endif
int tz_nn_launch_c6_b(int n, int p, const void* net_args, int max_positions, hipStream_t st);
int tz_nn_launch_c6_c(int n, int p, const void* net_args, int max_positions, hipStream_t st);
#if TZ_C6_PART == 0
int tz_nn_launch_c6(int n, const void* net_args, int max_positions, hipStream_t st) {
    const NetArgs& a = *static_cast<const NetArgs*>(net_args);
    const int small = net_small_p(max_positions);
    switch (n) {
        case 5:   // 8 boards per workgroup, square-major rows (26 of 117 (tap, tile) pairs skipped); small batches on 1, 2, 4 boards
            if (small == 0) return launch_c6<5, 8, 1, true>(a, max_positions, st);
            if (small == 4) return launch_c6<5, 4, 1, true>(a, max_positions, st);
            return tz_nn_launch_c6_b(5, small, net_args, max_positions, st);
        case 6:   // 4 boards, square-major rows (12 of 81 pairs skipped); below 1024 positions 2 boards
            return tz_nn_launch_c6_c(6, max_positions >= 1024 ? 4 : 2, net_args, max_positions, st);
    }
    return tz_fail(TZ_EINVAL, "TZ_PREC_F16C6: board sizes 5 and 6");
}
#elif TZ_C6_PART == 1
int tz_nn_launch_c6_b(int n, int p, const void* net_args, int max_positions, hipStream_t st) {
    const NetArgs& a = *static_cast<const NetArgs*>(net_args);
    if (n == 5 && p == 1) return launch_c6<5, 1, 1, false>(a, max_positions, st);
    if (n == 5 && p == 2) return launch_c6<5, 2, 1, false>(a, max_positions, st);
    return tz_fail(TZ_EINVAL, "TZ_PREC_F16C6: no such workgroup form");
}
#else
int tz_nn_launch_c6_c(int n, int p, const void* net_args, int max_positions, hipStream_t st) {
    const NetArgs& a = *static_cast<const NetArgs*>(net_args);
    if (n == 6 && p == 4) return launch_c6<6, 4, 2, true>(a, max_positions, st);
    if (n == 6 && p == 2) return launch_c6<6, 2, 2, false>(a, max_positions, st);
    return tz_fail(TZ_EINVAL, "TZ_PREC_F16C6: no such workgroup form");
}
#endif
