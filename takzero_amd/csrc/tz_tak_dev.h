// tz_tak_dev.h — Tak rules on the device, one wavefront (64 lanes) per game.
//
// Replaces what the reference gets from fast_tak::Game::{play, result, possible_moves}
// (call sites takzero/src/search/env.rs:39-59) for the tree kernels.  The game lives in LDS as a
// tz_state; squares map to lanes (sq = row*N + col, N*N <= 36 < 64), so board scans are single
// __ballot()s, road detection is a bitboard flood fill on wave-uniform 64-bit masks, and move
// generation is a work-item scan (square x {place | carry x direction}) with a wave prefix sum.
// Moves are identified by the reference's policy index (repr.rs:49-71).
//
// Move order follows fast-tak's possible_moves as pinned by runs/*.txt: squares
// file-major, Flat/Wall/Cap on empty squares, carry ascending, directions + - < >, drop sequences
// in descending lexicographic order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/takzero_hip.h"

#define TZ_REVERSIBLE_PLIES_LIMIT 100  // assumption, see DESIGN.md (fast-tak source unavailable)

namespace tzd {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// generation order of directions: + - < >  ; move_index direction slots: Up 0, Right 1, Down 2, Left 3
__device__ __forceinline__ int gen_dir_dx(int d) { return d == 2 ? -1 : d == 3 ? 1 : 0; }
__device__ __forceinline__ int gen_dir_dy(int d) { return d == 0 ? 1 : d == 1 ? -1 : 0; }
__device__ __forceinline__ int gen_dir_to_slot(int d) { return d == 0 ? 0 : d == 1 ? 2 : d == 2 ? 3 : 1; }
__device__ __forceinline__ int slot_dx(int s) { return s == 1 ? 1 : s == 3 ? -1 : 0; }
__device__ __forceinline__ int slot_dy(int s) { return s == 0 ? 1 : s == 2 ? -1 : 0; }

__device__ __forceinline__ uint64_t mask_bits(int k) { return k >= 64 ? ~0ull : ((1ull << k) - 1ull); }

template <int N>
struct Geo {
    static constexpr int NN = N * N;
    static constexpr int PATTERNS = (1 << N) - 2;
    static constexpr int OUT_CH = 3 + 4 * PATTERNS;
    static constexpr int ITEMS_PER_SQ = 1 + 4 * N;
    static constexpr int ITEMS = NN * ITEMS_PER_SQ;
    __device__ static constexpr uint64_t full() { return NN == 64 ? ~0ull : ((1ull << NN) - 1ull); }
    __device__ static constexpr uint64_t col0() {
        uint64_t m = 0;
        for (int y = 0; y < N; y++) m |= 1ull << (y * N);
        return m;
    }
    __device__ static constexpr uint64_t row0() { return (1ull << N) - 1ull; }
};

// ---------------------------------------------------------------- play  (Game::play)
// Executed by every lane on identical data; only lane 0 stores.  Caller __syncthreads() after.
template <int N>
__device__ void apply_move(tz_state* e, int move_idx) {
    constexpr int NN = N * N;
    const int channel = move_idx / NN, sq = move_idx % NN;
    if (lane_id() != 0) return;
    const int to_move = e->to_move;
    if (channel < 3) {
        const int color = e->ply < 2 ? 1 - to_move : to_move;  // opening rule
        e->colors[sq] = (uint64_t)color;
        e->height[sq] = 1;
        e->top[sq] = (uint8_t)(channel == 0 ? TZ_FLAT : channel == 1 ? TZ_WALL : TZ_CAP);
        if (channel == 2) e->caps[color]--; else e->stones[color]--;
        e->reversible_plies = 0;
    } else {
        const int slot = (channel - 3) / Geo<N>::PATTERNS;
        const int v = (channel - 3) % Geo<N>::PATTERNS + 1;
        const int p0 = __ffs(v) - 1;
        const int c = N - p0;
        const int h = e->height[sq];
        const uint64_t src = e->colors[sq];
        const uint64_t take = (src >> (h - c)) & mask_bits(c);
        const uint8_t top_piece = e->top[sq];
        e->height[sq] = (uint8_t)(h - c);
        e->colors[sq] = src & mask_bits(h - c);
        e->top[sq] = (uint8_t)(h - c > 0 ? TZ_FLAT : TZ_EMPTY);
        const int dsq = slot_dy(slot) * N + slot_dx(slot);
        int cur = sq, pos = 0;
        bool flattened = false;
        // drops: bit p0 starts the first square; every further set bit starts the next one
        int b = p0;
        while (b < N) {
            int len = 1;
            while (b + len < N && !((v >> (b + len)) & 1)) len++;
            cur += dsq;
            const int dh = e->height[cur];
            const bool last = b + len >= N;
            if (e->top[cur] == TZ_WALL) flattened = true;
            e->colors[cur] |= ((take >> pos) & mask_bits(len)) << dh;
            e->height[cur] = (uint8_t)(dh + len);
            e->top[cur] = last ? top_piece : (uint8_t)TZ_FLAT;
            pos += len;
            b += len;
        }
        e->reversible_plies = flattened ? 0 : (uint16_t)(e->reversible_plies + 1);
    }
    e->ply++;
    e->to_move = (uint8_t)(1 - to_move);
}

// ---------------------------------------------------------------- result  (Game::result)
template <int N>
__device__ __forceinline__ bool has_road(uint64_t bb) {
    constexpr uint64_t FULL = Geo<N>::full(), COL0 = Geo<N>::col0(), ROW0 = Geo<N>::row0();
    constexpr uint64_t COLL = COL0 << (N - 1), ROWL = ROW0 << (N * (N - 1));
    // left -> right
    uint64_t cur = bb & COL0;
    for (int it = 0; it < N * N && cur; it++) {
        uint64_t g = cur | ((cur << 1) & ~COL0) | ((cur >> 1) & ~COLL) | (cur << N) | (cur >> N);
        g &= bb & FULL;
        if (g == cur) break;
        cur = g;
    }
    if (cur & COLL) return true;
    cur = bb & ROW0;
    for (int it = 0; it < N * N && cur; it++) {
        uint64_t g = cur | ((cur << 1) & ~COL0) | ((cur >> 1) & ~COLL) | (cur << N) | (cur >> N);
        g &= bb & FULL;
        if (g == cur) break;
        cur = g;
    }
    return (cur & ROWL) != 0;
}

struct BoardScan {
    uint64_t road[2];   // flats + caps by colour
    uint64_t flats[2];  // top flats by colour
    uint64_t empty;
};

template <int N>
__device__ __forceinline__ BoardScan scan_board(const tz_state* e) {
    constexpr int NN = N * N;
    const int l = lane_id();
    int t = TZ_EMPTY, col = 0;
    if (l < NN) {
        t = e->top[l];
        const int h = e->height[l];
        col = h ? (int)((e->colors[l] >> (h - 1)) & 1ull) : 0;
    }
    BoardScan s;
    const bool in = l < NN;
    const bool roadp = in && (t == TZ_FLAT || t == TZ_CAP);
    s.road[0] = __ballot(roadp && col == 0);
    s.road[1] = __ballot(roadp && col == 1);
    s.flats[0] = __ballot(in && t == TZ_FLAT && col == 0);
    s.flats[1] = __ballot(in && t == TZ_FLAT && col == 1);
    s.empty = __ballot(in && t == TZ_EMPTY);
    return s;
}

// Environment::terminal (env.rs:47-59): TZ_TERMINAL_* from the side to move. Wave-uniform.
// *reason (optional): 0 ongoing, 1 road, 2 flat count (board full / reserves empty), 3 reversible-plies draw.
template <int N>
__device__ int terminal(const tz_state* e, int* reason = nullptr) {
    const BoardScan s = scan_board<N>(e);
    const int to_move = e->to_move, mover = 1 - to_move;
    int winner = -1, why = 0;  // 0 white, 1 black, 2 draw
    if (e->ply > 0) {
        if (has_road<N>(s.road[mover])) winner = mover;
        else if (has_road<N>(s.road[to_move])) winner = to_move;
        if (winner >= 0) why = 1;
    }
    if (winner < 0) {
        const bool depleted = (e->stones[0] == 0 && e->caps[0] == 0) || (e->stones[1] == 0 && e->caps[1] == 0);
        if (s.empty == 0 || depleted) {
            const int w = 2 * __popcll(s.flats[0]), b = 2 * __popcll(s.flats[1]) + e->half_komi;
            winner = w > b ? 0 : b > w ? 1 : 2;
            why = 2;
        } else if (e->reversible_plies >= TZ_REVERSIBLE_PLIES_LIMIT) {
            winner = 2;
            why = 3;
        }
    }
    if (reason) *reason = why;
    if (winner < 0) return TZ_TERMINAL_NONE;
    if (winner == 2) return TZ_TERMINAL_DRAW;
    return winner == to_move ? TZ_TERMINAL_WIN : TZ_TERMINAL_LOSS;
}

template <int N>
__device__ __forceinline__ int flat_diff(const tz_state* e) {
    const BoardScan s = scan_board<N>(e);
    return __popcll(s.flats[0]) - __popcll(s.flats[1]);
}

// ---------------------------------------------------------------- possible_moves
__device__ __forceinline__ int binom(int n, int k) {
    // n <= 5
    if (k < 0 || k > n || n < 0) return 0;
    const int tab[6][6] = {{1, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0}, {1, 2, 1, 0, 0, 0},
                           {1, 3, 3, 1, 0, 0}, {1, 4, 6, 4, 1, 0}, {1, 5, 10, 10, 5, 1}};
    return tab[n][k];
}

// number of legal drop sequences for carrying c pieces with `free_sq` enterable squares and an
// optional flattenable wall behind them
__device__ __forceinline__ int spread_count(int c, int free_sq, bool flatten) {
    int cnt = 0;
    const int kmax = free_sq < c ? free_sq : c;
    for (int k = 1; k <= kmax; k++) cnt += binom(c - 1, k - 1);
    if (flatten && c >= free_sq + 1) cnt += free_sq == 0 ? (c == 1 ? 1 : 0) : binom(c - 2, free_sq - 1);
    return cnt;
}

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(v, d);
        if (l >= d) v += o;
    }
    return v;
}

// scratch: reach[NN*4] bytes in LDS.  out: move indices (LDS or global), returns count (uniform).
// Caller must __syncthreads() before (env stable) — this function syncs internally after writing reach.
template <int N>
__device__ int gen_moves(const tz_state* e, uint8_t* reach, uint16_t* out, int out_cap) {
    constexpr int NN = N * N;
    const int l = lane_id();
    const int to_move = e->to_move;
    const bool opening = e->ply < 2;
    // per (square, direction): free squares and whether a wall follows
    if (l < NN) {
        const int y = l / N, x = l % N;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            int fr = 0, wall = 0, cx = x, cy = y;
            for (int s = 0; s < N - 1; s++) {
                cx += gen_dir_dx(d);
                cy += gen_dir_dy(d);
                if (cx < 0 || cy < 0 || cx >= N || cy >= N) break;
                const int t = e->top[cy * N + cx];
                if (t == TZ_CAP) break;
                if (t == TZ_WALL) { wall = 1; break; }
                fr++;
            }
            reach[l * 4 + d] = (uint8_t)(fr | (wall << 3));
        }
    }
    __syncthreads();
    const int stones = e->stones[to_move], caps = e->caps[to_move];
    int base = 0;
    for (int chunk = 0; chunk < Geo<N>::ITEMS; chunk += 64) {
        const int item = chunk + l;
        int cnt = 0, sq = 0, t = 0, c = 0, d = 0, fr = 0;
        bool flatten = false;
        if (item < Geo<N>::ITEMS) {
            const int sqo = item / Geo<N>::ITEMS_PER_SQ;  // file-major
            t = item % Geo<N>::ITEMS_PER_SQ;
            const int x = sqo / N, y = sqo % N;
            sq = y * N + x;
            const int top = e->top[sq], h = e->height[sq];
            if (t == 0) {
                if (top == TZ_EMPTY) cnt = opening ? 1 : ((stones > 0 ? 2 : 0) + (caps > 0 ? 1 : 0));
            } else if (!opening && top != TZ_EMPTY) {
                const int owner = (int)((e->colors[sq] >> (h - 1)) & 1ull);
                c = (t - 1) / 4 + 1;
                d = (t - 1) % 4;
                if (owner == to_move && c <= h) {
                    const int r = reach[sq * 4 + d];
                    fr = r & 7;
                    flatten = (r >> 3) && top == TZ_CAP;
                    cnt = spread_count(c, fr, flatten);
                }
            }
        }
        const int incl = wave_incl_scan(cnt);
        int off = base + incl - cnt;
        if (cnt > 0 && off + cnt <= out_cap) {
            if (t == 0) {
                if (opening) {
                    out[off] = (uint16_t)sq;
                } else {
                    if (stones > 0) {
                        out[off++] = (uint16_t)sq;
                        out[off++] = (uint16_t)(NN + sq);
                    }
                    if (caps > 0) out[off++] = (uint16_t)(2 * NN + sq);
                }
            } else {
                const int slot = gen_dir_to_slot(d);
                const int p0 = N - c, ncuts = c - 1;
                for (int w = 0; w < (1 << ncuts); w++) {
                    const int parts = __popc(w) + 1;
                    bool ok = parts <= fr;
                    if (!ok && flatten && parts == fr + 1) ok = ncuts == 0 ? (c == 1) : (w & 1);
                    if (!ok) continue;
                    // cut after piece j <-> bit (ncuts - j) of w <-> bit (p0 + j) of v
                    const int rev = ncuts ? (int)(__brev((unsigned)w) >> (32 - ncuts)) : 0;
                    const int v = (1 << p0) | (rev << (p0 + 1));
                    out[off++] = (uint16_t)((3 + slot * Geo<N>::PATTERNS + v - 1) * NN + sq);
                }
            }
        }
        base += __shfl(incl, 63);
    }
    __syncthreads();
    return base;
}

// ---------------------------------------------------------------- openings (env.rs:65-79)
// symmetry order is this engine's choice (unpinned): sym = rot*2 + mirror.
__device__ __forceinline__ void symmetry_apply(int n, int sym, int x, int y, int& ox, int& oy) {
    if (sym & 1) x = n - 1 - x;
    const int rot = (sym >> 1) & 3;
    for (int r = 0; r < rot; r++) {
        const int nx = n - 1 - y, ny = x;
        x = nx;
        y = ny;
    }
    ox = x;
    oy = y;
}

__device__ __forceinline__ void default_reserves(int n, int& stones, int& caps) {
    stones = n == 3 ? 10 : n == 4 ? 15 : n == 5 ? 21 : 30;
    caps = n >= 5 ? 1 : 0;
}

// lane 0 only
template <int N>
__device__ void write_opening(tz_state* e, int half_komi, int choice, bool with_moves) {
    {   // every byte, including the struct's tail padding: positions are compared and hashed as raw bytes
        uint32_t* wds = reinterpret_cast<uint32_t*>(e);
        for (int i = 0; i < (int)(sizeof(tz_state) / 4); i++) wds[i] = 0;
    }
    int st, cp;
    default_reserves(N, st, cp);
    e->stones[0] = e->stones[1] = (uint8_t)st;
    e->caps[0] = e->caps[1] = (uint8_t)cp;
    e->to_move = 0;
    e->n = N;
    e->half_komi = (int8_t)half_komi;
    e->pad0 = 0;
    e->ply = 0;
    e->reversible_plies = 0;
    if (!with_moves) return;
    const int sym = (choice >> 1) & 7, opposite = choice & 1;
    const int sqs[2][2] = {{0, 0}, {opposite ? N - 1 : 0, N - 1}};
    for (int i = 0; i < 2; i++) {
        int ox, oy;
        symmetry_apply(N, sym, sqs[i][0], sqs[i][1], ox, oy);
        const int sq = oy * N + ox, color = 1 - i;  // white places black's flat, then black places white's
        e->colors[sq] = (uint64_t)color;
        e->height[sq] = 1;
        e->top[sq] = TZ_FLAT;
        e->stones[color]--;
    }
    e->ply = 2;
    e->to_move = 0;
}

}  // namespace tzd
