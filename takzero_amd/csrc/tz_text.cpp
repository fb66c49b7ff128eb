// tz_text.cpp — host-side text and index helpers of libtakzero_hip.so: TPS <-> tz_state and
// PTN <-> move_index, the formats the reference reads and writes through takparse
// (call sites takzero/src/target.rs:56-73,99-143,215-268).  Pure host code,
// no rules: legality lives in the device kernels.
#include <cstdio>
#include <cstring>
#include <string>

#include "tz_engine.h"

namespace {
thread_local std::string g_err;
void reserves_for(int n, int& stones, int& caps) {
    stones = n == 3 ? 10 : n == 4 ? 15 : n == 5 ? 21 : n == 6 ? 30 : 0;
    caps = n >= 5 ? 1 : 0;
}
}  // namespace

void tz_set_error(const std::string& msg) { g_err = msg; }
int tz_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

extern "C" {

const char* tz_last_error(void) { return g_err.c_str(); }
int tz_version(void) { return 1; }

int tz_policy_size(int n) { return n < 3 || n > 6 ? TZ_EINVAL : n * n * (3 + 4 * ((1 << n) - 2)); }
int tz_input_channels(int n) { return n < 3 || n > 6 ? TZ_EINVAL : 2 * ((3 + (n - 1) + (n + 1)) + 2) + 2; }

int tz_state_from_tps(const char* tps, int n, int half_komi, tz_state* out) {
    if (!tps || !out || n < 3 || n > 6) return tz_fail(TZ_EINVAL, "tz_state_from_tps: bad argument");
    memset(out, 0, sizeof *out);
    int stones, caps;
    reserves_for(n, stones, caps);
    int used[2] = {0, 0}, used_caps[2] = {0, 0};
    const char* p = tps;
    int row = n - 1, col = 0;
    while (*p && *p != ' ') {
        const char c = *p;
        if (c == '/') {
            if (col != n || row == 0) return tz_fail(TZ_EPARSE, "TPS: wrong number of squares in a rank");
            row--;
            col = 0;
            p++;
        } else if (c == ',') {
            p++;
        } else if (c == 'x') {
            p++;
            int k = 1;
            if (*p >= '1' && *p <= '8') k = *p++ - '0';
            col += k;
            if (col > n) return tz_fail(TZ_EPARSE, "TPS: rank too long");
        } else if (c == '1' || c == '2') {
            if (col >= n) return tz_fail(TZ_EPARSE, "TPS: rank too long");
            const int sq = row * n + col;
            int h = 0;
            uint64_t bits = 0;
            while (*p == '1' || *p == '2') {
                if (h >= 64) return tz_fail(TZ_EPARSE, "TPS: stack too tall");
                const int color = *p - '1';
                if (color) bits |= 1ull << h;
                used[color]++;
                h++;
                p++;
            }
            uint8_t top = TZ_FLAT;
            if (*p == 'S') {
                top = TZ_WALL;
                p++;
            } else if (*p == 'C') {
                top = TZ_CAP;
                p++;
                const int color = (int)((bits >> (h - 1)) & 1);
                used[color]--;
                used_caps[color]++;
            }
            out->colors[sq] = bits;
            out->height[sq] = (uint8_t)h;
            out->top[sq] = top;
            col++;
        } else {
            return tz_fail(TZ_EPARSE, std::string("TPS: unexpected character '") + c + "'");
        }
    }
    if (row != 0 || col != n) return tz_fail(TZ_EPARSE, "TPS: wrong number of ranks");
    int to_move = 0, move_no = 0;
    if (sscanf(p, " %d %d", &to_move, &move_no) != 2 || to_move < 1 || to_move > 2 || move_no < 1)
        return tz_fail(TZ_EPARSE, "TPS: missing side to move / move number");
    for (int c = 0; c < 2; c++) {
        if (used[c] > stones || used_caps[c] > caps) return tz_fail(TZ_EPARSE, "TPS: more pieces than reserves");
        out->stones[c] = (uint8_t)(stones - used[c]);
        out->caps[c] = (uint8_t)(caps - used_caps[c]);
    }
    out->to_move = (uint8_t)(to_move - 1);
    out->n = (uint8_t)n;
    out->half_komi = (int8_t)half_komi;
    out->ply = (uint16_t)((move_no - 1) * 2 + (to_move - 1));
    out->reversible_plies = 0;
    return TZ_OK;
}

int tz_state_to_tps(const tz_state* s, char* buf, int buflen) {
    if (!s || !buf) return tz_fail(TZ_EINVAL, "tz_state_to_tps: null argument");
    const int n = s->n;
    std::string o;
    for (int row = n - 1; row >= 0; row--) {
        int run = 0;
        bool first = true;
        auto flush = [&]() {
            if (!run) return;
            if (!first) o += ',';
            o += 'x';
            if (run > 1) o += (char)('0' + run);
            run = 0;
            first = false;
        };
        for (int col = 0; col < n; col++) {
            const int sq = row * n + col;
            if (s->top[sq] == TZ_EMPTY) {
                run++;
                continue;
            }
            flush();
            if (!first) o += ',';
            first = false;
            for (int i = 0; i < s->height[sq]; i++) o += ((s->colors[sq] >> i) & 1) ? '2' : '1';
            if (s->top[sq] == TZ_WALL) o += 'S';
            if (s->top[sq] == TZ_CAP) o += 'C';
        }
        flush();
        if (row) o += '/';
    }
    o += ' ';
    o += s->to_move ? '2' : '1';
    o += ' ';
    o += std::to_string(s->ply / 2 + 1);
    if ((int)o.size() + 1 > buflen) return tz_fail(TZ_EINVAL, "tz_state_to_tps: buffer too small");
    memcpy(buf, o.c_str(), o.size() + 1);
    return TZ_OK;
}

// move_index layout: takzero/src/network/repr.rs:49-71
int tz_move_to_ptn(int n, uint16_t move_index, char* buf, int buflen) {
    if (n < 3 || n > 6 || !buf) return tz_fail(TZ_EINVAL, "tz_move_to_ptn: bad argument");
    const int nn = n * n, patterns = (1 << n) - 2;
    if (move_index >= nn * (3 + 4 * patterns)) return tz_fail(TZ_EINVAL, "tz_move_to_ptn: index out of range");
    const int channel = move_index / nn, sq = move_index % nn;
    std::string o;
    const char file = (char)('a' + sq % n), rank = (char)('1' + sq / n);
    if (channel < 3) {
        if (channel == 1) o += 'S';
        if (channel == 2) o += 'C';
        o += file;
        o += rank;
    } else {
        const int slot = (channel - 3) / patterns, v = (channel - 3) % patterns + 1;
        const int p0 = __builtin_ctz(v), carry = n - p0;
        if (carry != 1) o += (char)('0' + carry);
        o += file;
        o += rank;
        o += "+>-<"[slot];
        std::string drops;
        int b = p0;
        while (b < n) {
            int len = 1;
            while (b + len < n && !((v >> (b + len)) & 1)) len++;
            drops += (char)('0' + len);
            b += len;
        }
        if (drops.size() > 1) o += drops;
    }
    if ((int)o.size() + 1 > buflen) return tz_fail(TZ_EINVAL, "tz_move_to_ptn: buffer too small");
    memcpy(buf, o.c_str(), o.size() + 1);
    return TZ_OK;
}

int tz_move_from_ptn(int n, const char* ptn, uint16_t* move_index_out) {
    if (n < 3 || n > 6 || !ptn || !move_index_out) return tz_fail(TZ_EINVAL, "tz_move_from_ptn: bad argument");
    const int nn = n * n, patterns = (1 << n) - 2;
    std::string s(ptn);
    while (!s.empty() && strchr("*'?!", s.back())) s.pop_back();
    size_t i = 0;
    int count = 0;
    if (i < s.size() && s[i] >= '1' && s[i] <= '8') count = s[i++] - '0';
    int piece = -1;
    if (i < s.size() && (s[i] == 'F' || s[i] == 'S' || s[i] == 'C')) piece = s[i] == 'F' ? 0 : s[i] == 'S' ? 1 : 2, i++;
    if (i + 2 > s.size() || s[i] < 'a' || s[i] >= 'a' + n || s[i + 1] < '1' || s[i + 1] >= '1' + n)
        return tz_fail(TZ_EPARSE, std::string("PTN: bad square in '") + ptn + "'");
    const int sq = (s[i + 1] - '1') * n + (s[i] - 'a');
    i += 2;
    if (i == s.size()) {
        if (count) return tz_fail(TZ_EPARSE, "PTN: count on a placement");
        *move_index_out = (uint16_t)((piece < 0 ? 0 : piece) * nn + sq);
        return TZ_OK;
    }
    if (piece >= 0) return tz_fail(TZ_EPARSE, "PTN: piece letter on a spread");
    const char* dirs = "+>-<";
    const char* dp = strchr(dirs, s[i++]);
    if (!dp) return tz_fail(TZ_EPARSE, "PTN: bad direction");
    const int slot = (int)(dp - dirs);
    if (!count) count = 1;
    if (count > n) return tz_fail(TZ_EPARSE, "PTN: carry exceeds board size");
    const int p0 = n - count;
    int v = 1 << p0, acc = 0, sum = 0, parts = 0;
    if (i == s.size()) {
        sum = count;
        parts = 1;
    }
    for (; i < s.size(); i++) {
        if (s[i] < '1' || s[i] > '8') return tz_fail(TZ_EPARSE, "PTN: bad drop count");
        if (parts > 0) v |= 1 << (p0 + acc);
        acc += s[i] - '0';
        sum += s[i] - '0';
        parts++;
    }
    if (sum != count || parts >= n) return tz_fail(TZ_EPARSE, "PTN: drops do not match the carry");
    *move_index_out = (uint16_t)((3 + slot * patterns + v - 1) * nn + sq);
    return TZ_OK;
}

}  // extern "C"
