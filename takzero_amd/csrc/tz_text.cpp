// tz_text.cpp — host-side text and index helpers of libtakzero_hip.so: TPS <-> tz_state and
// PTN <-> move_index, the formats the reference reads and writes through takparse
// (call sites takzero/src/target.rs:56-73,99-143,215-268).  Pure host code,
// no rules: legality lives in the device kernels.
#include <charconv>
#include <cmath>
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

// ---- batched target lines (target.rs:56-73 Display, :99-143 FromStr): what selfplay / reanalyze append and learn
// tails, thousands per move — formatted and parsed here rather than line by line in the host language.
namespace {
// Rust's `Display for f32`: shortest decimal that round-trips, never an exponent, "1" for 1.0, "NaN", "inf"
void put_f32(std::string& o, float x) {
    if (std::isnan(x)) {
        o += "NaN";
        return;
    }
    if (std::isinf(x)) {
        o += x > 0 ? "inf" : "-inf";
        return;
    }
    // shortest round-trip digits from the scientific form, then laid out positionally (Rust pads with zeros where
    // to_chars' fixed form would print the exact binary value)
    char buf[64];
    const auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);
    const char* p = buf;
    if (*p == '-') {
        o += '-';
        p++;
    }
    char digits[32];
    int nd = 0;
    for (; p < r.ptr && *p != 'e'; p++)
        if (*p != '.') digits[nd++] = *p;
    int exp10 = 0;
    if (p < r.ptr && *p == 'e') {
        p++;
        const bool neg = *p == '-';
        if (*p == '-' || *p == '+') p++;
        for (; p < r.ptr; p++) exp10 = exp10 * 10 + (*p - '0');
        if (neg) exp10 = -exp10;
    }
    while (nd > 1 && digits[nd - 1] == '0') nd--;
    if (exp10 < 0) {                 // 0.000ddd
        o += "0.";
        o.append((size_t)(-exp10 - 1), '0');
        o.append(digits, nd);
    } else if (exp10 + 1 >= nd) {    // ddd000
        o.append(digits, nd);
        o.append((size_t)(exp10 + 1 - nd), '0');
    } else {                         // dd.ddd
        o.append(digits, exp10 + 1);
        o += '.';
        o.append(digits + exp10 + 1, nd - exp10 - 1);
    }
}
bool get_f32(const char* b, const char* e, float* out) {
    if (b < e && *b == '+') b++;  // Rust's f32::from_str accepts a leading '+', from_chars does not
    const auto r = std::from_chars(b, e, *out, std::chars_format::general);
    return r.ec == std::errc() && r.ptr == e && !std::isnan(*out);
}
}  // namespace

// `count` targets -> text.  moves / policy are [count][amax] (nmoves[i] entries used).  TZ_ECAPACITY if `cap` is too
// small (a line needs at most 160 + 32 * nmoves bytes).
int tz_format_targets(int n, int count, const tz_state* states, const uint16_t* moves, const float* policy,
                      const int32_t* nmoves, int amax, const float* value, const float* ube, char* out, uint64_t cap,
                      uint64_t* written_out) {
    if (!states || !moves || !policy || !nmoves || !value || !ube || !out || !written_out || count < 0 || amax <= 0)
        return tz_fail(TZ_EINVAL, "tz_format_targets: bad argument");
    std::string o;
    o.reserve((size_t)count * 1024);
    char buf[256];
    for (int i = 0; i < count; i++) {
        int rc = tz_state_to_tps(states + i, buf, sizeof buf);
        if (rc) return rc;
        o += buf;
        o += ';';
        put_f32(o, value[i]);
        o += ';';
        put_f32(o, ube[i]);
        o += ';';
        if (nmoves[i] < 0 || nmoves[i] > amax) return tz_fail(TZ_EINVAL, "tz_format_targets: nmoves out of range");
        for (int k = 0; k < nmoves[i]; k++) {
            if (k) o += ',';
            rc = tz_move_to_ptn(n, moves[(size_t)i * amax + k], buf, sizeof buf);
            if (rc) return rc;
            o += buf;
            o += ':';
            put_f32(o, policy[(size_t)i * amax + k]);
        }
        o += '\n';
    }
    *written_out = o.size();
    if (o.size() > cap) return tz_fail(TZ_ECAPACITY, "tz_format_targets: output buffer too small");
    memcpy(out, o.data(), o.size());
    return TZ_OK;
}

// Parses complete lines of `text` (up to max_targets of them).  Lines that do not parse are skipped, as the
// reference's filter_map(|line| line.parse().ok()) does (learn/src/main.rs:308); *consumed_out = bytes up to and
// including the last line looked at, so a half-written last line is left for the next call.
int tz_parse_targets(const char* text, uint64_t len, int n, int half_komi, int max_targets, int amax, tz_state* states,
                     uint16_t* moves, float* policy, int32_t* nmoves, float* value, float* ube, int32_t* count_out,
                     uint64_t* consumed_out, int32_t* skipped_out) {
    if (!text || !states || !moves || !policy || !nmoves || !value || !ube || !count_out || !consumed_out || amax <= 0)
        return tz_fail(TZ_EINVAL, "tz_parse_targets: bad argument");
    int count = 0, skipped = 0;
    uint64_t pos = 0;
    std::string tmp;
    while (pos < len && count < max_targets) {
        const char* b = text + pos;
        const char* nl = (const char*)memchr(b, '\n', len - pos);
        if (!nl) break;
        const char* e = nl;
        pos = (uint64_t)(nl - text) + 1;
        while (e > b && (e[-1] == '\r' || e[-1] == ' ')) e--;
        bool ok = false;
        do {
            const char* s1 = (const char*)memchr(b, ';', e - b);
            if (!s1) break;
            const char* s2 = (const char*)memchr(s1 + 1, ';', e - s1 - 1);
            if (!s2) break;
            const char* s3 = (const char*)memchr(s2 + 1, ';', e - s2 - 1);
            if (!s3) break;
            tmp.assign(b, s1);
            if (tz_state_from_tps(tmp.c_str(), n, half_komi, states + count)) break;
            if (!get_f32(s1 + 1, s2, value + count) || !get_f32(s2 + 1, s3, ube + count)) break;
            int k = 0;
            const char* p = s3 + 1;
            bool bad = p >= e;
            while (!bad && p < e) {
                const char* comma = (const char*)memchr(p, ',', e - p);
                const char* item_end = comma ? comma : e;
                const char* colon = (const char*)memchr(p, ':', item_end - p);
                if (!colon || k >= amax) {
                    bad = true;
                    break;
                }
                tmp.assign(p, colon);
                if (tz_move_from_ptn(n, tmp.c_str(), moves + (size_t)count * amax + k) ||
                    !get_f32(colon + 1, item_end, policy + (size_t)count * amax + k)) {
                    bad = true;
                    break;
                }
                k++;
                p = comma ? comma + 1 : e;
                if (comma && p >= e) bad = true;  // trailing comma
            }
            if (bad) break;
            nmoves[count] = k;
            ok = true;
        } while (0);
        if (ok) count++;
        else skipped++;
    }
    *count_out = count;
    *consumed_out = pos;
    if (skipped_out) *skipped_out = skipped;
    return TZ_OK;
}

}  // extern "C"
