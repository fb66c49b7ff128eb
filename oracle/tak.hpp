// oracle/tak.hpp — CPU restatement of the Tak rules the reference obtains from the un-vendored
// crates fast-tak 0.4.1 / takparse 0.6.0 (Cargo.lock:611-617,1564-1566).
//
// TEST INFRASTRUCTURE ONLY.  Nothing under takzero_amd/ may include, link or call this file;
// it exists so tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg have an
// independent checker for the HIP engine.
//
// What pins it: the reference's own tests at takzero/src/network/repr.rs:261-499
// (plane encodings, legal-move set, move_index layout), search/node/mcts.rs:346-411 (road wins),
// and the move *order* visible in runs/*.txt (file-major squares; Flat,Wall,Cap on empty squares;
// pickup count ascending; directions + - < >; drop sequences in descending lexicographic order).
// Parity unpinned (fast-tak source absent): the reversible-plies draw limit and its reset rule,
// and the order of the 8 board symmetries used by new_opening; both are single constants /
// tables below and are recorded as assumptions in DESIGN.md.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../include/takzero_hip.h"

namespace tzo {

enum Piece : uint8_t { NONE = 0, FLAT = 1, WALL = 2, CAP = 3 };
// generation order of directions as seen in runs/*.txt:  + - < >
enum Dir : uint8_t { UP = 0, DOWN = 1, LEFT = 2, RIGHT = 3 };
enum Result : uint8_t { ONGOING = 0, WHITE_WINS = 1, BLACK_WINS = 2, DRAW = 3 };

// assumption (unpinned): fast-tak ends a game as a draw once this many consecutive reversible
// plies (spreads that do not flatten a wall) have been played.
static constexpr int REVERSIBLE_PLIES_LIMIT = 100;

static inline void default_reserves(int n, int& stones, int& caps) {
    // python/action_space.py:63-64, repr.rs:308-309,367-368
    switch (n) {
        case 3: stones = 10; caps = 0; break;
        case 4: stones = 15; caps = 0; break;
        case 5: stones = 21; caps = 1; break;
        case 6: stones = 30; caps = 1; break;
        default: stones = 0; caps = 0; break;
    }
}

struct Move {
    uint8_t spread = 0;  // 0 place, 1 spread
    uint8_t piece = FLAT;  // for placements
    uint8_t x = 0, y = 0;  // column (file), row (rank-1)
    uint8_t dir = UP;
    uint8_t ndrops = 0;
    uint8_t drops[8] = {0};
    int carried() const {
        int c = 0;
        for (int i = 0; i < ndrops; i++) c += drops[i];
        return c;
    }
    bool operator==(const Move& o) const {
        if (spread != o.spread || x != o.x || y != o.y) return false;
        if (!spread) return piece == o.piece;
        if (dir != o.dir || ndrops != o.ndrops) return false;
        for (int i = 0; i < ndrops; i++)
            if (drops[i] != o.drops[i]) return false;
        return true;
    }
};

static inline int dir_dx(int d) { return d == LEFT ? -1 : d == RIGHT ? 1 : 0; }
static inline int dir_dy(int d) { return d == UP ? 1 : d == DOWN ? -1 : 0; }

// move_index: takzero/src/network/repr.rs:49-71.  Direction offsets there: Up 0, Right 1,
// Down 2, Left 3.  Pattern value v = pattern.mask() >> (8-N): lowest set bit at N-carried,
// each further set bit starts the next square (checked against repr.rs:438-448).
static inline int pattern_value(int n, const Move& m) {
    int c = m.carried();
    int p0 = n - c;
    int v = 1 << p0;
    int acc = 0;
    for (int i = 0; i + 1 < m.ndrops; i++) {
        acc += m.drops[i];
        v |= 1 << (p0 + acc);
    }
    return v;
}
static inline int move_index(int n, const Move& m) {
    int channel;
    if (!m.spread) {
        channel = m.piece == FLAT ? 0 : m.piece == WALL ? 1 : 2;
    } else {
        int patterns = (1 << n) - 2;
        int d = m.dir == UP ? 0 : m.dir == RIGHT ? 1 : m.dir == DOWN ? 2 : 3;
        channel = 3 + (pattern_value(n, m) - 1) + patterns * d;
    }
    return channel * n * n + m.y * n + m.x;
}
static inline Move move_from_index(int n, int idx) {
    Move m;
    int nn = n * n;
    int channel = idx / nn, sq = idx % nn;
    m.y = sq / n;
    m.x = sq % n;
    if (channel < 3) {
        m.spread = 0;
        m.piece = channel == 0 ? FLAT : channel == 1 ? WALL : CAP;
        return m;
    }
    int patterns = (1 << n) - 2;
    int d = (channel - 3) / patterns;
    int v = (channel - 3) % patterns + 1;
    m.spread = 1;
    m.dir = d == 0 ? UP : d == 1 ? RIGHT : d == 2 ? DOWN : LEFT;
    int p0 = __builtin_ctz(v);
    m.ndrops = 0;
    int cur = 0;
    for (int b = p0; b < n; b++) {
        if ((v >> b) & 1) {
            if (cur) m.drops[m.ndrops++] = cur;
            cur = 1;
        } else {
            cur++;
        }
    }
    m.drops[m.ndrops++] = cur;
    return m;
}

static inline std::string move_to_ptn(const Move& m) {
    std::string s;
    if (!m.spread) {
        if (m.piece == WALL) s += 'S';
        if (m.piece == CAP) s += 'C';
        s += char('a' + m.x);
        s += char('1' + m.y);
        return s;
    }
    int c = m.carried();
    if (c != 1) s += char('0' + c);
    s += char('a' + m.x);
    s += char('1' + m.y);
    s += m.dir == UP ? '+' : m.dir == DOWN ? '-' : m.dir == LEFT ? '<' : '>';
    if (m.ndrops > 1)
        for (int i = 0; i < m.ndrops; i++) s += char('0' + m.drops[i]);
    return s;
}
static inline bool move_from_ptn(const std::string& s_in, Move& m) {
    std::string s = s_in;
    // strip trailing annotations (* ' ? !)
    while (!s.empty() && (s.back() == '*' || s.back() == '\'' || s.back() == '?' || s.back() == '!'))
        s.pop_back();
    if (s.empty()) return false;
    size_t i = 0;
    m = Move();
    int count = 0;
    if (s[i] >= '1' && s[i] <= '8') count = s[i++] - '0';
    uint8_t piece = FLAT;
    bool explicit_piece = false;
    if (i < s.size() && (s[i] == 'S' || s[i] == 'C' || s[i] == 'F')) {
        piece = s[i] == 'S' ? WALL : s[i] == 'C' ? CAP : FLAT;
        explicit_piece = true;
        i++;
    }
    if (i + 2 > s.size()) return false;
    if (s[i] < 'a' || s[i] > 'h' || s[i + 1] < '1' || s[i + 1] > '8') return false;
    m.x = s[i] - 'a';
    m.y = s[i + 1] - '1';
    i += 2;
    if (i == s.size()) {
        if (count) return false;
        m.spread = 0;
        m.piece = piece;
        return true;
    }
    if (explicit_piece) return false;
    char d = s[i++];
    if (d == '+') m.dir = UP;
    else if (d == '-') m.dir = DOWN;
    else if (d == '<') m.dir = LEFT;
    else if (d == '>') m.dir = RIGHT;
    else return false;
    m.spread = 1;
    if (!count) count = 1;
    m.ndrops = 0;
    int sum = 0;
    for (; i < s.size(); i++) {
        if (s[i] < '1' || s[i] > '8' || m.ndrops >= 8) return false;
        m.drops[m.ndrops++] = s[i] - '0';
        sum += s[i] - '0';
    }
    if (m.ndrops == 0) {
        m.drops[0] = count;
        m.ndrops = 1;
    } else if (sum != count) {
        return false;
    }
    return true;
}

struct Stack {
    std::vector<uint8_t> colors;  // bottom -> top, 0 white 1 black
    uint8_t top = NONE;
    int height() const { return (int)colors.size(); }
};

struct Game {
    int n = 5;
    int half_komi = 0;
    Stack board[TZ_MAX_N][TZ_MAX_N];  // [y][x]
    int to_move = 0;
    int stones[2] = {0, 0};
    int caps[2] = {0, 0};
    int ply = 0;
    int reversible_plies = 0;

    Game() {}
    Game(int n_, int half_komi_) : n(n_), half_komi(half_komi_) {
        default_reserves(n, stones[0], caps[0]);
        default_reserves(n, stones[1], caps[1]);
    }

    int flat_diff() const {  // white top flats - black top flats
        int d = 0;
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) {
                const Stack& s = board[y][x];
                if (s.top == FLAT) d += s.colors.back() == 0 ? 1 : -1;
            }
        return d;
    }

    // ---- move generation in fast-tak order (see header comment) ----
    void possible_moves(std::vector<Move>& out) const {
        out.clear();
        if (ply < 2) {  // opening: place the opponent's flat
            for (int x = 0; x < n; x++)
                for (int y = 0; y < n; y++)
                    if (board[y][x].top == NONE) {
                        Move m;
                        m.x = x;
                        m.y = y;
                        m.piece = FLAT;
                        out.push_back(m);
                    }
            return;
        }
        for (int x = 0; x < n; x++)
            for (int y = 0; y < n; y++) {
                const Stack& s = board[y][x];
                if (s.top == NONE) {
                    Move m;
                    m.x = x;
                    m.y = y;
                    if (stones[to_move] > 0) {
                        m.piece = FLAT;
                        out.push_back(m);
                        m.piece = WALL;
                        out.push_back(m);
                    }
                    if (caps[to_move] > 0) {
                        m.piece = CAP;
                        out.push_back(m);
                    }
                    continue;
                }
                if (s.colors.back() != to_move) continue;
                int maxc = std::min(s.height(), n);
                for (int c = 1; c <= maxc; c++)
                    for (int d = 0; d < 4; d++) gen_spreads(x, y, c, d, s.top == CAP, out);
            }
    }

    void gen_spreads(int x, int y, int c, int d, bool cap_on_top, std::vector<Move>& out) const {
        // how many squares can be entered freely, and is the next one a wall we may flatten
        int free_sq = 0;
        bool wall_next = false;
        int cx = x, cy = y;
        for (;;) {
            cx += dir_dx(d);
            cy += dir_dy(d);
            if (cx < 0 || cy < 0 || cx >= n || cy >= n) break;
            uint8_t t = board[cy][cx].top;
            if (t == CAP) break;
            if (t == WALL) {
                wall_next = true;
                break;
            }
            free_sq++;
        }
        // compositions of c in descending lexicographic order == cut masks ascending with the
        // cut after the first piece as the most significant bit.
        int ncuts = c - 1;
        for (int w = 0; w < (1 << ncuts); w++) {
            int parts = __builtin_popcount(w) + 1;
            bool ok = parts <= free_sq;
            if (!ok && wall_next && cap_on_top && parts == free_sq + 1) {
                // the final drop must be the capstone alone
                bool last_is_one = (ncuts == 0) ? (c == 1) : (w & 1);
                ok = last_is_one;
            }
            if (!ok) continue;
            Move m;
            m.spread = 1;
            m.x = x;
            m.y = y;
            m.dir = d;
            m.ndrops = 0;
            int cur = 0;
            for (int j = 1; j <= c; j++) {
                cur++;
                bool cut = j < c && ((w >> (ncuts - j)) & 1);
                if (cut || j == c) {
                    m.drops[m.ndrops++] = cur;
                    cur = 0;
                }
            }
            out.push_back(m);
        }
    }

    bool is_legal(const Move& m) const {
        std::vector<Move> mv;
        possible_moves(mv);
        for (auto& o : mv)
            if (o == m) return true;
        return false;
    }

    // ---- play (no legality check beyond asserts; callers use possible_moves) ----
    bool play(const Move& m) {
        if (m.x >= n || m.y >= n) return false;
        if (!m.spread) {
            Stack& s = board[m.y][m.x];
            if (s.top != NONE) return false;
            int color = ply < 2 ? 1 - to_move : to_move;
            if (ply < 2 && m.piece != FLAT) return false;
            if (m.piece == CAP) {
                if (caps[color] <= 0) return false;
                caps[color]--;
            } else {
                if (stones[color] <= 0) return false;
                stones[color]--;
            }
            s.colors.push_back(color);
            s.top = m.piece;
            reversible_plies = 0;
        } else {
            Stack& src = board[m.y][m.x];
            int c = m.carried();
            if (ply < 2 || src.top == NONE || src.colors.back() != to_move || c < 1 || c > n ||
                c > src.height())
                return false;
            uint8_t top_piece = src.top;
            std::vector<uint8_t> carried(src.colors.end() - c, src.colors.end());
            src.colors.resize(src.colors.size() - c);
            src.top = src.colors.empty() ? NONE : FLAT;
            int cx = m.x, cy = m.y;
            size_t pos = 0;
            bool flattened = false;
            for (int i = 0; i < m.ndrops; i++) {
                cx += dir_dx(m.dir);
                cy += dir_dy(m.dir);
                if (cx < 0 || cy < 0 || cx >= n || cy >= n) return false;
                Stack& dst = board[cy][cx];
                bool last = i == m.ndrops - 1;
                if (dst.top == CAP) return false;
                if (dst.top == WALL) {
                    if (!(last && m.drops[i] == 1 && top_piece == CAP)) return false;
                    flattened = true;
                }
                for (int k = 0; k < m.drops[i]; k++) dst.colors.push_back(carried[pos++]);
                dst.top = last ? top_piece : FLAT;
            }
            reversible_plies = flattened ? 0 : reversible_plies + 1;
        }
        ply++;
        to_move = 1 - to_move;
        return true;
    }

    bool has_road(int color) const {
        bool road[TZ_MAX_N][TZ_MAX_N];
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) {
                const Stack& s = board[y][x];
                road[y][x] = (s.top == FLAT || s.top == CAP) && s.colors.back() == color;
            }
        auto connects = [&](bool horizontal) {
            bool seen[TZ_MAX_N][TZ_MAX_N] = {};
            std::vector<std::pair<int, int>> st;
            for (int i = 0; i < n; i++) {
                int y = horizontal ? i : 0, x = horizontal ? 0 : i;
                if (road[y][x] && !seen[y][x]) {
                    seen[y][x] = true;
                    st.push_back({y, x});
                }
            }
            while (!st.empty()) {
                auto [y, x] = st.back();
                st.pop_back();
                if (horizontal ? x == n - 1 : y == n - 1) return true;
                static const int dy[4] = {1, -1, 0, 0}, dx[4] = {0, 0, 1, -1};
                for (int k = 0; k < 4; k++) {
                    int ny = y + dy[k], nx = x + dx[k];
                    if (ny < 0 || nx < 0 || ny >= n || nx >= n) continue;
                    if (road[ny][nx] && !seen[ny][nx]) {
                        seen[ny][nx] = true;
                        st.push_back({ny, nx});
                    }
                }
            }
            return false;
        };
        return connects(true) || connects(false);
    }

    Result result() const {
        // The player who just moved is checked first (a move completing both roads wins for
        // the mover), SURVEY.md B.6.
        int mover = 1 - to_move;
        if (ply > 0) {
            if (has_road(mover)) return mover == 0 ? WHITE_WINS : BLACK_WINS;
            if (has_road(1 - mover)) return mover == 0 ? BLACK_WINS : WHITE_WINS;
        }
        bool full = true;
        for (int y = 0; y < n && full; y++)
            for (int x = 0; x < n; x++)
                if (board[y][x].top == NONE) {
                    full = false;
                    break;
                }
        bool depleted = (stones[0] == 0 && caps[0] == 0) || (stones[1] == 0 && caps[1] == 0);
        if (full || depleted) {
            int w = 0, b = 0;
            for (int y = 0; y < n; y++)
                for (int x = 0; x < n; x++) {
                    const Stack& s = board[y][x];
                    if (s.top == FLAT) (s.colors.back() == 0 ? w : b)++;
                }
            int ws = 2 * w, bs = 2 * b + half_komi;
            if (ws > bs) return WHITE_WINS;
            if (bs > ws) return BLACK_WINS;
            return DRAW;
        }
        if (reversible_plies >= REVERSIBLE_PLIES_LIMIT) return DRAW;
        return ONGOING;
    }

    // why the game is over, for the PTN result of a Replay line (takparse GameResult, target.rs:226-230):
    // 0 ongoing, 1 road, 2 flat count (board full / reserves empty), 3 reversible-plies draw
    int result_reason() const {
        if (result() == ONGOING) return 0;
        if (ply > 0 && (has_road(0) || has_road(1))) return 1;
        if (reversible_plies >= REVERSIBLE_PLIES_LIMIT) {
            Game copy = *this;
            copy.reversible_plies = 0;
            if (copy.result() == ONGOING) return 3;
        }
        return 2;
    }

    // Environment::terminal, takzero/src/search/env.rs:47-59
    int terminal() const {
        Result r = result();
        if (r == ONGOING) return TZ_TERMINAL_NONE;
        if (r == DRAW) return TZ_TERMINAL_DRAW;
        int winner = r == WHITE_WINS ? 0 : 1;
        return winner == to_move ? TZ_TERMINAL_WIN : TZ_TERMINAL_LOSS;
    }

    // ---- interchange ----
    void to_state(tz_state& s) const {
        memset(&s, 0, sizeof s);
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) {
                const Stack& st = board[y][x];
                int sq = y * n + x;
                uint64_t bits = 0;
                for (size_t i = 0; i < st.colors.size(); i++)
                    if (st.colors[i]) bits |= 1ull << i;
                s.colors[sq] = bits;
                s.height[sq] = (uint8_t)st.colors.size();
                s.top[sq] = st.top;
            }
        s.stones[0] = stones[0];
        s.stones[1] = stones[1];
        s.caps[0] = caps[0];
        s.caps[1] = caps[1];
        s.to_move = to_move;
        s.n = n;
        s.half_komi = (int8_t)half_komi;
        s.ply = ply;
        s.reversible_plies = reversible_plies;
    }
    static Game from_state(const tz_state& s) {
        Game g;
        g.n = s.n;
        g.half_komi = s.half_komi;
        for (int y = 0; y < g.n; y++)
            for (int x = 0; x < g.n; x++) {
                int sq = y * g.n + x;
                Stack& st = g.board[y][x];
                st.colors.clear();
                for (int i = 0; i < s.height[sq]; i++) st.colors.push_back((s.colors[sq] >> i) & 1);
                st.top = s.top[sq];
            }
        g.stones[0] = s.stones[0];
        g.stones[1] = s.stones[1];
        g.caps[0] = s.caps[0];
        g.caps[1] = s.caps[1];
        g.to_move = s.to_move;
        g.ply = s.ply;
        g.reversible_plies = s.reversible_plies;
        return g;
    }

    // TPS: ranks top -> bottom, stacks bottom -> top, "x"/"xK" empty runs,
    // then side to move and move number.  reversible_plies is not representable (target.rs:322-326).
    std::string to_tps() const {
        std::string out;
        for (int y = n - 1; y >= 0; y--) {
            int empties = 0;
            bool first = true;
            auto flush = [&]() {
                if (!empties) return;
                if (!first) out += ',';
                out += 'x';
                if (empties > 1) out += char('0' + empties);
                empties = 0;
                first = false;
            };
            for (int x = 0; x < n; x++) {
                const Stack& s = board[y][x];
                if (s.top == NONE) {
                    empties++;
                    continue;
                }
                flush();
                if (!first) out += ',';
                first = false;
                for (uint8_t c : s.colors) out += c ? '2' : '1';
                if (s.top == WALL) out += 'S';
                if (s.top == CAP) out += 'C';
            }
            flush();
            if (y) out += '/';
        }
        out += ' ';
        out += to_move ? '2' : '1';
        out += ' ';
        out += std::to_string(ply / 2 + 1);
        return out;
    }
    static bool from_tps(const std::string& tps, int n, int half_komi, Game& g) {
        g = Game(n, half_komi);
        size_t sp = tps.find(' ');
        if (sp == std::string::npos) return false;
        std::string b = tps.substr(0, sp);
        int y = n - 1, x = 0;
        size_t i = 0;
        int placed[2] = {0, 0}, placed_caps[2] = {0, 0};
        while (i < b.size()) {
            char c = b[i];
            if (c == '/') {
                if (x != n) return false;
                y--;
                x = 0;
                i++;
                if (y < 0) return false;
            } else if (c == ',') {
                i++;
            } else if (c == 'x') {
                int k = 1;
                i++;
                if (i < b.size() && b[i] >= '1' && b[i] <= '8') k = b[i++] - '0';
                x += k;
                if (x > n) return false;
            } else if (c == '1' || c == '2') {
                if (x >= n) return false;
                Stack& s = g.board[y][x];
                while (i < b.size() && (b[i] == '1' || b[i] == '2')) s.colors.push_back(b[i++] - '1');
                s.top = FLAT;
                if (i < b.size() && b[i] == 'S') {
                    s.top = WALL;
                    i++;
                } else if (i < b.size() && b[i] == 'C') {
                    s.top = CAP;
                    i++;
                }
                for (size_t k = 0; k < s.colors.size(); k++) {
                    bool is_cap = s.top == CAP && k + 1 == s.colors.size();
                    (is_cap ? placed_caps : placed)[s.colors[k]]++;
                }
                x++;
            } else {
                return false;
            }
        }
        if (y != 0 || x != n) return false;
        int to_move_c, move_no;
        if (sscanf(tps.c_str() + sp, " %d %d", &to_move_c, &move_no) != 2) return false;
        if (to_move_c < 1 || to_move_c > 2 || move_no < 1) return false;
        g.to_move = to_move_c - 1;
        g.ply = (move_no - 1) * 2 + g.to_move;
        for (int c = 0; c < 2; c++) {
            g.stones[c] -= placed[c];
            g.caps[c] -= placed_caps[c];
            if (g.stones[c] < 0 || g.caps[c] < 0) return false;
        }
        g.reversible_plies = 0;
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// network/repr.rs
static inline int stack_size(int n) { return 3 + (n - 1) + (n + 1); }           // repr.rs:123-129
static inline int input_channels(int n) { return 2 * (stack_size(n) + 2) + 2; }  // repr.rs:137-142
static inline int output_channels(int n) { return 3 + 4 * ((1 << n) - 2); }      // repr.rs:103-109

// game_repr, takzero/src/network/repr.rs:169-228.  buffer has input_channels*n*n zeros.
static inline void game_repr(const Game& g, float* buffer) {
    int n = g.n, nn = n * n, ss = stack_size(n);
    auto index = [&](int row, int col, int channel) { return nn * channel + n * row + col; };
    auto offset = [&](int color) { return (color != g.to_move ? 1 : 0) * ss; };
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            const Stack& s = g.board[y][x];
            if (s.top == NONE) continue;
            int topc = s.colors.back();
            int ch = (s.top == FLAT ? 0 : s.top == WALL ? 1 : 2) + offset(topc);
            buffer[index(y, x, ch)] = 1.0f;
            int i = 0;
            for (int k = (int)s.colors.size() - 2; k >= 0 && i < ss - 3; k--, i++)
                buffer[index(y, x, 3 + offset(s.colors[k]) + i)] = 1.0f;
        }
    int ds, dc;
    default_reserves(n, ds, dc);
    int mine = g.to_move, other = 1 - g.to_move;
    auto ratio = [](int a, int b) -> float {
        float r = (float)a / (float)b;  // NotNan::new(..).unwrap_or_default(): 0/0 -> 0
        return r != r ? 0.0f : r;
    };
    int base = 2 * ss * nn;
    for (int i = 0; i < nn; i++) {
        buffer[base + i] = ratio(g.stones[mine], ds);
        buffer[base + nn + i] = ratio(g.caps[mine], dc);
        buffer[base + 2 * nn + i] = ratio(g.stones[other], ds);
        buffer[base + 3 * nn + i] = ratio(g.caps[other], dc);
        if (g.to_move == 1) buffer[base + 4 * nn + i] = 1.0f;
    }
    float fcd = (float)g.flat_diff() - (float)g.half_komi / 2.0f;
    float per = fcd / (float)nn;
    for (int i = 0; i < nn; i++) buffer[base + 5 * nn + i] = per;
}

// ---------------------------------------------------------------------------------------------
// Openings, takzero/src/search/env.rs:65-79.  The order of fast-tak's 8 symmetries is not
// visible in the reference (unpinned); this table is the engine's documented choice:
// index = rot*2 + mirror, rot = quarter turns counter-clockwise, mirror = flip files first.
static inline void symmetry_apply(int n, int sym, int x, int y, int& ox, int& oy) {
    if (sym & 1) x = n - 1 - x;
    int rot = (sym >> 1) & 3;
    for (int r = 0; r < rot; r++) {
        int nx = n - 1 - y, ny = x;
        x = nx;
        y = ny;
    }
    ox = x;
    oy = y;
}
// opening_choice = symmetry*2 + opposite
static inline Game new_opening(int n, int half_komi, int opening_choice) {
    Game g(n, half_komi);
    int sym = (opening_choice >> 1) & 7, opposite = opening_choice & 1;
    int sq[2][2] = {{0, 0}, {opposite ? n - 1 : 0, n - 1}};  // a1 then aN / xN  (x=file,y=rank-1)
    for (int i = 0; i < 2; i++) {
        Move m;
        int ox, oy;
        symmetry_apply(n, sym, sq[i][0], sq[i][1], ox, oy);
        m.x = ox;
        m.y = oy;
        m.piece = FLAT;
        g.play(m);
    }
    return g;
}

}  // namespace tzo
