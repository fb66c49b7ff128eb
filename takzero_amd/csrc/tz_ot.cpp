// tz_ot.cpp — LibTorch model archives (.ot) and the flat .tzw container, read and written without LibTorch.
// Format notes and provenance: tz_ot.h.  Replaces, at the C ABI, VarStore::{save, load, load_partial, copy} as the
// reference's Network trait uses them (takzero/src/network/mod.rs:10-45; net6_simhash.rs:152-190).
#include "tz_ot.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "tz_engine.h"

namespace {

// ------------------------------------------------------------------------------------------------ little helpers
uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t rd64(const unsigned char* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
void wr16(std::vector<unsigned char>& o, uint32_t v) {
    o.push_back(v & 0xff);
    o.push_back((v >> 8) & 0xff);
}
void wr32(std::vector<unsigned char>& o, uint32_t v) {
    wr16(o, v & 0xffff);
    wr16(o, v >> 16);
}
void wrs(std::vector<unsigned char>& o, const std::string& s) { o.insert(o.end(), s.begin(), s.end()); }

uint32_t crc32_of(const unsigned char* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

// ------------------------------------------------------------------------------------------------ zip (reader)
struct ZipEntry {
    std::string name;
    uint64_t offset = 0, csize = 0, usize = 0;  // offset of the payload
    int method = 0;
};

int zip_entries(const unsigned char* d, size_t n, std::vector<ZipEntry>& out) {
    if (n < 22) return tz_fail(TZ_EPARSE, "model archive: file too short for a zip");
    size_t eocd = (size_t)-1;
    for (size_t i = n - 22;; i--) {   // the end record carries a comment of at most 64 KiB
        if (rd32(d + i) == 0x06054b50u) {
            eocd = i;
            break;
        }
        if (i == 0 || n - 22 - i > 70000) break;
    }
    if (eocd == (size_t)-1) return tz_fail(TZ_EPARSE, "model archive: no zip end-of-central-directory record (truncated file?)");
    uint64_t count = rd16(d + eocd + 10), cd_size = rd32(d + eocd + 12), cd_off = rd32(d + eocd + 16);
    if (eocd >= 20 && rd32(d + eocd - 20) == 0x07064b50u) {   // zip64 locator in front of it (LibTorch always writes one)
        const uint64_t z64 = rd64(d + eocd - 20 + 8);
        if (z64 <= n && n - z64 >= 56 && rd32(d + z64) == 0x06064b50u) {   // (no sums of file-supplied 64-bit values: they can wrap)
            count = rd64(d + z64 + 32);
            cd_size = rd64(d + z64 + 40);
            cd_off = rd64(d + z64 + 48);
        }
    }
    if (cd_off > n || cd_size > n - cd_off) return tz_fail(TZ_EPARSE, "model archive: central directory outside the file");
    if (count > cd_size / 46) return tz_fail(TZ_EPARSE, "model archive: more central directory entries than the directory can hold");
    size_t p = cd_off;
    for (uint64_t i = 0; i < count; i++) {
        if (p > n || n - p < 46 || rd32(d + p) != 0x02014b50u) return tz_fail(TZ_EPARSE, "model archive: bad central directory entry");
        ZipEntry e;
        e.method = rd16(d + p + 10);
        e.csize = rd32(d + p + 20);
        e.usize = rd32(d + p + 24);
        const size_t fl = rd16(d + p + 28), xl = rd16(d + p + 30), cl = rd16(d + p + 32);
        uint64_t lho = rd32(d + p + 42);
        if (fl + xl + cl > n - p - 46) return tz_fail(TZ_EPARSE, "model archive: truncated central directory");
        e.name.assign((const char*)d + p + 46, fl);
        for (size_t x = p + 46 + fl; x + 4 <= p + 46 + fl + xl;) {   // zip64 extended information
            const uint16_t id = rd16(d + x), len = rd16(d + x + 2);
            if ((size_t)len > p + 46 + fl + xl - (x + 4)) return tz_fail(TZ_EPARSE, "model archive: extra field of " + e.name + " runs past its header");
            if (id == 1) {
                size_t q = x + 4;
                if (e.usize == 0xFFFFFFFFu && q + 8 <= x + 4 + len) e.usize = rd64(d + q), q += 8;
                if (e.csize == 0xFFFFFFFFu && q + 8 <= x + 4 + len) e.csize = rd64(d + q), q += 8;
                if (lho == 0xFFFFFFFFu && q + 8 <= x + 4 + len) lho = rd64(d + q);
            }
            x += 4 + (size_t)len;
        }
        if (lho > n || n - lho < 30 || rd32(d + lho) != 0x04034b50u) return tz_fail(TZ_EPARSE, "model archive: bad local header of " + e.name);
        e.offset = lho + 30 + rd16(d + lho + 26) + rd16(d + lho + 28);
        if (e.offset > n || e.csize > n - e.offset) return tz_fail(TZ_EPARSE, "model archive: entry " + e.name + " runs past the end of the file");
        if (e.method == 0 && e.usize != e.csize) return tz_fail(TZ_EPARSE, "model archive: stored entry " + e.name + " with two different sizes");
        out.push_back(e);
        p += 46 + fl + xl + cl;
    }
    return TZ_OK;
}

// ------------------------------------------------------------------------------------------------ pickle (reader)
// Values of the subset a tensor archive needs.  Tuples, lists and dict items share `items` (dict: key, value, key, ...).
struct PV {
    enum Kind { NONE, BOOL, INT, FLOAT, STR, TUPLE, LIST, DICT, GLOBAL, OBJECT, STORAGE, TENSOR, MARK } kind = NONE;
    int64_t i = 0;
    double f = 0;
    std::string s;                         // STR; GLOBAL "module name"; STORAGE key
    std::vector<std::shared_ptr<PV>> items;
    // STORAGE: s = key, dtype in `i` (0 f32, 1 f64, 2 f16, 3 bf16), numel in `f`.  TENSOR: items[0] = storage, offset i, dims / strides
    std::vector<int64_t> dims, strides;
    std::shared_ptr<PV> state;             // OBJECT after BUILD
};
typedef std::shared_ptr<PV> P;
P mk(PV::Kind k) {
    P p = std::make_shared<PV>();
    p->kind = k;
    return p;
}

int storage_dtype(const std::string& global) {
    if (global == "torch FloatStorage") return 0;
    if (global == "torch DoubleStorage") return 1;
    if (global == "torch HalfStorage") return 2;
    if (global == "torch BFloat16Storage") return 3;
    return -1;
}

int unpickle(const unsigned char* d, size_t n, P& root) {
    std::vector<P> st;
    std::map<uint32_t, P> memo;
    size_t p = 0;
    auto need = [&](size_t k) { return k <= n - p; };   // p <= n always; a length from the file may be anything up to 2^64 - 1
    auto pop_to_mark = [&](std::vector<P>& out) -> bool {
        size_t m = st.size();
        while (m > 0 && st[m - 1]->kind != PV::MARK) m--;
        if (m == 0) return false;
        out.assign(st.begin() + m, st.end());
        st.resize(m - 1);
        return true;
    };
#define PK_FAIL(msg) return tz_fail(TZ_EPARSE, std::string("model archive: data.pkl: ") + msg)
    while (p < n) {
        const unsigned char op = d[p++];
        switch (op) {
            case 0x80: if (!need(1)) PK_FAIL("truncated"); p += 1; break;                 // PROTO
            case 0x95: if (!need(8)) PK_FAIL("truncated"); p += 8; break;                 // FRAME
            case '.': if (st.empty()) PK_FAIL("empty stack at STOP"); root = st.back(); return TZ_OK;
            case '(': st.push_back(mk(PV::MARK)); break;
            case 'N': st.push_back(mk(PV::NONE)); break;
            case 0x88: case 0x89: { P v = mk(PV::BOOL); v->i = op == 0x88; st.push_back(v); break; }
            case 'K': { if (!need(1)) PK_FAIL("truncated"); P v = mk(PV::INT); v->i = d[p]; p += 1; st.push_back(v); break; }
            case 'M': { if (!need(2)) PK_FAIL("truncated"); P v = mk(PV::INT); v->i = rd16(d + p); p += 2; st.push_back(v); break; }
            case 'J': { if (!need(4)) PK_FAIL("truncated"); P v = mk(PV::INT); v->i = (int32_t)rd32(d + p); p += 4; st.push_back(v); break; }
            case 0x8a: {  // LONG1
                if (!need(1)) PK_FAIL("truncated");
                const size_t len = d[p++];
                if (!need(len) || len > 8) PK_FAIL("unsupported LONG1");
                uint64_t u = 0;
                for (size_t k = 0; k < len; k++) u |= (uint64_t)d[p + k] << (8 * k);
                if (len && len < 8 && (d[p + len - 1] & 0x80)) u |= ~0ull << (8 * len);
                p += len;
                P v = mk(PV::INT);
                v->i = (int64_t)u;
                st.push_back(v);
                break;
            }
            case 'G': {  // BINFLOAT, big endian
                if (!need(8)) PK_FAIL("truncated");
                uint64_t u = 0;
                for (int k = 0; k < 8; k++) u = (u << 8) | d[p + k];
                p += 8;
                P v = mk(PV::FLOAT);
                memcpy(&v->f, &u, 8);
                st.push_back(v);
                break;
            }
            case 'X': case 'T': case 0x8c: case 'U': case 0x8d: {  // BINUNICODE, BINSTRING, SHORT_BINUNICODE, SHORT_BINSTRING, BINUNICODE8
                size_t len;
                if (op == 0x8c || op == 'U') { if (!need(1)) PK_FAIL("truncated"); len = d[p]; p += 1; }
                else if (op == 0x8d) { if (!need(8)) PK_FAIL("truncated"); len = rd64(d + p); p += 8; }
                else { if (!need(4)) PK_FAIL("truncated"); len = rd32(d + p); p += 4; }
                if (!need(len)) PK_FAIL("truncated string");
                P v = mk(PV::STR);
                v->s.assign((const char*)d + p, len);
                p += len;
                st.push_back(v);
                break;
            }
            case 'c': {  // GLOBAL "module\nname\n"
                std::string mod, name;
                while (p < n && d[p] != '\n') mod.push_back((char)d[p++]);
                p++;
                while (p < n && d[p] != '\n') name.push_back((char)d[p++]);
                p++;
                if (p > n) PK_FAIL("truncated GLOBAL");
                P v = mk(PV::GLOBAL);
                v->s = mod + " " + name;
                st.push_back(v);
                break;
            }
            case 0x93: {  // STACK_GLOBAL
                if (st.size() < 2 || st[st.size() - 1]->kind != PV::STR || st[st.size() - 2]->kind != PV::STR) PK_FAIL("bad STACK_GLOBAL");
                P v = mk(PV::GLOBAL);
                v->s = st[st.size() - 2]->s + " " + st[st.size() - 1]->s;
                st.resize(st.size() - 2);
                st.push_back(v);
                break;
            }
            case 'q': { if (!need(1) || st.empty()) PK_FAIL("bad BINPUT"); memo[d[p]] = st.back(); p += 1; break; }
            case 'r': { if (!need(4) || st.empty()) PK_FAIL("bad LONG_BINPUT"); memo[rd32(d + p)] = st.back(); p += 4; break; }
            case 0x94: { if (st.empty()) PK_FAIL("bad MEMOIZE"); const uint32_t k = (uint32_t)memo.size(); memo[k] = st.back(); break; }
            case 'h': case 'j': {
                uint32_t k;
                if (op == 'h') { if (!need(1)) PK_FAIL("truncated"); k = d[p]; p += 1; }
                else { if (!need(4)) PK_FAIL("truncated"); k = rd32(d + p); p += 4; }
                auto it = memo.find(k);
                if (it == memo.end()) PK_FAIL("BINGET of an unknown memo slot");
                st.push_back(it->second);
                break;
            }
            case ')': st.push_back(mk(PV::TUPLE)); break;
            case '}': st.push_back(mk(PV::DICT)); break;
            case ']': st.push_back(mk(PV::LIST)); break;
            case 0x85: case 0x86: case 0x87: {
                const size_t k = op - 0x84;
                if (st.size() < k) PK_FAIL("stack underflow in TUPLEn");
                P v = mk(PV::TUPLE);
                v->items.assign(st.end() - k, st.end());
                st.resize(st.size() - k);
                st.push_back(v);
                break;
            }
            case 't': {
                P v = mk(PV::TUPLE);
                if (!pop_to_mark(v->items)) PK_FAIL("TUPLE without MARK");
                st.push_back(v);
                break;
            }
            case 'l': {
                P v = mk(PV::LIST);
                if (!pop_to_mark(v->items)) PK_FAIL("LIST without MARK");
                st.push_back(v);
                break;
            }
            case 'a': {
                if (st.size() < 2 || st[st.size() - 2]->kind != PV::LIST) PK_FAIL("bad APPEND");
                st[st.size() - 2]->items.push_back(st.back());
                st.pop_back();
                break;
            }
            case 'e': {
                std::vector<P> it;
                if (!pop_to_mark(it) || st.empty() || st.back()->kind != PV::LIST) PK_FAIL("bad APPENDS");
                st.back()->items.insert(st.back()->items.end(), it.begin(), it.end());
                break;
            }
            case 's': {
                if (st.size() < 3 || st[st.size() - 3]->kind != PV::DICT) PK_FAIL("bad SETITEM");
                st[st.size() - 3]->items.push_back(st[st.size() - 2]);
                st[st.size() - 3]->items.push_back(st[st.size() - 1]);
                st.resize(st.size() - 2);
                break;
            }
            case 'u': {
                std::vector<P> it;
                if (!pop_to_mark(it) || st.empty() || st.back()->kind != PV::DICT || (it.size() & 1)) PK_FAIL("bad SETITEMS");
                st.back()->items.insert(st.back()->items.end(), it.begin(), it.end());
                break;
            }
            case 0x81: {  // NEWOBJ: cls, args -> object
                if (st.size() < 2) PK_FAIL("bad NEWOBJ");
                P v = mk(PV::OBJECT);
                v->s = st[st.size() - 2]->s;
                st.resize(st.size() - 2);
                st.push_back(v);
                break;
            }
            case 'b': {  // BUILD: object, state
                if (st.size() < 2) PK_FAIL("bad BUILD");
                P state = st.back();
                st.pop_back();
                if (st.back()->kind == PV::OBJECT) st.back()->state = state;
                else if (st.back()->kind == PV::DICT && state->kind == PV::DICT)
                    st.back()->items.insert(st.back()->items.end(), state->items.begin(), state->items.end());
                break;
            }
            case 'Q': {  // BINPERSID: ('storage', <StorageType>, key, location, numel)
                if (st.empty()) PK_FAIL("bad BINPERSID");
                P t = st.back();
                st.pop_back();
                if (t->kind != PV::TUPLE || t->items.size() < 5 || t->items[0]->kind != PV::STR || t->items[0]->s != "storage" ||
                    t->items[1]->kind != PV::GLOBAL || t->items[2]->kind != PV::STR || t->items[4]->kind != PV::INT)
                    PK_FAIL("persistent id is not a storage tuple");
                const int dt = storage_dtype(t->items[1]->s);
                if (dt < 0) PK_FAIL("unsupported storage type " + t->items[1]->s);
                P v = mk(PV::STORAGE);
                v->s = t->items[2]->s;
                v->i = dt;
                v->f = (double)t->items[4]->i;
                st.push_back(v);
                break;
            }
            case 'R': {  // REDUCE: callable, args
                if (st.size() < 2 || st.back()->kind != PV::TUPLE) PK_FAIL("bad REDUCE");
                P args = st.back();
                st.pop_back();
                P fn = st.back();
                st.pop_back();
                if (fn->kind != PV::GLOBAL) PK_FAIL("REDUCE of a non-global");
                if (fn->s == "torch._utils _rebuild_tensor_v2" || fn->s == "torch._utils _rebuild_tensor") {
                    if (args->items.size() < 4 || args->items[0]->kind != PV::STORAGE || args->items[1]->kind != PV::INT ||
                        args->items[2]->kind != PV::TUPLE || args->items[3]->kind != PV::TUPLE)
                        PK_FAIL("unexpected arguments of _rebuild_tensor_v2");
                    P v = mk(PV::TENSOR);
                    v->items.push_back(args->items[0]);
                    v->i = args->items[1]->i;
                    for (auto& x : args->items[2]->items) v->dims.push_back(x->i);
                    for (auto& x : args->items[3]->items) v->strides.push_back(x->i);
                    if (v->dims.size() != v->strides.size()) PK_FAIL("tensor size / stride mismatch");
                    st.push_back(v);
                } else if (fn->s == "torch._utils _rebuild_parameter") {   // Parameter(tensor, requires_grad, hooks)
                    if (args->items.empty() || args->items[0]->kind != PV::TENSOR) PK_FAIL("unexpected arguments of _rebuild_parameter");
                    st.push_back(args->items[0]);
                } else if (fn->s == "collections OrderedDict") {
                    st.push_back(mk(PV::DICT));
                } else {
                    PK_FAIL("unsupported callable " + fn->s);
                }
                break;
            }
            default: {
                char b[8];
                snprintf(b, sizeof b, "0x%02x", op);
                PK_FAIL(std::string("unsupported opcode ") + b);
            }
        }
    }
    PK_FAIL("no STOP");
#undef PK_FAIL
}

float half_to_float(uint16_t h) {
    const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s << 31;
        else {
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 1024)) mm <<= 1, sh++;
            u = (s << 31) | ((uint32_t)(113 - sh) << 23) | ((mm & 1023) << 13);
        }
    } else if (e == 31) u = (s << 31) | 0x7f800000u | (m << 13);
    else u = (s << 31) | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

bool ends_with(const std::string& s, const std::string& t) { return s.size() >= t.size() && !s.compare(s.size() - t.size(), t.size(), t); }

// ------------------------------------------------------------------------------------------------ pickle + zip (writer)
struct Pickler {   // emits what LibTorch's Pickler emits for a module of parameters (memo layout included)
    std::vector<unsigned char> o;
    uint32_t memo = 0;
    void put() {
        if (memo < 256) { o.push_back('q'); o.push_back((unsigned char)memo); }
        else { o.push_back('r'); wr32(o, memo); }
        memo++;
    }
    void get(uint32_t k) {
        if (k < 256) { o.push_back('h'); o.push_back((unsigned char)k); }
        else { o.push_back('j'); wr32(o, k); }
    }
    void str(const std::string& s) { o.push_back('X'); wr32(o, (uint32_t)s.size()); wrs(o, s); }
    void global(const std::string& mod, const std::string& name) { o.push_back('c'); wrs(o, mod + "\n" + name + "\n"); }
    void integer(int64_t v) {
        if (v >= 0 && v < 256) { o.push_back('K'); o.push_back((unsigned char)v); }
        else if (v >= 0 && v < 65536) { o.push_back('M'); wr16(o, (uint32_t)v); }
        else if (v >= INT32_MIN && v <= INT32_MAX) { o.push_back('J'); wr32(o, (uint32_t)(int32_t)v); }
        else { o.push_back(0x8a); o.push_back(8); for (int k = 0; k < 8; k++) o.push_back((unsigned char)((uint64_t)v >> (8 * k))); }
    }
};

struct ZipWriter {
    std::vector<unsigned char>& o;
    struct Rec { std::string name; uint32_t crc, size; uint64_t lho; };
    std::vector<Rec> recs;
    explicit ZipWriter(std::vector<unsigned char>& out) : o(out) {}
    void add(const std::string& name, const unsigned char* data, size_t n) {
        // payload aligned to 64 bytes with an "FB" padding extra field, as PyTorchStreamWriter does (mmap-friendly)
        const size_t base = o.size() + 30 + name.size() + 4;
        const size_t pad = (64 - base % 64) % 64;
        Rec r{name, crc32_of(data, n), (uint32_t)n, o.size()};
        wr32(o, 0x04034b50u);
        wr16(o, 20);       // version needed
        wr16(o, 0x0800);   // UTF-8 names
        wr16(o, 0);        // stored
        wr16(o, 0);
        wr16(o, 0);        // time, date
        wr32(o, r.crc);
        wr32(o, r.size);
        wr32(o, r.size);
        wr16(o, (uint32_t)name.size());
        wr16(o, (uint32_t)(4 + pad));
        wrs(o, name);
        o.push_back('F');
        o.push_back('B');
        wr16(o, (uint32_t)pad);
        o.insert(o.end(), pad, 'Z');
        o.insert(o.end(), data, data + n);
        recs.push_back(r);
    }
    void finish() {
        const size_t cd = o.size();
        for (auto& r : recs) {
            wr32(o, 0x02014b50u);
            wr16(o, 20);
            wr16(o, 20);
            wr16(o, 0x0800);
            wr16(o, 0);
            wr16(o, 0);
            wr16(o, 0);
            wr32(o, r.crc);
            wr32(o, r.size);
            wr32(o, r.size);
            wr16(o, (uint32_t)r.name.size());
            wr16(o, 0);
            wr16(o, 0);
            wr16(o, 0);
            wr16(o, 0);
            wr32(o, 0);
            wr32(o, (uint32_t)r.lho);
            wrs(o, r.name);
        }
        const size_t cd_size = o.size() - cd;
        wr32(o, 0x06054b50u);
        wr16(o, 0);
        wr16(o, 0);
        wr16(o, (uint32_t)recs.size());
        wr16(o, (uint32_t)recs.size());
        wr32(o, (uint32_t)cd_size);
        wr32(o, (uint32_t)cd);
        wr16(o, 0);
    }
};

bool bn_stats_first() {   // see ot_tch_names
    const char* e = getenv("TZ_TCH_BN_ORDER");
    return e && !strcmp(e, "stats_first");
}

}  // namespace

// ------------------------------------------------------------------------------------------------ API
static int ot_read_archive_impl(const unsigned char* data, size_t bytes, NamedTensors& out) {
    std::vector<ZipEntry> entries;
    int rc = zip_entries(data, bytes, entries);
    if (rc) return rc;
    const ZipEntry* pkl = nullptr;
    for (auto& e : entries)
        if (ends_with(e.name, "/data.pkl") || e.name == "data.pkl") pkl = &e;
    if (!pkl) return tz_fail(TZ_EPARSE, "model archive: no data.pkl (not a LibTorch archive)");
    if (pkl->method != 0) return tz_fail(TZ_EPARSE, "model archive: data.pkl is compressed (LibTorch stores it)");
    const std::string prefix = pkl->name.substr(0, pkl->name.size() - 8);   // "<stem>/"
    P root;
    if ((rc = unpickle(data + pkl->offset, pkl->usize, root))) return rc;
    // a module object (OutputArchive / torch.jit.save) carries its attributes as BUILD state; a plain dict (torch.save of a
    // state_dict) is accepted too
    P dict = root;
    if (root->kind == PV::OBJECT) dict = root->state;
    if (!dict || dict->kind != PV::DICT) return tz_fail(TZ_EPARSE, "model archive: data.pkl does not hold a module or a dict of tensors");
    std::map<std::string, const ZipEntry*> by_name;
    for (auto& e : entries) by_name[e.name] = &e;
    for (size_t i = 0; i + 1 < dict->items.size(); i += 2) {
        const P& k = dict->items[i];
        const P& v = dict->items[i + 1];
        if (k->kind != PV::STR || v->kind != PV::TENSOR) continue;   // e.g. `training`, `_is_full_backward_hook`
        const PV& stor = *v->items[0];
        auto it = by_name.find(prefix + "data/" + stor.s);
        if (it == by_name.end()) return tz_fail(TZ_EPARSE, "model archive: storage " + stor.s + " of " + k->s + " is missing");
        const ZipEntry& e = *it->second;
        if (e.method != 0) return tz_fail(TZ_EPARSE, "model archive: storage " + stor.s + " is compressed");
        const size_t esz = stor.i == 0 ? 4 : stor.i == 1 ? 8 : 2;
        const uint64_t numel_storage = e.usize / esz;
        HostTensor t;
        uint64_t total = 1;
        for (auto dv : v->dims) {
            if (dv < 0) return tz_fail(TZ_EPARSE, "model archive: negative dimension in " + k->s);
            t.dims.push_back((uint32_t)dv);
            if (dv != 0 && total > (numel_storage + 1) / (uint64_t)dv + 1) return tz_fail(TZ_EPARSE, "model archive: tensor " + k->s + " is larger than its storage");
            total *= (uint64_t)dv;
        }
        // a variable is a view of its own storage: more elements than the storage holds means a damaged size tuple (and must not
        // become an allocation)
        if (total > numel_storage) return tz_fail(TZ_EPARSE, "model archive: tensor " + k->s + " is larger than its storage");
        t.data.resize(total);
        const unsigned char* base = data + e.offset;
        std::vector<int64_t> idx(v->dims.size(), 0);
        for (uint64_t flat = 0; flat < total; flat++) {
            int64_t off = v->i;
            for (size_t a = 0; a < idx.size(); a++) off += idx[a] * v->strides[a];
            if (off < 0 || (uint64_t)off >= numel_storage) return tz_fail(TZ_EPARSE, "model archive: tensor " + k->s + " reads outside its storage");
            float f;
            if (stor.i == 0) memcpy(&f, base + 4 * off, 4);
            else if (stor.i == 1) { double dd; memcpy(&dd, base + 8 * off, 8); f = (float)dd; }
            else if (stor.i == 2) f = half_to_float(rd16(base + 2 * off));
            else { const uint32_t u = (uint32_t)rd16(base + 2 * off) << 16; memcpy(&f, &u, 4); }
            t.data[flat] = f;
            for (size_t a = idx.size(); a-- > 0;) {
                if (++idx[a] < v->dims[a]) break;
                idx[a] = 0;
            }
        }
        out.emplace_back(k->s, std::move(t));
    }
    if (out.empty()) return tz_fail(TZ_EPARSE, "model archive: no tensors in data.pkl");
    return TZ_OK;
}

int ot_write_archive(const NamedTensors& tensors, const std::string& stem, std::vector<unsigned char>& out) {
    // data.pkl: __torch__.Module object whose state maps attribute name -> _rebuild_tensor_v2(storage k, 0, size, stride, False, OrderedDict())
    Pickler pk;
    pk.o.push_back(0x80);
    pk.o.push_back(2);
    pk.global("__torch__", "Module");
    pk.put();                       // 0
    pk.o.push_back(')');
    pk.o.push_back(0x81);
    pk.o.push_back('}');
    pk.o.push_back('(');
    uint32_t m_rebuild = 0, m_storage = 0, m_float = 0, m_cpu = 0, m_odict = 0;
    for (size_t i = 0; i < tensors.size(); i++) {
        const HostTensor& t = tensors[i].second;
        uint64_t numel = 1;
        for (auto d : t.dims) numel *= d;
        pk.str(tensors[i].first);
        pk.put();
        if (i == 0) { pk.global("torch._utils", "_rebuild_tensor_v2"); m_rebuild = pk.memo; pk.put(); }
        else pk.get(m_rebuild);
        pk.o.push_back('(');
        pk.o.push_back('(');
        if (i == 0) {
            pk.str("storage"); m_storage = pk.memo; pk.put();
            pk.global("torch", "FloatStorage"); m_float = pk.memo; pk.put();
        } else {
            pk.get(m_storage);
            pk.get(m_float);
        }
        pk.str(std::to_string(i));
        pk.put();
        if (i == 0) { pk.str("cpu"); m_cpu = pk.memo; pk.put(); }
        else pk.get(m_cpu);
        pk.integer((int64_t)numel);
        pk.o.push_back('t');
        pk.o.push_back('Q');
        pk.put();
        pk.integer(0);
        pk.o.push_back('(');
        for (auto d : t.dims) pk.integer(d);
        pk.o.push_back('t');
        pk.o.push_back('(');
        for (size_t a = 0; a < t.dims.size(); a++) {
            uint64_t s = 1;
            for (size_t b = a + 1; b < t.dims.size(); b++) s *= t.dims[b];
            pk.integer((int64_t)s);
        }
        pk.o.push_back('t');
        pk.o.push_back(0x89);
        if (i == 0) { pk.global("collections", "OrderedDict"); m_odict = pk.memo; pk.put(); }
        else pk.get(m_odict);
        pk.o.push_back(')');
        pk.o.push_back('R');
        pk.o.push_back('t');
        pk.o.push_back('R');
    }
    pk.o.push_back('u');
    pk.o.push_back('b');
    pk.put();
    pk.o.push_back('.');
    // code/__torch__.py: the module class with its parameter list (what PythonPrint emits for a module without methods)
    std::string code = "class Module(Module):\n  __parameters__ = [";
    for (auto& t : tensors) code += "\"" + t.first + "\", ";
    code += "]\n  __buffers__ = []\n  __annotations__ = []\n";
    for (auto& t : tensors) {   // PythonPrint: a name that is a valid identifier is declared `name : Tensor`
        bool ident = !t.first.empty() && !(t.first[0] >= '0' && t.first[0] <= '9');
        for (char c : t.first) ident = ident && ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_');
        code += ident ? "  " + t.first + " : Tensor\n" : "  __annotations__[\"" + t.first + "\"] = Tensor\n";
    }
    out.clear();
    ZipWriter z(out);
    for (size_t i = 0; i < tensors.size(); i++) {
        const auto& dv = tensors[i].second.data;
        z.add(stem + "/data/" + std::to_string(i), reinterpret_cast<const unsigned char*>(dv.data()), dv.size() * 4);
    }
    z.add(stem + "/data.pkl", pk.o.data(), pk.o.size());
    z.add(stem + "/code/__torch__.py", reinterpret_cast<const unsigned char*>(code.data()), code.size());
    const unsigned char constants[] = {0x80, 2, ')', '.'};
    z.add(stem + "/constants.pkl", constants, sizeof constants);
    z.add(stem + "/version", reinterpret_cast<const unsigned char*>("3\n"), 2);
    z.add(stem + "/byteorder", reinterpret_cast<const unsigned char*>("little"), 6);
    z.finish();
    return TZ_OK;
}

int ot_canonical_names(const NamedTensors& in, TensorStore& out) {
    for (auto& kv : in) {
        std::string base = kv.first;
        bool dup = false;
        const size_t us = base.rfind("__");
        if (us != std::string::npos && us + 2 < base.size()) {
            bool digits = true;
            for (size_t i = us + 2; i < base.size(); i++) digits = digits && base[i] >= '0' && base[i] <= '9';
            if (digits) {
                base = base.substr(0, us);
                dup = true;
            }
        }
        const std::string pre = "core.res_block_";
        if (!base.compare(0, pre.size(), pre)) {
            const size_t dot = base.find('.', pre.size());
            if (dot != std::string::npos) {
                out[base.substr(0, dot) + (dup ? ".b." : ".a.") + base.substr(dot + 1)] = kv.second;
                continue;
            }
        }
        if (dup) return tz_fail(TZ_EPARSE, "model archive: duplicated variable outside a residual block: " + kv.first);
        out[base] = kv.second;
    }
    return TZ_OK;
}

void ot_tch_names(const TensorStore& in, NamedTensors& out) {
    // Creation order of Net::new (net5.rs:152-170, net6_simhash.rs:122-141): core (input conv, batch norm, blocks), policy,
    // value, ube, then the uncertainty side nets.  A path that already exists gets `__<variables registered so far>`
    // (tch nn::Path::add).  Inside nn::batch_norm, tch creates weight and bias (the `affine` pair) before running_mean and
    // running_var; TZ_TCH_BN_ORDER=stats_first gives the older order.  Neither is pinned by a file of the reference (none is
    // available, *.ot is git-ignored there): readers on this side do not depend on K, and the `.a.` / `.b.` mapping ignores it.
    std::map<std::string, bool> seen;
    size_t count = 0;
    auto add = [&](const std::string& name, const std::string& key) {
        auto it = in.find(key);
        if (it == in.end()) return;
        std::string final_name = name;
        if (seen.count(name)) final_name = name + "__" + std::to_string(count);
        seen[final_name] = true;
        out.emplace_back(final_name, it->second);
        count++;
    };
    const bool stats_first = bn_stats_first();
    auto bn = [&](const std::string& path, const std::string& key) {
        const char* a[4] = {"weight", "bias", "running_mean", "running_var"};
        const char* b[4] = {"running_mean", "running_var", "weight", "bias"};
        for (int i = 0; i < 4; i++) {
            const char* v = stats_first ? b[i] : a[i];
            add(path + "." + v, key + "." + v);
        }
    };
    add("core.input_conv2d.weight", "core.input_conv2d.weight");
    bn("core.batch_norm", "core.batch_norm");
    for (int blk = 0;; blk++) {
        const std::string p = "core.res_block_" + std::to_string(blk);
        if (!in.count(p + ".a.conv2d.weight")) break;
        for (const char* half : {".a", ".b"}) {
            add(p + ".conv2d.weight", p + half + ".conv2d.weight");
            bn(p + ".batch_norm", p + half + ".batch_norm");
        }
    }
    const char* rest[] = {"policy.conv2d.weight", "policy.conv2d.bias", "value.conv2d.weight", "value.conv2d.bias", "value.linear.weight",
                          "value.linear.bias", "ube.conv2d.weight", "ube.conv2d.bias", "ube.linear.weight", "ube.linear.bias",
                          "rnd_learning.input_linear.weight", "rnd_learning.input_linear.bias", "rnd_learning.hidden_linear.weight",
                          "rnd_learning.hidden_linear.bias", "rnd_learning.final_linear.weight", "rnd_learning.final_linear.bias",
                          "rnd_target.input_linear.weight", "rnd_target.input_linear.bias", "rnd_target.hidden_linear.weight",
                          "rnd_target.hidden_linear.bias", "rnd_target.final_linear.weight", "rnd_target.final_linear.bias", "min", "max",
                          "simhash_matrix"};
    std::map<std::string, bool> done;
    for (auto& kv : out) done[kv.first] = true;
    for (const char* r : rest) {
        add(r, r);
        done[r] = true;
    }
    for (auto& kv : in)   // anything else the store holds, in name order
        if (kv.first.compare(0, 5, "core.") && !done.count(kv.first)) add(kv.first, kv.first);
}

static int tzw_parse_impl(const unsigned char* p, size_t bytes, TensorStore& out) {
    if (bytes < 8 || memcmp(p, "TZW1", 4)) return tz_fail(TZ_EPARSE, "weights: bad magic");
    const uint32_t count = rd32(p + 4);
    size_t off = 8;
    for (uint32_t i = 0; i < count; i++) {
        if (off > bytes || bytes - off < 2) return tz_fail(TZ_EPARSE, "weights: truncated");
        const uint16_t ln = rd16(p + off);
        off += 2;
        if ((size_t)ln + 1 > bytes - off) return tz_fail(TZ_EPARSE, "weights: truncated");
        std::string name((const char*)p + off, ln);
        off += ln;
        const int nd = p[off++];
        HostTensor t;
        size_t size = 1;
        if (4 * (size_t)nd > bytes - off) return tz_fail(TZ_EPARSE, "weights: truncated");
        for (int d = 0; d < nd; d++) {
            t.dims.push_back(rd32(p + off));
            // the product may not pass what the rest of the blob can hold (and so cannot wrap either)
            if (t.dims.back() != 0 && size > (bytes / 4) / t.dims.back()) return tz_fail(TZ_EPARSE, "weights: tensor " + name + " larger than the blob");
            size *= t.dims.back();
            off += 4;
        }
        if (size > (bytes - off) / 4) return tz_fail(TZ_EPARSE, "weights: truncated");
        t.data.resize(size);
        memcpy(t.data.data(), p + off, 4 * size);
        off += 4 * size;
        out[name] = std::move(t);
    }
    return TZ_OK;
}

// The parsers are reached from the C ABI (tz_net_load_weights, tz_net_load_weights_mem, tz_net_broadcast's blobs): nothing a file
// can hold may leave them as an exception - a length the checks above let through that still cannot be allocated is TZ_ENOMEM, not
// a terminate() across extern "C".
template <typename F>
static int guarded(const char* what, F f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return tz_fail(TZ_ENOMEM, std::string(what) + ": out of memory");
    } catch (const std::exception& e) {
        return tz_fail(TZ_EPARSE, std::string(what) + ": " + e.what());
    }
}
int ot_read_archive(const unsigned char* data, size_t bytes, NamedTensors& out) {
    return guarded("model archive", [&] { return ot_read_archive_impl(data, bytes, out); });
}
int tzw_parse(const unsigned char* p, size_t bytes, TensorStore& out) {
    return guarded("weights", [&] { return tzw_parse_impl(p, bytes, out); });
}

void tzw_dump(const TensorStore& in, std::vector<unsigned char>& o) {
    o.clear();
    wrs(o, "TZW1");
    wr32(o, (uint32_t)in.size());
    for (auto& kv : in) {
        wr16(o, (uint32_t)kv.first.size());
        wrs(o, kv.first);
        o.push_back((unsigned char)kv.second.dims.size());
        for (auto d : kv.second.dims) wr32(o, d);
        const unsigned char* b = reinterpret_cast<const unsigned char*>(kv.second.data.data());
        o.insert(o.end(), b, b + 4 * kv.second.data.size());
    }
}

int weights_read_file(const char* path, TensorStore& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return tz_fail(TZ_EPARSE, std::string("weights: cannot open ") + path);
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> buf(sz > 0 ? sz : 0);
    const size_t rd = sz > 0 ? fread(buf.data(), 1, sz, f) : 0;
    fclose(f);
    if ((long)rd != sz) return tz_fail(TZ_EPARSE, std::string("weights: short read of ") + path);
    if (buf.size() >= 4 && !memcmp(buf.data(), "TZW1", 4)) return tzw_parse(buf.data(), buf.size(), out);
    if (buf.size() >= 4 && rd32(buf.data()) == 0x04034b50u) {
        NamedTensors named;
        int rc = ot_read_archive(buf.data(), buf.size(), named);
        if (rc) return rc;
        return ot_canonical_names(named, out);
    }
    return tz_fail(TZ_EPARSE, std::string("weights: ") + path + " is neither a LibTorch archive (.ot) nor a .tzw container");
}

int weights_write_file(const char* path, const TensorStore& in) {
    const std::string p(path), part = p + ".part";
    std::vector<unsigned char> bytes;
    if (ends_with(p, ".tzw")) {
        tzw_dump(in, bytes);
    } else {
        NamedTensors named;
        ot_tch_names(in, named);
        // the archive's directory name is the file name without its extension, as PyTorchStreamWriter derives it
        std::string stem = p.substr(p.find_last_of('/') == std::string::npos ? 0 : p.find_last_of('/') + 1);
        const size_t dot = stem.find_last_of('.');
        if (dot != std::string::npos && dot > 0) stem = stem.substr(0, dot);
        int rc = ot_write_archive(named, stem, bytes);
        if (rc) return rc;
    }
    FILE* f = fopen(part.c_str(), "wb");
    if (!f) return tz_fail(TZ_EINVAL, "weights: cannot create " + part);
    const size_t wr = fwrite(bytes.data(), 1, bytes.size(), f);
    if (fclose(f) != 0 || wr != bytes.size()) {
        remove(part.c_str());
        return tz_fail(TZ_EINVAL, "weights: short write to " + part);
    }
    if (rename(part.c_str(), p.c_str()) != 0) return tz_fail(TZ_EINVAL, "weights: cannot rename " + part);
    return TZ_OK;
}
