// Test harness: csrc/tz_ot.cpp (the native reader of LibTorch model archives) compiled alone under ASan + UBSan and fed damaged
// files.  A model file is written by another process (`learn`) while this one reads it: the reader must answer TZ_EPARSE (or load
// what is still consistent), never touch memory outside the file.   ot_fuzz <file> ...   prints one status per file.
#include <cstdio>
#include <string>

#include "../takzero_amd/csrc/tz_ot.h"
#include "takzero_hip.h"

static thread_local std::string g_err;
void tz_set_error(const std::string& msg) { g_err = msg; }
int tz_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

int main(int argc, char** argv) {
    int ok = 0, parse = 0, other = 0;
    for (int i = 1; i < argc; i++) {
        TensorStore st;
        const int rc = weights_read_file(argv[i], st);
        if (rc == 0) {
            ok++;
            // what loaded must be self-consistent
            for (auto& kv : st) {
                size_t n = 1;
                for (auto d : kv.second.dims) n *= d;
                if (n != kv.second.data.size()) return 3;
            }
        } else if (rc == TZ_EPARSE) {
            parse++;
        } else {
            other++;
        }
    }
    printf("ok %d parse_errors %d other %d\n", ok, parse, other);
    return 0;
}
