// Sanitizer fuzz of the native text entry points (tz_parse_targets / tz_format_targets, host code only):
// mutated target lines in exact-size heap buffers under AddressSanitizer + UBSan.  Built and run by tests/test_fuzz_text.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "takzero_hip.h"
int main(int argc, char** argv) {
    const int iterations = argc > 1 ? atoi(argv[1]) : 20000;
    const int n = 5, amax = 512, maxt = 64;
    std::vector<tz_state> st(maxt);
    std::vector<uint16_t> mv(maxt * amax);
    std::vector<float> pol(maxt * amax), val(maxt), ube(maxt);
    std::vector<int32_t> nm(maxt);
    std::string base = "x2,1221,x,1S/2,2C,2,1,x/x,212,21C,2S,2/2211S,2,21,1,1/x2,221S,2,x 2 23;0.5;1.25;a1:0.25,Sb2:0.5,3c3>12:0.25\n";
    std::mt19937 rng(1);
    long parsed = 0, skipped_total = 0;
    for (int it = 0; it < iterations; it++) {
        std::string s;
        int lines = 1 + rng() % 3;
        for (int l = 0; l < lines; l++) {
            std::string x = base;
            int muts = rng() % 4;
            for (int m = 0; m < muts; m++) {
                size_t p = rng() % x.size();
                switch (rng() % 4) {
                    case 0: x[p] = (char)(rng() % 256); break;
                    case 1: x.erase(p, 1 + rng() % 5); break;
                    case 2: x.insert(p, std::string(1 + rng() % 3, ";:,x/1S"[rng() % 7])); break;
                    case 3: x = x.substr(0, p); break;
                }
                if (x.empty()) x = "\n";
            }
            s += x;
        }
        int32_t cnt = 0, skp = 0;
        uint64_t used = 0;
        // exact-size heap copy so that any over-read is caught
        char* buf = (char*)malloc(s.size());
        memcpy(buf, s.data(), s.size());
        int rc = tz_parse_targets(buf, s.size(), n, 4, maxt, amax, st.data(), mv.data(), pol.data(), nm.data(), val.data(), ube.data(), &cnt, &used, &skp);
        free(buf);
        if (rc != 0 || used > s.size() || cnt < 0 || cnt > maxt) { printf("bad rc=%d used=%llu\n", rc, (unsigned long long)used); return 1; }
        parsed += cnt; skipped_total += skp;
        if (cnt > 0) {  // whatever parsed must format again
            std::vector<char> out(cnt * (200 + 32 * amax));
            uint64_t w = 0;
            rc = tz_format_targets(n, cnt, st.data(), mv.data(), pol.data(), nm.data(), amax, val.data(), ube.data(), out.data(), out.size(), &w);
            if (rc != 0) { printf("format rc=%d: %s\n", rc, tz_last_error()); return 1; }
        }
    }
    printf("ok parsed=%ld skipped=%ld\n", parsed, skipped_total);
    return 0;
}
