// test harness of takzero_amd/csrc/tz_fp8.h: stdin = floats (binary), stdout = their E4M3 codes (tests/test_fp8_host.py)
#include <cstdio>

#include "tz_fp8.h"

int main() {
    float f;
    while (fread(&f, 4, 1, stdin) == 1) {
        const unsigned char c = tz_f32_to_e4m3(f);
        fwrite(&c, 1, 1, stdout);
    }
    return 0;
}
