// test harness of takzero_amd/csrc/tz_fp6.h (tests/test_fp6_host.py).
//   mode "code":  stdin = floats (binary, in units of the block scale), stdout = their E2M3 codes (one byte each)
//   mode "scale": stdin = floats (block maxima),                        stdout = the E8M0 scale bytes
//   mode "pack":  stdin = 32 code bytes per record,                     stdout = the 24-byte operand strings
#include <cstdio>
#include <cstring>

#include "tz_fp6.h"

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "code";
    if (!strcmp(mode, "pack")) {
        uint8_t codes[32];
        while (fread(codes, 1, 32, stdin) == 32) {
            uint32_t out[6];
            tz_pack_fp6x32(codes, out);
            fwrite(out, 4, 6, stdout);
        }
        return 0;
    }
    float f;
    while (fread(&f, 4, 1, stdin) == 1) {
        const unsigned char c = !strcmp(mode, "scale") ? (unsigned char)tz_e2m3_block_scale_byte(f) : tz_f32_to_e2m3(f);
        fwrite(&c, 1, 1, stdout);
    }
    return 0;
}
