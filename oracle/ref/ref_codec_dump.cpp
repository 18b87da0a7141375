// TEST INFRASTRUCTURE (oracle/): dumps the compiled reference's decode tables.
//
// Links against the reference's own sources where they lie under
// /root/reference/mixed_precs_caching (see oracle/Makefile); nothing is copied.
// Output (little-endian fp32, written to the file named by argv[1]):
//   [0      .. 256)        EVLFU_8BIT::chars_buffer_to_floats of byte b      (evlfu_8.cpp:370-378)
//   [256    .. 256+512)    EVLFU_4BIT::chars_buffer_to_floats of byte b -> 2 (evlfu_4.cpp:319-341)
//   [768    .. 768+65536)  EVLFU_16BIT::chars_buffer_to_floats of ushort v   (evlfu_16.cpp:332-356)
// The 4-bit decoder indexes a 15-entry table with the nibble value; nibble 15
// is out of bounds in the reference so bytes containing a
// 15 nibble are dumped as NaN markers instead of being decoded.
#include "evlfu_4.hpp"
#include "evlfu_8.hpp"
#include "evlfu_16.hpp"
#include <cmath>
#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s out.bin\n", argv[0]); return 2; }
    std::vector<float> out(256 + 512 + 65536);
    EVLFU_8BIT *c8 = new EVLFU_8BIT(16, false, -1, false, "");
    EVLFU_4BIT *c4 = new EVLFU_4BIT(16, false, -1);
    EVLFU_16BIT *c16 = new EVLFU_16BIT(16, false, -1);
    {   // 8-bit: decode 36 bytes at a time
        char row[36]; float f[36];
        for (int base = 0; base < 256; base += 36) {
            for (int i = 0; i < 36; i++) row[i] = (char)((base + i) & 0xff);
            c8->chars_buffer_to_floats(row, f);
            for (int i = 0; i < 36 && base + i < 256; i++) out[base + i] = f[i];
        }
    }
    {   // 4-bit: 18 bytes -> 36 floats
        char row[18]; float f[36];
        for (int base = 0; base < 256; base += 18) {
            bool has15[18];
            for (int i = 0; i < 18; i++) {
                int b = (base + i) & 0xff;
                has15[i] = ((b & 15) == 15) || ((b >> 4) == 15);
                row[i] = has15[i] ? 0x77 : (char)b;  // 0x77 = (7,7) = safe placeholder
            }
            c4->chars_buffer_to_floats(row, f);
            for (int i = 0; i < 18 && base + i < 256; i++) {
                out[256 + 2 * (base + i) + 0] = has15[i] ? NAN : f[2 * i + 0];
                out[256 + 2 * (base + i) + 1] = has15[i] ? NAN : f[2 * i + 1];
            }
        }
    }
    {   // 16-bit: 36 ushorts at a time
        unsigned short row[36]; float f[36];
        for (int base = 0; base < 65536; base += 36) {
            for (int i = 0; i < 36; i++) row[i] = (unsigned short)((base + i) & 0xffff);
            c16->chars_buffer_to_floats((char *)row, f);
            for (int i = 0; i < 36 && base + i < 65536; i++) out[768 + base + i] = f[i];
        }
    }
    FILE *fp = fopen(argv[1], "wb");
    if (!fp) { perror("fopen"); return 1; }
    fwrite(out.data(), sizeof(float), out.size(), fp);
    fclose(fp);
    fflush(stdout);
    _exit(0);  // reader threads of the reference objects never join
}
