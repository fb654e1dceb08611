// Host check of exon_duckdb_amd/csrc/exg_float_el.hpp (the Eisel-Lemire decimal -> float32 the device's f32::from_str is
// built on) against glibc's correctly rounded strtof: random significands of every width at every decimal exponent the
// format reaches, small integers, exact halfway points (integers and negative powers of ten), and the `tie` flag the
// > 19-digit literals rely on.  Built and run by tests/test_float_el.py.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <random>
#include "exg_float_slow.hpp"
// literals from a file (one per line): the exact parser against strtof
static long check_file(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    static char line[8192];
    long n = 0, bad = 0;
    while (fgets(line, sizeof line, f)) {
        size_t len = strlen(line);
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        if (!len) continue;
        float want = strtof(line, nullptr);
        uint32_t wb, gb = 0;
        memcpy(&wb, &want, 4);
        const int rc = exg::f32_parse_exact((const uint8_t *)line, (int)len, &gb);
        n++;
        if (rc != 0 || (gb != wb && !(want != want && (gb & 0x7FFFFFFFu) > 0x7F800000u))) {
            if (bad < 20) printf("EXACT MISMATCH %.60s... rc %d got %08x want %08x\n", line, rc, gb, wb);
            bad++;
        }
    }
    fclose(f);
    printf("%ld literals from %s, %ld bad\n", n, path, bad);
    return bad;
}

int main(int argc, char **argv) {
    if (argc > 1) return check_file(argv[1]) != 0;
    std::mt19937_64 rng(12345);
    long bad = 0, n = 0;
    auto check = [&](uint64_t w, int q) {
        char buf[64]; snprintf(buf, sizeof buf, "%llue%d", (unsigned long long)w, q);
        float f = strtof(buf, nullptr); uint32_t want; memcpy(&want, &f, 4);
        uint32_t got = exg::el_f32_bits(w, q);
        n++;
        if (got != want) { if (bad < 20) printf("MISMATCH %s got %08x want %08x\n", buf, got, want); bad++; }
    };
    for (int q = -70; q <= 45; q++) {
        for (int k = 0; k < 12000; k++) {
            uint64_t w = rng();
            int bits = 1 + (int)(rng() % 64);
            if (bits < 64) w &= (1ull << bits) - 1;
            check(w, q);
        }
        for (uint64_t w = 0; w < 300; w++) check(w, q);
    }
    // halfway cases: (2m+1) * 2^k for small exponents expressed as decimal integers
    for (int k = 0; k < 40; k++) for (uint64_t m = (1u << 23); m < (1u << 23) + 2000; m++) { unsigned __int128 v = ((unsigned __int128)(2 * m + 1)) << k; if (v >> 64) continue; check((uint64_t)v, 0); check((uint64_t)v, -3); }
    // decimal halfway points with negative exponents: (2m+1) * 5^j * 10^-j = (2m+1) / 2^j
    for (int j = 1; j <= 17; j++) { uint64_t p5 = 1; for (int i = 0; i < j; i++) p5 *= 5; for (uint64_t m = (1u << 23); m < (1u << 23) + 3000; m++) { unsigned __int128 v = (unsigned __int128)(2 * m + 1) * p5; if (v >> 64) continue; check((uint64_t)v, -j); } }
    // the tie flag: (2m + 1) 2^k is exactly halfway; its neighbours are not
    long ties = 0;
    for (int k = 0; k < 38; k++) for (uint64_t m = (1u << 23); m < (1u << 23) + 1500; m++) {
        unsigned __int128 v = ((unsigned __int128)(2 * m + 1)) << k; if (v >> 63) continue;
        bool t = false; exg::el_f32_bits((uint64_t)v, 0, &t); ties++; if (!t) bad++;
        exg::el_f32_bits((uint64_t)v + 1, 0, &t); if (t) bad++;
        exg::el_f32_bits((uint64_t)v - 1, 0, &t); if (t) bad++;
    }
    printf("%ld checked, %ld ties, %ld bad\n", n, ties, bad);
    return bad != 0;
}
