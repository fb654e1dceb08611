// xxh_bench.cpp — development tool: the host's XXH64 (exg_xxh64.hpp: the checksum of zstd frames too big for one wavefront)
// over 1 GiB that is in no cache.  g++ -O3 -I exon_duckdb_amd/csrc -o /tmp/xxh_bench tools/xxh_bench.cpp && /tmp/xxh_bench
#include <chrono>
#include <cstdio>
#include <vector>

#include "exg_xxh64.hpp"

int main() {
    const size_t n = 1ull << 30;
    std::vector<uint8_t> buf(n);
    for (size_t i = 0; i < n; i += 64) buf[i] = (uint8_t)(i >> 6);
    for (int r = 0; r < 3; r++) {
        exg::Xxh64 h;
        const auto t0 = std::chrono::steady_clock::now();
        h.update(buf.data(), n);
        const auto t1 = std::chrono::steady_clock::now();
        printf("XXH64 of 1 GiB: %.2f GB/s (%016llx)\n", n / std::chrono::duration<double>(t1 - t0).count() / 1e9, (unsigned long long)h.digest());
    }
    // the specification's known answer for the empty input, and a short one
    exg::Xxh64 e;
    printf("empty: %016llx (ef46db3751d8e999)\n", (unsigned long long)e.digest());
    return 0;
}
