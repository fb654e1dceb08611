// exg_xxh64.hpp — XXH64 (seed 0) on the host, streaming: the Content_Checksum of a zstd frame too large to be hashed by one
// wavefront in reasonable time (the hash is a serial recurrence: ~0.55 GB/s per frame on the device, ~10 GB/s on a host
// core).  The device decodes, the bytes come back over PCIe in pieces and a host thread hashes them while the scan runs
// (exg_zstd.hip: host_verify).  Algorithm: the xxHash specification, XXH64.
#pragma once
#include <stdint.h>
#include <string.h>

namespace exg {

struct Xxh64 {
    static constexpr uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull,
                              P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
    uint64_t v[4] = {P1 + P2, P2, 0, 0 - P1};
    uint64_t total = 0;
    uint8_t tail[32];
    uint32_t n_tail = 0;

    static uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
    static uint64_t round(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
    static uint64_t merge(uint64_t h, uint64_t val) { return (h ^ round(0, val)) * P1 + P4; }
    static uint64_t rd64(const uint8_t *p) {
        uint64_t x;
        memcpy(&x, p, 8);
        return x;
    }
    void stripes(const uint8_t *p, size_t n_stripes) {
        uint64_t a = v[0], b = v[1], c = v[2], d = v[3];
        size_t i = 0;
        // Two stripes = one cache line per turn, the line 1 KiB ahead asked for: the bytes come from pinned memory a DMA
        // engine just wrote, never from a cache, and the four multiply chains leave the core's own prefetcher too little
        // to go on (tools/xxh_bench.cpp: 3.9 -> 9.6 GB/s on the build container's core; the distance was swept 256 .. 4096)
        for (; i + 2 <= n_stripes; i += 2, p += 64) {
            __builtin_prefetch(p + 1024, 0, 3);
            a = round(a, rd64(p));
            b = round(b, rd64(p + 8));
            c = round(c, rd64(p + 16));
            d = round(d, rd64(p + 24));
            a = round(a, rd64(p + 32));
            b = round(b, rd64(p + 40));
            c = round(c, rd64(p + 48));
            d = round(d, rd64(p + 56));
        }
        for (; i < n_stripes; i++, p += 32) {
            a = round(a, rd64(p));
            b = round(b, rd64(p + 8));
            c = round(c, rd64(p + 16));
            d = round(d, rd64(p + 24));
        }
        v[0] = a, v[1] = b, v[2] = c, v[3] = d;
    }
    void update(const uint8_t *p, size_t n) {
        total += n;
        if (n_tail) {
            const size_t take = n < 32 - n_tail ? n : 32 - n_tail;
            memcpy(tail + n_tail, p, take);
            n_tail += (uint32_t)take, p += take, n -= take;
            if (n_tail < 32) return;
            stripes(tail, 1);
            n_tail = 0;
        }
        stripes(p, n / 32);
        p += n / 32 * 32;
        n_tail = (uint32_t)(n % 32);
        memcpy(tail, p, n_tail);
    }
    uint64_t digest() const {
        uint64_t h;
        if (total >= 32) {
            h = rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18);
            h = merge(h, v[0]), h = merge(h, v[1]), h = merge(h, v[2]), h = merge(h, v[3]);
        } else {
            h = P5;
        }
        h += total;
        const uint8_t *t = tail, *end = tail + n_tail;
        while (t + 8 <= end) {
            h ^= round(0, rd64(t));
            h = rotl(h, 27) * P1 + P4;
            t += 8;
        }
        if (t + 4 <= end) {
            uint32_t w;
            memcpy(&w, t, 4);
            h ^= (uint64_t)w * P1;
            h = rotl(h, 23) * P2 + P3;
            t += 4;
        }
        while (t < end) {
            h ^= (*t++) * P5;
            h = rotl(h, 11) * P1;
        }
        h ^= h >> 33, h *= P2, h ^= h >> 29, h *= P3, h ^= h >> 32;
        return h;
    }
};

}  // namespace exg
