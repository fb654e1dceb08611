// ls_sim.cpp — development tool: statistics for the lane-serial DEFLATE step (exg_inflate_ls.hpp) on real streams.
// Reads raw DEFLATE members ([u32 n] then per member [u32 clen][u32 ulen][bytes]; tools/ls_sim.py writes them), decodes every
// Huffman block serially (the token starts = the truth) and then plays the step on it: 64 lanes, lane l owns the R bits
// from bitpos + R l, pass 1 decodes from G bits in front of the region (a guessed start), pass 2 from the exit of the lane
// in front.  Prints how many lanes a step keeps, how many loop iterations its passes take (the slowest lane's), and how many
// tokens have codes longer than the primary tables.
//   g++ -O2 -o /tmp/ls_sim tools/ls_sim.cpp && /tmp/ls_sim /tmp/ls_members.bin 256 128 9 9
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Huff {
    uint16_t count[16], symbol[288];
    bool build(const uint8_t *len, int n) {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
        for (int i = 0; i < n; i++)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        return true;
    }
};
struct Bits {
    const uint8_t *p;
    size_t n;
    uint32_t bit(size_t pos) const { return pos / 8 < n ? (p[pos / 8] >> (pos & 7)) & 1u : 0u; }
    uint32_t bits(size_t pos, int k) const {
        uint32_t v = 0;
        for (int i = 0; i < k; i++) v |= bit(pos + i) << i;
        return v;
    }
};
// returns symbol, sets len; -1 = invalid
static int decode(const Bits &b, size_t pos, const Huff &h, int *len) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; l++) {
        code |= (int)b.bit(pos + l - 1);
        int c = h.count[l];
        if (code - c < first) {
            *len = l;
            return h.symbol[index + (code - first)];
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    *len = 15;
    return -1;
}
static const uint16_t LB[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LE[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint8_t DE[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct Tok {
    int bits;   // total bits of the token; 0 = invalid, -1 = end of block
    int out;    // bytes it produces
    int longc;  // a code longer than the primary tables
    int dist;   // a match's distance
};
static int g_lit_bits = 9, g_dist_bits = 9, g_repair = 0;
static unsigned long long g_rounds = 0, it3 = 0, g_round_hist[16], g_mlen[5], g_mdist[5], g_short_near = 0;
static Tok token_at(const Bits &b, size_t pos, const Huff &hl, const Huff &hd) {
    Tok t{0, 0, 0, 0};
    int l;
    int s = decode(b, pos, hl, &l);
    if (s < 0) return t;
    if (l > g_lit_bits) t.longc = 1;
    if (s < 256) {
        t.bits = l;
        t.out = 1;
        return t;
    }
    if (s == 256) {
        t.bits = -1;
        return t;
    }
    if (s > 285) return t;
    int x = LE[s - 257];
    int len = LB[s - 257] + (int)b.bits(pos + l, x);
    int l2;
    int d = decode(b, pos + l + x, hd, &l2);
    if (d < 0 || d > 29) return t;
    if (l2 > g_dist_bits) t.longc = 1;
    static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    t.bits = l + x + l2 + DE[d];
    t.out = len;
    t.dist = DBASE[d] + (int)b.bits(pos + l + x + l2, DE[d]);
    return t;
}

int main(int argc, char **argv) {
    if (argc < 4) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    const int R = atoi(argv[2]), G = atoi(argv[3]);
    if (argc > 4) g_lit_bits = atoi(argv[4]);
    if (argc > 5) g_dist_bits = atoi(argv[5]);
    const int kLaneCap = argc > 6 ? atoi(argv[6]) : 1 << 30, kStepCap = argc > 7 ? atoi(argv[7]) : 1 << 30;
    g_repair = argc > 8 ? atoi(argv[8]) : 0;
    uint32_t nm;
    if (fread(&nm, 4, 1, f) != 1) return 1;
    unsigned long long steps = 0, lanes_kept = 0, it1 = 0, it2 = 0, out_total = 0, tok_total = 0, long_total = 0, blocks = 0, bits_total = 0;
    unsigned long long match_total = 0, hist_kept[65] = {0}, steps_full = 0, lane_bad = 0, lane_tried = 0, capped = 0;
    for (uint32_t m = 0; m < nm; m++) {
        uint32_t cl, ul;
        if (fread(&cl, 4, 1, f) != 1 || fread(&ul, 4, 1, f) != 1) break;
        std::vector<uint8_t> c(cl + 16);
        if (fread(c.data(), 1, cl, f) != cl) break;
        Bits b{c.data(), cl};
        size_t pos = 0;
        bool last = false;
        while (!last) {
            last = b.bit(pos);
            int type = (int)b.bits(pos + 1, 2);
            pos += 3;
            if (type == 0) {
                pos = (pos + 7) & ~(size_t)7;
                uint32_t len = b.bits(pos, 16);
                pos += 32 + 8 * (size_t)len;
                continue;
            }
            uint8_t lens[320];
            int nlit = 288, ndist = 30;
            if (type == 1) {
                for (int i = 0; i < 288; i++) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                for (int i = 0; i < 30; i++) lens[288 + i] = 5;
            } else {
                nlit = (int)b.bits(pos, 5) + 257;
                ndist = (int)b.bits(pos + 5, 5) + 1;
                int ncode = (int)b.bits(pos + 10, 4) + 4;
                pos += 14;
                static const uint8_t ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cll[19] = {0};
                for (int i = 0; i < ncode; i++) cll[ord[i]] = (uint8_t)b.bits(pos + 3 * i, 3);
                pos += 3 * ncode;
                Huff hc;
                hc.build(cll, 19);
                uint8_t all[320];
                int idx = 0;
                while (idx < nlit + ndist) {
                    int l;
                    int s = decode(b, pos, hc, &l);
                    pos += l;
                    if (s < 16) all[idx++] = (uint8_t)s;
                    else {
                        int rep, val = 0;
                        if (s == 16) { val = all[idx - 1]; rep = 3 + (int)b.bits(pos, 2); pos += 2; }
                        else if (s == 17) { rep = 3 + (int)b.bits(pos, 3); pos += 3; }
                        else { rep = 11 + (int)b.bits(pos, 7); pos += 7; }
                        while (rep--) all[idx++] = (uint8_t)val;
                    }
                }
                memcpy(lens, all, nlit);
                memset(lens + nlit, 0, 288 - nlit);
                memcpy(lens + 288, all + nlit, ndist);
            }
            Huff hl, hd;
            hl.build(lens, 288);
            hd.build(lens + 288, ndist);
            blocks++;
            // the step, repeated until the end-of-block token
            bool eob = false;
            while (!eob) {
                steps++;
                // pass 1: guessed exits
                long long exit1[64];
                int iters1 = 0;
                for (int l = 0; l < 64; l++) {
                    size_t q = l ? pos + (size_t)R * l - G : pos, end = pos + (size_t)R * (l + 1);
                    int n = 0;
                    long long ex = -1;
                    while (true) {
                        if (q >= end) { ex = (long long)q; break; }
                        Tok t = token_at(b, q, hl, hd);
                        n++;
                        if (t.bits <= 0) break;
                        q += t.bits;
                    }
                    exit1[l] = ex;
                    iters1 = std::max(iters1, n);
                }
                // rounds: every lane whose entry changed decodes its region again from the exit of the lane in front, until nothing
                // changes (lane 0's entry is the truth, so the truth reaches at least one more lane per round)
                int iters2 = 0, kept = 0;
                size_t entry = pos;
                if (g_repair) {
                    long long ent[64], ex[64];
                    bool dirty[64];
                    ent[0] = (long long)pos;
                    ex[0] = exit1[0];
                    dirty[0] = false;
                    for (int l = 1; l < 64; l++) { ent[l] = -2; ex[l] = exit1[l]; dirty[l] = false; }
                    int rounds = 0;
                    for (;;) {
                        bool any = false;
                        int S = 64;  // the first lane whose decode stopped (end of block, invalid code): the lanes behind it wait
                        for (int l = 63; l >= 0; l--)
                            if (ex[l] < 0) S = l;
                        for (int l = 1; l < 64; l++) {
                            dirty[l] = l <= S && ex[l - 1] != ent[l];
                            any |= dirty[l];
                        }
                        if (!any) break;
                        rounds++;
                        long long nex[64];
                        int itr = 0;
                        for (int l = 1; l < 64; l++) {
                            nex[l] = ex[l];
                            if (!dirty[l]) continue;
                            ent[l] = ex[l - 1];
                            if (ent[l] < 0) { nex[l] = -1; continue; }
                            size_t q = (size_t)ent[l], end = pos + (size_t)R * (l + 1);
                            int n = 0;
                            long long e2 = -1;
                            while (true) {
                                if (q >= end) { e2 = (long long)q; break; }
                                Tok t = token_at(b, q, hl, hd);
                                n++;
                                if (t.bits <= 0) break;
                                q += t.bits;
                            }
                            nex[l] = e2;
                            itr = std::max(itr, n);
                        }
                        for (int l = 1; l < 64; l++) ex[l] = nex[l];
                        iters2 += itr;
                        if (rounds > 70) { fprintf(stderr, "no convergence\n"); return 3; }
                    }
                    g_rounds += rounds;
                    g_round_hist[rounds > 15 ? 15 : rounds]++;
                    // the output pass over the lanes of the true chain (up to the end of the block)
                    int itr3 = 0;
                    for (int l = 0; l < 64; l++) {
                        size_t q = (size_t)ent[l] , end = pos + (size_t)R * (l + 1);
                        if (l == 0) q = pos;
                        int n = 0;
                        unsigned long long o = 0, toks = 0, longs = 0, matches = 0;
                        while (q < end) {
                            Tok t = token_at(b, q, hl, hd);
                            n++;
                            if (t.bits == -1) { eob = true; break; }
                            if (t.bits == 0) { fprintf(stderr, "bad token on the true chain\n"); return 2; }
                            q += t.bits; o += t.out; toks++; longs += t.longc; matches += t.out > 1;
                            if (t.out > 1) {
                                g_mlen[t.out <= 4 ? 0 : t.out <= 8 ? 1 : t.out <= 16 ? 2 : t.out <= 64 ? 3 : 4]++;
                                g_mdist[t.dist < 64 ? 0 : t.dist < 900 ? 1 : t.dist < 2048 ? 2 : t.dist < 8192 ? 3 : 4]++;
                                g_short_near += t.out <= 8 && t.dist >= 64 && t.dist < 900;
                            }
                        }
                        itr3 = std::max(itr3, n);
                        kept++;
                        out_total += o; tok_total += toks; long_total += longs; match_total += matches;
                        entry = q;
                        if (eob) { int l3; decode(b, q, hl, &l3); entry = q + l3; break; }
                    }
                    iters2 += itr3;  // (counted once: the x2 in the print is undone below)
                    it3 += itr3;
                } else {
                // pass 2: lane l from the exit of lane l - 1
                int iters2 = 0, kept = 0;
                size_t entry = pos;
                unsigned long long step_out = 0;
                for (int l = 0; l < 64; l++) {
                    size_t q = entry, end = pos + (size_t)R * (l + 1);
                    int n = 0, lane_out = 0;
                    bool stop = false;
                    unsigned long long o = 0, toks = 0, longs = 0, matches = 0;
                    while (q < end) {
                        Tok t = token_at(b, q, hl, hd);
                        n++;
                        if (t.bits == -1) { eob = true; q += 0; break; }
                        if (t.bits == 0) { fprintf(stderr, "bad token on the true chain\n"); return 2; }
                        if (lane_out + t.out > kLaneCap && lane_out) { stop = true; break; }
                        q += t.bits;
                        lane_out += t.out;
                        o += t.out;
                        toks++;
                        longs += t.longc;
                        matches += t.out > 1;
                    }
                    if (step_out + o > (unsigned long long)kStepCap && l > 0) { capped++; eob = false; break; }
                    iters2 = std::max(iters2, n);
                    kept++;
                    step_out += o;
                    out_total += o;
                    tok_total += toks;
                    long_total += longs;
                    match_total += matches;
                    if (eob) {
                        int l3;
                        decode(b, q, hl, &l3);
                        entry = q + l3;
                        break;
                    }
                    entry = q;
                    if (stop) break;
                    if (l < 63) {
                        lane_tried++;
                        if (exit1[l] != (long long)q) {  // (lane l's guess did not merge: the lanes behind start from a false entry)
                            lane_bad++;
                            break;
                        }
                    }
                }
                }
                bits_total += entry - pos;
                pos = entry;
                lanes_kept += kept;
                hist_kept[kept]++;
                steps_full += kept == 64;
                it1 += iters1;
                it2 += iters2;
            }
        }
    }
    printf("R %d G %d lit %d dist %d: members %u blocks %llu steps %llu, lanes kept %.1f (full %.0f %%), lane bad %.2f %%, capped %llu\n", R, G, g_lit_bits,
           g_dist_bits, nm, blocks, steps, (double)lanes_kept / steps, 100.0 * steps_full / steps, 100.0 * lane_bad / (lane_tried + 1), capped);
    printf("  iterations per step: pass1 %.1f pass2 %.1f (x2 with the output pass) = %.1f; tokens/lane-region %.1f; out bytes/step %.0f; in bits/step %.0f\n",
           (double)it1 / steps, (double)it2 / steps, (double)(it1 + 2 * it2) / steps, (double)tok_total / lanes_kept, (double)out_total / steps,
           (double)bits_total / steps);
    printf("  out %llu B, tokens %llu (%.3f per byte), matches %llu (%.1f %% of tokens, %.1f B each), long codes %.2f %% of tokens\n", out_total, tok_total,
           (double)tok_total / out_total, match_total, 100.0 * match_total / tok_total,
           match_total ? (double)(out_total - (tok_total - match_total)) / match_total : 0.0, 100.0 * long_total / tok_total);
    if (g_repair) {
        printf("  match lengths <=4 / <=8 / <=16 / <=64 / more: %llu %llu %llu %llu %llu; distances <64 / <900 / <2048 / <8192 / more: %llu %llu %llu %llu %llu; len <= 8 and 64 <= dist < 900: %llu\n",
               g_mlen[0], g_mlen[1], g_mlen[2], g_mlen[3], g_mlen[4], g_mdist[0], g_mdist[1], g_mdist[2], g_mdist[3], g_mdist[4], g_short_near);
        printf("  rounds histogram:");
        for (int i = 0; i < 16; i++) printf(" %llu", g_round_hist[i]);
        printf("\n");
    }
    if (g_repair)
        printf("  REPAIR: rounds per step %.2f, iterations per step: pass1 %.1f + rounds + output %.1f = %.1f -> per output byte %.4f\n", (double)g_rounds / steps,
               (double)it1 / steps, (double)it2 / steps, (double)(it1 + it2) / steps, (double)(it1 + it2) / out_total);
    else
        printf("  loop iterations per output byte: %.4f\n", (double)(it1 + 2 * it2) / out_total);
    return 0;
}
