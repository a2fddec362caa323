// inflate.hpp -- gzip members inflated from memory into one growing buffer, for the .fq.gz inputs the reference reads through needletail / flate2
// (src/seq_parse.rs:356-379, src/kmer_comp.rs:108-128).  zlib's gzread inflates ~0.3 GB/s of output on one core: 1 s for the 300 MB of a 100k-read sample,
// 25 x the step it feeds.  This decoder is written for that one job: the whole compressed file is mapped, the whole output is one buffer (it IS the
// window: matches copy from the output itself), the bit reader holds 64 bits and refills with one unaligned load, literal / length codes resolve through a 12-bit
// table (longer codes through sub-tables), matches are copied eight bytes at a time.  The member's CRC-32 is checked (flate2 does) with carry-less multiplication
// folding (PCLMULQDQ) where the CPU has it, verified against zlib's crc32 on first use.  zlib stays the test oracle (tests/test_io.py: byte-equal output on the
// fixtures, multi-member files, stored / fixed / dynamic blocks) and the fallback: whatever this decoder refuses, the zlib line reader reads -- or words the error.
#pragma once
#include <sys/mman.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace savont {
namespace gz {

typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;

// ---- an anonymous mapping that grows without copying (mremap moves page tables, not pages) and keeps its pages between uses.  No MADV_HUGEPAGE: with
// transparent_hugepage/defrag = madvise the first touch of every 2 MB compacts memory synchronously (a cold 300 MB output: 2.7 s against 1.6 s with 4 KB pages) --------
struct BigBuf {
    u8* p = nullptr; size_t cap = 0;
    BigBuf() = default; BigBuf(const BigBuf&) = delete; BigBuf& operator=(const BigBuf&) = delete;
    ~BigBuf() { if (p) munmap(p, cap); }
    bool reserve(size_t want) {
        if (want <= cap) return true;
        want = (want + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        void* q = p ? mremap(p, cap, want, MREMAP_MAYMOVE) : mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) return false;
        p = (u8*)q; cap = want;
        return true;
    }
};

// ---- CRC-32 (gzip polynomial) -------------------------------------------------------------------------------------------------------------------------
#if defined(__x86_64__)
// folding by carry-less multiplication, four 128-bit lanes per 64 bytes ("Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009;
// the constants are x^(n) mod P for the bit-reflected polynomial 0x1DB710641).  len >= 64 and a multiple of 16.
__attribute__((target("pclmul,sse4.1"))) inline u32 crc32_clmul(u32 crc, const u8* buf, size_t len) {
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4), k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k5k0 = _mm_set_epi64x(0x0000000000, 0x0163cd6124), poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00)); x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20)); x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = k1k2;
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00); x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11); x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = k3k4;                                                      // fold the four lanes into one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {                                             // single lanes
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);                         // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8); x1 = _mm_xor_si128(x1, x2);
    x0 = k5k0;
    x2 = _mm_srli_si128(x1, 4); x1 = _mm_and_si128(x1, x3); x1 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
    x0 = poly;                                                      // Barrett reduction 64 -> 32 bits
    x2 = _mm_and_si128(x1, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x10); x2 = _mm_and_si128(x2, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
    return (u32)_mm_extract_epi32(x1, 1);
}
#endif
// crc32 of buf[0, len) continuing from `crc` (zlib's convention: pass 0 first).  The folding kernel works on the raw register (no pre / post inversion).
inline u32 crc32_fast(u32 crc, const u8* buf, size_t len) {
#if defined(__x86_64__)
    static const int usable = [] {                                  // the CPU has PCLMULQDQ and the kernel agrees with zlib on a test vector, else zlib
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return 0;
        u8 t[256 + 48]; for (size_t i = 0; i < sizeof t; i++) t[i] = (u8)(i * 131 + 7);
        return (~crc32_clmul(~0x12345678u, t, sizeof t)) == (u32)crc32(0x12345678u, t, (uInt)sizeof t) ? 1 : 0;
    }();
    if (usable && len >= 64) {
        const size_t body = len & ~(size_t)15;
        crc = ~crc32_clmul(~crc, buf, body);
        buf += body; len -= body;
    }
#endif
    while (len) { const uInt n = (uInt)std::min<size_t>(len, (size_t)1 << 30); crc = (u32)crc32(crc, buf, n); buf += n; len -= n; }
    return crc;
}

// ---- deflate ------------------------------------------------------------------------------------------------------------------------------------------
// table entry: bits 0-7 = bits of input the symbol takes (for a sub-table entry: primary bits + its own), bits 8-9 = kind, bits 10-15 = extra bits (kind 1)
// or index bits of the sub-table (kind 3), bits 16-31 = literal / base value / sub-table offset.  0 = no such code.
enum : u32 { K_LIT = 0, K_BASE = 1, K_END = 2, K_SUB = 3 };
constexpr int LIT_BITS = 12, DIST_BITS = 8, PRE_BITS = 7;
inline u32 mk(u32 kind, u32 len, u32 extra, u32 value) { return len | (kind << 8) | (extra << 10) | (value << 16); }

struct Tables {
    u32 lit[(1 << LIT_BITS) + 1024];      // sub-tables behind the primary table: at most 2^15 / 2^11 x ... bounded by the Kraft sum: < 2^(15 - 11) x 288 -> 1024 is generous (checked)
    u32 dist[(1 << DIST_BITS) + 512];
    u32 pre[1 << PRE_BITS];
};
static const u16 LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const u8 LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const u16 DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const u8 DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// canonical Huffman code of lens[0, n) into a table of `tb` primary bits (+ sub-tables up to `cap` entries in all); which: 0 literal / length, 1 distance, 2 precode.
// false: over-subscribed code, a code that does not fit, or (literal / length) an incomplete code other than the single-code case RFC 1951 allows for distances.
inline bool build_table(const u8* lens, int n, u32* tab, int tb, size_t cap, int which) {
    int count[16] = {0}; for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int maxlen = 0; for (int l = 1; l <= 15; l++) if (count[l]) maxlen = l;
    for (size_t i = 0; i < ((size_t)1 << tb); i++) tab[i] = 0;
    if (maxlen == 0) return which != 2;                             // no codes: a block without matches (distances) -- a literal / length code without symbol 256 is refused by the caller
    long left = 1; for (int l = 1; l <= 15; l++) { left = left * 2 - count[l]; if (left < 0) return false; }     // over-subscribed
    if (left > 0 && (which == 2 || maxlen != 1)) return false;      // incomplete: only the one-code case is legal (a single distance, RFC 1951 3.2.7), as zlib's inflate_table has it
    u32 next[16]; { u32 c = 0; for (int l = 1; l <= 15; l++) { c = (c + (u32)count[l - 1]) << 1; next[l] = c; } }
    // sub-tables: for every primary prefix, the longest code under it decides its size
    u8 sub_bits[1 << LIT_BITS]; memset(sub_bits, 0, (size_t)1 << tb);
    struct Code { u16 sym; u8 len; u32 rev; };
    std::vector<Code> codes; codes.reserve((size_t)n);
    for (int s = 0; s < n; s++) {
        const int l = lens[s]; if (!l) continue;
        u32 c = next[l]++, r = 0; for (int b = 0; b < l; b++) r |= ((c >> b) & 1u) << (l - 1 - b);
        codes.push_back(Code{(u16)s, (u8)l, r});
        if (l > tb) { const u32 pfx = r & ((1u << tb) - 1); if (l - tb > sub_bits[pfx]) sub_bits[pfx] = (u8)(l - tb); }
    }
    size_t used = (size_t)1 << tb;
    for (u32 pfx = 0; pfx < (1u << tb); pfx++) if (sub_bits[pfx]) {
        const size_t sz = (size_t)1 << sub_bits[pfx];
        if (used + sz > ((size_t)1 << tb) + cap) return false;
        tab[pfx] = mk(K_SUB, 0, sub_bits[pfx], (u32)used);
        for (size_t i = 0; i < sz; i++) tab[used + i] = 0;
        used += sz;
    }
    for (const Code& c : codes) {
        u32 e;
        if (which == 0) e = c.sym < 256 ? mk(K_LIT, c.len, 0, c.sym) : c.sym == 256 ? mk(K_END, c.len, 0, 0) : c.sym <= 285 ? mk(K_BASE, c.len, LEN_EXTRA[c.sym - 257], LEN_BASE[c.sym - 257]) : 0;
        else if (which == 1) e = c.sym < 30 ? mk(K_BASE, c.len, DIST_EXTRA[c.sym], DIST_BASE[c.sym]) : 0;
        else e = mk(K_LIT, c.len, 0, c.sym);
        if (e == 0) continue;                                       // symbols 286 / 287, distances 30 / 31: in the code, never valid in the data (a zero entry is an error when met)
        if (c.len <= tb) { for (u32 i = c.rev; i < (1u << tb); i += 1u << c.len) tab[i] = e; }
        else {
            const u32 pfx = c.rev & ((1u << tb) - 1), sb = sub_bits[pfx], off = tab[pfx] >> 16;
            for (u32 i = c.rev >> tb; i < (1u << sb); i += 1u << (c.len - tb)) tab[off + i] = e;
        }
    }
    if (which == 0) {
        // TWO literals per look-up where both codes fit the primary index: the decoder's pace on literal runs is the chain look-up -> shift -> look-up (one L1 load
        // latency per symbol), and FASTQ is mostly literals -- two-bit-ish codes for the bases, five to seven bits for the quality characters.  An entry whose first code
        // is a literal of L1 bits is joined with the literal the remaining index bits decide completely (its code is no longer than what is left): bit 10 flags the pair,
        // bits 24-31 carry the second byte, bits 0-7 the bits of both.
        static thread_local u32 single[1 << LIT_BITS];
        memcpy(single, tab, sizeof(u32) << tb);
        for (u32 i = 0; i < (1u << tb); i++) {
            const u32 e = single[i];
            if ((e >> 8 & 3) != K_LIT || (e & 0xFF) == 0) continue;
            const u32 l1 = e & 0xFF;
            if (l1 >= (u32)tb) continue;
            const u32 e2 = single[i >> l1];
            if ((e2 >> 8 & 3) != K_LIT || (e2 & 0xFF) == 0 || l1 + (e2 & 0xFF) > (u32)tb) continue;
            tab[i] = (l1 + (e2 & 0xFF)) | (K_LIT << 8) | (1u << 10) | ((e >> 16 & 0xFF) << 16) | ((e2 >> 16 & 0xFF) << 24);
        }
    }
    return true;
}

struct Reader {                                                     // 64-bit bit buffer over [in, end)
    const u8* in; const u8* end; u64 buf = 0; int cnt = 0;
    inline void refill() {
        if (in + 8 <= end) { u64 w; memcpy(&w, in, 8); buf |= w << cnt; const int nb = (63 - cnt) >> 3; in += nb; cnt += nb * 8; }
        else while (cnt <= 56 && in < end) { buf |= (u64)*in++ << cnt; cnt += 8; }
    }
    inline u32 peek(int n) const { return (u32)(buf & (((u64)1 << n) - 1)); }
    inline void drop(int n) { buf >>= n; cnt -= n; }
    inline bool take(int n, u32& v) { if (cnt < n) { refill(); if (cnt < n) return false; } v = peek(n); drop(n); return true; }
    inline void to_byte() { const int r = cnt & 7; drop(r); in -= cnt >> 3; buf = 0; cnt = 0; }   // whole bytes still in the buffer go back to the input
};

// one deflate stream from r into out (grown as needed; `len` = bytes valid so far, matches may reach back to out.p[member_start] -- the first byte of THIS gzip member:
// zlib and flate2 reject a distance into the previous member's output as "invalid distance too far back"); false on corrupt / truncated input
inline bool inflate_stream(Reader& r, BigBuf& out, size_t& len, Tables& T, size_t member_start = 0) {
    static const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (;;) {
        u32 final_, type;
        if (!r.take(1, final_) || !r.take(2, type)) return false;
        if (type == 0) {                                            // stored
            r.to_byte();
            if (r.in + 4 > r.end) return false;
            const u32 n = r.in[0] | (r.in[1] << 8), nn = r.in[2] | (r.in[3] << 8);
            r.in += 4;
            if ((n ^ nn) != 0xFFFF || r.in + n > r.end) return false;
            if (!out.reserve(len + n + 64)) return false;
            memcpy(out.p + len, r.in, n); len += n; r.in += n;
        } else if (type == 1 || type == 2) {
            u8 lens[320];
            int nlit = 288, ndist = 30;
            if (type == 1) {
                for (int i = 0; i < 144; i++) lens[i] = 8; for (int i = 144; i < 256; i++) lens[i] = 9; for (int i = 256; i < 280; i++) lens[i] = 7; for (int i = 280; i < 288; i++) lens[i] = 8;
                for (int i = 0; i < 30; i++) lens[288 + i] = 5;
                ndist = 30;
                u8 dl[32]; for (int i = 0; i < 32; i++) dl[i] = 5;     // the fixed distance code has 32 codes of 5 bits (30 and 31 never occur)
                if (!build_table(lens, 288, T.lit, LIT_BITS, 1024, 0) || !build_table(dl, 32, T.dist, DIST_BITS, 512, 1)) return false;
            } else {
                u32 hlit, hdist, hclen;
                if (!r.take(5, hlit) || !r.take(5, hdist) || !r.take(4, hclen)) return false;
                nlit = (int)hlit + 257; ndist = (int)hdist + 1; const int ncl = (int)hclen + 4;
                if (nlit > 286 || ndist > 30) return false;
                u8 cl[19] = {0};
                for (int i = 0; i < ncl; i++) { u32 v; if (!r.take(3, v)) return false; cl[order[i]] = (u8)v; }
                if (!build_table(cl, 19, T.pre, PRE_BITS, 0, 2)) return false;
                int i = 0;
                while (i < nlit + ndist) {
                    if (r.cnt < 14) r.refill();
                    const u32 e = T.pre[r.peek(PRE_BITS)];
                    if ((e & 0xFF) == 0 || (int)(e & 0xFF) > r.cnt) return false;
                    r.drop((int)(e & 0xFF));
                    const u32 sym = e >> 16;
                    if (sym < 16) lens[i++] = (u8)sym;
                    else {
                        u32 rep, v = 0;
                        if (sym == 16) { if (i == 0 || !r.take(2, rep)) return false; v = lens[i - 1]; rep += 3; }
                        else if (sym == 17) { if (!r.take(3, rep)) return false; rep += 3; }
                        else { if (!r.take(7, rep)) return false; rep += 11; }
                        if (i + (int)rep > nlit + ndist) return false;
                        while (rep--) lens[i++] = (u8)v;
                    }
                }
                if (lens[256] == 0) return false;                   // no end-of-block code
                if (!build_table(lens, nlit, T.lit, LIT_BITS, 1024, 0) || !build_table(lens + nlit, ndist, T.dist, DIST_BITS, 512, 1)) return false;
            }
            // ---- the symbols of the block
            for (;;) {
                if (out.cap < len + ((size_t)1 << 16) && !out.reserve(std::max(len + ((size_t)1 << 16), out.cap + out.cap / 2))) return false;   // room for a run of symbols (258 bytes per match + the 8-byte copy overshoot); grows by halves when the size hint was short
                u8* o = out.p + len; u8* const o_safe = out.p + out.cap - 300;
                bool end_block = false;
                if (r.cnt < 48) r.refill();
                u32 e = T.lit[r.peek(LIT_BITS)];                     // the entry of the NEXT symbol is looked up as soon as its bits are known -- before the bytes of the current one are stored
                while (o < o_safe) {
                    if ((e >> 8 & 3) == K_SUB) e = T.lit[(e >> 16) + ((u32)(r.buf >> LIT_BITS) & ((1u << (e >> 10 & 63)) - 1))];
                    const int nb = (int)(e & 0xFF);
                    if (nb == 0 || nb > r.cnt) return false;
                    r.drop(nb);
                    const u32 kind = e >> 8 & 3;
                    if (kind == K_LIT) {                             // one literal, or two (the second byte of a single is overwritten by what follows)
                        const u32 cur = e;
                        if (r.cnt < 48) r.refill();
                        e = T.lit[r.peek(LIT_BITS)];
                        o[0] = (u8)(cur >> 16); o[1] = (u8)(cur >> 24); o += 1 + (cur >> 10 & 1);
                        continue;
                    }
                    if (kind == K_END) { end_block = true; break; }
                    const int xb = (int)(e >> 10 & 63);
                    if (xb > r.cnt) return false;                   // (only at the end of a truncated input: a refill leaves >= 48 bits, a length + distance pair takes at most 48)
                    u32 mlen = (e >> 16) + r.peek(xb); r.drop(xb);
                    u32 d = T.dist[r.peek(DIST_BITS)];
                    if ((d >> 8 & 3) == K_SUB) d = T.dist[(d >> 16) + ((u32)(r.buf >> DIST_BITS) & ((1u << (d >> 10 & 63)) - 1))];
                    const int db = (int)(d & 0xFF);
                    if (db == 0 || db > r.cnt) return false;
                    r.drop(db);
                    const int dx = (int)(d >> 10 & 63);
                    if (dx > r.cnt) { r.refill(); if (dx > r.cnt) return false; }
                    const size_t dist = (size_t)(d >> 16) + r.peek(dx); r.drop(dx);
                    if (r.cnt < 48) r.refill();
                    e = T.lit[r.peek(LIT_BITS)];
                    if (dist > (size_t)(o - out.p) - member_start) return false;   // before the start of this member's output
                    const u8* s = o - dist;
                    if (dist >= 8) {
                        u8* const stop = o + mlen;
                        do { u64 w; memcpy(&w, s, 8); memcpy(o, &w, 8); s += 8; o += 8; } while (o < stop);
                        o = stop;
                    } else if (dist == 1) { memset(o, *s, mlen); o += mlen; }
                    else { while (mlen--) *o++ = *s++; }
                }
                len = (size_t)(o - out.p);
                if (end_block) break;
            }
        } else return false;
        if (final_) return true;
    }
}

// every gzip member of [src, src + n) into out / len; stops (successfully) at bytes that do not start another member, as zlib's gzread does.
// why: set on failure.  CRC-32 and ISIZE of every member are checked.
inline bool gunzip_all(const u8* src, size_t n, BigBuf& out, size_t& len, std::string& why) {
    static thread_local Tables T;
    const u8* p = src; const u8* const end = src + n;
    len = 0;
    bool first = true;
    while (end - p >= 18 && p[0] == 0x1f && p[1] == 0x8b) {
        if (p[2] != 8 || (p[3] & 0xE0)) { why = "unknown gzip method / flags"; return false; }
        const u8 flg = p[3];
        const u8* q = p + 10;
        if (flg & 4) { if (end - q < 2) { why = "truncated gzip header"; return false; } const size_t xl = q[0] | (q[1] << 8); q += 2; if ((size_t)(end - q) < xl) { why = "truncated gzip header"; return false; } q += xl; }
        if (flg & 8) { while (q < end && *q) q++; if (q >= end) { why = "truncated gzip header"; return false; } q++; }
        if (flg & 16) { while (q < end && *q) q++; if (q >= end) { why = "truncated gzip header"; return false; } q++; }
        if (flg & 2) { if (end - q < 2) { why = "truncated gzip header"; return false; } q += 2; }
        if (first) {                                                // ISIZE of the LAST member: exact for the usual one-member file; a hint otherwise
            const u32 isz = end[-4] | (end[-3] << 8) | (end[-2] << 16) | ((u32)end[-1] << 24);
            out.reserve(std::max<size_t>((size_t)isz, n * 3) + ((size_t)1 << 20));
            first = false;
        }
        Reader r{q, end};
        const size_t start = len;
        if (!inflate_stream(r, out, len, T, start)) { why = "corrupt or truncated deflate stream"; return false; }
        r.to_byte();
        if (end - r.in < 8) { why = "truncated gzip trailer"; return false; }
        const u32 want_crc = r.in[0] | (r.in[1] << 8) | (r.in[2] << 16) | ((u32)r.in[3] << 24), want_len = r.in[4] | (r.in[5] << 8) | (r.in[6] << 16) | ((u32)r.in[7] << 24);
        if ((u32)(len - start) != want_len) { why = "gzip length check failed"; return false; }
        if (crc32_fast(0, out.p + start, len - start) != want_crc) { why = "gzip CRC-32 check failed"; return false; }
        p = r.in + 8;
    }
    if (first) { why = "not a gzip file"; return false; }
    if (end - p >= 2 && end - p < 18 && p[0] == 0x1f && p[1] == 0x8b) { why = "truncated trailing gzip member"; return false; }   // zlib reports a truncated stream here
    return true;
}

}  // namespace gz
}  // namespace savont
