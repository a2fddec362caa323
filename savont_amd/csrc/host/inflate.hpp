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
#include <memory>
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
        madvise(q, want, MADV_HUGEPAGE);                                 // hundreds of MB that are written once from end to end: 2 MB pages where the system hands them out on request (a hint; ignored elsewhere)
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
// Stops: `stops` (ascending bit positions relative to `base`, may be null) are block starts other threads decode from (gunzip_parallel); the stream ends early -- *stopped_at =
// that position -- when a block ends exactly on one of them.
inline bool inflate_stream(Reader& r, BigBuf& out, size_t& len, Tables& T, size_t member_start = 0, const u8* base = nullptr, const std::vector<size_t>* stops = nullptr, size_t* stopped_at = nullptr) {
    static const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    if (stopped_at) *stopped_at = 0;
    for (;;) {
        if (stops) {                                                 // between two blocks: is this where another thread took over?
            const size_t pos = (size_t)(r.in - base) * 8 - (size_t)r.cnt;
            if (std::binary_search(stops->begin(), stops->end(), pos)) { *stopped_at = pos; return true; }
        }
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

// ---- one member on several threads (round 6) -----------------------------------------------------------------------------------------------------------------
// A .fq.gz sample is ONE deflate stream (the basecallers write plain gzip), and a lone `savont asv` run waited 0.6 s of one core for it: six steps' worth.  Deflate has no
// index, but a decoder can start at any BLOCK boundary if it keeps the bytes it cannot know yet symbolic (Kerbiriou & Chikhi, "Parallel decompression of gzip-compressed
// files and random access to DNA sequences", 2019):
//   * the compressed member is cut into T pieces; piece k > 0 starts at the first position behind its cut that parses as a dynamic-Huffman block header with complete codes
//     and decodes a few thousand symbols of text without an error (find_block_start);
//   * piece k decodes into 16-bit symbols behind a window of 32768 MARKERS: a match that reaches back before the piece copies markers instead of bytes, and markers are copied
//     like any other symbol from then on;
//   * a piece stops when a block ends exactly on the start of a later piece.  A guessed start no piece arrives at was not a boundary of THIS stream: its piece is dropped and
//     the piece before simply went on (so a wrong guess costs time, never bytes);
//   * in stream order, the markers of a piece are replaced by the 32768 bytes in front of it, which are final by then (the replacement itself runs on the pool); the member's
//     CRC-32 and length are checked over the whole output as before.
// zlib stays the oracle: byte-equal output (tests/test_io.py), and whatever this path refuses goes the sequential way.
struct U16Buf {
    u16* p = nullptr; size_t cap = 0;                               // in elements
    U16Buf() = default; U16Buf(const U16Buf&) = delete; U16Buf& operator=(const U16Buf&) = delete;
    ~U16Buf() { if (p) munmap(p, cap * 2); }
    bool reserve(size_t want) {
        if (want <= cap) return true;
        size_t bytes = (want * 2 + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        void* q = p ? mremap(p, cap * 2, bytes, MREMAP_MAYMOVE) : mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) return false;
        madvise(q, bytes, MADV_HUGEPAGE);
        p = (u16*)q; cap = bytes / 2;
        return true;
    }
};
constexpr size_t MARK_WIN = 32768;
inline Reader reader_at_bit(const u8* base, const u8* end, size_t bit) { Reader r{base + (bit >> 3), end}; r.refill(); r.drop((int)(bit & 7)); return r; }
inline size_t reader_bit(const Reader& r, const u8* base) { return (size_t)(r.in - base) * 8 - (size_t)r.cnt; }

// the header of a dynamic block at r -> T.lit / T.dist; false on anything a deflate encoder cannot have written
inline bool read_dynamic_header(Reader& r, Tables& T) {
    static const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    u8 lens[320];
    u32 hlit, hdist, hclen;
    if (!r.take(5, hlit) || !r.take(5, hdist) || !r.take(4, hclen)) return false;
    const int nlit = (int)hlit + 257, ndist = (int)hdist + 1, ncl = (int)hclen + 4;
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
    if (lens[256] == 0) return false;
    return build_table(lens, nlit, T.lit, LIT_BITS, 1024, 0) && build_table(lens + nlit, ndist, T.dist, DIST_BITS, 512, 1);
}
inline bool fixed_tables(Tables& T) {
    u8 lens[288];
    for (int i = 0; i < 144; i++) lens[i] = 8; for (int i = 144; i < 256; i++) lens[i] = 9; for (int i = 256; i < 280; i++) lens[i] = 7; for (int i = 280; i < 288; i++) lens[i] = 8;
    u8 dl[32]; for (int i = 0; i < 32; i++) dl[i] = 5;
    return build_table(lens, 288, T.lit, LIT_BITS, 1024, 0) && build_table(dl, 32, T.dist, DIST_BITS, 512, 1);
}
// the symbols of ONE block (tables in T) into 16-bit symbols at out.p[len..]; matches copy symbols, markers included.  max_syms: stop after that many (the block-start
// test); text_only: every literal must be a byte a FASTA / FASTQ file holds.  -> 1 end of block, 2 max_syms reached, 0 error
inline int block_symbols_u16(Reader& r, U16Buf& out, size_t& len, const Tables& T, size_t max_syms, bool text_only) {
    size_t n_syms = 0;
    for (;;) {
        if (out.cap < len + 600 && !out.reserve(std::max(len + ((size_t)1 << 20), out.cap + out.cap / 2))) return 0;
        u16* o = out.p + len; u16* const o_safe = out.p + out.cap - 300;
        if (r.cnt < 48) r.refill();
        u32 e = T.lit[r.peek(LIT_BITS)];                             // as in the byte decoder: the entry of the NEXT symbol is looked up as soon as its bits are known
        while (o < o_safe) {
            if ((e >> 8 & 3) == K_SUB) e = T.lit[(e >> 16) + ((u32)(r.buf >> LIT_BITS) & ((1u << (e >> 10 & 63)) - 1))];
            const int nb = (int)(e & 0xFF);
            if (nb == 0 || nb > r.cnt) return 0;
            r.drop(nb);
            const u32 kind = e >> 8 & 3;
            if (kind == K_LIT) {
                const u32 cur = e;
                if (r.cnt < 48) r.refill();
                e = T.lit[r.peek(LIT_BITS)];
                const u32 b0 = cur >> 16 & 0xFF, two = cur >> 10 & 1, b1 = cur >> 24;
                if (text_only) {
                    if (!((b0 >= 32 && b0 < 127) || b0 == 10 || b0 == 13 || b0 == 9)) return 0;
                    if (two && !((b1 >= 32 && b1 < 127) || b1 == 10 || b1 == 13 || b1 == 9)) return 0;
                }
                o[0] = (u16)b0; o[1] = (u16)b1; o += 1 + two;
            } else if (kind == K_END) { len = (size_t)(o - out.p); return 1; }
            else {
                const int xb = (int)(e >> 10 & 63);
                if (xb > r.cnt) return 0;
                u32 mlen = (e >> 16) + r.peek(xb); r.drop(xb);
                u32 d = T.dist[r.peek(DIST_BITS)];
                if ((d >> 8 & 3) == K_SUB) d = T.dist[(d >> 16) + ((u32)(r.buf >> DIST_BITS) & ((1u << (d >> 10 & 63)) - 1))];
                const int db = (int)(d & 0xFF);
                if (db == 0 || db > r.cnt) return 0;
                r.drop(db);
                const int dx = (int)(d >> 10 & 63);
                if (dx > r.cnt) { r.refill(); if (dx > r.cnt) return 0; }
                const size_t dist = (size_t)(d >> 16) + r.peek(dx); r.drop(dx);
                if (r.cnt < 48) r.refill();
                e = T.lit[r.peek(LIT_BITS)];
                if (dist > (size_t)(o - out.p)) return 0;            // before the marker window: more than 32768 back
                const u16* s2 = o - dist;
                if (dist >= 4) { u16* const stop = o + mlen; do { u64 w; memcpy(&w, s2, 8); memcpy(o, &w, 8); s2 += 4; o += 4; } while (o < stop); o = stop; }
                else while (mlen--) *o++ = *s2++;
            }
            if (max_syms && ++n_syms >= max_syms) { len = (size_t)(o - out.p); return 2; }
        }
        len = (size_t)(o - out.p);
    }
}
// blocks from bit `start` (a block boundary) into out (whose first MARK_WIN symbols are the markers) until a block ends on one of `stops`, or the stream's final block ends
inline bool inflate_blocks_u16(const u8* base, const u8* end, size_t start, U16Buf& out, size_t& len, Tables& T, const std::vector<size_t>& stops, size_t& end_bit, bool& final_seen) {
    Reader r = reader_at_bit(base, end, start);
    final_seen = false;
    for (bool first = true;; first = false) {
        if (!first) { const size_t pos = reader_bit(r, base); if (std::binary_search(stops.begin(), stops.end(), pos)) { end_bit = pos; return true; } }
        u32 fin, type;
        if (!r.take(1, fin) || !r.take(2, type)) return false;
        if (type == 0) {
            r.to_byte();
            if (r.in + 4 > r.end) return false;
            const u32 n = r.in[0] | (r.in[1] << 8), nn = r.in[2] | (r.in[3] << 8);
            r.in += 4;
            if ((n ^ nn) != 0xFFFF || r.in + n > r.end) return false;
            if (!out.reserve(len + n + 600)) return false;
            for (u32 i = 0; i < n; i++) out.p[len + i] = r.in[i];
            len += n; r.in += n;
        } else if (type == 1 || type == 2) {
            if (type == 1 ? !fixed_tables(T) : !read_dynamic_header(r, T)) return false;
            if (block_symbols_u16(r, out, len, T, 0, false) != 1) return false;
        } else return false;
        if (fin) { final_seen = true; end_bit = reader_bit(r, base); return true; }
    }
}
// the first bit position in [from, from + span) that looks like the start of a dynamic block of a TEXT stream: header with complete codes, 8192 symbols (or the whole
// block) decoded without an error, every literal a text byte.  ~0: none
inline size_t find_block_start(const u8* base, const u8* end, size_t from, size_t span, Tables& T, U16Buf& scratch) {
    const size_t total_bits = (size_t)(end - base) * 8;
    for (size_t b = from; b < from + span && b + 64 < total_bits; b++) {
        const size_t byte = b >> 3; const int sh = (int)(b & 7);
        u32 w; memcpy(&w, base + byte, 4); w >>= sh;
        if ((w & 7) != 4) continue;                                  // BFINAL 0, BTYPE 2 (bits: final, then type low bit first)
        if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;  // HLIT, HDIST
        Reader r = reader_at_bit(base, end, b + 3);
        if (!read_dynamic_header(r, T)) continue;
        if (!scratch.reserve(MARK_WIN + ((size_t)1 << 16))) return ~(size_t)0;
        size_t len = MARK_WIN;
        if (block_symbols_u16(r, scratch, len, T, 8192, true) == 0) continue;
        return b;
    }
    return ~(size_t)0;
}
#ifdef GZ_PHASE_TIMING
inline double g_gz_phase[6];                                              // tools/micro/inflate_par_bench.cpp: block starts, pieces, markers, CRC (seconds)
inline double gz_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#define GZ_T(x) const double x = gz_now()
#define GZ_ACC(k, a, b) g_gz_phase[k] += (b) - (a)
#else
#define GZ_T(x)
#define GZ_ACC(k, a, b)
#endif
struct ParHooks { void (*run)(size_t n, void (*f)(size_t, void*), void* ctx) = nullptr; unsigned threads = 1; };   // how the caller runs n jobs side by side (io.cpp: the worker pool)
inline ParHooks& par_hooks() { static ParHooks h; return h; }
struct Piece { size_t start = 0, end_bit = 0, len = 0; bool ok = false, final_seen = false, used = false; U16Buf buf; };
// one member's deflate stream [q, end) on `threads` threads into out[start .. ): true + *trailer = the byte behind the stream when it worked; false: nothing was written that
// matters (the caller decodes sequentially)
inline bool inflate_member_parallel(const u8* q, const u8* end, BigBuf& out, size_t& len, unsigned threads, const u8** trailer) {
    const size_t n = (size_t)(end - q);
    if (threads < 2 || n < ((size_t)4 << 20) || !par_hooks().run) return false;
    const unsigned Tn = (unsigned)std::min<size_t>(std::min<unsigned>(threads, 16), n / ((size_t)2 << 20));
    if (Tn < 2) return false;
    std::vector<std::unique_ptr<Piece>> pieces_tl;                        // the symbol buffers (2 bytes per output byte of the later pieces) go back to the system when the member is done
    while (pieces_tl.size() < Tn) pieces_tl.emplace_back(new Piece());
    struct PieceView { std::vector<std::unique_ptr<Piece>>* v; Piece& operator[](size_t k) { return *(*v)[k]; } } pc{&pieces_tl};
    // cuts: the byte-wise decoder of piece 0 is ~1.5-2 x as fast as the symbol decoder of the others (16-bit stores, no look-ahead), measured on FASTQ text
    constexpr double FIRST_SHARE = 1.8;
    const double share = (double)n / (FIRST_SHARE + (Tn - 1));
    struct Ctx { const u8* q; const u8* end; std::vector<std::unique_ptr<Piece>>* pc; std::vector<size_t> cut; std::vector<size_t> stops; BigBuf* out; size_t* len; size_t member_start; bool ok0 = false; size_t end0 = 0; bool final0 = false; size_t expect_syms = 0; } cx;
    cx.q = q; cx.end = end; cx.pc = &pieces_tl; cx.out = &out; cx.len = &len; cx.member_start = len;
    cx.cut.resize(Tn);
    {   // symbols a later piece will hold: its share of the compressed bytes at the member's overall ratio (ISIZE of the trailer: exact for files below 4 GB, a hint otherwise -- believed up to 16 : 1; a piece that outgrows its mapping grows it as before) + 12 %
        const u32 isz = end[-4] | (end[-3] << 8) | (end[-2] << 16) | ((u32)end[-1] << 24);
        if (isz > n && (size_t)isz < n * 16) cx.expect_syms = (size_t)((double)isz / (FIRST_SHARE + (Tn - 1)) * 1.12);
    }
    for (unsigned k = 1; k < Tn; k++) cx.cut[k] = (size_t)((FIRST_SHARE + (k - 1)) * share) * 8;
    // phase 1: every later piece finds its block start
    GZ_T(t_p1);
    for (unsigned k = 0; k < Tn; k++) { pc[k].start = ~(size_t)0; pc[k].ok = false; pc[k].used = false; pc[k].len = 0; pc[k].final_seen = false; }
    par_hooks().run(Tn - 1, [](size_t j, void* v) {
        Ctx& c = *(Ctx*)v; Piece& p = *(*c.pc)[j + 1];
        static thread_local Tables T;
        p.start = find_block_start(c.q, c.end, c.cut[j + 1], (size_t)8 << 20, T, p.buf);     // a block is tens of KB of compressed data: the next header is near
    }, &cx);
    for (unsigned k = 1; k < Tn; k++) if (pc[k].start != ~(size_t)0) cx.stops.push_back(pc[k].start);
    std::sort(cx.stops.begin(), cx.stops.end());
    cx.stops.erase(std::unique(cx.stops.begin(), cx.stops.end()), cx.stops.end());
    if (cx.stops.empty()) return false;
    // phase 2: piece 0 with the byte decoder straight into `out`, the others into symbols
    GZ_T(t_p2); GZ_ACC(0, t_p1, t_p2);
    par_hooks().run(Tn, [](size_t k, void* v) {
        Ctx& c = *(Ctx*)v; Piece& p = *(*c.pc)[k];
        static thread_local Tables T;
        if (k == 0) {
            Reader r{c.q, c.end};
            size_t at = 0;
            c.ok0 = inflate_stream(r, *c.out, *c.len, T, c.member_start, c.q, &c.stops, &at);
            if (c.ok0) { if (at) c.end0 = at; else { r.to_byte(); c.end0 = (size_t)(r.in - c.q) * 8; c.final0 = true; } }
            return;
        }
        if (p.start == ~(size_t)0) return;
        if (!p.buf.reserve(MARK_WIN + std::max<size_t>((size_t)4 << 20, c.expect_syms))) return;      // one mapping for the whole piece when the trailer's ISIZE says what to expect (growing it 2 MB-wise was an mremap per step)
        for (size_t i = 0; i < MARK_WIN; i++) p.buf.p[i] = (u16)(0x8000u | i);             // marker i = the byte i of the 32768 in front of this piece
        size_t l = MARK_WIN;
        p.ok = inflate_blocks_u16(c.q, c.end, p.start, p.buf, l, T, c.stops, p.end_bit, p.final_seen);
        p.len = l - MARK_WIN;
    }, &cx);
    if (!cx.ok0) return false;
    // phase 3: follow the chain of pieces that really meet, replace markers in stream order
    GZ_T(t_p3); GZ_ACC(1, t_p2, t_p3);
    // the chain first (which pieces, in which order: bit positions only), then the 32 KB in front of every piece -- the tail of the piece before it, resolved in stream
    // order: 32768 look-ups per piece -- and then ALL pieces replace their markers side by side in one pass (until round 6's last session: piece after piece, each one
    // spread over the threads, seven barriers and a tail per member)
    size_t at = cx.end0; bool fin = cx.final0;
    std::vector<Piece*> chain; size_t total = len;
    while (!fin) {
        Piece* nx = nullptr;
        for (unsigned k = 1; k < Tn; k++) if (pc[k].start == at && pc[k].ok && !pc[k].used) { nx = &pc[k]; break; }
        if (!nx) return false;                                       // the stream goes on where no piece started (cannot happen: a piece only stops on another's start) or that piece failed
        nx->used = true; chain.push_back(nx); total += nx->len;
        at = nx->end_bit; fin = nx->final_seen;
    }
    if (!chain.empty()) {
        if (len < MARK_WIN) return false;                            // (a piece in front of which fewer than 32768 bytes lie: pieces are megabytes apart)
        if (!out.reserve(total + 64)) return false;
        std::vector<u8> wins((chain.size() > 1 ? chain.size() - 1 : 0) * MARK_WIN);
        struct RCtx { std::vector<Piece*>* chain; std::vector<const u8*> win; std::vector<u8*> dst; unsigned parts; } rc{&chain, {}, {}, std::max(1u, par_hooks().threads)};
        rc.win.resize(chain.size()); rc.dst.resize(chain.size());
        size_t off = len;
        for (size_t k = 0; k < chain.size(); k++) {
            rc.dst[k] = out.p + off; off += chain[k]->len;
            if (k == 0) { rc.win[0] = out.p + len - MARK_WIN; continue; }
            u8* w = wins.data() + (k - 1) * MARK_WIN; const Piece& pv = *chain[k - 1]; const u8* wp = rc.win[k - 1]; const u16* src = pv.buf.p + MARK_WIN;
            for (size_t i = 0; i < MARK_WIN; i++) {                  // byte i of the window = output byte (len - MARK_WIN + i) of the piece before, or a byte of ITS window when that piece is shorter than a window
                const long long j2 = (long long)pv.len - (long long)MARK_WIN + (long long)i;
                if (j2 < 0) { w[i] = wp[MARK_WIN + j2]; continue; }
                const u16 x = src[j2]; w[i] = x & 0x8000u ? wp[x & 0x7FFFu] : (u8)x;
            }
            rc.win[k] = w;
        }
        par_hooks().run(chain.size() * rc.parts, [](size_t t, void* v) {
            RCtx& r2 = *(RCtx*)v;
            const size_t k = t / r2.parts, j2 = t % r2.parts; const Piece& pk = *(*r2.chain)[k];
            const size_t lo = pk.len * j2 / r2.parts, hi = pk.len * (j2 + 1) / r2.parts;
            const u16* src = pk.buf.p + MARK_WIN; u8* dst = r2.dst[k]; const u8* win = r2.win[k];
            for (size_t i = lo; i < hi; i++) { const u16 x = src[i]; dst[i] = x & 0x8000u ? win[x & 0x7FFFu] : (u8)x; }
        }, &rc);
        len = total;
    }
    *trailer = q + ((at + 7) >> 3);
    GZ_T(t_p4); GZ_ACC(2, t_p3, t_p4);
    return true;
}

// CRC-32 of a member's bytes: on one thread, or in pieces side by side whose values zlib's crc32_combine joins (a member inflated on several threads is hundreds of MB:
// 30-55 ms of one core otherwise, a seventh of the whole load)
inline u32 crc32_member(const u8* p, size_t n, unsigned threads) {
    const size_t parts = std::min<size_t>(std::min<size_t>(threads, 64), n / ((size_t)4 << 20));
    if (parts < 2 || !par_hooks().run) return crc32_fast(0, p, n);
    struct Ctx { const u8* p; size_t n, parts; u32 crc[64]; } cx{p, n, parts, {}};
    par_hooks().run(parts, [](size_t k, void* v) { Ctx& c = *(Ctx*)v; const size_t lo = c.n * k / c.parts, hi = c.n * (k + 1) / c.parts; c.crc[k] = crc32_fast(0, c.p + lo, hi - lo); }, &cx);
    uLong crc = cx.crc[0];
    for (size_t k = 1; k < parts; k++) crc = crc32_combine(crc, cx.crc[k], (z_off_t)(n * (k + 1) / parts - n * k / parts));
    return (u32)crc;
}

// every gzip member of [src, src + n) into out / len; stops (successfully) at bytes that do not start another member, as zlib's gzread does.
// why: set on failure.  CRC-32 and ISIZE of every member are checked.
inline bool gunzip_all(const u8* src, size_t n, BigBuf& out, size_t& len, std::string& why, unsigned threads = 1) {
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
        const u8* tr = nullptr;
        if (threads > 1 && inflate_member_parallel(q, end, out, len, threads, &tr)) r.in = tr;       // r.buf is empty: the trailer starts on a byte
        else {
            len = start;
            r = Reader{q, end};
            if (!inflate_stream(r, out, len, T, start)) { why = "corrupt or truncated deflate stream"; return false; }
            r.to_byte();
        }
        if (end - r.in < 8) { why = "truncated gzip trailer"; return false; }
        const u32 want_crc = r.in[0] | (r.in[1] << 8) | (r.in[2] << 16) | ((u32)r.in[3] << 24), want_len = r.in[4] | (r.in[5] << 8) | (r.in[6] << 16) | ((u32)r.in[7] << 24);
        if ((u32)(len - start) != want_len) { why = "gzip length check failed"; return false; }
        GZ_T(t_c0);
        if (crc32_member(out.p + start, len - start, threads) != want_crc) { why = "gzip CRC-32 check failed"; return false; }
        GZ_T(t_c1); GZ_ACC(3, t_c0, t_c1);
        p = r.in + 8;
    }
    if (first) { why = "not a gzip file"; return false; }
    if (end - p >= 2 && end - p < 18 && p[0] == 0x1f && p[1] == 0x8b) { why = "truncated trailing gzip member"; return false; }   // zlib reports a truncated stream here
    return true;
}

}  // namespace gz
}  // namespace savont
