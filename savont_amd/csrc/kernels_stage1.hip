// kernels_stage1.hip -- K0 (2-bit pack), K1 (split k-mer emit), K2 (fused emit + hash count).
//
// Reference semantics: src/types.rs:92-101 (encoding), src/seeding.rs:975-1068 (split_kmer_mid),
// src/seq_parse.rs:362-373 (` rc` reads), :455-460 (count by full canonical k-mer, strand = bit 63),
// :33-46 (strand / multiplicity filter).
//
// Layout / mapping (MI355X): one 64-lane wavefront per read.  Lane l handles k-mer END positions
// base+l, so the quality bytes of a chunk are one coalesced 64-byte line and the packed words
// are 4-5 consecutive dwords served from L1.  The k-mer is rebuilt from the packed words with a
// 64-bit funnel (no rolling state => no inter-lane dependency); the reverse complement is one
// v_bfrev pair + a pair swap.  Counting goes straight into an open-addressing table in HBM with
// one 64-bit CAS (claims the slot) + one 32-bit atomic add per k-mer: no 1.2 GB intermediate list.
#include "svt_internal.hpp"

// ------------------------------------------------------------------------------------------------
// K0: ASCII -> 2-bit words + non-ACGT mask + per-read flags
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 code_of(u8 b, u32& bad) {
    u32 u = b & 0xDF;                     // upper-case fold
    u32 c;
    if (u == 'A') c = 0; else if (u == 'C') c = 1; else if (u == 'G') c = 2; else if (u == 'T' || u == 'U') c = 3; else { c = 0; bad = 1; }
    if (b <= 3) { c = b; }                // row 0 of BYTE_TO_SEQ: bytes 0..3 map to themselves (types.rs:93)
    return c;
}

// one wave per read, lane = word: the 16 bases of a word are ONE 16-byte load (a read starts at any byte offset: gfx950 takes unaligned dwordx4 loads), the tail of a read
// goes byte by byte; the same wave then scans the read's qualities 16 bytes per lane for "all equal" and writes the read's flags (two kernels and 16 byte loads per lane
// until round 6)
__global__ void __launch_bounds__(256) k_pack(const u8* __restrict__ ascii, const u8* __restrict__ qual, const u64* __restrict__ off, const u64* __restrict__ woff,
                                              u32 n, u32* __restrict__ packed, u16* __restrict__ nmask, u8* __restrict__ flags) {
    u32 r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    u32 lane = d_lane();
    u64 o = off[r]; u32 len = (u32)(off[r + 1] - o);
    u64 wo = woff[r]; u32 nw = (u32)(woff[r + 1] - wo);       // includes the pad words
    int hasn = 0;
    for (u32 w = lane; w < nw; w += 64) {
        u32 word = 0, m = 0;
        const u32 b0 = w * 16;
        if (b0 + 16 <= len) {
            uint4 v; __builtin_memcpy(&v, ascii + o + b0, 16);
            const u32 d[4] = {v.x, v.y, v.z, v.w};
            #pragma unroll
            for (u32 j = 0; j < 16; j++) {
                u32 bad = 0;
                const u32 c = code_of((u8)(d[j >> 2] >> (8 * (j & 3))), bad);
                word |= c << (30 - 2 * j);
                m |= bad << (15 - j);
            }
        } else {
            #pragma unroll
            for (u32 j = 0; j < 16; j++) {
                const u32 i = b0 + j;
                if (i < len) {
                    u32 bad = 0;
                    const u32 c = code_of(ascii[o + i], bad);
                    word |= c << (30 - 2 * j);
                    m |= bad << (15 - j);
                }
            }
        }
        packed[wo + w] = word;
        nmask[wo + w] = (u16)m;
        hasn |= (m != 0);
    }
    int diff = 0;
    if (qual && len) {
        const u32 q0 = qual[o], q4 = q0 * 0x01010101u;
        for (u32 base = 0; base < len; base += 1024) {                             // 1024 bytes per round; the first round nearly always finds two different qualities
            const u32 i = base + lane * 16;
            if (i + 16 <= len) { uint4 v; __builtin_memcpy(&v, qual + o + i, 16); diff |= ((v.x ^ q4) | (v.y ^ q4) | (v.z ^ q4) | (v.w ^ q4)) != 0; }
            else for (u32 j = i; j < len; j++) diff |= (qual[o + j] != q0);
            if (__ballot(diff)) break;
        }
    }
    const ull anydiff = __ballot(diff), anyn = __ballot(hasn);
    if (lane == 0) flags[r] = (u8)(((qual && anydiff == 0) ? 1 : 0) | (anyn ? 2 : 0));
}

int launch_pack(svt_ctx* c, svt_batch* b, const u8* d_ascii) {
    if (b->n == 0) return SVT_OK;
    ProfScope ps(c, "k_pack", (double)b->total_bases * (b->has_qual ? 2.0 : 1.0) + b->total_words * 6.0, b->n);
    u32 blocks = (b->n + 3) / 4;
    hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, c->stream, d_ascii, b->view().qual, b->d_off, b->d_woff, b->n, b->d_packed, b->d_nmask, b->d_flags);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// ------------------------------------------------------------------------------------------------
// K1 / K2
// ------------------------------------------------------------------------------------------------
#define HT_MAX_PROBE 4096u
__device__ __forceinline__ void ht_insert(HtEntry* __restrict__ t, u64 mask, u64 key, u32 strand, u32* __restrict__ overflow) {
    u64 h = d_mm_hash64(key) & mask;
    u32 probes = 0;
    while (true) {
        if (++probes > HT_MAX_PROBE) { *overflow = 1; return; }      // table too small for this input: the host retries with 4x the capacity
        if ((probes & 63u) == 0 && *(volatile u32*)overflow) return;  // somebody found it full already: the pass is void, do not walk 4096 slots per key to learn the same (a table sized from the step before can meet a sample with many more keys)
        ull cur = t[h].key;                       // a non-empty slot never changes again: a stale EMPTY is resolved by the CAS
        if (cur == key) break;
        if (cur == SVT_EMPTY_KEY) {
            ull old = atomicCAS(&t[h].key, SVT_EMPTY_KEY, (ull)key);
            if (old == SVT_EMPTY_KEY || old == key) break;
        }
        h = (h + 1) & mask;
    }
    atomicAdd(&t[h].c[strand], 1u);
}

template <bool COUNT>
__global__ void __launch_bounds__(256) k_split_kmers(BatchView bv, u32 k, u32 min_bq, const u8* __restrict__ rc_flags,
                                                     const u64* __restrict__ out_off, u64* __restrict__ out, u32* __restrict__ out_cnt,
                                                     HtEntry* __restrict__ ht, u64 ht_mask, u32* __restrict__ overflow) {
    u32 r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= bv.n) return;
    const u32 lane = d_lane();
    const u64 o = bv.off[r];
    const u32 len = (u32)(bv.off[r + 1] - o);
    if (len < k) { if (!COUNT && lane == 0) out_cnt[r] = 0; return; }          // seeding.rs:982
    const u32* w = bv.packed + bv.woff[r];
    const u16* nm = bv.nmask + bv.woff[r];
    const u8 fl = bv.flags[r];
    const bool use_q = bv.qual && !(fl & 1);                                    // seeding.rs:1004-1008
    const bool rc = rc_flags && rc_flags[r];
    const bool has_n = fl & 2;
    const u8* q = bv.qual + o;
    const u32 npos = len - k + 1;
    const u32 mid_k = k / 2;
    const u64 split_mask = ~(3ull << (k - 1));
    u32 cnt = 0;
    const u64 obase = COUNT ? 0 : out_off[r];
    for (u32 base = 0; base < npos; base += 64) {
        u32 e = base + lane;                       // emission index = (k-mer end) - (k-1) in the possibly-rc'd read
        bool valid = e < npos;
        u32 p = valid ? (rc ? (len - k - e) : e) : 0;   // window start in stored coordinates
        u64 fo = d_window64(w, p) >> (64 - 2 * k);
        u64 f, rv;
        if (rc) {
            // the rc'd read's forward k-mer is revcomp(window) with non-ACGT bases forced to A on BOTH
            // strands' views (utils.rs:51-65 maps them to 'N', types.rs:92-101 maps 'N' to 0)
            u64 F = fo;
            if (has_n) F |= d_nmask_kmer(nm, p, k);
            f = d_revcomp(F, k); rv = F;
        } else { f = fo; rv = d_revcomp(fo, k); }
        u64 sf = f & split_mask, sr = rv & split_mask;
        bool ok = valid && (sf != sr);                                          // :1044
        if (use_q && ok) ok = ((u8)(q[p + mid_k] - 33)) >= min_bq;              // :1010-1011,:1049 (mid base is strand-symmetric)
        bool canon = sf < sr;                                                   // :1053
        u64 km = canon ? f : rv;
        if (COUNT) {
            if (ok) ht_insert(ht, ht_mask, km, canon ? 1u : 0u, overflow);
        } else {
            ull m = __ballot(ok);
            if (ok) out[obase + cnt + d_rank(m)] = km | ((u64)canon << 63);     // :1063
            cnt += __popcll(m);
        }
    }
    if (!COUNT && lane == 0) out_cnt[r] = cnt;
}

// K2, windowed: amplicon reads are the same few sequences over and over, so the k-mers that END inside one 64-position window
// of many reads are a few thousand distinct values hit ~1000 times each.  A block takes one window of WIN_READS consecutive reads
// (a wave per read at a time, lane = position, as above), counts into an LDS table first (64-bit ds_cmpst claims a slot, ds_add
// counts) and sends to the HBM table only what does not find a slot within WIN_PROBES probes (sequencing-error k-mers, mostly
// singletons) plus, at the end, one (key, rev, fwd) triple per occupied slot.  The sums are order-independent: the table is the one
// the read-per-wave kernel builds, with ~6x fewer device-scope atomics (every one of which is a 32-byte memory-side write).
// Windows are dealt to blockIdx so that a window's blocks share an XCD (blockIdx % 8) and hence the L2 copies of its table lines.
#define WIN_SLOTS 4096u
#define WIN_READS 1024u
#define WIN_PROBES 4u
__device__ __forceinline__ void ht_insert_n(HtEntry* __restrict__ t, u64 mask, u64 key, u64 hash, u32 c0, u32 c1, u32* __restrict__ overflow) {
    u64 h = hash & mask;
    u32 probes = 0;
    while (true) {
        if (++probes > HT_MAX_PROBE) { *overflow = 1; return; }
        if ((probes & 63u) == 0 && *(volatile u32*)overflow) return;
        ull cur = t[h].key;
        if (cur == key) break;
        if (cur == SVT_EMPTY_KEY) {
            ull old = atomicCAS(&t[h].key, SVT_EMPTY_KEY, (ull)key);
            if (old == SVT_EMPTY_KEY || old == key) break;
        }
        h = (h + 1) & mask;
    }
    // both counts of a window's slot in ONE 64-bit add (the pair is eight bytes, eight-byte aligned; a count stays below 2^32, so nothing carries from c[0] into c[1])
    if (c0 && c1) atomicAdd(reinterpret_cast<ull*>(&t[h].c[0]), (ull)c0 | ((ull)c1 << 32));
    else if (c0) atomicAdd(&t[h].c[0], c0);
    else if (c1) atomicAdd(&t[h].c[1], c1);
}

#define WIN_MQ 128u
// the first `cnt` (<= 64) queued misses of a wave go to the HBM table, a lane each; the rest (< 64) move to the front.  One wave: its LDS requests are served in order
__device__ __forceinline__ void win_flush_misses(ull* mq, const u32 qn, const u32 cnt, HtEntry* __restrict__ ht, u64 ht_mask, u32* __restrict__ overflow) {
    const u32 lane = d_lane();
    __builtin_amdgcn_wave_barrier();
    const ull mine = lane < cnt ? mq[lane] : 0;
    const u32 rest = qn - cnt;
    const ull moved = lane < rest ? mq[cnt + lane] : 0;
    if (lane < cnt) { const u64 km = mine & ~(1ull << 63); const u32 canon = (u32)(mine >> 63); ht_insert_n(ht, ht_mask, km, d_mm_hash64(km), canon ? 0u : 1u, canon ? 1u : 0u, overflow); }
    __builtin_amdgcn_wave_barrier();
    if (lane < rest) mq[lane] = moved;
    __builtin_amdgcn_wave_barrier();
}
// the window's slots go to the HBM table (512 threads): a thread's eight slots look their first table entry up TOGETHER (eight independent loads in flight; most keys are
// there already -- the same k-mers fill the windows of every block) and add straight to it when it holds their key; the others take the probing path
template <u32 SLOTS = WIN_SLOTS, u32 THREADS = 512u>
__device__ __forceinline__ void win_flush_slots(const ull* skey, const u32* scnt, HtEntry* __restrict__ ht, u64 ht_mask, u32* __restrict__ overflow) {
    constexpr u32 SPT = SLOTS / THREADS;
    ull keys[SPT], cur[SPT]; u64 hh[SPT];
    #pragma unroll
    for (u32 j = 0; j < SPT; j++) { keys[j] = skey[threadIdx.x + THREADS * j]; hh[j] = keys[j] != SVT_EMPTY_KEY ? (d_mm_hash64(keys[j]) & ht_mask) : 0; }
    #pragma unroll
    for (u32 j = 0; j < SPT; j++) cur[j] = keys[j] != SVT_EMPTY_KEY ? ht[hh[j]].key : SVT_EMPTY_KEY;
    #pragma unroll
    for (u32 j = 0; j < SPT; j++) {
        if (keys[j] == SVT_EMPTY_KEY) continue;
        const u32 i = threadIdx.x + THREADS * j, c0 = scnt[2 * i], c1 = scnt[2 * i + 1];
        if (cur[j] == keys[j]) {
            if (c0 && c1) atomicAdd(reinterpret_cast<ull*>(&ht[hh[j]].c[0]), (ull)c0 | ((ull)c1 << 32));
            else if (c0) atomicAdd(&ht[hh[j]].c[0], c0);
            else if (c1) atomicAdd(&ht[hh[j]].c[1], c1);
        } else ht_insert_n(ht, ht_mask, keys[j], d_mm_hash64(keys[j]), c0, c1, overflow);
    }
}
__global__ void __launch_bounds__(512) k_split_kmers_count_win(BatchView bv, u32 k, u32 min_bq, const u8* __restrict__ rc_flags, u32 nwin, u32 nw8,
                                                               HtEntry* __restrict__ ht, u64 ht_mask, u32* __restrict__ overflow) {
    constexpr u32 U = 4;                                   // reads in flight per wave: their loads are issued before any is used
    extern __shared__ ull win_lds[];
    ull* skey = win_lds;                                   // [WIN_SLOTS]
    u32* scnt = (u32*)(skey + WIN_SLOTS);                  // [WIN_SLOTS][2]
    // round 6: the k-mers that find no slot of the LDS table (a quarter of a window's: sequencing-error k-mers, mostly singletons) used to go to the HBM table where they were
    // met -- a dependent load (+ a claim) per sub-step of the unrolled loop, with a quarter of the lanes taking part.  They queue up per wave instead (k-mer, strand in bit 63) and
    // go out 64 at a time: one round of table latency per 64 inserts, every lane busy.
    ull* mq = (ull*)(scnt + 2 * WIN_SLOTS) + (size_t)(threadIdx.x >> 6) * WIN_MQ;   // [waves][WIN_MQ]
    u32 qn = 0;                                            // entries waiting (wave-uniform, < 128)
    const u32 y = blockIdx.x >> 3;
    const u32 win = (blockIdx.x & 7u) + 8u * (y % nw8), grp = y / nw8;
    if (win >= nwin) return;
    for (u32 i = threadIdx.x; i < WIN_SLOTS; i += blockDim.x) { skey[i] = SVT_EMPTY_KEY; scnt[2 * i] = 0; scnt[2 * i + 1] = 0; }
    __syncthreads();
    const u32 lane = d_lane(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const u32 r0 = grp * WIN_READS, r1 = min(bv.n, r0 + WIN_READS);
    const u32 mid_k = k / 2;
    const u64 split_mask = ~(3ull << (k - 1));
    const u32 per_wave = (WIN_READS + nwaves - 1) / nwaves;
    const u32 wr0 = r0 + wave * per_wave, wr1 = min(r1, wr0 + per_wave);
    for (u32 cb = wr0; cb < wr1; cb += 64) {
        // lane-parallel read descriptors of the next 64 reads of this wave, handed out below by v_readlane
        const u32 rr = cb + lane;
        u64 m_off = 0, m_woff = 0; u32 m_len = 0, m_fl = 0;
        if (rr < wr1) {
            m_off = bv.off[rr]; m_len = (u32)(bv.off[rr + 1] - m_off); m_woff = bv.woff[rr];
            m_fl = (u32)bv.flags[rr] | ((rc_flags && rc_flags[rr]) ? 4u : 0u);
            if (m_len < k || win * 64u >= m_len - k + 1) m_fl |= 8u;                // seeding.rs:982, or nothing of this read in the window
        } else m_fl = 8u;
        const u32 cn = min(64u, wr1 - cb);
        for (u32 it = 0; it < cn; it += U) {
            u32 w0[U], w1[U], w2[U], nm0[U], nm1[U], nm2[U], pp[U], fl[U]; u8 qv[U]; bool valid[U];
            #pragma unroll
            for (u32 u = 0; u < U; u++) {
                const u32 src = min(it + u, 63u);
                const u64 o = ((u64)(u32)__builtin_amdgcn_readlane((int)(m_off >> 32), src) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)m_off, src);
                const u64 wo = ((u64)(u32)__builtin_amdgcn_readlane((int)(m_woff >> 32), src) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)m_woff, src);
                const u32 len = (u32)__builtin_amdgcn_readlane((int)m_len, src);
                fl[u] = (it + u < cn) ? (u32)__builtin_amdgcn_readlane((int)m_fl, src) : 8u;
                const bool skip = fl[u] & 8u;
                const u32 npos = skip ? 0u : len - k + 1;
                const u32 e = win * 64u + lane;
                valid[u] = e < npos;
                const u32 p = valid[u] ? ((fl[u] & 4u) ? (len - k - e) : e) : 0u;
                pp[u] = p;
                w0[u] = w1[u] = w2[u] = 0; nm0[u] = nm1[u] = nm2[u] = 0; qv[u] = 255;
                if (!skip) {
                    const u32* w = bv.packed + wo + (p >> 4);
                    w0[u] = w[0]; w1[u] = w[1]; w2[u] = w[2];
                    if ((fl[u] & 6u) == 6u) { const u16* m = bv.nmask + wo + (p >> 4); nm0[u] = m[0]; nm1[u] = m[1]; nm2[u] = m[2]; }
                    if (bv.qual && !(fl[u] & 1u) && valid[u]) qv[u] = bv.qual[o + p + mid_k];           // seeding.rs:1004-1008
                }
            }
            #pragma unroll
            for (u32 u = 0; u < U; u++) {
                if (fl[u] & 8u) continue;
                const u32 sh = (pp[u] & 15u) * 2;
                const u64 A = ((u64)w0[u] << 32) | w1[u];
                const u64 fo = (sh == 0 ? A : ((A << sh) | ((u64)w2[u] >> (32 - sh)))) >> (64 - 2 * k);
                u64 f, rv;
                if (fl[u] & 4u) {
                    u64 F = fo;
                    if (fl[u] & 2u) {
                        const u32 ob = pp[u] & 15u;
                        const u64 M = ((u64)nm0[u] << 32) | ((u64)nm1[u] << 16) | (u64)nm2[u];
                        u64 x = (M >> (48 - ob - k)) & ((1ull << k) - 1);
                        x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull; x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
                        x = (x | (x << 2)) & 0x3333333333333333ull; x = (x | (x << 1)) & 0x5555555555555555ull;
                        F |= x | (x << 1);
                    }
                    f = d_revcomp(F, k); rv = F;
                } else { f = fo; rv = d_revcomp(fo, k); }
                const u64 sf = f & split_mask, sr = rv & split_mask;
                bool ok = valid[u] && (sf != sr);                                   // :1044
                if (bv.qual && !(fl[u] & 1u) && ok) ok = ((u8)(qv[u] - 33)) >= min_bq;   // :1010-1011,:1049
                const bool canon = sf < sr;                                         // :1053
                const u64 km = canon ? f : rv;
                bool miss = false;
                if (ok) {
                    const u64 hash = d_mm_hash64(km);
                    u32 h = (u32)(hash >> 40) & (WIN_SLOTS - 1);
                    bool placed = false;
                    for (u32 t = 0; t < WIN_PROBES; t++) {
                        ull cur = skey[h];
                        if (cur == SVT_EMPTY_KEY) cur = atomicCAS(&skey[h], SVT_EMPTY_KEY, (ull)km);
                        if (cur == SVT_EMPTY_KEY || cur == km) { atomicAdd(&scnt[2 * h + (canon ? 1u : 0u)], 1u); placed = true; break; }
                        h = (h + 1) & (WIN_SLOTS - 1);
                    }
                    miss = !placed;
                }
                const ull mb = __ballot(miss);
                if (mb) {                                                           // wave-uniform
                    if (miss) mq[qn + d_rank(mb)] = (ull)km | ((ull)(canon ? 1 : 0) << 63);
                    qn += (u32)__popcll(mb);
                    if (qn >= 64) { win_flush_misses(mq, qn, 64u, ht, ht_mask, overflow); qn -= 64; }
                }
            }
        }
    }
    if (qn) win_flush_misses(mq, qn, qn, ht, ht_mask, overflow);
    __syncthreads();
    win_flush_slots(skey, scnt, ht, ht_mask, overflow);
}

// K2, windowed, a LANE per read (round 6; the default).  The kernel above puts the 64 positions of a window on the lanes and walks the reads: every read costs a
// set of v_readlane + scalar bookkeeping, a 64-bit funnel, a bit-reversal and the seeding hash per position, and its control flow (four unrolled reads, strand and N
// branches, a four-probe loop) is ~160 scalar and ~150 vector instructions per read and window (SQ counters, profiles/r06_stage1_pmc_before.txt) -- the scalar unit of a CU
// is shared by its four SIMDs, and it was the busier one.  Here a lane owns a READ and the wave walks the window's 64 positions together:
//  * the k-mer and its reverse complement ROLL (two shifts and an OR each per position; the stream of the window's 64 new bases is two registers built once per read);
//  * COUNTING NEEDS NO EMISSION ORDER: a position contributes the same (k-mer, strand) whatever window it is counted in, so a ` rc` read (seq_parse.rs:362-373) is walked
//    in STORED coordinates like any other read and only swaps the roles of the two strands -- X = the stored window (non-ACGT forced to T when the read is ` rc`,
//    so that its reverse complement carries A: utils.rs:51-65, types.rs:92-101), Y = revcomp(X); the k-mer kept is the one with the smaller split value either way,
//    and the strand bit is (X is the smaller) XOR rc;
//  * the mid-base quality test (seeding.rs:1004-1011) of the 64 positions is one 64-bit mask per read, built from four 16-byte loads before the walk;
//  * the LDS table is probed with a cheap 12-bit hash of its own (placement inside a window's table is private to the block; the HBM table keeps mm_hash64).
// The HBM table receives the same sums as from either other kernel (tests/test_gpu_kernels.py::test_count_kernels_agree).
__device__ __forceinline__ u64 d_spread32(u32 bits) {                    // bit j -> bits 2j, 2j+1
    u64 x = bits;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull; x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull; x = (x | (x << 1)) & 0x5555555555555555ull;
    return x | (x << 1);
}
__device__ __forceinline__ u32 d_win_mix(u64 km) {                       // 32 mixed bits of a k-mer for the window's LDS structures: low 12 = table bucket, high 16 = first-sighting bit
    const u32 lo = (u32)km, hi = (u32)(km >> 32);
    u32 x = lo ^ ((hi << 13) | (hi >> 19));
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    return x;
}
// one block per CU: 16 waves share a table of 8192 slots (128 KB) over a window of 2048 reads -- the real k-mers of a window (~2000, both strands) fill a quarter of it
#define WIN_WALK 6u                                                      // pairs of slots a k-mer may lie behind its bucket
// the first `cnt` (<= 64) queued k-mers of a wave, a lane each; the rest (< 64) move to the front.  A k-mer is given a slot of the window's table only on its SECOND
// sighting in the block (a bit per 16-bit hash value remembers the first): sequencing-error k-mers are mostly seen once and would otherwise fill the table before the
// real k-mers of the later positions arrive -- and a real k-mer without a slot goes to the HBM table a thousand times per block, from many lanes at once (same-address
// atomics are served one by one: measured, 3 ms per launch when a quarter of them had none).  First sightings go to the HBM table; seen k-mers walk from their bucket
// (linear probing, pairs of slots) to their slot or to the first free one; a walk that finds neither within WIN_WALK pairs ends in the HBM table as well.
template <u32 WL_SLOTS, u32 WL_FILTW>
__device__ __forceinline__ void win_slow_batch(ull* mq, const u32 qn, const u32 cnt, ull* skey, u32* scnt, u32* filt, HtEntry* __restrict__ ht, u64 ht_mask, u32* __restrict__ overflow) {
    const u32 lane = d_lane();
    __builtin_amdgcn_wave_barrier();
    const ull mine = lane < cnt ? mq[lane] : 0;
    const u32 rest = qn - cnt;
    const ull moved = lane < rest ? mq[cnt + lane] : 0;
    if (lane < cnt) {
        const u64 km = mine & ~(1ull << 63); const u32 cbit = (u32)(mine >> 63);
        const u32 hx = d_win_mix(km);
        const u32 f = (hx >> 16) & (WL_FILTW * 32u - 1u), bit = 1u << (f & 31u);
        int slot = -1;
        if (atomicOr(&filt[f >> 5], bit) & bit) {
            u32 b_ = (hx & (WL_SLOTS - 1)) & ~1u;
            for (u32 pr = 0; pr < WIN_WALK; pr++) {
                ull a0 = skey[b_], a1 = skey[b_ + 1];
                if (a0 == SVT_EMPTY_KEY) { a0 = atomicCAS(&skey[b_], SVT_EMPTY_KEY, (ull)km); if (a0 == SVT_EMPTY_KEY) a0 = km; }
                if (a0 == km) { slot = (int)b_; break; }
                if (a1 == SVT_EMPTY_KEY) { a1 = atomicCAS(&skey[b_ + 1], SVT_EMPTY_KEY, (ull)km); if (a1 == SVT_EMPTY_KEY) a1 = km; }
                if (a1 == km) { slot = (int)b_ + 1; break; }
                b_ = (b_ + 2) & (WL_SLOTS - 1);
            }
        }
        if (slot >= 0) atomicAdd(&scnt[2 * slot + cbit], 1u);
        else ht_insert_n(ht, ht_mask, km, d_mm_hash64(km), cbit ? 0u : 1u, cbit ? 1u : 0u, overflow);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < rest) mq[lane] = moved;
    __builtin_amdgcn_wave_barrier();
}
template <u32 WL_THREADS, u32 WL_SLOTS, u32 WL_READS, u32 WL_FILTW>
__global__ void __launch_bounds__(WL_THREADS) k_split_kmers_count_lanes(BatchView bv, u32 k, u32 min_bq, const u8* __restrict__ rc_flags, u32 nwin, u32 nw8,
                                                                 HtEntry* __restrict__ ht, u64 ht_mask, u32* __restrict__ overflow) {
    extern __shared__ ull win_lds[];
    ull* skey = win_lds;                                   // [WL_SLOTS]
    u32* scnt = (u32*)(skey + WL_SLOTS);                  // [WL_SLOTS][2]
    u32* filt = scnt + 2 * WL_SLOTS;                      // [WL_FILTW]
    ull* mq = (ull*)(filt + WL_FILTW) + (size_t)(threadIdx.x >> 6) * WIN_MQ;   // [waves][WIN_MQ]
    u32 qn = 0;
    const u32 y = blockIdx.x >> 3;
    const u32 win = (blockIdx.x & 7u) + 8u * (y % nw8), grp = y / nw8;
    if (win >= nwin) return;
    for (u32 i = threadIdx.x; i < WL_SLOTS; i += blockDim.x) { skey[i] = SVT_EMPTY_KEY; scnt[2 * i] = 0; scnt[2 * i + 1] = 0; }
    for (u32 i = threadIdx.x; i < WL_FILTW; i += blockDim.x) filt[i] = 0;
    __syncthreads();
    const u32 lane = d_lane(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const u32 r0 = grp * WL_READS, r1 = min(bv.n, r0 + WL_READS);
    const u32 per_wave = (WL_READS + nwaves - 1) / nwaves;
    const u32 wr0 = r0 + wave * per_wave, wr1 = min(r1, wr0 + per_wave);
    const u32 p0 = win * 64u, km1 = k - 1, mid_k = k / 2;  // k is odd, 3..31 (check_k): km1 = 2..30, both split bits lie in the low dword
    const u32 split_lo = ~(3u << (k - 1));
    const u64 kmask = (1ull << (2 * k)) - 1;
    const u32 sel = km1 >> 4, sh = (km1 & 15u) * 2;
    for (u32 cb = wr0; cb < wr1; cb += 64) {
        const u32 r = cb + lane;
        u32 cnt = 0, fl = 0, nw = 0; u64 o = 0, wo = 0; bool rc = false;
        if (r < wr1) {
            o = bv.off[r]; const u32 len = (u32)(bv.off[r + 1] - o); wo = bv.woff[r]; nw = (u32)(bv.woff[r + 1] - wo);
            fl = bv.flags[r]; rc = rc_flags && rc_flags[r];
            if (len >= k && p0 < len - k + 1) cnt = min(64u, len - k + 1 - p0);      // seeding.rs:982; positions of this read inside the window
        }
        if (__ballot(cnt != 0) == 0) continue;
        u64 X = 0, Y = 0, S0 = 0, S1 = 0, okm = 0;
        if (cnt) {
            // bases p0 .. p0 + 95 of the read: six words from word 4 win on (p0 is a multiple of 64); words past the read's own are its zero pad
            const u32* wp = bv.packed + wo; const u32 w0i = 4u * win, wl = nw - 1;
            u32 W[6];
            #pragma unroll
            for (u32 j = 0; j < 6; j++) W[j] = wp[min(w0i + j, wl)];
            X = (((u64)W[0] << 32) | W[1]) >> (64 - 2 * km1);                         // the first k - 1 bases; the walk adds one per step
            const u32 B0 = sel ? W[1] : W[0], B1 = sel ? W[2] : W[1], B2 = sel ? W[3] : W[2], B3 = sel ? W[4] : W[3], B4 = sel ? W[5] : W[4];
            S0 = ((u64)B0 << 32) | B1; S1 = ((u64)B2 << 32) | B3;                     // bases k - 1 .. k + 62 of the window: the base step t adds
            if (sh) { S0 = (S0 << sh) | ((u64)B2 >> (32 - sh)); S1 = (S1 << sh) | ((u64)B4 >> (32 - sh)); }
            if (rc && (fl & 2u)) {                                                   // ` rc` read with non-ACGT bases: stored as A, counted as T on this strand
                const u16* mp = bv.nmask + wo;
                u32 N[6];
                #pragma unroll
                for (u32 j = 0; j < 6; j++) N[j] = mp[min(w0i + j, wl)];
                const u64 hi = ((u64)N[0] << 48) | ((u64)N[1] << 32) | ((u64)N[2] << 16) | (u64)N[3];
                const u32 lo = (N[4] << 16) | N[5];
                X |= d_spread32((u32)(hi >> (64 - km1)));
                const u64 ns = (hi << km1) | ((u64)lo >> (32 - km1));
                S0 |= d_spread32((u32)(ns >> 32)); S1 |= d_spread32((u32)ns);
            }
            Y = d_revcomp(X, km1) << 2;
            okm = cnt == 64 ? ~0ull : ((1ull << cnt) - 1);
            if (bv.qual && !(fl & 1u)) {                                             // seeding.rs:1004-1011: the quality of the middle base of every position
                const u8* qp = bv.qual + o + p0 + mid_k;                             // (the batch's quality array has 64 bytes of slack behind its last read)
                uint4 q4[4];
                #pragma unroll
                for (u32 j = 0; j < 4; j++) __builtin_memcpy(&q4[j], qp + 16 * j, 16);
                u64 qm = 0;
                #pragma unroll
                for (u32 j = 0; j < 4; j++) {
                    const u32 d[4] = {q4[j].x, q4[j].y, q4[j].z, q4[j].w};
                    #pragma unroll
                    for (u32 b = 0; b < 16; b++) if ((u8)((d[b >> 2] >> (8 * (b & 3))) - 33) >= min_bq) qm |= 1ull << (16 * j + b);
                }
                okm &= qm;
            }
        }
        const ull rcm = __ballot(rc);
        #pragma unroll 1
        for (u32 half = 0; half < 2; half++) {
            const u64 S = half ? S1 : S0;
            #pragma unroll 1
            for (u32 tt = 0; tt < 32; tt++) {
                const u32 t = half * 32 + tt;
                if (__ballot((okm >> t) != 0) == 0) { half = 2; break; }              // nobody has a position left
                const u32 cbase = (u32)(S >> (62 - 2 * tt)) & 3u;
                X = ((X << 2) | cbase) & kmask;
                Y = (Y >> 2) | ((u64)(3u - cbase) << (2 * km1));
                const u32 sfl = (u32)X & split_lo, srl = (u32)Y & split_lo;
                const u64 sf = (X & 0xFFFFFFFF00000000ull) | sfl, sr = (Y & 0xFFFFFFFF00000000ull) | srl;
                const bool xlt = sf < sr;
                const bool ok = ((okm >> t) & 1) && sf != sr;                          // seeding.rs:1044
                const u64 km = xlt ? X : Y;
                const ull canon_m = __ballot(xlt) ^ rcm;                              // seeding.rs:1053, the strands of a ` rc` read swapped
                const u32 cbit = (u32)(canon_m >> lane) & 1u;
                // the window's table: a bucket of two slots per k-mer, one 16-byte read.  Found there: one ds_add.  Anything else -- a first sighting, a k-mer that is
                // about to get its slot, one that lies further along its probe walk -- is queued (k-mer, strand in bit 63) and handled 64 at a time by win_slow_batch
                // with every lane busy: per-lane slow paths inside this loop would be issued in nearly every step for the one or two lanes that need them
                const u32 bkt = (d_win_mix(km) & (WL_SLOTS - 1)) & ~1u;
                bool todo = false;
                if (ok) {
                    const ull c0 = skey[bkt], c1 = skey[bkt + 1];
                    if (c0 == km) atomicAdd(&scnt[2 * bkt + cbit], 1u);
                    else if (c1 == km) atomicAdd(&scnt[2 * bkt + 2 + cbit], 1u);
                    else todo = true;
                }
                const ull mb = __ballot(todo);
                if (mb) {
                    if (todo) mq[qn + d_rank(mb)] = (ull)km | ((ull)cbit << 63);
                    qn += (u32)__popcll(mb);
                    if (qn >= 64) { win_slow_batch<WL_SLOTS, WL_FILTW>(mq, qn, 64u, skey, scnt, filt, ht, ht_mask, overflow); qn -= 64; }
                }
            }
        }
    }
    if (qn) win_slow_batch<WL_SLOTS, WL_FILTW>(mq, qn, qn, skey, scnt, filt, ht, ht_mask, overflow);
    __syncthreads();
    win_flush_slots<WL_SLOTS, WL_THREADS>(skey, scnt, ht, ht_mask, overflow);
}

int launch_split_emit(svt_ctx* c, const svt_batch* b, u32 k, u8 min_bq, const u8* d_rc, const u64* d_out_off, u64* d_out, u32* d_cnt) {
    if (b->n == 0) return SVT_OK;
    double bytes = (double)b->total_words * 4.0 + (b->has_qual ? (double)b->total_bases : 0.0) + 8.0 * (double)b->total_bases;
    ProfScope ps(c, "k_split_kmers_emit", bytes, b->n);
    hipLaunchKernelGGL(k_split_kmers<false>, dim3((b->n + 3) / 4), dim3(256), 0, c->stream, b->view(), k, (u32)min_bq, d_rc, d_out_off, d_out, d_cnt,
                       (HtEntry*)nullptr, (u64)0, (u32*)nullptr);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

int launch_count_insert(svt_ctx* c, const svt_batch* b, u32 k, u8 min_bq, const u8* d_rc, u32* d_overflow) {
    if (b->n == 0) return SVT_OK;
    // algorithmic bytes: packed + qual read, 16 B table entry read-modify-write per k-mer
    double bytes = (double)b->total_words * 4.0 + (b->has_qual ? (double)b->total_bases : 0.0) + 16.0 * (double)b->total_bases;
    ProfScope ps(c, "k_split_kmers_count", bytes, b->n);
    const bool per_read = c->opt().count_kernel == 1;                              // svt_set_option("count_kernel", 1)
    if (per_read || b->max_len < k) {
        hipLaunchKernelGGL(k_split_kmers<true>, dim3((b->n + 3) / 4), dim3(256), 0, c->stream, b->view(), k, (u32)min_bq, d_rc, (const u64*)nullptr,
                           (u64*)nullptr, (u32*)nullptr, c->ht, c->ht_cap - 1, d_overflow);
    } else {
        const u32 nwin = (b->max_len - k + 1 + 63) / 64, nw8 = (nwin + 7) / 8, ngrp = (b->n + WIN_READS - 1) / WIN_READS;
        if (c->opt().count_kernel == 2) {                                          // the round-5 window kernel (a wave per read, lane = position): comparison runs
            const size_t sh = (size_t)WIN_SLOTS * 16 + (size_t)(512 / 64) * WIN_MQ * 8;      // the window's table + a miss queue per wave
            DYN_LDS_ONCE(c, 10, k_split_kmers_count_win, sh);
            hipLaunchKernelGGL(k_split_kmers_count_win, dim3(8 * nw8 * ngrp), dim3(512), sh, c->stream, b->view(), k, (u32)min_bq, d_rc, nwin, nw8,
                               c->ht, c->ht_cap - 1, d_overflow);
        } else {
            // two shapes: ONE 16-wave workgroup per CU (8192 slots, 2048 reads per block, 64 K first-sighting bits: 152 KB of LDS) -- the fastest alone on the chip -- and
            // 8 waves with 4096 slots, 1024 reads and 32 K bits (76 KB: two per CU, or one beside a K12 workgroup's 82 KB when samples are in flight; count_kernel 3)
            if (c->opt().count_kernel == 3) {
                const size_t sh = (size_t)4096 * 16 + (size_t)1024 * 4 + (size_t)8 * WIN_MQ * 8;
                const u32 ngl = (b->n + 1023) / 1024;
                DYN_LDS_ONCE(c, 12, (k_split_kmers_count_lanes<512, 4096, 1024, 1024>), sh);
                hipLaunchKernelGGL((k_split_kmers_count_lanes<512, 4096, 1024, 1024>), dim3(8 * nw8 * ngl), dim3(512), sh, c->stream, b->view(), k, (u32)min_bq, d_rc, nwin, nw8,
                                   c->ht, c->ht_cap - 1, d_overflow);
            } else {
                const size_t sh = (size_t)8192 * 16 + (size_t)2048 * 4 + (size_t)16 * WIN_MQ * 8;
                const u32 ngl = (b->n + 2047) / 2048;
                DYN_LDS_ONCE(c, 11, (k_split_kmers_count_lanes<1024, 8192, 2048, 2048>), sh);
                hipLaunchKernelGGL((k_split_kmers_count_lanes<1024, 8192, 2048, 2048>), dim3(8 * nw8 * ngl), dim3(1024), sh, c->stream, b->view(), k, (u32)min_bq, d_rc, nwin, nw8,
                                   c->ht, c->ht_cap - 1, d_overflow);
            }
        }
    }
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// ---- table init / merge / compact -------------------------------------------------------------
__global__ void k_ht_init(HtEntry* t, u64 cap) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 stride = (u64)gridDim.x * blockDim.x;
    for (; i < cap; i += stride) { HtEntry e; e.key = SVT_EMPTY_KEY; e.c[0] = 0; e.c[1] = 0; t[i] = e; }
}
int launch_ht_init(svt_ctx* c) {
    ProfScope ps(c, "k_ht_init", 16.0 * (double)c->ht_cap, (double)c->ht_cap);
    hipLaunchKernelGGL(k_ht_init, dim3(2048), dim3(256), 0, c->stream, c->ht, c->ht_cap);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

__global__ void k_ht_merge(HtEntry* t, u64 mask, const u64* k, const u32* r, const u32* f, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 key = k[i];
    u64 h = d_mm_hash64(key) & mask;
    while (true) {                                  // svt_count_merge sizes the table for the merged content first
        ull cur = t[h].key;
        if (cur == key) break;
        if (cur == SVT_EMPTY_KEY) {
            ull old = atomicCAS(&t[h].key, SVT_EMPTY_KEY, (ull)key);
            if (old == SVT_EMPTY_KEY || old == key) break;
        }
        h = (h + 1) & mask;
    }
    if (r[i] && f[i]) atomicAdd(reinterpret_cast<ull*>(&t[h].c[0]), (ull)r[i] | ((ull)f[i] << 32));
    else if (r[i]) atomicAdd(&t[h].c[0], r[i]);
    else if (f[i]) atomicAdd(&t[h].c[1], f[i]);
}
int launch_ht_merge(svt_ctx* c, const u64* d_k, const u32* d_r, const u32* d_f, u64 n) {
    if (n == 0) return SVT_OK;
    ProfScope ps(c, "k_ht_merge", 32.0 * (double)n, (double)n);
    hipLaunchKernelGGL(k_ht_merge, dim3((u32)((n + 255) / 256)), dim3(256), 0, c->stream, c->ht, c->ht_cap - 1, d_k, d_r, d_f, n);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

#define CMP_STAGE 256          // per-wave LDS staging entries (flushed with ONE global atomic when > 192 are pending)
__global__ void __launch_bounds__(256) k_ht_compact(const HtEntry* __restrict__ t, u64 cap, int mode, u64* __restrict__ ok, u32* __restrict__ orv,
                                                    u32* __restrict__ of, ull* __restrict__ counters) {
    // Streaming scan, 16 B per lane per step.  Kept entries (a few % of the slots) are staged per WAVE in LDS and appended
    // to the output in bulk, so the single append cursor sees ~1 atomic per 200 kept entries instead of 1 per wave step;
    // `distinct` is accumulated in registers (one atomic per wave at the end).
    __shared__ HtEntry stage[4][CMP_STAGE];
    const u32 lane = d_lane(), wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // explicitly wave-uniform
    HtEntry* st = stage[wave];
    const u64 i = ((u64)blockIdx.x * 4 + wave) * 64 + lane;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 n_present = 0, pending = 0;                 // pending is wave-uniform
    auto flush = [&]() {
        ull pos = 0;
        if (lane == 0) pos = atomicAdd(&counters[1], (ull)pending);
        pos = __shfl(pos, 0);
        if (ok) for (u32 x = lane; x < pending; x += 64) { HtEntry e = st[x]; ok[pos + x] = e.key; orv[pos + x] = e.c[0]; of[pos + x] = e.c[1]; }
        pending = 0;
    };
    for (u64 base = i - lane; base < cap; base += stride) {           // whole waves step together
        const u64 j = base + lane;
        HtEntry e; e.key = SVT_EMPTY_KEY; e.c[0] = e.c[1] = 0;
        if (j < cap) e = t[j];
        const bool present = e.key != SVT_EMPTY_KEY;
        bool keep;
        if (mode == 2) keep = present;
        else if (mode == 1) keep = present && e.c[0] > 2;                                        // seq_parse.rs:35-38
        else keep = present && e.c[0] > 0 && e.c[1] > 0 && (e.c[0] + e.c[1]) > 2;              // seq_parse.rs:41
        n_present += present;
        const ull mk = __ballot(keep);
        if (mk) {
            if (keep) st[pending + d_rank(mk)] = e;
            pending += __popcll(mk);
            if (pending > CMP_STAGE - 64) flush();
        }
    }
    if (pending) flush();
    #pragma unroll
    for (int s = 32; s >= 1; s >>= 1) n_present += __shfl_xor(n_present, s);
    if (lane == 0 && n_present) atomicAdd(&counters[0], (ull)n_present);
}
int launch_ht_compact(svt_ctx* c, int mode, u64* d_k, u32* d_r, u32* d_f, ull* d_counters) {
    ProfScope ps(c, "k_ht_compact", 16.0 * (double)c->ht_cap, (double)c->ht_cap);
    // a wave ends with one atomic on the append cursor: 64 slots per lane keep those to a few thousand for a table sized from the batch before (8 M slots at 100k reads)
    const u32 grid = (u32)std::min<u64>(2048, std::max<u64>(256, c->ht_cap / (256 * 64)));
    hipLaunchKernelGGL(k_ht_compact, dim3(grid), dim3(256), 0, c->stream, c->ht, c->ht_cap, mode, d_k, d_r, d_f, d_counters);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
