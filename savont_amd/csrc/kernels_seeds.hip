// kernels_seeds.hip -- K3 (open-syncmer minimizers + SNPmer scan + est_id + quality bins),
// K4 (LSH signatures + sorted distinct minimizer sets), K6-prep (SNPmer bitset rows).
//
// Reference semantics: seeding::get_twin_read_syncmer src/seeding.rs:317-658 (incl. the s-mer
// initialisation quirk :392-397), estimate_sequence_identity_vec :801-817, quality binning :578-602
// + src/types.rs:447-467, per-read high-frequency flags src/kmer_comp.rs:169-202,
// TwinRead::compute_lsh_signatures src/types.rs:719-747.
//
// Mapping (MI355X): one wavefront (= one 64-thread workgroup) per read.  Lane l owns k-mer end
// position base+l of each 64-position chunk, so packed words and quality bytes are read as
// coalesced lines; the 11-hash syncmer window is exchanged through a 74-entry LDS ring (one
// ds_write_b64 + eleven ds_read_b64 per lane per chunk, conflict-free: consecutive lanes read
// consecutive 8-byte slots); hits are compacted in read order with ballot + mbcnt into LDS and
// flushed once per read with coalesced stores.
#include "svt_internal.hpp"

__device__ __forceinline__ u32 base_at(const u32* w, int x) {     // 2-bit code of base x (x >= 0)
    return (w[x >> 4] >> (30 - 2 * (x & 15))) & 3u;
}

__device__ __forceinline__ bool snp_lookup(const SnpTable& t, u64 km, u32& val) {
    if (t.n_sites == 0) return false;
    u32 h = snp_slot_hash(km) & t.mask;
    while (true) {
        u64 key = t.keys[h];
        if (key == km) { val = t.vals[h]; return true; }
        if (key == SVT_EMPTY_KEY) return false;
        h = (h + 1) & t.mask;
    }
}
__device__ __forceinline__ bool hf_contains(const SnpTable& t, u64 km) {
    int lo = 0, hi = (int)t.n_hf - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        u64 v = t.hf[mid];
        if (v == km) return true;
        if (v < km) lo = mid + 1; else hi = mid - 1;
    }
    return false;
}

// Round 4: the kernel is bound by VALU issue, not by memory (ISA count: the run-time syncmer loop was 11 trips of ~50 instructions and the lane-ordered
// f64 sum of est_id 265 instructions per 64 bases -- together two thirds of a chunk).  So: the window size is a template parameter (the eleven
// ds_read_b64 are issued back to back and compared from registers), the reverse-complemented s-mer is the top of the reverse-complemented k-mer
// (no second bit reversal), a single-wave workgroup needs no s_barrier around its own LDS traffic (the LDS serves one wave's requests in order), the
// minimizers go straight to their per-read region in HBM, and est_id has its own kernel with one LANE per read (k_est_id below): the same
// sequential additions, 64 reads per wave instruction instead of one.
template <int WIN>      // syncmer window = k - s + 1 (= c); 0: run-time value
__global__ void __launch_bounds__(64) k_seeds(BatchView bv, SnpTable st, SeedsDev sd,
                                              u32 k, u32 cpar, u32 min_bq, int use_qual, u32 maxm, u32 maxs, u32 read_lo) {
    extern __shared__ __align__(16) unsigned char smem[];
    u64* H = (u64*)smem;                          // [80]: 16 history slots + 64 current
    u64* skm = H + 80;                            // [maxs]   bit63 = quality pass
    u32* spos = (u32*)(skm + maxs);               // [maxs]
    const u32 r = blockIdx.x + read_lo;          // the launch covers reads [read_lo, read_lo + gridDim.x): one rank's block under svt_set_shard
    if (r >= bv.n) return;
    const u32 lane = threadIdx.x;
    const u64 o = bv.off[r];
    const u32 len = (u32)(bv.off[r + 1] - o);
    const u64 mb = sd.mini_base[r];
    if (len < k) {                                                              // seeding.rs:339 -> None
        if (lane == 0) { sd.mini_cnt[r] = 0; sd.snp_cnt[r] = 0; sd.snp_base[r] = 0; sd.status[r] = 1; }
        return;
    }
    const u32* w = bv.packed + bv.woff[r];
    const u8 fl = bv.flags[r];
    const bool have_q = use_qual && bv.qual;
    const bool eq_q = have_q && (fl & 1);                                       // :372-380
    const u8* q = bv.qual + o;
    const u32 sl = k - cpar + 1;                                                // :363
    const u32 win = WIN ? (u32)WIN : k - sl + 1;                                // :368 (= cpar)
    const u32 midw = (k - sl) / 2;                                              // :528
    const u32 mid_k = k / 2;
    const u64 split_mask = ~(3ull << (k - 1));
    const u64 s_mask = ~0ull >> (64 - 2 * sl);
    const u32 npos = len - k + 1;
    u32 mcnt = 0, scnt = 0; bool overflow = false;
    for (u32 base = 0; base < npos; base += 64) {
        const u32 e = base + lane;               // k-mer start; end i = e + k - 1
        const bool valid = e < npos;
        const u32 p = valid ? e : 0;
        const u64 f = d_window64(w, p) >> (64 - 2 * k);
        const u64 rv = d_revcomp(f, k);
        const bool canon = (f & split_mask) < (rv & split_mask);               // :429 ties -> reverse
        const u64 km = canon ? f : rv;
        // canonical s-mer ending at i (last sl bases of the k-mer); its reverse complement is the FIRST sl bases of rv.  Quirk for the first sl-1 windows.
        u64 sf = f & s_mask, sr = rv >> (2 * (k - sl));
        if (base == 0 && e + 1 < sl) {                                          // i < k+sl-2: bases 0..sl-2 seeded the register (:392-397)
            sf = 0;
            const int i = (int)(e + k - 1);
            for (u32 j = 0; j < sl; j++) {
                int x = i - (int)sl + 1 + (int)j;
                if (x < (int)k - 1) x -= (int)(k - sl);
                sf = (sf << 2) | base_at(w, x);
            }
            sr = d_revcomp(sf, sl);
        }
        const u64 h = d_mm_hash64(sf < sr ? sf : sr);                           // :446-452
        __builtin_amdgcn_wave_barrier();                                        // one wave: its LDS requests are served in order, nothing to wait for
        H[16 + lane] = h;
        __builtin_amdgcn_wave_barrier();
        // syncmer test: window = hashes of ends i-(win-1) .. i ; H index of end (base+k-1+j) is 16+j
        bool sync = valid && (e + 1 >= win);                                    // window full (:527)
        if (WIN) {
            u64 hv[WIN ? WIN : 1];
            #pragma unroll
            for (int j = 0; j < WIN; j++) hv[j] = H[16 + lane - (WIN - 1) + j];
            constexpr int MIDW = WIN ? (WIN - 1) / 2 : 0;                            // = (k - s) / 2: win = k - s + 1
            const u64 mh = hv[MIDW];
            bool le = false;
            #pragma unroll
            for (int j = 0; j < WIN; j++) if (j != MIDW) le |= hv[j] <= mh;         // :533
            sync = sync && !le;
        } else if (sync) {
            const u64 mh = H[16 + lane - (win - 1) + midw];
            for (u32 j = 0; j < win; j++) {
                u64 hv = H[16 + lane - (win - 1) + j];
                if (j != midw && hv <= mh) sync = false;
            }
        }
        ull mm = __ballot(sync);
        if (sync) {                                                             // ~2 lanes in 64: straight into the read's region, flags bit0 = not high-frequency (kmer_comp.rs:179), bit1 = canon
            u32 d = mcnt + d_rank(mm);
            if (d < maxm) { sd.mini_pos[mb + d] = e; sd.mini_kmer[mb + d] = km; sd.mini_flags[mb + d] = (u8)((hf_contains(st, km) ? 0 : 1) | (canon ? 2 : 0)); }
        }
        mcnt += __popcll(mm);
        // SNPmer probe (:509-525)
        u32 val;
        bool hit = valid && snp_lookup(st, km, val);
        bool pass = true;
        if (hit && have_q && !eq_q) pass = ((u8)(q[e + mid_k] - 33)) > min_bq;  // strict (:517)
        if (hit && !have_q) pass = 60 > min_bq;                                 // :513-515
        ull ms = __ballot(hit);
        if (hit) { u32 d = scnt + d_rank(ms); if (d < maxs) { spos[d] = e; skm[d] = km | ((u64)pass << 63); } else overflow = true; }
        scnt += __popcll(ms);
        __builtin_amdgcn_wave_barrier();
        if (lane >= 48) H[lane - 48] = h;                                       // keep the last 16 hashes for the next chunk
    }
    __builtin_amdgcn_wave_barrier();
    overflow = __ballot(overflow) != 0;
    if (scnt > maxs) scnt = maxs;
    // ---- SNPmer dedup (:550-559): drop every split k-mer seen more than once (counted before the quality test)
    u32 fin = 0; ull base_out = 0;
    // pass 1: count survivors; pass 2: write.  survivors keep read order.
    for (int pass2 = 0; pass2 < 2; pass2++) {
        if (pass2) {
            if (lane == 0) base_out = atomicAdd(sd.snp_cursor, (ull)fin);
            base_out = __shfl(base_out, 0);
            if (base_out + fin > sd.snp_cap) { overflow = true; break; }
        }
        u32 run = 0;
        for (u32 b0 = 0; b0 < scnt; b0 += 64) {
            u32 i = b0 + lane; bool keep = false; u64 v = 0;
            if (i < scnt) {
                v = skm[i]; u64 sp = v & split_mask & ~(1ull << 63);
                u32 c = 0;
                for (u32 j = 0; j < scnt; j++) c += ((skm[j] & split_mask & ~(1ull << 63)) == sp);
                keep = (c == 1) && (v >> 63);
            }
            ull mk = __ballot(keep);
            if (pass2 && keep) {
                u64 d = base_out + run + d_rank(mk); u64 km = v & ~(1ull << 63);
                sd.snp_pos[d] = spos[i]; sd.snp_kmer[d] = km; sd.snp_flags[d] = hf_contains(st, km) ? 0 : 1;   // kmer_comp.rs:198
            }
            run += __popcll(mk);
        }
        if (!pass2) fin = run;
    }
    // ---- quality bins (seeding.rs:578-602): min of each 4 raw bytes -> 4-bit code, two per byte
    if (have_q && sd.qualbins) {
        const u64 qo = sd.qb_off[r];
        const u32 nb = (len + 3) / 4;
        for (u32 b0 = 0; b0 < nb; b0 += 64) {
            u32 bi = b0 + lane; u32 code = 0;
            if (bi < nb) {
                u32 mn = 255;
                for (u32 j = 0; j < 4; j++) { u32 x = bi * 4 + j; if (x < len) mn = min(mn, (u32)q[x]); }
                code = d_qual_bin((u8)mn);
            }
            u32 nxt = __shfl_down(code, 1);
            if (!(lane & 1) && bi < nb) sd.qualbins[qo + (bi >> 1)] = (u8)(code | ((bi + 1 < nb ? nxt : 0) << 4));
        }
    }
    if (lane == 0) {
        sd.mini_cnt[r] = mcnt < maxm ? mcnt : maxm;
        sd.snp_cnt[r] = overflow ? 0 : fin; sd.snp_base[r] = base_out;
        sd.status[r] = overflow ? 2 : 0;
    }
}

// Round 6: the rank-table form of K3 (s = k - c + 1 <= 7, which holds the defaults k = 17, c = 11).
//   * mm_hash64 of the canonical s-mer is only ever COMPARED (`<` / `<=`, src/seeding.rs:527-537), and it is a bijection of its 64-bit argument: the rank of the hash among the 4^s
//     possible inputs decides every comparison identically.  `rank[forward s-mer]` (u16, built on the host once per s: hash of min(s-mer, its reverse complement), ranked) sits in
//     LDS: one ds_read_u16 replaces the reverse complement of the s-mer, the minimum, seven rounds of 64-bit shift / add / xor and -- in the window test -- eleven 64-bit compares
//     (now five v_min3_u32 and one compare on 32-bit ring entries).  The s-mer initialisation quirk (:392-397) only changes WHICH 2s bits are looked up.
//   * the SNPmer probe of every position (:509-525) starts with one bit of an LDS bitmap: the occupancy of the probe's first slot in the open-addressing table (the table is
//     filled to an eighth): a clear bit is a miss, and only the lanes with a set bit go to the table in HBM / L2 (before: one scattered 8-byte load per base).
//   * DEDUP_SNPMERS (:550-559) through a per-wave LDS hash table on the split k-mer (an atomicCAS per hit) instead of comparing every hit with every other, twice.
//   * the high-frequency flag of the minimizers (kmer_comp.rs:179, a binary search in HBM per accepted window: a chain of dependent loads inside the position loop) moved to
//     K4 (k_lsh_sets), which reads every minimizer anyway.
// Workgroups are persistent (the tables are loaded once per workgroup) and sixteen waves wide; a wave draws reads from a counter.
#define SEEDS_CHUNK 512u      // entries of the SNPmer arrays a wave of k_seeds_rt takes from the cursor at a time (what a wave leaves unused at the end is never read: the lists are addressed through snp_base / snp_cnt)
__device__ __forceinline__ u64 d_window64_regs(u32 w0, u32 w1, u32 w2, u32 p) {     // d_window64 on the three words already loaded (words p/16 .. p/16 + 2)
    const u32 o = (p & 15) * 2;
    const u64 A = ((u64)w0 << 32) | w1;
    return o == 0 ? A : ((A << o) | ((u64)w2 >> (32 - o)));
}
__device__ __forceinline__ u32 umin3(u32 a, u32 b, u32 c) { u32 r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int WIN>      // syncmer window = k - s + 1 (= c); 0: run-time value
__global__ void __launch_bounds__(1024) k_seeds_rt(BatchView bv, SnpTable st, SeedsDev sd, const u16* __restrict__ g_rank, const u32* __restrict__ g_occ, u32 occ_bits_mask,
                                                   u32 k, u32 cpar, u32 min_bq, int use_qual, u32 maxm, u32 maxs, u32 read_lo, u32 read_hi) {
    extern __shared__ __align__(16) unsigned char smem[];
    const u32 sl = k - cpar + 1;                                                // :363
    const u32 rt_n = 1u << (2 * sl);
    u16* RT = (u16*)smem;
    u32* BM = (u32*)(smem + ((rt_n * 2 + 15) & ~15u));
    const u32 occ_words = (occ_bits_mask >> 5) + 1;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 htn = 2 * maxs;                                                    // slots of the dedup table (a power of two: maxs is)
    unsigned char* wbase = (unsigned char*)(BM + occ_words) + (size_t)wave * (80 * 4 + (size_t)maxs * 12 + (size_t)htn * 4);
    u32* H = (u32*)wbase;                         // [80]: 16 history slots + 64 current (ranks)
    u64* skm = (u64*)(H + 80);                    // [maxs]   bit63 = quality pass
    u32* spos = (u32*)(skm + maxs);               // [maxs]   bit31 = split k-mer seen more than once
    u32* HT = spos + maxs;                        // [2 maxs] 0 = empty, else hit index + 1
    for (u32 i = threadIdx.x; i < rt_n / 2; i += blockDim.x) ((u32*)RT)[i] = ((const u32*)g_rank)[i];
    for (u32 i = threadIdx.x; i < occ_words; i += blockDim.x) BM[i] = g_occ[i];
    __syncthreads();
    const u32 win = WIN ? (u32)WIN : cpar;                                      // :368
    const u32 midw = (k - sl) / 2;                                              // :528
    const u32 mid_k = k / 2;
    const u64 split_mask = ~(3ull << (k - 1));
    const u32 s_mask = rt_n - 1;
    const bool have_q = use_qual && bv.qual;
    // a wave's reads: read_lo + its index among the grid's waves, then in steps of the grid's waves (reads of one sample are alike in length: no counter to draw from --
    // 10^5 returning atomics on ONE address cost ~20 ns each, which was the whole run time of the round-5 kernel: its SNPmer cursor took one per read)
    const u32 n_waves = gridDim.x * (blockDim.x >> 6);
    ull chunk_base = 0; u32 chunk_left = 0;                                     // this wave's piece of the SNPmer arrays: one atomic on the cursor per SEEDS_CHUNK entries, not per read
    for (u32 r = read_lo + blockIdx.x * (blockDim.x >> 6) + wave; r < read_hi && r < bv.n; r += n_waves) {
        const u64 o = bv.off[r];
        const u32 len = (u32)(bv.off[r + 1] - o);
        const u64 mb = sd.mini_base[r];
        if (len < k) {                                                          // seeding.rs:339 -> None
            if (lane == 0) { sd.mini_cnt[r] = 0; sd.snp_cnt[r] = 0; sd.snp_base[r] = 0; sd.status[r] = 1; }
            continue;
        }
        const u32* w = bv.packed + bv.woff[r];
        const bool eq_q = have_q && (bv.flags[r] & 1);                          // :372-380
        const u8* q = bv.qual + o;
        const u32 npos = len - k + 1;
        u32 mcnt = 0, scnt = 0; bool overflow = false;
        u32 cq_n = 0;                                                              // candidates waiting for the table: < 128, in read order (wave-uniform count)
        u32* const cq_e = HT; u64* const cq_km = (u64*)(HT + 128);                 // 128 x (position, canonical k-mer) in the room of the dedup table, which is not in use yet (2 maxs >= 512 words)
        auto flush_candidates = [&](const u32 cnt) {                               // the first cnt (<= 64) candidates: table probe, quality test, hits appended in order
            __builtin_amdgcn_wave_barrier();
            const bool mine = lane < cnt;
            const u32 ce = mine ? cq_e[lane] : 0; const u64 ckm = mine ? cq_km[lane] : 0;
            const u32 rest = cq_n - cnt;                                           // < 64: they move to the front
            u32 me = 0; u64 mk2 = 0;
            if (lane < rest) { me = cq_e[cnt + lane]; mk2 = cq_km[cnt + lane]; }
            bool hit = false;
            u32 hs = snp_slot_hash(ckm) & st.mask;
            bool go = mine;
            while (go) {
                const u64 key = st.keys[hs];
                if (key == ckm) { hit = true; go = false; }
                else if (key == SVT_EMPTY_KEY) go = false;
                else hs = (hs + 1) & st.mask;
            }
            bool pass = true;
            if (hit && have_q && !eq_q) pass = ((u8)(q[ce + mid_k] - 33)) > min_bq;  // strict (:517)
            if (hit && !have_q) pass = 60 > min_bq;                                 // :513-515
            const ull ms = __ballot(hit);
            if (hit) { const u32 d = scnt + d_rank(ms); if (d < maxs) { spos[d] = ce; skm[d] = ckm | ((u64)pass << 63); } else overflow = true; }
            scnt += __popcll(ms);
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) { cq_e[lane] = me; cq_km[lane] = mk2; }
            cq_n = rest;
            __builtin_amdgcn_wave_barrier();
        };
        u32 nw0, nw1, nw2;
        { const u32 p0 = lane < npos ? lane : 0; nw0 = w[p0 >> 4]; nw1 = w[(p0 >> 4) + 1]; nw2 = w[(p0 >> 4) + 2]; }
        for (u32 base = 0; base < npos; base += 64) {
            const u32 e = base + lane;               // k-mer start; end i = e + k - 1
            const bool valid = e < npos;
            const u32 p = valid ? e : 0;
            const u64 f = d_window64_regs(nw0, nw1, nw2, p) >> (64 - 2 * k);
            {   // the words of the NEXT chunk's windows are asked for now: their latency passes behind this chunk's work
                const u32 pn = (e + 64 < npos) ? e + 64 : 0;
                const u32 an = pn >> 4;
                nw0 = w[an]; nw1 = w[an + 1]; nw2 = w[an + 2];
            }
            const u64 rv = d_revcomp(f, k);
            const bool canon = (f & split_mask) < (rv & split_mask);           // :429 ties -> reverse
            const u64 km = canon ? f : rv;
            // the s-mer ending at i = the last sl bases of the k-mer; for the first sl - 1 windows the register was seeded by bases 0 .. sl-2 (:392-397)
            u32 sf = (u32)f & s_mask;
            if (base == 0 && e + 1 < sl) {
                sf = 0;
                const int i = (int)(e + k - 1);
                for (u32 j = 0; j < sl; j++) {
                    int x = i - (int)sl + 1 + (int)j;
                    if (x < (int)k - 1) x -= (int)(k - sl);
                    sf = (sf << 2) | base_at(w, x);
                }
            }
            const u32 h = RT[sf];                                               // rank of mm_hash64(canonical s-mer): decides :446-452 / :527-537 like the hash itself
            __builtin_amdgcn_wave_barrier();                                    // one wave: its LDS requests are served in order, nothing to wait for
            H[16 + lane] = h;
            __builtin_amdgcn_wave_barrier();
            bool sync = valid && (e + 1 >= win);                                // window full (:527)
            if (WIN) {
                u32 hv[WIN ? WIN : 1];
                #pragma unroll
                for (int j = 0; j < WIN; j++) hv[j] = H[16 + lane - (WIN - 1) + j];
                constexpr int MIDW = WIN ? (WIN - 1) / 2 : 0;                   // = (k - s) / 2: win = k - s + 1
                u32 others = 0xFFFFFFFFu;                                       // minimum over the window without its middle
                if (WIN == 11) {
                    others = umin3(hv[0], hv[1], hv[2]); others = umin3(others, hv[3], hv[4]); others = umin3(others, hv[6], hv[7]); others = umin3(others, hv[8], hv[9]); others = min(others, hv[10]);
                } else {
                    #pragma unroll
                    for (int j = 0; j < WIN; j++) if (j != MIDW) others = min(others, hv[j]);
                }
                sync = sync && hv[MIDW] < others;                               // :533 (any other <= the middle rejects)
            } else if (sync) {
                const u32 mh = H[16 + lane - (win - 1) + midw];
                for (u32 j = 0; j < win; j++) {
                    const u32 hv = H[16 + lane - (win - 1) + j];
                    if (j != midw && hv <= mh) sync = false;
                }
            }
            const ull mm = __ballot(sync);
            if (sync) {                                                         // ~2 lanes in 64: straight into the read's region; flags bit1 = canon, bit0 (not high-frequency) is set by K4
                const u32 d = mcnt + d_rank(mm);
                if (d < maxm) { sd.mini_pos[mb + d] = e; sd.mini_kmer[mb + d] = km; sd.mini_flags[mb + d] = (u8)(canon ? 2 : 0); }
            }
            mcnt += __popcll(mm);
            // SNPmer probe (:509-525).  The occupancy bit of the probe's first slot decides most positions inside the CU; the others queue up in LDS and go to the table 64 at a
            // time (flush_candidates): one round of dependent loads per 64 candidates instead of one per chunk of 64 positions -- the wave waits for the table ~6 times per read
            if (st.n_sites) {
                const u32 hs = snp_slot_hash(km) & st.mask;
                const u32 ob = hs & occ_bits_mask;
                const bool cand = valid && ((BM[ob >> 5] >> (ob & 31)) & 1u);
                const ull mc = __ballot(cand);
                if (cand) { const u32 d = cq_n + d_rank(mc); cq_e[d] = e; cq_km[d] = km; }
                cq_n += __popcll(mc);
                if (cq_n >= 64) flush_candidates(64u);
            }
            __builtin_amdgcn_wave_barrier();
            if (lane >= 48) H[lane - 48] = h;                                       // keep the last 16 ranks for the next chunk
        }
        if (cq_n) flush_candidates(cq_n);
        __builtin_amdgcn_wave_barrier();
        overflow = __ballot(overflow) != 0;
        if (scnt > maxs) scnt = maxs;
        for (u32 i = lane; i < htn; i += 64) HT[i] = 0;                           // the queue's room becomes the dedup table
        __builtin_amdgcn_wave_barrier();
        // ---- SNPmer dedup (:550-559): drop every split k-mer seen more than once (counted before the quality test).  Every hit goes into the wave's table by its split k-mer;
        // a hit that finds its split k-mer already there flags both
        const u64 sp_mask = split_mask & ~(1ull << 63);
        for (u32 b0 = 0; b0 < scnt; b0 += 64) {
            const u32 i = b0 + lane;
            if (i < scnt) {
                const u64 sp = skm[i] & sp_mask;
                u32 slot = snp_slot_hash(sp) & (htn - 1);
                for (;;) {
                    const u32 old = atomicCAS(&HT[slot], 0u, i + 1);
                    if (old == 0) break;
                    if ((skm[old - 1] & sp_mask) == sp) { atomicOr(&spos[old - 1], 0x80000000u); atomicOr(&spos[i], 0x80000000u); break; }
                    slot = (slot + 1) & (htn - 1);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        u32 fin = 0;
        for (u32 b0 = 0; b0 < scnt; b0 += 64) { const u32 i = b0 + lane; fin += __popcll(__ballot(i < scnt && !(spos[i] >> 31) && (skm[i] >> 63))); }
        if (fin > chunk_left) {                                                  // wave-uniform
            const u32 take = fin > SEEDS_CHUNK ? fin : SEEDS_CHUNK;
            ull cb = 0;
            if (lane == 0) cb = atomicAdd(sd.snp_cursor, (ull)take);
            chunk_base = __shfl(cb, 0); chunk_left = take;
        }
        const ull base_out = chunk_base;
        chunk_base += fin; chunk_left -= fin;
        if (base_out + fin > sd.snp_cap) overflow = true;
        else {
            u32 run = 0;
            for (u32 b0 = 0; b0 < scnt; b0 += 64) {                                // survivors keep read order
                const u32 i = b0 + lane;
                const bool keep = i < scnt && !(spos[i] >> 31) && (skm[i] >> 63);
                const ull mk = __ballot(keep);
                if (keep) {
                    const u64 d = base_out + run + d_rank(mk); const u64 km = skm[i] & ~(1ull << 63);
                    sd.snp_pos[d] = spos[i]; sd.snp_kmer[d] = km; sd.snp_flags[d] = hf_contains(st, km) ? 0 : 1;   // kmer_comp.rs:198
                }
                run += __popcll(mk);
            }
        }
        // ---- quality bins (seeding.rs:578-602): min of each 4 raw bytes -> 4-bit code, two per byte
        if (have_q && sd.qualbins) {
            const u64 qo = sd.qb_off[r];
            const u32 nb = (len + 3) / 4;
            for (u32 b0 = 0; b0 < nb; b0 += 64) {
                const u32 bi = b0 + lane; u32 code = 0;
                if (bi < nb) {
                    u32 mn = 255;
                    for (u32 j = 0; j < 4; j++) { const u32 x = bi * 4 + j; if (x < len) mn = min(mn, (u32)q[x]); }
                    code = d_qual_bin((u8)mn);
                }
                const u32 nxt = __shfl_down(code, 1);
                if (!(lane & 1) && bi < nb) sd.qualbins[qo + (bi >> 1)] = (u8)(code | ((bi + 1 < nb ? nxt : 0) << 4));
            }
        }
        if (lane == 0) {
            sd.mini_cnt[r] = mcnt < maxm ? mcnt : maxm;
            sd.snp_cnt[r] = overflow ? 0 : fin; sd.snp_base[r] = base_out;
            sd.status[r] = overflow ? 2 : 0;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// est_id (seeding.rs:801-817): 100 - 100 * mean(10^(-q/10)), the mean as ONE sequential f64 sum in read order (the reference folds an iterator), so the
// additions cannot be reassociated.  One LANE per read: every lane walks its own read and adds table[q - 33] base by base -- the reference's order
// exactly -- while a wave instruction serves 64 reads (the round-3 kernel spent two v_readlane + one v_add_f64 of a whole wave on every base of ONE read).
// The 256-entry table sits in LDS; quality bytes are fetched 16 at a time (unaligned dwordx4), the next piece while the current one is summed.
__global__ void __launch_bounds__(256) k_est_id(BatchView bv, SeedsDev sd, const double* __restrict__ ptable, u32 k, int use_qual, u32 read_lo, u32 read_hi) {
    __shared__ double tab[256];
    tab[threadIdx.x] = ptable[threadIdx.x];
    __syncthreads();
    const u32 r = read_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= read_hi || r >= bv.n) return;
    const u64 o = bv.off[r];
    const u32 len = (u32)(bv.off[r + 1] - o);
    const bool have_q = use_qual && bv.qual;
    const bool ok = have_q && !(bv.flags[r] & 1) && len >= k;                   // all-equal qualities -> None (:571-576); len < k -> no twin read at all
    double est = 0.0;
    if (ok) {
        const u8* q = bv.qual + o;
        double sum = 0.0;
        u32 i = 0;
        uint4 nx = make_uint4(0, 0, 0, 0);
        if (len >= 16) __builtin_memcpy(&nx, q, 16);
        for (; i + 16 <= len; i += 16) {
            const u32 d[4] = {nx.x, nx.y, nx.z, nx.w};
            __builtin_memcpy(&nx, q + i + 16, 16);                                  // the piece after this one, asked for before this one is summed (the quality array has 64 bytes of slack behind its last read)
            double t[16];
            #pragma unroll
            for (u32 j = 0; j < 16; j++) t[j] = tab[(u8)((d[j >> 2] >> (8 * (j & 3))) - 33)];
            #pragma unroll
            for (u32 j = 0; j < 16; j++) sum += t[j];
        }
        for (; i < len; i++) sum += tab[(u8)(q[i] - 33)];
        est = 100.0 - (sum / (double)len * 100.0);
    }
    sd.est_id[r] = est; sd.est_valid[r] = ok ? 1 : 0;
}

int launch_seeds(svt_ctx* c, svt_batch* b, u32 k, u32 cpar, u8 min_bq, int use_qual, u32 maxm, u32 maxs, u32 read_lo, u32 read_hi) {
    if (read_hi <= read_lo) return SVT_OK;
    size_t sh = 80 * 8 + (size_t)maxs * 12;
    // algorithmic bytes per read: packed + quals in, 10 B per minimizer/SNPmer + 8 + L/8 out
    double bytes = (double)b->total_words * 4.0 + (use_qual && b->has_qual ? (double)b->total_bases * 1.125 : 0.0) + (double)b->total_bases / 11.0 * 13.0 + 32.0 * b->n;
    const double part = (double)(read_hi - read_lo) / (double)b->n;
    {
        ProfScope ps(c, "k_seeds", bytes * part, read_hi - read_lo);
        const u32 win = cpar;                                                     // k - (k - c + 1) + 1
        const u32 sl = k - cpar + 1;
        bool rt = sl <= 7 && c->d_rank && c->rank_s == sl && c->snp_occ && !c->opt().seeds_hash;
        if (rt) {
            // persistent workgroups of W waves sharing the rank table and the occupancy bitmap; per wave: ring + hit list + dedup table
            const size_t shared_b = (((size_t)2 << (2 * sl)) + 15 & ~(size_t)15) + (size_t)((c->snp_occ_mask >> 5) + 1) * 4, per_wave = 80 * 4 + (size_t)maxs * 12 + (size_t)maxs * 8;
            u32 waves = 16;
            while (waves > 1 && shared_b + waves * per_wave > (size_t)150 * 1024) waves >>= 1;
            const size_t lds = shared_b + waves * per_wave;
            if (lds > (size_t)160 * 1024) rt = false;
            else {
                static int cus = 0;
                if (!cus) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, c->device) == hipSuccess) cus = pr.multiProcessorCount; if (cus <= 0) cus = 256; }
                const u32 per_cu = (u32)std::max<size_t>(1, std::min<size_t>((size_t)160 * 1024 / lds, 32 / waves));
                const u32 nr = read_hi - read_lo;
                const u32 grid = std::max<u32>(1, std::min<u32>((nr + waves - 1) / waves, (u32)cus * per_cu));
                #define SEEDS_RT(W) do { DYN_LDS_ONCE(c, 6 + (W == 11 ? 0 : W == 9 ? 1 : W == 13 ? 2 : 3), k_seeds_rt<W>, 160 * 1024); \
                    hipLaunchKernelGGL(k_seeds_rt<W>, dim3(grid), dim3(waves * 64), lds, c->stream, b->view(), c->snp_table(), b->seeds, c->d_rank, c->snp_occ, c->snp_occ_mask, k, cpar, (u32)min_bq, use_qual, maxm, maxs, read_lo, read_hi); } while (0)
                if (win == 11) SEEDS_RT(11); else if (win == 9) SEEDS_RT(9); else if (win == 13) SEEDS_RT(13); else SEEDS_RT(0);
                #undef SEEDS_RT
            }
        }
        if (!rt) {
            #define SEEDS_LAUNCH(W) hipLaunchKernelGGL(k_seeds<W>, dim3(read_hi - read_lo), dim3(64), sh, c->stream, b->view(), c->snp_table(), b->seeds, k, cpar, (u32)min_bq, use_qual, maxm, maxs, read_lo)
            if (win == 11) SEEDS_LAUNCH(11); else if (win == 9) SEEDS_LAUNCH(9); else if (win == 13) SEEDS_LAUNCH(13); else SEEDS_LAUNCH(0);
            #undef SEEDS_LAUNCH
        }
        HIPCHK(c, hipGetLastError());
    }
    {
        ProfScope ps(c, "k_est_id", (use_qual && b->has_qual ? (double)b->total_bases : 0.0) * part + 9.0 * (read_hi - read_lo), read_hi - read_lo);
        hipLaunchKernelGGL(k_est_id, dim3((read_hi - read_lo + 255) / 256), dim3(256), 0, c->stream, b->view(), b->seeds, c->d_ptable, k, use_qual, read_lo, read_hi);
        HIPCHK(c, hipGetLastError());
    }
    return SVT_OK;
}

// ------------------------------------------------------------------------------------------------
// K4: LSH signatures (types.rs:719-747) + sorted distinct minimizer set (for K5/K7)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
    #pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        u64 o = __shfl_xor(v, s);
        v = o < v ? o : v;
    }
    return v;
}
// minimum of a u32 over the wave, in every lane (uniform): four row shifts, two row broadcasts -- each a v_min_u32 with a DPP operand, no LDS crossbar --
// then the value of lane 63.  Lanes without a source keep their own value (min(v, v)).
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xF, 0xF, false));   // row_shr:1
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xF, 0xF, false));   // row_shr:2
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xF, 0xF, false));   // row_shr:4
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xF, 0xF, false));   // row_shr:8
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast:15 into rows 1 and 3
    v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast:31 into rows 2 and 3
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

// Round 4 (the kernel is VALU-bound: ~10k wave instructions per read): the wave-wide minimum of the 64-bit hashes is found on their HIGH words with
// a DPP reduction; the low words only decide among lanes that tie there (two of a read's ~135 hashes share 32 high bits with probability 2e-6, the
// general path stays for them); lanes hash only the element slots the read fills (m ~ 135 of a 512-slot capacity); the bitonic sort runs over the
// read's own power of two instead of the launch's.
template <int EPL>   // elements per lane
__global__ void __launch_bounds__(64) k_lsh_sets(SeedsDev sd, SnpTable st, u32 n, u32 np2_cap, u32 read_lo) {
    extern __shared__ __align__(16) unsigned char smem[];
    u64* keys = (u64*)smem;                       // [np2_cap] bitonic sort buffer
    const u32 r = blockIdx.x + read_lo;          // the launch covers reads [read_lo, read_lo + gridDim.x): one rank's block under svt_set_shard
    if (r >= n) return;
    const u32 lane = threadIdx.x;
    const u32 m = sd.mini_cnt[r];
    const u64 mb = sd.mini_base[r];
    // ---- LSH: bottom-3 by FxHash64(table, kmer), duplicates kept (stable order irrelevant: equal hash <=> equal k-mer)
    // Round 6: lane = (table, third of the read's minimizers).  Until then a lane hashed ITS three minimizers for every table and each table's three smallest came out of
    // three wave-wide minimum searches (7 DPP steps, ballots, a cross-lane fetch): 20 x 3 of them were half of the kernel's ~4300 vector instructions per read.  Now lane
    // 20 s + t walks the minimizers of third s for table t alone (the k-mers sit in LDS; the twenty lanes of a third read the same address), keeps its own sorted three, and
    // lane t merges the triples of lanes t + 20 and t + 40 into its own: no wave-wide search at all.
    static_assert(SVT_LSH_TABLES * 3 <= 64 && SVT_LSH_BUCKET == 3, "lane = (table, third)");
    const u32 ml = min(m, np2_cap);               // (m <= np2_cap: the launch sizes the buffer from the minimizer capacity)
    for (u32 i = lane; i < ml; i += 64) keys[i] = sd.mini_kmer[mb + i];
    __builtin_amdgcn_wave_barrier();              // one wave per workgroup: its LDS requests are served in order
    if (m >= SVT_LSH_BUCKET) {
        const u32 t = lane % SVT_LSH_TABLES, third = lane / SVT_LSH_TABLES;
        const u32 per = (ml + 2) / 3, i0 = third * per, i1 = third < 3 ? min(ml, i0 + per) : 0;
        const u64 seedh = d_fx_word(0, (u64)t);
        u64 h0 = ~0ull, h1 = ~0ull, h2 = ~0ull, k0 = 0, k1 = 0, k2 = 0;
        // (h, kv) into the sorted three, by selects: an equal hash goes behind the one that is there (duplicates are kept)
        #define PUT3(h_, kv_) do { const u64 ph_ = (h_), pk_ = (kv_); const bool l0_ = ph_ < h0, l1_ = ph_ < h1, l2_ = ph_ < h2;                 \
            h2 = l1_ ? h1 : l2_ ? ph_ : h2; k2 = l1_ ? k1 : l2_ ? pk_ : k2; h1 = l0_ ? h0 : l1_ ? ph_ : h1; k1 = l0_ ? k0 : l1_ ? pk_ : k1;            \
            h0 = l0_ ? ph_ : h0; k0 = l0_ ? pk_ : k0; } while (0)
        for (u32 j = 0; j < per; j++) {
            const u32 i = i0 + j;
            if (i < i1) { const u64 kv = keys[i]; PUT3(d_fx_word(seedh, kv), kv); }
        }
        // NOTE: a real hash can equal ~0 only with probability 2^-64 per element; such an element would be
        // treated as "absent" here.  Documented in DESIGN.md (cannot be produced by 34..46-bit k-mers in practice).
        #pragma unroll
        for (u32 o = 1; o <= 2; o++) {
            const int src = (int)(lane + o * SVT_LSH_TABLES) & 63;
            const u64 a0 = __shfl(h0, src), a1 = __shfl(h1, src), a2 = __shfl(h2, src), b0 = __shfl(k0, src), b1 = __shfl(k1, src), b2 = __shfl(k2, src);
            if (lane < SVT_LSH_TABLES) { PUT3(a0, b0); PUT3(a1, b1); PUT3(a2, b2); }        // (an absent entry, hash ~0, is smaller than nothing)

        }
        #undef PUT3
        if (lane < SVT_LSH_TABLES) sd.lsh[(u64)r * SVT_LSH_TABLES + lane] = k0 ^ (k1 * 2ull) ^ (k2 * 3ull);
        if (lane == 0) sd.lsh_valid[r] = 1;
    } else {
        if (lane < SVT_LSH_TABLES) sd.lsh[(u64)r * SVT_LSH_TABLES + lane] = 0;
        if (lane == 0) sd.lsh_valid[r] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- sorted distinct set: key = kmer<<18 | index<<2 | solid<<1 | canon ; bitonic sort in LDS over the read's own power of two
    u32 np2 = 64; while (np2 < m) np2 <<= 1;
    if (np2 > np2_cap) np2 = np2_cap;
    for (u32 i = lane; i < np2; i += 64) {
        u64 key = ~0ull;
        if (i < m) {
            // bit0 of the flags (kmer_comp.rs:179: the k-mer is not a high-frequency one) is decided here, once per minimizer of the read, and written back for svt_seeds_fetch
            const u64 kmv = sd.mini_kmer[mb + i];
            const u32 f = (sd.mini_flags[mb + i] & 2u) | (hf_contains(st, kmv) ? 0u : 1u);
            sd.mini_flags[mb + i] = (u8)f;
            key = (kmv << 18) | ((u64)i << 2) | ((u64)(f & 1) << 1) | ((f >> 1) & 1);
        }
        keys[i] = key;
    }
    __builtin_amdgcn_wave_barrier();              // one wave per workgroup: its LDS requests are served in order
    for (u32 sz = 2; sz <= np2; sz <<= 1) {
        for (u32 st = sz >> 1; st > 0; st >>= 1) {
            for (u32 t = lane; t < (np2 >> 1); t += 64) {
                u32 i = 2 * t - (t & (st - 1));       // lower index of the pair
                u32 j = i + st;
                bool up = ((i & sz) == 0);
                u64 a = keys[i], b = keys[j];
                if ((a > b) == up) { keys[i] = b; keys[j] = a; }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    u32 out = 0, solid = 0;
    for (u32 b0 = 0; b0 < np2; b0 += 64) {
        u32 i = b0 + lane;
        u64 key = keys[i];
        bool real = key != ~0ull;
        bool first = real && (i == 0 || (keys[i - 1] >> 18) != (key >> 18));
        ull mk = __ballot(first);
        if (first) sd.set_kmer[mb + out + d_rank(mk)] = (key >> 18) | ((key & 1) << 63);
        out += __popcll(mk);
        // kmer_comp.rs:163-183: keep index iff in-read multiplicity <= MAX_KMER_COUNT_IN_READ (500) and not high-frequency
        bool sol = real && ((key >> 1) & 1);
        if (sol && m > 500) {
            u32 run = 1;
            for (int j = (int)i - 1; j >= 0 && (keys[j] >> 18) == (key >> 18) && run <= 500; j--) run++;
            for (u32 j = i + 1; j < np2 && (keys[j] >> 18) == (key >> 18) && run <= 500; j++) run++;
            sol = run <= 500;
        }
        solid += __popcll(__ballot(sol));
    }
    if (lane == 0) { sd.set_cnt[r] = out; sd.n_solid[r] = solid; }
}

int launch_lsh_sets(svt_ctx* c, svt_batch* b, u32 np2, u32 read_lo, u32 read_hi) {
    if (read_hi <= read_lo) return SVT_OK;
    const u32 nr = read_hi - read_lo;
    double bytes = (double)b->seeds.mini_cap * 0 + (double)b->n * (160.0 + 6.0 * 135.0);   // ~1 KB/read
    ProfScope ps(c, "k_lsh_sets", bytes * nr / b->n, nr);
    size_t sh = (size_t)np2 * 8;
    u32 epl = (np2 + 63) / 64;
    if (epl <= 4) hipLaunchKernelGGL(k_lsh_sets<4>, dim3(nr), dim3(64), sh, c->stream, b->seeds, c->snp_table(), b->n, np2, read_lo);
    else if (epl <= 8) hipLaunchKernelGGL(k_lsh_sets<8>, dim3(nr), dim3(64), sh, c->stream, b->seeds, c->snp_table(), b->n, np2, read_lo);
    else if (epl <= 16) hipLaunchKernelGGL(k_lsh_sets<16>, dim3(nr), dim3(64), sh, c->stream, b->seeds, c->snp_table(), b->n, np2, read_lo);
    else return svt_fail(c, SVT_ERR_ARG, "reads longer than 6144 bases are not supported by k_lsh_sets");
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// ------------------------------------------------------------------------------------------------
// K6-prep: SNPmer bitset rows.  Exact re-expression of asv_cluster.rs:356-383: DEDUP_SNPMERS
// (seeding.rs:550-559) leaves <= 1 SNPmer per site per read and the SNPmer set holds exactly two
// alleles per site (kmer_comp.rs:71-78), so (presence bit, allele bit) per site is lossless.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_snp_bits(SeedsDev sd, SnpTable st, u32 n, u32 read_lo) {
    extern __shared__ __align__(16) unsigned char smem[];
    u32* pa = (u32*)smem;                 // [2*words] presence all
    u32* pf = pa + 2 * st.words;          // presence filtered
    u32* al = pf + 2 * st.words;          // allele
    const u32 r = blockIdx.x + read_lo;          // the launch covers reads [read_lo, read_lo + gridDim.x): one rank's block under svt_set_shard
    if (r >= n) return;
    const u32 lane = threadIdx.x;
    for (u32 i = lane; i < 6 * st.words; i += 64) pa[i] = 0;
    __syncthreads();
    const u32 cnt = sd.snp_cnt[r]; const u64 sb = sd.snp_base[r];
    for (u32 i = lane; i < cnt; i += 64) {
        u64 km = sd.snp_kmer[sb + i]; u32 val;
        if (snp_lookup(st, km, val)) {
            u32 site = val >> 1, w = site >> 5, bit = 1u << (site & 31);
            atomicOr(&pa[w], bit);
            if (sd.snp_flags[sb + i] & 1) atomicOr(&pf[w], bit);
            if (val & 1) atomicOr(&al[w], bit);
        }
    }
    __syncthreads();
    u32* gpa = (u32*)(sd.p_all + (u64)r * st.words); u32* gpf = (u32*)(sd.p_filt + (u64)r * st.words); u32* gal = (u32*)(sd.allele + (u64)r * st.words);
    for (u32 i = lane; i < 2 * st.words; i += 64) { gpa[i] = pa[i]; gpf[i] = pf[i]; gal[i] = al[i]; }
    // sparse form: non-zero 64-bit presence words in ascending word order (<= cnt of them)
    const u64* pa64 = (const u64*)pa; const u64* pf64 = (const u64*)pf; const u64* al64 = (const u64*)al;
    u32 out = 0;
    for (u32 b0 = 0; b0 < st.words; b0 += 64) {
        u32 w = b0 + lane;
        u64 v = w < st.words ? pa64[w] : 0;
        ull mk = __ballot(v != 0);
        if (v != 0) { u64 d = sb + out + d_rank(mk); sd.nz_idx[d] = w; sd.nz_pa[d] = v; sd.nz_pf[d] = pf64[w]; sd.nz_a[d] = al64[w]; }
        out += __popcll(mk);
    }
    if (lane == 0) sd.nz_cnt[r] = out;
}

int launch_snp_bits(svt_ctx* c, svt_batch* b, u32 read_lo, u32 read_hi) {
    if (read_hi <= read_lo || c->words == 0) return SVT_OK;
    const u32 nr = read_hi - read_lo;
    ProfScope ps(c, "k_snp_bits", (double)nr * (24.0 * c->words + 9.0 * 50.0), nr);
    size_t sh = (size_t)c->words * 24;
    hipLaunchKernelGGL(k_snp_bits, dim3(nr), dim3(64), sh, c->stream, b->seeds, c->snp_table(), b->n, read_lo);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// Mean of table[bin] over the 4-bit quality bins of every read (src/alignment.rs:254-260: the mean base accuracy that ranks the reads of a
// cluster for the POA).  One thread per read adds its bins in order -- the same f64 additions in the same order as the reference's fold, then
// one IEEE division -- so the values, and with them the stable sort by accuracy, are bit-identical to the host computation they replace.
__global__ void k_qualbin_mean(const u8* __restrict__ qualbins, const u64* __restrict__ qb_off, const u64* __restrict__ off, u32 n, const double* __restrict__ table, double* __restrict__ out) {
    __shared__ double tab[16];
    if (threadIdx.x < 16) tab[threadIdx.x] = table[threadIdx.x];
    __syncthreads();
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const u64 nb = (off[r + 1] - off[r] + 3) / 4;
    const u8* qb = qualbins + qb_off[r];
    double tot = 0.0;
    for (u64 b = 0; b < nb; b++) tot += tab[(qb[b >> 1] >> (4 * (b & 1))) & 15];
    out[r] = nb ? tot / (double)nb : 1.0;
}
int launch_qualbin_mean(svt_ctx* c, const svt_batch* b, const double* d_table, double* d_out) {
    if (b->n == 0) return SVT_OK;
    ProfScope ps(c, "k_qualbin_mean", (double)b->seeds.qb_bytes + 8.0 * b->n, (double)b->n);
    hipLaunchKernelGGL(k_qualbin_mean, dim3((b->n + 127) / 128), dim3(128), 0, c->stream, b->seeds.qualbins, b->seeds.qb_off, b->d_off, b->n, d_table, d_out);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
