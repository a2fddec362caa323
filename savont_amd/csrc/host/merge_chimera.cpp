// merge_chimera.cpp -- Stage 5 (merge similar consensuses) and Stage 6 (chimera detection) above the C-ABI.
// Reference: src/alignment.rs:1162-1208 (remove_similar_seqs_kmers), :1213-1517 (merge_similar_consensuses), :98-188
// (calculate_adjusted_errors), src/chimera.rs:37-269 (detect_chimeras), :274-399 (calculate_match_lengths), :465-494.
//
// Every consensus-vs-consensus alignment of the reference is a minimap2 call (lrhq index of all consensuses in Stage 5, one
// map-ont aligner per (query, parent) in Stage 6).  Here they are batched GPU calls on ONE resident batch of consensuses:
// K7 (svt_minimizer_shared_counts) votes the strand, K8 (svt_align_nm) prefilters by NM, K9 (svt_align_pileup) gives the
// traceback from which the CIGAR the reference walks is rebuilt.  minimap2 is third-party: its local alignment is replaced by
// the K8/K9 contract; everything downstream of the CIGAR follows the reference statement by statement,
// including two behaviours that look unintended but decide the output:
//   * src/alignment.rs:1491 rebuilds every consensus with ConsensusSequence::new => appended_depth (the only product of the
//     low-quality -> high-quality mapping pass, :1252-1293) is reset to 0; that pass is therefore not executed here;
//   * src/chimera.rs:454 stores similarities under (j, i) with j > i while every lookup (:143,:153,:175,:227) asks for
//     (min, max): all lookups miss, so similarity_score = 1.0, parent_similarity = 0.0, chimera_score = 0, and "detection
//     step 2" (:220-250) can never fire.  calculate_pairwise_similarities is therefore not executed either.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <unordered_map>

#include "asv_pipeline.hpp"
#include "worker_pool.hpp"

namespace savont {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

static void chk5(svt_ctx* c, int rc, const char* what) {
    if (rc != SVT_OK) throw Error{rc, std::string(what) + ": " + svt_last_error(c)};
}
static u32 band5(const ClusterArgs& args, u32 n, u32 m) {
    if (args.align_band) return args.align_band;
    const u32 mx = std::max(n, m), df = n > m ? n - m : m - n;
    return std::min<u32>(std::max((mx + 12) / 13, df), 511);
}
static inline u64 mm_hash64(u64 key) {                                           // src/seeding.rs:18-28
    key = (~key) + (key << 21); key = key ^ key >> 24; key = (key + (key << 3)) + (key << 8); key = key ^ key >> 14;
    key = (key + (key << 2)) + (key << 4); key = key ^ key >> 28; key = key + (key << 31);
    return key;
}
static inline u64 byte_to_seq(u8 b) {                                            // src/types.rs:92-101
    switch (b) { case 1: case 'C': case 'c': return 1; case 2: case 'G': case 'g': return 2; case 3: case 'T': case 't': case 'U': case 'u': return 3; default: return 0; }
}
static std::vector<u8> revcomp(const std::vector<u8>& s) {                        // src/utils.rs reverse_complement
    std::vector<u8> r(s.rbegin(), s.rend());
    for (auto& b : r) b = (b == 'A') ? 'T' : (b == 'C') ? 'G' : (b == 'G') ? 'C' : (b == 'T') ? 'A' : b;
    return r;
}
static size_t position_min(const std::vector<u64>& w) {                          // src/seeding.rs:91-97: LAST minimum
    size_t best = 0;
    for (size_t i = 1; i < w.size(); i++) if (w[i] <= w[best]) best = i;
    return best;
}

// seeding::minimizer_seeds_positions (src/seeding.rs:99-186), k-mer values only.  Kept as written there: the first element is
// the last CANONICAL K-MER of the first window (not a hash, :145), and rolling_kmer_f is not masked while the first k+w-1
// bases are loaded (:123-141), so those "canonical" comparisons see up to 2(k+w-1) bits.
std::vector<u64> minimizer_seeds(const u8* s, size_t len, size_t w, size_t k) {
    std::vector<u64> out;
    if (len < k + w - 1) return out;
    u64 f = 0, r = 0, canonical_kmer = 0;
    const u64 rshift = 2 * (k - 1), max_mask = ~0ull >> (64 - 2 * k), rev_mask = ~(3ull << (2 * k - 2));
    std::vector<u64> win(w, ~0ull);
    for (size_t i = 0; i < k + w - 1; i++) {
        const u64 nf = byte_to_seq(s[i]), nr = 3 - nf;
        f <<= 2; f |= nf; r >>= 2; r |= nr << rshift;
        if (i >= k - 1) { canonical_kmer = f < r ? f : r; win[i + 1 - k] = mm_hash64(canonical_kmer); }
    }
    size_t min_pos = position_min(win); u64 min_val = win[min_pos];
    out.push_back(canonical_kmer);
    for (size_t i = k + w - 1; i < len; i++) {
        const u64 nf = byte_to_seq(s[i]), nr = 3 - nf;
        f <<= 2; f |= nf; f &= max_mask; r >>= 2; r &= rev_mask; r |= nr << rshift;
        const u64 h = mm_hash64(f < r ? f : r);
        const size_t g = i + 1 - k;
        win[g % w] = h;
        if (h < min_val) { min_val = h; min_pos = g % w; out.push_back(h); }
        else if (min_pos == g % w) { min_pos = position_min(win); min_val = win[min_pos]; out.push_back(min_val); }
    }
    return out;
}

// remove_similar_seqs_kmers, src/alignment.rs:1162-1208.  The reference emits the survivors in HashMap order; here: input order.
static std::vector<ConsensusSequence> remove_similar_seqs_kmers(std::vector<ConsensusSequence> cons) {
    const size_t adapter_buffer = 25, n = cons.size();
    std::vector<std::vector<u64>> minis(n); std::vector<char> has(n, 0);
    par_for(n, [&](size_t i) {
        const std::vector<u8>& s = cons[i].sequence;
        if (s.size() < 100) return;                                               // :1169 (dropped altogether)
        minis[i] = minimizer_seeds(s.data() + adapter_buffer, s.size() - 2 * adapter_buffer, 10, 21);
        has[i] = 1;
    });
    // possible_greater_ids (:1182-1201) = the consensuses more than twice as deep that hold the FIRST list element, intersected with
    // the holders of every further element: i.e. the deeper consensuses whose minimizer set contains all of e's elements.
    std::vector<std::vector<u64>> uniq(n);
    for (size_t i = 0; i < n; i++) if (has[i]) { uniq[i] = minis[i]; std::sort(uniq[i].begin(), uniq[i].end()); uniq[i].erase(std::unique(uniq[i].begin(), uniq[i].end()), uniq[i].end()); }
    std::vector<ConsensusSequence> out; std::vector<char> drop(n, 0);
    for (size_t e = 0; e < n; e++) {
        if (!has[e]) { drop[e] = 1; continue; }
        if (minis[e].empty()) continue;                                           // shorter than k + w - 1 after trimming: kept (:1202)
        for (size_t id = 0; id < n; id++) {
            if (!has[id] || !(cons[id].depth / 2 > cons[e].depth)) continue;
            if (!std::binary_search(uniq[id].begin(), uniq[id].end(), minis[e][0])) continue;    // holders of the first element (:1186-1191)
            if (std::includes(uniq[id].begin(), uniq[id].end(), uniq[e].begin(), uniq[e].end())) { drop[e] = 1; break; }
        }
    }
    for (size_t e = 0; e < n; e++) if (!drop[e]) out.push_back(std::move(cons[e]));
    return out;
}

// ---- consensus-vs-consensus alignments (the reference's minimap2 calls) ---------------------------------------------
struct PairAlignment {                         // minimap2 vocabulary: QUERY consensus mapped onto TARGET consensus
    bool mapped = false, rev = false; int32_t nm = 0;
    u32 query_start = 0, query_end = 0, target_start = 0, target_end = 0;        // query coordinates in the ALIGNED orientation
    std::vector<std::pair<u32, u8>> cigar;     // (len, op) op 0 = M, 1 = I (in query), 2 = D
};
struct ConsensusBatch {
    svt_ctx* ctx; svt_batch* b = nullptr; std::vector<u64> off;
    ConsensusBatch(svt_ctx* c, const std::vector<ConsensusSequence>& cons, const ClusterArgs& args) : ctx(c), off(1, 0) {
        std::vector<u8> seq;
        for (auto& x : cons) { seq.insert(seq.end(), x.decompressed.begin(), x.decompressed.end()); off.push_back(seq.size()); }
        if (seq.empty()) seq.push_back('A');
        chk5(ctx, svt_batch_upload(ctx, seq.data(), nullptr, off.data(), (u32)cons.size(), &b), "svt_batch_upload(consensuses)");
        const int rc = svt_extract_seeds(ctx, b, args.kmer_size, args.c, args.minimum_base_quality, 0);
        if (rc != SVT_OK) { svt_batch_free(ctx, b); b = nullptr; chk5(ctx, rc, "svt_extract_seeds(consensuses)"); }
    }
    ~ConsensusBatch() { if (b) svt_batch_free(ctx, b); }
    u32 len(u32 i) const { return (u32)(off[i + 1] - off[i]); }
};
// strand of q relative to t (K7): returns mapped flags + rev flags
static void strand_votes(const ConsensusBatch& cb, const std::vector<u32>& q, const std::vector<u32>& t, std::vector<u8>& mapped, std::vector<u8>& rev) {
    const size_t n = q.size(); mapped.assign(n, 0); rev.assign(n, 0);
    if (!n) return;
    std::vector<u32> shared(n), same(n);
    chk5(cb.ctx, svt_minimizer_shared_counts(cb.ctx, cb.b, cb.b, q.data(), t.data(), n, shared.data(), same.data()), "svt_minimizer_shared_counts(consensuses)");
    for (size_t i = 0; i < n; i++) { mapped[i] = shared[i] > 0; rev[i] = (shared[i] - same[i]) > same[i]; }
}
// ---- the pair lists of stages 5 and 6 over the ranks of a shard (round 5) -----------------------------------------------
// Every rank holds the same consensus set and makes the same calls here, so a list of independent pairs is dealt out like the K5 pairs of Stage 2: rank r takes the
// contiguous slice [n r / W, n (r + 1) / W), the per-pair results -- an nm, or an alignment as (flags, nm, span, CIGAR) -- are gathered (svt_shard_allgatherv: the
// library's exchange path, RCCL or the hook), and every rank goes on with the whole list.  Results are those of the one-rank run: every pair has exactly one owner.
struct PairSlice { u32 rank = 0, world = 1; size_t lo = 0, hi = 0; bool on = false; };
// The slicing is only right if every rank holds the SAME pair list here.  That is the construction (identical consensus sets, identical calls), but a rank-local error or a
// caller that runs merge / chimera per rank on different data would make the gathers below meet with different sizes -- garbage alignments that pass the length checks, or a
// wait until the deadline.  So the ranks compare a hash of (n, q, t) first (one u64 per rank through the library's exchange path) and fail loudly on a mismatch; and no list
// is dealt out while the caller has paused the slicing (svt_shard_pause: the ranks make different calls).  ADVICE r05.
static PairSlice pair_slice(svt_ctx* ctx, size_t n, const std::vector<u32>& q, const std::vector<u32>& t) {
    PairSlice s; s.hi = n;
    svt_shard_info(ctx, &s.rank, &s.world);
    if (s.world <= 1 || n < 8 * (size_t)s.world) { s.world = std::max<u32>(s.world, 1); return s; }
    if (svt_shard_pause(ctx, 0) == 1) { svt_shard_pause(ctx, 1); return s; }            // paused by the caller: this rank's list is its own
    u64 h = 1469598103934665603ull;
    auto mix = [&](u64 x) { h = (h ^ x) * 1099511628211ull; };
    mix((u64)n);
    for (size_t i = 0; i < n; i++) mix(((u64)q[i] << 32) | t[i]);
    std::vector<u64> all(s.world, 0);
    chk5(ctx, svt_shard_allgather_u64(ctx, h, all.data()), "svt_shard_allgather_u64(pair list hash)");
    for (u32 r = 0; r < s.world; r++) if (all[r] != h) throw Error{SVT_ERR_STATE, "stages 5/6: the ranks of the shard hold different pair lists (rank " + std::to_string(r) + " differs from rank " + std::to_string(s.rank) + "); nothing was dealt out"};
    s.on = true; s.lo = n * s.rank / s.world; s.hi = n * (s.rank + 1) / s.world;
    return s;
}
static void sharded_align_nm(svt_ctx* ctx, const svt_batch* Q, const svt_batch* T, const std::vector<u32>& q, const std::vector<u32>& t, const std::vector<u8>& rev,
                             const std::vector<u32>& band, std::vector<int32_t>& nm) {
    const size_t n = q.size(); nm.assign(n, 0);
    if (!n) return;
    const PairSlice sl = pair_slice(ctx, n, q, t);
    if (sl.hi > sl.lo) chk5(ctx, svt_align_nm(ctx, Q, T, q.data() + sl.lo, t.data() + sl.lo, rev.data() + sl.lo, band.data() + sl.lo, sl.hi - sl.lo, nm.data() + sl.lo), "svt_align_nm(consensuses)");
    if (!sl.on) return;
    std::vector<u64> bytes(sl.world);
    for (u32 r = 0; r < sl.world; r++) bytes[r] = (n * (r + 1) / sl.world - n * r / sl.world) * 4;
    std::vector<int32_t> all(n);
    chk5(ctx, svt_shard_allgatherv(ctx, nm.data() + sl.lo, bytes.data(), all.data()), "svt_shard_allgatherv(consensus nm)");
    nm.swap(all);
}

// K9 for (query q[i] onto target t[i]) -> CIGAR; processed in slabs so the traceback cells stay bounded on the host
static std::vector<PairAlignment> align_pairs_local(const ConsensusBatch& cb, const std::vector<u32>& q, const std::vector<u32>& t,
                                              const std::vector<u8>& mapped, const std::vector<u8>& rev, const ClusterArgs& args) {
    const size_t n = q.size();
    std::vector<PairAlignment> out(n);
    const size_t SLAB = 16384;
    for (size_t s0 = 0; s0 < n; s0 += SLAB) {
        const size_t s1 = std::min(n, s0 + SLAB);
        std::vector<u32> kq, kt, band, src; std::vector<u8> kr;
        for (size_t i = s0; i < s1; i++) if (mapped[i]) { kq.push_back(t[i]); kt.push_back(q[i]); kr.push_back(rev[i]); band.push_back(band5(args, cb.len(t[i]), cb.len(q[i]))); src.push_back((u32)i); }
        const size_t m = kq.size();
        if (!m) continue;
        std::vector<u64> cell_off(m + 1, 0);
        for (size_t i = 0; i < m; i++) cell_off[i + 1] = cell_off[i] + cb.len(kq[i]);
        static thread_local std::vector<u64> cells;                               // kept between calls: ~20 MB that the library overwrites whole (zero-filling them cost as much as reading them)
        if (cells.size() < cell_off[m]) cells.resize(cell_off[m]);
        std::vector<u32> span(m * 4); std::vector<int32_t> nm(m);
        chk5(cb.ctx, svt_align_pileup(cb.ctx, cb.b, cb.b, kq.data(), kt.data(), kr.data(), band.data(), m, cell_off.data(), cells.data(), span.data(), nm.data()), "svt_align_pileup(consensuses)");
        for (size_t i = 0; i < m; i++) {
            PairAlignment& a = out[src[i]];
            if (nm[i] == INT32_MAX || nm[i] < 0) continue;
            a.mapped = true; a.rev = kr[i] != 0; a.nm = nm[i];
            a.target_start = span[i * 4]; a.target_end = span[i * 4 + 1]; a.query_start = span[i * 4 + 2]; a.query_end = span[i * 4 + 3];
            const u64* row = &cells[cell_off[i]];
            auto push = [&](u32 len, u8 op) { if (!len) return; if (!a.cigar.empty() && a.cigar.back().second == op) a.cigar.back().first += len; else a.cigar.push_back({len, op}); };
            u32 run = 0;                                                       // aligned columns without an insertion behind them: nearly all of a consensus x consensus alignment
            for (u32 p = a.target_start; p < a.target_end; p++) {
                const u64 c = row[p]; const u32 code = (u32)(c & 7), ins = (u32)((c >> 18) & 0xFF);
                if (code < 4 && !ins) { run++; continue; }
                push(run, 0); run = 0;
                if (code < 4) push(1, 0); else if (code == 4) push(1, 2);
                push(ins, 1);
            }
            push(run, 0);
        }
    }
    return out;
}
static std::vector<PairAlignment> align_pairs(const ConsensusBatch& cb, const std::vector<u32>& q, const std::vector<u32>& t,
                                              const std::vector<u8>& mapped, const std::vector<u8>& rev, const ClusterArgs& args) {
    const size_t n = q.size();
    const PairSlice sl = pair_slice(cb.ctx, n, q, t);
    if (!sl.on) return align_pairs_local(cb, q, t, mapped, rev, args);
    // this rank's slice, then (mapped | rev << 1, nm, span[4], CIGAR as (len << 2 | op) words) of every pair, slice after slice in rank order
    const std::vector<u32> qs(q.begin() + sl.lo, q.begin() + sl.hi), ts(t.begin() + sl.lo, t.begin() + sl.hi);
    const std::vector<u8> ms(mapped.begin() + sl.lo, mapped.begin() + sl.hi), rs_(rev.begin() + sl.lo, rev.begin() + sl.hi);
    const std::vector<PairAlignment> mine = align_pairs_local(cb, qs, ts, ms, rs_, args);
    std::vector<u32> buf;
    for (const PairAlignment& a : mine) {
        buf.push_back((a.mapped ? 1u : 0u) | (a.rev ? 2u : 0u)); buf.push_back((u32)a.nm);
        buf.push_back(a.query_start); buf.push_back(a.query_end); buf.push_back(a.target_start); buf.push_back(a.target_end);
        buf.push_back((u32)a.cigar.size());
        for (auto& op : a.cigar) buf.push_back((op.first << 2) | op.second);
    }
    std::vector<u64> bytes(sl.world, 0);
    chk5(cb.ctx, svt_shard_allgather_u64(cb.ctx, buf.size() * 4, bytes.data()), "svt_shard_allgather_u64(pair alignments)");
    u64 total = 0; for (u64 b : bytes) total += b;
    std::vector<u32> all(total / 4 + 1);
    chk5(cb.ctx, svt_shard_allgatherv(cb.ctx, buf.data(), bytes.data(), all.data()), "svt_shard_allgatherv(pair alignments)");
    std::vector<PairAlignment> out(n);
    size_t w = 0;
    for (size_t i = 0; i < n; i++) {
        if (w + 7 > total / 4) throw Error{SVT_ERR_EXCHANGE, "align_pairs: the gathered pair alignments are shorter than the pair list (the ranks diverged)"};
        PairAlignment& a = out[i];
        a.mapped = (all[w] & 1) != 0; a.rev = (all[w] & 2) != 0; a.nm = (int32_t)all[w + 1];
        a.query_start = all[w + 2]; a.query_end = all[w + 3]; a.target_start = all[w + 4]; a.target_end = all[w + 5];
        const u32 nops = all[w + 6]; w += 7;
        if (w + nops > total / 4) throw Error{SVT_ERR_EXCHANGE, "align_pairs: a gathered CIGAR runs past the buffer (the ranks diverged)"};
        a.cigar.resize(nops);
        for (u32 x = 0; x < nops; x++) a.cigar[x] = {all[w + x] >> 2, (u8)(all[w + x] & 3)};
        w += nops;
    }
    return out;
}

// has_homopolymer_context, src/alignment.rs:75-96
static bool has_homopolymer_context(const std::vector<u8>& seq, size_t pos, size_t window) {
    if (seq.empty()) return false;
    const size_t start = pos >= window ? pos - window : 0, end = std::min(pos + window + 1, seq.size());
    if (end <= start + 2) return false;
    for (size_t i = start; i <= (end >= 3 ? end - 3 : 0); i++) if (i + 2 < seq.size() && seq[i] == seq[i + 1] && seq[i + 1] == seq[i + 2]) return true;
    return false;
}
// calculate_adjusted_errors, src/alignment.rs:101-188
static size_t calculate_adjusted_errors(const std::vector<std::pair<u32, u8>>& cigar, const std::vector<u8>& q, const std::vector<u8>& t, size_t query_start, size_t target_start) {
    size_t err = 0, qp = query_start, tp = target_start; const size_t buffer = 35;
    for (auto& op : cigar) {
        const size_t len = op.first;
        if (op.second == 0) {
            for (size_t x = 0; x < len; x++) {
                if (qp < q.size() && tp < t.size() && q[qp] != t[tp] && q[qp] != 'N' && t[tp] != 'N' && qp > buffer && qp + buffer < q.size()) err++;
                qp++; tp++;
            }
        } else if (op.second == 1) {
            const bool hp = has_homopolymer_context(q, qp, 2) || has_homopolymer_context(t, tp, 2);
            if (!hp && qp > buffer && qp + len + buffer < q.size()) err += len < 10 ? 1 : len;
            qp += len;
        } else {
            const bool hp = has_homopolymer_context(q, qp, 2) || has_homopolymer_context(t, tp, 2);
            if (!hp && tp > buffer && tp + len + buffer < t.size()) err += len < 10 ? 1 : len;
            tp += len;
        }
    }
    return err;
}

// ==================================================================================================
// Stage 5: alignment::merge_similar_consensuses (src/alignment.rs:1213-1517)
// ==================================================================================================
std::vector<ConsensusSequence> merge_similar_consensuses(const ReadSet& rs, std::vector<ConsensusSequence> consensuses_in,
                                                         const std::vector<ConsensusSequence>& /*low_qual: see header*/, const ClusterArgs& args) {
    if (consensuses_in.empty()) return consensuses_in;                            // :1220
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tsec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    auto tm0 = tnow();
    std::vector<ConsensusSequence> cons = remove_similar_seqs_kmers(std::move(consensuses_in));   // :1228
    auto tm1 = tnow(); trace_add("5.dedup", tsec(tm0, tm1));
    const size_t n = cons.size();
    if (n == 0) return cons;
    struct Mapping { u32 q, t; size_t nm, t_depth; };
    std::vector<Mapping> mappings;
    {
        ConsensusBatch cb(rs.ctx, cons, args);
        auto tm2 = tnow(); trace_add("5.batch+seeds", tsec(tm1, tm2));
        std::vector<u32> pq, pt;
        for (u32 i = 0; i < n; i++) for (u32 j = i + 1; j < n; j++) { pq.push_back(i); pt.push_back(j); }
        std::vector<u8> mapped, rev;
        strand_votes(cb, pq, pt, mapped, rev);
        auto tm3 = tnow(); trace_add("5.k7", tsec(tm2, tm3));
        // NM prefilter (:1319 `alignment.nm > 30` skips the mapping): overlap edit distance is symmetric, one K8 pass per unordered pair
        std::vector<u32> fq, ft, band; std::vector<u8> fr;
        for (size_t i = 0; i < pq.size(); i++) if (mapped[i]) { fq.push_back(pt[i]); ft.push_back(pq[i]); fr.push_back(rev[i]); band.push_back(band5(args, cb.len(pt[i]), cb.len(pq[i]))); }
        std::vector<int32_t> nm;
        sharded_align_nm(rs.ctx, cb.b, cb.b, fq, ft, fr, band, nm);
        std::vector<u32> q2, t2; std::vector<u8> m2, r2;
        for (size_t i = 0; i < fq.size(); i++) if (nm[i] >= 0 && nm[i] <= 30) {
            q2.push_back(ft[i]); t2.push_back(fq[i]); m2.push_back(1); r2.push_back(fr[i]);      // i -> j
            q2.push_back(fq[i]); t2.push_back(ft[i]); m2.push_back(1); r2.push_back(fr[i]);      // j -> i
        }
        auto tm4 = tnow(); trace_add("5.k8", tsec(tm3, tm4));
        std::vector<PairAlignment> al = align_pairs(cb, q2, t2, m2, r2, args);
        auto tm5 = tnow(); trace_add("5.k9", tsec(tm4, tm5));
        std::vector<std::vector<u8>> rc_of(n);                                    // a query's reverse complement, made when its first reverse pair asks for it (it was made per pair: 3 % of a 2-CPU step)
        for (size_t i = 0; i < al.size(); i++) {
            const PairAlignment& a = al[i];
            if (!a.mapped) continue;
            const std::vector<u8>& qs = cons[q2[i]].decompressed; const std::vector<u8>& ts = cons[t2[i]].decompressed;
            if ((size_t)(a.query_end - a.query_start) < qs.size() * 3 / 4 || a.nm > 30) continue;   // :1319
            if (a.rev && rc_of[q2[i]].empty() && !qs.empty()) rc_of[q2[i]] = revcomp(qs);
            size_t adj = a.rev ? calculate_adjusted_errors(a.cigar, rc_of[q2[i]], ts, a.query_start, a.target_start)      // :1326-1334 (coordinates already in rc space)
                               : calculate_adjusted_errors(a.cigar, qs, ts, a.query_start, a.target_start);
            if ((size_t)a.nm < adj) adj = (size_t)a.nm;                           // :1349-1353
            mappings.push_back({q2[i], t2[i], adj, cons[t2[i]].depth});
        }
    }
    std::sort(mappings.begin(), mappings.end(), [](const Mapping& a, const Mapping& b) { return a.q != b.q ? a.q < b.q : a.t < b.t; });
    std::map<size_t, size_t> merge_map;                                           // :1374
    size_t mi = 0;
    for (size_t query_idx = 0; query_idx < n; query_idx++) {
        struct VT { size_t t, nm, depth; };
        std::vector<VT> valid;
        for (; mi < mappings.size() && mappings[mi].q == query_idx; mi++) {
            const Mapping& m = mappings[mi];
            if (m.q == m.t) continue;
            const size_t qd = cons[query_idx].depth, td = m.t_depth;
            const double rel = (double)qd / (double)td;
            double thr = std::pow(0.5, (double)m.nm * 0.75 + 1.25);               // :1393
            bool ok;
            if (m.nm == 0) {
                thr = 0.999999;
                if (qd == td) { if (query_idx > m.t) valid.push_back({m.t, m.nm, td}); continue; }   // :1398-1406
            }
            ok = (rel < thr) || (1.0 / rel < thr);                                // :1413
            if (ok) valid.push_back({m.t, m.nm, td});
        }
        if (valid.empty()) continue;
        struct QR { size_t a, nm, depth, b; };
        std::vector<QR> q2r, r2q;
        for (auto& v : valid) {
            if (cons[v.t].depth == cons[query_idx].depth) { if (v.nm == 0 && query_idx > v.t) merge_map[query_idx] = v.t; continue; }   // :1423-1431
            else if (cons[v.t].depth > cons[query_idx].depth) q2r.push_back({v.t, v.nm, v.depth, query_idx});
            else r2q.push_back({query_idx, v.nm, cons[query_idx].depth, v.t});
        }
        if (!q2r.empty()) { std::stable_sort(q2r.begin(), q2r.end(), [](const QR& a, const QR& b) { return a.depth > b.depth; }); merge_map[query_idx] = q2r[0].a; }   // :1439-1443
        for (auto& x : r2q) if (!merge_map.count(x.b)) merge_map[x.b] = query_idx;   // :1445-1449
    }
    std::vector<std::vector<u32>> new_clusters(n);
    for (size_t i = 0; i < n; i++) new_clusters[i] = cons[i].cluster;
    std::map<size_t, size_t> merged_into;
    for (size_t q = 0; q < n; q++) {                                              // :1458-1466
        auto it = merge_map.find(q);
        if (it == merge_map.end()) continue;
        size_t fin = it->second, guard = 0;
        for (auto nx = merge_map.find(fin); nx != merge_map.end() && guard <= n; nx = merge_map.find(fin), guard++) fin = nx->second;
        if (guard > n) throw Error{SVT_ERR_STATE, "merge_similar_consensuses: circular merge chain (the reference does not terminate on this input)"};
        merged_into[q] = fin;
    }
    for (auto& kv : merged_into) {                                                // :1469-1482
        std::vector<u32> mv = new_clusters[kv.first];
        new_clusters[kv.second].insert(new_clusters[kv.second].end(), mv.begin(), mv.end());
        new_clusters[kv.first].clear();
    }
    std::vector<ConsensusSequence> out;
    for (size_t i = 0; i < n; i++) if (!new_clusters[i].empty()) {                // :1487-1495
        ConsensusSequence c; c.sequence = cons[i].sequence; c.hp_lengths = cons[i].hp_lengths; c.depth = new_clusters[i].size(); c.id = cons[i].id; c.cluster = new_clusters[i];
        decompress(c);
        out.push_back(std::move(c));
    }
    std::stable_sort(out.begin(), out.end(), [](const ConsensusSequence& a, const ConsensusSequence& b) { return a.depth > b.depth; });   // :1503
    return out;
}

// calculate_match_lengths, src/chimera.rs:274-399.  -1 = None
static void calculate_match_lengths(const PairAlignment& a, const std::vector<u8>& q, const std::vector<u8>& t, const ClusterArgs& args, long& left_out, long& right_out) {
    size_t left = 0, right = 0; const size_t pcr_slack = 15, allow = args.chimera_allowable_errors;
    {
        size_t errs = 0, qp = a.query_start, tp = a.target_start;
        for (auto& op : a.cigar) {
            if (errs > allow) break;
            const size_t len = op.first;
            if (op.second == 0) {
                for (size_t i = 0; i < len; i++) if (qp + i < q.size() && tp + i < t.size()) {
                    if (q[qp + i] == t[tp + i]) left++;
                    else { errs++; if (errs > allow && qp + i >= pcr_slack) break; }
                }
                qp += len; tp += len;
            } else if (op.second == 1) qp += len; else tp += len;
        }
    }
    {
        size_t errs = 0, qp = a.query_end, tp = a.target_end;
        for (auto it = a.cigar.rbegin(); it != a.cigar.rend(); ++it) {
            if (errs > allow) break;
            const size_t len = it->first;
            if (it->second == 0) {
                for (size_t i = 0; i < len; i++) {
                    if (q[qp - i - 1] == t[tp - i - 1]) right++;
                    else { errs++; if (errs > allow && qp - i + pcr_slack <= q.size()) break; }
                }
                qp -= len; tp -= len;
            } else if (it->second == 1) qp -= len; else tp -= len;
        }
    }
    const size_t min_len = args.chimera_detect_length ? args.chimera_detect_length : std::max<size_t>(args.min_read_length / 10, 100);   // :383
    long r = (long)right, l = (long)left;
    if (right < min_len || left >= right) r = -1;                                 // :385-387
    if (left < min_len || right >= left) l = -1;                                  // :389-391
    if (a.rev) { left_out = r; right_out = l; } else { left_out = l; right_out = r; }   // :393-398
}

// ==================================================================================================
// Stage 6: chimera::detect_chimeras + filter_chimeras (src/chimera.rs:37-269, :465-494) -> consensuses without the chimeras;
// chimera_idx (optional) receives the indices (into the input) that were removed.
// ==================================================================================================
std::vector<ConsensusSequence> detect_and_filter_chimeras(const ReadSet& rs, std::vector<ConsensusSequence> cons, const ClusterArgs& args, std::vector<u32>* chimera_idx) {
    const size_t n = cons.size();
    if (chimera_idx) chimera_idx->clear();
    if (n == 0) return cons;
    std::vector<u32> pq, pt;
    for (u32 q = 0; q < n; q++) for (u32 r = 0; r < n; r++) {
        if (r == q) continue;
        if (cons[r].depth <= cons[q].depth * 3) continue;                         // :79
        pq.push_back(q); pt.push_back(r);
    }
    std::vector<char> is_chimera(n, 0);
    if (!pq.empty()) {
        ConsensusBatch cb(rs.ctx, cons, args);
        std::vector<u8> mapped, rev;
        strand_votes(cb, pq, pt, mapped, rev);
        std::vector<PairAlignment> al = align_pairs(cb, pq, pt, mapped, rev, args);
        size_t i = 0;
        for (u32 q = 0; q < n; q++) {
            std::vector<std::pair<u32, size_t>> lefts, rights;                    // (ref, len) in ref order
            const std::vector<u8> q_rc = revcomp(cons[q].decompressed);
            for (; i < pq.size() && pq[i] == q; i++) {
                if (!al[i].mapped) continue;
                long l = -1, r = -1;
                calculate_match_lengths(al[i], al[i].rev ? q_rc : cons[q].decompressed, cons[pt[i]].decompressed, args, l, r);
                if (l >= 0) lefts.push_back({pt[i], (size_t)l});
                if (r >= 0) rights.push_back({pt[i], (size_t)r});
            }
            const double qlen = (double)cons[q].decompressed.size();
            for (auto& L : lefts) for (auto& R : rights) {                        // :165-217 with parent_similarity = 0.0 (see header)
                if (L.first == R.first) continue;
                const double parent_similarity = 0.0;
                const double cov = (double)(L.second + R.second) / qlen;
                if (cov >= std::min(0.9 * std::max(parent_similarity, 0.7), 0.8) && (cov < 1.5 || (parent_similarity < 0.99 && cov < 1.8))) { is_chimera[q] = 1; break; }
            }
        }
    }
    std::vector<ConsensusSequence> out;
    for (size_t i = 0; i < n; i++) {
        if (is_chimera[i]) { if (chimera_idx) chimera_idx->push_back((u32)i); continue; }
        out.push_back(std::move(cons[i]));
    }
    return out;
}

}  // namespace savont
