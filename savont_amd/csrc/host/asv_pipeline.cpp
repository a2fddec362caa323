// asv_pipeline.cpp -- host logic of the `savont asv` hot path above the C-ABI (see asv_pipeline.hpp).
// Reference citations are file:line relative to the reference root (bluenote-1577/savont v0.6.4).
#include "asv_pipeline.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <unordered_map>

#include "stats.hpp"

namespace savont {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

static void chk(svt_ctx* c, int rc, const char* what) {
    if (rc != SVT_OK) throw Error{rc, std::string(what) + ": " + svt_last_error(c)};
}

// ==================================================================================================
// Stage 1a: seq_parse::read_to_split_kmers (src/seq_parse.rs:12-78).  The three-level thread pipeline
// of :316-497 (reader -> split_kmer_mid workers -> kmer%threads hash-map shards) is ONE fused GPU pass.
// ==================================================================================================
KmerCountTable read_to_split_kmers(const ReadSet& rs, const ClusterArgs& args, u64* n_distinct) {
    u64 nd = 0, nk = 0;
    chk(rs.ctx, svt_count_split_kmers(rs.ctx, rs.batch, args.kmer_size, args.minimum_base_quality,
                                      rs.rc_flags.empty() ? nullptr : rs.rc_flags.data(), args.single_strand ? 1 : 0, &nd, &nk),
        "svt_count_split_kmers");
    std::vector<u64> km(nk); std::vector<u32> rev(nk), fwd(nk);
    chk(rs.ctx, svt_count_fetch(rs.ctx, km.data(), rev.data(), fwd.data()), "svt_count_fetch");
    if (n_distinct) *n_distinct = nd;
    if (nk < nd / 1000)                                                        // :69-72 (process::exit(1) in the reference)
        throw Error{1, "Less than 0.1% of SNPmers have counts > 1 in both strands and > 2 multiplicity. Consider --single-strand"};
    KmerCountTable t(nk);
    for (u64 i = 0; i < nk; i++) t[i] = {km[i], {rev[i], fwd[i]}};
    return t;
}

// ==================================================================================================
// Stage 1b: kmer_comp::get_snpmers_inplace_sort (src/kmer_comp.rs:454-642).  Host: the filtered table
// is small (1e4..1e6 rows) and already in the sort order of :480, so this is one linear pass.
// ==================================================================================================
KmerGlobalInfo get_snpmers_inplace_sort(const KmerCountTable& table, u32 k, const ClusterArgs& args) {
    KmerGlobalInfo info;
    const size_t n = table.size();
    if (n == 0) throw Error{1, "No k-mers found. Exiting."};                  // :469-472
    std::vector<u32> counts(n);
    for (size_t i = 0; i < n; i++) counts[i] = table[i].second.first + table[i].second.second;
    std::vector<u32> sorted = counts;
    std::sort(sorted.begin(), sorted.end());
    const u32 thresh = std::max<u32>(sorted[n - n / 100000 - 1], 100);       // :474
    info.high_freq_thresh = thresh;
    const u64 sm = 3ull << (k - 1);
    struct E { u64 kmer; u32 c0, c1; };
    std::vector<E> group;
    auto flush = [&]() {
        if (group.size() > 1) {                                               // :507-519
            std::stable_sort(group.begin(), group.end(), [](const E& a, const E& b) { return a.c0 + a.c1 > b.c0 + b.c1; });   // :554
            const u64 nn = (u64)group[0].c0 + group[0].c1, succ = (u64)group[1].c0 + group[1].c1;
            const bool cond1 = binomial_test(nn, succ, 0.025) > 0.05;        // :557-559 (cond2 is dead: k < 5 never holds)
            if (!cond1) {
                const u32 a = group[0].c0, b = group[1].c0, c = group[0].c1, d = group[1].c1;
                const u32 t0 = std::max(a, c), t1 = std::max(b, d), t2 = std::min(c, a), t3 = std::min(d, b);   // :575-578
                const double p_value = fisher_two_tail(t0, t1, t2, t3);      // :579
                double odds = 0.0;
                if (!(t0 == 0 || t1 == 0 || t2 == 0 || t3 == 0)) odds = ((double)t0 * (double)t3) / ((double)t1 * (double)t2);
                const bool skip = !args.single_strand && odds == 0.0;        // :586-590
                if (!skip && (p_value > 0.005 || (odds < 1.5 && odds > 1. / 1.5))) {   // :593
                    SnpmerInfo s;
                    s.split_kmer = group[0].kmer & ~sm;
                    s.mid_bases[0] = (u8)((group[0].kmer & sm) >> (k - 1)); s.mid_bases[1] = (u8)((group[1].kmer & sm) >> (k - 1));
                    s.counts[0] = (u32)nn; s.counts[1] = (u32)succ; s.k = (u8)k;
                    info.snpmer_info.push_back(s);
                }
            }
        }
        group.clear();
    };
    u64 cur = ~0ull;
    for (size_t i = 0; i < n; i++) {
        const u32 c0 = table[i].second.first, c1 = table[i].second.second;
        if (c0 + c1 > thresh) info.high_freq_kmers.push_back(table[i].first);   // :494-496
        if (!args.single_strand && (c0 == 0 || c1 == 0)) continue;              // :498-502
        const u64 split = table[i].first & ~sm;
        if (split != cur) { flush(); cur = split; }
        group.push_back({table[i].first, c0, c1});
    }
    flush();
    std::sort(info.snpmer_info.begin(), info.snpmer_info.end(), [](const SnpmerInfo& a, const SnpmerInfo& b) { return a.split_kmer < b.split_kmer; });  // :632
    std::sort(info.high_freq_kmers.begin(), info.high_freq_kmers.end());
    return info;
}

// ==================================================================================================
// Stage 1c: kmer_comp::twin_reads_from_snpmers (src/kmer_comp.rs:68-258) + ordering of src/main.rs:529-548.
// The second FASTQ decode + per-read worker threads of the reference are one seed-extraction pass over
// the reads already resident in HBM.
// ==================================================================================================
TwinReads twin_reads_from_snpmers(const ReadSet& rs, const KmerGlobalInfo& info, const ClusterArgs& args) {
    const u32 n = rs.n, k = args.kmer_size;
    std::vector<u64> split(info.snpmer_info.size()); std::vector<u8> m0(split.size()), m1(split.size());
    for (size_t i = 0; i < split.size(); i++) { split[i] = info.snpmer_info[i].split_kmer; m0[i] = info.snpmer_info[i].mid_bases[0]; m1[i] = info.snpmer_info[i].mid_bases[1]; }
    chk(rs.ctx, svt_set_snpmers(rs.ctx, k, split.data(), m0.data(), m1.data(), (u32)split.size(), info.high_freq_kmers.data(), (u32)info.high_freq_kmers.size()), "svt_set_snpmers");
    chk(rs.ctx, svt_extract_seeds(rs.ctx, rs.batch, k, args.c, args.minimum_base_quality, 1), "svt_extract_seeds");
    u64 nm = 0, ns = 0, nq = 0;
    chk(rs.ctx, svt_seeds_sizes(rs.ctx, rs.batch, &nm, &ns, &nq), "svt_seeds_sizes");
    std::vector<u64> mini_off(n + 1), snp_off(n + 1), lsh((size_t)n * SVT_LSH_TABLES);
    std::vector<u8> snp_flags(ns), est_valid(n), lsh_valid(n), status(n);
    std::vector<double> est(n); std::vector<u32> n_unique(n), n_solid(n);
    svt_seeds_out o; memset(&o, 0, sizeof o);
    o.mini_off = mini_off.data(); o.snp_off = snp_off.data(); o.snp_flags = snp_flags.data(); o.est_id = est.data(); o.est_valid = est_valid.data();
    o.lsh = lsh.data(); o.lsh_valid = lsh_valid.data(); o.n_unique = n_unique.data(); o.n_solid = n_solid.data(); o.status = status.data();
    chk(rs.ctx, svt_seeds_fetch(rs.ctx, rs.batch, &o), "svt_seeds_fetch");
    // intake filters
    std::vector<u32> order;
    for (u32 i = 0; i < n; i++) {
        const u64 len = rs.offsets[i + 1] - rs.offsets[i];
        if (len < args.min_read_length || len > args.max_read_length) continue;     // kmer_comp.rs:117
        if (status[i] != 0) continue;                                               // seeding.rs:339 (None)
        if (n_solid[i] < len / args.c / 20) continue;                               // kmer_comp.rs:185
        order.push_back(i);
    }
    std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return rs.ids[a] < rs.ids[b]; });             // kmer_comp.rs:233
    std::vector<u32> kept;
    for (u32 i : order) if (!est_valid[i] || est[i] >= args.quality_value_cutoff) kept.push_back(i);                 // kmer_comp.rs:248
    auto e100 = [&](u32 i) { return est_valid[i] ? est[i] : 100.0; };
    std::stable_sort(kept.begin(), kept.end(), [&](u32 a, u32 b) { return e100(a) > e100(b); });                     // main.rs:538
    TwinReads tw;
    tw.n = (u32)kept.size();
    tw.words = svt_snpmer_words(rs.ctx);
    std::vector<u64> pf((size_t)n * tw.words), al((size_t)n * tw.words);
    if (tw.words) chk(rs.ctx, svt_snpmer_bits_fetch(rs.ctx, rs.batch, nullptr, pf.data(), al.data()), "svt_snpmer_bits_fetch");
    tw.orig = kept;
    tw.length.resize(tw.n); tw.file_idx.resize(tw.n); tw.n_mini.resize(tw.n); tw.n_unique.resize(tw.n); tw.n_snp_filtered.resize(tw.n);
    tw.est_id.resize(tw.n); tw.est_valid.resize(tw.n); tw.lsh.resize((size_t)tw.n * SVT_LSH_TABLES); tw.lsh_valid.resize(tw.n);
    tw.p_filt.resize((size_t)tw.n * tw.words); tw.allele.resize((size_t)tw.n * tw.words);
    size_t without = 0;
    for (u32 t = 0; t < tw.n; t++) {
        const u32 i = kept[t];
        tw.length[t] = (u32)(rs.offsets[i + 1] - rs.offsets[i]);
        tw.file_idx[t] = rs.file_idx.empty() ? 0 : rs.file_idx[i];
        tw.n_mini[t] = (u32)(mini_off[i + 1] - mini_off[i]); tw.n_unique[t] = n_unique[i];
        u32 f = 0; for (u64 j = snp_off[i]; j < snp_off[i + 1]; j++) f += snp_flags[j] & 1;
        tw.n_snp_filtered[t] = f; if (f == 0) without++;
        tw.est_id[t] = est[i]; tw.est_valid[t] = est_valid[i];
        memcpy(&tw.lsh[(size_t)t * SVT_LSH_TABLES], &lsh[(size_t)i * SVT_LSH_TABLES], SVT_LSH_TABLES * 8);
        tw.lsh_valid[t] = lsh_valid[i];
        if (tw.words) {
            memcpy(&tw.p_filt[(size_t)t * tw.words], &pf[(size_t)i * tw.words], tw.words * 8);
            memcpy(&tw.allele[(size_t)t * tw.words], &al[(size_t)i * tw.words], tw.words * 8);
        }
    }
    tw.auto_low_polymorphism = tw.n > 0 && (double)without / (double)tw.n > 0.75;   // main.rs:539-543
    return tw;
}

static bool cluster_less(const std::vector<u32>& a, const std::vector<u32>& b) {   // (len desc, first asc)
    if (a.size() != b.size()) return a.size() > b.size();
    const u32 fa = a.empty() ? 0 : a[0], fb = b.empty() ? 0 : b[0];
    return fa < fb;
}

// ==================================================================================================
// Stage 2: asv_cluster::cluster_reads_by_kmers (src/asv_cluster.rs:72-249).
// The loop is order-dependent, so it stays sequential on the host -- but the expensive part of every
// iteration, the minimizer-set similarity of the read against its LSH candidates (:131-143, a 135x135
// `Vec::contains` scan per candidate), is batched: candidates of a BLOCK of reads are resolved by one
// svt_minimizer_shared_counts call.  A read whose candidate list could be changed by a representative
// created earlier in the same block is re-queued ("dirty"), so the result is exactly the sequential one.
// ==================================================================================================
std::vector<std::vector<u32>> cluster_reads_by_kmers(const ReadSet& rs, const TwinReads& tw, const ClusterArgs& args) {
    const u32 n = tw.n, k = args.kmer_size;
    const double threshold = args.primary_clustering_threshold;
    const size_t top_n = 10;                                                   // :84
    std::vector<std::unordered_map<u64, std::vector<u32>>> buckets(SVT_LSH_TABLES);
    std::vector<u32> assign(n);
    size_t pos = 0, B = 64;
    std::vector<std::vector<u32>> check;
    std::vector<u32> pa, pb, shared, scratch;
    std::vector<size_t> poff;
    while (pos < n) {
        const size_t end = std::min<size_t>(n, pos + B), nb = end - pos;
        check.assign(nb, {}); pa.clear(); pb.clear(); poff.assign(nb + 1, 0);
        for (size_t r = pos; r < end; r++) {                                   // query_read_against_bucket_index :303-337
            scratch.clear();
            if (tw.lsh_valid[r])
                for (u32 t = 0; t < SVT_LSH_TABLES; t++) {
                    auto it = buckets[t].find(tw.lsh[r * SVT_LSH_TABLES + t]);
                    if (it != buckets[t].end()) scratch.insert(scratch.end(), it->second.begin(), it->second.end());
                }
            std::vector<u32>& ck = check[r - pos];
            if (!scratch.empty()) {
                std::sort(scratch.begin(), scratch.end());
                std::vector<std::pair<u32, u32>> cands;                        // (hits, id)
                for (size_t i = 0; i < scratch.size();) { size_t j = i; while (j < scratch.size() && scratch[j] == scratch[i]) j++; cands.push_back({(u32)(j - i), scratch[i]}); i = j; }
                std::sort(cands.begin(), cands.end(), [](const auto& a, const auto& b) { return a > b; });   // :111 (hits desc, id desc)
                const u32 max_hits = cands[0].first;
                for (auto& c : cands) { if (c.first == max_hits || ck.size() < top_n) ck.push_back(c.second); else break; }   // :118-125
            }
            poff[r - pos] = pa.size();
            for (u32 c : ck) { pa.push_back(tw.orig[r]); pb.push_back(tw.orig[c]); }
        }
        poff[nb] = pa.size();
        shared.assign(pa.size(), 0);
        if (!pa.empty()) chk(rs.ctx, svt_minimizer_shared_counts(rs.ctx, rs.batch, rs.batch, pa.data(), pb.data(), pa.size(), shared.data(), nullptr), "svt_minimizer_shared_counts");
        std::vector<char> dirty(nb, 0);
        size_t r = pos;
        for (; r < end; r++) {
            if (dirty[r - pos]) break;
            double best_sim = 0.0; int best = -1;
            const std::vector<u32>& ck = check[r - pos];
            for (size_t j = 0; j < ck.size(); j++) {
                const u32 count = shared[poff[r - pos] + j];
                const double ratio = (double)count / (double)std::max(tw.n_unique[r], tw.n_mini[ck[j]]);     // :143
                const double sim = std::pow(ratio, 1.0 / (double)k);                                          // :144
                if (sim > best_sim) { best_sim = sim; best = (int)ck[j]; }
            }
            if (best >= 0 && best_sim > threshold) assign[r] = (u32)best;                                     // :152
            else {                                                                                            // :176-186 new representative
                assign[r] = (u32)r;
                if (tw.lsh_valid[r]) {
                    for (u32 t = 0; t < SVT_LSH_TABLES; t++) buckets[t][tw.lsh[r * SVT_LSH_TABLES + t]].push_back((u32)r);
                    for (size_t r2 = r + 1; r2 < end; r2++) {
                        if (dirty[r2 - pos] || !tw.lsh_valid[r2]) continue;
                        for (u32 t = 0; t < SVT_LSH_TABLES; t++)
                            if (tw.lsh[r2 * SVT_LSH_TABLES + t] == tw.lsh[r * SVT_LSH_TABLES + t]) { dirty[r2 - pos] = 1; break; }
                    }
                }
            }
        }
        const size_t resolved = r - pos;
        pos = r;
        if (resolved == nb) B = std::min<size_t>(B * 2, 16384); else B = std::max<size_t>(16, std::min<size_t>(B, resolved * 2 + 16));
    }
    std::map<u32, std::vector<u32>> cm;
    for (u32 r = 0; r < n; r++) cm[assign[r]].push_back(r);                   // members ascending (:216-218)
    std::vector<std::vector<u32>> clusters;
    for (auto& kv : cm) clusters.push_back(std::move(kv.second));
    std::stable_sort(clusters.begin(), clusters.end(), cluster_less);         // :212 (equal sizes: smaller first member)
    std::vector<std::vector<u32>> kept;
    for (auto& c : clusters) if (c.size() >= args.min_cluster_size) kept.push_back(std::move(c));   // :221
    return kept;
}

// ==================================================================================================
// SNPmer consensus algebra on bitsets (src/asv_cluster.rs:840-1003).  A consensus is (presence, allele);
// position / count fields of ConsensusPoly only order the list and are never compared (:892, :968-994).
// ==================================================================================================
struct Bits { std::vector<u64> p, a; };
static inline u32 popc(u64 x) { return (u32)__builtin_popcountll(x); }

static void build_consensus(const TwinReads& tw, const std::vector<u32>& cluster, Bits& out, std::vector<u32>& c0, std::vector<u32>& c1) {
    const u32 W = tw.words;
    c0.assign((size_t)W * 64, 0); c1.assign((size_t)W * 64, 0);
    for (u32 rid : cluster) {
        const u64* p = &tw.p_filt[(size_t)rid * W]; const u64* a = &tw.allele[(size_t)rid * W];
        for (u32 w = 0; w < W; w++) {
            u64 bits = p[w];
            while (bits) { const u32 b = (u32)__builtin_ctzll(bits); bits &= bits - 1; if ((a[w] >> b) & 1) c1[w * 64 + b]++; else c0[w * 64 + b]++; }
        }
    }
    out.p.assign(W, 0); out.a.assign(W, 0);
    const u32 thr = std::max<u32>(1, (u32)(cluster.size() / 6));              // :878
    for (u32 s = 0; s < W * 64; s++) {
        const bool one = c1[s] > c0[s];                                        // tie -> allele 0 = smaller mid base
        const u32 best = one ? c1[s] : c0[s];
        if (best >= thr && best > 0) { out.p[s >> 6] |= 1ull << (s & 63); if (one) out.a[s >> 6] |= 1ull << (s & 63); }
    }
}
static void compare_consensus(const Bits& x, const Bits& y, u32& m, u32& mm) {   // :968-994
    m = mm = 0;
    for (size_t w = 0; w < x.p.size(); w++) { const u64 both = x.p[w] & y.p[w], d = x.a[w] ^ y.a[w]; m += popc(both & ~d); mm += popc(both & d); }
}
static u32 cons_len(const Bits& x) { u32 n = 0; for (u64 w : x.p) n += popc(w); return n; }
static bool concordant(const Bits& x, const Bits& y) {                           // :997-1003
    u32 m, mm; compare_consensus(x, y, m, mm);
    return mm == 0 && m >= std::min(cons_len(x), std::max<u32>(cons_len(y), 2));
}

// recluster_one_round_top_n (top_n = None), src/asv_cluster.rs:1146-1270
static void recluster_one_round(const TwinReads& tw, std::vector<std::vector<u32>>& clusters, u32& num_merges) {
    struct Item { std::vector<u32> members; Bits cons; };
    std::vector<Item> all; std::vector<u32> c0, c1;
    for (auto& cl : clusters) { if (cl.empty()) continue; Item it; it.members = cl; build_consensus(tw, cl, it.cons, c0, c1); all.push_back(std::move(it)); }
    std::stable_sort(all.begin(), all.end(), [](const Item& a, const Item& b) { return cluster_less(a.members, b.members); });   // :1170
    std::vector<char> merged(all.size(), 0);
    std::vector<std::vector<u32>> out;
    num_merges = 0;
    for (size_t i = 0; i < all.size(); i++) {
        if (merged[i]) continue;
        for (size_t j = i + 1; j < all.size(); j++) {
            if (merged[j]) continue;
            const Bits& ci = all[i].cons; const Bits& cj = all[j].cons;        // consensus of i is NOT rebuilt inside the j loop (:1201-1202)
            bool conc = concordant(ci, cj) && concordant(cj, ci);
            u32 m, mm; compare_consensus(ci, cj, m, mm);
            const size_t li = all[i].members.size(), lj = all[j].members.size();
            const size_t max_len = std::max(li, lj), min_len = std::min(li, lj);
            if (mm == 0 && (double)m > (double)std::min(cons_len(ci), cons_len(cj)) * 0.975 && max_len / min_len > 50) conc = true;   // :1212-1215
            if (mm == 0 && max_len / min_len > 500 && min_len <= 2) conc = true;                                                    // :1220
            if (conc) { all[i].members.insert(all[i].members.end(), all[j].members.begin(), all[j].members.end()); merged[j] = 1; num_merges++; }
        }
        out.push_back(all[i].members);
    }
    std::stable_sort(out.begin(), out.end(), cluster_less);                    // :1266
    clusters.swap(out);
}

// reassign_reads_to_best_cluster, src/asv_cluster.rs:1007-1130: reads x cluster consensuses is one GPU tile
static void reassign_reads(const ReadSet& rs, const TwinReads& tw, std::vector<std::vector<u32>>& clusters, const ClusterArgs& args) {
    const size_t nc = clusters.size();
    if (nc == 0) return;
    const u32 W = tw.words;
    std::vector<u64> cp(nc * W), ca(nc * W); std::vector<u32> c0, c1; Bits b;
    for (size_t i = 0; i < nc; i++) { build_consensus(tw, clusters[i], b, c0, c1); if (W) { memcpy(&cp[i * W], b.p.data(), W * 8); memcpy(&ca[i * W], b.a.data(), W * 8); } }
    std::vector<u32> rows, twin_of;
    for (auto& cl : clusters) for (u32 r : cl) { rows.push_back(tw.orig[r]); twin_of.push_back(r); }
    std::vector<u32> best(rows.size(), 0);
    if (W && !rows.empty()) {
        svt_bitset* S = nullptr;
        chk(rs.ctx, svt_bitset_upload(rs.ctx, cp.data(), ca.data(), (u32)nc, &S), "svt_bitset_upload");
        int rc = svt_snpmer_best_column(rs.ctx, rs.batch, SVT_VIEW_FILTERED, rows.data(), (u32)rows.size(), S, best.data(), nullptr);
        svt_bitset_free(rs.ctx, S);
        chk(rs.ctx, rc, "svt_snpmer_best_column");
    }
    std::vector<std::vector<u32>> out(nc);
    for (size_t i = 0; i < rows.size(); i++) out[best[i]].push_back(twin_of[i]);
    std::vector<std::vector<u32>> kept;
    for (auto& cl : out) if (!cl.empty() && cl.size() >= args.min_cluster_size) { std::sort(cl.begin(), cl.end()); kept.push_back(std::move(cl)); }   // :1121-1124
    clusters.swap(kept);
}

// ==================================================================================================
// Stage 3: asv_cluster::cluster_reads_by_snpmers (src/asv_cluster.rs:561-795) + recluster (:1272-1433).
// Greedy per k-mer cluster.  Per block of reads ONE tile call returns, for every read, the compatible
// columns (mismatches == 0, matches > 0, :481-483) among (a) the representatives that exist at block
// start and (b) the EARLIER reads of the same block (any of which may have become a representative by the
// time the read is decided) -- so the sequential decision is exact without re-running anything.
// ==================================================================================================
std::vector<std::vector<u32>> cluster_reads_by_snpmers(const ReadSet& rs, const TwinReads& tw, const std::vector<std::vector<u32>>& kmer_clusters,
                                                       const ClusterArgs& args, std::vector<std::vector<u32>>* pre, std::vector<u32>* pre_group) {
    if (args.low_polymorphism) {                                               // :570-580
        std::vector<std::vector<u32>> cl;
        for (auto& c : kmer_clusters) if (c.size() >= args.min_cluster_size) cl.push_back(c);
        std::stable_sort(cl.begin(), cl.end(), cluster_less);
        return cl;
    }
    std::map<u32, std::vector<std::vector<u32>>> groups;
    std::vector<u32> o_row, o_col, o_mm, cols, rows, cnt;
    for (u32 g = 0; g < kmer_clusters.size(); g++) {
        const std::vector<u32>& kc = kmer_clusters[g];
        if (kc.empty()) continue;
        std::vector<u32> reps;                              // twin ids, in creation order (= `representatives`, :606)
        std::unordered_map<u32, u32> rep_pos, rep_size;     // twin id -> index in reps / current size
        std::unordered_map<u32, u32> assign;
        size_t pos = 0;
        const size_t B = 1024;
        while (pos < kc.size()) {
            const size_t end = std::min(kc.size(), pos + B), nb = end - pos;
            const u32 R = (u32)reps.size();
            rows.resize(nb); cols.resize(R + nb);
            for (u32 i = 0; i < R; i++) cols[i] = tw.orig[reps[i]];
            for (size_t i = 0; i < nb; i++) { rows[i] = tw.orig[kc[pos + i]]; cols[R + i] = rows[i]; }
            u64 n_out = 0, cap = std::max<u64>(4096, (u64)nb * 64);
            while (true) {
                o_row.resize(cap); o_col.resize(cap); o_mm.resize(cap);
                int rc = svt_snpmer_compat_lists(rs.ctx, rs.batch, SVT_VIEW_ALL, rows.data(), (u32)nb, rs.batch, SVT_VIEW_ALL, nullptr, cols.data(), (u32)cols.size(),
                                                 SVT_LIST_COMPATIBLE, 1, R, o_row.data(), o_col.data(), o_mm.data(), cap, &n_out);
                if (rc == SVT_ERR_OVERFLOW) { cap = n_out + 1024; continue; }
                chk(rs.ctx, rc, "svt_snpmer_compat_lists");
                break;
            }
            // bucket the triples by row (counting sort)
            cnt.assign(nb + 1, 0);
            for (u64 i = 0; i < n_out; i++) cnt[o_row[i] + 1]++;
            for (size_t i = 0; i < nb; i++) cnt[i + 1] += cnt[i];
            std::vector<std::pair<u32, u32>> lst(n_out);    // (col, matches)
            { std::vector<u32> fill(cnt.begin(), cnt.end() - 1); for (u64 i = 0; i < n_out; i++) lst[fill[o_row[i]]++] = {o_col[i], o_mm[i] >> 16}; }
            for (size_t i = 0; i < nb; i++) {
                const u32 rid = kc[pos + i];
                const bool iterative = reps.size() > 1000;                     // :615
                int best = -1;
                if (iterative) {                                               // :413-464; find_any -> FIRST compatible representative
                    u32 best_pos = ~0u;
                    for (u32 j = cnt[i]; j < cnt[i + 1]; j++) {
                        const u32 col = lst[j].first; u32 p;
                        if (col < R) p = col;
                        else { auto it = rep_pos.find(kc[pos + (col - R)]); if (it == rep_pos.end()) continue; p = it->second; }
                        if (p < best_pos) best_pos = p;
                    }
                    if (best_pos != ~0u) best = (int)reps[best_pos];
                } else {                                                       // :467-510
                    std::array<int64_t, 3> bk{0, 0, 0}; bool have = false;
                    for (u32 j = cnt[i]; j < cnt[i + 1]; j++) {
                        const u32 col = lst[j].first; u32 cand;
                        if (col < R) cand = reps[col];
                        else { cand = kc[pos + (col - R)]; if (!rep_pos.count(cand)) continue; }
                        std::array<int64_t, 3> key{-(int64_t)lst[j].second, (int64_t)rep_size[cand], (int64_t)cand};   // :494
                        if (!have || key < bk) { bk = key; have = true; }
                    }
                    if (have) best = (int)bk[2];
                }
                if (best >= 0) { assign[rid] = (u32)best; rep_size[(u32)best] += 1; }       // :386-394
                else { rep_pos[rid] = (u32)reps.size(); reps.push_back(rid); assign[rid] = rid; rep_size[rid] = 1; }   // :397-410
            }
            pos = end;
        }
        std::map<u32, std::vector<u32>> cm;
        for (auto& kv : assign) cm[kv.second].push_back(kv.first);
        std::vector<std::vector<u32>> local;
        for (auto& kv : cm) { std::sort(kv.second.begin(), kv.second.end()); local.push_back(std::move(kv.second)); }
        std::stable_sort(local.begin(), local.end(), cluster_less);            // :687
        std::vector<std::vector<u32>> kept;
        for (auto& cl : local) if (cl.size() >= args.min_cluster_size) kept.push_back(std::move(cl));   // :692
        groups[g] = std::move(kept);
    }
    if (pre) { pre->clear(); if (pre_group) pre_group->clear(); for (auto& kv : groups) for (auto& cl : kv.second) { pre->push_back(cl); if (pre_group) pre_group->push_back(kv.first); } }
    // recluster_using_consensus_reps :1272-1433
    u32 iteration = 0;
    while (true) {
        if (iteration >= args.max_iterations_recluster) break;                // :1296
        iteration++;
        u32 total_merges = 0;
        std::map<u32, std::vector<std::vector<u32>>> next;
        for (auto& kv : groups) {
            std::vector<std::vector<u32>> cl = kv.second; u32 merges = 0;
            recluster_one_round(tw, cl, merges);
            total_merges += merges;
            reassign_reads(rs, tw, cl, args);
            if (!cl.empty()) next[kv.first] = std::move(cl);                   // :1339
        }
        groups.swap(next);
        if (total_merges == 0) break;                                          // :1367
    }
    std::vector<std::vector<u32>> fin;
    for (auto& kv : groups) for (auto& cl : kv.second) if (!cl.empty()) fin.push_back(cl);
    std::stable_sort(fin.begin(), fin.end(), cluster_less);                    // :1387
    std::vector<std::vector<u32>> kept;
    for (auto& cl : fin) if (cl.size() >= args.min_cluster_size) kept.push_back(std::move(cl));
    return kept;
}

// ==================================================================================================
// Stage 7: alignment::refine_asv_depths_with_em (src/alignment.rs:1723-2039), SNPmer path.
// Three GPU passes over ALL reads (the reference: rayon par_iter with a minimap2 index build per read):
//   K6 overlap lists reads x ASVs -> K7 minimizer intersections of the candidate pairs -> K8 nm of the ties.
// ==================================================================================================
static u32 band_for(const ClusterArgs& args, u32 n, u32 m) {
    if (args.align_band) return args.align_band;
    const u32 mx = std::max(n, m), df = n > m ? n - m : m - n;
    return std::min<u32>(std::max((mx + 9) / 10, df), 511);
}

static void run_em(const std::map<std::vector<u32>, u64>& eq, u64 total_assigned, size_t n_asv, std::vector<double>& ab) {   // :1957-2009
    ab.assign(n_asv, 1.0 / (double)n_asv);
    const double thr = 0.01 / (double)total_assigned;
    u32 iter = 0;
    while (true) {
        iter++;
        std::vector<double> nw(n_asv, 0.0);
        for (auto& kv : eq) {                                                  // sorted class order (the reference iterates a RandomState HashMap)
            double den = 0.0; for (u32 a : kv.first) den += ab[a];
            if (den > 0.0) for (u32 a : kv.first) nw[a] += (double)kv.second * ab[a] / den;
        }
        double tot = 0.0; for (double x : nw) tot += x;
        if (tot > 0.0) for (double& x : nw) x /= (double)total_assigned;
        double mx = 0.0; for (size_t i = 0; i < n_asv; i++) mx = std::max(mx, std::fabs(ab[i] - nw[i]));
        ab.swap(nw);
        if (mx < thr || iter >= 10000) break;
    }
}

EmResult refine_asv_depths_with_em(const ReadSet& rs, const TwinReads& tw, svt_batch* asvs, const std::vector<u64>& asv_off, const ClusterArgs& args) {
    EmResult em;
    const size_t na = asv_off.size() - 1, nr = tw.n;
    const u32 k = args.kmer_size;
    em.depth.assign(na, 0); em.unambig.assign(na, 0); em.ambig.assign(na, 0); em.leq10.assign(na, 0);
    em.read_n_best.assign(nr, 0); em.read_first.assign(nr, 0); em.read_nm.assign(nr, -1); em.read_class.assign(nr, {});
    if (na == 0 || nr == 0) { em.kept_original = true; return em; }
    // ASV twin reads: kmer_comp::twin_reads_from_fasta (src/kmer_comp.rs:39-66): qualities None, no filtering
    chk(rs.ctx, svt_extract_seeds(rs.ctx, asvs, k, args.c, args.minimum_base_quality, 0), "svt_extract_seeds(asvs)");
    std::vector<u32> asv_unique(na);
    { svt_seeds_out o; memset(&o, 0, sizeof o); o.n_unique = asv_unique.data(); chk(rs.ctx, svt_seeds_fetch(rs.ctx, asvs, &o), "svt_seeds_fetch(asvs)"); }
    // K6: candidates = ASVs sharing >= 1 SNPmer site with the read (find_compatible_candidates keys, :1791)
    std::vector<u32> rows(nr), cols(na);
    for (size_t i = 0; i < nr; i++) rows[i] = tw.orig[i];
    for (size_t i = 0; i < na; i++) cols[i] = (u32)i;
    std::vector<u32> o_row, o_col, o_mm;
    u64 n_out = 0, cap = std::max<u64>(4096, (u64)nr * 16);
    while (true) {
        o_row.resize(cap); o_col.resize(cap); o_mm.resize(cap);
        int rc = svt_snpmer_compat_lists(rs.ctx, rs.batch, SVT_VIEW_ALL, rows.data(), (u32)nr, asvs, SVT_VIEW_ALL, nullptr, cols.data(), (u32)na,
                                         SVT_LIST_OVERLAP, 0, 0, o_row.data(), o_col.data(), o_mm.data(), cap, &n_out);
        if (rc == SVT_ERR_OVERFLOW) { cap = n_out + 1024; continue; }
        chk(rs.ctx, rc, "svt_snpmer_compat_lists(stage7)");
        break;
    }
    // K7 on every candidate pair
    std::vector<u32> pa(n_out), pb(n_out), shared(n_out), same(n_out);
    for (u64 i = 0; i < n_out; i++) { pa[i] = tw.orig[o_row[i]]; pb[i] = o_col[i]; }
    if (n_out) chk(rs.ctx, svt_minimizer_shared_counts(rs.ctx, rs.batch, asvs, pa.data(), pb.data(), n_out, shared.data(), same.data()), "svt_minimizer_shared_counts(stage7)");
    // group by read, ascending ASV inside a read (deterministic stand-in for FxHashMap iteration order; only ties it could
    // affect are removed by the sort at :1892)
    std::vector<u64> ord(n_out);
    for (u64 i = 0; i < n_out; i++) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](u64 a, u64 b) { return o_row[a] != o_row[b] ? o_row[a] < o_row[b] : o_col[a] < o_col[b]; });
    const double minfrac = std::pow(0.950, (int)k);                           // powi :1806
    struct Tie { u32 read, asv; u8 rev; u32 band; };
    std::vector<Tie> ties; std::vector<size_t> tie_off(nr + 1, 0);
    size_t p = 0;
    for (size_t r = 0; r < nr; r++) {
        tie_off[r] = ties.size();
        std::vector<std::array<u32, 3>> best;     // (asv, mismatches, index)
        for (; p < n_out && o_row[ord[p]] == r; p++) {
            const u64 i = ord[p];
            const u32 mm = shared[i], mism = o_mm[i] & 0xFFFF, asv = o_col[i];
            if (mm == 0) continue;                                                                           // :1801
            if ((double)mm / (double)std::min(tw.n_unique[r], asv_unique[asv]) < minfrac) continue;          // :1805-1808
            const double ratio = (double)mism / (double)mm / (double)args.c;                                 // :1811
            if (ratio <= 0.0050) best.push_back({asv, mism, (u32)i});                                        // :1829-1833
        }
        if (best.empty()) continue;
        u32 lowest = ~0u; for (auto& b : best) lowest = std::min(lowest, b[1]);                              // :1841-1843
        for (auto& b : best) if (b[1] == lowest) {
            const u64 i = b[2];
            const bool rev = (shared[i] - same[i]) > same[i];                 // strand vote (K8 contract)
            const u32 la = (u32)(asv_off[b[0] + 1] - asv_off[b[0]]);
            ties.push_back({(u32)r, b[0], (u8)rev, band_for(args, la, tw.length[r])});
        }
    }
    tie_off[nr] = ties.size();
    // K8
    std::vector<u32> qi(ties.size()), ti(ties.size()), band(ties.size()); std::vector<u8> rev(ties.size()); std::vector<int32_t> nm(ties.size());
    for (size_t i = 0; i < ties.size(); i++) { qi[i] = ties[i].asv; ti[i] = tw.orig[ties[i].read]; rev[i] = ties[i].rev; band[i] = ties[i].band; }
    if (!ties.empty()) chk(rs.ctx, svt_align_nm(rs.ctx, asvs, rs.batch, qi.data(), ti.data(), rev.data(), band.data(), ties.size(), nm.data()), "svt_align_nm");
    std::map<std::vector<u32>, u64> eq;
    for (size_t r = 0; r < nr; r++) {
        int32_t best_nm = INT32_MAX;
        for (size_t i = tie_off[r]; i < tie_off[r + 1]; i++) if (nm[i] != INT32_MAX) best_nm = std::min(best_nm, nm[i]);   // empty mapping -> skipped (:1859-1861)
        std::vector<u32> cls;
        for (size_t i = tie_off[r]; i < tie_off[r + 1]; i++) if (nm[i] != INT32_MAX && nm[i] == best_nm) cls.push_back(ties[i].asv);
        if (cls.empty()) { em.filtered++; continue; }                                                         // :1817-1837, :1921-1924
        std::sort(cls.begin(), cls.end());                                                                    // :1892
        if (cls.size() == 1) em.unambig[cls[0]]++; else for (u32 a : cls) em.ambig[a]++;                    // :1898-1908
        if (best_nm <= 10) for (u32 a : cls) em.leq10[a]++;                                                   // :1910-1915
        eq[cls]++; em.total_assigned++;
        em.read_n_best[r] = (u32)cls.size(); em.read_first[r] = cls[0]; em.read_nm[r] = best_nm; em.read_class[r] = cls;
    }
    if (eq.empty()) { em.kept_original = true; return em; }                                                   // :1952-1955
    std::vector<double> ab; run_em(eq, em.total_assigned, na, ab);
    for (size_t a = 0; a < na; a++) em.depth[a] = (u64)std::llround(ab[a] * (double)em.total_assigned);       // :2015
    return em;
}

// Stage 7b: alignment::compute_per_sample_depths (src/alignment.rs:2044-2215).  The per-read mapping is the same
// function of (read, ASV set) as in Stage 7, so the reference's per-sample recomputation collapses to an EM per sample.
std::vector<std::vector<u64>> compute_per_sample_depths(const TwinReads& tw, const EmResult& em, u32 n_samples, size_t n_asv) {
    std::vector<std::vector<u64>> res(n_asv, std::vector<u64>(n_samples, 0));
    for (u32 s = 0; s < n_samples; s++) {
        std::map<std::vector<u32>, u64> eq; u64 total = 0;
        for (u32 i = 0; i < tw.n; i++) if (tw.file_idx[i] == s && !em.read_class[i].empty()) { eq[em.read_class[i]]++; total++; }
        if (eq.empty() || total == 0) continue;                               // :2179-2181
        std::vector<double> ab; run_em(eq, total, n_asv, ab);
        for (size_t a = 0; a < n_asv; a++) res[a][s] = (u64)std::llround(ab[a] * (double)total);             // :2209-2211
    }
    return res;
}

}  // namespace savont
