// asv_pipeline.cpp -- host logic of the `savont asv` hot path above the C-ABI (see asv_pipeline.hpp).
// Reference citations are file:line relative to the reference root (bluenote-1577/savont v0.6.4).
#include "asv_pipeline.hpp"
#include "worker_pool.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>

#include "stats.hpp"

#include <ctime>
namespace savont {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;
typedef uint16_t u16;

// env-gated host tracer (SAVONT_TRACE=1): accumulates wall time per label, dumped by trace_dump()
namespace {
struct TraceAcc { double s = 0; u64 n = 0; double cpu = 0; };   // cpu: process CPU seconds between the label's start and end (all threads: attributable when ONE sample is in flight)
inline double process_cpu_seconds() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
std::map<std::string, TraceAcc>& trace_map() { static std::map<std::string, TraceAcc> m; return m; }
std::mutex& trace_mutex() { static std::mutex m; return m; }
bool trace_on() { static int on = -1; if (on < 0) { const char* e = getenv("SAVONT_TRACE"); on = (e && *e == '1') ? 1 : 0; } return on == 1; }
struct Trace {
    const char* name; std::chrono::steady_clock::time_point t0; double c0 = 0; bool on;
    explicit Trace(const char* n) : name(n), on(trace_on()) { if (on) { t0 = std::chrono::steady_clock::now(); c0 = process_cpu_seconds(); } }
    ~Trace() { if (on) { const double d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), dc = process_cpu_seconds() - c0; std::lock_guard<std::mutex> l(trace_mutex()); auto& a = trace_map()[name]; a.s += d; a.n++; a.cpu += dc; } }
};
}  // namespace
bool trace_enabled() { return trace_on(); }
void trace_add(const char* name, double seconds, double cpu_seconds) { if (!trace_on()) return; std::lock_guard<std::mutex> l(trace_mutex()); auto& a = trace_map()[name]; a.s += seconds; a.n++; a.cpu += cpu_seconds; }
double trace_cpu_now() { return process_cpu_seconds(); }
void trace_dump() {
    if (!trace_on()) return;
    for (auto& kv : trace_map()) fprintf(stderr, "[savont-trace] %-36s %9.3f ms  x%llu  cpu %9.3f ms\n", kv.first.c_str(), kv.second.s * 1e3, (unsigned long long)kv.second.n, kv.second.cpu * 1e3);
    trace_map().clear();
}

static void chk(svt_ctx* c, int rc, const char* what) {
    if (rc != SVT_OK) throw Error{rc, std::string(what) + ": " + svt_last_error(c)};
}

// host worker threads for the embarrassingly parallel host loops (the reference uses its rayon pool for the same loops)
static unsigned host_threads() {
    static unsigned t = 0;
    if (!t) { const char* e = getenv("SAVONT_THREADS"); t = e ? (unsigned)atoi(e) : std::min(32u, std::max(1u, std::thread::hardware_concurrency())); if (!t) t = 1; }
    return t;
}
template <class F> static void parallel_ranges(size_t n, size_t min_chunk, F f) {      // f(chunk_index, lo, hi) on the persistent worker pool
    size_t T = std::min<size_t>(std::min<size_t>(host_threads(), WorkerPool::get().size()), std::max<size_t>(1, n / std::max<size_t>(1, min_chunk)));
    if (T <= 1) { f(0, 0, n); return; }
    const size_t per = (n + T - 1) / T;
    par_for(T, [&](size_t t) { const size_t lo = t * per, hi = std::min(n, lo + per); if (lo < hi) f(t, lo, hi); });
}

// stable sort on the worker pool: sorted chunks, then pairwise stable merges (the result equals std::stable_sort)
template <class T, class Cmp> static void parallel_stable_sort(std::vector<T>& v, Cmp cmp) {
    const size_t n = v.size();
    size_t parts = std::min<size_t>(std::min<size_t>(host_threads(), WorkerPool::get().size()), n / 8192);
    if (parts < 2) { std::stable_sort(v.begin(), v.end(), cmp); return; }
    size_t p2 = 1; while (p2 * 2 <= parts) p2 *= 2;                             // power of two: clean merge tree
    std::vector<size_t> cut(p2 + 1);
    for (size_t i = 0; i <= p2; i++) cut[i] = n * i / p2;
    par_for(p2, [&](size_t i) { std::stable_sort(v.begin() + cut[i], v.begin() + cut[i + 1], cmp); });
    for (size_t w = 1; w < p2; w *= 2)
        par_for(p2 / (2 * w), [&](size_t i) { std::inplace_merge(v.begin() + cut[2 * w * i], v.begin() + cut[2 * w * i + w], v.begin() + cut[2 * w * i + 2 * w], cmp); });
}

// ==================================================================================================
// Stage 1a: seq_parse::read_to_split_kmers (src/seq_parse.rs:12-78).  The three-level thread pipeline
// of :316-497 (reader -> split_kmer_mid workers -> kmer%threads hash-map shards) is ONE fused GPU pass.
// ==================================================================================================
void count_split_kmers_device(const ReadSet& rs, const ClusterArgs& args, u64* n_distinct, u64* n_kept) {
    u64 nd = 0, nk = 0;
    Trace t_("1a.count.total");
    chk(rs.ctx, svt_count_split_kmers(rs.ctx, rs.batch, args.kmer_size, args.minimum_base_quality,
                                      rs.rc_flags.empty() ? nullptr : rs.rc_flags.data(), args.single_strand ? 1 : 0, &nd, &nk),
        "svt_count_split_kmers");
    if (n_distinct) *n_distinct = nd;
    if (n_kept) *n_kept = nk;
    if (nk < nd / 1000)                                                        // :69-72 (process::exit(1) in the reference)
        throw Error{1, "Less than 0.1% of SNPmers have counts > 1 in both strands and > 2 multiplicity. Consider --single-strand"};
}
// the B1 return value: the sorted table copied out of HBM (tests, callers that want all of it; the pipeline itself does not)
KmerCountTable fetch_count_table(const ReadSet& rs, u64 n_kept) {
    std::vector<u64> km(n_kept); std::vector<u32> rev(n_kept), fwd(n_kept);
    chk(rs.ctx, svt_count_fetch(rs.ctx, km.data(), rev.data(), fwd.data()), "svt_count_fetch");
    KmerCountTable t(n_kept);
    for (u64 i = 0; i < n_kept; i++) t[i] = {km[i], {rev[i], fwd[i]}};
    return t;
}
KmerCountTable read_to_split_kmers(const ReadSet& rs, const ClusterArgs& args, u64* n_distinct) {
    u64 nk = 0;
    count_split_kmers_device(rs, args, n_distinct, &nk);
    return fetch_count_table(rs, nk);
}

// ==================================================================================================
// Stage 1b: kmer_comp::get_snpmers_inplace_sort (src/kmer_comp.rs:454-642).  Of the sorted table only two short selections
// matter: the entries in groups of >= 2 alleles (the statistics of :543-623) and the entries with total > 100 (the order
// statistic of :474 and the high-frequency list of :494-496).  The device makes both selections (svt_count_candidates_*); for a
// table that is already on the host (multi-GPU merge done elsewhere, tests) candidates_from_table makes the same two.
// ==================================================================================================
SnpCandidates candidates_from_device(const ReadSet& rs) {
    SnpCandidates c; u64 ng = 0, nh = 0;
    chk(rs.ctx, svt_count_candidates_sizes(rs.ctx, &c.n_table, &ng, &nh), "svt_count_candidates_sizes");
    c.g_kmer.resize(ng); c.g_rev.resize(ng); c.g_fwd.resize(ng); c.h_kmer.resize(nh); c.h_rev.resize(nh); c.h_fwd.resize(nh);
    chk(rs.ctx, svt_count_candidates_fetch(rs.ctx, c.g_kmer.data(), c.g_rev.data(), c.g_fwd.data(), c.h_kmer.data(), c.h_rev.data(), c.h_fwd.data()), "svt_count_candidates_fetch");
    return c;
}
SnpCandidates candidates_from_table(const KmerCountTable& table, u32 k, const ClusterArgs& args) {
    SnpCandidates c; c.n_table = table.size();
    const u64 sm = 3ull << (k - 1);
    const size_t n = table.size();
    size_t run_start = 0, run_len = 0; u64 cur = ~0ull;
    std::vector<size_t> run;                                                    // indices of the current group's members
    auto flush = [&]() { if (run.size() >= 2) for (size_t i : run) { c.g_kmer.push_back(table[i].first); c.g_rev.push_back(table[i].second.first); c.g_fwd.push_back(table[i].second.second); } run.clear(); };
    (void)run_start; (void)run_len;
    for (size_t i = 0; i < n; i++) {
        const u32 c0 = table[i].second.first, c1 = table[i].second.second;
        if ((u64)c0 + c1 > 100) { c.h_kmer.push_back(table[i].first); c.h_rev.push_back(c0); c.h_fwd.push_back(c1); }
        if (!args.single_strand && (c0 == 0 || c1 == 0)) continue;              // :498-502 (a no-op on a B1 table: its filter already requires both)
        const u64 split = table[i].first & ~sm;
        if (split != cur) { flush(); cur = split; }
        run.push_back(i);
    }
    flush();
    return c;
}
KmerGlobalInfo get_snpmers_inplace_sort(const KmerCountTable& table, u32 k, const ClusterArgs& args) {
    if (table.empty()) throw Error{1, "No k-mers found. Exiting."};           // :469-472
    return snpmers_from_candidates(candidates_from_table(table, k, args), k, args);
}
KmerGlobalInfo snpmers_from_candidates(const SnpCandidates& cand, u32 k, const ClusterArgs& args) {
    KmerGlobalInfo info;
    const size_t n = cand.n_table;
    if (n == 0) throw Error{1, "No k-mers found. Exiting."};                  // :469-472
    Trace t_all("1b.total");
    // :468-474: ONE order statistic of the sorted totals, the q-th largest with q = n/100000 + 1.  Totals <= 100 cannot raise
    // max(., 100), so the entries with total > 100 decide it.
    const size_t q = n / 100000 + 1, nh = cand.h_kmer.size();
    u32 thresh = 100;
    if (nh >= q) {
        std::vector<u32> tot(nh);
        for (size_t i = 0; i < nh; i++) tot[i] = cand.h_rev[i] + cand.h_fwd[i];
        std::nth_element(tot.begin(), tot.begin() + (q - 1), tot.end(), std::greater<u32>());
        thresh = std::max<u32>(tot[q - 1], 100);                              // :474
    }
    info.high_freq_thresh = thresh;
    for (size_t i = 0; i < nh; i++) if (cand.h_rev[i] + cand.h_fwd[i] > thresh) info.high_freq_kmers.push_back(cand.h_kmer[i]);   // :494-496
    const u64 sm = 3ull << (k - 1);
    struct E { u64 kmer; u32 c0, c1; };
    // groups (:490-519): runs of equal masked k-mer among the candidates, every one with >= 2 alleles
    const size_t ne = cand.g_kmer.size();
    std::vector<size_t> gstart;
    u64 cur = ~0ull;
    for (size_t i = 0; i < ne; i++) { const u64 split = cand.g_kmer[i] & ~sm; if (split != cur) { gstart.push_back(i); cur = split; } }
    gstart.push_back(ne);
    const size_t ng = gstart.size() - 1;
    std::vector<SnpmerInfo> res(ng); std::vector<char> ok(ng, 0);
    // workers (:543-623): the statistics of every group, independent of each other
    parallel_ranges(ng, 256, [&](size_t, size_t lo, size_t hi) {
        E group[2];
        for (size_t g = lo; g < hi; g++) {
            // :554 sorts the group's alleles by falling total (stable) and reads the first two: the first maximum, and the first maximum of the rest -- two scans of
            // the two to four entries instead of a vector and a sort per group (there are ~10^5 groups per 100k reads, nearly all of them error alleles)
            size_t i0 = gstart[g];
            for (size_t i = gstart[g] + 1; i < gstart[g + 1]; i++) if ((u64)cand.g_rev[i] + cand.g_fwd[i] > (u64)cand.g_rev[i0] + cand.g_fwd[i0]) i0 = i;
            size_t i1 = i0 == gstart[g] ? gstart[g] + 1 : gstart[g];
            for (size_t i = gstart[g]; i < gstart[g + 1]; i++) if (i != i0 && (u64)cand.g_rev[i] + cand.g_fwd[i] > (u64)cand.g_rev[i1] + cand.g_fwd[i1]) i1 = i;
            group[0] = E{cand.g_kmer[i0], cand.g_rev[i0], cand.g_fwd[i0]}; group[1] = E{cand.g_kmer[i1], cand.g_rev[i1], cand.g_fwd[i1]};
            const u64 nn = (u64)group[0].c0 + group[0].c1, succ = (u64)group[1].c0 + group[1].c1;
            // :557-569 `binomial_test(n, succ, 0.025) > 0.05 -> not a SNPmer` (cond2 is dead: k < 5 never holds).  Nearly every group is a sequencing-error allele
            // beside the true one: succ ~ 0.3 % of n, far BELOW the mean 0.025 n of the null.  Cantelli's inequality decides those without the incomplete beta function
            // (three lgamma + a continued fraction per group: 25 ms of CPU per 100k-read step, a fifth of what a rank has at 2 CPUs): for t = mean - succ > 0,
            // P(X <= succ) <= var / (var + t^2), so 1 - cdf(succ) >= 1 - var / (var + t^2), which is > 0.05 as soon as t > 0.23 sd; the margin below (t > sd + 1: the bound
            // then says >= 0.5) keeps the screen away from anything the rounding of beta_reg could decide differently.
            { const double mean = 0.025 * (double)nn, var = mean * 0.975, t = mean - (double)succ; if (t > 1.0 && (t - 1.0) * (t - 1.0) > var) continue; }
            if (binomial_test(nn, succ, 0.025) > 0.05) continue;
            const u32 a = group[0].c0, b = group[1].c0, c = group[0].c1, d = group[1].c1;
            const u32 t0 = std::max(a, c), t1 = std::max(b, d), t2 = std::min(c, a), t3 = std::min(d, b);   // :575-578
            double odds = 0.0;
            if (!(t0 == 0 || t1 == 0 || t2 == 0 || t3 == 0)) odds = ((double)t0 * (double)t3) / ((double)t1 * (double)t2);
            if (!args.single_strand && odds == 0.0) continue;                  // :586-590
            // :593 `p_value > 0.005 || (odds < 1.5 && odds > 1/1.5)`: the p-value (:579, a hypergeometric walk of thousands of terms at
            // 100k reads) only decides when the odds ratio is outside that range -- it is not stored anywhere, so it is evaluated lazily
            if (!(odds < 1.5 && odds > 1. / 1.5) && !(fisher_two_tail(t0, t1, t2, t3) > 0.005)) continue;
            SnpmerInfo s2;
            s2.split_kmer = group[0].kmer & ~sm;
            s2.mid_bases[0] = (u8)((group[0].kmer & sm) >> (k - 1)); s2.mid_bases[1] = (u8)((group[1].kmer & sm) >> (k - 1));
            s2.counts[0] = (u32)nn; s2.counts[1] = (u32)succ; s2.k = (u8)k;
            res[g] = s2; ok[g] = 1;
        }
    });
    for (size_t g = 0; g < ng; g++) if (ok[g]) info.snpmer_info.push_back(res[g]);
    if (args.no_snpmers) info.snpmer_info.clear();                            // :525,:689 "Skipping snpmer detection": no sites, the high-frequency list stays
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
    TwinReads tw; twin_reads_from_snpmers(rs, info, args, tw); return tw;
}
// in place: the vectors of `tw` (160 B of LSH signatures per read alone) keep their storage from the previous call on the same reads
void twin_reads_from_snpmers(const ReadSet& rs, const KmerGlobalInfo& info, const ClusterArgs& args, TwinReads& tw) {
    const u32 n = rs.n, k = args.kmer_size;
    std::vector<u64> split(info.snpmer_info.size()); std::vector<u8> m0(split.size()), m1(split.size());
    for (size_t i = 0; i < split.size(); i++) { split[i] = info.snpmer_info[i].split_kmer; m0[i] = info.snpmer_info[i].mid_bases[0]; m1[i] = info.snpmer_info[i].mid_bases[1]; }
    Trace t0_("1c.total");
    { Trace t_("1c.set_snpmers");
    std::vector<u32> weight(split.size());
    for (size_t i = 0; i < split.size(); i++) weight[i] = info.snpmer_info[i].counts[0] + info.snpmer_info[i].counts[1];
    chk(rs.ctx, svt_set_snpmers(rs.ctx, k, split.data(), m0.data(), m1.data(), weight.data(), (u32)split.size(), info.high_freq_kmers.data(), (u32)info.high_freq_kmers.size()), "svt_set_snpmers"); }
    { Trace t_("1c.extract_seeds");
    chk(rs.ctx, svt_extract_seeds(rs.ctx, rs.batch, k, args.c, args.minimum_base_quality, 1), "svt_extract_seeds"); }
    Trace t1_("1c.order+gather");
    // Round 4: the intake filters, the sort and the gather of the per-read records run on the device (svt_twin_order / svt_twin_gather): the host receives
    // the records of the kept reads already in twin order -- 12 bytes per read for the order, ~190 per twin read for the records -- instead of fetching
    // every array in read order, filtering, sorting 10^5..10^6 reads and gathering 160 bytes of signatures per read (0.12 s of a 1 M-read step).
    std::vector<u32> kept(n); std::vector<u64> key(n);
    u32 nk = 0;
    { Trace t_("1c.order");
    chk(rs.ctx, svt_twin_order(rs.ctx, rs.batch, args.min_read_length, args.max_read_length, args.c, args.quality_value_cutoff, &nk, kept.data(), key.data()), "svt_twin_order"); }
    kept.resize(nk);
    // The reference sorts the reads by id (kmer_comp.rs:233, stable), filters (:248), then sorts by estimated identity, descending and stable
    // (main.rs:538): the result is the lexicographic order (identity desc, id asc, input order).  The device sorted by identity, stable in input
    // order; an id is compared only where identities tie: every run of equal keys is re-sorted by (id, input order)
    for (u32 x = 0; x + 1 < nk;) {
        u32 y = x + 1;
        while (y < nk && key[y] == key[x]) y++;
        if (y - x > 1) std::stable_sort(kept.begin() + x, kept.begin() + y, [&](u32 a, u32 b) { return rs.ids[a] < rs.ids[b]; });   // the run is in ascending read index: stable keeps it among equal ids
        x = y;
    }
    tw.n = nk;
    tw.words = svt_snpmer_words(rs.ctx);
    tw.orig = kept;
    tw.length.resize(tw.n); tw.file_idx.resize(tw.n); tw.n_mini.resize(tw.n); tw.n_unique.resize(tw.n); tw.n_snp_filtered.resize(tw.n);
    tw.est_id.resize(tw.n); tw.est_valid.resize(tw.n); tw.lsh.resize((size_t)tw.n * SVT_LSH_TABLES); tw.lsh_valid.resize(tw.n);
    { Trace t_("1c.gather");
    chk(rs.ctx, svt_twin_gather(rs.ctx, rs.batch, kept.data(), tw.n, tw.length.data(), tw.n_mini.data(), tw.n_unique.data(), tw.n_snp_filtered.data(), tw.est_id.data(), tw.est_valid.data(),
                                tw.lsh.data(), tw.lsh_valid.data()), "svt_twin_gather"); }
    std::vector<u8> no_snp(tw.n, 0);
    for (size_t t2 = 0; t2 < tw.n; t2++) { tw.file_idx[t2] = rs.file_idx.empty() ? 0 : rs.file_idx[kept[t2]]; no_snp[t2] = tw.n_snp_filtered[t2] == 0; }
    size_t without = 0; for (u8 x : no_snp) without += x;
    tw.auto_low_polymorphism = tw.n > 0 && (double)without / (double)tw.n > 0.75;   // main.rs:539-543
}

// signature -> the dense indices of the representatives that carry it (one table of the LSH index, src/asv_cluster.rs:176-186, :303-337).
// Open addressing over (key, head) with the values chained through `next`: a lookup is one or two cache lines instead of the node walk of
// std::unordered_map (2 M lookups per 100k reads), and the values of a key are visited newest first -- the callers only count hits per
// representative, so the order is immaterial.
struct SigTable {
    std::vector<u64> key; std::vector<int32_t> head;            // head < 0: empty slot
    std::vector<u32> val; std::vector<int32_t> next;
    size_t used = 0;
    SigTable() { key.assign(1024, 0); head.assign(1024, -1); }
    static inline u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; return x; }
    inline size_t slot_of(u64 k) const {
        const size_t m = key.size() - 1; size_t i = (size_t)mix(k) & m;
        while (head[i] >= 0 && key[i] != k) i = (i + 1) & m;
        return i;
    }
    void grow() {
        std::vector<u64> ok; std::vector<int32_t> oh; ok.swap(key); oh.swap(head);
        key.assign(ok.size() * 2, 0); head.assign(oh.size() * 2, -1);
        for (size_t i = 0; i < ok.size(); i++) if (oh[i] >= 0) { const size_t j = slot_of(ok[i]); key[j] = ok[i]; head[j] = oh[i]; }
    }
    void insert(u64 k, u32 v) {
        if ((used + 1) * 2 > key.size()) grow();
        const size_t i = slot_of(k);
        if (head[i] < 0) { key[i] = k; used++; }
        val.push_back(v); next.push_back(head[i]); head[i] = (int32_t)val.size() - 1;
    }
    template <class F> inline void for_each(u64 k, F f) const { const size_t i = slot_of(k); for (int32_t x = head[i]; x >= 0; x = next[x]) f(val[x]); }
};

static bool cluster_less(const std::vector<u32>& a, const std::vector<u32>& b) {   // (len desc, first asc)
    if (a.size() != b.size()) return a.size() > b.size();
    const u32 fa = a.empty() ? 0 : a[0], fb = b.empty() ? 0 : b[0];
    return fa < fb;
}

// ==================================================================================================
// Stage 2: asv_cluster::cluster_reads_by_kmers (src/asv_cluster.rs:72-249).
// The reference's loop is written read after read, but a read's decision depends on EARLIER reads only through the
// representatives they create (:176-186).  So the loop runs in BLOCKS, and inside a block every read is decided in parallel
// against the representatives that exist at block start:
//   pass 1   candidates among the block-start representatives (query_read_against_bucket_index :303-337: the host's bucket
//            walk or the device's svt_lsh_candidates), their minimizer-set similarity by ONE svt_minimizer_shared_counts call
//            (:131-143, a 135 x 135 `Vec::contains` scan per candidate in the reference), and the decision of :144-152 -- all of
//            it per read, on the worker pool, and in a pooled multi-rank run on the rank's slice [nb r / W, nb (r + 1) / W) of the
//            block only (the decisions, one word per read, are all-gathered: svt_shard_allgatherv);
//   pass 2   a read whose best block-start candidate does not pass the threshold is a POTENTIAL new representative.  Every
//            later read of the block that shares an LSH signature with one gets that pair verified too (same slices);
//   fix-up   the ordered part: only the potentials and the reads that share a signature with an earlier potential are walked in
//            read order, exactly as the reference takes them (the candidate rule :111-125 re-applied to the union of the
//            block-start list and the representatives created inside the block).  Everything else keeps its pass-1 decision.
//            The rare read that becomes a representative WITHOUT having been a potential (its best candidate fell out of the
//            top-10 list) ends the block at the first later read that shares a signature with it (the "cut").
// The result is the sequential one for every block schedule (tests force schedules and compare with the oracle).
// ==================================================================================================
namespace {
// one u32 string per rank, rank 0's first -> `all`, word offset of every rank's part in `off` (world + 1); known_words: the lengths when every rank can compute them (saves the size exchange)
void exchange_words(svt_ctx* ctx, u32 world, const std::vector<u32>& mine, const std::vector<u64>* known_words, std::vector<u32>& all, std::vector<u64>& off) {
    std::vector<u64> bytes(world);
    if (known_words) for (u32 r = 0; r < world; r++) bytes[r] = (*known_words)[r] * 4;
    else chk(ctx, svt_shard_allgather_u64(ctx, (u64)mine.size() * 4, bytes.data()), "svt_shard_allgather_u64");
    off.assign(world + 1, 0);
    for (u32 r = 0; r < world; r++) off[r + 1] = off[r] + bytes[r] / 4;
    all.resize(off[world] + 1);
    chk(ctx, svt_shard_allgatherv(ctx, mine.data(), bytes.data(), all.data()), "svt_shard_allgatherv");
}
struct ShardPauseGuard {                                            // the ranks make DIFFERENT calls inside Stage 2 (each its slice): the tile slicing below the C-ABI is off meanwhile
    svt_ctx* c; bool on;
    ShardPauseGuard(svt_ctx* c_, bool on_) : c(c_), on(on_) { if (on) svt_shard_pause(c, 1); }
    ~ShardPauseGuard() { if (on) svt_shard_pause(c, 0); }
};
constexpr u32 DEC_POTENTIAL = 0x80000000u, DEC_SELF = 0x40000000u;   // a read's pass-1 decision: the twin id of its representative, or: becomes a representative unless the block adds a better one / is its own cluster and never a representative (no LSH signature, :176-186)
}  // namespace

std::vector<std::vector<u32>> cluster_reads_by_kmers(const ReadSet& rs, const TwinReads& tw, const ClusterArgs& args) {
    const u32 n = tw.n, k = args.kmer_size;
    const double threshold = args.primary_clustering_threshold;
    const size_t top_n = 10;                                                   // :84
    std::vector<SigTable> buckets(SVT_LSH_TABLES);                                 // signature -> dense representative indices
    std::vector<u32> reps;                                                         // dense index -> twin id (creation order => ascending id)
    std::vector<u32> assign(n);
    // memo of ratio.powf(1/k) (:144): a pure function of (count, denominator, k), evaluated by the same libm call; kept across calls of this thread (8 MB of -1.0 per call
    // otherwise).  The pool threads of a pass fill it side by side: a slot is written with the bits of the one value it can hold (relaxed atomics)
    static thread_local std::vector<double> pow_cache_tl; static thread_local u32 pow_cache_k = 0;
    if (pow_cache_tl.empty() || pow_cache_k != k) { pow_cache_tl.assign((size_t)1024 * 1024, -1.0); pow_cache_k = k; }
    double* const pow_cache = pow_cache_tl.data();
    auto sim_of = [pow_cache, k](u32 count, u32 den) -> double {
        if (den < 1024 && count < 1024) {
            double* slot = pow_cache + (size_t)den * 1024 + count;
            u64 b = __atomic_load_n((const u64*)slot, __ATOMIC_RELAXED); double c; memcpy(&c, &b, 8);
            if (c < 0.0) { c = std::pow((double)count / (double)den, 1.0 / (double)k); memcpy(&b, &c, 8); __atomic_store_n((u64*)slot, b, __ATOMIC_RELAXED); }
            return c;
        }
        return std::pow((double)count / (double)den, 1.0 / (double)k);
    };
    typedef std::pair<u32, u32> HitId;                                          // (hits, twin id); list order = (hits desc, id desc), :111
    auto by_rank = [](const HitId& a, const HitId& b) { return a > b; };
    const Tuning& tn = args.tuning;
    size_t pos = 0, B = std::max<size_t>(1, tn.stage2_first_block);
    u64 n_blocks = 0, n_cuts = 0, n_pairs1 = 0, n_pairs2 = 0, n_fix_reads = 0;
    u32 sh_rank = 0, sh_world = 1;
    svt_shard_info(rs.ctx, &sh_rank, &sh_world);
    if (svt_shard_pause(rs.ctx, 0) == 1) { svt_shard_pause(rs.ctx, 1); sh_world = 1; sh_rank = 0; }   // the caller has paused the slicing: its ranks make different calls, nothing may be dealt out here
    ShardPauseGuard pause_guard(rs.ctx, sh_world > 1);
    bool dev_lists = (tn.stage2_device < 0 ? WorkerPool::get().threads() <= 10 : tn.stage2_device != 0) && rs.ctx != nullptr;
    if (sh_world > 1) {                                                          // the ranks must agree on who builds the lists (a rank's fallbacks differ otherwise): the device only if every rank chose it
        std::vector<u64> votes(sh_world, 0);
        chk(rs.ctx, svt_shard_allgather_u64(rs.ctx, dev_lists ? 1 : 0, votes.data()), "svt_shard_allgather_u64");
        for (u64 v : votes) dev_lists = dev_lists && v != 0;
    }
    double serial_s = 0.0;                                                       // seconds of the parts every rank repeats (the ordered fix-up, the parse of the gathered records, the final grouping)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(now() - t0).count(); };
    std::vector<std::vector<HitId>>& l0 = rs.stage2.l0;                         // per read of MY slice: verify list against the representatives at block start (the vectors keep their storage from block to block and call to call)
    std::vector<std::vector<std::pair<u32, u32>>>& ext = rs.stage2.ext;         // per read of MY slice: (earlier potential u of the block, shared signatures)
    auto fresh_lists = [](auto& lists, size_t nb_) { if (lists.size() < nb_) lists.resize(nb_); for (size_t x = 0; x < nb_; x++) lists[x].clear(); };
    std::vector<u32> pa, pb, shared, shared2, dec, dec_mine, wire, wire_all, dq, dcnt, doff, xcnt, xoff_d, pidx, pr, lim;
    std::vector<size_t> poff, xoff;
    std::vector<u64> woff, known(sh_world);
    std::vector<char> potential, is_new, handled;
    std::vector<int64_t> fix_of;                                                // block read -> word offset of its record in wire_all (-1: none)
    while (pos < n) {
        size_t end = std::min<size_t>(n, pos + B), nb = end - pos;
        n_blocks++;
        const size_t my_lo = sh_world > 1 ? nb * sh_rank / sh_world : 0, my_hi = sh_world > 1 ? nb * (sh_rank + 1) / sh_world : nb, ns = my_hi - my_lo;
        fresh_lists(l0, ns); poff.assign(ns + 1, 0);
        // ---- pass 1: candidates among the representatives that exist at block start (query_read_against_bucket_index :303-337), my slice
        Trace t_cand("2.candidates");
        // device lists: hits(read, representative) = tables with equal signatures, every pair of the slice compared directly (the bucket walk below counts the same
        // thing); the list rule is applied on the device; a read with more than DEV_CAP touched representatives falls to the bucket walk
        const u32 DEV_CAP = 64;
        u32* dout = nullptr;
        if (dev_lists) {
            dq.resize(ns); dcnt.assign(ns, 0); doff.assign(ns, 0);
            for (size_t x = 0; x < ns; x++) dq[x] = tw.orig[pos + my_lo + x];
            if (!reps.empty() && ns) {
                std::vector<u32> dr(reps.size());
                for (size_t d = 0; d < reps.size(); d++) dr[d] = tw.orig[reps[d]];
                const u32 capacity = (u32)std::min<size_t>((size_t)ns * 16 + 64, (size_t)1 << 26); u32 used = 0;   // the list rule keeps ~10 per read: a read that finds the array full falls to the bucket walk
                if (rs.stage2.dout_words < (size_t)capacity * 2) { rs.stage2.dout.reset(new u32[(size_t)capacity * 2]); rs.stage2.dout_words = (size_t)capacity * 2; }
                dout = rs.stage2.dout.get();
                chk(rs.ctx, svt_lsh_candidates(rs.ctx, rs.batch, dq.data(), (u32)ns, dr.data(), (u32)reps.size(), nullptr, 0, (u32)top_n, DEV_CAP, capacity, dcnt.data(), doff.data(), dout, &used), "svt_lsh_candidates");
            }
            parallel_ranges(ns, 2048, [&](size_t, size_t lo_, size_t hi_) {
                for (size_t x = lo_; x < hi_; x++) {
                    if (dcnt[x] == 0xFFFFFFFFu) continue;
                    std::vector<HitId>& ck = l0[x];
                    const u32* e = dout + (size_t)doff[x] * 2;
                    for (u32 q = 0; q < dcnt[x]; q++) ck.push_back({e[2 * q], reps[e[2 * q + 1]]});
                }
            });
        }
        bool walk = !dev_lists;
        if (dev_lists) for (size_t x = 0; x < ns && !walk; x++) walk = dcnt[x] == 0xFFFFFFFFu;
        if (walk) parallel_ranges(ns, 256, [&](size_t, size_t lo_, size_t hi_) {          // the index is read-only while a block's candidates are collected
            std::vector<u16> hits_l(reps.size(), 0); std::vector<u32> touched_l; std::vector<HitId> cands_l;
            for (size_t x = lo_; x < hi_; x++) {
                const size_t r = pos + my_lo + x;
                if (dev_lists && dcnt[x] != 0xFFFFFFFFu) continue;
                touched_l.clear();
                if (tw.lsh_valid[r])
                    for (u32 t = 0; t < SVT_LSH_TABLES; t++) {
                        buckets[t].for_each(tw.lsh[r * SVT_LSH_TABLES + t], [&](u32 d) { if (hits_l[d]++ == 0) touched_l.push_back(d); });
                    }
                if (touched_l.empty()) continue;
                cands_l.clear();
                for (u32 d : touched_l) { cands_l.push_back({(u32)hits_l[d], reps[d]}); hits_l[d] = 0; }
                std::sort(cands_l.begin(), cands_l.end(), by_rank);           // :111
                const u32 max_hits = cands_l[0].first;
                std::vector<HitId>& ck = l0[x];
                for (auto& c : cands_l) { if (c.first == max_hits || ck.size() < top_n) ck.push_back(c); else break; }   // :118-125
            }
        });
        for (size_t x = 0; x < ns; x++) poff[x + 1] = poff[x] + l0[x].size();
        pa.resize(poff[ns]); pb.resize(poff[ns]);
        parallel_ranges(ns, 2048, [&](size_t, size_t lo_, size_t hi_) {
            for (size_t x = lo_; x < hi_; x++) { size_t o = poff[x]; const u32 a_ = tw.orig[pos + my_lo + x]; for (auto& c : l0[x]) { pa[o] = a_; pb[o] = tw.orig[c.second]; o++; } }
        });
        t_cand.~Trace(); new (&t_cand) Trace("2.resolve");
        shared.assign(pa.size(), 0); n_pairs1 += pa.size();
        if (!pa.empty()) { Trace t_("2.k5_calls"); chk(rs.ctx, svt_minimizer_shared_counts(rs.ctx, rs.batch, rs.batch, pa.data(), pb.data(), pa.size(), shared.data(), nullptr), "svt_minimizer_shared_counts"); }
        // ---- the decision of :131-152 against the block-start representatives, every read of my slice on its own.  A read whose best match does not pass the threshold
        // is a POTENTIAL representative (one created inside the block can only ADD candidates; the rare read that loses its best candidate to the top-10 cut and becomes
        // a representative anyway ends the block, see the fix-up)
        dec_mine.resize(ns);
        { Trace t_("2.decide");
        parallel_ranges(ns, 1024, [&](size_t, size_t lo_, size_t hi_) {
            for (size_t x = lo_; x < hi_; x++) {
                const size_t r = pos + my_lo + x;
                if (!tw.lsh_valid[r]) { dec_mine[x] = DEC_SELF; continue; }      // never inserted into the index (:176-186), never a candidate list either
                double best_sim = 0.0; int64_t best = -1;
                for (size_t j = 0; j < l0[x].size(); j++) {
                    const double sim = sim_of(shared[poff[x] + j], std::max(tw.n_unique[r], tw.n_mini[l0[x][j].second]));   // :143-144
                    if (sim > best_sim) { best_sim = sim; best = (int64_t)l0[x][j].second; }
                }
                dec_mine[x] = (best >= 0 && best_sim > threshold) ? (u32)best : DEC_POTENTIAL;                           // :152
            }
        }); }
        if (sh_world > 1) {
            Trace t_("2.exchange");
            for (u32 r = 0; r < sh_world; r++) known[r] = nb * (r + 1) / sh_world - nb * r / sh_world;
            exchange_words(rs.ctx, sh_world, dec_mine, &known, dec, woff);
            dec.resize(nb);
        } else dec.swap(dec_mine);
        // ---- pass 2: every later read of the block that shares a signature with a potential representative gets that pair verified too (my slice against the potentials of the WHOLE block)
        Trace t_p2("2.pass2");
        potential.assign(nb, 0); pidx.clear(); pr.clear(); lim.resize(nb);
        for (size_t x = 0; x < nb; x++) { lim[x] = (u32)pidx.size(); if (dec[x] & DEC_POTENTIAL) { potential[x] = 1; pidx.push_back((u32)x); pr.push_back(tw.orig[pos + x]); } }
        const size_t n_pot = pidx.size();
        fresh_lists(ext, ns);
        bool walk2 = !dev_lists;
        if (dev_lists && n_pot && ns) {
            // the potentials are the references, a read sees those before it (ref_limit = how many potentials precede it); a read that shares a signature with more than 256 of them falls to the map walk below
            const u32 XCAP = 256, capacity = (u32)std::min<size_t>((size_t)ns * 8 + 4096, (size_t)1 << 26); u32 used = 0;
            xcnt.assign(ns, 0); xoff_d.assign(ns, 0);
            if (rs.stage2.xout_words < (size_t)capacity * 2) { rs.stage2.xout.reset(new u32[(size_t)capacity * 2]); rs.stage2.xout_words = (size_t)capacity * 2; }
            u32* xout = rs.stage2.xout.get();
            chk(rs.ctx, svt_lsh_candidates(rs.ctx, rs.batch, dq.data(), (u32)ns, pr.data(), (u32)n_pot, lim.data() + my_lo, 1, 0, XCAP, capacity, xcnt.data(), xoff_d.data(), xout, &used), "svt_lsh_candidates(pass 2)");
            for (size_t x = 0; x < ns; x++) {
                if (xcnt[x] == 0xFFFFFFFFu) { walk2 = true; continue; }
                std::vector<std::pair<u32, u32>>& e = ext[x];
                const u32* src = xout + (size_t)xoff_d[x] * 2;
                for (u32 q = 0; q < xcnt[x]; q++) e.push_back({pidx[src[2 * q]], src[2 * q + 1]});
            }
        }
        if (walk2 && n_pot && ns) {
            std::vector<SigTable> psig(SVT_LSH_TABLES);                             // signature -> the potentials that carry it (open addressing, chained values: no allocation per key)
            std::vector<u64> pbloom((size_t)SVT_LSH_TABLES * 1024, 0);             // 64 Kbit per table: almost every later read misses every table, skip its 20 table lookups
            auto bloom_bit = [](u64 sig) { return (u32)((sig * 0x9E3779B97F4A7C15ull) >> 48); };
            par_for(SVT_LSH_TABLES, [&](size_t t) {
                for (u32 x : pidx) { if (x >= my_hi) break; const u64 sg = tw.lsh[(pos + x) * SVT_LSH_TABLES + t]; psig[t].insert(sg, x); const u32 bb = bloom_bit(sg); pbloom[t * 1024 + (bb >> 6)] |= 1ull << (bb & 63); }
            });
            parallel_ranges(ns, 512, [&](size_t, size_t lo_, size_t hi_) {
                std::vector<std::pair<u32, u32>> tmp;
                for (size_t xs = lo_; xs < hi_; xs++) {
                    const size_t x = my_lo + xs, r = pos + x;
                    if (!tw.lsh_valid[r]) continue;
                    if (dev_lists && xcnt[xs] != 0xFFFFFFFFu) continue;
                    tmp.clear();
                    for (u32 t = 0; t < SVT_LSH_TABLES; t++) {
                        const u64 sg = tw.lsh[r * SVT_LSH_TABLES + t]; const u32 bb = bloom_bit(sg);
                        if (!((pbloom[(size_t)t * 1024 + (bb >> 6)] >> (bb & 63)) & 1)) continue;
                        psig[t].for_each(sg, [&](u32 u) { if (u < x) tmp.push_back({u, 1}); });
                    }
                    if (tmp.empty()) continue;
                    std::sort(tmp.begin(), tmp.end());
                    std::vector<std::pair<u32, u32>>& e = ext[xs];
                    for (auto& p : tmp) { if (!e.empty() && e.back().first == p.first) e.back().second++; else e.push_back(p); }
                }
            });
        }
        // bound the second launch: a block whose extra pairs would explode (few matches yet, e.g. the very first reads) is shortened at the read where my slice's pairs pass the cap
        const size_t PAIR_CAP = (size_t)tn.stage2_pair_cap;
        size_t cut_at = nb, ns_used = ns;
        xoff.assign(ns + 1, 0);
        for (size_t x = 0; x < ns; x++) {
            xoff[x + 1] = xoff[x] + ext[x].size();
            if (xoff[x + 1] > PAIR_CAP && my_lo + x > 0) { cut_at = my_lo + x; ns_used = x; break; }
        }
        pa.resize(xoff[ns_used]); pb.resize(xoff[ns_used]);
        parallel_ranges(ns_used, 2048, [&](size_t, size_t lo_, size_t hi_) {
            for (size_t x = lo_; x < hi_; x++) { size_t o = xoff[x]; const u32 a_ = tw.orig[pos + my_lo + x]; for (auto& e : ext[x]) { pa[o] = a_; pb[o] = tw.orig[pos + e.first]; o++; } }
        });
        t_p2.~Trace(); new (&t_p2) Trace("2.k5_second+records");
        shared2.assign(pa.size(), 0); n_pairs2 += pa.size();
        if (!pa.empty()) { Trace t_("2.k5_calls"); chk(rs.ctx, svt_minimizer_shared_counts(rs.ctx, rs.batch, rs.batch, pa.data(), pb.data(), pa.size(), shared2.data(), nullptr), "svt_minimizer_shared_counts"); }
        // ---- the records of the ordered fix-up: for every read of my slice that shares a signature with an earlier potential, its block-start list and its in-block list with the
        // verified counts.  Words: cut_at, then per read: x, |l0|, |ext|, (hits, twin id, shared) x |l0|, (u, hits, shared) x |ext|
        wire.clear(); wire.push_back((u32)cut_at);
        for (size_t x = 0; x < ns_used; x++) {
            if (ext[x].empty()) continue;
            wire.push_back((u32)(my_lo + x)); wire.push_back((u32)l0[x].size()); wire.push_back((u32)ext[x].size());
            for (size_t j = 0; j < l0[x].size(); j++) { wire.push_back(l0[x][j].first); wire.push_back(l0[x][j].second); wire.push_back(shared[poff[x] + j]); }
            for (size_t j = 0; j < ext[x].size(); j++) { wire.push_back(ext[x][j].first); wire.push_back(ext[x][j].second); wire.push_back(shared2[xoff[x] + j]); }
        }
        if (sh_world > 1) { Trace t_("2.exchange"); exchange_words(rs.ctx, sh_world, wire, nullptr, wire_all, woff); }
        else { wire_all.swap(wire); woff.assign(2, 0); woff[1] = wire_all.size(); }
        t_p2.~Trace(); new (&t_p2) Trace("2.tail");
        // ---- from here on every rank does the same work: the ordered fix-up
        const auto t_serial = now();
        Trace t_fix("2.fixup");
        fix_of.assign(nb, -1);
        for (u32 r = 0; r < (sh_world > 1 ? sh_world : 1u); r++) {
            u64 q = woff[r]; const u64 qe = woff[r + 1];
            if (q >= qe) throw Error{SVT_ERR_STATE, "Stage 2: a rank sent no fix-up record header"};
            cut_at = std::min<size_t>(cut_at, wire_all[q++]);
            while (q < qe) { const u32 x = wire_all[q]; if (x >= nb || q + 3 > qe) throw Error{SVT_ERR_STATE, "Stage 2: malformed fix-up record"}; fix_of[x] = (int64_t)q; q += 3 + 3 * ((u64)wire_all[q + 1] + wire_all[q + 2]); }
            if (q != qe) throw Error{SVT_ERR_STATE, "Stage 2: malformed fix-up record"};
        }
        if (cut_at < nb) { nb = cut_at; end = pos + nb; n_cuts++; }
        is_new.assign(nb, 0);
        size_t first_dirty = nb;                                                // the block ends before the first read whose candidates could not be foreseen
        auto new_representative = [&](size_t x) {                               // :176-186
            const size_t r = pos + x;
            assign[r] = (u32)r;
            const u32 dense = (u32)reps.size(); reps.push_back((u32)r);
            for (u32 t = 0; t < SVT_LSH_TABLES; t++) buckets[t].insert(tw.lsh[r * SVT_LSH_TABLES + t], dense);
            is_new[x] = 1;
        };
        std::vector<std::pair<HitId, u32>> all;                                  // (hits, twin id) -> verified count
        for (size_t x = 0; x < nb && x < first_dirty; x++) {
            if (!potential[x] && fix_of[x] < 0) continue;                        // decided in pass 1
            const size_t r = pos + x;
            n_fix_reads++;
            if (fix_of[x] < 0) { new_representative(x); continue; }              // a potential no earlier potential shares a signature with: nothing can have been added to its candidates
            const u32* rec = wire_all.data() + fix_of[x];
            const u32 n0 = rec[1], nx = rec[2]; const u32* c0 = rec + 3; const u32* cx = c0 + 3 * (size_t)n0;
            bool has_new = false;
            for (u32 j = 0; j < nx; j++) if (is_new[cx[3 * j]]) { has_new = true; break; }
            if (!has_new) {                                                      // none of them became a representative: the pass-1 decision stands
                if (potential[x]) new_representative(x); else assign[r] = dec[x];
                continue;
            }
            // the representatives created earlier in this block join the candidates: re-apply the list rule (:111-125) to the union;
            // l0 is a prefix of the sorted block-start candidates that is long enough for any outcome of the rule
            all.clear();
            for (u32 j = 0; j < n0; j++) all.push_back({HitId(c0[3 * j], c0[3 * j + 1]), c0[3 * j + 2]});
            for (u32 j = 0; j < nx; j++) if (is_new[cx[3 * j]]) all.push_back({HitId(cx[3 * j + 1], (u32)(pos + cx[3 * j])), cx[3 * j + 2]});
            std::sort(all.begin(), all.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
            const u32 max_hits = all[0].first.first;
            size_t taken = 0; double best_sim = 0.0; int64_t best = -1;
            for (auto& c : all) {
                if (!(c.first.first == max_hits || taken < top_n)) break;
                taken++;
                const double sim = sim_of(c.second, std::max(tw.n_unique[r], tw.n_mini[c.first.second]));
                if (sim > best_sim) { best_sim = sim; best = (int64_t)c.first.second; }
            }
            if (best >= 0 && best_sim > threshold) { assign[r] = (u32)best; continue; }                        // :152
            new_representative(x);
            if (!potential[x]) {
                // not foreseen (its best candidate fell out of the top-10 list): the pairs of later reads with this representative
                // were not verified -> the block ends at the first later read that shares a signature with it
                for (size_t x2 = x + 1; x2 < first_dirty; x2++) {
                    if (!tw.lsh_valid[pos + x2]) continue;
                    bool hit = false;
                    for (u32 t = 0; t < SVT_LSH_TABLES && !hit; t++) hit = tw.lsh[(pos + x2) * SVT_LSH_TABLES + t] == tw.lsh[r * SVT_LSH_TABLES + t];
                    if (hit) { first_dirty = x2; break; }
                }
            }
        }
        const size_t resolved = std::min(first_dirty, nb);
        if (resolved < nb) n_cuts++;
        t_fix.~Trace(); new (&t_fix) Trace("2.assign");
        parallel_ranges(resolved, 4096, [&](size_t, size_t lo_, size_t hi_) {    // everything the fix-up did not touch keeps its pass-1 decision
            for (size_t x = lo_; x < hi_; x++) {
                if (potential[x] || fix_of[x] >= 0) continue;
                assign[pos + x] = (dec[x] & DEC_SELF) ? (u32)(pos + x) : dec[x];
            }
        });
        serial_s += since(t_serial);
        pos += resolved;
        if (resolved == B) B = std::min<size_t>(B * 2, std::max<size_t>(1, tn.stage2_max_block)); else if (resolved < nb) B = std::max<size_t>(std::max<size_t>(1, tn.stage2_first_block), std::max(resolved, (B * 3) / 4));
        if (trace_on() && pos >= n) fprintf(stderr, "[savont-trace] stage2 blocks %llu cuts %llu reps %zu, pairs verified %llu + %llu, %llu reads through the ordered fix-up\n", (unsigned long long)n_blocks, (unsigned long long)n_cuts, reps.size(), (unsigned long long)n_pairs1, (unsigned long long)n_pairs2, (unsigned long long)n_fix_reads);
    }
    // clusters: members ascending (:216-218); a representative is the smallest member of its cluster
    const auto t_group = now();
    std::vector<u32> size_of(n, 0);
    for (u32 r = 0; r < n; r++) size_of[assign[r]]++;
    std::vector<u32> slot(n, 0xFFFFFFFFu);
    std::vector<std::vector<u32>> clusters;
    for (u32 r = 0; r < n; r++) if (size_of[r]) { slot[r] = (u32)clusters.size(); clusters.emplace_back(); clusters.back().reserve(size_of[r]); }
    for (u32 r = 0; r < n; r++) clusters[slot[assign[r]]].push_back(r);
    std::stable_sort(clusters.begin(), clusters.end(), cluster_less);         // :212 (equal sizes: smaller first member)
    std::vector<std::vector<u32>> kept;
    for (auto& c : clusters) if (c.size() >= args.min_cluster_size) kept.push_back(std::move(c));   // :221
    rs.stage2.serial_seconds = serial_s + since(t_group);
    return kept;
}

// ==================================================================================================
// SNPmer consensus algebra on bitsets (src/asv_cluster.rs:840-1003).  A consensus is (presence, allele);
// position / count fields of ConsensusPoly only order the list and are never compared (:892, :968-994).
// Consensus rows are BUILT on the GPU (svt_snpmer_consensus, one launch for all clusters of all k-mer
// groups); the host only compares them.
// ==================================================================================================
struct Cons { const u64* p; const u64* a; u32 len; };
static inline u32 popc(u64 x) { return (u32)__builtin_popcountll(x); }
static void compare_consensus(const Cons& x, const Cons& y, u32 W, u32& m, u32& mm) {   // :968-994
    m = mm = 0;
    for (u32 w = 0; w < W; w++) { const u64 both = x.p[w] & y.p[w], d = x.a[w] ^ y.a[w]; m += popc(both & ~d); mm += popc(both & d); }
}
static bool concordant(const Cons& x, const Cons& y, u32 m, u32 mm) {                   // :997-1003 (matches/mismatches are symmetric)
    return mm == 0 && m >= std::min(x.len, std::max<u32>(y.len, 2));
}

// recluster_one_round_top_n (top_n = None), src/asv_cluster.rs:1146-1270; cons[i] = consensus of clusters[i] built
// ONCE before the loop (:1154-1167); the consensus of i is NOT rebuilt inside its j loop (:1201-1202).
static void recluster_one_round(std::vector<std::vector<u32>>& clusters, const std::vector<Cons>& cons, u32 W, u32& num_merges) {
    struct Item { std::vector<u32> members; Cons cons; };
    std::vector<Item> all;
    for (size_t i = 0; i < clusters.size(); i++) { if (clusters[i].empty()) continue; all.push_back(Item{clusters[i], cons[i]}); }
    std::stable_sort(all.begin(), all.end(), [](const Item& a, const Item& b) { return cluster_less(a.members, b.members); });   // :1170
    std::vector<char> merged(all.size(), 0);
    std::vector<std::vector<u32>> out;
    num_merges = 0;
    for (size_t i = 0; i < all.size(); i++) {
        if (merged[i]) continue;
        for (size_t j = i + 1; j < all.size(); j++) {
            if (merged[j]) continue;
            const Cons& ci = all[i].cons; const Cons& cj = all[j].cons;
            u32 m, mm; compare_consensus(ci, cj, W, m, mm);
            bool conc = concordant(ci, cj, m, mm) && concordant(cj, ci, m, mm);                                                     // :1203-1206
            const size_t li = all[i].members.size(), lj = all[j].members.size();
            const size_t max_len = std::max(li, lj), min_len = std::min(li, lj);
            if (mm == 0 && (double)m > (double)std::min(ci.len, cj.len) * 0.975 && max_len / min_len > 50) conc = true;           // :1212-1215
            if (mm == 0 && max_len / min_len > 500 && min_len <= 2) conc = true;                                                    // :1220
            if (conc) { all[i].members.insert(all[i].members.end(), all[j].members.begin(), all[j].members.end()); merged[j] = 1; num_merges++; }
        }
        out.push_back(std::move(all[i].members));
    }
    std::stable_sort(out.begin(), out.end(), cluster_less);                    // :1266
    clusters.swap(out);
}

// consensus rows of every cluster of every group in ONE launch; returns host rows (and the device set if wanted)
typedef std::map<u32, std::vector<std::vector<u32>>> Groups;
static void consensus_all(const ReadSet& rs, const TwinReads& tw, const Groups& groups, std::vector<u64>& P, std::vector<u64>& A, svt_bitset** set) {
    std::vector<u64> off(1, 0); std::vector<u32> mem;
    for (auto& kv : groups) for (auto& cl : kv.second) { for (u32 r : cl) mem.push_back(tw.orig[r]); off.push_back(mem.size()); }
    const u32 nc = (u32)off.size() - 1;
    P.assign((size_t)nc * tw.words, 0); A.assign((size_t)nc * tw.words, 0);
    if (set) *set = nullptr;
    if (nc == 0) return;
    Trace t_("3.consensus_calls");
    chk(rs.ctx, svt_snpmer_consensus(rs.ctx, rs.batch, off.data(), mem.data(), nc, P.data(), A.data(), set), "svt_snpmer_consensus");
}

// one iteration of recluster_using_consensus_reps (:1307-1350) over ALL k-mer groups: 3 GPU calls in total
// A group whose iteration merged nothing and reassigned nothing is at a fixed point of the (deterministic) iteration: it moves to
// `settled` and later iterations, which the reference runs over every group until no group merges (:1296-1367), skip it.
static u32 recluster_iteration(const ReadSet& rs, const TwinReads& tw, Groups& groups, Groups& settled, const ClusterArgs& args) {
    const u32 W = tw.words;
    std::vector<u64> P, A;
    consensus_all(rs, tw, groups, P, A, nullptr);
    // the groups are independent of each other: flat views for the worker pool
    std::vector<u32> gkey; std::vector<std::vector<std::vector<u32>>*> gcl; std::vector<size_t> cfirst;
    { size_t ci = 0; for (auto& kv : groups) { gkey.push_back(kv.first); gcl.push_back(&kv.second); cfirst.push_back(ci); ci += kv.second.size(); } }
    const size_t ng = gkey.size();
    std::vector<std::vector<std::vector<u32>>> before(ng);
    std::vector<u32> merges_of(ng, 0);
    { Trace t_("3.recluster.one_round");
    par_for(ng, [&](size_t g) {                                                // merge inside every group (host, O(C^2 W))
        std::vector<std::vector<u32>>& cls = *gcl[g];
        before[g] = cls;
        std::vector<Cons> cons(cls.size());
        for (size_t i = 0; i < cls.size(); i++) {
            const size_t ci = cfirst[g] + i;
            cons[i].p = &P[ci * W]; cons[i].a = &A[ci * W]; cons[i].len = 0;
            for (u32 w = 0; w < W; w++) cons[i].len += popc(cons[i].p[w]);
        }
        u32 merges = 0; recluster_one_round(cls, cons, W, merges); merges_of[g] = merges;
    }); }
    u32 total_merges = 0; for (u32 m : merges_of) total_merges += m;
    // reassign_reads_to_best_cluster (:1007-1130): consensus of the MERGED clusters, then every read x every cluster of its group
    svt_bitset* S = nullptr;
    consensus_all(rs, tw, groups, P, A, &S);
    std::vector<size_t> rfirst(ng + 1, 0); std::vector<u32> cbase(ng + 1, 0);
    for (size_t g = 0; g < ng; g++) { size_t nr = 0; for (auto& cl : *gcl[g]) nr += cl.size(); rfirst[g + 1] = rfirst[g] + nr; cbase[g + 1] = cbase[g] + (u32)gcl[g]->size(); }
    const size_t nrows = rfirst[ng];
    std::vector<u32> rows(nrows), twin_of(nrows), lo(nrows), hi(nrows);
    par_for(ng, [&](size_t g) {
        size_t ri = rfirst[g];
        for (auto& cl : *gcl[g]) for (u32 r : cl) { rows[ri] = tw.orig[r]; twin_of[ri] = r; lo[ri] = cbase[g]; hi[ri] = cbase[g + 1]; ri++; }
    });
    std::vector<u32> best(nrows, 0);
    if (nrows && S) {
        Trace t_("3.best_column_calls");
        int rc = svt_snpmer_best_column(rs.ctx, rs.batch, SVT_VIEW_FILTERED, rows.data(), (u32)nrows, S, lo.data(), hi.data(), best.data(), nullptr);
        svt_bitset_free(rs.ctx, S);
        chk(rs.ctx, rc, "svt_snpmer_best_column");
    } else { if (S) svt_bitset_free(rs.ctx, S); for (size_t i = 0; i < nrows; i++) best[i] = lo[i]; }
    std::vector<std::vector<std::vector<u32>>> kept_of(ng); std::vector<char> is_settled(ng, 0);
    par_for(ng, [&](size_t g) {
        const u32 nc = cbase[g + 1] - cbase[g];
        std::vector<std::vector<u32>> out(nc);
        for (size_t ri = rfirst[g]; ri < rfirst[g + 1]; ri++) out[best[ri] - cbase[g]].push_back(twin_of[ri]);
        std::vector<std::vector<u32>>& kept = kept_of[g];
        for (auto& cl : out) if (!cl.empty() && cl.size() >= args.min_cluster_size) { std::sort(cl.begin(), cl.end()); kept.push_back(std::move(cl)); }   // :1121-1124
        is_settled[g] = merges_of[g] == 0 && kept == before[g];
    });
    Groups next;
    for (size_t g = 0; g < ng; g++) {
        if (kept_of[g].empty()) continue;                                      // :1339
        if (is_settled[g]) settled[gkey[g]] = std::move(kept_of[g]); else next[gkey[g]] = std::move(kept_of[g]);
    }
    groups.swap(next);
    return total_merges;
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
    Groups groups;
    std::vector<u32> rep_pos(tw.n, ~0u), rep_size(tw.n, 0), assign(tw.n, ~0u);   // indexed by twin id: groups are disjoint, threads never share an entry
    // The greedy loop of a k-mer cluster is order-dependent, but the k-mer clusters are independent of each other
    // (src/asv_cluster.rs:596 iterates them one after the other): one task per group on the worker pool, each on its own
    // svt_fork context (own stream: the small K6 launches of different groups overlap on the GPU).  Largest groups first.
    std::vector<u32> gidx;
    for (u32 g = 0; g < kmer_clusters.size(); g++) if (!kmer_clusters[g].empty()) gidx.push_back(g);
    std::stable_sort(gidx.begin(), gidx.end(), [&](u32 a, u32 b) { return kmer_clusters[a].size() > kmer_clusters[b].size(); });
    // Multi-GPU (svt_set_shard): the groups are independent through the whole of Stage 3 -- greedy loops AND reclustering -- so every rank runs the
    // groups it owns (largest first onto the least loaded rank: the same deterministic assignment on every rank), host decisions and K6 tiles alike,
    // and the clusters are gathered at the end.  The one thing the groups share is the reclustering loop's exit test (no merge anywhere), which becomes
    // a sum over the ranks.  Tile slicing inside the library is paused meanwhile: the ranks make different calls here.
    u32 sh_rank = 0, sh_world = 1;
    svt_shard_info(rs.ctx, &sh_rank, &sh_world);
    const bool by_group = sh_world > 1 && gidx.size() >= 2 * (size_t)sh_world;
    struct PauseTiles { svt_ctx* c; bool on; int was; ~PauseTiles() { if (on) svt_shard_pause(c, was); } } pause_tiles{rs.ctx, by_group, 0};
    if (by_group) {
        pause_tiles.was = std::max(0, svt_shard_pause(rs.ctx, 1));
        std::vector<u64> load(sh_world, 0); std::vector<u32> mine;
        for (u32 g : gidx) {
            u32 best = 0; for (u32 r = 1; r < sh_world; r++) if (load[r] < load[best]) best = r;
            load[best] += kmer_clusters[g].size();
            if (best == sh_rank) mine.push_back(g);
        }
        gidx.swap(mine);
    }
    // (group key, its clusters) of every rank, in key order: what the single-rank run holds in `groups`
    auto gather_groups = [&](Groups& gs) {
        std::vector<u32> buf;
        for (auto& kv : gs) { buf.push_back(kv.first); buf.push_back((u32)kv.second.size()); for (auto& cl : kv.second) { buf.push_back((u32)cl.size()); buf.insert(buf.end(), cl.begin(), cl.end()); } }
        std::vector<u64> bytes(sh_world, 0);
        chk(rs.ctx, svt_shard_allgather_u64(rs.ctx, (u64)buf.size() * 4, bytes.data()), "svt_shard_allgather_u64");
        u64 total = 0; for (u64 b : bytes) total += b;
        std::vector<u32> all(total / 4 + 1);
        chk(rs.ctx, svt_shard_allgatherv(rs.ctx, buf.data(), bytes.data(), all.data()), "svt_shard_allgatherv");
        Groups out;
        for (size_t i = 0; i < total / 4;) {
            const u32 key = all[i++], nc = all[i++];
            std::vector<std::vector<u32>>& dst = out[key];
            for (u32 c = 0; c < nc; c++) { const u32 len = all[i++]; dst.emplace_back(all.begin() + i, all.begin() + i + len); i += len; }
        }
        gs.swap(out);
    };
    std::vector<std::vector<std::vector<u32>>> group_out(kmer_clusters.size());
    // the decisions of one block of one group, in read order: `cnt` / `lst` hold, per row of the block, the compatible columns (column index inside the
    // group's column list: < R an existing representative, else R + position in the block) with their match counts
    auto decide_block = [&](const std::vector<u32>& kc, std::vector<u32>& reps, size_t pos, size_t nb, u32 R, const u32* cnt, const std::pair<u32, u32>* lst) {
        for (size_t i = 0; i < nb; i++) {
            const u32 rid = kc[pos + i];
            const bool iterative = reps.size() > 1000;                     // :615
            int best = -1;
            if (iterative) {                                               // :413-464; find_any -> FIRST compatible representative
                u32 best_pos = ~0u;
                for (u32 j = cnt[i]; j < cnt[i + 1]; j++) {
                    const u32 col = lst[j].first; u32 p;
                    if (col < R) p = col;
                    else { p = rep_pos[kc[pos + (col - R)]]; if (p == ~0u) continue; }
                    if (p < best_pos) best_pos = p;
                }
                if (best_pos != ~0u) best = (int)reps[best_pos];
            } else {                                                       // :467-510
                std::array<int64_t, 3> bk{0, 0, 0}; bool have = false;
                for (u32 j = cnt[i]; j < cnt[i + 1]; j++) {
                    const u32 col = lst[j].first; u32 cand;
                    if (col < R) cand = reps[col];
                    else { cand = kc[pos + (col - R)]; if (rep_pos[cand] == ~0u) continue; }
                    std::array<int64_t, 3> key{-(int64_t)lst[j].second, (int64_t)rep_size[cand], (int64_t)cand};   // :494
                    if (!have || key < bk) { bk = key; have = true; }
                }
                if (have) best = (int)bk[2];
            }
            if (best >= 0) { assign[rid] = (u32)best; rep_size[(u32)best] += 1; }       // :386-394
            else { rep_pos[rid] = (u32)reps.size(); reps.push_back(rid); assign[rid] = rid; rep_size[rid] = 1; }   // :397-410
        }
    };
    auto finish_group = [&](u32 g, const std::vector<u32>& reps) {
        const std::vector<u32>& kc = kmer_clusters[g];
        std::vector<std::vector<u32>> local(reps.size());
        for (u32 r : kc) local[rep_pos[assign[r]]].push_back(r);             // kc ascending => members ascending (:684-686)
        std::stable_sort(local.begin(), local.end(), cluster_less);            // :687
        std::vector<std::vector<u32>> kept;
        for (auto& cl : local) if (cl.size() >= args.min_cluster_size) kept.push_back(std::move(cl));   // :692
        group_out[g] = std::move(kept);
    };
    const Tuning& tn0 = args.tuning;
    auto block_size = [&](size_t n_reps, size_t pos) { return std::max<size_t>(1, n_reps == 0 ? tn0.stage3_first_block : (pos < tn0.stage3_switch ? tn0.stage3_block : tn0.stage3_max_block)); };
    bool waves_done = false;
    if (tn0.stage3_waves) {
        // ---- one device call per WAVE: the next block of EVERY unfinished group as a segment of one launch (svt_snpmer_compat_lists_seg).  The
        // greedy loop of a group is order-dependent, the groups are independent: ~5 calls per step instead of one per (group, block) -- ~116 at
        // 100k reads, each with its uploads, three launches and a wait (profiles/r02_kernel_stats.csv: 2 082 launches of 38 us in 18 steps).
        struct GS { u32 g; size_t pos = 0; std::vector<u32> reps; size_t nb = 0; u32 R = 0; };
        std::vector<GS> st(gidx.size());
        for (size_t x = 0; x < gidx.size(); x++) { st[x].g = gidx[x]; for (u32 r : kmer_clusters[gidx[x]]) { rep_pos[r] = ~0u; rep_size[r] = 0; assign[r] = ~0u; } }
        std::vector<u32> rows, cols, seg_row_off, seg_col_off, act, cnt;
        // the output arrays of the wave calls (16 entries per row of the first wave: 13 MB at 100k reads) are written by the library up to n_out: no zero fill
        std::unique_ptr<u32[]> o_buf; u64 o_cap = 0; u32 *o_col = nullptr, *o_mm = nullptr;
        std::vector<std::pair<u32, u32>> lst;
        bool fits = true;
        while (fits) {
            rows.clear(); cols.clear(); seg_row_off.assign(1, 0); seg_col_off.assign(1, 0); act.clear();
            for (size_t x = 0; x < st.size(); x++) {
                GS& s_ = st[x]; const std::vector<u32>& kc = kmer_clusters[s_.g];
                if (s_.pos >= kc.size()) continue;
                const size_t end = std::min(kc.size(), s_.pos + block_size(s_.reps.size(), s_.pos));
                s_.nb = end - s_.pos; s_.R = (u32)s_.reps.size();
                for (u32 r : s_.reps) cols.push_back(tw.orig[r]);
                for (size_t i = 0; i < s_.nb; i++) { const u32 o = tw.orig[kc[s_.pos + i]]; rows.push_back(o); cols.push_back(o); }
                seg_row_off.push_back((u32)rows.size()); seg_col_off.push_back((u32)cols.size()); act.push_back((u32)x);
            }
            if (act.empty()) { waves_done = true; break; }
            u64 n_out = 0, cap = std::max<u64>(4096, (u64)rows.size() * 16);
            while (true) {
                Trace t_("3.compat_calls");
                if (cap > o_cap) { o_buf.reset(new u32[(size_t)cap * 2]); o_cap = cap; }
                o_col = o_buf.get(); o_mm = o_col + o_cap;
                cnt.resize(rows.size() + 1);
                // row by row (svt_snpmer_compat_rows_seg): cnt = the rows' offsets, made on the device -- the host built them from unordered triples before (a
                // count, a prefix sum and a scatter over ~10^6 records per step)
                const int rc = svt_snpmer_compat_rows_seg(rs.ctx, rs.batch, SVT_VIEW_ALL, rows.data(), (u32)rows.size(), seg_row_off.data(), cols.data(), seg_col_off.data(), (u32)act.size(),
                                                          SVT_LIST_COMPATIBLE, cnt.data(), o_col, o_mm, cap, &n_out);
                if (rc == SVT_ERR_OVERFLOW) { cap = n_out + 1024; continue; }
                if (rc == SVT_ERR_TOOWIDE && st[act[0]].pos == 0 && act.size() == st.size()) { fits = false; break; }   // SNPmer rows too wide for the LDS tile (a property of the table width: the same on every rank and in the first wave): the per-group path below.  Every other failure is fatal
                chk(rs.ctx, rc, "svt_snpmer_compat_rows_seg");
                break;
            }
            if (!fits) break;
            Trace t_g("3.greedy.host");
            lst.resize(n_out);
            for (u64 i = 0; i < n_out; i++) lst[i] = {o_col[i], o_mm[i] >> 16};
            par_for(act.size(), [&](size_t a_) {
                GS& s_ = st[act[a_]];
                decide_block(kmer_clusters[s_.g], s_.reps, s_.pos, s_.nb, s_.R, cnt.data() + seg_row_off[a_], lst.data());
                s_.pos += s_.nb;
            });
        }
        if (waves_done) { Trace t_fin("3.group.finish"); par_for(st.size(), [&](size_t x) { finish_group(st[x].g, st[x].reps); }); }
    }
    if (!waves_done) {
    if (!rs.forks) const_cast<ReadSet&>(rs).forks = std::make_shared<ForkPool>(rs.ctx);
    ForkPool& fork_pool = *rs.forks;
    par_for(gidx.size(), [&](size_t gx) {
        const u32 g = gidx[gx];
        struct Lease { ForkPool& p; svt_ctx* c; explicit Lease(ForkPool& pp) : p(pp), c(pp.acquire()) {} ~Lease() { p.release(c); } } lease(fork_pool);
        svt_ctx* ctx = lease.c;
        std::vector<u32> o_row, o_col, o_mm, cols, rows, cnt, fill;
        std::vector<std::pair<u32, u32>> lst;
        const std::vector<u32>& kc = kmer_clusters[g];
        std::vector<u32> reps;                              // twin ids, in creation order (= `representatives`, :606)
        // twin id -> index in reps (~0 = not a representative) / current cluster size / assignment: flat arrays, reset per group
        for (u32 r : kc) { rep_pos[r] = ~0u; rep_size[r] = 0; assign[r] = ~0u; }
        size_t pos = 0;
        // Block schedule: a short first block (no representatives exist yet, so every in-block pair has to be listed), then 2048 reads
        // per block and, once the group's representatives are mostly established, 16384.  From then on the device reports an in-block "earlier read" column only when that read has no compatible
        // existing representative (triangular mode 2): nothing else can become a representative inside the block.
        while (pos < kc.size()) {
            const size_t end = std::min(kc.size(), pos + block_size(reps.size(), pos)), nb = end - pos;
            const u32 R = (u32)reps.size();
            rows.resize(nb); cols.resize(R + nb);
            for (u32 i = 0; i < R; i++) cols[i] = tw.orig[reps[i]];
            for (size_t i = 0; i < nb; i++) { rows[i] = tw.orig[kc[pos + i]]; cols[R + i] = rows[i]; }
            u64 n_out = 0, cap = std::max<u64>(4096, (u64)nb * 64);
            while (true) {
                Trace t_("3.compat_calls");
                o_row.resize(cap); o_col.resize(cap); o_mm.resize(cap);
                int rc = svt_snpmer_compat_lists(ctx, rs.batch, SVT_VIEW_ALL, rows.data(), (u32)nb, rs.batch, SVT_VIEW_ALL, nullptr, cols.data(), (u32)cols.size(),
                                                 SVT_LIST_COMPATIBLE, 2, R, nullptr, o_row.data(), o_col.data(), o_mm.data(), cap, &n_out);
                if (rc == SVT_ERR_OVERFLOW) { cap = n_out + 1024; continue; }
                chk(ctx, rc, "svt_snpmer_compat_lists");
                break;
            }
            Trace t_g("3.greedy.host");
            // bucket the triples by row (counting sort)
            cnt.assign(nb + 1, 0);
            for (u64 i = 0; i < n_out; i++) cnt[o_row[i] + 1]++;
            for (size_t i = 0; i < nb; i++) cnt[i + 1] += cnt[i];
            lst.resize(n_out);                              // (col, matches)
            fill.assign(cnt.begin(), cnt.end() - 1);
            for (u64 i = 0; i < n_out; i++) lst[fill[o_row[i]]++] = {o_col[i], o_mm[i] >> 16};
            decide_block(kc, reps, pos, nb, R, cnt.data(), lst.data());
            pos = end;
        }
        Trace t_fin("3.group.finish");
        finish_group(g, reps);
    });
    }
    for (u32 g : gidx) groups[g] = std::move(group_out[g]);
    Trace t_rc("3.recluster.total");
    if (pre) {
        Groups snapshot = groups;                                              // test hook: the clusters before reclustering, of all ranks
        if (by_group) gather_groups(snapshot);
        pre->clear(); if (pre_group) pre_group->clear();
        for (auto& kv : snapshot) for (auto& cl : kv.second) { pre->push_back(cl); if (pre_group) pre_group->push_back(kv.first); }
    }
    // recluster_using_consensus_reps :1272-1433
    Groups settled;
    u32 iteration = 0;
    while (true) {
        if (iteration >= args.max_iterations_recluster) break;                // :1296
        iteration++;
        u64 total_merges = groups.empty() ? 0 : recluster_iteration(rs, tw, groups, settled, args);
        if (by_group) {                                                        // the exit test is over ALL groups: a rank whose groups have settled keeps meeting the others
            std::vector<u64> all(sh_world, 0);
            chk(rs.ctx, svt_shard_allgather_u64(rs.ctx, total_merges, all.data()), "svt_shard_allgather_u64");
            total_merges = 0; for (u64 x : all) total_merges += x;
        }
        if (total_merges == 0) break;                                          // :1367
    }
    for (auto& kv : settled) groups[kv.first] = std::move(kv.second);
    if (by_group) gather_groups(groups);
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
    return std::min<u32>(std::max((mx + 12) / 13, df), 511);
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

// Low-polymorphism mode (src/alignment.rs:1527-1719, refine_asv_depths_with_minimap2): every read against ALL ASVs.  One K7
// launch over reads x ASVs (a pair is a hit when it shares a minimizer; the vote gives the strand), one K8 launch on the hits, the
// the ASV with the strictly lowest NM is the read's class; counters and EM as in the SNPmer path.  `mapq > 0` (:1579-1581): minimap2
// sets a primary's mapq to 0 when its DP score is not strictly above the second-best target's (mm_set_mapq) and secondary hits carry
// mapq 0, so a read that several ASVs fit equally well has no valid hit; under the K8 contract: a tie at the lowest NM drops the read.
// Stage 7 `nm` (src/alignment.rs:1848-1862) under the selected contract: K8, K8a near the unit-cost optimum, or K8a in the whole band
// (tuning.nm_contract)
static int stage7_nm(const ReadSet& rs, svt_batch* asvs, const ClusterArgs& args, const u32* qi, const u32* ti, const u8* rev, const u32* band, size_t n, int32_t* nm) {
    if (args.tuning.nm_contract == 1) return svt_align_nm_affine_near(rs.ctx, asvs, rs.batch, qi, ti, rev, band, n, nm, nullptr, nullptr);
    if (args.tuning.nm_contract == 2) return svt_align_nm_affine(rs.ctx, asvs, rs.batch, qi, ti, rev, band, n, nm, nullptr);
    return svt_align_nm(rs.ctx, asvs, rs.batch, qi, ti, rev, band, n, nm);
}
static void em_read_classes_all_vs_all(const ReadSet& rs, const TwinReads& tw, svt_batch* asvs, const std::vector<u64>& asv_off, const ClusterArgs& args, size_t lo, size_t hi, EmResult& em) {
    const size_t na = asv_off.size() - 1;
    chk(rs.ctx, svt_extract_seeds(rs.ctx, asvs, args.kmer_size, args.c, args.minimum_base_quality, 0), "svt_extract_seeds(asvs)");
    const size_t RB = std::max<size_t>(1, ((size_t)4 << 20) / std::max<size_t>(1, na));      // reads per slab: <= 4M pairs in flight
    for (size_t r0 = lo; r0 < hi; r0 += RB) {
        const size_t r1 = std::min(hi, r0 + RB), np = (r1 - r0) * na;
        std::vector<u32> pa(np), pb(np), shared(np), same(np);
        for (size_t r = r0; r < r1; r++) for (size_t a = 0; a < na; a++) { pa[(r - r0) * na + a] = tw.orig[r]; pb[(r - r0) * na + a] = (u32)a; }
        chk(rs.ctx, svt_minimizer_shared_counts(rs.ctx, rs.batch, asvs, pa.data(), pb.data(), np, shared.data(), same.data()), "svt_minimizer_shared_counts(low polymorphism)");
        std::vector<u32> qi, ti, band, src; std::vector<u8> rev;
        for (size_t i = 0; i < np; i++) if (shared[i]) {
            const size_t r = r0 + i / na, a = i % na;
            qi.push_back((u32)a); ti.push_back(tw.orig[r]); rev.push_back((shared[i] - same[i]) > same[i] ? 1 : 0);
            band.push_back(band_for(args, (u32)(asv_off[a + 1] - asv_off[a]), tw.length[r])); src.push_back((u32)i);
        }
        std::vector<int32_t> nm(qi.size());
        if (!qi.empty()) chk(rs.ctx, stage7_nm(rs, asvs, args, qi.data(), ti.data(), rev.data(), band.data(), qi.size(), nm.data()), "svt_align_nm(low polymorphism)");
        size_t x = 0;
        for (size_t r = r0; r < r1; r++) {
            const size_t xb = x; int32_t best_nm = INT32_MAX;
            for (; x < src.size() && src[x] / na == r - r0; x++) if (nm[x] != INT32_MAX) best_nm = std::min(best_nm, nm[x]);
            std::vector<u32> cls;
            for (size_t y = xb; y < x; y++) if (nm[y] != INT32_MAX && nm[y] == best_nm) cls.push_back(qi[y]);              // ascending ASV (:1599)
            if (cls.empty()) continue;                                                                                  // :1584-1587 (counted by em_finish)
            if (cls.size() > 1) continue;      // `mapq > 0` (:1579-1581): minimap2 gives mapq 0 to a primary whose DP score is not strictly above the second-best target's -> dropped
            em.read_n_best[r] = (u32)cls.size(); em.read_first[r] = cls[0]; em.read_nm[r] = best_nm; em.read_class[r] = cls;
            if (em.keep_mappings) { em.read_lines[r].clear(); for (u32 a : cls) em.read_lines[r].push_back(EmResult::MapLine{a, 0, best_nm}); }   // :1604-1608
        }
    }
}

void em_init(const TwinReads& tw, size_t na, EmResult& em, bool keep_mappings) {
    const size_t nr = tw.n;
    // the per-read class lists of the result this one replaces keep their storage (a pipeline runs Stage 7 once per step on ~10^5 reads: 10^5 frees and
    // 10^5 allocations of a few words each per step otherwise)
    std::vector<std::vector<u32>> lists = std::move(em.read_class);
    em = EmResult();
    em.keep_mappings = keep_mappings;
    if (keep_mappings) em.read_lines.assign(nr, {});
    em.depth.assign(na, 0); em.unambig.assign(na, 0); em.ambig.assign(na, 0); em.leq10.assign(na, 0);
    em.read_n_best.assign(nr, 0); em.read_first.assign(nr, 0); em.read_nm.assign(nr, -1);
    lists.resize(nr);
    for (auto& l : lists) l.clear();
    em.read_class = std::move(lists);
}

// the per-read part of refine_asv_depths_with_em (src/alignment.rs:1786-1896) for the twin reads [lo, hi): independent per read
void em_read_classes(const ReadSet& rs, const TwinReads& tw, svt_batch* asvs, const std::vector<u64>& asv_off, const ClusterArgs& args, size_t lo, size_t hi, EmResult& em) {
    const size_t na = asv_off.size() - 1;
    hi = std::min<size_t>(hi, tw.n); if (lo >= hi || na == 0) return;
    const size_t nr = hi - lo;
    const u32 k = args.kmer_size;
    Trace t_all7("7.classes");
    if (args.low_polymorphism) { em_read_classes_all_vs_all(rs, tw, asvs, asv_off, args, lo, hi, em); return; }   // :1730-1732
    // ASV twin reads: kmer_comp::twin_reads_from_fasta (src/kmer_comp.rs:39-66): qualities None, no filtering
    { Trace t_("7.asv_seeds"); chk(rs.ctx, svt_extract_seeds(rs.ctx, asvs, k, args.c, args.minimum_base_quality, 0), "svt_extract_seeds(asvs)"); }
    // K6: candidates = ASVs sharing >= 1 SNPmer site with the read (find_compatible_candidates keys, :1791)
    std::vector<u32> rows(nr);
    for (size_t i = 0; i < nr; i++) rows[i] = tw.orig[lo + i];
    // exact device-side pre-filter: a pair survives :1829-1833 only if mism / minimizer_matches / c <= 0.005, and
    // minimizer_matches <= |read minimizer set|, so mism / |set| / c > 0.005 already decides it (f64 division is monotone)
    std::vector<u32> max_mism(nr);
    for (size_t i = 0; i < nr; i++) {
        const double nu = (double)tw.n_unique[lo + i];
        u32 m = (u32)(0.0050 * (double)args.c * nu) + 2;
        while (m > 0 && (double)m / nu / (double)args.c > 0.0050) m--;
        max_mism[i] = tw.n_unique[lo + i] ? m : 0;
    }
    // K6 -> K7 -> f64 filters -> per-read lowest-mismatch ties, on device-resident lists (svt_read_asv_ties); only the ties come back
    struct Tie { u32 read, asv; u8 rev; u32 band; u32 mm; };                   // read = position in [lo, hi); mm = SNPmer mismatches (temp dump only)
    std::vector<Tie> ties; std::vector<size_t> tie_off(nr + 1, 0);
    {
        Trace t_("7.ties");
        const double minfrac = std::pow(0.950, (int)k);                       // powi :1806
        std::vector<u32> t_row, t_col, t_mm; std::vector<u8> t_rev;
        u64 n_t = 0, n_cand = 0, cap = std::max<u64>(4096, (u64)nr * 3);
        while (true) {
            t_row.resize(cap); t_col.resize(cap); t_rev.resize(cap); if (em.keep_mappings) t_mm.resize(cap);
            int rc = svt_read_asv_ties(rs.ctx, rs.batch, rows.data(), (u32)nr, asvs, (u32)na, max_mism.data(), minfrac, (double)args.c,
                                       t_row.data(), t_col.data(), t_rev.data(), em.keep_mappings ? t_mm.data() : nullptr, cap, &n_t, &n_cand);
            if (rc == SVT_ERR_OVERFLOW) { cap = n_t + 1024; continue; }
            chk(rs.ctx, rc, "svt_read_asv_ties");
            break;
        }
        // read order, ascending ASV inside a read (deterministic stand-in for FxHashMap iteration order; the only ties it could
        // affect are removed by the sort at :1892): counting sort by read, then a small sort per read
        std::vector<u64> start(nr + 1, 0);
        for (u64 i = 0; i < n_t; i++) start[t_row[i] + 1]++;
        for (size_t r = 0; r < nr; r++) start[r + 1] += start[r];
        std::vector<u64> fill(start.begin(), start.end() - 1);
        ties.resize(n_t);
        for (u64 i = 0; i < n_t; i++) ties[fill[t_row[i]]++] = Tie{t_row[i], t_col[i], t_rev[i], 0, em.keep_mappings ? t_mm[i] : 0};
        for (size_t r = 0; r < nr; r++) {
            if (start[r + 1] - start[r] > 1) std::sort(ties.begin() + start[r], ties.begin() + start[r + 1], [](const Tie& x, const Tie& y) { return x.asv < y.asv; });
            for (u64 i = start[r]; i < start[r + 1]; i++) ties[i].band = band_for(args, (u32)(asv_off[ties[i].asv + 1] - asv_off[ties[i].asv]), tw.length[lo + r]);
            tie_off[r] = start[r];
        }
        tie_off[nr] = n_t;
        if (trace_enabled()) fprintf(stderr, "[savont-trace] stage7: %zu reads x %zu ASVs, %llu candidate pairs after K6, %zu tied pairs to K8\n", nr, na, (unsigned long long)n_cand, ties.size());
    }
    Trace t_host7("7.host_k8+prep");
    // K8
    std::vector<u32> qi(ties.size()), ti(ties.size()), band(ties.size()); std::vector<u8> rev(ties.size()); std::vector<int32_t> nm(ties.size());
    for (size_t i = 0; i < ties.size(); i++) { qi[i] = ties[i].asv; ti[i] = tw.orig[lo + ties[i].read]; rev[i] = ties[i].rev; band[i] = ties[i].band; }
    if (!ties.empty()) { Trace t_("7.k8"); chk(rs.ctx, stage7_nm(rs, asvs, args, qi.data(), ti.data(), rev.data(), band.data(), ties.size(), nm.data()), "svt_align_nm"); }
    t_host7.~Trace(); new (&t_host7) Trace("7.host_classes");
    // per read: the ties at the best NM are its class (independent per read)
    const size_t n_parts = std::max<size_t>(1, std::min<size_t>(WorkerPool::get().size(), nr / 4096));
    par_for(n_parts, [&](size_t pi) {
        std::vector<u32> cls;
        for (size_t r = nr * pi / n_parts; r < nr * (pi + 1) / n_parts; r++) {
            int32_t best_nm = INT32_MAX;
            for (size_t i = tie_off[r]; i < tie_off[r + 1]; i++) if (nm[i] != INT32_MAX) best_nm = std::min(best_nm, nm[i]);   // empty mapping -> skipped (:1859-1861)
            if (em.keep_mappings) {                                                                           // best_alns in ascending nm, first five (:1865, :1877)
                std::vector<EmResult::MapLine>& ln = em.read_lines[lo + r];
                ln.clear();
                for (size_t i = tie_off[r]; i < tie_off[r + 1]; i++) if (nm[i] != INT32_MAX) ln.push_back(EmResult::MapLine{ties[i].asv, ties[i].mm, nm[i]});
                std::stable_sort(ln.begin(), ln.end(), [](const EmResult::MapLine& x, const EmResult::MapLine& y) { return x.b < y.b; });
                if (ln.size() > 5) ln.resize(5);
            }
            cls.clear();
            for (size_t i = tie_off[r]; i < tie_off[r + 1]; i++) if (nm[i] != INT32_MAX && nm[i] == best_nm) cls.push_back(ties[i].asv);
            if (cls.empty()) continue;                                                                        // :1817-1837, :1921-1924 (counted by em_finish)
            std::sort(cls.begin(), cls.end());                                                                // :1892
            em.read_n_best[lo + r] = (u32)cls.size(); em.read_first[lo + r] = cls[0]; em.read_nm[lo + r] = best_nm; em.read_class[lo + r].assign(cls.begin(), cls.end());
        }
    });
}

// counters, equivalence classes and EM from the per-read classes (src/alignment.rs:1898-2031); the counters and class counts are
// sums over reads, so the reads are folded in parallel slices and the slices added up in order
void em_finish(const TwinReads& tw, size_t na, EmResult& em) {
    const size_t nr = tw.n;
    Trace t_("7.host_eq_em");
    em.depth.assign(na, 0); em.unambig.assign(na, 0); em.ambig.assign(na, 0); em.leq10.assign(na, 0); em.filtered = 0; em.total_assigned = 0; em.kept_original = false;
    if (na == 0 || nr == 0) { em.kept_original = true; return; }
    struct Part { std::map<std::vector<u32>, u64> eq; std::vector<u64> unambig, ambig, leq10; u64 filtered = 0, assigned = 0; };
    const size_t n_parts = std::max<size_t>(1, std::min<size_t>(WorkerPool::get().size(), nr / 4096));
    std::vector<Part> parts(n_parts);
    par_for(n_parts, [&](size_t pi) {
        Part& P = parts[pi];
        P.unambig.assign(na, 0); P.ambig.assign(na, 0); P.leq10.assign(na, 0);
        u64* last = nullptr; const std::vector<u32>* last_key = nullptr;              // reads come cluster by cluster: most repeat the class of the read before (one vector compare instead of a walk down the map)
        for (size_t r = nr * pi / n_parts; r < nr * (pi + 1) / n_parts; r++) {
            const std::vector<u32>& cls = em.read_class[r];
            if (cls.empty()) { P.filtered++; continue; }
            if (cls.size() == 1) P.unambig[cls[0]]++; else for (u32 a : cls) P.ambig[a]++;                   // :1898-1908
            if (em.read_nm[r] <= 10) for (u32 a : cls) P.leq10[a]++;                                          // :1910-1915
            if (last && *last_key == cls) ++*last;
            else { auto it = P.eq.try_emplace(cls, 0).first; last = &it->second; last_key = &it->first; ++*last; }
            P.assigned++;
        }
    });
    std::map<std::vector<u32>, u64> eq;
    for (Part& P : parts) {
        for (auto& kv : P.eq) eq[kv.first] += kv.second;
        for (size_t a = 0; a < na; a++) { em.unambig[a] += P.unambig[a]; em.ambig[a] += P.ambig[a]; em.leq10[a] += P.leq10[a]; }
        em.filtered += P.filtered; em.total_assigned += P.assigned;
    }
    if (eq.empty()) { em.kept_original = true; return; }                                                      // :1952-1955 / :1643-1646
    std::vector<double> ab; run_em(eq, em.total_assigned, na, ab);
    for (size_t a = 0; a < na; a++) em.depth[a] = (u64)std::llround(ab[a] * (double)em.total_assigned);       // :2015 / :1700
}

void refine_asv_depths_with_em(const ReadSet& rs, const TwinReads& tw, svt_batch* asvs, const std::vector<u64>& asv_off, const ClusterArgs& args, bool keep_mappings, EmResult& em) {
    const size_t na = asv_off.size() - 1;
    Trace t_all7("7.total");
    em_init(tw, na, em, keep_mappings);                                       // `em` may hold the result of an earlier step: its storage is reused
    if (na == 0 || tw.n == 0) { em.kept_original = true; return; }
    em_read_classes(rs, tw, asvs, asv_off, args, 0, tw.n, em);
    em_finish(tw, na, em);
}
EmResult refine_asv_depths_with_em(const ReadSet& rs, const TwinReads& tw, svt_batch* asvs, const std::vector<u64>& asv_off, const ClusterArgs& args, bool keep_mappings) {
    EmResult em;
    refine_asv_depths_with_em(rs, tw, asvs, asv_off, args, keep_mappings, em);
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
