// consensus.cpp -- Stage 4 of `savont asv` above the C-ABI: POA consensus per cluster, pile-ups of the
// cluster's reads against it, quality -> error-rate map, Bayesian per-position confidence, end trimming / masking and the
// low-quality split.  Reference: src/alignment.rs:233-412 (align_and_consensus), :416-659 (generate_consensus_pileups),
// :663-786 (estimate_quality_error_rates), :864-1160 (analyze_pileup_consensuses).
//
// What runs where: orientation votes (K7) and every read-vs-consensus alignment with traceback (K9, the reference's minimap2
// map-ont + CIGAR walk) are GPU calls batched over ALL clusters; the POA itself is CPU work, one cluster per host thread, as it
// is in the reference (spoars under rayon).  --use-hpc (src/cli.rs:118-120): the reads are homopolymer-compressed before the POA and before the pile-ups, see hpc_with_quality below.
// Third-party pieces that cannot be pinned (spoars POA, minimap2 strand / CIGAR) are restated: see poa.hpp and DESIGN.md 7.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <ctime>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>

#include "asv_pipeline.hpp"
#include "poa.hpp"
#include "worker_pool.hpp"

namespace savont {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

static void chk4(svt_ctx* c, int rc, const char* what) {
    if (rc != SVT_OK) throw Error{rc, std::string(what) + ": " + svt_last_error(c)};
}
static u32 band_of(const ClusterArgs& args, u32 n, u32 m) {
    if (args.align_band) return args.align_band;
    const u32 mx = std::max(n, m), df = n > m ? n - m : m - n;
    return std::min<u32>(std::max((mx + 12) / 13, df), 511);
}
// qual_seq of the reads (4-bit bins, src/types.rs:447-467): computed on the GPU by svt_extract_seeds(use_qual=1), fetched once
void ensure_qualbins(const ReadSet& rs) {
    if (!rs.qualbin_off.empty()) return;
    u64 nm = 0, ns = 0, nq = 0;
    chk4(rs.ctx, svt_seeds_sizes(rs.ctx, rs.batch, &nm, &ns, &nq), "svt_seeds_sizes");
    rs.qualbin_off.assign(rs.n + 1, 0); rs.qualbins.assign(nq, 0);
    if (nq == 0) return;
    svt_seeds_out o; memset(&o, 0, sizeof o);
    o.qualbin_off = rs.qualbin_off.data(); o.qualbins = rs.qualbins.data();
    chk4(rs.ctx, svt_seeds_fetch(rs.ctx, rs.batch, &o), "svt_seeds_fetch(qualbins)");
}

// dna_seq of a TwinRead decoded to upper-case ACGT (non-ACGT -> A, src/seeding.rs:604-626), optionally reverse-complemented: two 256-entry
// tables, no branches (12 MB of bases per 100k-read sample pass through here)
namespace {
struct BaseTables {
    u8 fwd[256], rc[256];
    BaseTables() {
        for (int c = 0; c < 256; c++) {
            u8 b = (u8)(c & 0xDF);
            if (b == 'U') b = 'T';
            if (b != 'A' && b != 'C' && b != 'G' && b != 'T') b = 'A';
            fwd[c] = b; rc[c] = (b == 'A') ? 'T' : (b == 'C') ? 'G' : (b == 'G') ? 'C' : 'A';
        }
    }
};
const BaseTables g_base_tables;
}  // namespace
static std::vector<u8> read_seq(const ReadSet& rs, u32 orig, bool rc) {
    const u64 o = rs.offsets[orig], len = rs.offsets[orig + 1] - o;
    std::vector<u8> s(len);
    const u8* src = rs.host_seq.data() + o;
    if (!rc) for (u64 i = 0; i < len; i++) s[i] = g_base_tables.fwd[src[i]];
    else for (u64 i = 0; i < len; i++) s[i] = g_base_tables.rc[src[len - 1 - i]];
    return s;
}
// qual_seq decoded (bin*3+33) and expanded x4 to the read length (src/alignment.rs:248-273)
static std::vector<u8> read_qual(const ReadSet& rs, u32 orig, bool rc) {
    const u64 len = rs.offsets[orig + 1] - rs.offsets[orig];
    std::vector<u8> q(len, 33);
    if (!rs.qualbins.empty()) {
        const u8* qb = rs.qualbins.data() + rs.qualbin_off[orig];
        const u64 nb = len / 4;
        for (u64 b = 0; b < nb; b++) { const u8 v = (u8)(((qb[b >> 1] >> (4 * (b & 1))) & 15) * 3 + 33); q[4 * b] = v; q[4 * b + 1] = v; q[4 * b + 2] = v; q[4 * b + 3] = v; }
        for (u64 i = 4 * nb; i < len; i++) { const u64 bin = i >> 2; q[i] = (u8)(((qb[bin >> 1] >> (4 * (bin & 1))) & 15) * 3 + 33); }
    }
    if (rc) std::reverse(q.begin(), q.end());
    return q;
}

// utils::homopolymer_compress_with_quality (src/utils.rs:136-190): one base per run (runs capped at 255), its minimum quality, its length
static void hpc_with_quality(const std::vector<u8>& s, const std::vector<u8>& q, std::vector<u8>& os, std::vector<u8>& oq, std::vector<u8>& ol) {
    os.clear(); oq.clear(); ol.clear();
    if (s.empty() || s.size() != q.size()) return;                              // :137-139
    u8 cur = s[0], mq = q[0]; u32 run = 1;
    for (size_t i = 1; i < s.size(); i++) {
        if (s[i] == cur && run < 255) { run++; mq = std::min(mq, q[i]); }
        else { os.push_back(cur); oq.push_back(mq); ol.push_back((u8)run); cur = s[i]; run = 1; mq = q[i]; }
    }
    os.push_back(cur); oq.push_back(mq); ol.push_back((u8)run);
}
// utils::homopolymer_compress (src/utils.rs:70-109), sequence only
static std::vector<u8> hpc(const std::vector<u8>& s) {
    std::vector<u8> o;
    if (s.empty()) return o;
    u8 cur = s[0]; u32 run = 1;
    for (size_t i = 1; i < s.size(); i++) { if (s[i] == cur && run < 255) run++; else { o.push_back(cur); cur = s[i]; run = 1; } }
    o.push_back(cur);
    return o;
}

// DP matrices (tens of MB, touched once per alignment) are leased from a process-wide pool: fresh allocations per cluster cost
// more in page faults than the DP itself when ~100 clusters run on ~100 host threads.
namespace {
struct ScratchPool { std::mutex m; std::vector<std::vector<int>*> free; } g_scratch;
struct ScratchLease {
    std::vector<int>* buf;
    ScratchLease() { std::lock_guard<std::mutex> l(g_scratch.m); if (g_scratch.free.empty()) buf = new std::vector<int>(); else { buf = g_scratch.free.back(); g_scratch.free.pop_back(); } }
    ~ScratchLease() { std::lock_guard<std::mutex> l(g_scratch.m); g_scratch.free.push_back(buf); }
};
}  // namespace

static std::atomic<u64> g_poa_cells{0}, g_poa_rows{0}, g_poa_maxdev{0}, g_poa_n{0};
// generate_consensus_poa, src/alignment.rs:193-231
std::vector<u8> poa_consensus(const std::vector<std::vector<u8>>& seqs, const std::vector<std::vector<u8>>& quals, u64* graph_nodes, bool wide_cells, bool no_band) {
    if (graph_nodes) *graph_nodes = 0;
    if (seqs.empty()) return {};
    size_t tot = 0; for (auto& s : seqs) tot += s.size();
    const size_t ref_len = tot / seqs.size();                                   // :211
    u32 max_dev = 0; for (auto& s : seqs) max_dev = std::max<u32>(max_dev, (u32)std::llabs((long long)ref_len - (long long)s.size()));
    if (no_band) for (auto& s : seqs) max_dev = std::max<u32>(max_dev, (u32)s.size() + 1);       // :217 --no-band: spoa's unbanded engine = a band that holds every column (the 32-bit DP takes it)
    PoaGraph g; g.wide_cells = wide_cells;
    ScratchLease lease; g.use_scratch(lease.buf);
    for (size_t i = 0; i < seqs.size(); i++) {
        std::vector<u32> w(quals[i].begin(), quals[i].end());
        PoaGraph::Alignment al = g.align(seqs[i], max_dev, 0.1);               // BandConfig{base: max_deviation, frac: 0.1} :220
        g.add_alignment(al, seqs[i], w);
    }
    if (trace_enabled()) { g_poa_cells += g.cells_done; g_poa_rows += g.rows_done; g_poa_maxdev += max_dev; g_poa_n++; }
    if (graph_nodes) *graph_nodes = g.nodes.size();
    return g.consensus();
}

// generate_consensus_poa for all clusters in ONE launch with the graphs resident on the device (K12, svt_poa_graphs): the host packs the
// reads, the kernel aligns / fuses / keeps its topological order for every read of every cluster, the final graphs come back and
// PoaGraph::consensus() reads the heaviest bundle.  A cluster outside the kernel's limits, or one the kernel gave up on (status != 0:
// an end-cell tie it cannot order the way spoa's sort would, a capacity), runs on the host DP -- same results either way.
// device_share (percent): that share of the clusters goes to the device, the host DP works on the others WHILE the launch runs (the chain of a
// 75-read cluster takes the device ~130 ms whatever the number of clusters, a host core ~9 ms: alone the host pool is faster, a busy pool is not).
static std::vector<std::vector<u8>> poa_consensus_resident(svt_ctx* ctx, const std::vector<PoaInput>& in, bool wide_cells, std::vector<u64>* graph_nodes, int device_share) {
    const size_t n = in.size();
    std::vector<std::vector<u8>> out(n);
    if (graph_nodes) graph_nodes->assign(n, 0);
    std::vector<u32> dev, host;
    std::vector<u32> max_dev(n, 0);
    for (size_t i = 0; i < n; i++) {
        const auto& seqs = in[i].seqs;
        if (seqs.empty()) continue;
        size_t tot = 0, longest = 0; for (auto& s : seqs) { tot += s.size(); longest = std::max(longest, s.size()); }
        const size_t ref_len = tot / seqs.size();                                // :211
        for (auto& s : seqs) max_dev[i] = std::max<u32>(max_dev[i], (u32)std::llabs((long long)ref_len - (long long)s.size()));
        const u64 bw = (u64)max_dev[i] + (u64)(0.1 * (double)longest) + 1;
        const bool fits = longest <= 5440 && bw <= 640;
        const bool mine = device_share >= 100 || (int)((i * 37u) % 100u) < device_share;          // a fixed, spread-out subset
        (fits && mine ? dev : host).push_back((u32)i);
    }
    const size_t n_host_first = host.size();
    const double k0 = trace_cpu_now();
    std::vector<u64> cl_off(dev.size() + 1, 0), seq_off(1, 0); std::vector<u32> band;
    for (size_t x = 0; x < dev.size(); x++) {
        const auto& seqs = in[dev[x]].seqs;
        cl_off[x + 1] = cl_off[x] + seqs.size();
        for (auto& s : seqs) { seq_off.push_back(seq_off.back() + s.size()); band.push_back(max_dev[dev[x]] + (u32)(int)(0.1 * (double)s.size()) + 1u); }   // BandConfig{base, frac: 0.1} :220
    }
    std::vector<u8> seq(seq_off.back() + 1), wts(seq_off.back() + 1);
    par_for(dev.size(), [&](size_t x) {
        const PoaInput& pi = in[dev[x]];
        for (size_t r = 0; r < pi.seqs.size(); r++) {
            const u64 o = seq_off[cl_off[x] + r];
            if (!pi.seqs[r].empty()) { memcpy(seq.data() + o, pi.seqs[r].data(), pi.seqs[r].size()); memcpy(wts.data() + o, pi.quals[r].data(), pi.seqs[r].size()); }
        }
    });
    std::vector<svt_poa_result> res(dev.size()); std::vector<u64> node_off(dev.size() + 1, 0), edge_off(dev.size() + 1, 0);
    std::vector<u8> code; std::vector<uint16_t> al; std::vector<u32> ed;
    const double k1 = trace_cpu_now();
    if (!dev.empty()) chk4(ctx, svt_poa_graphs_submit(ctx, (u32)dev.size(), cl_off.data(), seq_off.data(), seq.data(), wts.data(), band.data()), "svt_poa_graphs_submit");
    // the host engine's clusters while the launch runs
    par_for(n_host_first, [&](size_t t) { const u32 i = host[t]; u64 gn = 0; out[i] = poa_consensus(in[i].seqs, in[i].quals, &gn, wide_cells); if (graph_nodes) (*graph_nodes)[i] = gn; });
    std::vector<u64> cons_off(dev.size() + 1, 0); std::vector<u8> cons_dev;
    bool need_graphs = false;                                                    // a graph whose consensus the device left to the host (too large for its LDS)
    if (!dev.empty()) {
        chk4(ctx, svt_poa_graphs_wait(ctx, res.data(), node_off.data(), edge_off.data()), "svt_poa_graphs_wait");
        u64 tot = 0;
        for (size_t x = 0; x < dev.size(); x++) if (res[x].status == 0) { if (res[x].cons_len == 0xFFFFFFFFu) need_graphs = true; else tot += res[x].cons_len; }
        cons_dev.resize(tot + 1);
        chk4(ctx, svt_poa_consensus_fetch(ctx, res.data(), cons_off.data(), cons_dev.data()), "svt_poa_consensus_fetch");
        if (need_graphs) { code.resize(node_off.back() + 1); al.resize(node_off.back() * 8 + 8); ed.resize(edge_off.back() * 3 + 3); }
        chk4(ctx, svt_poa_graphs_fetch(ctx, need_graphs ? code.data() : nullptr, need_graphs ? al.data() : nullptr, need_graphs ? ed.data() : nullptr), "svt_poa_graphs_fetch");
    }
    const double k2 = trace_cpu_now();
    u64 gave_up = 0, ties = 0, rows = 0, far = 0; u32 why[16] = {0}; u64 tk[6] = {0, 0, 0, 0, 0, 0}, tk_max = 0; size_t slowest = 0;
    for (size_t x = 0; x < dev.size(); x++) { u64 t = 0; for (int k = 0; k < 6; k++) t += res[x].ticks[k]; if (t > tk_max) { tk_max = t; slowest = x; for (int k = 0; k < 6; k++) tk[k] = res[x].ticks[k]; } }
    for (size_t x = 0; x < dev.size(); x++) { ties += res[x].tie_reads; rows += res[x].rows_done; far += res[x].far_rows; if (res[x].status != 0) { host.push_back(dev[x]); gave_up++; why[res[x].status & 15]++; } }
    // graphs -> consensus on the pool; the clusters the kernel handed back ride along as tasks of the same loop
    par_for(dev.size() + (host.size() - n_host_first), [&](size_t t) {
        if (t >= dev.size()) { const u32 i = host[n_host_first + t - dev.size()]; u64 gn = 0; out[i] = poa_consensus(in[i].seqs, in[i].quals, &gn, wide_cells); if (graph_nodes) (*graph_nodes)[i] = gn; return; }
        if (res[t].status != 0) return;
        if (graph_nodes) (*graph_nodes)[dev[t]] = res[t].n_nodes;
        if (res[t].cons_len != 0xFFFFFFFFu) { out[dev[t]].assign(cons_dev.begin() + cons_off[t], cons_dev.begin() + cons_off[t + 1]); return; }   // K12c
        PoaGraph g;
        g.import_graph(code.data() + node_off[t], al.data() + node_off[t] * 8, res[t].n_nodes, ed.data() + edge_off[t] * 3, res[t].n_edges);
        out[dev[t]] = g.consensus();
    });
    if (trace_enabled()) {
        const double k3 = trace_cpu_now();
        fprintf(stderr, "[savont-trace] poa resident: %zu clusters on the device (%llu handed back: nodes %u edges %u aligned %u spill %u preds %u tie %u wait %u sweep %u far-list %u lds %u), %zu on the host; %llu k rows, %llu tie reads, %llu far rows; CPU seconds: pack %.3f launch+fetch %.3f consensus %.3f; slowest cluster, ms: descriptors %.2f DP %.2f end cell %.2f traceback %.2f fuse %.2f order %.2f\n",
                dev.size(), (unsigned long long)gave_up, why[1], why[2], why[3], why[4], why[5] + why[9], why[6], why[10], why[11], why[12], why[13], n_host_first, (unsigned long long)(rows / 1000), (unsigned long long)ties, (unsigned long long)far, k1 - k0, k2 - k1, k3 - k2, tk[0] * 1e-5, tk[1] * 1e-5, tk[2] * 1e-5, tk[3] * 1e-5, tk[4] * 1e-5, tk[5] * 1e-5);
        if (!dev.empty()) { fprintf(stderr, "[savont-trace] poa resident, slowest cluster, per wave tasks / not-ready polls:"); for (int k = 0; k < 8; k++) fprintf(stderr, " %u/%u", res[slowest].tasks[k], res[slowest].spins[k]); fprintf(stderr, "\n"); }
    }
    return out;
}

// The same with the sequences named, not copied: cluster i = the reads refs[i].orig (batch numbering) in the orientations refs[i].rev.  The device gathers
// letters and weights from the resident batch (svt_poa_graphs_submit_reads); only the clusters of the host engine's share, and the ones the kernel hands
// back, are materialised on the host (make_input).  Round 4: building, flattening and uploading 24 MB of sequences per 100k-read step cost ~20 ms of CPU.
struct PoaRefs { std::vector<u32> orig, len; std::vector<u8> rev; };
static std::vector<std::vector<u8>> poa_consensus_resident_refs(svt_ctx* ctx, const svt_batch* batch, const std::vector<PoaRefs>& refs, const std::function<PoaInput(size_t)>& make_input,
                                                                bool wide_cells, int device_share) {
    const size_t n = refs.size();
    std::vector<std::vector<u8>> out(n);
    std::vector<u32> dev, host;
    std::vector<u32> max_dev(n, 0);
    for (size_t i = 0; i < n; i++) {
        const auto& len = refs[i].len;
        if (len.empty()) continue;
        size_t tot = 0, longest = 0; for (u32 l : len) { tot += l; longest = std::max<size_t>(longest, l); }
        const size_t ref_len = tot / len.size();                                 // :211
        for (u32 l : len) max_dev[i] = std::max<u32>(max_dev[i], (u32)std::llabs((long long)ref_len - (long long)l));
        const u64 bw = (u64)max_dev[i] + (u64)(0.1 * (double)longest) + 1;
        const bool fits = longest <= 5440 && bw <= 640;
        const bool mine = device_share >= 100 || (int)((i * 37u) % 100u) < device_share;          // a fixed, spread-out subset
        (fits && mine ? dev : host).push_back((u32)i);
    }
    const size_t n_host_first = host.size();
    const double k0 = trace_cpu_now();
    std::vector<u64> cl_off(dev.size() + 1, 0); std::vector<u32> band, ridx; std::vector<u8> rv;
    for (size_t x = 0; x < dev.size(); x++) {
        const PoaRefs& r = refs[dev[x]];
        cl_off[x + 1] = cl_off[x] + r.len.size();
        for (size_t q = 0; q < r.len.size(); q++) { band.push_back(max_dev[dev[x]] + (u32)(int)(0.1 * (double)r.len[q]) + 1u); ridx.push_back(r.orig[q]); rv.push_back(r.rev[q]); }   // BandConfig{base, frac: 0.1} :220
    }
    std::vector<svt_poa_result> res(dev.size()); std::vector<u64> node_off(dev.size() + 1, 0), edge_off(dev.size() + 1, 0);
    std::vector<u8> code; std::vector<uint16_t> al; std::vector<u32> ed;
    const double k1 = trace_cpu_now();
    if (!dev.empty()) chk4(ctx, svt_poa_graphs_submit_reads(ctx, batch, (u32)dev.size(), cl_off.data(), ridx.data(), rv.data(), band.data()), "svt_poa_graphs_submit_reads");
    // the host engine's clusters while the launch runs
    par_for(n_host_first, [&](size_t t) { const u32 i = host[t]; const PoaInput pi = make_input(i); u64 gn = 0; out[i] = poa_consensus(pi.seqs, pi.quals, &gn, wide_cells); });
    std::vector<u64> cons_off(dev.size() + 1, 0); std::vector<u8> cons_dev;
    bool need_graphs = false;
    if (!dev.empty()) {
        chk4(ctx, svt_poa_graphs_wait(ctx, res.data(), node_off.data(), edge_off.data()), "svt_poa_graphs_wait");
        u64 tot = 0;
        for (size_t x = 0; x < dev.size(); x++) if (res[x].status == 0) { if (res[x].cons_len == 0xFFFFFFFFu) need_graphs = true; else tot += res[x].cons_len; }
        cons_dev.resize(tot + 1);
        chk4(ctx, svt_poa_consensus_fetch(ctx, res.data(), cons_off.data(), cons_dev.data()), "svt_poa_consensus_fetch");
        if (need_graphs) { code.resize(node_off.back() + 1); al.resize(node_off.back() * 8 + 8); ed.resize(edge_off.back() * 3 + 3); }
        chk4(ctx, svt_poa_graphs_fetch(ctx, need_graphs ? code.data() : nullptr, need_graphs ? al.data() : nullptr, need_graphs ? ed.data() : nullptr), "svt_poa_graphs_fetch");
    }
    const double k2 = trace_cpu_now();
    u64 gave_up = 0;
    for (size_t x = 0; x < dev.size(); x++) if (res[x].status != 0) { host.push_back(dev[x]); gave_up++; }
    // graphs -> consensus on the pool; the clusters the kernel handed back ride along as tasks of the same loop
    par_for(dev.size() + (host.size() - n_host_first), [&](size_t t) {
        if (t >= dev.size()) { const u32 i = host[n_host_first + t - dev.size()]; const PoaInput pi = make_input(i); u64 gn = 0; out[i] = poa_consensus(pi.seqs, pi.quals, &gn, wide_cells); return; }
        if (res[t].status != 0) return;
        if (res[t].cons_len != 0xFFFFFFFFu) { out[dev[t]].assign(cons_dev.begin() + cons_off[t], cons_dev.begin() + cons_off[t + 1]); return; }   // K12c
        PoaGraph g;
        g.import_graph(code.data() + node_off[t], al.data() + node_off[t] * 8, res[t].n_nodes, ed.data() + edge_off[t] * 3, res[t].n_edges);
        out[dev[t]] = g.consensus();
    });
    if (trace_enabled()) {
        const double k3 = trace_cpu_now();
        fprintf(stderr, "[savont-trace] poa resident (reads gathered on the device): %zu clusters on the device (%llu handed back), %zu on the host; CPU seconds: lists %.3f launch+fetch %.3f consensus %.3f\n",
                dev.size(), (unsigned long long)gave_up, n_host_first, k1 - k0, k2 - k1, k3 - k2);
    }
    return out;
}

std::vector<std::vector<u8>> poa_consensus_batch(svt_ctx* ctx, const std::vector<PoaInput>& in, int engine, bool wide_cells, std::vector<u64>* graph_nodes, bool no_band) {
    const size_t n = in.size();
    std::vector<std::vector<u8>> out(n);
    if (no_band) engine = 0;                                                    // K12's rings are sized by the band: the unbanded DP is the host engine's
    if (engine < 0) engine = (ctx != nullptr && WorkerPool::get().threads() <= 10) ? 2 : 0;         // auto: measured on MI355X + EPYC 9575F, 100k reads per step: K12 beats the host DP at 2, 4 and 8 CPUs per process (156 / 134 / 102 ms per step against 410 / 264 / 128), loses at 16 (88 against 55)
    if (engine >= 2 && ctx != nullptr) return poa_consensus_resident(ctx, in, wide_cells, graph_nodes, engine == 2 ? 100 : engine - 100);   // 2: all clusters; 100 + s: s percent of them
    if (graph_nodes) graph_nodes->assign(n, 0);
    par_for(n, [&](size_t i) { u64 gn = 0; out[i] = poa_consensus(in[i].seqs, in[i].quals, &gn, wide_cells, no_band); if (graph_nodes) (*graph_nodes)[i] = gn; });
    return out;
}

// ==================================================================================================
// Stage 4a: alignment::align_and_consensus (src/alignment.rs:233-412)
// ==================================================================================================
// Stage 4a for the clusters ci with ci % world == rank (the clusters are independent: src/alignment.rs:241 is a par_iter over them);
// the other entries stay empty.  world = 1: all clusters.  A pooled multi-rank run all-gathers the raw consensuses afterwards.
std::vector<std::vector<u8>> poa_raw_consensuses(const ReadSet& rs, const TwinReads& tw, const std::vector<std::vector<u32>>& all_clusters, const ClusterArgs& args, u32 rank, u32 world) {
    const size_t max_seqs_consensus = 75;                                       // :234
    const size_t nc = all_clusters.size();
    static const std::vector<u32> no_members;
    std::vector<const std::vector<u32>*> mine(nc);
    for (size_t ci = 0; ci < nc; ci++) mine[ci] = (world <= 1 || ci % world == rank) ? &all_clusters[ci] : &no_members;
    struct ClusterView { const std::vector<const std::vector<u32>*>& v; const std::vector<u32>& operator[](size_t i) const { return *v[i]; } } clusters{mine};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    auto t0 = now(); const double c0 = trace_cpu_now();
    ensure_qualbins(rs);
    if (rs.host_seq.empty() && rs.n) throw Error{SVT_ERR_ARG, "align_and_consensus: the ReadSet holds no host copy of the reads"};
    struct Plan { u32 seed; std::vector<u32> picks; };                          // cluster-local indices
    std::vector<Plan> plan(nc);
    auto t1 = now(); const double c1 = trace_cpu_now();
    double acc_of_bin[16];
    for (u32 b = 0; b < 16; b++) acc_of_bin[b] = 1.0 - std::pow(10.0, -((double)(b * 3)) / 10.0);   // :255 per decoded bin quality
    // :254-260 mean of 1 - 10^(-(q-33)/10) over the 4-bit bins, per read: svt_qualbin_mean adds the bins of every read in bin order on the
    // device (the table is evaluated here, by this libm), the values are the fold's bit for bit
    std::vector<double> avg_of(tw.n, 1.0);
    if (!rs.qualbins.empty() && rs.n) {
        std::vector<double> mean(rs.n, 1.0);
        chk4(rs.ctx, svt_qualbin_mean(rs.ctx, rs.batch, acc_of_bin, mean.data()), "svt_qualbin_mean");
        for (size_t t = 0; t < tw.n; t++) avg_of[t] = mean[tw.orig[t]];
    }
    par_for(nc, [&](size_t ci) {
        const std::vector<u32>& cl = clusters[ci];
        const size_t n = cl.size();
        if (n == 0) return;
        std::vector<double> avg(n);
        for (size_t i = 0; i < n; i++) avg[i] = avg_of[cl[i]];
        // :282-288 ask for ONE element of the reads stably sorted by length (the 90th percentile) and for the FIRST 75 of the reads stably sorted by falling mean
        // quality.  A stable sort is the order of the key (value, input index), so the element is an nth_element and the prefix a partial_sort under that key:
        // O(n) and O(n log 75) instead of two full sorts of clusters of 10^4 reads (5.7 ms of CPU per 100k-read step, a twentieth of what a rank has at 2 CPUs)
        std::vector<std::pair<u32, u32>> len_i(n);
        for (size_t i = 0; i < n; i++) len_i[i] = {tw.length[cl[i]], (u32)i};
        const size_t q90 = (size_t)((double)n * 0.9);
        std::nth_element(len_i.begin(), len_i.begin() + q90, len_i.end());        // pairs compare (length, index): the stable order of :282
        std::vector<u32> by_q(n);
        for (size_t i = 0; i < n; i++) by_q[i] = (u32)i;
        const size_t take = std::min(max_seqs_consensus, n);
        std::partial_sort(by_q.begin(), by_q.begin() + take, by_q.end(), [&](u32 a, u32 b) { return avg[a] != avg[b] ? avg[a] > avg[b] : a < b; });   // :286, stable
        plan[ci].seed = len_i[q90].second;                                      // :287 90th-percentile length
        by_q.resize(take);                                                      // :288
        std::sort(by_q.begin(), by_q.end());                                    // mappings.sort_by_key(|k| k.0) :312
        for (u32 i : by_q) if (i != plan[ci].seed) plan[ci].picks.push_back(i);
    });
    std::vector<u32> pa, pb; std::vector<size_t> poff(nc + 1, 0);
    for (size_t ci = 0; ci < nc; ci++) {
        poff[ci] = pa.size();
        for (u32 i : plan[ci].picks) { pa.push_back(tw.orig[clusters[ci][i]]); pb.push_back(tw.orig[clusters[ci][plan[ci].seed]]); }
    }
    poff[nc] = pa.size();
    auto t2 = now(); const double c2 = trace_cpu_now();
    // strand of every picked read relative to its seed (the reference: minimap2 map-ont strand, :291-305) -> K7 vote
    std::vector<u32> shared(pa.size()), same(pa.size());
    if (!pa.empty()) chk4(rs.ctx, svt_minimizer_shared_counts(rs.ctx, rs.batch, rs.batch, pa.data(), pb.data(), pa.size(), shared.data(), same.data()), "svt_minimizer_shared_counts(stage4a)");
    auto t3 = now(); const double c3 = trace_cpu_now();
    // the sequences of a cluster, by reference: seed first (:315), then the picked reads that share a minimizer with it, in their voted orientation
    auto refs_of = [&](size_t ci) -> PoaRefs {
        PoaRefs r;
        const std::vector<u32>& cl = clusters[ci];
        if (cl.empty()) return r;
        auto add = [&](u32 orig, bool rev) { r.orig.push_back(orig); r.rev.push_back(rev ? 1 : 0); r.len.push_back((u32)(rs.offsets[orig + 1] - rs.offsets[orig])); };
        add(tw.orig[cl[plan[ci].seed]], false);
        for (size_t x = 0; x < plan[ci].picks.size(); x++) {
            const size_t pi = poff[ci] + x;
            if (shared[pi] == 0) continue;                                      // no alignment found (:323-326)
            add(tw.orig[cl[plan[ci].picks[x]]], (shared[pi] - same[pi]) > same[pi]);
            if (r.orig.size() > max_seqs_consensus) break;                      // :358
        }
        return r;
    };
    auto input_of = [&](const PoaRefs& r) -> PoaInput {
        PoaInput in;
        for (size_t q = 0; q < r.orig.size(); q++) { in.seqs.push_back(read_seq(rs, r.orig[q], r.rev[q] != 0)); in.quals.push_back(read_qual(rs, r.orig[q], r.rev[q] != 0)); }
        if (args.use_hpc) {                                                     // :363-375 HPC compress all sequences before POA
            std::vector<u8> hs, hq, hl;
            for (size_t i = 0; i < in.seqs.size(); i++) { hpc_with_quality(in.seqs[i], in.quals[i], hs, hq, hl); in.seqs[i] = hs; in.quals[i] = hq; }
        }
        return in;
    };
    int engine = args.tuning.poa_engine == 3 ? 100 + args.tuning.poa_device_share : args.tuning.poa_engine;
    if (args.no_band) engine = 0;                                               // --no-band (:198,217): the host engine, unbanded
    if (engine < 0) engine = (rs.ctx != nullptr && WorkerPool::get().threads() <= 10) ? 2 : 0;        // as poa_consensus_batch decides
    const bool by_reference = engine >= 2 && rs.ctx != nullptr && rs.batch != nullptr && !args.use_hpc && args.tuning.poa_cells != 32;   // the device gathers the reads itself; HPC inputs exist on the host only
    std::vector<std::vector<u8>> cons_all;
    auto t3b = now(); double c3b = trace_cpu_now();
    if (by_reference) {
        std::vector<PoaRefs> refs(nc);
        par_for(nc, [&](size_t ci) { refs[ci] = refs_of(ci); });
        t3b = now(); c3b = trace_cpu_now();
        cons_all = poa_consensus_resident_refs(rs.ctx, rs.batch, refs, [&](size_t ci) { return input_of(refs[ci]); }, false, engine == 2 ? 100 : engine - 100);
    } else {
        std::vector<PoaInput> inputs(nc);
        par_for(nc, [&](size_t ci) { inputs[ci] = input_of(refs_of(ci)); });
        t3b = now(); c3b = trace_cpu_now();
        cons_all = poa_consensus_batch(rs.ctx, inputs, engine, args.tuning.poa_cells == 32, nullptr, args.no_band);
    }
    if (args.use_hpc) for (auto& c : cons_all) c = hpc(c);                      // :383 "compress the consensus again to ensure it's fully HPC"
    auto t4 = now(); const double c4 = trace_cpu_now();
    if (trace_enabled()) {
        trace_add("4a.qualbins", secs(t0, t1), c1 - c0); trace_add("4a.plan", secs(t1, t2), c2 - c1); trace_add("4a.k7", secs(t2, t3), c3 - c2);
        trace_add("4a.inputs", secs(t3, t3b), c3b - c3); trace_add("4a.poa", secs(t3b, t4), c4 - c3b);
        fprintf(stderr, "[savont-trace] poa: %llu clusters, %.1f M cells, %.1f k rows, mean max_dev %.1f\n", (unsigned long long)g_poa_n.load(), g_poa_cells.load() / 1e6, g_poa_rows.load() / 1e3, (double)g_poa_maxdev.load() / std::max<u64>(1, g_poa_n.load()));
        g_poa_cells = 0; g_poa_rows = 0; g_poa_maxdev = 0; g_poa_n = 0;
    }
    return cons_all;
}
// raw consensus per cluster -> the consensus list of align_and_consensus (src/alignment.rs:385-402)
std::vector<ConsensusSequence> assemble_consensuses(const std::vector<std::vector<u32>>& clusters, std::vector<std::vector<u8>> cons_all) {
    std::vector<ConsensusSequence> res;
    for (size_t ci = 0; ci < clusters.size(); ci++) {
        std::vector<u8>& cons = cons_all[ci];
        if (cons.size() < 40) continue;                                         // :385-389
        ConsensusSequence c; c.sequence = std::move(cons); c.depth = clusters[ci].size(); c.id = ci; c.cluster = clusters[ci];
        res.push_back(std::move(c));
    }
    std::stable_sort(res.begin(), res.end(), [](const ConsensusSequence& a, const ConsensusSequence& b) { return a.depth > b.depth; });   // :402
    return res;
}
std::vector<ConsensusSequence> align_and_consensus(const ReadSet& rs, const TwinReads& tw, const std::vector<std::vector<u32>>& clusters, const ClusterArgs& args) {
    return assemble_consensuses(clusters, poa_raw_consensuses(rs, tw, clusters, args, 0, 1));
}

// ==================================================================================================
// Stage 4b-d: alignment::generate_consensus_pileups (src/alignment.rs:416-659) + estimate_quality_error_rates (:663-786) +
// analyze_pileup_consensuses (:864-1160), fused.  All (consensus, read) alignments of all clusters are ONE K9 launch whose
// rows stay in HBM (svt_pileup_create); the per-column work of the two statistics functions -- depth, error fraction,
// per-quality error histogram, the two log-likelihood sums -- is K10 (svt_pileup_stats / svt_pileup_loglik).  The host keeps
// what is sequential or transcendental: the quality -> error-rate map, ln(), log-sum-exp, trimming, masking, the split.
// Returns the low-quality consensuses; `consensuses` keeps the rest.  keep (test hook): host copy of the pile-up entries.
// ==================================================================================================
static double log_sum_exp(double a, double b) {                                 // :789-795
    const double mx = std::max(a, b);
    if (std::isinf(mx) && mx < 0) return -INFINITY;
    return mx + std::log(std::exp(a - mx) + std::exp(b - mx));
}

std::vector<ConsensusSequence> polish_consensuses(const ReadSet& rs, const TwinReads& tw, std::vector<ConsensusSequence>& consensuses, const ClusterArgs& args,
                                                  std::map<u8, double>* qmap_out, Pileups* keep) {
    const size_t nc = consensuses.size();
    if (qmap_out) qmap_out->clear();
    if (keep) { keep->clear(); keep->resize(nc); }
    if (nc == 0) return {};
    ensure_qualbins(rs);
    // Multi-GPU (svt_set_shard): the polish is per cluster like the POA -- pile-ups, column statistics and the Bayesian calls of a cluster need nothing
    // of the others except the quality -> error map, whose inputs are integer histograms: every rank polishes the clusters it owns (largest pile-up first
    // onto the least loaded rank), the histograms are summed over the ranks, the polished sequences are gathered.  Not with the pile-up test hook.
    u32 sh_rank = 0, sh_world = 1;
    svt_shard_info(rs.ctx, &sh_rank, &sh_world);
    const bool by_cluster = sh_world > 1 && keep == nullptr && nc >= 2 * (size_t)sh_world;
    std::vector<u8> mine(nc, 1);
    struct PauseTiles { svt_ctx* c; bool on; int was; ~PauseTiles() { if (on) svt_shard_pause(c, was); } } pause_tiles{rs.ctx, by_cluster, 0};
    if (by_cluster) {
        pause_tiles.was = std::max(0, svt_shard_pause(rs.ctx, 1));             // the ranks make different K7 / K9 / K10 calls below
        std::vector<std::pair<size_t, size_t>> order;                          // (pile-up size, cluster), largest first; ties by index: the same on every rank
        for (size_t ci = 0; ci < nc; ci++) order.push_back({std::min<size_t>(consensuses[ci].cluster.size(), 250) * std::max<size_t>(consensuses[ci].sequence.size(), 1), ci});
        std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        std::vector<u64> load(sh_world, 0);
        for (auto& o : order) { u32 best = 0; for (u32 r = 1; r < sh_world; r++) if (load[r] < load[best]) best = r; load[best] += o.first; mine[o.second] = best == sh_rank; }
    }
    const auto pt0 = std::chrono::steady_clock::now(); const double pc0 = trace_cpu_now();
    std::vector<u8> cseq; std::vector<u64> coff(1, 0);
    for (auto& c : consensuses) { cseq.insert(cseq.end(), c.sequence.begin(), c.sequence.end()); coff.push_back(cseq.size()); }
    svt_batch* cb = nullptr; svt_batch* hb = nullptr; svt_pileup* pile = nullptr;
    chk4(rs.ctx, svt_batch_upload(rs.ctx, cseq.data(), nullptr, coff.data(), (u32)nc, &cb), "svt_batch_upload(consensus)");
    std::vector<u32> depth, err; std::vector<double> lr, ln; std::map<u8, double> qmap;
    try {
        chk4(rs.ctx, svt_extract_seeds(rs.ctx, cb, args.kmer_size, args.c, args.minimum_base_quality, 0), "svt_extract_seeds(consensus)");
        // ---- which reads are piled onto which consensus (:447-451), strand by K7 (the reference: minimap2 map-ont strand)
        std::vector<u32> qi, ti;
        for (size_t ci = 0; ci < nc; ci++) {
            if (!mine[ci]) continue;                                           // multi-GPU: another rank piles up and calls this cluster
            const size_t m = std::min<size_t>(consensuses[ci].cluster.size(), 250);                            // MAX_SEQS_CONSENSUS :421,447
            for (size_t i = 0; i < m; i++) { qi.push_back((u32)ci); ti.push_back(tw.orig[consensuses[ci].cluster[i]]); }
        }
        const size_t np = qi.size();
        // --use-hpc (:480): the piled reads are homopolymer-compressed (minimum quality and length per run) into a batch of their own,
        // pair i = read i of it; K7 votes and K9 aligns that batch, K9 takes qualities and run lengths from its tags
        const svt_batch* tb = rs.batch;
        std::vector<u64> hoff;
        if (args.use_hpc && np) {
            std::vector<std::vector<u8>> hs(np), hq(np), hl(np);
            par_for(np, [&](size_t i) { hpc_with_quality(read_seq(rs, ti[i], false), read_qual(rs, ti[i], false), hs[i], hq[i], hl[i]); });
            hoff.assign(np + 1, 0);
            for (size_t i = 0; i < np; i++) hoff[i + 1] = hoff[i] + hs[i].size();
            std::vector<u8> fs(hoff[np]), fq(hoff[np]), fl(hoff[np]);
            par_for(np, [&](size_t i) { std::copy(hs[i].begin(), hs[i].end(), fs.begin() + hoff[i]); std::copy(hq[i].begin(), hq[i].end(), fq.begin() + hoff[i]); std::copy(hl[i].begin(), hl[i].end(), fl.begin() + hoff[i]); });
            chk4(rs.ctx, svt_batch_upload(rs.ctx, fs.data(), nullptr, hoff.data(), (u32)np, &hb), "svt_batch_upload(hpc reads)");
            chk4(rs.ctx, svt_batch_set_tags(rs.ctx, hb, fq.data(), fl.data()), "svt_batch_set_tags");
            chk4(rs.ctx, svt_extract_seeds(rs.ctx, hb, args.kmer_size, args.c, args.minimum_base_quality, 0), "svt_extract_seeds(hpc reads)");
            tb = hb;
            for (size_t i = 0; i < np; i++) ti[i] = (u32)i;
        }
        auto tlen = [&](u32 t) -> u32 { return hb ? (u32)(hoff[t + 1] - hoff[t]) : (u32)(rs.offsets[t + 1] - rs.offsets[t]); };
        std::vector<u32> shared(np), same(np);
        if (np) chk4(rs.ctx, svt_minimizer_shared_counts(rs.ctx, tb, cb, ti.data(), qi.data(), np, shared.data(), same.data()), "svt_minimizer_shared_counts(stage4b)");
        std::vector<u32> q2, t2, band; std::vector<u8> rev; std::vector<u64> grp_off(nc + 1, 0);
        for (size_t i = 0; i < np; i++) {
            if (shared[i] == 0) continue;                                       // no mapping (:485-486)
            q2.push_back(qi[i]); t2.push_back(ti[i]); rev.push_back((shared[i] - same[i]) > same[i] ? 1 : 0);
            band.push_back(band_of(args, (u32)(coff[qi[i] + 1] - coff[qi[i]]), tlen(ti[i])));
            grp_off[qi[i] + 1]++;
        }
        for (size_t g = 0; g < nc; g++) grp_off[g + 1] += grp_off[g];
        const size_t n2 = q2.size();
        const auto pt1 = std::chrono::steady_clock::now(); const double pc1 = trace_cpu_now();
        std::vector<int32_t> nm(std::max<size_t>(n2, 1));
        chk4(rs.ctx, svt_pileup_create(rs.ctx, cb, tb, q2.data(), t2.data(), rev.data(), band.data(), n2, grp_off.data(), (u32)nc, &pile, nullptr, nm.data()), "svt_pileup_create");
        const auto pt2 = std::chrono::steady_clock::now(); const double pc2 = trace_cpu_now();
        // column numbering of K10: group after group, empty groups have no columns
        std::vector<u64> col_off(nc + 1, 0);
        for (size_t g = 0; g < nc; g++) col_off[g + 1] = col_off[g] + (grp_off[g + 1] > grp_off[g] ? consensuses[g].sequence.size() : 0);
        const u64 ncol = col_off[nc];
        // ---- estimate_quality_error_rates: top 10 % of the clusters by depth (:669-680)
        std::vector<std::pair<size_t, size_t>> by_depth;
        for (size_t i = 0; i < nc; i++) by_depth.push_back({i, consensuses[i].depth});
        std::stable_sort(by_depth.begin(), by_depth.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
        const size_t take = std::min(nc, (size_t)std::llround(0.1 * (double)nc));
        std::vector<u8> selected(nc, 0);
        for (size_t t = 0; t < take; t++) selected[by_depth[t].first] = 1;
        depth.assign(std::max<u64>(ncol, 1), 0); err.assign(std::max<u64>(ncol, 1), 0);
        u64 qt[256], qe[256];
        chk4(rs.ctx, svt_pileup_stats(rs.ctx, pile, selected.data(), depth.data(), err.data(), qt, qe), "svt_pileup_stats");
        if (by_cluster) {                                                        // integer sums over the ranks: the same totals, whatever the split
            std::vector<u64> part(512), all((size_t)512 * sh_world), bytes(sh_world, 4096);
            for (int q = 0; q < 256; q++) { part[q] = qt[q]; part[256 + q] = qe[q]; }
            chk4(rs.ctx, svt_shard_allgatherv(rs.ctx, part.data(), bytes.data(), all.data()), "svt_shard_allgatherv(quality histograms)");
            for (int q = 0; q < 256; q++) { qt[q] = 0; qe[q] = 0; for (u32 r = 0; r < sh_world; r++) { qt[q] += all[(size_t)512 * r + q]; qe[q] += all[(size_t)512 * r + 256 + q]; } }
        }
        for (int q = 0; q < 256; q++) if (qt[q]) qmap[(u8)q] = (double)(1 + qe[q]) / (double)(1 + qt[q]);      // prior (1,1) :687,:728; rate :782-785
        // ---- ln tables (the device only adds)
        const double DEFAULT_ERR_RATE = 0.02;                                    // src/constants.rs:35
        auto rate = [&](u8 q) { auto it = qmap.find(q); return it == qmap.end() ? DEFAULT_ERR_RATE : it->second; };
        const double indel_err = rate(48);                                       // :874-879
        std::vector<double> tab(512);
        for (int q = 0; q < 256; q++) { const double er = rate((u8)q), acc = 1.0 - er; tab[2 * q] = std::log(acc); tab[2 * q + 1] = std::log(er); }
        lr.assign(std::max<u64>(ncol, 1), 0.0); ln.assign(std::max<u64>(ncol, 1), 0.0);
        chk4(rs.ctx, svt_pileup_loglik(rs.ctx, pile, tab.data(), std::log(indel_err), std::log(1.0 - indel_err), lr.data(), ln.data()), "svt_pileup_loglik");
        if (args.use_hpc) {                                                      // :586-656 consensus hp_lengths = median run length per column
            std::vector<u8> med(std::max<u64>(ncol, 1), 1);
            chk4(rs.ctx, svt_pileup_hp_median(rs.ctx, pile, med.data()), "svt_pileup_hp_median");
            for (size_t ci = 0; ci < nc; ci++) {
                if (!mine[ci]) continue;
                const size_t len = consensuses[ci].sequence.size();
                if (grp_off[ci + 1] > grp_off[ci]) consensuses[ci].hp_lengths.assign(med.begin() + col_off[ci], med.begin() + col_off[ci] + len);
                else consensuses[ci].hp_lengths.assign(len, 1);                  // nothing aligned: the placeholder stays (:623-625)
            }
        }
        if (keep) {                                                              // test hook: the same rows as Vec<Pileup> entries
            std::vector<u64> cells(std::max<u64>(svt_pileup_cells(pile), 1)), cell_off(n2 + 1);
            chk4(rs.ctx, svt_pileup_fetch(rs.ctx, pile, cells.data(), cell_off.data()), "svt_pileup_fetch");
            static const u8 ACGT[4] = {'A', 'C', 'G', 'T'};
            for (size_t ci = 0; ci < nc; ci++) (*keep)[ci].resize(consensuses[ci].sequence.size());
            for (size_t i = 0; i < n2; i++) {
                std::vector<PileupColumn>& cols = (*keep)[q2[i]];
                const u64* row = &cells[cell_off[i]];
                for (size_t p = 0; p < cols.size(); p++) {
                    const u64 c = row[p]; const u32 code = (u32)(c & 7);
                    if (code < 4) cols[p].entries.push_back(PileupEntry{0, ACGT[code], (u8)((c >> 8) & 0xFF), (u8)(c >> 56)});   // add_base :538
                    else if (code == 4) cols[p].entries.push_back(PileupEntry{1, 0, 0, 0});                                  // add_deletion :562
                    if ((c >> 16) & 3) cols[p].entries.push_back(PileupEntry{2, ACGT[(c >> 32) & 3], (u8)((c >> 40) & 0xFF), 0});   // add_insertion :554
                }
            }
        }
        const auto pt3 = std::chrono::steady_clock::now(); const double pc3 = trace_cpu_now();
        if (trace_enabled()) {
            auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
            trace_add("4b.batch+k7", sec(pt0, pt1), pc1 - pc0); trace_add("4b.k9_pileup", sec(pt1, pt2), pc2 - pc1); trace_add("4b.k10_stats", sec(pt2, pt3), pc3 - pc2);
        }
        svt_pileup_free(rs.ctx, pile); pile = nullptr;
        svt_batch_free(rs.ctx, cb); cb = nullptr;
        if (hb) { svt_batch_free(rs.ctx, hb); hb = nullptr; }
        // ---- analyze_pileup_consensuses on the column statistics
        const size_t bad_length_threshold = 100;                                 // :872
        const size_t min_coverage_abs = std::max<size_t>(args.min_cluster_size * 3 / 4, 2);   // :873
        const double post_threshold = std::min(args.posterior_threshold_ln, (double)(args.min_cluster_size * 3));   // :995
        for (size_t ci = 0; ci < nc; ci++) {
            if (!mine[ci]) continue;
            ConsensusSequence& cons = consensuses[ci];
            const size_t len = cons.sequence.size();
            if (len == 0) continue;                                              // :895 empty pile-up
            const bool has = grp_off[ci + 1] > grp_off[ci];                   // nobody mapped: every column is empty (depth 0, lr = ln = 0)
            std::vector<u32> zero_d; if (!has) zero_d.assign(len, 0);
            const u32* d = has ? &depth[col_off[ci]] : zero_d.data();
            size_t maxd = 0; for (size_t p = 0; p < len; p++) maxd = std::max<size_t>(maxd, d[p]);
            const size_t min_cov = std::max(maxd / 3, min_coverage_abs);         // :894
            size_t start = 0, end = len;
            for (size_t i = 0; i < len; i++) if (d[i] >= min_cov) { start = i; break; }                                   // :905-914
            for (size_t i = len; i-- > 0;) if (d[i] >= min_cov) { end = i + 1; break; }                                    // :917-926
            std::vector<size_t> low_conf;
            size_t left_start = 0, right_end = len;                              // an untrimmed pile-up spans everything (:928-931)
            if (start < end) {
                left_start = start; right_end = end;                             // :1092-1093
                for (size_t p = start; p < end; p++) {
                    const double a = has ? ln[col_off[ci] + p] : 0.0, b = has ? lr[col_off[ci] + p] : 0.0;
                    // alt_post = a - (mx + ln(e^(a-mx) + e^(b-mx))) <= a - b: a column whose reference allele leads by more than the threshold (+ 1: far beyond any
                    // rounding of the three libm calls) cannot pass the test below -- nearly every column, and two exp + one log each otherwise
                    if (b - a > post_threshold + 1.0) continue;
                    const double alt_post = a - log_sum_exp(b, a);               // :991-992
                    if (alt_post > -post_threshold) low_conf.push_back(p);       // :996,:1025
                }
            }
            const size_t start_polish = bad_length_threshold + left_start;
            const size_t end_polish = right_end >= bad_length_threshold ? right_end - bad_length_threshold : 0;
            size_t lc_left = left_start; bool have_l = false;
            for (size_t p : low_conf) if (p < start_polish) { lc_left = have_l ? std::max(lc_left, p) : p; have_l = true; }    // :1098-1099
            size_t lc_right = right_end; bool have_r = false;
            for (size_t p : low_conf) if (p >= end_polish) { lc_right = have_r ? std::min(lc_right, p) : p; have_r = true; }   // :1100-1101
            for (size_t p = 0; p < lc_left && p < len; p++) cons.sequence[p] = 'N';                                            // :1104-1109
            for (size_t p = lc_right; p < len; p++) cons.sequence[p] = 'N';                                                    // :1110-1115
            for (size_t p : low_conf) {
                if (args.mask_low_quality) cons.sequence[p] = 'N';               // :1119-1121
                if (p > lc_left && p < lc_right) cons.low_quality_positions.push_back(p);   // :1122-1125
            }
        }
        if (by_cluster) {
            // (cluster, masked sequence, low-quality positions, run lengths) of the clusters every rank polished, applied to the others' copies
            std::vector<u8> buf;
            auto put32 = [&](u32 v) { const u8* p = (const u8*)&v; buf.insert(buf.end(), p, p + 4); };
            for (size_t ci = 0; ci < nc; ci++) {
                if (!mine[ci]) continue;
                const ConsensusSequence& c = consensuses[ci];
                put32((u32)ci); put32((u32)c.sequence.size()); buf.insert(buf.end(), c.sequence.begin(), c.sequence.end());
                put32((u32)c.low_quality_positions.size()); for (size_t p : c.low_quality_positions) put32((u32)p);
                put32((u32)c.hp_lengths.size()); buf.insert(buf.end(), c.hp_lengths.begin(), c.hp_lengths.end());
            }
            std::vector<u64> bytes(sh_world, 0);
            chk4(rs.ctx, svt_shard_allgather_u64(rs.ctx, buf.size(), bytes.data()), "svt_shard_allgather_u64(polish)");
            u64 total = 0; for (u64 b : bytes) total += b;
            std::vector<u8> all(total + 1);
            chk4(rs.ctx, svt_shard_allgatherv(rs.ctx, buf.data(), bytes.data(), all.data()), "svt_shard_allgatherv(polish)");
            auto get32 = [&](size_t& i) { u32 v; memcpy(&v, &all[i], 4); i += 4; return v; };
            for (size_t i = 0; i < total;) {
                const u32 ci = get32(i), sl = get32(i);
                ConsensusSequence& c = consensuses[ci];
                c.sequence.assign(all.begin() + i, all.begin() + i + sl); i += sl;
                const u32 nl = get32(i); c.low_quality_positions.clear();
                for (u32 x = 0; x < nl; x++) c.low_quality_positions.push_back(get32(i));
                const u32 nh = get32(i); c.hp_lengths.assign(all.begin() + i, all.begin() + i + nh); i += nh;
            }
        }
    } catch (...) { if (pile) svt_pileup_free(rs.ctx, pile); if (cb) svt_batch_free(rs.ctx, cb); if (hb) svt_batch_free(rs.ctx, hb); throw; }
    if (qmap_out) *qmap_out = qmap;
    auto lq = [&](const ConsensusSequence& c) {                                 // lq_criteria :1157-1160
        const size_t n = c.low_quality_positions.size();
        return n > 0 && c.depth / (n * n) < args.n_depth_cutoff;
    };
    std::vector<ConsensusSequence> low, keepv;
    for (auto& c : consensuses) { if (lq(c)) low.push_back(c); else keepv.push_back(c); }
    consensuses.swap(keepv);
    return low;
}

// ConsensusSequence::decompress (src/types.rs:212-217, utils::homopolymer_decompress src/utils.rs:114-130): every base repeated by its
// run length (all 1 without --use-hpc), then leading / trailing N trimmed
void decompress(ConsensusSequence& c) {
    std::vector<u8> full;
    const std::vector<u8>* src = &c.sequence;
    if (!c.hp_lengths.empty() && c.hp_lengths.size() == c.sequence.size()) {     // a length mismatch returns the sequence as it is (:115-118)
        for (size_t i = 0; i < c.sequence.size(); i++) full.insert(full.end(), c.hp_lengths[i], c.sequence[i]);
        src = &full;
    }
    size_t a = 0, b = src->size();
    while (a < b && (*src)[a] == 'N') a++;
    while (b > a && (*src)[b - 1] == 'N') b--;
    if (a >= b) { a = 0; b = src->size(); }
    c.decompressed.assign(src->begin() + a, src->begin() + b);
}

}  // namespace savont
