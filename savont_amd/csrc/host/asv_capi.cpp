// asv_capi.cpp -- extern "C" surface of libsavont_asv.so (host pipeline) for the Python harness and
// for a Rust `run_cluster` that wants whole stages rather than kernels.  Also hosts the deterministic
// synthetic amplicon generator used by bench.py / tests.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <string>

#include <unistd.h>
#include "asv_pipeline.hpp"
#include "inflate.hpp"
#include "savont_asv.h"                 // the declarations of everything below: the compiler holds the two together
#include "sampler.hpp"
// memcpy whose source may be the data() of an EMPTY vector (a null pointer with a zero count is not a valid memcpy call)
static inline void cpy(void* d, const void* s, size_t n) { if (n) memcpy(d, s, n); }
#include "worker_pool.hpp"
#include "stats.hpp"

using namespace savont;
typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;


struct svh_pipeline {
    svt_ctx* ctx = nullptr;
    ClusterArgs args;
    ReadSet rs;
    svt_batch* asvs = nullptr; std::vector<u64> asv_off;
    KmerCountTable table; u64 n_distinct = 0, n_kept = 0;
    int table_where = 0;                       // 0 none, 1 in HBM (counted by this pipeline), 2 on the host (fetched or set by the caller)
    KmerGlobalInfo info;
    TwinReads tw;
    std::vector<std::vector<u32>> kmer_clusters, snp_clusters, snp_pre; std::vector<u32> snp_pre_group;
    EmResult em;
    std::vector<ConsensusSequence> consensuses, low_qual; std::map<u8, double> qmap;
    std::vector<u32> chimera_ids; u32 n_after_merge = 0;
    bool keep_pileups = false; Pileups pileups; std::vector<ConsensusSequence> raw_consensuses;   // test hook (svh_keep_pileups)
    // svh_load_fastx: the parse buffers persist between loads (warm pages) and are page-locked for the upload while their storage does not move
    RawBytes ing_seq, ing_qual; void* pinned[2] = {nullptr, nullptr}; size_t pinned_cap[2] = {0, 0};
    std::vector<std::vector<u8>> poa_raw; int poa_which = 1;                                       // pooled multi-rank run: raw consensus per cluster (mine, then everyone's)
    std::string err;
    std::map<std::string, double> seconds;
    std::string temp_dir;                       // non-empty: every stage writes the reference's intermediate file(s) there (svh_set_temp_dir)
    bool engines_logged = false;                // the engines chosen by the CPU share of the process are reported once (stderr, like the reference's log::info lines)
};

// Two implementation choices follow the CPU share of the process unless the caller pins them (poa_engine, stage2_device = -1): who runs the Stage-4a POA and who
// builds the Stage-2 candidate lists.  Results are identical either way (tests run the alternatives against the same oracle), speed and host load are not: the
// pipeline says once what it chose, so that two runs of the same command on hosts of different size can be told apart.
static void log_engines_once(svh_pipeline* p) {
    if (p->engines_logged) return;
    p->engines_logged = true;
    const Tuning& t = p->args.tuning;
    const unsigned th = (unsigned)WorkerPool::get().threads();
    const char* poa = t.poa_engine < 0 ? (th <= 10 ? "K12 on the device (auto: <= 10 worker threads)" : "host DP (auto: > 10 worker threads)")
                    : t.poa_engine == 0 ? "host DP (pinned)" : t.poa_engine == 2 ? "K12 on the device (pinned)" : "split: K12 for poa_device_share percent of the clusters, host DP for the rest (pinned)";
    const char* s2 = t.stage2_device < 0 ? (th <= 10 ? "device (auto)" : "host bucket walk (auto)") : t.stage2_device ? "device (pinned)" : "host bucket walk (pinned)";
    fprintf(stderr, "[savont] engines: Stage-4a POA = %s; Stage-2 candidate lists = %s; %u worker threads.  Same results either way; svh_set_option poa_engine / stage2_device pins them.\n", poa, s2, th);
}

namespace {
struct StageTimer {
    svh_pipeline* p; std::string name; std::chrono::steady_clock::time_point t0;
    StageTimer(svh_pipeline* p_, const char* n) : p(p_), name(n), t0(std::chrono::steady_clock::now()) {}
    ~StageTimer() { p->seconds[name] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
template <class F> int guarded(svh_pipeline* p, F f) {
    savont::sampler::arm_thread();                                              // development sampler only (SAVONT_SAMPLE): this caller thread's CPU timer
    try { f(); return 0; }
    catch (const Error& e) { p->err = e.msg; return e.code ? e.code : -1; }
    catch (const std::exception& e) { p->err = e.what(); return -100; }
}
void fetch_clusters(const std::vector<std::vector<u32>>& cl, u64* off, u32* mem) {
    u64 o = 0;
    for (size_t i = 0; i < cl.size(); i++) { off[i] = o; for (u32 x : cl[i]) mem[o++] = x; }
    off[cl.size()] = o;
}
u64 total_members(const std::vector<std::vector<u32>>& cl) { u64 t = 0; for (auto& x : cl) t += x.size(); return t; }
}  // namespace

extern "C" {

void svh_default_args(svh_args* a) {
    ClusterArgs d;
    a->kmer_size = d.kmer_size; a->c = d.c; a->min_read_length = d.min_read_length; a->max_read_length = d.max_read_length;
    a->quality_value_cutoff = d.quality_value_cutoff; a->minimum_base_quality = d.minimum_base_quality; a->single_strand = d.single_strand;
    a->min_cluster_size = d.min_cluster_size; a->max_iterations_recluster = d.max_iterations_recluster;
    a->primary_clustering_threshold = d.primary_clustering_threshold; a->low_polymorphism = d.low_polymorphism; a->align_band = d.align_band;
    a->n_depth_cutoff = d.n_depth_cutoff; a->mask_low_quality = d.mask_low_quality; a->posterior_threshold_ln = d.posterior_threshold_ln;
    a->chimera_allowable_errors = d.chimera_allowable_errors; a->chimera_detect_length = d.chimera_detect_length; a->skip_chimera_detection = d.skip_chimera_detection; a->use_hpc = d.use_hpc ? 1 : 0;
    a->no_snpmers = d.no_snpmers ? 1 : 0; a->no_band = d.no_band ? 1 : 0;
}

// The library is compiled for x86-64-v3 (csrc/Makefile: HOST_MARCH): on a host without AVX2 / BMI2 / POPCNT it says so when it is loaded instead of dying on an
// illegal instruction somewhere inside a stage.
__attribute__((constructor)) static void svh_check_host_cpu() {
#if defined(__x86_64__) && defined(__AVX2__)
    __builtin_cpu_init();
    if (!__builtin_cpu_supports("avx2") || !__builtin_cpu_supports("bmi2") || !__builtin_cpu_supports("popcnt") || !__builtin_cpu_supports("fma")) {
        fprintf(stderr, "[savont] libsavont_asv.so was built for x86-64-v3 (AVX2, BMI2, POPCNT, FMA) and this CPU lacks one of them: rebuild with `make HOST_MARCH=x86-64`\n");
        abort();
    }
#endif
}

int svh_create(int device_id, const svh_args* a, svh_pipeline** out) {
    savont::sampler::start_once(); if (savont::sampler::g_path) { savont::WorkerPool::thread_hook().store(+[] { savont::sampler::arm_thread(); }); }
    *out = nullptr;
    svt_ctx* ctx = nullptr;
    int rc = svt_create(device_id, &ctx);
    if (rc != SVT_OK) return rc;                     // no GPU -> loud failure, no CPU path
    svh_pipeline* p = new svh_pipeline();
    p->ctx = ctx;
    if (a) {
        ClusterArgs& d = p->args;
        d.kmer_size = a->kmer_size; d.c = a->c; d.min_read_length = a->min_read_length; d.max_read_length = a->max_read_length;
        d.quality_value_cutoff = a->quality_value_cutoff; d.minimum_base_quality = (u8)a->minimum_base_quality; d.single_strand = a->single_strand != 0;
        d.min_cluster_size = a->min_cluster_size; d.max_iterations_recluster = a->max_iterations_recluster;
        d.primary_clustering_threshold = a->primary_clustering_threshold; d.low_polymorphism = a->low_polymorphism != 0; d.align_band = a->align_band;
        d.n_depth_cutoff = a->n_depth_cutoff; d.mask_low_quality = a->mask_low_quality != 0; d.posterior_threshold_ln = a->posterior_threshold_ln;
        d.chimera_allowable_errors = a->chimera_allowable_errors; d.chimera_detect_length = a->chimera_detect_length; d.skip_chimera_detection = a->skip_chimera_detection != 0; d.use_hpc = a->use_hpc != 0;
        d.no_snpmers = a->no_snpmers != 0; d.no_band = a->no_band != 0;
    }
    p->rs.ctx = ctx;
    *out = p;
    return 0;
}
// Implementation choices (identical results) are pipeline state set through this call, never process environment.  Host keys:
// stage2_first_block, stage2_max_block, stage2_pair_cap, stage2_device (-1 by CPU share | 0 host | 1 device: candidate lists of a block), stage3_first_block, stage3_block, stage3_max_block, stage3_switch,
// poa_engine (-1 by CPU share | 0 host DP | 2 K12, graphs resident on the device | 3 K12 for a share), poa_cells (16 | 32), nm_contract (0 K8 | 1 K8a near the unit-cost optimum | 2 K8a whole band, Stage 7); every other key goes to svt_set_option of the device layer.
int svh_set_option(svh_pipeline* p, const char* key, int64_t value) {
    if (!p || !key) return -1;
    Tuning& t = p->args.tuning; const std::string k = key;
    auto pos = [&](uint32_t& dst) { if (value < 1 || value > (1 << 24)) { p->err = "svh_set_option: value out of range for '" + k + "'"; return SVT_ERR_ARG; } dst = (uint32_t)value; return 0; };
    if (k == "stage2_first_block") return pos(t.stage2_first_block);
    if (k == "stage2_max_block") return pos(t.stage2_max_block);
    if (k == "stage3_first_block") return pos(t.stage3_first_block);
    if (k == "stage3_block") return pos(t.stage3_block);
    if (k == "stage3_max_block") return pos(t.stage3_max_block);
    if (k == "stage3_switch") return pos(t.stage3_switch);
    if (k == "stage2_pair_cap") { if (value < 1) { p->err = "svh_set_option: stage2_pair_cap must be positive"; return SVT_ERR_ARG; } t.stage2_pair_cap = (uint64_t)value; return 0; }
    if (k == "stage3_waves") { t.stage3_waves = value != 0; return 0; }
    if (k == "stage2_device") { if (value < -1 || value > 1) { p->err = "svh_set_option: stage2_device is -1 (by CPU share), 0 (host bucket walk) or 1 (device)"; return SVT_ERR_ARG; } t.stage2_device = (int)value; return 0; }
    if (k == "poa_engine") { if (value < -1 || value > 3 || value == 1) { p->err = "svh_set_option: poa_engine is -1 (by CPU share), 0 (host), 2 (K12, device-resident graphs) or 3 (K12 for poa_device_share percent of the clusters, host DP for the others)"; return SVT_ERR_ARG; } t.poa_engine = (int)value; return 0; }
    if (k == "nm_contract") { if (value < 0 || value > 2) { p->err = "svh_set_option: nm_contract is 0 (K8), 1 (K8a near the unit-cost optimum) or 2 (K8a, whole band)"; return SVT_ERR_ARG; } t.nm_contract = (int)value; return 0; }
    if (k == "poa_device_share") { if (value < 0 || value > 100) { p->err = "svh_set_option: poa_device_share is a percentage"; return SVT_ERR_ARG; } t.poa_device_share = (int)value; return 0; }
    if (k == "gz_threads") { if (value < 0 || value > 64) return -1; set_gz_threads((int)value); return 0; }     // threads one gzip member is inflated on; 0 (default) = up to eight when no other file is being inflated, else one (process-wide)
    if (k == "gz_inflate") { set_gz_inflate(value != 0); return 0; }     // 1 (default): gz inputs through host/inflate.hpp; 0: zlib's gzread (comparison runs; process-wide)
    if (k == "poa_cells") { if (value != 16 && value != 32) { p->err = "svh_set_option: poa_cells is 16 or 32"; return SVT_ERR_ARG; } t.poa_cells = (int)value; return 0; }
    const int rc = svt_set_option(p->ctx, key, value);
    if (rc != SVT_OK) p->err = svt_last_error(p->ctx);
    return rc;
}
// `<out>/temp/` of the reference (src/main.rs:55-58): when set, the stages write kmer_clusters_stage2.tsv, snpmer_clusters_before_reclust2.5.tsv,
// final_snpmer_clusters_stage3.tsv, consensus_sequences.fasta, low_quality_clusters.tsv, clusters_after_quality_filter_stage4.tsv,
// low_quality_consensus_sequences.fasta, final_clusters_merged_stage5.tsv, merged_consensus_sequences.fasta and final_asvs_for_em.fasta and read_to_asv_mappings.tsv in the
// reference's formats.  NULL / "" switches it off.
int svh_set_temp_dir(svh_pipeline* p, const char* dir) { p->temp_dir = dir ? dir : ""; return 0; }
void svh_trace_dump(void) { trace_dump(); }                                // SAVONT_TRACE=1: print and clear the host timers (e.g. after warm-up)
void svh_destroy(svh_pipeline* p) {
    if (!p) return;
    trace_dump();
    if (p->rs.batch) svt_batch_free(p->ctx, p->rs.batch);
    if (p->asvs) svt_batch_free(p->ctx, p->asvs);
    for (int k = 0; k < 2; k++) if (p->pinned[k]) svt_host_unpin(p->ctx, p->pinned[k]);
    svt_destroy(p->ctx);
    delete p;
}
const char* svh_last_error(svh_pipeline* p) { return p->err.c_str(); }
svt_ctx* svh_ctx(svh_pipeline* p) { return p->ctx; }
double svh_stage_seconds(svh_pipeline* p, const char* name) { auto it = p->seconds.find(name); return it == p->seconds.end() ? -1.0 : it->second; }

// reads: uploads to HBM (this is the PCIe step; everything after it works on resident data)
static void set_reads_impl(svh_pipeline* p, const u8* seq, const u8* qual, const u64* offsets, u32 n, std::vector<std::string> ids, const u32* file_idx, bool keep_host_copy = true) {
    StageTimer t(p, "upload");
    ReadSet& rs = p->rs;
    if (rs.batch) { svt_batch_free(p->ctx, rs.batch); rs.batch = nullptr; }
    rs.n = n; rs.offsets.assign(offsets, offsets + n + 1);
    rs.rc_flags.assign(n, 0);
    for (u32 i = 0; i < n; i++) {
        const std::string& id = ids[i];
        size_t e = id.find_last_not_of(" \t\r\n\f\v");                       // src/seq_parse.rs:362-366
        if (e != std::string::npos) {
            size_t b = id.find_last_of(" \t\r\n\f\v", e);
            size_t s = (b == std::string::npos) ? 0 : b + 1;
            rs.rc_flags[i] = (id.compare(s, e - s + 1, "rc") == 0) ? 1 : 0;
        }
    }
    rs.ids = std::move(ids);
    if (file_idx) rs.file_idx.assign(file_idx, file_idx + n); else rs.file_idx.clear();
    if (keep_host_copy) rs.host_seq.assign(seq, seq + offsets[n]);
    rs.qualbin_off.clear(); rs.qualbins.clear();
    int rc = svt_batch_upload(p->ctx, seq, qual, offsets, n, &rs.batch);
    if (rc != SVT_OK) throw Error{rc, std::string("svt_batch_upload: ") + svt_last_error(p->ctx)};
}
int svh_set_reads(svh_pipeline* p, const u8* seq, const u8* qual, const u64* offsets, u32 n, const char* ids_joined, const u32* file_idx) {
    return guarded(p, [&] {
        std::vector<std::string> ids; ids.reserve(n);
        const char* q = ids_joined;
        for (u32 i = 0; i < n; i++) {
            std::string id;
            if (q) { const char* e = strchr(q, '\n'); id = e ? std::string(q, e) : std::string(q); q = e ? e + 1 : q + strlen(q); }
            else { char buf[32]; snprintf(buf, sizeof buf, "read_%08u", i); id = buf; }
            ids.push_back(std::move(id));
        }
        set_reads_impl(p, seq, qual, offsets, n, std::move(ids), file_idx);
    });
}
// FASTA/FASTQ (gz or plain) files, '\n'-joined paths, one sample per file (file_idx = position in the list) -> reads in HBM
int svh_load_fastx(svh_pipeline* p, const char* paths_joined, u32* n_reads) {
    return guarded(p, [&] {
        RawBytes& seq = p->ing_seq; RawBytes& qual = p->ing_qual;
        // The two ingest vectors stay page-locked between loads (hipHostRegister on data() / capacity()).  A registered block must never be
        // freed, and a parse that outgrows the capacity reallocates: so the registration is kept only when the parse provably fits -- every file is
        // plain text (bases + qualities of a file are each smaller than the file) and the sizes sum to at most the smaller capacity; otherwise both
        // are unpinned BEFORE the vectors are touched and pinned again once their storage is final.
        { u64 need = 0; bool known = true;
          for (const char* q = paths_joined; q && *q;) {
              const char* e = strchr(q, '\n'); std::string path = e ? std::string(q, e) : std::string(q); q = e ? e + 1 : nullptr;
              if (path.empty()) continue;
              FILE* f = fopen(path.c_str(), "rb"); unsigned char mg[2] = {0, 0};
              if (!f || fread(mg, 1, 2, f) != 2 || (mg[0] == 0x1f && mg[1] == 0x8b) || (mg[0] == 'B' && mg[1] == 'Z')) known = false;
              else { fseek(f, 0, SEEK_END); need += (u64)ftell(f); }
              if (f) fclose(f);
          }
          const bool fits = known && need + 1 <= std::min(seq.capacity(), qual.capacity());
          if (!fits) for (int k = 0; k < 2; k++) if (p->pinned[k]) { svt_host_unpin(p->ctx, p->pinned[k]); p->pinned[k] = nullptr; p->pinned_cap[k] = 0; } }
        seq.clear(); qual.clear();
        std::vector<u64> off(1, 0); std::vector<std::string> ids; std::vector<u32> file_idx; bool any_qual = false;
        { StageTimer t(p, "ingest");
        std::vector<std::string> files;                                       // position in the list = sample index; empty names keep their position
        for (const char* q = paths_joined; q && *q;) { const char* e = strchr(q, '\n'); files.push_back(e ? std::string(q, e) : std::string(q)); q = e ? e + 1 : nullptr; }
        read_fastx_files(files, seq, qual, off, ids, file_idx, any_qual); }
        if (seq.empty()) seq.push_back('A');
        // page-lock the two buffers while their storage stays where it is (a later, larger file moves it: lock again)
        RawBytes* bufs[2] = {&seq, &qual};
        for (int k = 0; k < 2; k++) {
            RawBytes& v = *bufs[k];
            if (v.capacity() < ((size_t)8 << 20)) continue;
            if (p->pinned[k] == (void*)v.data() && p->pinned_cap[k] == v.capacity()) continue;
            if (p->pinned[k]) throw Error{SVT_ERR_STATE, "svh_load_fastx: an ingest buffer moved while it was page-locked"};   // cannot happen: see the top of this function
            if (svt_host_pin(p->ctx, v.data(), v.capacity()) == SVT_OK) { p->pinned[k] = v.data(); p->pinned_cap[k] = v.capacity(); }
        }
        const u32 n = (u32)ids.size();
        set_reads_impl(p, seq.data(), any_qual ? qual.data() : nullptr, off.data(), n, std::move(ids), file_idx.data(), false);
        p->rs.host_seq.resize(seq.size());                         // Stage 4a reads the bases on the host; the ingest buffer is parsed into again by the next load
        { const size_t nb = seq.size(), CH = (size_t)8 << 20, nch = (nb + CH - 1) / CH;      // copied on the pool: 150 MB per 100k reads (and on a first load the page faults of the copy)
          par_for(nch, [&](size_t k) { const size_t lo = k * CH, hi = std::min(nb, lo + CH); memcpy(p->rs.host_seq.data() + lo, seq.data() + lo, hi - lo); }); }
        if (n_reads) *n_reads = p->rs.n;
    });
}
// final_asvs.fasta, feature-table.tsv, final_clusters.tsv in out_dir (src/main.rs:153-199).  sample_names: '\n'-joined;
// pooled != 0 (and more than one sample) writes per-sample depths (--pooled-samples), else the first name labels the one column.
int svh_write_outputs(svh_pipeline* p, const char* out_dir, const char* sample_names_joined, int pooled) {
    return guarded(p, [&] {
        StageTimer t(p, "write");
        std::vector<std::string> names;
        for (const char* q = sample_names_joined; q && *q;) { const char* e = strchr(q, '\n'); names.push_back(e ? std::string(q, e) : std::string(q)); q = e ? e + 1 : nullptr; }
        if (names.empty()) names.push_back("sample");
        std::vector<std::vector<u64>> per_sample;
        const bool do_pool = pooled && names.size() > 1;
        if (do_pool) per_sample = compute_per_sample_depths(p->tw, p->em, (u32)names.size(), p->em.depth.size());
        std::vector<FinalAsv> fin = finalize_asvs(p->consensuses, p->em, do_pool ? &per_sample : nullptr);
        const std::string dir = out_dir;
        write_consensus_fasta(fin, dir + "/final_asvs.fasta", "final");
        if (!do_pool) names.resize(1);
        write_feature_table(fin, dir + "/feature-table.tsv", names);
        write_clusters_tsv(fin, p->rs, p->tw, dir + "/final_clusters.tsv", "final");
    });
}

// K0 again from the ASCII bases in HBM (option keep_ascii set before the reads were uploaded): lets a benchmark step start from unpacked reads
int svh_repack(svh_pipeline* p) {
    return guarded(p, [&] { StageTimer t(p, "pack"); int rc = svt_batch_repack(p->ctx, p->rs.batch); if (rc != SVT_OK) throw Error{rc, std::string("svt_batch_repack: ") + svt_last_error(p->ctx)}; });
}
int svh_read_to_split_kmers(svh_pipeline* p) {
    return guarded(p, [&] { StageTimer t(p, "count"); p->table.clear(); p->table_where = 0; count_split_kmers_device(p->rs, p->args, &p->n_distinct, &p->n_kept); p->table_where = 1; });
}
u64 svh_count_distinct(svh_pipeline* p) { return p->n_distinct; }
u64 svh_count_size(svh_pipeline* p) { return p->table_where == 2 ? p->table.size() : p->n_kept; }
int svh_count_fetch(svh_pipeline* p, u64* km, u32* rev, u32* fwd) {            // the whole sorted table (B1), copied out of HBM on demand
    return guarded(p, [&] {
        if (p->table_where == 1) { p->table = fetch_count_table(p->rs, p->n_kept); p->table_where = 2; }
        for (size_t i = 0; i < p->table.size(); i++) { km[i] = p->table[i].first; rev[i] = p->table[i].second.first; fwd[i] = p->table[i].second.second; }
    });
}
int svh_set_count_table(svh_pipeline* p, const u64* km, const u32* rev, const u32* fwd, u64 n) {   // multi-GPU: table merged elsewhere
    p->table.resize(n);
    for (u64 i = 0; i < n; i++) p->table[i] = {km[i], {rev[i], fwd[i]}};
    p->table_where = 2;
    return 0;
}

int svh_get_snpmers(svh_pipeline* p) {
    return guarded(p, [&] {
        StageTimer t(p, "snpmers");
        if (p->table_where == 1) p->info = snpmers_from_candidates(candidates_from_device(p->rs), p->args.kmer_size, p->args);
        else p->info = get_snpmers_inplace_sort(p->table, p->args.kmer_size, p->args);
    });
}
u32 svh_snpmer_count(svh_pipeline* p) { return (u32)p->info.snpmer_info.size(); }
void svh_snpmer_fetch(svh_pipeline* p, u64* split, u8* m0, u8* m1, u32* c0, u32* c1) {
    for (size_t i = 0; i < p->info.snpmer_info.size(); i++) {
        const SnpmerInfo& s = p->info.snpmer_info[i];
        split[i] = s.split_kmer; m0[i] = s.mid_bases[0]; m1[i] = s.mid_bases[1]; if (c0) c0[i] = s.counts[0]; if (c1) c1[i] = s.counts[1];
    }
}
u32 svh_high_freq_thresh(svh_pipeline* p) { return (u32)p->info.high_freq_thresh; }
u32 svh_high_freq_count(svh_pipeline* p) { return (u32)p->info.high_freq_kmers.size(); }
void svh_high_freq_fetch(svh_pipeline* p, u64* k) { cpy(k, p->info.high_freq_kmers.data(), p->info.high_freq_kmers.size() * 8); }
int svh_set_snpmers(svh_pipeline* p, const u64* split, const u8* m0, const u8* m1, u32 n, const u64* hf, u32 n_hf) {
    p->info.snpmer_info.clear();
    for (u32 i = 0; i < n; i++) { SnpmerInfo s; s.split_kmer = split[i]; s.mid_bases[0] = m0[i]; s.mid_bases[1] = m1[i]; s.counts[0] = s.counts[1] = 0; s.k = (u8)p->args.kmer_size; p->info.snpmer_info.push_back(s); }
    p->info.high_freq_kmers.assign(hf, hf + n_hf);
    return 0;
}

int svh_twin_reads(svh_pipeline* p) {
    return guarded(p, [&] { StageTimer t(p, "twin_reads"); twin_reads_from_snpmers(p->rs, p->info, p->args, p->tw); });
}
u32 svh_twin_count(svh_pipeline* p) { return p->tw.n; }
int svh_auto_low_polymorphism(svh_pipeline* p) { return p->tw.auto_low_polymorphism; }
void svh_twin_meta(svh_pipeline* p, u32* orig, u32* length, double* est, u8* ev, u32* n_mini, u32* n_unique, u32* n_snp_filt, u64* lsh, u8* lsh_valid) {
    const TwinReads& t = p->tw;
    if (orig) cpy(orig, t.orig.data(), t.n * 4); if (length) cpy(length, t.length.data(), t.n * 4);
    if (est) cpy(est, t.est_id.data(), t.n * 8); if (ev) cpy(ev, t.est_valid.data(), t.n);
    if (n_mini) cpy(n_mini, t.n_mini.data(), t.n * 4); if (n_unique) cpy(n_unique, t.n_unique.data(), t.n * 4);
    if (n_snp_filt) cpy(n_snp_filt, t.n_snp_filtered.data(), t.n * 4);
    if (lsh) cpy(lsh, t.lsh.data(), (size_t)t.n * SVT_LSH_TABLES * 8); if (lsh_valid) cpy(lsh_valid, t.lsh_valid.data(), t.n);
}

int svh_cluster_reads_by_kmers(svh_pipeline* p) {
    log_engines_once(p);
    return guarded(p, [&] {
        StageTimer t(p, "cluster_kmers"); p->kmer_clusters = cluster_reads_by_kmers(p->rs, p->tw, p->args);
        p->seconds["cluster_kmers.serial"] = p->rs.stage2.serial_seconds;       // the ordered fix-up + final grouping: what every rank of a pooled run repeats
        if (!p->temp_dir.empty()) write_kmer_clusters_tsv(p->kmer_clusters, p->temp_dir + "/kmer_clusters_stage2.tsv");
    });
}
int svh_cluster_reads_by_snpmers(svh_pipeline* p) {
    return guarded(p, [&] {
        StageTimer t(p, "cluster_snpmers");
        ClusterArgs a = p->args;
        if (p->tw.auto_low_polymorphism) a.low_polymorphism = true;              // src/main.rs:76-79
        p->snp_clusters = cluster_reads_by_snpmers(p->rs, p->tw, p->kmer_clusters, a, &p->snp_pre, &p->snp_pre_group);
        if (!p->temp_dir.empty()) {
            if (!a.low_polymorphism) write_pre_recluster_tsv(p->snp_pre, p->snp_pre_group, p->temp_dir + "/snpmer_clusters_before_reclust2.5.tsv");
            if (!a.low_polymorphism) write_snpmer_clusters_tsv(p->snp_clusters, p->rs, p->tw, p->temp_dir + "/final_snpmer_clusters_stage3.tsv");   // the low-polymorphism pass-through returns before both (:570-580)
        }
    });
}
u32 svh_cluster_count(svh_pipeline* p, int which) { return (u32)(which == 0 ? p->kmer_clusters : which == 1 ? p->snp_clusters : p->snp_pre).size(); }
u64 svh_cluster_total(svh_pipeline* p, int which) { return total_members(which == 0 ? p->kmer_clusters : which == 1 ? p->snp_clusters : p->snp_pre); }
void svh_clusters_fetch(svh_pipeline* p, int which, u64* off, u32* mem, u32* group) {
    fetch_clusters(which == 0 ? p->kmer_clusters : which == 1 ? p->snp_clusters : p->snp_pre, off, mem);
    if (which == 2 && group) cpy(group, p->snp_pre_group.data(), p->snp_pre_group.size() * 4);
}

// Stage 4b-d on the POA consensuses in p->consensuses (+ the temp files of src/alignment.rs:405-408, :1130-1141, src/main.rs:112)
static void stage4_after_poa(svh_pipeline* p) {
    if (p->keep_pileups) p->raw_consensuses = p->consensuses;
    if (!p->temp_dir.empty()) write_consensus_fasta(as_records(p->consensuses), p->temp_dir + "/consensus_sequences.fasta", "initial");
    p->low_qual = polish_consensuses(p->rs, p->tw, p->consensuses, p->args, &p->qmap, p->keep_pileups ? &p->pileups : nullptr);
    if (!p->temp_dir.empty()) {
        write_clusters_tsv(as_records(p->low_qual), p->rs, p->tw, p->temp_dir + "/low_quality_clusters.tsv", "low_quality", false);
        write_clusters_tsv(as_records(p->consensuses), p->rs, p->tw, p->temp_dir + "/clusters_after_quality_filter_stage4.tsv", "prefilter", false);
    }
    for (auto& c : p->consensuses) decompress(c);
    for (auto& c : p->low_qual) decompress(c);
    if (!p->temp_dir.empty()) write_consensus_fasta(as_records(p->low_qual), p->temp_dir + "/low_quality_consensus_sequences.fasta", "lowqual");
}

// ---- pooled multi-rank run: the sharded halves of stages 1a, 4a and 7 --------------------------------------
// Stage 1a on the read block [lo, hi) of the resident batch; the table stays in HBM (C1: svh_count_export_device -> all-gather ->
// svh_count_merge_begin / svh_count_merge_device on every rank -> svh_count_finalize)
int svh_count_partial_device(svh_pipeline* p, u32 lo, u32 hi, u64* n_distinct) {
    return guarded(p, [&] {
        StageTimer t(p, "count.partial");
        svt_batch* sl = nullptr;
        int rc = svt_batch_slice(p->ctx, p->rs.batch, lo, hi, &sl);
        if (rc != SVT_OK) throw Error{rc, std::string("svt_batch_slice: ") + svt_last_error(p->ctx)};
        rc = svt_count_partial_device(p->ctx, sl, p->args.kmer_size, p->args.minimum_base_quality, p->rs.rc_flags.empty() ? nullptr : p->rs.rc_flags.data() + lo, n_distinct);
        svt_batch_free(p->ctx, sl);
        if (rc != SVT_OK) throw Error{rc, std::string("svt_count_partial_device: ") + svt_last_error(p->ctx)};
        p->table.clear(); p->table_where = 0;
    });
}
int svh_count_export_device(svh_pipeline* p, u64* d_kmer, u32* d_rev, u32* d_fwd, u64 cap, u64* n) {
    return guarded(p, [&] { int rc = svt_count_export_device(p->ctx, d_kmer, d_rev, d_fwd, cap, n); if (rc != SVT_OK) throw Error{rc, std::string("svt_count_export_device: ") + svt_last_error(p->ctx)}; });
}
int svh_count_merge_begin(svh_pipeline* p, u64 total_entries) {
    return guarded(p, [&] { int rc = svt_count_merge_begin(p->ctx, total_entries); if (rc != SVT_OK) throw Error{rc, std::string("svt_count_merge_begin: ") + svt_last_error(p->ctx)}; });
}
int svh_count_merge_device(svh_pipeline* p, const u64* d_kmer, const u32* d_rev, const u32* d_fwd, u64 n) {
    return guarded(p, [&] { StageTimer t(p, "count.merge"); int rc = svt_count_merge_device(p->ctx, d_kmer, d_rev, d_fwd, n); if (rc != SVT_OK) throw Error{rc, std::string("svt_count_merge_device: ") + svt_last_error(p->ctx)}; });
}
int svh_count_finalize(svh_pipeline* p) {
    return guarded(p, [&] {
        StageTimer t(p, "count.finalize");
        int rc = svt_count_finalize(p->ctx, p->args.kmer_size, p->args.single_strand ? 1 : 0, &p->n_distinct, &p->n_kept);
        if (rc != SVT_OK) throw Error{rc, std::string("svt_count_finalize: ") + svt_last_error(p->ctx)};
        p->table.clear(); p->table_where = 1;
        if (p->n_kept < p->n_distinct / 1000)                                       // src/seq_parse.rs:69-72 on the MERGED table
            throw Error{1, "Less than 0.1% of SNPmers have counts > 1 in both strands and > 2 multiplicity. Consider --single-strand"};
    });
}
// Stage 4a for the clusters ci % world == rank; the raw consensuses are then exchanged (export / import) and svh_consensus_polish
// runs the rest of Stage 4 on all of them
int svh_consensus_poa(svh_pipeline* p, int which, u32 rank, u32 world) {
    return guarded(p, [&] {
        StageTimer t(p, "consensus.poa");
        const auto& cl = which == 0 ? p->kmer_clusters : which == 1 ? p->snp_clusters : p->snp_pre;
        p->poa_which = which;
        p->poa_raw = poa_raw_consensuses(p->rs, p->tw, cl, p->args, rank, world);
    });
}
u32 svh_consensus_raw_count(svh_pipeline* p) { return (u32)p->poa_raw.size(); }
u64 svh_consensus_raw_bytes(svh_pipeline* p) { u64 t = 0; for (auto& c : p->poa_raw) t += c.size(); return t; }
void svh_consensus_raw_export(svh_pipeline* p, u32* len, u8* bytes) {
    u64 o = 0;
    for (size_t i = 0; i < p->poa_raw.size(); i++) { len[i] = (u32)p->poa_raw[i].size(); cpy(bytes + o, p->poa_raw[i].data(), p->poa_raw[i].size()); o += p->poa_raw[i].size(); }
}
// raw consensuses of ANOTHER rank: entry i replaces the local one when the local one is empty (every cluster has exactly one owner)
int svh_consensus_raw_import(svh_pipeline* p, const u32* len, const u8* bytes, u32 n, u64 n_bytes) {
    return guarded(p, [&] {
        if (n != p->poa_raw.size()) throw Error{SVT_ERR_ARG, "svh_consensus_raw_import: cluster count differs between ranks"};
        u64 tot = 0; for (u32 i = 0; i < n; i++) tot += len[i];
        if (tot != n_bytes) throw Error{SVT_ERR_ARG, "svh_consensus_raw_import: the byte buffer does not have sum(len) entries"};
        u64 o = 0;
        for (u32 i = 0; i < n; i++) {
            if (len[i]) {                                                          // every cluster has exactly one owner: the same bytes may arrive again (an all-gather returns the own part too), other bytes may not
                if (p->poa_raw[i].empty()) p->poa_raw[i].assign(bytes + o, bytes + o + len[i]);
                else if (p->poa_raw[i].size() != len[i] || memcmp(p->poa_raw[i].data(), bytes + o, len[i]) != 0) throw Error{SVT_ERR_ARG, "svh_consensus_raw_import: a cluster arrived from two owners with different consensuses"};
            }
            o += len[i];
        }
    });
}
int svh_consensus_polish(svh_pipeline* p) {
    return guarded(p, [&] {
        const auto& cl = p->poa_which == 0 ? p->kmer_clusters : p->poa_which == 1 ? p->snp_clusters : p->snp_pre;
        p->consensuses = assemble_consensuses(cl, std::move(p->poa_raw)); p->poa_raw.clear();
        StageTimer t2(p, "consensus.polish");
        stage4_after_poa(p);
    });
}
// Stage 7 in halves: per-read classes of the twin reads [lo, hi) (svh_em_classes), exchanged as flat arrays (C2), then svh_em_finish
static ClusterArgs em_args(svh_pipeline* p) { ClusterArgs a = p->args; if (p->tw.auto_low_polymorphism) a.low_polymorphism = true; return a; }   // src/main.rs:76-79
int svh_em_begin(svh_pipeline* p) { return guarded(p, [&] { em_init(p->tw, p->asv_off.empty() ? 0 : p->asv_off.size() - 1, p->em); }); }
int svh_em_classes(svh_pipeline* p, u32 lo, u32 hi) {
    return guarded(p, [&] { StageTimer t(p, "em.classes"); if (p->asv_off.size() > 1) em_read_classes(p->rs, p->tw, p->asvs, p->asv_off, em_args(p), lo, hi, p->em); });
}
u64 svh_em_classes_members(svh_pipeline* p, u32 lo, u32 hi) { u64 t = 0; for (u32 r = lo; r < hi && r < p->em.read_class.size(); r++) t += p->em.read_class[r].size(); return t; }
void svh_em_classes_export(svh_pipeline* p, u32 lo, u32 hi, u32* n_best, int32_t* nm, u32* members) {
    u64 o = 0;
    for (u32 r = lo; r < hi; r++) { const auto& c = p->em.read_class[r]; n_best[r - lo] = (u32)c.size(); nm[r - lo] = p->em.read_nm[r]; for (u32 a : c) members[o++] = a; }
}
int svh_em_classes_import(svh_pipeline* p, u32 lo, u32 hi, const u32* n_best, const int32_t* nm, const u32* members, u64 n_members) {
    return guarded(p, [&] {
        if (hi > p->em.read_class.size() || lo > hi) throw Error{SVT_ERR_ARG, "svh_em_classes_import: range outside the twin reads"};
        // the arrays come from another rank: a rank that diverged (other ASV set, other read block) must fail here, not corrupt the counters
        const u64 n_asvs = p->asv_off.empty() ? 0 : p->asv_off.size() - 1;
        u64 tot = 0; for (u32 r = lo; r < hi; r++) tot += n_best[r - lo];
        if (tot != n_members) throw Error{SVT_ERR_ARG, "svh_em_classes_import: the member list does not have sum(n_best) entries"};
        for (u64 x = 0; x < tot; x++) if (members[x] >= n_asvs) throw Error{SVT_ERR_ARG, "svh_em_classes_import: ASV index outside the ASV set of this rank"};
        u64 o = 0;
        for (u32 r = lo; r < hi; r++) {
            const u32 n = n_best[r - lo];
            p->em.read_class[r].assign(members + o, members + o + n); o += n;
            p->em.read_n_best[r] = n; p->em.read_nm[r] = nm[r - lo]; p->em.read_first[r] = n ? p->em.read_class[r][0] : 0;
        }
    });
}
int svh_em_finish(svh_pipeline* p) {
    return guarded(p, [&] { StageTimer t(p, "em.finish"); em_finish(p->tw, p->asv_off.empty() ? 0 : p->asv_off.size() - 1, p->em); });
}

// ---- the whole of `savont asv` in one call, on one GPU or sharded over the ranks of a communicator -------------------------------------
// svh_set_shard_comm: every rank's pipeline joins an RCCL communicator made from the 128 id bytes of svt_shard_comm_id (one rank makes them,
// the caller hands them round).  From then on svh_run_asv deals the work of a pooled read set out over the ranks and the library issues
// every exchange itself (grouped collectives on device memory): a Rust `main.rs` needs these two calls and nothing else for N GPUs.
int svh_set_shard_comm(svh_pipeline* p, u32 rank, u32 world, const u8* comm_id) {
    return guarded(p, [&] { int rc = svt_set_shard_comm(p->ctx, rank, world, comm_id); if (rc != SVT_OK) throw Error{rc, std::string("svt_set_shard_comm: ") + svt_last_error(p->ctx)}; });
}
// One stage of a sharded step has returned `rc` on this rank.  The ranks AGREE on the outcome before anyone goes on (one u64 per rank through the same exchange
// path as everything else), because a rank that went on alone would wait in the next collective for peers that have returned:
//   * every rank 0: go on;
//   * an error every rank shares (bad arguments, "Less than 0.1% of SNPmers ...": the inputs are replicated): every rank returns ITS code and message, as the
//     reference's single process would; the communicator stays usable;
//   * an error only some ranks have: those return theirs, the others SVT_ERR_EXCHANGE naming the first failed rank;
//   * SVT_ERR_EXCHANGE here, or an agreement that cannot complete (a peer died or hangs): the library has aborted the communicator (collectives have a
//     deadline, svt_set_option "shard_timeout_s"), so no rank waits for ever; this rank returns SVT_ERR_EXCHANGE.
// Nothing ends the host process (round 4 called _exit(70): ADVICE r04).
static int shard_agree(svh_pipeline* p, int rc) {
    if (rc == SVT_ERR_EXCHANGE) { svt_shard_abort(p->ctx, "a sharded stage failed in an exchange"); return rc; }
    u32 rank = 0, world = 1; svt_shard_info(p->ctx, &rank, &world);
    if (world <= 1) return rc;
    const int was = svt_shard_pause(p->ctx, 0);                     // the agreement is the same call on every rank whatever the stage paused
    u64 all[64] = {0};
    const int g = svt_shard_allgather_u64(p->ctx, (u64)(u32)rc, all);
    if (was >= 0) svt_shard_pause(p->ctx, was);
    if (g != SVT_OK) {
        const std::string why = svt_last_error(p->ctx);
        svt_shard_abort(p->ctx, "the ranks could not agree on the outcome of a sharded stage");
        if (rc == 0) { p->err = "a peer rank left the sharded step: " + why; return SVT_ERR_EXCHANGE; }
        return rc;
    }
    if (rc != 0) return rc;
    for (u32 r = 0; r < world; r++)
        if ((u32)all[r] != 0) { p->err = "rank " + std::to_string(r) + " of " + std::to_string(world) + " failed in this stage of the sharded step (code " + std::to_string((int)(u32)all[r]) + "); see its error"; return SVT_ERR_EXCHANGE; }
    return 0;
}
// the sharded halves of C1 / Stage 4a / C2, each ONE call on every rank
int svh_count_shard_merge(svh_pipeline* p) {
    return guarded(p, [&] {
        StageTimer t(p, "count.merge");
        int rc = svt_count_shard_merge(p->ctx, p->args.kmer_size, p->args.single_strand ? 1 : 0, &p->n_distinct, &p->n_kept);
        if (rc != SVT_OK) throw Error{rc, std::string("svt_count_shard_merge: ") + svt_last_error(p->ctx)};
        p->table.clear(); p->table_where = 1;
        if (p->n_kept < p->n_distinct / 1000)                                       // src/seq_parse.rs:69-72 on the MERGED table
            throw Error{1, "Less than 0.1% of SNPmers have counts > 1 in both strands and > 2 multiplicity. Consider --single-strand"};
    });
}
// every rank called SNPmers on the identical merged table: a digest of the list is gathered and compared (a rank that diverged fails loudly)
int svh_snpmers_check_ranks(svh_pipeline* p) {
    return guarded(p, [&] {
        u64 h = 0xcbf29ce484222325ull;
        for (const SnpmerInfo& s : p->info.snpmer_info) { for (u64 v : {(u64)s.split_kmer, (u64)s.mid_bases[0] | (u64)s.mid_bases[1] << 8 | (u64)s.counts[0] << 16 | (u64)s.counts[1] << 40}) { h ^= v; h *= 0x100000001b3ull; h ^= h >> 29; } }
        u32 rank = 0, world = 1; svt_shard_info(p->ctx, &rank, &world);
        std::vector<u64> all(std::max<u32>(world, 1));
        int rc = svt_shard_allgather_u64(p->ctx, h, all.data());
        if (rc != SVT_OK) throw Error{rc, std::string("svt_shard_allgather_u64: ") + svt_last_error(p->ctx)};
        for (u32 r = 0; r < world; r++) if (all[r] != all[0]) throw Error{SVT_ERR_STATE, "the SNPmer lists of the ranks differ (the merged count tables are not identical)"};
    });
}
// raw consensuses of the clusters this rank owns -> all of them on every rank (every cluster has exactly one owner)
int svh_consensus_gather(svh_pipeline* p) {
    return guarded(p, [&] {
        StageTimer t(p, "consensus.allgather");
        u32 rank = 0, world = 1; svt_shard_info(p->ctx, &rank, &world);
        if (world <= 1) return;
        const size_t nc = p->poa_raw.size();
        std::vector<u8> mine(nc * 4);
        for (size_t i = 0; i < nc; i++) { const u32 l = (u32)p->poa_raw[i].size(); cpy(mine.data() + 4 * i, &l, 4); }
        for (size_t i = 0; i < nc; i++) mine.insert(mine.end(), p->poa_raw[i].begin(), p->poa_raw[i].end());
        std::vector<u64> bytes(world);
        int rc = svt_shard_allgather_u64(p->ctx, mine.size(), bytes.data());
        if (rc != SVT_OK) throw Error{rc, std::string("svt_shard_allgather_u64: ") + svt_last_error(p->ctx)};
        u64 tot = 0; for (u64 b : bytes) tot += b;
        std::vector<u8> all(tot + 1);
        rc = svt_shard_allgatherv(p->ctx, mine.data(), bytes.data(), all.data());
        if (rc != SVT_OK) throw Error{rc, std::string("svt_shard_allgatherv: ") + svt_last_error(p->ctx)};
        u64 o = 0;
        for (u32 r = 0; r < world; r++) {
            if (r != rank) {
                if (bytes[r] < nc * 4) throw Error{SVT_ERR_STATE, "svh_consensus_gather: cluster count differs between ranks"};
                const u8* len = all.data() + o; u64 q = o + nc * 4;
                for (size_t i = 0; i < nc; i++) {
                    u32 l; cpy(&l, len + 4 * i, 4);
                    if (q + l > o + bytes[r]) throw Error{SVT_ERR_STATE, "svh_consensus_gather: a rank's record is shorter than its lengths say"};
                    if (l) {
                        if (p->poa_raw[i].empty()) p->poa_raw[i].assign(all.data() + q, all.data() + q + l);
                        else if (p->poa_raw[i].size() != l || memcmp(p->poa_raw[i].data(), all.data() + q, l) != 0) throw Error{SVT_ERR_STATE, "svh_consensus_gather: a cluster arrived from two owners with different consensuses"};
                    }
                    q += l;
                }
            }
            o += bytes[r];
        }
    });
}
// C2 (src/alignment.rs:1918-1920): the per-read classes of every rank's read block -> all reads on every rank
static inline u32 block_lo(u64 n, u32 r, u32 W) { return (u32)(n * r / W); }
int svh_em_classes_gather(svh_pipeline* p) {
    return guarded(p, [&] {
        StageTimer t(p, "em.allgather");
        u32 rank = 0, world = 1; svt_shard_info(p->ctx, &rank, &world);
        if (world <= 1) return;
        const u32 nt = p->tw.n, lo = block_lo(nt, rank, world), hi = block_lo(nt, rank + 1, world);
        const u64 n_asvs = p->asv_off.empty() ? 0 : p->asv_off.size() - 1;
        std::vector<u32> mine; mine.reserve((size_t)(hi - lo) * 3);
        for (u32 r = lo; r < hi; r++) { mine.push_back((u32)p->em.read_class[r].size()); mine.push_back((u32)p->em.read_nm[r]); }
        for (u32 r = lo; r < hi; r++) for (u32 a : p->em.read_class[r]) mine.push_back(a);
        std::vector<u64> bytes(world);
        int rc = svt_shard_allgather_u64(p->ctx, mine.size() * 4, bytes.data());
        if (rc != SVT_OK) throw Error{rc, std::string("svt_shard_allgather_u64: ") + svt_last_error(p->ctx)};
        u64 tot = 0; for (u64 b : bytes) tot += b;
        std::vector<u32> all(tot / 4 + 1);
        rc = svt_shard_allgatherv(p->ctx, mine.data(), bytes.data(), all.data());
        if (rc != SVT_OK) throw Error{rc, std::string("svt_shard_allgatherv: ") + svt_last_error(p->ctx)};
        u64 o = 0;
        for (u32 r = 0; r < world; r++) {
            const u32 rlo = block_lo(nt, r, world), rhi = block_lo(nt, r + 1, world);
            const u64 words = bytes[r] / 4;
            if (r != rank) {
                if (words < (u64)(rhi - rlo) * 2) throw Error{SVT_ERR_STATE, "svh_em_classes_gather: a rank's record is shorter than its read block"};
                const u32* hd = all.data() + o; u64 q = o + (u64)(rhi - rlo) * 2;
                for (u32 x = rlo; x < rhi; x++) {
                    const u32 n = hd[2 * (x - rlo)];
                    if (q + n > o + words) throw Error{SVT_ERR_STATE, "svh_em_classes_gather: a rank's member list is shorter than its class sizes say"};
                    for (u32 j = 0; j < n; j++) if (all[q + j] >= n_asvs) throw Error{SVT_ERR_STATE, "svh_em_classes_gather: ASV index outside the ASV set of this rank"};
                    p->em.read_class[x].assign(all.begin() + q, all.begin() + q + n); q += n;
                    p->em.read_n_best[x] = n; p->em.read_nm[x] = (int32_t)hd[2 * (x - rlo) + 1]; p->em.read_first[x] = n ? p->em.read_class[x][0] : 0;
                }
            }
            o += words;
        }
    });
}
// src/main.rs:49-152 (run_cluster) from the resident reads to the EM depths.  With a communicator of world > 1 (svh_set_shard_comm, or a hook set
// on svh_ctx with svt_set_shard) the stages are dealt out as DESIGN.md section 9 says: counting and Stage 7 by read block, the K5 / K6 tiles under the
// replicated greedy loops by slice, Stage 3 by k-mer cluster, POA and polish by cluster -- results identical to the one-rank run.
int svh_run_asv(svh_pipeline* p) {
    u32 rank = 0, world = 1; svt_shard_info(p->ctx, &rank, &world);
    const bool sh = world > 1;
    // the tile slicing is paused and resumed below as the ranks' calls diverge and meet again: whatever the way out, the context leaves as it came
    struct PauseGuard { svt_ctx* c; int was; ~PauseGuard() { if (was >= 0) svt_shard_pause(c, was); } } guard{p->ctx, sh ? svt_shard_pause(p->ctx, 0) : -1};
    if (sh && guard.was > 0) svt_shard_pause(p->ctx, guard.was);
    auto step = [&](int rc) { return sh ? shard_agree(p, rc) : rc; };
    int rc = 0;
#define RUN(x) do { if ((rc = step(x)) != 0) return rc; } while (0)
    if (!sh) {
        RUN(svh_read_to_split_kmers(p)); RUN(svh_get_snpmers(p));
    } else {
        svt_shard_pause(p->ctx, 1);                                                   // the ranks make DIFFERENT calls: no tile slicing
        u64 nd = 0;
        RUN(svh_count_partial_device(p, block_lo(p->rs.n, rank, world), block_lo(p->rs.n, rank + 1, world), &nd));
        RUN(svh_count_shard_merge(p)); RUN(svh_get_snpmers(p)); RUN(svh_snpmers_check_ranks(p));
        svt_shard_pause(p->ctx, 0);                                                   // the same calls on every rank from here ...
    }
    RUN(svh_twin_reads(p)); RUN(svh_cluster_reads_by_kmers(p)); RUN(svh_cluster_reads_by_snpmers(p));
    if (!sh) {
        RUN(svh_consensus(p, 1));
    } else {
        svt_shard_pause(p->ctx, 1);                                                   // ... to here: POA by cluster and Stage 7 by read block are rank-dependent
        RUN(svh_consensus_poa(p, 1, rank, world)); RUN(svh_consensus_gather(p)); RUN(svh_consensus_polish(p));
    }
    RUN(svh_merge_similar_consensuses(p)); RUN(svh_detect_chimeras(p)); RUN(svh_consensus_to_asvs(p));
    if (!sh) {
        RUN(svh_refine_asv_depths_with_em(p));
    } else {
        RUN(svh_em_begin(p)); RUN(svh_em_classes(p, block_lo(p->tw.n, rank, world), block_lo(p->tw.n, rank + 1, world)));
        RUN(svh_em_classes_gather(p)); RUN(svh_em_finish(p));
    }
#undef RUN
    return 0;
}

// ---- Stage 4: consensus + pile-up confidence (src/main.rs:84-110) ----------------------------------------
// which: 1 = SNPmer clusters (the reference's input), 0 = k-mer clusters, 2 = clusters before reclustering
int svh_consensus(svh_pipeline* p, int which) {
    return guarded(p, [&] {
        StageTimer t(p, "consensus");
        const auto& cl = which == 0 ? p->kmer_clusters : which == 1 ? p->snp_clusters : p->snp_pre;
        { StageTimer t1(p, "consensus.poa"); p->consensuses = align_and_consensus(p->rs, p->tw, cl, p->args); }
        StageTimer t2(p, "consensus.polish");
        stage4_after_poa(p);
    });
}
// Stage 5 + 6 on the Stage-4 result (src/main.rs:115-130): p->consensuses becomes the final consensus list
int svh_merge_similar_consensuses(svh_pipeline* p) {
    return guarded(p, [&] {
        StageTimer t(p, "merge");
        p->consensuses = merge_similar_consensuses(p->rs, std::move(p->consensuses), p->low_qual, p->args);
        p->n_after_merge = (u32)p->consensuses.size();
        if (!p->temp_dir.empty()) {                                                   // src/alignment.rs:1506-1513
            write_clusters_tsv(as_records(p->consensuses), p->rs, p->tw, p->temp_dir + "/final_clusters_merged_stage5.tsv", "final", false);
            write_consensus_fasta(as_records(p->consensuses), p->temp_dir + "/merged_consensus_sequences.fasta", "merged");
        }
    });
}
int svh_detect_chimeras(svh_pipeline* p) {
    return guarded(p, [&] {
        StageTimer t(p, "chimera");
        p->chimera_ids.clear();
        if (p->args.skip_chimera_detection) return;
        std::vector<u32> idx; std::vector<u64> ids;
        for (auto& c : p->consensuses) ids.push_back(c.id);
        p->consensuses = detect_and_filter_chimeras(p->rs, std::move(p->consensuses), p->args, &idx);
        for (u32 i : idx) p->chimera_ids.push_back((u32)ids[i]);
    });
}
u32 svh_chimera_count(svh_pipeline* p) { return (u32)p->chimera_ids.size(); }
void svh_chimera_fetch(svh_pipeline* p, u32* ids) { cpy(ids, p->chimera_ids.data(), p->chimera_ids.size() * 4); }   // debug ids (cluster index) of removed consensuses
// window minimizers of the Stage-5 de-duplication (src/seeding.rs:99-186), host only; returns the count (<= cap)
u64 svh_minimizer_seeds(const u8* seq, u64 len, u32 w, u32 k, u64* out, u64 cap) {
    std::vector<u64> v = minimizer_seeds(seq, len, w, k);
    for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    return v.size();
}
// set = 0: kept consensuses, 1: low-quality consensuses; sequences are the decompressed (N-trimmed) ones
u32 svh_consensus_count(svh_pipeline* p, int set) { return (u32)(set ? p->low_qual : p->consensuses).size(); }
u64 svh_consensus_bases(svh_pipeline* p, int set) { u64 t = 0; for (auto& c : (set ? p->low_qual : p->consensuses)) t += c.decompressed.size(); return t; }
void svh_consensus_fetch(svh_pipeline* p, int set, u8* seq, u64* off, u64* depth, u64* id, u32* n_lowq) {
    const auto& v = set ? p->low_qual : p->consensuses;
    u64 o = 0;
    for (size_t i = 0; i < v.size(); i++) {
        off[i] = o; cpy(seq + o, v[i].decompressed.data(), v[i].decompressed.size()); o += v[i].decompressed.size();
        if (depth) depth[i] = v[i].depth; if (id) id[i] = v[i].id; if (n_lowq) n_lowq[i] = (u32)v[i].low_quality_positions.size();
    }
    off[v.size()] = o;
}
// test hooks: the pile-ups and the consensuses as they were BEFORE analyze_pileup_consensuses (set 2)
void svh_keep_pileups(svh_pipeline* p, int keep) { p->keep_pileups = keep != 0; }
u64 svh_pileup_entries(svh_pipeline* p, u32 ci) { u64 t = 0; for (auto& c : p->pileups[ci]) t += c.entries.size(); return t; }
void svh_pileup_fetch(svh_pipeline* p, u32 ci, u64* col_off, u8* kind, u8* base, u8* qual) {
    u64 o = 0; size_t i = 0;
    for (auto& c : p->pileups[ci]) { col_off[i++] = o; for (auto& e : c.entries) { kind[o] = e.kind; base[o] = e.base; qual[o] = e.qual; o++; } }
    col_off[i] = o;
}
void svh_pileup_fetch_hp(svh_pipeline* p, u32 ci, u8* hp) { u64 o = 0; for (auto& c : p->pileups[ci]) for (auto& e : c.entries) hp[o++] = e.hp; }
u32 svh_raw_consensus_count(svh_pipeline* p) { return (u32)p->raw_consensuses.size(); }
u64 svh_raw_consensus_len(svh_pipeline* p, u32 ci) { return p->raw_consensuses[ci].sequence.size(); }
void svh_raw_consensus_fetch(svh_pipeline* p, u32 ci, u8* seq, u64* depth, u64* id, u64* n_members) {
    const ConsensusSequence& c = p->raw_consensuses[ci];
    cpy(seq, c.sequence.data(), c.sequence.size()); *depth = c.depth; *id = c.id; *n_members = c.cluster.size();
}
u32 svh_quality_map(svh_pipeline* p, u8* q, double* rate) { u32 i = 0; for (auto& kv : p->qmap) { if (q) { q[i] = kv.first; rate[i] = kv.second; } i++; } return i; }
// stateless POA (host only): n sequences + per-base weights -> consensus; returns its length (<= cap) or -1
int svh_poa_consensus(const u8* seq, const u8* weights, const u64* off, u32 n, u8* out, u64 cap, u64* graph_nodes, int wide_cells) {
    try {
        std::vector<std::vector<u8>> s(n), w(n);
        for (u32 i = 0; i < n; i++) { s[i].assign(seq + off[i], seq + off[i + 1]); if (weights) w[i].assign(weights + off[i], weights + off[i + 1]); else w[i].assign(s[i].size(), 1); }
        std::vector<u8> c = poa_consensus(s, w, graph_nodes, (wide_cells & 1) != 0, (wide_cells & 2) != 0);   // bit 1: --no-band (src/alignment.rs:198,217)
        if (c.size() > cap) return -1;
        cpy(out, c.data(), c.size());
        return (int)c.size();
    } catch (...) { return -1; }
}
// clusters: cl_off[n_clusters+1] ranges over the n sequences; consensus of every cluster -> out (concatenated) + out_off
// engine: 0 host DP, 2 K12 (graphs resident on the device), 3 K12 for poa_device_share percent of the clusters; graph_nodes (nullable): nodes of every cluster's final graph
int svh_poa_consensus_batch(svh_pipeline* p, int engine, const u8* seq, const u8* weights, const u64* off, const u64* cl_off, u32 n_clusters, u8* out, u64* out_off, u64 cap, u64* graph_nodes) {
    return guarded(p, [&] {
        std::vector<PoaInput> in(n_clusters);
        for (u32 c = 0; c < n_clusters; c++) for (u64 i = cl_off[c]; i < cl_off[c + 1]; i++) {
            in[c].seqs.emplace_back(seq + off[i], seq + off[i + 1]);
            if (weights) in[c].quals.emplace_back(weights + off[i], weights + off[i + 1]); else in[c].quals.emplace_back(off[i + 1] - off[i], (u8)1);
        }
        std::vector<u64> gn;
        if (engine < -1 || engine > 3 || engine == 1) throw Error{SVT_ERR_ARG, "svh_poa_consensus_batch: engine is -1 (by CPU share), 0 (host DP), 2 (K12) or 3 (K12 for poa_device_share percent of the clusters)"};
        const int eng = engine == 3 ? 100 + p->args.tuning.poa_device_share : engine;       // the share encoding poa_consensus_batch reads (as poa_raw_consensuses does)
        auto res = poa_consensus_batch(engine ? p->ctx : nullptr, in, eng, false, &gn);
        if (graph_nodes) for (u32 c = 0; c < n_clusters; c++) graph_nodes[c] = gn[c];
        u64 o = 0;
        for (u32 c = 0; c < n_clusters; c++) { out_off[c] = o; if (o + res[c].size() > cap) throw Error{SVT_ERR_OVERFLOW, "svh_poa_consensus_batch: output buffer too small"}; cpy(out + o, res[c].data(), res[c].size()); o += res[c].size(); }
        out_off[n_clusters] = o;
    });
}
// the kept consensuses become the ASV set of Stage 7 (the reference runs Stage 5/6 in between)
int svh_consensus_to_asvs(svh_pipeline* p) {
    return guarded(p, [&] {
        std::vector<u8> seq; std::vector<u64> off(1, 0);
        for (auto& c : p->consensuses) { seq.insert(seq.end(), c.decompressed.begin(), c.decompressed.end()); off.push_back(seq.size()); }
        if (p->asvs) { svt_batch_free(p->ctx, p->asvs); p->asvs = nullptr; }
        p->asv_off = off;
        if (!p->temp_dir.empty()) write_consensus_fasta(as_records(p->consensuses), p->temp_dir + "/final_asvs_for_em.fasta", "em_refinement");   // src/alignment.rs:1747-1749
        int rc = svt_batch_upload(p->ctx, seq.data(), nullptr, off.data(), (u32)p->consensuses.size(), &p->asvs);
        if (rc != SVT_OK) throw Error{rc, std::string("svt_batch_upload(asvs): ") + svt_last_error(p->ctx)};
    });
}

int svh_set_asvs(svh_pipeline* p, const u8* seq, const u64* offsets, u32 n) {
    return guarded(p, [&] {
        StageTimer t(p, "upload_asvs");
        if (p->asvs) { svt_batch_free(p->ctx, p->asvs); p->asvs = nullptr; }
        p->asv_off.assign(offsets, offsets + n + 1);
        for (auto& x : p->asv_off) x -= offsets[0];
        int rc = svt_batch_upload(p->ctx, seq, nullptr, offsets, n, &p->asvs);
        if (rc != SVT_OK) throw Error{rc, std::string("svt_batch_upload(asvs): ") + svt_last_error(p->ctx)};
    });
}
int svh_refine_asv_depths_with_em(svh_pipeline* p) {
    return guarded(p, [&] {
        StageTimer t(p, "em");
        ClusterArgs a = p->args;
        if (p->tw.auto_low_polymorphism) a.low_polymorphism = true;              // src/main.rs:76-79
        const bool dump = !p->temp_dir.empty();
        refine_asv_depths_with_em(p->rs, p->tw, p->asvs, p->asv_off, a, dump, p->em);
        if (dump) {                                                              // src/alignment.rs:1538-1541 / :1739-1742
            std::vector<size_t> ids;                                             // the ASV set is the Stage 4-6 result (ids known) or one handed in by svh_set_asvs (index)
            const size_t na = p->asv_off.empty() ? 0 : p->asv_off.size() - 1;
            if (p->consensuses.size() == na) for (auto& c : p->consensuses) ids.push_back(c.id);
            write_read_to_asv_mappings(p->em, ids, p->rs, p->tw, a.low_polymorphism, p->temp_dir + "/read_to_asv_mappings.tsv");
            p->em.read_lines.clear(); p->em.read_lines.shrink_to_fit();
        }
    });
}
void svh_em_fetch(svh_pipeline* p, u64* depth, u64* un, u64* am, u64* l10, u64* total, u64* filtered, int* kept_original) {
    size_t n = p->em.depth.size();
    if (depth) cpy(depth, p->em.depth.data(), n * 8); if (un) cpy(un, p->em.unambig.data(), n * 8);
    if (am) cpy(am, p->em.ambig.data(), n * 8); if (l10) cpy(l10, p->em.leq10.data(), n * 8);
    if (total) *total = p->em.total_assigned; if (filtered) *filtered = p->em.filtered; if (kept_original) *kept_original = p->em.kept_original;
}
void svh_em_read_assignments(svh_pipeline* p, u32* nb, int32_t* nm, u32* first) {
    size_t n = p->em.read_n_best.size();
    if (nb) cpy(nb, p->em.read_n_best.data(), n * 4); if (nm) cpy(nm, p->em.read_nm.data(), n * 4); if (first) cpy(first, p->em.read_first.data(), n * 4);
}
int svh_compute_per_sample_depths(svh_pipeline* p, u32 n_samples, u64* out) {
    return guarded(p, [&] {
        StageTimer t(p, "per_sample");
        auto r = compute_per_sample_depths(p->tw, p->em, n_samples, p->em.depth.size());
        for (size_t a = 0; a < r.size(); a++) for (u32 s = 0; s < n_samples; s++) out[a * n_samples + s] = r[a][s];
    });
}

// stateless ingest check (no GPU): record count, bases, FNV-1a over ids / sequences / qualities of one file; returns 0 or -1
int svh_fastx_digest(const char* path, u64* n_records, u64* n_bases, int* has_qual, u64* digest, char* err, u64 err_cap) {
    try {
        RawBytes seq, qual; std::vector<u64> off; std::vector<std::string> ids; bool q = false;
        std::vector<std::string> files; std::vector<u32> file_idx;          // '\n'-joined paths: the files of svh_load_fastx, inflated and parsed side by side
        for (const char* c = path; c && *c;) { const char* e = strchr(c, '\n'); files.push_back(e ? std::string(c, e) : std::string(c)); c = e ? e + 1 : nullptr; }
        read_fastx_files(files, seq, qual, off, ids, file_idx, q);
        u64 h = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t n) { const u8* b = (const u8*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
        for (size_t i = 0; i < ids.size(); i++) { mix(ids[i].data(), ids[i].size()); mix("\n", 1); mix(seq.data() + off[i], off[i + 1] - off[i]); mix("\n", 1); if (q) mix(qual.data() + off[i], off[i + 1] - off[i]); mix("\n", 1); }
        *n_records = ids.size(); *n_bases = seq.size(); *has_qual = q; *digest = h;
        return 0;
    } catch (const Error& e) { if (err && err_cap) { strncpy(err, e.msg.c_str(), err_cap - 1); err[err_cap - 1] = 0; } return -1; }
}

// stateless gz check (no GPU): one .gz file inflated whole by zlib (decoder 0) or by host/inflate.hpp (decoder 1) -> bytes, FNV-1a of the bytes, seconds of the
// inflate alone (CRC check included, hashing not).  tests/test_io.py holds the two decoders against each other; bench.py reports the seconds.  Returns 0 or -1.
int svh_gunzip_digest(const char* path, int decoder, u64* n_bytes, u64* digest, double* seconds, char* err, u64 err_cap) {
    auto fail = [&](const std::string& m) { if (err && err_cap) { strncpy(err, m.c_str(), err_cap - 1); err[err_cap - 1] = 0; } return -1; };
    if (!path || !n_bytes || !digest) return fail("svh_gunzip_digest: null argument");
    std::vector<u8> zout; gz::BigBuf own; const u8* data = nullptr; size_t len = 0;
    const auto t0 = std::chrono::steady_clock::now();
    if (decoder == 0) {
        gzFile f = gzopen(path, "rb");
        if (!f) return fail(std::string("cannot open ") + path);
        gzbuffer(f, 1 << 20);
        zout.resize((size_t)1 << 24);
        for (;;) {
            if (zout.size() - len < ((size_t)1 << 22)) zout.resize(zout.size() * 2);
            const int n = gzread(f, zout.data() + len, (unsigned)std::min<size_t>(zout.size() - len, (size_t)1 << 30));
            if (n < 0) { gzclose(f); return fail("zlib: corrupt or truncated gzip stream"); }
            if (n == 0) break;
            len += (size_t)n;
        }
        int en = 0; gzerror(f, &en); gzclose(f);
        if (en != Z_OK && en != Z_STREAM_END) return fail("zlib: corrupt or truncated gzip stream");
        data = zout.data();
    } else {
        FILE* fp = fopen(path, "rb");
        if (!fp) return fail(std::string("cannot open ") + path);
        std::vector<u8> src; u8 chunk[1 << 16]; size_t n;
        while ((n = fread(chunk, 1, sizeof chunk, fp)) > 0) src.insert(src.end(), chunk, chunk + n);
        fclose(fp);
        std::string why;
        const auto t1 = std::chrono::steady_clock::now();
        (void)gz_threads_now();                                                  // installs the worker-pool hooks of the parallel path
        if (!gz::gunzip_all(src.data(), src.size(), own, len, why, decoder >= 2 ? (unsigned)decoder : 1u)) return fail("inflate.hpp: " + why);     // decoder n >= 2: one member on n threads
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        data = own.p;
    }
    if (seconds && decoder == 0) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    u64 h = 1469598103934665603ull;
    for (size_t i = 0; i < len; i++) { h ^= data[i]; h *= 1099511628211ull; }
    *n_bytes = len; *digest = h;
    return 0;
}

// ---- stateless host entry points (no GPU): statistics + SNPmer calling on a given count table -------------
double svh_binomial_test(u64 n, u64 k, double p) { return binomial_test(n, k, p); }
double svh_fisher_two_tail(u32 a, u32 b, u32 c, u32 d) { return fisher_two_tail(a, b, c, d); }
// table must be in the order of svt_count_fetch; outputs sized by the caller (<= n/2 sites, <= n high-freq); returns n_sites
int svh_snpmers_from_table(const u64* km, const u32* rev, const u32* fwd, u64 n, u32 k, int single_strand,
                           u64* split, u8* m0, u8* m1, u32* c0, u32* c1, u64* hf, u32* n_hf, u32* thresh) {
    try {
        KmerCountTable t(n);
        for (u64 i = 0; i < n; i++) t[i] = {km[i], {rev[i], fwd[i]}};
        ClusterArgs a; a.single_strand = single_strand != 0; a.kmer_size = k;
        KmerGlobalInfo info = get_snpmers_inplace_sort(t, k, a);
        for (size_t i = 0; i < info.snpmer_info.size(); i++) {
            const SnpmerInfo& s = info.snpmer_info[i];
            split[i] = s.split_kmer; m0[i] = s.mid_bases[0]; m1[i] = s.mid_bases[1]; c0[i] = s.counts[0]; c1[i] = s.counts[1];
        }
        if (!info.high_freq_kmers.empty()) cpy(hf, info.high_freq_kmers.data(), info.high_freq_kmers.size() * 8);
        *n_hf = (u32)info.high_freq_kmers.size(); *thresh = (u32)info.high_freq_thresh;
        return (int)info.snpmer_info.size();
    } catch (...) { return -1; }
}

// ---- synthetic amplicon reads: deterministic, splitmix64/xoshiro256** -----------------
static inline u64 splitmix(u64& s) { u64 z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
struct Xo { u64 s[4]; explicit Xo(u64 seed) { for (auto& x : s) x = splitmix(seed); }
    static inline u64 rotl(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
    u64 next() { u64 r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    u32 below(u32 n) { return (u32)(((next() >> 32) * (u64)n) >> 32); } };

// haplotypes: concatenated ASCII + offsets[n_hap+1]; weights[n_hap] (relative abundances).
// Output buffers must hold n_reads * (max_hap_len * 1.2 + 16) bytes; returns total bases written.
u64 svh_synth_reads(const u8* hap_seq, const u64* hap_off, u32 n_hap, const double* weights, u32 n_reads, u64 seed,
                    u8* seq_out, u8* qual_out, u64* off_out, u32* hap_of_read, u8* strand_of_read) {
    Xo rng(seed);
    std::vector<double> cum(n_hap); double tot = 0; for (u32 i = 0; i < n_hap; i++) { tot += weights[i]; cum[i] = tot; }
    static const char comp[4] = {'T', 'G', 'C', 'A'}; static const char acgt[4] = {'A', 'C', 'G', 'T'};
    auto code = [](u8 b) { switch (b) { case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 0; } };
    u64 o = 0;
    std::vector<u8> tmp;
    double perr_of_q[64];                                                        // 10^(-q/10), q <= 50: the same doubles the per-base pow() gave (150 M calls per 100k reads)
    for (u32 q = 0; q < 64; q++) perr_of_q[q] = std::pow(10.0, -(double)q / 10.0);
    for (u32 r = 0; r < n_reads; r++) {
        double u = rng.uni() * tot; u32 h = (u32)(std::lower_bound(cum.begin(), cum.end(), u) - cum.begin()); if (h >= n_hap) h = n_hap - 1;
        const u8* hs = hap_seq + hap_off[h]; u32 hl = (u32)(hap_off[h + 1] - hap_off[h]);
        bool rev = rng.next() & 1;
        tmp.resize(hl);
        if (!rev) cpy(tmp.data(), hs, hl); else for (u32 i = 0; i < hl; i++) tmp[i] = (u8)comp[code(hs[hl - 1 - i])];
        off_out[r] = o;
        // per-read quality regime: mostly Q30-40 with an occasional worse read
        const double worse = rng.uni();
        const u32 qmode = worse < 0.08 ? 18 + rng.below(8) : 30 + rng.below(11);
        for (u32 i = 0; i < hl; i++) {
            // per-base quality: discretised around the read mode, 5 % low tail (Q5-15); never constant per read
            u32 q; const double t = rng.uni();
            if (t < 0.05) q = 5 + rng.below(11); else { int d = (int)rng.below(13) - 6; int qq = (int)qmode + d; q = (u32)(qq < 2 ? 2 : (qq > 50 ? 50 : qq)); }
            const double perr = perr_of_q[q];
            const bool hp = i > 0 && tmp[i] == tmp[i - 1];
            double pe = perr; const double e = rng.uni();
            // errors consistent with the emitted quality: 40/30/30 sub/ins/del, homopolymer indels x3
            const double p_sub = 0.4 * pe, p_ins = 0.3 * pe * (hp ? 3.0 : 1.0), p_del = 0.3 * pe * (hp ? 3.0 : 1.0);
            if (e < p_sub) { u32 b = (code(tmp[i]) + 1 + rng.below(3)) & 3; seq_out[o] = (u8)acgt[b]; qual_out[o] = (u8)(q + 33); o++; }
            else if (e < p_sub + p_ins) { seq_out[o] = tmp[i]; qual_out[o] = (u8)(q + 33); o++; seq_out[o] = hp ? tmp[i] : (u8)acgt[rng.below(4)]; qual_out[o] = (u8)(q + 33); o++; }
            else if (e < p_sub + p_ins + p_del) { /* deletion */ }
            else { seq_out[o] = tmp[i]; qual_out[o] = (u8)(q + 33); o++; }
        }
        if (hap_of_read) hap_of_read[r] = h; if (strand_of_read) strand_of_read[r] = rev;
    }
    off_out[n_reads] = o;
    return o;
}

}  // extern "C"
