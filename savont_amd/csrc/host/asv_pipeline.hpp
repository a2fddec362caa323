// asv_pipeline.hpp -- C++ host side of the `savont asv` hot path, ABOVE the C-ABI of
// include/savont_hip.h.  It mirrors the reference's stage functions (same names, argument meaning
// and error behaviour; src/main.rs:49-201 is the call order) and keeps exactly the logic the
// reference runs sequentially on the CPU (greedy representative choice, cluster merging, EM);
// every data-parallel inner loop is a call into libsavont_hip.so.  Nothing here links oracle/.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "savont_hip.h"

namespace savont {

// Implementation choices with IDENTICAL results (block schedules, engines); set through svh_set_option, never through the environment.
// Tests shrink the blocks so that the cut / re-queue paths that only large inputs reach run on small ones.
struct Tuning {
    uint32_t stage2_first_block = 256, stage2_max_block = 32768;   // Stage 2 blocks double from first to max
    uint64_t stage2_pair_cap = (uint64_t)2 << 20;                  // pass-2 pairs per block before the block is shortened
    int stage2_device = -1;                                        // the candidate lists of a Stage-2 block: 1 from the device (svt_lsh_candidates: every (read, representative) pair compared directly), 0 the host's bucket walk on the worker pool, -1 by the CPU share of this process (device at <= 10 pool threads: measured 1.38 against 1.28 M reads/s at 2 CPUs, 2.25 against 2.28 M at 16)
    uint32_t stage3_first_block = 128, stage3_block = 2048, stage3_max_block = 16384, stage3_switch = 4096;
    int stage3_waves = 1;                                          // 1: the next block of every k-mer cluster in ONE device call (svt_snpmer_compat_lists_seg); 0: one call per (cluster, block) on forked contexts
    int poa_engine = -1;                                           // -1 by the CPU share of this process (K12 when it has at most 10 CPUs: the host DP needs 0.6 CPU-s per 100k-read step, K12 130-180 ms of latency and no CPU; the host DP otherwise), 0 host DP on the worker pool, 2 K12: graphs resident on the GPU, one launch, 3 K12 for poa_device_share percent of the clusters while the host DP does the others
    int poa_device_share = 35;
    int poa_cells = 16;                                            // 32: the plain int32 DP (equality tests of the SIMD 16-bit paths)
    int nm_contract = 1;                                           // Stage 7 `nm`: 0 = K8 unit-cost overlap distance, 1 = K8a (minimap2-style affine local nm) near the unit-cost optimum, 2 = K8a in the whole band
};

struct ClusterArgs {                       // src/cli.rs:46-187 (fields on the hot path)
    Tuning tuning;
    uint32_t kmer_size = 17;               // :153
    uint32_t c = 11;                       // :83
    uint32_t min_read_length = 1100;       // :87
    uint32_t max_read_length = 2000;       // :91
    double quality_value_cutoff = 98.0;    // :95
    uint8_t minimum_base_quality = 25;     // :99
    bool single_strand = false;            // :103
    bool no_snpmers = false;               // :145 (hidden) skip SNPmer detection: no sites, the high-frequency list stays (src/kmer_comp.rs:525,689)
    bool no_band = false;                  // :183 (hidden) unbanded POA (src/alignment.rs:198,217): the host engine with a band that holds every column; K12 is not used
    bool use_hpc = false;                  // :118-120 homopolymer-compressed Stage 4 (POA + pile-ups; the consensus is decompressed before Stage 5)
    uint32_t min_cluster_size = 12;        // :107
    uint32_t max_iterations_recluster = 10;  // :132
    double primary_clustering_threshold = 0.95;  // :185
    bool low_polymorphism = false;         // :143
    uint32_t align_band = 0;               // K8 band half width; 0 = max(ceil(max(Lq,Lt)/13), |Lq-Lt|) capped at 511
    uint32_t n_depth_cutoff = 250;         // :116
    double posterior_threshold_ln = 30.0;  // :129
    bool mask_low_quality = false;         // :125
    uint32_t chimera_allowable_errors = 1; // :166
    uint32_t chimera_detect_length = 0;    // :170 (0 = None -> max(min_read_length / 10, 100))
    bool skip_chimera_detection = false;   // :141
};

struct SnpmerInfo {                        // src/types.rs:818-824
    uint64_t split_kmer; uint8_t mid_bases[2]; uint32_t counts[2]; uint8_t k;
};
struct KmerGlobalInfo {                    // src/types.rs:800-808 (live fields)
    std::vector<SnpmerInfo> snpmer_info;
    std::vector<uint64_t> high_freq_kmers; // sorted
    double high_freq_thresh = 0;
};
typedef std::vector<std::pair<uint64_t, std::pair<uint32_t, uint32_t>>> KmerCountTable;   // (kmer, [rev, fwd])

struct Error { int code; std::string msg; };

// idle svt_fork contexts of one root context; acquire() refreshes the fork's view of the root's tables
struct ForkPool {
    svt_ctx* root; std::mutex m; std::vector<svt_ctx*> idle;
    explicit ForkPool(svt_ctx* r) : root(r) {}
    svt_ctx* acquire() {
        svt_ctx* c = nullptr;
        { std::lock_guard<std::mutex> l(m); if (!idle.empty()) { c = idle.back(); idle.pop_back(); } }
        if (!c) { std::lock_guard<std::mutex> l(m); if (svt_fork(root, &c) != SVT_OK) throw Error{SVT_ERR_HIP, std::string("svt_fork: ") + svt_last_error(root)}; }
        svt_fork_refresh(c);
        return c;
    }
    void release(svt_ctx* c) { std::lock_guard<std::mutex> l(m); idle.push_back(c); }
};   // the reference's process::exit(1) sites surface as Error

// the ingest arrays: byte vectors whose resize() leaves the new elements as they are -- the parser writes every one of them, and std::vector's zero fill of 2 x 150 MB per
// 100k-read load was a serial memset (and, on a first load, the page faults of both arrays on one thread instead of on the parser's)
template <class T> struct default_init_alloc : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_alloc<U>; };
    default_init_alloc() = default;
    template <class U> default_init_alloc(const default_init_alloc<U>&) {}
    template <class U> void construct(U* p) { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
typedef std::vector<uint8_t, default_init_alloc<uint8_t>> RawBytes;
// the reads of one run, resident in HBM
struct ReadSet {
    svt_ctx* ctx = nullptr;
    svt_batch* batch = nullptr;
    uint32_t n = 0;
    std::vector<uint64_t> offsets;             // n+1
    std::vector<std::string> ids;              // full header text
    std::vector<uint8_t> rc_flags;             // last header token == "rc" (src/seq_parse.rs:362-366)
    std::vector<uint32_t> file_idx;
    RawBytes host_seq;                         // ASCII copy of the reads (Stage 4a POA input; the reference keeps dna_seq per TwinRead)
    std::shared_ptr<struct ForkPool> forks;    // contexts for the worker threads of Stage 3 (svt_fork), created on first use
    mutable std::vector<uint64_t> qualbin_off; // 4-bit quality bins (qual_seq), fetched from the GPU on first use by Stage 4
    mutable std::vector<uint8_t> qualbins;
    // Stage 1c fetch buffers, kept across calls: at 1 M reads they are ~250 MB, and allocating + zero-filling + unmapping them cost more
    // per step than the seed kernel itself (every element is overwritten by svt_seeds_fetch)
    struct SeedFetch {
        std::vector<uint64_t> mini_off, snp_off, lsh; std::vector<uint8_t> snp_flags, est_valid, lsh_valid, status;
        std::vector<double> est; std::vector<uint32_t> n_unique, n_solid;
    };
    mutable SeedFetch seed_fetch;
    // Stage 2 working lists, kept across calls for the same reason: one candidate list and one list of earlier block reads per read of a block (2 x 10^5 small
    // vectors per 100k-read step), and the two device-list buffers (16 / 8 pairs of words per read, overwritten by svt_lsh_candidates: no zero fill)
    struct Stage2Scratch {
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> l0, ext;
        std::unique_ptr<uint32_t[]> dout, xout; size_t dout_words = 0, xout_words = 0;
        double serial_seconds = 0.0;            // of the last cluster_reads_by_kmers: the parts every rank of a pooled run repeats (ordered fix-up, final grouping)
    };
    mutable Stage2Scratch stage2;
};

// Vec<TwinRead> of the reference (src/types.rs:386-412), as SoA over the reads that survive intake,
// in the FINAL order of src/main.rs:538.  Lists live on the GPU; the host keeps what the greedy logic needs.
struct TwinReads {
    uint32_t n = 0;
    std::vector<uint32_t> orig;                // index into the ReadSet / svt_batch
    std::vector<uint32_t> length, file_idx, n_mini, n_unique, n_snp_filtered;
    std::vector<double> est_id; std::vector<uint8_t> est_valid;
    std::vector<uint64_t> lsh; std::vector<uint8_t> lsh_valid;       // n*20, n
    uint32_t words = 0;                        // 64-bit words per SNPmer bitset row (rows stay in HBM)
    bool auto_low_polymorphism = false;        // src/main.rs:539-543
};

struct EmResult {                              // what refine_asv_depths_with_em writes into the consensuses
    std::vector<uint64_t> depth, unambig, ambig, leq10;
    uint64_t total_assigned = 0, filtered = 0;
    std::vector<uint32_t> read_n_best, read_first; std::vector<int32_t> read_nm;
    std::vector<std::vector<uint32_t>> read_class;
    bool kept_original = false;                // src/alignment.rs:1952-1955
    // read_to_asv_mappings.tsv (temp directory only; :1604-1608, :1874-1886): per twin read up to five (ASV, column 3, column 4) lines --
    // SNPmer path: the aligned ties in ascending nm (then ASV) with their SNPmer mismatches and nm; low-polymorphism path: the best ASVs with nm
    bool keep_mappings = false;
    struct MapLine { uint32_t asv; uint32_t a; int32_t b; };
    std::vector<std::vector<MapLine>> read_lines;
};

// ---- stage functions (reference names) ------------------------------------------------------------
// src/seq_parse.rs:12-78
KmerCountTable read_to_split_kmers(const ReadSet& rs, const ClusterArgs& args, uint64_t* n_distinct = nullptr);
// src/kmer_comp.rs:454-642 (host: statistics on the small filtered table)
struct SnpCandidates {                     // the two selections of the sorted count table that Stage 1b reads (table order)
    uint64_t n_table = 0;
    std::vector<uint64_t> g_kmer, h_kmer;  // g: entries in groups of >= 2 alleles; h: entries with rev + fwd > 100
    std::vector<uint32_t> g_rev, g_fwd, h_rev, h_fwd;
};
void count_split_kmers_device(const ReadSet& rs, const ClusterArgs& args, uint64_t* n_distinct, uint64_t* n_kept);
KmerCountTable fetch_count_table(const ReadSet& rs, uint64_t n_kept);
SnpCandidates candidates_from_device(const ReadSet& rs);
SnpCandidates candidates_from_table(const KmerCountTable& table, uint32_t k, const ClusterArgs& args);
KmerGlobalInfo snpmers_from_candidates(const SnpCandidates& cand, uint32_t k, const ClusterArgs& args);
KmerGlobalInfo get_snpmers_inplace_sort(const KmerCountTable& table, uint32_t k, const ClusterArgs& args);
// src/kmer_comp.rs:68-258 + src/main.rs:529-548
TwinReads twin_reads_from_snpmers(const ReadSet& rs, const KmerGlobalInfo& info, const ClusterArgs& args);
void twin_reads_from_snpmers(const ReadSet& rs, const KmerGlobalInfo& info, const ClusterArgs& args, TwinReads& tw);   // in place (storage reused)
// src/asv_cluster.rs:72-249
std::vector<std::vector<uint32_t>> cluster_reads_by_kmers(const ReadSet& rs, const TwinReads& tw, const ClusterArgs& args);
// src/asv_cluster.rs:561-795 (+ recluster :1272-1433); pre = clusters before reclustering
std::vector<std::vector<uint32_t>> cluster_reads_by_snpmers(const ReadSet& rs, const TwinReads& tw,
                                                            const std::vector<std::vector<uint32_t>>& kmer_clusters, const ClusterArgs& args,
                                                            std::vector<std::vector<uint32_t>>* pre = nullptr, std::vector<uint32_t>* pre_group = nullptr);
// src/alignment.rs:1723-2039; asvs = ASV sequences already uploaded + seeded
EmResult refine_asv_depths_with_em(const ReadSet& rs, const TwinReads& tw, svt_batch* asvs, const std::vector<uint64_t>& asv_offsets, const ClusterArgs& args, bool keep_mappings = false);
void refine_asv_depths_with_em(const ReadSet& rs, const TwinReads& tw, svt_batch* asvs, const std::vector<uint64_t>& asv_offsets, const ClusterArgs& args, bool keep_mappings, EmResult& em);   // the same into `em`, whose storage (a result of an earlier step) is reused
// its two halves (a pooled multi-rank run shards the first over read blocks and all-gathers the per-read classes, C2):
//   em_init          sizes the result;  em_read_classes  fills read_class / read_nm / read_n_best / read_first of the twin reads [lo, hi)
//   em_finish        counters, equivalence classes and EM from the per-read classes (src/alignment.rs:1898-2031)
void em_init(const TwinReads& tw, size_t n_asv, EmResult& em, bool keep_mappings = false);
void em_read_classes(const ReadSet& rs, const TwinReads& tw, svt_batch* asvs, const std::vector<uint64_t>& asv_offsets, const ClusterArgs& args, size_t lo, size_t hi, EmResult& em);
void em_finish(const TwinReads& tw, size_t n_asv, EmResult& em);
// src/alignment.rs:2044-2215; [n_asv][n_samples]
std::vector<std::vector<uint64_t>> compute_per_sample_depths(const TwinReads& tw, const EmResult& em, uint32_t n_samples, size_t n_asv);

// ---- Stage 4 (src/alignment.rs:233-1160) -----------------------------------------------------------
struct ConsensusSequence {                     // src/types.rs:162-190
    std::vector<uint8_t> sequence, decompressed;
    std::vector<uint8_t> hp_lengths;           // --use-hpc: run length per base of `sequence` (median over the pile-up, :586-656); empty = all 1
    size_t depth = 0, appended_depth = 0, id = 0;
    std::vector<uint32_t> cluster;             // twin read indices
    std::vector<size_t> low_quality_positions;
};
struct PileupEntry { uint8_t kind, base, qual, hp; };   // kind 0 Base (hp = its run length under --use-hpc, else 0), 1 Deletion, 2 Insertion (first base / quality)
struct PileupColumn { std::vector<PileupEntry> entries; };
typedef std::vector<std::vector<PileupColumn>> Pileups;
void ensure_qualbins(const ReadSet& rs);
std::vector<ConsensusSequence> align_and_consensus(const ReadSet& rs, const TwinReads& tw, const std::vector<std::vector<uint32_t>>& clusters, const ClusterArgs& args);
// the two halves of align_and_consensus: the POA of the clusters ci % world == rank (others empty), and the list assembly
std::vector<std::vector<uint8_t>> poa_raw_consensuses(const ReadSet& rs, const TwinReads& tw, const std::vector<std::vector<uint32_t>>& clusters, const ClusterArgs& args, uint32_t rank, uint32_t world);
std::vector<ConsensusSequence> assemble_consensuses(const std::vector<std::vector<uint32_t>>& clusters, std::vector<std::vector<uint8_t>> cons_all);
// Stage 4b-d fused (pile-ups in HBM, K10 column statistics): returns the low-quality consensuses, `consensuses` keeps the rest
std::vector<ConsensusSequence> polish_consensuses(const ReadSet& rs, const TwinReads& tw, std::vector<ConsensusSequence>& consensuses, const ClusterArgs& args,
                                                  std::map<uint8_t, double>* qmap_out = nullptr, Pileups* keep = nullptr);
void decompress(ConsensusSequence& c);
// ---- Stage 5 / 6 (src/alignment.rs:1162-1517, src/chimera.rs) ---------------------------------------
std::vector<uint64_t> minimizer_seeds(const uint8_t* s, size_t len, size_t w, size_t k);   // src/seeding.rs:99-186 (k-mer values)
std::vector<ConsensusSequence> merge_similar_consensuses(const ReadSet& rs, std::vector<ConsensusSequence> consensuses,
                                                         const std::vector<ConsensusSequence>& low_qual, const ClusterArgs& args);
std::vector<ConsensusSequence> detect_and_filter_chimeras(const ReadSet& rs, std::vector<ConsensusSequence> consensuses, const ClusterArgs& args,
                                                          std::vector<uint32_t>* chimera_idx = nullptr);
// generate_consensus_poa (src/alignment.rs:193-231): sequences + per-base weights (quality bytes) -> consensus
std::vector<uint8_t> poa_consensus(const std::vector<std::vector<uint8_t>>& seqs, const std::vector<std::vector<uint8_t>>& quals, uint64_t* graph_nodes = nullptr, bool wide_cells = false, bool no_band = false);
// the same for many clusters; engine (Tuning::poa_engine): 0 the host DP on the worker pool, 1 the DP of every round in one K11 launch,
// 2 everything in one K12 launch with the graphs resident on the device; graph_nodes (optional): nodes of every cluster's final graph
struct PoaInput { std::vector<std::vector<uint8_t>> seqs, quals; };
std::vector<std::vector<uint8_t>> poa_consensus_batch(svt_ctx* ctx, const std::vector<PoaInput>& in, int engine = 0, bool wide_cells = false, std::vector<uint64_t>* graph_nodes = nullptr, bool no_band = false);

// ---- formats either side of the path (src/main.rs:140-200, writers; needletail ingest) ---------------------------------
struct FinalAsv {
    std::vector<uint8_t> sequence; size_t depth = 0, debug_id = 0; long long chimera_score = 0;
    uint64_t unambig = 0, ambig = 0, leq10 = 0; std::vector<uint64_t> per_sample; std::vector<uint32_t> cluster;
};
void read_fastx_files(const std::vector<std::string>& files, RawBytes& seq, RawBytes& qual, std::vector<uint64_t>& off, std::vector<std::string>& ids,
                      std::vector<uint32_t>& file_idx, bool& any_qual);   // io.cpp: the files of a run, several side by side on the pool
void set_gz_threads(int n);    // io.cpp: threads one gzip member is inflated on (0 = by the situation)
unsigned gz_threads_now();
void set_gz_inflate(int on);   // io.cpp: 1 = gz inputs through host/inflate.hpp (default), 0 = zlib
size_t read_fastx_file(const std::string& path, RawBytes& seq, RawBytes& qual, std::vector<uint64_t>& offsets,
                       std::vector<std::string>& ids, bool& any_qual, bool keep_buffer = true);   // keep_buffer: the thread keeps the buffer a gz file was inflated into (warm pages for the next load); false on pool threads
std::vector<FinalAsv> finalize_asvs(const std::vector<ConsensusSequence>& consensuses, const EmResult& em, const std::vector<std::vector<uint64_t>>* per_sample);
void write_consensus_fasta(const std::vector<FinalAsv>& asvs, const std::string& path, const std::string& prefix);
void write_feature_table(const std::vector<FinalAsv>& asvs, const std::string& path, const std::vector<std::string>& sample_names);
void write_clusters_tsv(const std::vector<FinalAsv>& asvs, const ReadSet& rs, const TwinReads& tw, const std::string& path, const std::string& prefix, bool by_index = true);
// the reference's `<out>/temp/` files (stage-level parity probes)
std::vector<FinalAsv> as_records(const std::vector<ConsensusSequence>& cons);
void write_read_to_asv_mappings(const EmResult& em, const std::vector<size_t>& consensus_ids, const ReadSet& rs, const TwinReads& tw, bool low_polymorphism, const std::string& path);
void write_kmer_clusters_tsv(const std::vector<std::vector<uint32_t>>& clusters, const std::string& path);
void write_pre_recluster_tsv(const std::vector<std::vector<uint32_t>>& pre, const std::vector<uint32_t>& group, const std::string& path);
void write_snpmer_clusters_tsv(const std::vector<std::vector<uint32_t>>& clusters, const ReadSet& rs, const TwinReads& tw, const std::string& path);

bool trace_enabled();
void trace_add(const char* name, double seconds, double cpu_seconds = 0.0);   // main thread only
double trace_cpu_now();                              // process CPU seconds (all threads): attributable to a phase when ONE sample is in flight
void trace_dump();   // SAVONT_TRACE=1: print accumulated host timings to stderr

}  // namespace savont
