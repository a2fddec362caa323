/* savont_asv.h -- C ABI of libsavont_asv.so: the STAGE-level boundary of the MI355X-native `savont asv` hot path.
 *
 * The reference has no FFI for this path; its stages are plain Rust functions called in sequence from run_cluster
 * (src/main.rs:64-142) on borrowed slices (edges B1-B5).  This header declares one group of entry
 * points per call edge, each citing the reference function it stands for; a savont maintainer who wants whole stages rather
 * than kernels binds these (INTEGRATION.md shows the Rust block generated from this file), one who wants kernels binds
 * savont_hip.h.  libsavont_asv.so is C++ above the kernel ABI (there is no Rust toolchain in the build image).
 *
 * Conventions: an opaque svh_pipeline* per host thread owns one svt_ctx (its streams, scratch, reads in HBM).  Functions
 * returning int give 0 or a negative svt_* error code (svh_last_error has the text; the reference's process::exit(1) sites
 * surface this way, nothing aborts).  Results are fetched into caller-allocated arrays after a size query.  No CPU fallback:
 * svh_create fails with SVT_ERR_NODEVICE without a gfx950 device.
 */
#ifndef SAVONT_ASV_H
#define SAVONT_ASV_H
#include <stdint.h>
#include "savont_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* the libraries are built with -fvisibility=hidden: exactly the prototypes between this push and the pop below are exported */
#pragma GCC visibility push(default)

typedef struct svh_pipeline svh_pipeline;
/* ClusterArgs of the reference (src/cli.rs:40-190), the fields this path reads; svh_default_args fills the reference's defaults */
typedef struct svh_args {
    uint32_t kmer_size, c, min_read_length, max_read_length;
    double quality_value_cutoff;
    uint32_t minimum_base_quality, single_strand, min_cluster_size, max_iterations_recluster;
    double primary_clustering_threshold;
    uint32_t low_polymorphism, align_band;
    uint32_t n_depth_cutoff, mask_low_quality;
    double posterior_threshold_ln;
    uint32_t chimera_allowable_errors, chimera_detect_length, skip_chimera_detection, use_hpc;
    uint32_t no_snpmers, no_band;   /* the reference's two hidden flags on this path (src/cli.rs:145,183): SNPmer calling returns no sites (src/kmer_comp.rs:525,689); the POA of stage 4a runs unbanded, on the host engine (src/alignment.rs:198,217) */
} svh_args;


/* ---- pipeline object, options, diagnostics (no reference counterpart: `ClusterArgs` of src/cli.rs:40-190 arrives as svh_args) ---- */
void svh_default_args(svh_args* a);
int svh_create(int device_id, const svh_args* a, svh_pipeline** out);
/* Implementation choices are pipeline state set through svh_set_option, never process environment.  With one exception they give identical
 * results (tests run the alternatives against the same oracle):
 *   stage2_first_block, stage2_max_block, stage2_pair_cap, stage3_first_block, stage3_block, stage3_max_block, stage3_switch   block schedules
 *   stage3_waves      1 (default) one multi-cluster K6 call per wave of the greedy loops | 0 one call per block per cluster on forked contexts
 *   poa_engine        -1 (default) by the CPU share of the process: K12 when its worker pool has <= 10 threads, else the host DP |
 *                     0 host DP | 2 K12 (graphs resident on the device) | 3 K12 for poa_device_share percent of the clusters
 *   poa_device_share  0..100 (engine 3), poa_cells 16 | 32 (host DP cell width)
 *   nm_contract       the ONE option that changes results, by design: what `nm` of src/alignment.rs:1848-1862 means --
 *                     1 (default) minimap2-style affine local nm inside the band around the unit-cost optimum | 2 the same in the whole band |
 *                     0 banded unit-cost overlap distance
 *   gz_inflate        1 (default) gz inputs through host/inflate.hpp | 0 zlib's gzread (comparison runs); process-wide
 *   gz_threads        threads ONE gzip member is inflated on (round 6: later pieces start at block boundaries found by trial decode and carry markers for the 32 KB in
 *                     front of them, replaced in stream order): 0 (default) up to eight pool threads when no other file is being inflated in this process, else one |
 *                     n >= 1 exactly n; process-wide.  Byte-identical output either way (zlib is the test oracle); what the parallel path refuses, the sequential one reads
 *   every other key goes to svt_set_option of the device layer (include/savont_hip.h). */
int svh_set_option(svh_pipeline* p, const char* key, int64_t value);
int svh_set_temp_dir(svh_pipeline* p, const char* dir);
void svh_trace_dump(void);
void svh_destroy(svh_pipeline* p);
const char* svh_last_error(svh_pipeline* p);
svt_ctx* svh_ctx(svh_pipeline* p);
double svh_stage_seconds(svh_pipeline* p, const char* name);

/* ---- reads in, files out: needletail ingest of src/seq_parse.rs:316-373 / src/kmer_comp.rs:112-127 (one decode, reads stay in HBM);
 * write_consensus_fasta src/alignment.rs:830-860, write_feature_table src/main.rs:381-400, write_clusters_tsv src/alignment.rs:799-826 ---- */
int svh_set_reads(svh_pipeline* p, const uint8_t* seq, const uint8_t* qual, const uint64_t* offsets, uint32_t n, const char* ids_joined, const uint32_t* file_idx);
int svh_load_fastx(svh_pipeline* p, const char* paths_joined, uint32_t* n_reads);
int svh_write_outputs(svh_pipeline* p, const char* out_dir, const char* sample_names_joined, int pooled);
int svh_repack(svh_pipeline* p);
int svh_fastx_digest(const char* path, uint64_t* n_records, uint64_t* n_bases, int* has_qual, uint64_t* digest, char* err, uint64_t err_cap);
/* stateless gz check (no GPU): the .gz file inflated whole by zlib (decoder 0) or by the library's own decoder (1: host/inflate.hpp, what svh_load_fastx uses for the
 * reference's usual input format; n >= 2: the same with ONE member inflated on n threads, src/seq_parse.rs:356-379 through needletail / flate2) -> inflated bytes, FNV-1a of them, seconds of the inflate alone (CRC check included) */
int svh_gunzip_digest(const char* path, int decoder, uint64_t* n_bytes, uint64_t* digest, double* seconds, char* err, uint64_t err_cap);

/* ---- edge B1, src/main.rs:501: seq_parse::read_to_split_kmers (src/seq_parse.rs:12-78) -> the kept (k-mer, [rev, fwd]) table ---- */
int svh_read_to_split_kmers(svh_pipeline* p);
uint64_t svh_count_distinct(svh_pipeline* p);
uint64_t svh_count_size(svh_pipeline* p);
int svh_count_fetch(svh_pipeline* p, uint64_t* km, uint32_t* rev, uint32_t* fwd);
int svh_set_count_table(svh_pipeline* p, const uint64_t* km, const uint32_t* rev, const uint32_t* fwd, uint64_t n);

/* ---- edge B2, src/main.rs:520: kmer_comp::get_snpmers_inplace_sort (src/kmer_comp.rs:454-642) -> KmerGlobalInfo (SNPmer sites, high-frequency k-mers) ---- */
int svh_get_snpmers(svh_pipeline* p);
uint32_t svh_snpmer_count(svh_pipeline* p);
void svh_snpmer_fetch(svh_pipeline* p, uint64_t* split, uint8_t* m0, uint8_t* m1, uint32_t* c0, uint32_t* c1);
uint32_t svh_high_freq_thresh(svh_pipeline* p);
uint32_t svh_high_freq_count(svh_pipeline* p);
void svh_high_freq_fetch(svh_pipeline* p, uint64_t* k);
int svh_set_snpmers(svh_pipeline* p, const uint64_t* split, const uint8_t* m0, const uint8_t* m1, uint32_t n, const uint64_t* hf, uint32_t n_hf);
int svh_snpmers_from_table(const uint64_t* km, const uint32_t* rev, const uint32_t* fwd, uint64_t n, uint32_t k, int single_strand, uint64_t* split, uint8_t* m0, uint8_t* m1, uint32_t* c0, uint32_t* c1, uint64_t* hf, uint32_t* n_hf, uint32_t* thresh);
double svh_binomial_test(uint64_t n, uint64_t k, double p);
double svh_fisher_two_tail(uint32_t a, uint32_t b, uint32_t c, uint32_t d);

/* ---- edge B3, src/main.rs:537-543: kmer_comp::twin_reads_from_snpmers (src/kmer_comp.rs:68-258) -> the ordered TwinRead list; auto low-polymorphism switch ---- */
int svh_twin_reads(svh_pipeline* p);
uint32_t svh_twin_count(svh_pipeline* p);
int svh_auto_low_polymorphism(svh_pipeline* p);
void svh_twin_meta(svh_pipeline* p, uint32_t* orig, uint32_t* length, double* est, uint8_t* ev, uint32_t* n_mini, uint32_t* n_unique, uint32_t* n_snp_filt, uint64_t* lsh, uint8_t* lsh_valid);

/* ---- edge B4, src/main.rs:83,87: cluster_reads_by_kmers (src/asv_cluster.rs:72-249), cluster_reads_by_snpmers (:561-795 incl. reclustering);
 * which = 0 k-mer clusters, 1 final SNPmer clusters, 2 SNPmer clusters before reclustering ---- */
int svh_cluster_reads_by_kmers(svh_pipeline* p);
int svh_cluster_reads_by_snpmers(svh_pipeline* p);
uint32_t svh_cluster_count(svh_pipeline* p, int which);
uint64_t svh_cluster_total(svh_pipeline* p, int which);
void svh_clusters_fetch(svh_pipeline* p, int which, uint64_t* off, uint32_t* mem, uint32_t* group);

/* ---- stages 4-6, src/main.rs:92-131: alignment::align_and_consensus + generate_consensus_pileups + analyze_pileup_consensuses (src/alignment.rs:233-1160),
 * merge_similar_consensuses (:1162-1517), chimera::filter_chimeras (src/chimera.rs:37-494); set = 0 kept consensuses, 1 low-quality ones ---- */
int svh_consensus(svh_pipeline* p, int which);
int svh_merge_similar_consensuses(svh_pipeline* p);
int svh_detect_chimeras(svh_pipeline* p);
uint32_t svh_chimera_count(svh_pipeline* p);
void svh_chimera_fetch(svh_pipeline* p, uint32_t* ids);
uint32_t svh_consensus_count(svh_pipeline* p, int set);
uint64_t svh_consensus_bases(svh_pipeline* p, int set);
void svh_consensus_fetch(svh_pipeline* p, int set, uint8_t* seq, uint64_t* off, uint64_t* depth, uint64_t* id, uint32_t* n_lowq);
uint32_t svh_quality_map(svh_pipeline* p, uint8_t* q, double* rate);
uint64_t svh_minimizer_seeds(const uint8_t* seq, uint64_t len, uint32_t w, uint32_t k, uint64_t* out, uint64_t cap);
void svh_keep_pileups(svh_pipeline* p, int keep);
uint64_t svh_pileup_entries(svh_pipeline* p, uint32_t ci);
void svh_pileup_fetch(svh_pipeline* p, uint32_t ci, uint64_t* col_off, uint8_t* kind, uint8_t* base, uint8_t* qual);
void svh_pileup_fetch_hp(svh_pipeline* p, uint32_t ci, uint8_t* hp);
uint32_t svh_raw_consensus_count(svh_pipeline* p);
uint64_t svh_raw_consensus_len(svh_pipeline* p, uint32_t ci);
void svh_raw_consensus_fetch(svh_pipeline* p, uint32_t ci, uint8_t* seq, uint64_t* depth, uint64_t* id, uint64_t* n_members);
int svh_poa_consensus(const uint8_t* seq, const uint8_t* weights, const uint64_t* off, uint32_t n, uint8_t* out, uint64_t cap, uint64_t* graph_nodes, int wide_cells);
int svh_poa_consensus_batch(svh_pipeline* p, int engine, const uint8_t* seq, const uint8_t* weights, const uint64_t* off, const uint64_t* cl_off, uint32_t n_clusters, uint8_t* out, uint64_t* out_off, uint64_t cap, uint64_t* graph_nodes);

/* ---- edge B5, src/main.rs:142: alignment::refine_asv_depths_with_em (src/alignment.rs:1723-2039), compute_per_sample_depths (:2044-2215) ---- */
int svh_consensus_to_asvs(svh_pipeline* p);
int svh_set_asvs(svh_pipeline* p, const uint8_t* seq, const uint64_t* offsets, uint32_t n);
int svh_refine_asv_depths_with_em(svh_pipeline* p);
void svh_em_fetch(svh_pipeline* p, uint64_t* depth, uint64_t* un, uint64_t* am, uint64_t* l10, uint64_t* total, uint64_t* filtered, int* kept_original);
void svh_em_read_assignments(svh_pipeline* p, uint32_t* nb, int32_t* nm, uint32_t* first);
int svh_compute_per_sample_depths(svh_pipeline* p, uint32_t n_samples, uint64_t* out);

/* ---- the sharded halves of stages 1a / 4a / 7 for a pooled multi-rank run (savont_amd/pooled.py): partial count tables on
 * device buffers, Stage-4a for the clusters ci % world == rank, read classes of a block of reads, EM on the gathered classes ---- */
int svh_count_partial_device(svh_pipeline* p, uint32_t lo, uint32_t hi, uint64_t* n_distinct);
int svh_count_export_device(svh_pipeline* p, uint64_t* d_kmer, uint32_t* d_rev, uint32_t* d_fwd, uint64_t cap, uint64_t* n);
int svh_count_merge_begin(svh_pipeline* p, uint64_t total_entries);
int svh_count_merge_device(svh_pipeline* p, const uint64_t* d_kmer, const uint32_t* d_rev, const uint32_t* d_fwd, uint64_t n);
int svh_count_finalize(svh_pipeline* p);
int svh_consensus_poa(svh_pipeline* p, int which, uint32_t rank, uint32_t world);
uint32_t svh_consensus_raw_count(svh_pipeline* p);
uint64_t svh_consensus_raw_bytes(svh_pipeline* p);
void svh_consensus_raw_export(svh_pipeline* p, uint32_t* len, uint8_t* bytes);
int svh_consensus_raw_import(svh_pipeline* p, const uint32_t* len, const uint8_t* bytes, uint32_t n, uint64_t n_bytes);
int svh_consensus_polish(svh_pipeline* p);
int svh_em_begin(svh_pipeline* p);
int svh_em_classes(svh_pipeline* p, uint32_t lo, uint32_t hi);
uint64_t svh_em_classes_members(svh_pipeline* p, uint32_t lo, uint32_t hi);
void svh_em_classes_export(svh_pipeline* p, uint32_t lo, uint32_t hi, uint32_t* n_best, int32_t* nm, uint32_t* members);
int svh_em_classes_import(svh_pipeline* p, uint32_t lo, uint32_t hi, const uint32_t* n_best, const int32_t* nm, const uint32_t* members, uint64_t n_members);
int svh_em_finish(svh_pipeline* p);

/* ---- `savont asv` in ONE call, on one GPU or over the ranks of an RCCL communicator (src/main.rs:49-152 run_cluster) ----
 * svh_set_shard_comm: every rank's pipeline joins a communicator made from the SVT_COMM_ID_BYTES id bytes of svt_shard_comm_id (savont_hip.h:
 * one rank makes them, the caller hands them round).  svh_run_asv then runs stages 1-7 on the resident reads; with world > 1 it deals the work
 * out -- counting (src/seq_parse.rs:316-497) and Stage 7 (src/alignment.rs:1786, a par_iter over all reads) by read block, Stage 3
 * (src/asv_cluster.rs:593) by k-mer cluster, POA and polish (src/alignment.rs:241,426) by cluster, the K5 tiles of Stage 2 by slice under the
 * replicated greedy loop -- and the LIBRARY issues every exchange (grouped RCCL collectives on device memory; merge points
 * src/seq_parse.rs:434-487, src/alignment.rs:1918-1920).  Results are identical to the one-rank run.  After every stage the ranks agree on its outcome
 * (one word per rank through the same exchange path): an error every rank shares is returned by every rank with the communicator intact; an error only some
 * ranks have reaches the others as SVT_ERR_EXCHANGE naming the rank; an exchange that fails, or makes no progress for "shard_timeout_s" seconds, aborts the
 * communicator (savont_hip.h) and returns SVT_ERR_EXCHANGE -- no rank waits for ever and the host process is never ended by the library.  svh_count_shard_merge / svh_snpmers_check_ranks / svh_consensus_gather / svh_em_classes_gather are the single
 * exchanges svh_run_asv is made of (each ONE call on every rank; read blocks are [n r / W, n (r + 1) / W)). */
int svh_set_shard_comm(svh_pipeline* p, uint32_t rank, uint32_t world, const uint8_t* comm_id);
int svh_run_asv(svh_pipeline* p);
int svh_count_shard_merge(svh_pipeline* p);
int svh_snpmers_check_ranks(svh_pipeline* p);
int svh_consensus_gather(svh_pipeline* p);
int svh_em_classes_gather(svh_pipeline* p);

/* ---- deterministic synthetic amplicon reads of bench.py and the larger tests (no reference counterpart) ---- */
uint64_t svh_synth_reads(const uint8_t* hap_seq, const uint64_t* hap_off, uint32_t n_hap, const double* weights, uint32_t n_reads, uint64_t seed, uint8_t* seq_out, uint8_t* qual_out, uint64_t* off_out, uint32_t* hap_of_read, uint8_t* strand_of_read);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
