/*
 * savont_oracle.h -- C API of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is a plain, single-threaded-by-default C++17 restatement of the
 * `savont asv` hot path of bluenote-1577/savont v0.6.4 (reference files cited per
 * function in savont_oracle.cpp).  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library.  The product
 * (savont_amd/, include/savont_hip.h) never links, imports or executes it.
 *
 * PARITY PINNING: see the header of savont_oracle.cpp.
 */
#ifndef SAVONT_ORACLE_H
#define SAVONT_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

typedef struct orc_params {
    uint32_t k;                      /* cli.rs:153  default 17 */
    uint32_t c;                      /* cli.rs:83   default 11 */
    uint32_t min_read_length;        /* cli.rs:87   default 1100 */
    uint32_t max_read_length;        /* cli.rs:91   default 2000 */
    double   quality_value_cutoff;   /* cli.rs:95   default 98.0 */
    uint32_t minimum_base_quality;   /* cli.rs:99   default 25 */
    uint32_t single_strand;          /* cli.rs:103  default 0 */
    uint32_t min_cluster_size;       /* cli.rs:107  default 12 */
    uint32_t max_iterations_recluster; /* cli.rs:132 default 10 */
    double   primary_clustering_threshold; /* cli.rs:185 default 0.95 */
    uint32_t align_band;             /* K8 band half-width; 0 = auto (see orc_band_for) */
    uint32_t threads;                /* CPU threads for the embarrassingly parallel loops */
    uint32_t low_polymorphism;       /* cli.rs:143  default 0 (also forced by the caller when orc_auto_low_polymorphism, main.rs:76-79) */
    uint32_t no_snpmers;             /* cli.rs:145 (hidden) default 0: SNPmer calling returns an empty list (kmer_comp.rs:525,689) */
    uint32_t no_band;                /* cli.rs:183 (hidden) default 0: the POA of stage 4a runs unbanded (alignment.rs:198,217) */
    uint32_t nm_contract;            /* Stage-7 nm: 0 = K8 unit-cost overlap distance (the product's contract), 1 = K8a minimap2-style affine local nm (study only) */
} orc_params;

void orc_default_params(orc_params* p);

orc_ctx* orc_create(const orc_params* p);
void     orc_destroy(orc_ctx*);
const char* orc_last_error(orc_ctx*);

/* ---- leaf functions (stateless; used for kernel-level parity tests) ---- */
uint64_t orc_mm_hash64(uint64_t key);                               /* seeding.rs:18-28 */
uint64_t orc_fx_hash_pair(uint64_t seed, uint64_t kmer);            /* types.rs:733-736 */
uint8_t  orc_byte_to_seq(uint8_t b);                                /* types.rs:92-101 */
uint8_t  orc_qual_bin(uint8_t ascii);                               /* types.rs:447-467 */
/* packs ASCII to the 2-bit layout used by the product (16 bases / u32, base i at bits 30-2*(i%16),
 * i.e. first base in the high bits like the reference's k-mer integers, types.rs:1119-1128) */
void     orc_pack_2bit(const uint8_t* seq, uint64_t len, uint32_t* words);
uint64_t orc_kmer_from_ascii(const uint8_t* s, uint32_t k);         /* first base in high bits */
uint64_t orc_revcomp_kmer(uint64_t kmer, uint32_t k);
void     orc_reverse_complement(const uint8_t* seq, uint64_t len, uint8_t* out); /* utils.rs:51-65 */
/* types.rs:622-663 TwinRead::kmer_from_position: canonical (split value, ties -> forward) k-mer at `pos` of the stored sequence; ~0 if out of range */
uint64_t orc_kmer_from_position(const uint8_t* seq, uint64_t len, uint32_t pos, uint32_t k);
/* seeding.rs:975-1068; returns number of emitted u64 (out may be NULL to count) */
uint64_t orc_split_kmer_mid(const uint8_t* seq, const uint8_t* qual, uint64_t len,
                            uint32_t k, uint8_t min_bq, uint64_t* out);
double   orc_estimate_identity(const uint8_t* qual, uint64_t len, int* valid); /* seeding.rs:801-817, :372-380 */
double   orc_binomial_test(uint64_t n, uint64_t k, double p);       /* utils.rs:37-49 */
double   orc_fisher_two_tail(uint32_t a, uint32_t b, uint32_t c, uint32_t d); /* kmer_comp.rs:579 */
/* LSH signatures types.rs:719-747; sig[20], valid[20] */
void     orc_lsh_signatures(const uint64_t* kmers, uint32_t n, uint64_t* sig, uint8_t* valid);
/* K8 contract: banded overlap edit distance (see savont_oracle.cpp) */
int32_t  orc_band_for(uint32_t qlen, uint32_t tlen);
int32_t  orc_align_nm(const uint8_t* q, uint32_t qlen, const uint8_t* t, uint32_t tlen,
                      int reverse_target, uint32_t band);
/* K8a: minimap2-style nm (best LOCAL alignment, a=2 b=4 gap=min(4+2l,24+l), fewest nm among the top-scoring alignments;
 * see savont_oracle.cpp).  Returns nm (-1: no positive-score alignment in the band); out[5] = nm, score, q_end, t_end, #cells at max */
int32_t  orc_align_nm_affine(const uint8_t* q, uint32_t qlen, const uint8_t* t, uint32_t tlen,
                             int reverse_target, uint32_t band, int32_t* out);
/* K9 contract (pile-up rows, src/alignment.rs:449-575): same DP as orc_align_nm with a deterministic traceback.
 * q = consensus (reference side), t = read (ASCII) with its 4-bit quality bins (may be NULL -> quality 33).
 * Predecessor priority: diagonal, then up (deletion in the read), then left (insertion in the read); end cell =
 * smallest value on the last row / last column, ties -> smallest i+j, then smallest j-i; the walk stops at i==0 or j==0.
 * cells[qlen] (one per consensus position): bits 0-2 code (0-3 aligned read base, 4 deletion, 7 not covered),
 * bits 8-15 quality of the aligned base (bin*3+33 of the read position, src/alignment.rs:458-477),
 * bits 16-17 number of inserted read bases kept AFTER this position (<= MAX_INSERTION_LENGTH = 2, src/constants.rs:3),
 * bits 18-25 full insertion length (capped 255), bits 32-33 / 34-35 the kept inserted bases, bits 40-47 / 48-55 their qualities.
 * span[4] = q_start, q_end, t_start, t_end (t in the orientation that was aligned).  Returns nm. */
int32_t  orc_align_pileup_row(const uint8_t* q, uint32_t qlen, const uint8_t* t, uint32_t tlen, const uint8_t* t_qualbins,
                              int reverse_target, uint32_t band, uint64_t* cells, uint32_t* span);
/* K7 contract on two plain sequences: distinct shared open-syncmer minimizers and how many carry the same canonical flag */
void     orc_strand_vote(const uint8_t* a, uint32_t alen, const uint8_t* b, uint32_t blen, uint32_t k, uint32_t c,
                         uint32_t* shared, uint32_t* same);
/* utils.rs:70-130 homopolymer helpers (doc-comment golden vectors) */
uint64_t orc_hpc(const uint8_t* seq, uint64_t len, uint8_t* out_seq, uint8_t* out_len);
uint64_t orc_hpc_qual(const uint8_t* seq, const uint8_t* qual, uint64_t len, uint8_t* out_seq, uint8_t* out_qual, uint8_t* out_len);   /* utils.rs:136-190 */
int32_t orc_align_pileup_row_tags(const uint8_t* q, uint32_t qlen, const uint8_t* t, uint32_t tlen, const uint8_t* qual, const uint8_t* hp, int reverse_target,
                                  uint32_t band, uint64_t* cells, uint32_t* span);   /* K9 on a homopolymer-compressed read: alignment.rs:480-538 */

/* ---- pipeline (stateful) ---- */
/* reads: concatenated ASCII, offsets[n+1]; qual may be NULL; ids: '\n'-joined full header
 * texts; file_idx may be NULL.  The call COPIES everything. */
int orc_set_reads(orc_ctx*, const uint8_t* seq, const uint8_t* qual, const uint64_t* offsets,
                  uint32_t n_reads, const char* ids_joined, const uint32_t* file_idx);

/* Stage 1a: seq_parse.rs:12-78,316-497 -> filtered table sorted by (masked k-mer, mid base) */
int      orc_count_split_kmers(orc_ctx*);
uint64_t orc_count_raw_distinct(orc_ctx*);
uint64_t orc_count_size(orc_ctx*);
void     orc_count_fetch(orc_ctx*, uint64_t* kmer, uint32_t* rev, uint32_t* fwd);

/* test hook (multi-rank driver tests): install a count table merged / filtered / sorted elsewhere */
int      orc_set_count_table(orc_ctx*, const uint64_t* kmer, const uint32_t* rev, const uint32_t* fwd, uint64_t n, uint64_t raw_distinct);

/* Stage 1b: kmer_comp.rs:454-642 */
int      orc_get_snpmers(orc_ctx*);
uint32_t orc_snpmer_count(orc_ctx*);
void     orc_snpmer_fetch(orc_ctx*, uint64_t* split_kmer, uint8_t* mid0, uint8_t* mid1,
                          uint32_t* cnt0, uint32_t* cnt1);
uint32_t orc_high_freq_thresh(orc_ctx*);
uint32_t orc_high_freq_count(orc_ctx*);
void     orc_high_freq_fetch(orc_ctx*, uint64_t* kmers); /* sorted ascending */
/* inject SNPmers / high-freq k-mers computed elsewhere (used to test later stages in isolation) */
int      orc_set_snpmers(orc_ctx*, const uint64_t* split_kmer, const uint8_t* mid0,
                         const uint8_t* mid1, uint32_t n, const uint64_t* high_freq, uint32_t n_hf);

/* Stage 1c: kmer_comp.rs:68-258 + main.rs:529-548.  Result: twin reads in FINAL order
 * (sorted by id, est_id cutoff, stable sort by est_id desc). */
int      orc_twin_reads(orc_ctx*);
uint32_t orc_twin_count(orc_ctx*);
int      orc_auto_low_polymorphism(orc_ctx*);
/* per twin read i: original read index, length, est_id, counts */
void     orc_twin_meta(orc_ctx*, uint32_t* orig_index, uint32_t* length, double* est_id,
                       uint8_t* est_valid, uint32_t* n_mini, uint32_t* n_mini_kept,
                       uint32_t* n_snp, uint32_t* n_snp_kept);
/* flattened lists in twin order (size = sum of the counts above) */
void     orc_twin_minimizers(orc_ctx*, uint32_t* pos, uint64_t* kmer, uint8_t* kept);
void     orc_twin_snpmers(orc_ctx*, uint32_t* pos, uint64_t* kmer, uint8_t* kept);
void     orc_twin_lsh(orc_ctx*, uint64_t* sig /*n*20*/, uint8_t* valid /*n*20*/);
/* 4-bit quality bins of ORIGINAL read `orig` (seeding.rs:578-602); returns count */
uint64_t orc_read_qual_bins(orc_ctx*, uint32_t orig, uint8_t* bins);
/* raw per-read seed extraction on ORIGINAL read `orig` (no filtering), for kernel parity:
 * returns counts through n_mini/n_snp; arrays may be NULL to size */
int      orc_read_seeds(orc_ctx*, uint32_t orig, uint32_t* n_mini, uint32_t* mini_pos,
                        uint64_t* mini_kmer, uint32_t* n_snp, uint32_t* snp_pos, uint64_t* snp_kmer);

/* Stage 2: asv_cluster.rs:72-249.  Clusters as CSR over twin indices. */
int      orc_cluster_by_kmers(orc_ctx*);
uint32_t orc_kmer_cluster_count(orc_ctx*);
uint64_t orc_kmer_cluster_total(orc_ctx*);
void     orc_kmer_clusters_fetch(orc_ctx*, uint64_t* offsets /*nc+1*/, uint32_t* members);

/* Stage 3: asv_cluster.rs:561-795 + :1272-1433 */
int      orc_cluster_by_snpmers(orc_ctx*);
uint32_t orc_snpmer_cluster_count(orc_ctx*);
uint64_t orc_snpmer_cluster_total(orc_ctx*);
void     orc_snpmer_clusters_fetch(orc_ctx*, uint64_t* offsets, uint32_t* members);
/* pre-recluster clusters (temp/snpmer_clusters_before_reclust2.5.tsv) */
uint32_t orc_snpmer_pre_cluster_count(orc_ctx*);
uint64_t orc_snpmer_pre_cluster_total(orc_ctx*);
void     orc_snpmer_pre_clusters_fetch(orc_ctx*, uint64_t* offsets, uint32_t* members, uint32_t* group);

/* Stage 7: alignment.rs:1723-2039 (SNPmer path).  ASVs: concatenated ASCII + offsets. */
int      orc_set_asvs(orc_ctx*, const uint8_t* seq, const uint64_t* offsets, uint32_t n_asvs);
int      orc_refine_depths_em(orc_ctx*);
/* per ASV (input order): depth after EM (0 = removed), unambig, ambig, leq10 counters */
void     orc_em_fetch(orc_ctx*, uint64_t* depth, uint64_t* unambig, uint64_t* ambig, uint64_t* leq10);
uint64_t orc_em_total_assigned(orc_ctx*);
uint64_t orc_em_filtered(orc_ctx*);
/* per twin read: number of tied best ASVs (0 = filtered), best nm (-1 if none), first best ASV */
void     orc_em_read_assignments(orc_ctx*, uint32_t* n_best, int32_t* best_nm, uint32_t* first_asv);
/* per twin read: class members (tied best ASVs, ascending) as CSR; returns the member count; arrays may be NULL to size */
uint64_t orc_em_read_classes(orc_ctx*, uint64_t* off, uint32_t* members);
/* Stage 7b alignment.rs:2044-2215: depth matrix [n_asv][n_samples] row-major */
int      orc_per_sample_depths(orc_ctx*, uint32_t n_samples, uint64_t* out);

/* timing of the last stage call in seconds (steady clock) */
double   orc_last_stage_seconds(orc_ctx*);

#ifdef __cplusplus
}
#endif
#endif
