/*
 * savont_hip.h -- C-ABI of libsavont_hip.so: the MI355X (gfx950) device layer of the
 * `savont asv` read-clustering hot path.  This is the drop-in boundary: plain pointers and
 * sizes, no C++/torch types.  A Rust caller binds it with `extern "C"` (INTEGRATION.md shows
 * the stub); the C++ host pipeline (savont_amd/csrc/host/) and the Python harness bind the same
 * symbols.
 *
 * The reference (bluenote-1577/savont v0.6.4) has no FFI for this path: stages are Rust
 * functions called in sequence from `run_cluster` (src/main.rs:64-142).  Each entry point below
 * names the reference interface whose DATA-PARALLEL part it replaces (file:line relative to the
 * reference root).  Order-dependent greedy logic (asv_cluster.rs) stays on the host and consumes
 * the lists these calls return.
 *
 * Conventions
 *  - every function returns 0 (SVT_OK) or a negative error; svt_last_error(ctx) has the text.
 *    Nothing here calls exit()/abort() (the reference's process::exit conventions stay with
 *    the caller: src/seq_parse.rs:69-72, src/kmer_comp.rs:469-472).
 *  - all host buffers are caller-owned; sizes are queried first (svt_*_sizes) then fetched.
 *  - one svt_ctx per host thread / per GPU (one process per GPU); calls are synchronous.
 *  - k-mers are 2k-bit integers, first base in the HIGH bits, A=0 C=1 G=2 T=3, every other byte
 *    = 0 (src/types.rs:92-101, :1119-1138).
 *  - the library fails loudly (SVT_ERR_NODEVICE) when no gfx950 device is present; there is no
 *    CPU fallback.
 */
#ifndef SAVONT_HIP_H
#define SAVONT_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the libraries are built with -fvisibility=hidden: exactly the prototypes between this push and the pop below are exported */
#pragma GCC visibility push(default)

#define SVT_OK 0
#define SVT_ERR_ARG (-1)
#define SVT_ERR_HIP (-2)
#define SVT_ERR_STATE (-3)
#define SVT_ERR_OVERFLOW (-4)
#define SVT_ERR_NODEVICE (-5)
#define SVT_ERR_TOOWIDE (-6)    /* the SNPmer rows do not fit the LDS tile of the segmented K6 call: use svt_snpmer_compat_lists per group (nothing else returns it) */
#define SVT_ERR_EXCHANGE (-7)   /* a collective of a sharded call failed, timed out ("shard_timeout_s") or was aborted (exchange hook / RCCL): the communicator of the context is gone, the step cannot be retried on it */

#define SVT_LSH_TABLES 20u      /* src/constants.rs:67 */
#define SVT_LSH_BUCKET 3u       /* src/constants.rs:68 */

typedef struct svt_ctx svt_ctx;
typedef struct svt_batch svt_batch;     /* a set of sequences resident in HBM (reads, or ASVs) */
typedef struct svt_bitset svt_bitset;   /* a set of SNPmer bitset rows resident in HBM (cluster consensuses) */

/* ---- context ----------------------------------------------------------------------------
 * The first svt_device_count / svt_create of a process sets GPU_MAX_HW_QUEUES=16 in its environment when the variable is unset (the samples in flight are
 * separate HIP streams; the runtime's default of 4 hardware queues serialises them behind the longest kernel).  A host with threads of its own should export
 * the variable before it starts them: the library then writes nothing (setenv is not safe against a concurrent getenv). */
int         svt_version(void);
int         svt_device_count(void);
int         svt_create(int device_id, svt_ctx** out);
void        svt_destroy(svt_ctx* ctx);
const char* svt_last_error(const svt_ctx* ctx);
/* Kernel / copy-path selection is context state, never process environment.  Options live on the root context (a fork reads its
 * parent's).  The reference has no counterpart: these choose between implementations with IDENTICAL results (tests run both).
 *   "k8_kernel"        0 bit-parallel lane-per-pair (default) | 1 anti-diagonal wavefront
 *   "k9_kernel"        0 by launch size (default) | 1 anti-diagonal wavefront | 2 bit-parallel, 64-bit direction window per column
 *                      (walks that leave it run again, svt_get_option "k9_pairs" / "k9_again_pairs" / "k9_redo_pairs" count them) | 3 bit-parallel, full slab
 *   "shard_seeds"      under svt_set_shard: 1 = svt_extract_seeds runs this rank's read block only and gathers the seed arrays (default 0: every
 *                      rank extracts all seeds -- ~4.7 KB per read would cross the links for ~45 ns of kernel time per read)
 *   "count_kernel"     0 windowed LDS counting, a lane per read, one 16-wave workgroup per CU (default) | 3 the same with 8-wave workgroups of 76 KB | 1 wave per read into the HBM table | 2 windowed, a wave per read (the kernel of rounds 2-5)
 *   "consensus_dense"  0 sparse-row consensus kernel (default) | 1 dense rows
 *   "consensus_chunk"  members per block of the sparse consensus kernel (0 = 256)
 *   "pin_staging"      1 = small calls staged through pinned memory (default 0)
 *   "zero_copy"        0 = no zero-copy I/O for small calls (default 1)
 *   "sync_block"       1 = waits that leave the core to other threads instead of spinning in hipStreamSynchronize: a short burst of hipStreamQuery, then
 *                      queries between sleeps of 40 us .. 1 ms; the K12 launch (svt_poa_graphs_wait) is awaited through a word in page-locked host memory
 *                      that a one-lane kernel sets, without runtime calls (default 0; for callers that keep several contexts busy from one process)
 *   "k8a_pk16"         K8a (svt_align_nm_affine*, queue launch): 1 (default) = pairs with bands <= 39 and |n - m| <= 64 run through the packed 16-bit cell (two pairs per
 *                      lane group in the halves of every register; a per-pair certificate, the pairs without it rerun through the 32-bit cell: identical results) | 0 = 32-bit cell only |
 *                      2 = packed only for bands <= 39 | 3 = bands <= 39 chosen by band and length difference alone (the two comparison settings of DESIGN.md section 3)
 *   "seeds_hash"       K3 (svt_extract_seeds): 0 (default) = the rank-table kernel when s = k - c + 1 <= 7 (rank of mm_hash64 of the canonical s-mer from a table in LDS,
 *                      persistent sixteen-wave workgroups); 1 = the kernel that evaluates mm_hash64 per base in 64-bit arithmetic (always used for s >= 8)
 *   "keep_ascii"       1 = svt_batch_upload keeps the ASCII bases in HBM for svt_batch_repack (default 0)
 *   "k9_window"        bits of the direction window the bit-parallel K9 keeps per pair-column in its first pass: 32 (default: +-16 diagonals around the expected
 *                      one, 8 bytes per column) | 64 (round 3); walks that leave the window run again around their end diagonal with 64 bits, then with the full slab
 *   (svt_get_option only) "k8a_packed_pairs" / "k8a_redo_pairs": pairs K8a has sent through the packed cell on this context / pairs of those without a certificate (rerun through the 32-bit cell)
 *   (svt_get_option only) "poa_clusters" / "poa_handed_back" / "poa_cons_device": clusters K12 has taken on this context / clusters it ended with a status / clusters whose consensus K12c delivered
 *                      (ties between unrelated end rows, capacity limits: svt_poa_result.status) and left to the caller's host engine
 *   "poa_rows"         K12's DP engine: 2 the anti-diagonal engine (default: lane = graph row, 64-row blocks pipelined over the waves of a workgroup) |
 *                      0 the chunk pipeline over eight waves (round 3) | 1 the row engine (one wave per cluster, a graph row per step; used when every
 *                      band of the launch fits 512 columns and every base is one of ACGT, the chunk pipeline otherwise).  Identical graphs from all three.
 *   "shard_world1"     test option: with a ONE-rank communicator (svt_set_shard_comm, world = 1) the sharded code paths still run, every exchange a
 *                      grouped broadcast from the rank to itself (default 0)
 * Unknown keys and out-of-range values return SVT_ERR_ARG. */
int         svt_set_option(svt_ctx* ctx, const char* key, int64_t value);
int         svt_get_option(svt_ctx* ctx, const char* key, int64_t* value);
/* A second context on the same device for ANOTHER host thread: own stream, scratch and error text; it shares the parent's
 * read-only tables (SNPmer table set by svt_set_snpmers, site order).  Batches and bitsets made by the parent can be passed to
 * calls on the fork (they are only read).  Lets independent order-dependent loops (one per k-mer cluster in Stage 3,
 * src/asv_cluster.rs:596-700) run side by side.  The parent must outlive its forks (svt_destroy(parent) destroys them);
 * call svt_fork_refresh after the parent's tables changed; a fork's profile entries are reported through the parent. */
int         svt_fork(svt_ctx* parent, svt_ctx** out);
/* Page-lock a host buffer the caller will upload from repeatedly (the read arrays of an ingest loop: svt_batch_upload then moves them with one
 * DMA instead of a chain of staged copies).  The buffer must stay alive and at the same address until svt_host_unpin.  Returns SVT_OK or
 * SVT_ERR_HIP (the caller simply goes on with pageable memory).  No reference counterpart: the reference has no device. */
int         svt_host_pin(svt_ctx* ctx, void* ptr, uint64_t bytes);
int         svt_host_unpin(svt_ctx* ctx, void* ptr);
/* Multi-GPU tile sharding (the reference shards its pair loops over rayon threads: src/asv_cluster.rs:99-196, :593-716).
 * With a shard set, the calls whose work is a list of independent tiles -- svt_minimizer_shared_counts (pairs), svt_snpmer_compat_lists_seg
 * (row tiles), svt_extract_seeds (reads) -- run only this rank's contiguous slice of the tiles on this GPU and complete their device-side
 * results through `exchange`: an IN-PLACE all-gather-v on device memory.  On entry the bytes [elem_off[rank] * elem_bytes,
 * elem_off[rank + 1] * elem_bytes) from dev_base hold this rank's part (the library has synchronised its stream); on return the whole range
 * [elem_off[0], elem_off[world]) must hold every rank's part, and the hook's own work must be complete.  Every rank makes the same calls with
 * the same arguments in the same order (the greedy decisions above the kernels stay replicated and deterministic), so the hooks meet.
 * Results are identical to the unsharded call.  world <= 1 or exchange == NULL switches sharding off.  Forked contexts do not inherit it. */
typedef int (*svt_exchange_fn)(void* user, void* dev_base, uint64_t elem_bytes, const uint64_t* elem_off);
int         svt_set_shard(svt_ctx* ctx, uint32_t rank, uint32_t world, svt_exchange_fn exchange, void* user);
/* The same with the collective issued by the library itself over RCCL (xGMI): no callback, nothing above the C-ABI takes part in an exchange.
 * svt_shard_comm_id fills `id` (SVT_COMM_ID_BYTES bytes: an ncclUniqueId) on ONE rank; the caller hands these bytes to every rank by whatever
 * means it has (a file, MPI, torch.distributed, a socket); then EVERY rank calls svt_set_shard_comm(ctx, rank, world, id), which creates the
 * communicator (ncclCommInitRank: collective, blocks until all ranks have called) on the context's device.  From then on every exchange of the
 * calls above is ONE grouped collective -- an ncclBroadcast per rank's slice between ncclGroupStart / ncclGroupEnd, in place, on the context's
 * stream (no host wait, no staging copy); arrays that travel together share one group.  svt_set_shard / svt_destroy release the communicator.
 * RCCL is bound at run time (librccl.so.1); SVT_ERR_STATE when it cannot be loaded, SVT_ERR_EXCHANGE when RCCL reports an error or a collective makes no
 * progress for "shard_timeout_s" seconds (svt_set_option; default 180: a peer died, returned early or issued another collective).  In both cases the library
 * has ABORTED the communicator (ncclCommAbort) before it returns: this rank's queued collectives leave the stream, the peers' run into their own deadline, and
 * every later exchange on the context fails at once with SVT_ERR_EXCHANGE until a new communicator is set -- no rank waits for ever, none has to be killed.
 * svt_shard_abort does the same on request: a caller whose step failed for a reason the peers do not share calls it so that they stop waiting for this rank.
 * world = 1 is allowed (a one-rank communicator; sharding is off unless the
 * test option "shard_world1" is set).  Replaces the rayon merge points src/seq_parse.rs:434-487 (C1) and src/alignment.rs:1918-1920 (C2). */
#define SVT_COMM_ID_BYTES 128
int         svt_shard_comm_id(uint8_t* id);
int         svt_set_shard_comm(svt_ctx* ctx, uint32_t rank, uint32_t world, const uint8_t* id);
int         svt_shard_abort(svt_ctx* ctx, const char* why);
/* For host code above the library that shards by OBJECT instead of by tile (Stage 3 runs the greedy loops of a rank's own k-mer clusters --
 * src/asv_cluster.rs:596 walks them one after the other although they are independent -- and gathers the resulting clusters):
 * svt_shard_info reports the shard (rank 0 of 1 when none is set); svt_shard_pause(1) switches the tile slicing of the calls above off while
 * the ranks make DIFFERENT calls (the hook stays installed), svt_shard_pause(0) back on -- it returns the previous state (0 / 1), negative on error; svt_shard_allgather_u64 gathers one value per rank;
 * svt_shard_allgatherv gathers one byte string per rank (bytes[r] from svt_shard_allgather_u64) into `all`, rank 0's first -- both go
 * through the exchange hook on a device staging buffer, so they meet the same ordering rule as every other exchange. */
int         svt_shard_info(const svt_ctx* ctx, uint32_t* rank, uint32_t* world);
int         svt_shard_pause(svt_ctx* ctx, int on);
int         svt_shard_allgather_u64(svt_ctx* ctx, uint64_t mine, uint64_t* all);
int         svt_shard_allgatherv(svt_ctx* ctx, const void* mine, const uint64_t* bytes, void* all);
int         svt_fork_refresh(svt_ctx* fork);

/* per-kernel device timing with HIP events on the context's own stream (bench.py roofline).  on = 1: every kernel; on = 2: only the kernels a roofline
 * is quoted for (the POA engine, the affine aligner's span and its forward pass) -- two events per launch cost host time and queue slots, and a 100k-read
 * step is ~180 launches: bench.py times its steps at level 2 and fills the all-kernel table from steps outside the timed region; 0: off */
int  svt_profile_enable(svt_ctx* ctx, int on);
void svt_profile_reset(svt_ctx* ctx);
int  svt_profile_count(svt_ctx* ctx);
/* name_out: >= 64 bytes.  ms = sum of event-measured durations; bytes = algorithmic bytes the
 * launches moved, units = algorithmic units (reads / pairs / k-mers) */
int  svt_profile_get(svt_ctx* ctx, int idx, char* name_out, uint64_t* launches, double* ms,
                     double* algo_bytes, double* units);

/* measured streaming-copy rate of the device's HBM (16-byte grid-stride copy of `bytes` bytes, best of `iters`), in GB/s of
 * read + written bytes: the denominator bench.py reports next to the 8 TB/s datasheet figure */
int  svt_hbm_copy_peak(svt_ctx* ctx, uint64_t bytes, int iters, double* gb_per_s);

/* ---- a1: 2-bit packing.  src/types.rs:92-134,400; src/seeding.rs:604-626 ------------------ */
/* seq: concatenated ASCII, offsets[n+1]; qual: concatenated raw quality bytes or NULL.
 * The library packs on the device: 16 bases per u32, base i at bits 30-2*(i%16), each read
 * starting on a word boundary; a 1-bit/base non-ACGT mask is kept for the ` rc` path. */
int      svt_batch_upload(svt_ctx* ctx, const uint8_t* seq, const uint8_t* qual,
                          const uint64_t* offsets, uint32_t n, svt_batch** out);
void     svt_batch_free(svt_ctx* ctx, svt_batch* b);
/* multi-GPU: a view of reads [lo, hi) of a resident batch (no copy; the parent must outlive it; free with svt_batch_free).  Accepted by
 * the calls that only read bases / qualities: svt_count_partial(_device), svt_split_kmers_emit.  rc_flags of such a call are the slice's. */
int      svt_batch_slice(svt_ctx* ctx, const svt_batch* parent, uint32_t lo, uint32_t hi, svt_batch** out);
/* K0 again on a batch uploaded under the "keep_ascii" option: rewrites the packed words, the mask and the flags from the ASCII bases
 * in HBM (so that a benchmark step can start from unpacked reads resident in HBM); SVT_ERR_STATE without the option */
int      svt_batch_repack(svt_ctx* ctx, svt_batch* b);
/* --use-hpc (src/alignment.rs:480, src/utils.rs:136): per-base tags of a batch of homopolymer-compressed reads -- qual[i] the minimum
 * quality and hp_len[i] the length of the run base i stands for, both total_bases long in batch order.  With tags on the TARGET batch,
 * svt_align_pileup / svt_pileup_create take the read qualities from them and put the run length into bits 56-63 of Base cells. */
int      svt_batch_set_tags(svt_ctx* ctx, svt_batch* b, const uint8_t* qual, const uint8_t* hp_len);
uint32_t svt_batch_size(const svt_batch* b);
/* test hook: packed words ((len+15)/16 u32) and non-ACGT mask (same count of u16) of one read */
int      svt_batch_fetch_packed(svt_ctx* ctx, const svt_batch* b, uint32_t read,
                                uint32_t* words, uint16_t* nonacgt);

/* ---- a2: seeding::split_kmer_mid, src/seeding.rs:975-1068 --------------------------------- */
/* Emits, per read r, its canonical split k-mers with the strand bit in bit 63 into
 * out[out_offsets[r] ..], count in out_counts[r].  Order inside a read is ascending k-mer END
 * position of the (possibly reverse-complemented) read, i.e. the reference's push order.
 * rc_flags[r]!=0 reproduces the ` rc` header handling of src/seq_parse.rs:362-373 (NULL = none).
 * out_offsets[r] must leave room for max(0, len_r-k+1) entries. */
int svt_split_kmers_emit(svt_ctx* ctx, const svt_batch* b, uint32_t k, uint8_t min_bq,
                         const uint8_t* rc_flags, const uint64_t* out_offsets,
                         uint64_t* out, uint32_t* out_counts);

/* ---- a3: seq_parse::read_to_split_kmers, src/seq_parse.rs:12-78,316-497 -------------------- */
/* Fused emit + count over the whole batch.  Keeps k-mers with both strand counts > 0 and
 * total > 2 (single_strand: rev count > 2), src/seq_parse.rs:33-46.  n_distinct = table size
 * before the filter (the caller applies the 0.1 % rule of :69-72). */
int svt_count_split_kmers(svt_ctx* ctx, const svt_batch* b, uint32_t k, uint8_t min_bq,
                          const uint8_t* rc_flags, int single_strand,
                          uint64_t* n_distinct, uint64_t* n_kept);
/* sorted by (masked k-mer, mid base) = sort key of src/kmer_comp.rs:480; rev = counts[0], fwd = counts[1].
 * The sorted table lives in HBM; this copies all n_kept entries out (the B1 return value of src/seq_parse.rs:12-17). */
int svt_count_fetch(svt_ctx* ctx, uint64_t* kmer, uint32_t* rev, uint32_t* fwd);
/* What kmer_comp::get_snpmers_inplace_sort (src/kmer_comp.rs:454-642) reads from that table, selected on the device, both in
 * table order: (g) the entries whose masked k-mer is shared with a neighbour -- the groups of >= 2 alleles of :507-519, the only
 * input of the binomial / Fisher tests of :543-623; (h) the entries with rev + fwd > 100 -- the order statistic of :474
 * (thresh = max(q-th largest total, 100), q = n_table / 100000 + 1) and the high-frequency k-mers of :494-496 depend on these
 * alone.  Valid after svt_count_split_kmers / svt_count_finalize. */
int svt_count_candidates_sizes(svt_ctx* ctx, uint64_t* n_table, uint64_t* n_group_entries, uint64_t* n_heavy);
int svt_count_candidates_fetch(svt_ctx* ctx, uint64_t* g_kmer, uint32_t* g_rev, uint32_t* g_fwd,
                               uint64_t* h_kmer, uint32_t* h_rev, uint32_t* h_fwd);
/* multi-GPU (C1): export / merge partial tables.  Export returns ALL distinct entries of this
 * rank's table (unfiltered, unsorted); merge adds entries into the table of this ctx; finalize
 * applies the filter + sort over the merged table. */
int svt_count_partial(svt_ctx* ctx, const svt_batch* b, uint32_t k, uint8_t min_bq,
                      const uint8_t* rc_flags, uint64_t* n_distinct);
int svt_count_export(svt_ctx* ctx, uint64_t* kmer, uint32_t* rev, uint32_t* fwd);
int svt_count_merge(svt_ctx* ctx, const uint64_t* kmer, const uint32_t* rev, const uint32_t* fwd, uint64_t n);
int svt_count_finalize(svt_ctx* ctx, uint32_t k, int single_strand, uint64_t* n_distinct, uint64_t* n_kept);
/* The same exchange with the tables staying in HBM (RCCL all-gather of device buffers, no host hop): a rank counts its read block
 * (svt_batch_slice of the resident batch), exports ALL distinct entries into caller-provided DEVICE buffers of capacity >= n_distinct,
 * the caller all-gathers them, then every rank re-creates an empty table sized for the total (svt_count_merge_begin), merges every
 * rank's buffer (svt_count_merge_device; sums, so the order is irrelevant) and calls svt_count_finalize. */
int svt_count_partial_device(svt_ctx* ctx, const svt_batch* b, uint32_t k, uint8_t min_bq, const uint8_t* rc_flags, uint64_t* n_distinct);
int svt_count_export_device(svt_ctx* ctx, uint64_t* d_kmer, uint32_t* d_rev, uint32_t* d_fwd, uint64_t cap, uint64_t* n);
int svt_count_merge_begin(svt_ctx* ctx, uint64_t total_entries);
int svt_count_merge_device(svt_ctx* ctx, const uint64_t* d_kmer, const uint32_t* d_rev, const uint32_t* d_fwd, uint64_t n);
/* C1 in ONE call under a shard (svt_set_shard_comm or svt_set_shard): this rank has counted its read block (svt_count_partial_device); the library
 * gathers the sizes, then the three entry arrays of every rank as one grouped collective on device memory, merges all entries into a fresh table
 * (sums) and applies the filter + sort of svt_count_finalize.  Every rank ends with the identical table; nothing above the C-ABI handles an entry.
 * Replaces the per-thread hash maps keyed `kmer % threads` of src/seq_parse.rs:434-487.  Without a shard it is svt_count_finalize. */
int svt_count_shard_merge(svt_ctx* ctx, uint32_t k, int single_strand, uint64_t* n_distinct, uint64_t* n_kept);

/* ---- a4 result upload: SnpmerInfo list of kmer_comp::get_snpmers_inplace_sort -------------- */
/* split_kmer[] ascending (src/kmer_comp.rs:632); both alleles form the SNPmer set
 * (src/kmer_comp.rs:71-78); high_freq[] = k-mers above the threshold (:494-496), any order.
 * site_weight[] (nullable) = SnpmerInfo.counts[0]+counts[1]: orders the INTERNAL bit positions of the bitset rows
 * (heaviest site first) so that the bits a read sets cluster in few words; it never changes a result. */
int svt_set_snpmers(svt_ctx* ctx, uint32_t k, const uint64_t* split_kmer, const uint8_t* mid0,
                    const uint8_t* mid1, const uint32_t* site_weight, uint32_t n_sites,
                    const uint64_t* high_freq, uint32_t n_hf);

/* ---- a5/a6/a7: seeding::get_twin_read_syncmer src/seeding.rs:317-658,
 *      per-read filters src/kmer_comp.rs:163-206, LSH src/types.rs:719-747 -------------------- */
/* has_qual=0 reproduces `qualities = None` (ASV FASTA path, src/kmer_comp.rs:59). */
int svt_extract_seeds(svt_ctx* ctx, svt_batch* b, uint32_t k, uint32_t c, uint8_t min_bq, int use_qual);
int svt_seeds_sizes(svt_ctx* ctx, const svt_batch* b, uint64_t* n_mini, uint64_t* n_snp, uint64_t* n_qualbin_bytes);
typedef struct svt_seeds_out {
    /* per read (n entries unless noted); any pointer may be NULL to skip */
    uint64_t* mini_off;      /* n+1 */
    uint32_t* mini_pos;      /* n_mini: k-mer start position (i+1-k) */
    uint64_t* mini_kmer;     /* n_mini: canonical k-mer (Kmer48 value) */
    uint8_t*  mini_flags;    /* n_mini: bit0 = not high-frequency ("solid", kmer_comp.rs:179), bit1 = forward strand canonical */
    uint64_t* snp_off;       /* n+1 */
    uint32_t* snp_pos;       /* n_snp */
    uint64_t* snp_kmer;      /* n_snp */
    uint8_t*  snp_flags;     /* n_snp: bit0 = not high-frequency (kmer_comp.rs:198) */
    double*   est_id;        /* n: 100 - 100*mean(10^(-q/10)) in the reference's summation order */
    uint8_t*  est_valid;     /* n: 0 = None (all qualities equal, or no qualities) */
    uint64_t* lsh;           /* n*20 */
    uint8_t*  lsh_valid;     /* n: 1 if >= 3 minimizers */
    uint32_t* n_unique;      /* n: |set(minimizer k-mers)| */
    uint32_t* n_solid;       /* n: minimizers with in-read multiplicity <= 500 and not high-frequency (kmer_comp.rs:163-183) */
    uint64_t* qualbin_off;   /* n+1 byte offsets into qualbins */
    uint8_t*  qualbins;      /* 4-bit bins (types.rs:447-467), two per byte, low nibble first */
    uint8_t*  status;        /* n: 0 ok, 1 = read shorter than k (None, seeding.rs:339), 2 = SNPmer buffer overflow */
} svt_seeds_out;
int svt_seeds_fetch(svt_ctx* ctx, const svt_batch* b, const svt_seeds_out* out);
/* mean[r] = (sum over the 4-bit quality bins of read r, in order, of table16[bin]) / #bins, 1.0 for a read without bins: with
 * table16[b] = 1 - 10^(-3b/10) the mean base accuracy that ranks a cluster's reads for the POA (src/alignment.rs:254-260).  The f64
 * additions run in bin order and the table is the caller's, so the values equal the reference's fold bit for bit. */
int svt_qualbin_mean(svt_ctx* ctx, const svt_batch* b, const double* table16, double* mean);

/* ---- a6 on the device: which reads become twin reads, and in which order (kmer_comp::twin_reads_from_snpmers src/kmer_comp.rs:117,185,233,248;
 * get_twin_reads_from_kmer_info src/main.rs:538).  svt_twin_order applies the intake filters to the seeded batch -- length in [min_len, max_len]
 * (:117), seeds exist (src/seeding.rs:339), solid minimizers >= len / c / 20 in integer arithmetic (:185), estimated identity >= cutoff or absent (:248) --
 * and sorts the survivors by identity descending, STABLE in read order (a 64-bit radix sort of the identity's bit pattern; a read without an estimate
 * counts as 100.0, :538).  order[0 .. *n_kept) receives the read indices, est_key the sort keys in that order: equal keys are equal identities -- the
 * reference breaks such ties by the read id (:233), which the device does not hold, so the CALLER re-sorts every run of equal keys by (id, read index)
 * before it goes on (runs are rare: an identity is a mean of ~1500 error probabilities).  Capacity of both arrays: the batch's read count.
 * svt_twin_gather returns, for read indices in any order, the per-read records the host stages consume, in THAT order: length, minimizer count, distinct
 * minimizer count, SNPmers that are not high-frequency (src/kmer_comp.rs:190-202), identity + valid flag, the 20 LSH signatures (lsh may be NULL) + valid flag. */
int svt_twin_order(svt_ctx* ctx, const svt_batch* b, uint32_t min_len, uint32_t max_len, uint32_t c, double cutoff, uint32_t* n_kept, uint32_t* order, uint64_t* est_key);
int svt_twin_gather(svt_ctx* ctx, const svt_batch* b, const uint32_t* order, uint32_t n, uint32_t* length, uint32_t* n_mini, uint32_t* n_unique, uint32_t* n_snp_filtered,
                    double* est_id, uint8_t* est_valid, uint64_t* lsh, uint8_t* lsh_valid);

/* ---- a9 candidate lists of a Stage-2 block: src/asv_cluster.rs:303-337 (query_read_against_bucket_index) + :111-125 (the list rule) -----------------
 * The reference counts, for one read, in how many of the 20 LSH tables each representative shares its signature (a walk over 20 hash buckets).  Here a
 * whole block of queries meets a list of references on the device, every pair compared directly: hits(i, j) = number of tables in which read q_idx[i]
 * and read r_idx[j] have equal signatures (both must have signatures: a query without gets the count 0).  Only the references j < ref_limit[i] count
 * for query i when ref_limit is given.
 *   mode 0: the list rule -- the references with hits > 0 ordered by (hits, j) descending (the caller lists its representatives in ascending read id, so j
 *           descending is id descending, :111); kept: every reference with the maximum hit count, or the first top_n, whichever is longer.
 *           entries {hits, j}.
 *   mode 1: every reference with hits > 0 in ascending j: entries {j, hits}.
 * The lists lie back to back in `out` (two words per entry, `capacity` entries): query i owns out_cnt[i] entries from entry out_off[i]; *n_out = entries used.
 * out_cnt[i] = 0xFFFFFFFF when more than `cap` (<= 256) references share a signature with the query, or when `out` was full: the caller builds that one
 * list itself. */
int svt_lsh_candidates(svt_ctx* ctx, const svt_batch* b, const uint32_t* q_idx, uint32_t n_q, const uint32_t* r_idx, uint32_t n_ref, const uint32_t* ref_limit,
                       uint32_t mode, uint32_t top_n, uint32_t cap, uint32_t capacity, uint32_t* out_cnt, uint32_t* out_off, uint32_t* out, uint32_t* n_out);

/* ---- a9 verify loop / a13 minimizer overlap: src/asv_cluster.rs:131-143, src/alignment.rs:1798-1799 */
/* For each pair (a_idx[i] in batch A, b_idx[i] in batch B): shared[i] = |set(A) ∩ set(B)| over
 * distinct minimizer k-mers; same_strand[i] = how many of those have equal canonical-orientation
 * flags at their first occurrence (K8 strand vote).  Denominators: n_unique / mini counts. */
int svt_minimizer_shared_counts(svt_ctx* ctx, const svt_batch* A, const svt_batch* B,
                                const uint32_t* a_idx, const uint32_t* b_idx, uint64_t n_pairs,
                                uint32_t* shared, uint32_t* same_strand);

/* ---- a10/a12/a13 SNPmer compatibility: src/asv_cluster.rs:356-383 -------------------------- */
/* views of a read's SNPmers: ALL = snpmer_kmers(), FILTERED = snpmers_vec() */
#define SVT_VIEW_ALL 0
#define SVT_VIEW_FILTERED 1
/* list filters */
#define SVT_LIST_COMPATIBLE 0   /* mismatches == 0 && matches > 0  (asv_cluster.rs:481-483) */
#define SVT_LIST_OVERLAP 1      /* matches + mismatches > 0        (asv_cluster.rs:364-380 keys) */
/* number of 64-bit words per bitset row = ceil(n_sites/64) */
uint32_t svt_snpmer_words(const svt_ctx* ctx);
/* order[b] = caller's site index stored at bit position b of the bitset rows (n_sites entries) */
int svt_snpmer_site_order(const svt_ctx* ctx, uint32_t* order);
/* fetch bitset rows of a batch: presence_all, presence_filtered, allele (n * words u64 each) */
int svt_snpmer_bits_fetch(svt_ctx* ctx, const svt_batch* b, uint64_t* p_all, uint64_t* p_filt, uint64_t* allele);
/* upload consensus rows (src/asv_cluster.rs:840-894 results as bitsets) */
int  svt_bitset_upload(svt_ctx* ctx, const uint64_t* presence, const uint64_t* allele, uint32_t n_rows, svt_bitset** out);
void svt_bitset_free(svt_ctx* ctx, svt_bitset* s);
/* rows x cols tile -> sparse triples.  Rows: row_idx[] into batch R with row_view.  Cols: either
 * col batch C (+col_view) with col_idx[], or a bitset set S (C == NULL) with col_idx[] (NULL = all).
 * triangular != 0: col j is only compared with row i when j_position < tri_base + i
 * (cols [tri_base, n_cols) are the block's own rows: in-block "earlier read" columns, col tri_base + i == row i).
 * triangular == 2 (greedy assignment, src/asv_cluster.rs:596-660: a read becomes a representative only if NO existing
 * representative is compatible with it): an in-block column is reported only if its own row has no listed pair among the
 * first tri_base columns -- the other in-block columns can never be representatives, the caller would discard their pairs.
 * row_max_mismatch (nullable): per row, pairs with more mismatches are dropped on the device (Stage 7: the ratio rule
 * of src/alignment.rs:1811,1829-1833 bounds the mismatches a surviving pair can have by 0.005*c*|read minimizers|).
 * Output triples (row position, col position, matches<<16|mismatches) unordered; returns
 * SVT_ERR_OVERFLOW with *n_out = needed when cap is too small. */
int svt_snpmer_compat_lists(svt_ctx* ctx, const svt_batch* R, int row_view, const uint32_t* row_idx, uint32_t n_rows,
                            const svt_batch* C, int col_view, const svt_bitset* S, const uint32_t* col_idx, uint32_t n_cols,
                            int filter, int triangular, uint32_t tri_base, const uint32_t* row_max_mismatch,
                            uint32_t* out_row, uint32_t* out_col, uint32_t* out_mm, uint64_t cap, uint64_t* n_out);

/* The same for MANY k-mer clusters in one call (one wave of the greedy loops of src/asv_cluster.rs:596-660, which walks the k-mer clusters
 * one after the other although they are independent): segment s has the rows row_idx[seg_row_off[s] .. seg_row_off[s+1]) and the columns
 * col_idx[seg_col_off[s] .. seg_col_off[s+1]) = the cluster's existing representatives followed by the segment's rows themselves (in that
 * order), all reads of batch R under `view`.  Triangular mode 2 of svt_snpmer_compat_lists per segment: a representative column is listed
 * with every row it is compatible with; an in-block column only with LATER rows, and only if its own row met no compatible representative.
 * Triples: out_row = position in row_idx (global), out_col = column position INSIDE the segment, out_mm = matches << 16 | mismatches,
 * unordered.  SVT_ERR_OVERFLOW with *n_out = needed when cap is too small; SVT_ERR_STATE when the SNPmer rows do not fit the LDS tile
 * (more than 9 600 sites: call svt_snpmer_compat_lists per cluster). */
int svt_snpmer_compat_lists_seg(svt_ctx* ctx, const svt_batch* R, int view, const uint32_t* row_idx, uint32_t n_rows, const uint32_t* seg_row_off,
                                const uint32_t* col_idx, const uint32_t* seg_col_off, uint32_t n_seg, int filter,
                                uint32_t* out_row, uint32_t* out_col, uint32_t* out_mm, uint64_t cap, uint64_t* n_out);
/* The same call with the entries handed back ROW BY ROW, the form the greedy loop reads them in (src/asv_cluster.rs:596-660 walks the compatible
 * representatives of one read after the other): out_off[0 .. n_rows] = offsets (out_off[n_rows] = *n_out), the entries of row r are
 * out_col / out_mm [out_off[r], out_off[r + 1]), in no particular order inside a row.  The records are counted and dropped into their rows on the
 * device (the host did three passes over ~10^6 triples per 100k-read step for it).  cap < 2^32.  Errors as above; on SVT_ERR_OVERFLOW out_off is zero. */
int svt_snpmer_compat_rows_seg(svt_ctx* ctx, const svt_batch* R, int view, const uint32_t* row_idx, uint32_t n_rows, const uint32_t* seg_row_off,
                               const uint32_t* col_idx, const uint32_t* seg_col_off, uint32_t n_seg, int filter,
                               uint32_t* out_off, uint32_t* out_col, uint32_t* out_mm, uint64_t cap, uint64_t* n_out);
/* a12-a14 fused for Stage 7 (src/alignment.rs:1786-1846): for the reads row_idx[0..n_rows) of batch R against the first n_asvs
 * sequences of batch A (both seeded): candidates = pairs sharing a SNPmer site with mismatches <= row_max_mismatch[row] (NULL =
 * no bound); a candidate survives if shared != 0, shared / min(|read minimizer set|, |ASV set|) >= min_frac (f64, :1805-1808) and
 * mismatches / shared / c_param <= 0.005 (f64, :1811-1833); per read the survivors with the lowest mismatch count are returned
 * (:1841-1846) with the K7 strand vote (1 = reverse) and, when tie_mismatches is not NULL, that mismatch count (a column of
 * read_to_asv_mappings.tsv, :1877-1885).  Unordered; tie_row = position in row_idx.  SVT_ERR_OVERFLOW with
 * *n_ties = needed when cap is too small.  n_candidates (optional) = size of the candidate list. */
int svt_read_asv_ties(svt_ctx* ctx, const svt_batch* R, const uint32_t* row_idx, uint32_t n_rows, const svt_batch* A, uint32_t n_asvs,
                      const uint32_t* row_max_mismatch, double min_frac, double c_param,
                      uint32_t* tie_row, uint32_t* tie_col, uint8_t* tie_rev, uint32_t* tie_mismatches, uint64_t cap, uint64_t* n_ties,
                      uint64_t* n_candidates);
/* a11: build_consensus_snpmers_top_n (top_n = None), src/asv_cluster.rs:840-894, for MANY clusters in one call.
 * Clusters are a CSR over read indices of batch R (cl_off[n_clusters+1], members[]); the FILTERED view
 * (snpmers_vec()) is used.  Per site: consensus allele = most common allele among the members (tie -> the
 * allele with the smaller mid base), kept iff its count >= max(1, len/6).  Outputs n_clusters*words u64 each
 * (host, may be NULL) and, if out_set != NULL, the same rows as a device-resident svt_bitset. */
int svt_snpmer_consensus(svt_ctx* ctx, const svt_batch* R, const uint64_t* cl_off, const uint32_t* members, uint32_t n_clusters,
                         uint64_t* presence, uint64_t* allele, svt_bitset** out_set);
/* reassign_reads_to_best_cluster, src/asv_cluster.rs:1057-1097: per row the FIRST column of [col_lo[i], col_hi[i])
 * (NULL = all columns) with the lexicographically smallest (mismatches, -matches); best_col[i] = column index,
 * score = matches<<16|mismatches */
int svt_snpmer_best_column(svt_ctx* ctx, const svt_batch* R, int row_view, const uint32_t* row_idx, uint32_t n_rows,
                           const svt_bitset* S, const uint32_t* col_lo, const uint32_t* col_hi,
                           uint32_t* best_col, uint32_t* best_score);

/* ---- a14 (K8): replaces minimap2 `nm`, src/alignment.rs:1848-1862 -------------------------- */
/* Banded overlap edit distance (contract in DESIGN.md section 3 / oracle header): query = sequence
 * q_idx[i] of batch Q (ASV), target = sequence t_idx[i] of batch T (read), reverse[i]!=0 aligns
 * the reverse complement of the target; band[i] = half width w (|j-i| <= w), w <= 511. */
int svt_align_nm(svt_ctx* ctx, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx,
                 const uint32_t* t_idx, const uint8_t* reverse, const uint32_t* band,
                 uint64_t n_pairs, int32_t* nm);

/* K8a, the "affine contract" of the same `nm`: minimap2's map-ont / lr:hq scoring (a = 2, b = 4,
 * gap(l) = min(4 + 2 l, 24 + l)), best LOCAL alignment inside the band, nm = mismatches + gap bases along it, among the
 * alignments of maximum score the one with the fewest nm.  Same arguments as svt_align_nm; nm[i] = INT32_MAX when nothing
 * aligns (score 0); score (may be NULL) receives the alignment score.  The kernel carries nm in 12 bits beside the score: the result is
 * exact whenever the optimum's nm is below 4096 (any pair of reads / consensuses this path aligns); a more distant optimum is not representable. */
int svt_align_nm_affine(svt_ctx* ctx, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx,
                        const uint32_t* t_idx, const uint8_t* reverse, const uint32_t* band,
                        uint64_t n_pairs, int32_t* nm, int32_t* score);

/* K8a near the unit-cost optimum, what Stage 7 calls for `nm` (src/alignment.rs:1848-1862) by default: the same affine alignment, inside the
 * band the unit-cost optimum allows.  For every pair the library first takes the unit-cost overlap distance d inside min(band, 255) and the
 * diagonal e = j - i of its end cell (lowest value; ties: smallest i + j, then smallest j - i -- the cell svt_align_pileup walks back from);
 * an overlap alignment of cost d that ends on e never leaves |j - i| <= |e| + d, and the affine DP runs in
 * |j - i| <= min(band, |e| + d + 8).  minimap2 aligns around its chain, not in a fixed band; this confines the DP the same way and costs a
 * quarter of the whole band.  Pairs without an end cell inside min(band, 255) keep their band.  band_used (may be NULL) receives the band of
 * every pair; nm / score as svt_align_nm_affine. */
int svt_align_nm_affine_near(svt_ctx* ctx, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx,
                             const uint32_t* t_idx, const uint8_t* reverse, const uint32_t* band,
                             uint64_t n_pairs, int32_t* nm, int32_t* score, uint32_t* band_used);

/* ---- a16 (K9): pile-up rows, src/alignment.rs:449-575 (the minimap2 map-ont + CIGAR walk of generate_consensus_pileups) */
/* Same banded DP as svt_align_nm plus a deterministic traceback (priority diagonal > deletion > insertion; end cell = smallest
 * value on the last row/column, ties -> smallest i+j then smallest j-i).  Query q_idx[i] of batch Q is the consensus
 * (reference side), target t_idx[i] of batch T the read; T's quality bins are used when svt_extract_seeds(T, use_qual=1) ran.
 * cells: one u64 per consensus position, rows concatenated at cell_off[i] (cell_off[n_pairs] = total); encoding:
 * bits 0-2 code (0-3 read base, 4 deletion, 7 not covered), 8-15 quality (bin*3+33), 16-17 kept inserted bases after this
 * position (<= 2 = MAX_INSERTION_LENGTH), 18-25 insertion length, 32-35 the kept bases, 40-55 their qualities, 56-63 the homopolymer
 * run length of a Base entry when the target batch carries tags (svt_batch_set_tags; 0 otherwise).
 * span: 4 per pair (q_start, q_end, t_start, t_end; target coordinates in the aligned orientation). */
int svt_align_pileup(svt_ctx* ctx, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx, const uint32_t* t_idx,
                     const uint8_t* reverse, const uint32_t* band, uint64_t n_pairs, const uint64_t* cell_off,
                     uint64_t* cells, uint32_t* span, int32_t* nm);

/* ---- a17 (K10): pile-ups that stay in HBM + per-column statistics, src/alignment.rs:663-786 and :893-1029 ---------------- */
/* svt_pileup_create runs K9 for n_pairs (consensus, read) pairs and keeps the rows on the device.  Pairs are grouped:
 * grp_off[n_groups+1] are pair ranges, every pair of a group has the same query (one group = one consensus and the reads
 * piled onto it, in push order).  Columns are numbered group after group (col = prefix sum of the groups' query lengths;
 * an empty group has no columns).  span (4 per pair) may be NULL; nm as in svt_align_pileup. */
typedef struct svt_pileup svt_pileup;
int      svt_pileup_create(svt_ctx* ctx, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx, const uint32_t* t_idx,
                           const uint8_t* reverse, const uint32_t* band, uint64_t n_pairs, const uint64_t* grp_off, uint32_t n_groups,
                           svt_pileup** out, uint32_t* span, int32_t* nm);
void     svt_pileup_free(svt_ctx* ctx, svt_pileup* p);
uint64_t svt_pileup_cells(const svt_pileup* p);      /* total u64 cells (sum of query lengths over pairs) */
uint64_t svt_pileup_columns(const svt_pileup* p);    /* total columns (sum of query lengths over non-empty groups) */
/* test hook: rows (svt_align_pileup encoding) and their offsets (n_pairs+1) back to the host; either may be NULL */
int      svt_pileup_fetch(svt_ctx* ctx, const svt_pileup* p, uint64_t* cells, uint64_t* cell_off);
/* per column: depth = pile-up entries (base / deletion, plus one per insertion), err = entries that are not a base equal to
 * the consensus base (src/alignment.rs:701-719).  qual_total / qual_err (256 each): over the Base entries of the columns
 * with err/depth < 0.05 in the groups with grp_selected[g] != 0 (NULL = none): entries per quality value and how many of
 * them differ from the consensus base (:722-734, without the (1,1) prior). */
int      svt_pileup_stats(svt_ctx* ctx, const svt_pileup* p, const uint8_t* grp_selected, uint32_t* depth, uint32_t* err,
                          uint64_t* qual_total, uint64_t* qual_err);
/* per column: the median run length over the Base entries (bits 56-63; even counts: integer mean of the two middle values; no Base
 * entry: 1) -- the consensus' hp_lengths under --use-hpc (src/alignment.rs:586-625, :652-656). */
int      svt_pileup_hp_median(svt_ctx* ctx, const svt_pileup* p, uint8_t* median);
/* per column: lr = sum ln P(entry | consensus base right), ln = sum ln P(entry | wrong), in push order (:946-987).
 * ln_table[2q] = ln(1 - error_rate(q)), ln_table[2q+1] = ln(error_rate(q)) for q = 0..255, evaluated by the CALLER (the
 * device adds, it never calls log); a deletion adds (ln_indel_err, ln_indel_acc), an insertion (ln er(q0), ln(1-er(q0)))
 * with q0 the quality of its first base. */
int      svt_pileup_loglik(svt_ctx* ctx, const svt_pileup* p, const double* ln_table, double ln_indel_err, double ln_indel_acc,
                           double* lr, double* ln);

/* ---- a15 (K12): the whole Stage-4a POA of many clusters in ONE launch, the partial-order graphs resident on the device ----
 * Replaces the per-cluster loop of src/alignment.rs:193-231 (spoars: graph.add_alignment(engine.align(seq, graph), seq, weights) for
 * every sequence of a cluster, src/alignment.rs:222-227) for the clusters of src/alignment.rs:241 (a par_iter over clusters).
 * Cluster c owns the sequences [cl_off[c], cl_off[c+1]) (the seed first, src/alignment.rs:315); sequence s = seq[seq_off[s] ..
 * seq_off[s+1]) with one weight byte per base (the quality byte, :224) and the band half-width seq_band[s] = max length deviation
 * + (int)(0.1 * len) + 1 (BandConfig, :209-221; evaluated by the caller in double precision).  Alignment contract (the CPU twin is
 * savont_amd/csrc/host/poa.hpp; scores 3 / -8 / -6, overlap mode): a node v may align with the sequence columns [c_v - band, c_v + band]
 * around the rounded mean position c_v of its fused bases; cell(v, j) = max over predecessors p of {cell(p, j-1) + (code == seq[j-1] ?
 * match : mismatch), cell(p, j) + gap} and cell(v, j-1) + gap, floored at -30000; the virtual source row and column 0 are 0 (free leading
 * overhangs); the end cell is the first maximum, in (row, column) order, over the sink rows and column seq_len (free trailing overhangs);
 * the traceback prefers (mis)match (first predecessor reaching the value), then deletion, then insertion; fusing = spoa's add_alignment.
 * The kernel keeps its own topological order: results are identical unless equal maxima in unrelated end rows would
 * have to be ordered the way spoa's depth-first sort orders them; such a cluster -- and one that exceeds a capacity (nodes 4*max_len
 * + 2048, six aligned siblings per node, band half-width 640, sequence length 5440) -- comes back with status != 0 and no graph:
 * the caller redoes it with its CPU engine.  res[c].n_nodes / n_edges size the graph; node_off / edge_off (n_clusters + 1 each)
 * receive the offsets of every cluster's nodes / edges in the arrays of svt_poa_graphs_fetch, which must be the next call on ctx:
 *   code[v]            the node's letter
 *   aligned[8 v ..]    {count, up to six aligned node ids (cluster-local), 0}, in spoa's list order
 *   edges[3 e ..]      {tail, head, weight} in creation order: a node's in-edges / out-edges in list order are the edges naming
 *                      it as head / tail in this order. */
typedef struct svt_poa_result {
    int32_t status; uint32_t n_nodes, n_edges, ties, rows_done, tie_reads, far_rows, cons_len;   /* cons_len: length of the cluster's consensus (K12c), 0xFFFFFFFF when the device left it to the caller */
    uint64_t ticks[6];   /* diagnostics: ticks of the 100 MHz device clock in row descriptors, DP, end cell, traceback, fuse, order splice */
    uint32_t spins[8], tasks[8];   /* per wave of the cluster's workgroup: polls that found a neighbour wave not ready; (row, chunk) tasks computed */
} svt_poa_result;
int svt_poa_graphs(svt_ctx* ctx, uint32_t n_clusters, const uint64_t* cl_off, const uint64_t* seq_off, const uint8_t* seq, const uint8_t* weights,
                   const uint32_t* seq_band, svt_poa_result* res, uint64_t* node_off, uint64_t* edge_off);
/* the same in two halves: _submit queues the uploads and the launch and returns (the caller's arrays must stay alive and no other call
 * may use ctx's scratch until _wait); _wait blocks until the launch is done, fills res / node_off / edge_off and prepares the fetch.
 * The caller's own engine can work on other clusters in between (savont_amd/csrc/host/consensus.cpp splits a step's clusters that way). */
int svt_poa_graphs_submit(svt_ctx* ctx, uint32_t n_clusters, const uint64_t* cl_off, const uint64_t* seq_off, const uint8_t* seq, const uint8_t* weights,
                          const uint32_t* seq_band);
/* _submit with the sequences taken from a RESIDENT batch instead of host arrays: sequence s = read read_idx[s] of b, reverse-complemented when reverse[s]
 * (reverse may be NULL); its letters are the read's 2-bit codes decoded (A C G T; every other input byte was stored as A, src/types.rs:92-101) and its
 * weights the read's 4-bit quality bins decoded as src/alignment.rs:248-273 does (bin * 3 + 33 for the four bases of a bin; 33 without qualities), in
 * the same orientation -- what the caller would have built on the host, without the copies.  Lengths come from the batch. */
int svt_poa_graphs_submit_reads(svt_ctx* ctx, const svt_batch* b, uint32_t n_clusters, const uint64_t* cl_off, const uint32_t* read_idx, const uint8_t* reverse, const uint32_t* seq_band);
int svt_poa_graphs_wait(svt_ctx* ctx, svt_poa_result* res, uint64_t* node_off, uint64_t* edge_off);
int svt_poa_graphs_fetch(svt_ctx* ctx, uint8_t* code, uint16_t* aligned, uint32_t* edges);   /* three NULLs: no copy, the result just ends */
/* K12c: the consensus of every finished graph, computed on the device behind the graphs (spoa's depth-first order with aligned nodes adjacent + heaviest bundle
 * with branch completion, src/alignment.rs:223-231 generate_consensus; ties decided as the CPU engine decides them).  After svt_poa_graphs_wait and BEFORE
 * svt_poa_graphs_fetch (which ends the result): cons_off (n_clusters + 1) receives the offsets, cons the letters (capacity: the sum of res[j].cons_len over the
 * clusters with status 0 and cons_len != 0xFFFFFFFF -- a graph too large for the workgroup's LDS has that flag and an empty slot: the caller fetches the
 * graph and walks it itself).  cons may be NULL when that sum is 0. */
int svt_poa_consensus_fetch(svt_ctx* ctx, const svt_poa_result* res, uint64_t* cons_off, uint8_t* cons);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
