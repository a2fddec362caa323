// svt_internal.hpp -- shared declarations of libsavont_hip.so (gfx950 only; no compatibility paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/savont_hip.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;
typedef unsigned long long ull;

#define SVT_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define SVT_PAD_WORDS 2u          // zero words appended to every packed read (window reads never fault)

// ---------------------------------------------------------------------------------------------
// device-resident views
// ---------------------------------------------------------------------------------------------
struct BatchView {
    u32 n;
    const u64* off;        // [n+1] base offsets into qual / ascii order
    const u64* woff;       // [n+1] word offsets into packed / nmask (includes SVT_PAD_WORDS per read)
    const u32* packed;     // 2-bit, MSB-first: base i of a read at bits 30-2*(i%16) of word i/16
    const u16* nmask;      // non-ACGT mask, bit 15-(i%16) of u16 i/16 (same word indexing as packed)
    const u8* qual;        // raw quality bytes or nullptr
    const u8* flags;       // [n] bit0 = all qualities equal, bit1 = read has a non-ACGT base
};

struct HtEntry { ull key; u32 c[2]; };   // c[0] = reverse-canonical count, c[1] = forward (seq_parse.rs:456-459)

struct SnpTable {          // open-addressing table over both alleles of every SNPmer site
    const u64* keys;       // SVT_EMPTY_KEY = empty
    const u32* vals;       // site<<1 | allele bit (allele bit = 1 for the larger mid base)
    u32 mask;              // capacity-1 (power of two)
    const u64* hf;         // sorted high-frequency k-mers
    u32 n_hf;
    u32 n_sites;
    u32 words;             // ceil(n_sites/64)
};

struct SeedsDev {          // per-batch outputs of svt_extract_seeds (all device pointers)
    bool valid = false;
    u32 k = 0, c = 0;
    u64 mini_cap = 0, snp_cap = 0;
    u32 max_set = 0;           // the largest per-read minimizer capacity: no read's sorted set is larger (sizes the LDS copy of K5 / K7)
    u64* mini_base = nullptr;  // [n] start of each read's region inside mini_* (fixed capacity layout)
    u32* mini_cnt = nullptr;   // [n]
    u32* mini_pos = nullptr;   // [mini_cap]
    u64* mini_kmer = nullptr;  // [mini_cap]  bit63 = forward strand canonical
    u8* mini_flags = nullptr;  // [mini_cap]  bit0 solid, bit1 canon
    u64* set_kmer = nullptr;   // [mini_cap]  sorted distinct k-mers, bit63 = canon flag of first occurrence
    u32* set_cnt = nullptr;    // [n]
    u32* n_solid = nullptr;    // [n]
    u64* snp_base = nullptr;   // [n] (assigned by an atomic cursor inside the kernel)
    u32* snp_cnt = nullptr;    // [n]
    u32* snp_pos = nullptr;    // [snp_cap]
    u64* snp_kmer = nullptr;   // [snp_cap]
    u8* snp_flags = nullptr;   // [snp_cap]
    ull* snp_cursor = nullptr; // [1]
    double* est_id = nullptr;  // [n]
    u8* est_valid = nullptr;   // [n]
    u64* lsh = nullptr;        // [n*20]
    u8* lsh_valid = nullptr;   // [n]
    u64* qb_off = nullptr;     // [n+1]
    u8* qualbins = nullptr;
    u64 qb_bytes = 0;
    u8* status = nullptr;      // [n]
    bool rows_partial = false; // svt_set_shard: the dense bitset rows p_all / p_filt / allele hold only this rank's reads (the kernels of stages 2-3 read the sparse nz_* form; ensure_dense_rows completes them where a path needs them)
    bool mini_partial = false; // svt_set_shard: mini_pos / mini_kmer / mini_flags hold only this rank's reads (no stage reads them; svt_seeds_fetch completes them on demand)
    u8* meta_block = nullptr;  // ONE allocation holding mini_base | qb_off (sent together) and est_id | set_cnt | n_solid | mini_cnt | snp_cnt | est_valid |
    u64 meta_fetch_off = 0, meta_fetch_bytes = 0;   // lsh_valid | status | snp_cursor (fetched together): the per-read records the host reads after extraction
    // SNPmer bitsets (row-major, `words` u64 per read)
    u32 words = 0;
    u64* p_all = nullptr; u64* p_filt = nullptr; u64* allele = nullptr;
    // sparse form of the same rows: the non-zero 64-bit words of presence_all, indexed like the SNPmer lists
    // (region of read r starts at snp_base[r], holds nz_cnt[r] <= snp_cnt[r] entries)
    u32* nz_cnt = nullptr; u32* nz_idx = nullptr; u64* nz_pa = nullptr; u64* nz_pf = nullptr; u64* nz_a = nullptr;
};

struct svt_batch {
    u32 n = 0;
    u64 total_bases = 0, total_words = 0;
    u32 max_len = 0;
    bool has_qual = false;
    std::vector<u64> h_off, h_woff;   // host copies
    u64* d_off = nullptr; u64* d_woff = nullptr; u32* d_packed = nullptr; u16* d_nmask = nullptr;
    u8* d_qual = nullptr; u8* d_flags = nullptr;
    u8* d_ascii = nullptr;            // kept only under the "keep_ascii" option
    u8* d_block = nullptr;            // small batches: d_off, d_woff, the ASCII bases and d_qual are regions of this one allocation
    u8* d_tag_qual = nullptr; u8* d_tag_hp = nullptr;   // svt_batch_set_tags: per-base quality byte and homopolymer run length (--use-hpc reads)
    const svt_batch* slice_of = nullptr;   // non-null: a view of reads [lo, hi) of that batch (svt_batch_slice); the device arrays belong to it
    SeedsDev seeds;
    BatchView view() const { return BatchView{n, d_off, d_woff, d_packed, d_nmask, d_qual, d_flags}; }
};

struct svt_bitset {
    u32 n_rows = 0, words = 0;
    u64* p = nullptr; u64* a = nullptr;
};

struct ProfEntry { std::string name; u64 launches = 0; double ms = 0, bytes = 0, units = 0; };
struct PendingEvt { int idx; hipEvent_t a, b; };

// kernel / copy-path selections (svt_set_option); read through svt_ctx::opt() so that forks follow their root context
struct SvtOptions {
    int k8_kernel = 0;          // 0 = bit-parallel (default), 1 = anti-diagonal wavefront
    int shard_seeds = 0;        // under svt_set_shard: 1 = svt_extract_seeds runs the rank's read block only and gathers the seed arrays (default 0: replicated)
    int k9_kernel = 0;          // 0 = by launch size, 1 = anti-diagonal wavefront, 2 = bit-parallel (windowed slab), 3 = bit-parallel, full slab
    int count_table_hint = 1;   // 1 = the table of a batch is sized from the distinct count of the batch this context counted before (same order of size); 0 = from the positions alone
    int count_kernel = 0;       // 0 = windowed LDS counting, a lane per read (default), 1 = wave per read straight into the HBM table, 2 = windowed, a wave per read (rounds 2-5)
    int consensus_dense = 0;    // 1 = dense-row consensus kernel
    int consensus_chunk = 0;    // members per block of the sparse consensus kernel (0 = default 256)
    int pin_staging = 0;        // 1 = stage small calls through pinned host memory (SDMA path)
    int zero_copy = 1;          // 0 = no zero-copy I/O for small calls
    int sync_block = 0;         // 1 = wait on a blocking event instead of spinning in hipStreamSynchronize
    int shard_world1 = 0;       // test option: a one-rank RCCL communicator still runs the sharded paths (exchanges = broadcasts to self)
    int poa_rows = 2;           // K12's DP engine: 2 = the anti-diagonal engine (lane = graph row, 64-row blocks pipelined over the waves; round 4: DP 134 -> 63 ms per 75-read cluster), 1 = the row engine (one wave per cluster, a graph row per step) when the bands fit, 0 = the chunk pipeline over eight waves (round 3)
    int k9_window = 32;         // bits of the direction window K9's windowed slab keeps per pair-column: 64 (round 3) or 32 (half the slab; walks that leave it run again)
    int shard_timeout_s = 180;  // seconds a wait behind a grouped collective of the shard communicator may last before the communicator is aborted (a peer never joined)
    int k8a_queue = 1;          // K8a: 1 = ONE launch, the waves draw (class, pairs) tasks from a queue in falling cost (round 5); 0 = a launch per band class on side streams (round 4)
    int k8a_pk16 = 1;           // K8a: 1 = pairs with bands <= 39 and |n - m| <= 64 go through the packed 16-bit cell (two pairs per lane group; a certificate per pair, the others rerun through the 32-bit cell); 0 = the 32-bit cell for all
    int k8a_g16 = 1;            // K8a: 0 = no sixteen-pair classes (round 4's eight pairs per wave at most; comparison runs)
    int seeds_hash = 0;         // K3: 1 = round 5's kernel (mm_hash64 of every canonical s-mer in 64-bit arithmetic, a wave per workgroup); 0 = the rank-table kernel when s = k - c + 1 <= 7 (comparison runs, tests)
    int keep_ascii = 0;         // 1 = svt_batch_upload keeps the ASCII bases in HBM so that svt_batch_repack can redo K0 (bench: the pack is part of a timed step)
};

struct svt_ctx {
    int device = 0;
    SvtOptions options;
    const SvtOptions& opt() const { return parent ? parent->options : options; }
    hipStream_t stream = nullptr;
    hipEvent_t ev_block = nullptr;   // blocking-wait event (ctx_sync)
    unsigned* sync_word = nullptr; unsigned sync_seq = 0;   // page-locked host word a one-lane kernel sets behind a long launch (ctx_sync_long)
    static constexpr int N_SIDE = 7;
    hipStream_t side[N_SIDE] = {}; hipEvent_t side_go = nullptr, side_done[N_SIDE] = {};   // side streams: independent launches of one call (the band classes of K8a) run side by side, their tails overlap
    std::string err;
    // counting table
    HtEntry* ht = nullptr; u64 ht_cap = 0; u64 ht_distinct = 0; u64 ht_positions = 0;
    bool ht_fresh = false;                                   // the table holds exactly one counted batch (no merge since)
    u64 ht_hint_distinct = 0, ht_hint_positions = 0;          // the last counted batch of this context: distinct keys and k-mer positions (sizes the next table)
    std::vector<u64> cnt_kmer; std::vector<u32> cnt_rev, cnt_fwd;   // host copy of the table (mode 2: always; sorted table: filled on demand)
    // sorted table in HBM (count_collect modes 0/1) + the two Stage-1b selections over it (fetched eagerly: they are short)
    u64* tab_kmer = nullptr; u32* tab_rev = nullptr; u32* tab_fwd = nullptr; u64 tab_n = 0, tab_cap = 0; bool tab_valid = false, tab_on_host = false;
    void* tab_tmp = nullptr; size_t tab_tmp_bytes = 0;
    std::vector<u64> grp_kmer, heavy_kmer; std::vector<u32> grp_rev, grp_fwd, heavy_rev, heavy_fwd;
    // SNPmer table
    u32 k = 0;
    u64* snp_keys = nullptr; u32* snp_vals = nullptr; u32 snp_mask = 0; u64* d_hf = nullptr; u32 n_hf = 0;
    u32 n_sites = 0, words = 0;
    u32* snp_occ = nullptr; u32 snp_occ_mask = 0;   // K3 (rank-table kernel): one bit per slot (mod snp_occ_mask + 1, <= 2^18 bits = 32 KB of LDS) of the SNPmer table: set = occupied.  A probe whose first slot's bit is clear is a miss
    u16* d_rank = nullptr; u32 rank_s = 0;          // K3: rank of mm_hash64(canonical s-mer) among the 4^s forward s-mers, s = rank_s <= 7 (built on first use, kept until s changes)
    std::vector<u32> site_order;   // internal bit position -> caller's site index
    double* d_ptable = nullptr;   // 256 entries: 10^(-x/10)
    // scratch
    void* scratch = nullptr; size_t scratch_bytes = 0;
    // pinned host staging for the many small calls of the greedy stages (one DMA each way instead of a staged copy per array)
    void* pin = nullptr; size_t pin_bytes = 0;
    void* zc = nullptr; size_t zc_bytes = 0;      // zero-copy I/O of small calls
    // multi-GPU tile sharding (svt_set_shard)
    u32 sh_rank = 0, sh_world = 1; int (*sh_fn)(void*, void*, uint64_t, const uint64_t*) = nullptr; void* sh_user = nullptr;
    void* sh_comm = nullptr;                  // svt_set_shard_comm: an RCCL communicator (ncclComm_t) the library owns; the exchanges are grouped broadcasts on `stream`
    hipEvent_t sh_mark = nullptr; bool sh_mark_set = false;   // recorded in front of the first collective since the last completed wait: that wait's deadline runs from the completion of this event
    bool sh_inflight = false;                 // a grouped collective has been enqueued on `stream` since the last completed wait: that wait has a deadline
    bool sh_failed = false; std::string sh_fail_why;   // the communicator was aborted (svt_shard_abort, a timed-out or failed collective): every exchange fails until a new one is set
    bool sh_paused = false;                   // svt_shard_pause: the hook stays, the tile slicing is off
    int sh_depth = 0;                         // inside a ShardGroup (one grouped collective for several arrays)
    u64 sh_calls = 0, sh_bytes = 0;           // exchanges made / bytes they covered (svt_get_option "shard_exchanges", "shard_bytes")
    // pinned staging of the packed copies (UpPack / DownPack, capi.hip): one buffer per direction, busy until the next stream sync
    void* pk[2] = {nullptr, nullptr}; size_t pk_bytes[2] = {0, 0}; bool pk_busy[2] = {false, false};
    u64 poa_clusters = 0, poa_handed_back = 0, poa_cons_device = 0;                  // K12: clusters launched / clusters the kernel ended with a status (the caller's host engine redoes them) (svt_get_option)
    u64 k8a_packed = 0, k8a_redo = 0;         // K8a: pairs sent through the packed 16-bit cell / of those, pairs it gave no certificate for (rerun through the 32-bit cell) (svt_get_option)
    u64 k9_pairs = 0, k9_again_pairs = 0, k9_redo_pairs = 0;   // K9 windowed slab: pairs walked / walked again around the end diagonal / with the full slab (svt_get_option)
    // profiling
    bool prof = false; int prof_level = 1; std::vector<ProfEntry> prof_entries; std::vector<PendingEvt> pending; std::vector<hipEvent_t> prof_events;   // prof_level 2: only the kernels a roofline is quoted for; prof_events: events to reuse
    // K12 (svt_poa_graphs): where the compacted final graphs of the last run sit inside the scratch buffer, until svt_poa_graphs_fetch
    struct { bool valid = false, pending = false; u64 n_nodes = 0, n_edges = 0; size_t off_code = 0, off_al = 0, off_edge = 0, off_cons = 0; std::vector<u64> cons_slot; size_t off_jobs = 0, off_outs = 0, off_noff = 0, off_eoff = 0, off_arena = 0; u32 n_clusters = 0; int C = 1; } poa_last;
    // forks (svt_fork): contexts of other host threads that share this context's read-only tables
    svt_ctx* parent = nullptr; std::vector<svt_ctx*> forks;
    u32 func_attr_done = 0;                   // dynamic-LDS attributes (hipFuncSetAttribute is per DEVICE) this context has set: one bit per kernel, see DYN_LDS_ONCE -- a context belongs to one device and one thread at a time
    bool profiling() const { return parent ? parent->prof : prof; }
    SnpTable snp_table() const { return SnpTable{snp_keys, snp_vals, snp_mask, d_hf, n_hf, n_sites, words}; }
};

// error helpers -----------------------------------------------------------------------------------
int svt_fail(svt_ctx* c, int code, const std::string& msg);
#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return svt_fail((ctx), SVT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
// the dynamic-LDS ceiling of a kernel, set once per CONTEXT (not once per process: the attribute belongs to the device, and contexts of several devices and host threads run this code)
#define DYN_LDS_ONCE(ctx, bit, func, bytes)                                                                    \
    do {                                                                                                       \
        if (!((ctx)->func_attr_done & (1u << (bit)))) {                                                        \
            HIPCHK((ctx), hipFuncSetAttribute((const void*)(func), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            (ctx)->func_attr_done |= 1u << (bit);                                                             \
        }                                                                                                      \
    } while (0)

// profiling scope: records HIP events around the launches issued while it lives
struct ProfScope {
    svt_ctx* c; int idx = -1; hipEvent_t a = nullptr, b = nullptr; hipStream_t st = nullptr;
    ProfScope(svt_ctx* ctx, const char* name, double bytes, double units, hipStream_t on = nullptr);   // on: the stream of the launch when it is not the context's own (side streams of K8a)
    ~ProfScope();
};
void prof_add_bytes(svt_ctx* c, const char* name, double bytes);
void prof_add_units(svt_ctx* c, const char* name, double units);
void prof_note_units(svt_ctx* c, const char* name, double units);   // output bytes known only after the launch (emitted list entries)
void* svt_scratch(svt_ctx* c, size_t bytes);   // grows a reusable device scratch buffer; nullptr on failure

// host-side launchers implemented in the .hip files -------------------------------------------------
int launch_pack(svt_ctx* c, svt_batch* b, const u8* d_ascii);
int launch_split_emit(svt_ctx* c, const svt_batch* b, u32 k, u8 min_bq, const u8* d_rc, const u64* d_out_off, u64* d_out, u32* d_cnt);
int launch_count_insert(svt_ctx* c, const svt_batch* b, u32 k, u8 min_bq, const u8* d_rc, u32* d_overflow);
int launch_ht_init(svt_ctx* c);
int launch_ht_merge(svt_ctx* c, const u64* d_k, const u32* d_r, const u32* d_f, u64 n);
int launch_ht_compact(svt_ctx* c, int mode /*0 filter,1 single_strand filter,2 all*/, u64* d_k, u32* d_r, u32* d_f, ull* d_counters /*[2]: distinct, kept*/);
int launch_seeds(svt_ctx* c, svt_batch* b, u32 k, u32 cpar, u8 min_bq, int use_qual, u32 maxm, u32 maxs, u32 read_lo, u32 read_hi);
int launch_lsh_sets(svt_ctx* c, svt_batch* b, u32 np2, u32 read_lo, u32 read_hi);
int launch_qualbin_mean(svt_ctx* c, const svt_batch* b, const double* d_table, double* d_out);
int launch_snp_bits(svt_ctx* c, svt_batch* b, u32 read_lo, u32 read_hi);
int launch_set_intersect(svt_ctx* c, const svt_batch* A, const svt_batch* B, const u32* d_a, const u32* d_b, u64 n, u32* d_shared, u32* d_same);
int launch_lsh_candidates(svt_ctx* c, const svt_batch* B, const u32* d_q, u32 n_q, const u32* d_r, u32 n_ref, const u32* d_lim, u32 mode, u32 top_n, u32 cap, u32 capacity, u32* d_cursor, u32* d_cnt, u32* d_off, u32* d_out);
int launch_gather_cols_t(svt_ctx* c, const u64* srcP, const u64* srcA, const u32* d_idx, u32 n, u32 words, ulonglong2* dstPA);
int launch_compat_lists(svt_ctx* c, const SeedsDev& rows, int row_view, const u32* d_row_idx, u32 n_rows, const ulonglong2* colPA, u32 n_cols, u32 words,
                        int filter, int triangular, u32 tri_base, const u32* d_row_max_x, u32* o_row, u32* o_col, u32* o_mm, u64 cap, ull* d_counter);
int launch_compat_lists_cs(svt_ctx* c, const SeedsDev& rows, int row_view, const u32* d_row_idx, u32 n_rows, const SeedsDev& cols, int col_view, const u32* d_col_idx, u32 n_cols,
                           u32 words, int filter, int triangular, u32 tri_base, const u32* d_row_max_x, u32* o_row, u32* o_col, u32* o_mm, u64 cap, ull* d_counter,
                           u32 col_lo = 0, u32* d_row_has = nullptr, const u32* d_sel_list = nullptr, const u32* d_sel_count = nullptr);
size_t seg_desc_bytes(); size_t seg_tile_bytes(); int compat_seg_rt(u32 words);
int launch_compat_lists_seg(svt_ctx* c, const SeedsDev& rows, int row_view, const u32* d_row_idx, u32 n_rows, const void* d_tiles, u32 n_tiles, const void* d_segs, u32 max_reps,
                            const SeedsDev& cols, int col_view, const u32* d_col_idx, u32 n_cols, u32 words, int filter, int phase,
                            u32* o_row, u32* o_col, u32* o_mm, u64 cap, ull* d_counter, u32* d_row_has, const u32* d_sel, const u32* d_sel_count);
int launch_unflagged_cols_seg(svt_ctx* c, const u32* d_flags, const u32* d_row_seg, const void* d_segs, u32 n, u32* d_sel, u32* d_count);
int launch_rec_row_count(svt_ctx* c, const u32* d_rec, const ull* d_count, u64 n_host, u64 cap, u32* d_row_cnt);   // records per row; d_count: the producing launch's counter (clamped to cap), else n_host records
int launch_rec_fill(svt_ctx* c, const u32* d_rec, u64 n, const u32* d_off, u32* d_cur, u32* d_col, u32* d_mm);       // every record into its row's slots (d_cur zeroed)
int launch_unflagged_cols(svt_ctx* c, const u32* d_flags, u32 n, u32 tri_base, u32* d_sel, u32* d_count);
int launch_best_column(svt_ctx* c, const SeedsDev& rows, int row_view, const u32* d_row_idx, u32 n_rows, const ulonglong2* colPA, u32 n_cols, u32 words,
                       const u32* d_lo, const u32* d_hi, u32* best_col, u32* best_score);
size_t consensus_counter_bytes(u32 n_clusters, u32 words);
int launch_consensus(svt_ctx* c, const SeedsDev& rows, const u64* d_cl_off, const u32* d_members, u32 n_clusters, u64 n_members, u32 words, u64* d_p, u64* d_a,
                     ull* d_counters, u64 max_cluster);
int launch_align(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band,
                 const u32* d_sel, u64 n_sel, int rclass, int32_t* d_nm, u32 max_qlen, u32 max_tlen, double algo_bytes, double cells);
#define SVT_MAX_SNPMER_SITES (1u << 20)   // bit positions of the SNPmer rows: 16384 64-bit words; the LDS-tiled Stage-3 kernels take <= 1200 words and hand wider sets to the dense-column kernels (SVT_ERR_TOOWIDE)
#define AFF_NCLS 18
static const int AFF_P[AFF_NCLS] = {8, 10, 12, 14, 16, 18, 20, 6, 8, 10, 12, 14, 16, 10, 12, 16, 16, 16};   // diagonals per lane and pairs per wavefront of the K8a band classes (kernels_affine.hip)
static const int AFF_G[AFF_NCLS] = {16, 16, 16, 16, 16, 16, 16, 8, 8, 8, 8, 8, 8, 4, 4, 4, 2, 1};
#define AFF16_NCLS 10
static const int AFF16_P[AFF16_NCLS] = {8, 10, 12, 14, 16, 18, 20, 12, 14, 16};   // diagonals per lane and lanes per pair of the packed-cell classes of K8a (kernels_affine.hip: aff16_pairs): class id = AFF_NCLS + index;
static const int AFF16_LG[AFF16_NCLS] = {4, 4, 4, 4, 4, 4, 4, 8, 8, 8};            // 128 / LG pairs per wavefront, bands <= LG P / 2 - 1 (ascending: the first class that holds a band carries the fewest diagonals)
int affine16_class_of(u32 w);
#define AFF_LDS_BUDGET ((size_t)20 * 1024)   // LDS bytes a wave may take for its pairs' sequences: sixteen 1.5 kb pairs take 12 KB, sixteen 4.3 kb pairs 35 KB (-> eight per wave)
int affine_class_of(u32 w, u32 lds_words, int max_g);
double affine_task_cost(int cls, u32 steps);
const char* affine_class_name(int cls);
int launch_align_affine_queue(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band, const u32* d_sel,
                              const void* d_tasks, u32 n_tasks, u32* d_counter, int max_g, int32_t* d_nm, int32_t* d_score, u32 max_qlen, u32 max_tlen, double algo_bytes, double cells,
                              u32* d_redo, u64 n_packed, u32 n_packed_tasks);
int launch_align_affine(svt_ctx* c, hipStream_t on, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band,
                        const u32* d_sel, u64 n_sel, int cls, int32_t* d_nm, int32_t* d_score, u32 max_qlen, u32 max_tlen, double algo_bytes, double cells);
int launch_align_bp(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band,
                    const u32* d_sel, u64 n_sel, int rclass, int32_t* d_nm, double algo_bytes, double cells);
int launch_table_sort(svt_ctx* c, u32 k, u64 n, const u64* km, const u32* rv, const u32* fw, u64* okm, u32* orv, u32* ofw,
                      u64* key_a, u64* key_b, u32* idx_a, u32* idx_b, void* temp, size_t temp_bytes, size_t* need_bytes);
int launch_table_select(svt_ctx* c, u32 k, u64 n, const u64* km, const u32* rv, const u32* fw, u8* fl_grp, u8* fl_heavy, u32* out_grp, u32* out_heavy, u32* d_counts,
                        void* temp, size_t temp_bytes, size_t* need_bytes);
int launch_twin_order(svt_ctx* c, const svt_batch* b, u32 min_len, u32 max_len, u32 cpar, double cutoff, u8* d_flag, u64* d_key_all, u32* d_idx_a, u32* d_idx_b, u64* d_key_a, u64* d_key_b,
                      u32* d_count, u32 n_kept_known, void* temp, size_t temp_bytes, size_t* need_bytes, int phase);
int launch_twin_gather(svt_ctx* c, const svt_batch* b, const u32* d_order, u32 n, u32* o_len, u32* o_nmini, u32* o_nuniq, u32* o_nsnpf, double* o_est, u8* o_ev, u8* o_lv, u64* o_lsh);
int launch_table_gather(svt_ctx* c, const u32* idx, u64 n, const u64* km, const u32* rv, const u32* fw, u64* okm, u32* orv, u32* ofw);
u64 align_tb_dwords(int rclass, u32 max_qlen, u32 max_tlen);
u64 align_tb_dwords_bp(int rclass, u32 max_tlen, bool full, int win_bits = 64);
int launch_align_tb_bp(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band,
                       const u32* d_sel, u64 n_sel, int rclass, int32_t* d_nm, u32 max_tlen, u32* d_tb, u64* d_cells, const u64* d_cell_off, u32* d_span,
                       int mode, u64* d_keys, u32* d_redo, const u32* d_remap, double band_cells = 0.0, int win_bits = 64);
int launch_tie_passes(svt_ctx* c, const u32* d_row_idx, const u32* o_row, const u32* o_col, const u32* o_mm, u64 n, u32* a_idx, int phase,
                      const u32* shared, const u32* same, const u32* r_unique, const u32* a_unique, double min_frac, double cpar,
                      u32* lowest, u8* keep, u32* t_row, u32* t_col, u8* t_rev, u32* t_mm, u64 cap, ull* counter, u8* done);
int launch_candidate_select(svt_ctx* c, const u32* o_row, const u32* o_col, const u32* o_mm, u64 n, u32* rmin, const u8* done, int mode,
                            u32* s_row, u32* s_col, u32* s_mm, ull* counter);
u32 poa_graph_stride(int C);
u64 poa_graph_arena_bytes(u32 ncap, u32 ecap, u32 lmax, int C);
size_t poa_graph_job_bytes();
size_t poa_graph_out_bytes();
int poa_graph_max_band(int C);
int launch_poa_graph(svt_ctx* c, int C, u32 n_clusters, u32 lmax, const void* d_jobs, u8* d_arenas, const u8* d_seqs, const u8* d_wts, const u64* d_seq_off, const u32* d_band, void* d_outs, double cells);
int launch_poa_consensus(svt_ctx* c, int C, u32 n_clusters, const void* d_jobs, const u8* d_arenas, void* d_outs, u8* d_cons);
int launch_poa_gather(svt_ctx* c, const svt_batch* B, const u32* d_read_idx, const u8* d_rev, const u64* d_seq_off, u32 n_seqs, u8* d_seq, u8* d_wts, double bytes);
int launch_poa_graph_export(svt_ctx* c, int C, u32 n_clusters, const void* d_jobs, const u8* d_arenas, const void* d_outs, const u64* d_node_off, const u64* d_edge_off, u8* o_code, u16* o_al, u32* o_edge);
int launch_pileup_stats(svt_ctx* c, const svt_batch* Q, const u64* d_cells, const u64* d_cell_off, const u32* d_pair_q, const u64* d_grp_off, const u64* d_col_off,
                        const u8* d_grp_sel, const void* d_tiles, u32 n_tiles, u64 n_cells, u32* d_depth, u32* d_err, ull* d_total, ull* d_errs);
int launch_pileup_hp_median(svt_ctx* c, const u64* d_cells, const u64* d_cell_off, const u64* d_grp_off, const u64* d_col_off, const void* d_tiles, u32 n_tiles, u64 n_cells, u8* d_out);
int launch_pileup_loglik(svt_ctx* c, const svt_batch* Q, const u64* d_cells, const u64* d_cell_off, const u32* d_pair_q, const u64* d_grp_off, const u64* d_col_off,
                         const void* d_tiles, u32 n_tiles, u64 n_cells, const double* d_tab, double indel_lr, double indel_ln, double* d_lr, double* d_ln);
int launch_align_tb(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const u32* d_q, const u32* d_t, const u8* d_rev, const u32* d_band,
                    const u32* d_sel, u64 n_sel, int rclass, int32_t* d_nm, u32 max_qlen, u32 max_tlen, u32* d_tb, u64* d_cells, const u64* d_cell_off, u32* d_span);
int launch_csr_gather(svt_ctx* c, const svt_batch* b, int which, const u64* d_dst_off, u32* d_pos, u64* d_kmer, u8* d_flags);

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ u64 d_mm_hash64(u64 key) {            // src/seeding.rs:18-28
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}
__device__ __forceinline__ u64 d_fx_word(u64 h, u64 w) {         // fxhash 0.2.1 (types.rs:733-736 call site)
    return (((h << 5) | (h >> 59)) ^ w) * 0x517cc1b727220a95ull;
}
__host__ __device__ __forceinline__ u32 snp_slot_hash(u64 km) {  // slot hash of the SNPmer table (not a reference hash)
    u32 x = (u32)km ^ ((u32)(km >> 32) * 0x9E3779B1u);
    x *= 0x85EBCA6Bu;
    x ^= x >> 15;
    return x;
}
// reverse the order of the 32 two-bit groups of x
__device__ __forceinline__ u64 d_revpairs64(u64 x) {
    x = __brevll(x);
    return ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
}
__device__ __forceinline__ u64 d_revcomp(u64 kmer, u32 k) {      // reverse complement of a 2k-bit k-mer
    return d_revpairs64(~kmer) >> (64 - 2 * k);
}
// 32 bases starting at base p (p >= 0) of a packed read, MSB-first.  Reads words p/16 .. p/16+2
// (the SVT_PAD_WORDS zero words make that safe for every p < len).
__device__ __forceinline__ u64 d_window64(const u32* w, u32 p) {
    u32 a = p >> 4, o = (p & 15) * 2;
    u64 A = ((u64)w[a] << 32) | w[a + 1];
    if (o == 0) return A;
    return (A << o) | ((u64)w[a + 2] >> (32 - o));
}
// k mask bits (1 per base) starting at base p, returned as a 2k-bit value with BOTH bits set per masked base
__device__ __forceinline__ u64 d_nmask_kmer(const u16* m, u32 p, u32 k) {
    u32 a = p >> 4, o = p & 15;
    u64 A = ((u64)m[a] << 32) | ((u64)m[a + 1] << 16) | (u64)m[a + 2];   // 48 mask bits, MSB-first
    u64 bits = (A >> (48 - o - k)) & ((1ull << k) - 1);                  // k bits, first base highest
    // spread: bit j -> bits 2j,2j+1
    u64 x = bits;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x | (x << 1);
}
__device__ __forceinline__ u32 d_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ u32 d_rank(ull mask) {                // number of set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0));
}
__device__ __forceinline__ u8 d_qual_bin(u8 b) {                 // src/types.rs:447-467
    if (b <= 34) return 0;
    if (b >= 77) return 15;
    return (u8)((b - 35) / 3 + 1);
}
#endif
