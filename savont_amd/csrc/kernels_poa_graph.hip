// kernels_poa_graph.hip -- K12: Stage-4a POA (src/alignment.rs:193-231, generate_consensus_poa) with the partial-order graphs
// RESIDENT on the device: one workgroup (8 waves) owns one cluster for all of its reads -- alignment, traceback, fusing the path
// into the graph and keeping a topological order all happen inside ONE launch for all clusters.  The inputs are NAMED, not copied (k_poa_gather takes letters and
// weights from the resident 2-bit reads and quality bins), and the consensus of every finished graph is walked on the device (k_poa_consensus); the graphs
// themselves (nodes, aligned sets, weighted edges) are exported only when the caller asks for them or a consensus has to be redone on the host.
//
// It computes what savont_amd/csrc/host/poa.hpp (its CPU twin) and oracle/poa_oracle.py compute, alignment by alignment:
//   cell(i, j) = max over predecessor rows p (in in-edge order) of {cell(p, j-1) + (code == seq[j-1] ? 3 : -8), cell(p, j) - 6}
//   and cell(i, j-1) - 6; row 0 / column 0 free (overlap mode); band = band_base + (int)(0.1 L) + 1 columns around the rounded
//   mean position of the bases fused into the node; end cell = first maximum over sink rows and column L; traceback prefers
//   (mis)match, then deletion, then insertion.
// What is organised differently from spoa's bookkeeping (tools/poa_order_model.py checks each claim against the oracle):
//   * ORDER.  spoa re-sorts the whole graph depth-first after every read.  The DP needs only SOME topological order, so the
//     kernel keeps one in which every aligned set is a contiguous block and splices a read's new nodes in behind the block of
//     the path node that precedes them (new edges between old nodes only shortcut existing paths, so the old order stays valid).
//     The only order-dependent choice of the algorithm is the end cell among EQUAL maxima in different rows (spoa: the first in
//     its own order).  When the first tied row (this order) is an ancestor of all other tied rows -- the homopolymer case, 12 % of
//     the reads -- every topological order agrees; any other tie ends the cluster with a status and the host DP redoes it.
//   * FUSE is position-parallel: every sequence position decides on its own (same node / aligned sibling / new node); new ids
//     come from a prefix sum in the host's creation order (unaligned prefix, unaligned suffix, then path order); the edges are
//     (node of p-1 -> node of p) for every p.  Same graph as the serial add_alignment, in-edge and aligned-list order included.
//   * DP: three engines share the graph code (template parameter ENG); all give the same cells, back-pointers and end cell.
//     ENG 2, THE DEFAULT (round 4; svt option poa_rows = 2): anti-diagonal sweeps with lane = graph row.  A wave takes a block of 64 consecutive rows and
//       steps it one anti-diagonal (row + column) at a time; the waves of a workgroup run behind one another on the blocks (block b + 1 reads anti-diagonal A - 1
//       of block b: progress words in LDS, no barrier).  The cell to the left is the lane's own last value, the row before arrives by one DPP move, any other
//       predecessor inside the block by one LDS read per step from the RING (the last TR anti-diagonals of each of the wave's 64 rows), predecessors in the
//       block before from that block's EXPORT rows, anything farther from the row's copy in HBM (rows known before the DP starts; `far_rows`).  A candidate
//       carries its traceback priority in its low six bits, so ONE maximum is value and back-pointer; four anti-diagonals per loop trip, one back-pointer
//       dword per lane and trip.  0.43 us per graph row for the DP on an idle chip (profiles/r04_poa.md): a lone wavefront issues one instruction per ~3 ns
//       whatever it depends on, and a step is ~31 vector instructions; the floor of a step that does nothing but the recurrence is 0.15 us per row.
//     ENG 1 (poa_rows = 1): the row engine -- one wave per cluster, one graph row per step, the band's columns on the lanes (C per lane), the insertion chain by a
//       prefix maximum of five DPP steps; for bands that fit 64 C columns.  0.9 us per row.
//     ENG 0 (poa_rows = 0): round 3's chunk pipeline -- columns cut into chunks of 64 C absolute columns, wave w owns the chunks k = w (mod 8), the waves run
//       behind one another over the rows (LDS ring of the last R rows per wave); a (row, chunk) task is ~200 instructions.  0.9 us per row; kept as the engine
//       every band fits (the anti-diagonal engine hands a cluster back with a status when a block's sweep outgrows its export area).
//     Back-pointers (move + predecessor ordinal, 1 byte per cell) go to HBM; the traceback is one wave walking them, a RUN of diagonal moves per ballot.
// Bound: NOT HBM and not MFMA -- the dependent chain of graph rows of ONE cluster (every read is fused into the graph before the next aligns).  A 75-read x 1.5 kb
// cluster is ~146 k rows: 60 ms of DP + 11 ms of traceback + 3 ms of bookkeeping with the default engine (163 ms in round 3), against 9 ms per cluster on ONE
// AVX-512 core (poa.hpp) -- but the launch holds 105 of the chip's ~8000 wave slots and no host core.  Who runs the POA is the caller's choice
// (svh_set_option "poa_engine": host DP, K12, or a split; by default K12 when the process has <= 10 worker threads); the choice is logged once per pipeline.
// HBM traffic = 1 back-pointer byte per band cell (+ the gathered inputs, 2 bytes per base).
#include "svt_internal.hpp"
#include <type_traits>

namespace {

constexpr int PW = 8;                 // waves per workgroup
constexpr int PNT = PW * 64;          // threads
constexpr int PAL = 6;                // aligned siblings kept per node
constexpr int PSPILL = 512;           // spill rows per cluster
constexpr int PTIE = 32;              // tied end rows examined
constexpr int PMETA = 12;             // dwords per row descriptor
constexpr int PWFD = 12;              // dwords per row descriptor of the anti-diagonal engine
constexpr u32 PNIL = 0xFFFFFFFFu;
constexpr int PNEG = -30000;
constexpr int SM = 3, SX = -8, SG = -6;

struct PoaGJob {                      // one cluster
    u64 arena;                        // byte offset of the cluster's arena
    u64 seq_first;                    // index of its first sequence in seq_off / seq_band
    u32 n_seqs, ncap, ecap, lmax;
};
struct PoaGOut { int32_t status; u32 n_nodes, n_edges, ties, rows_done, tie_reads, far_rows, pad; u64 ticks[6]; u32 spins[PW], tasks[PW]; };   // per wave: polls that found the neighbours not ready, chunk tasks done.  ticks: 100 MHz clock spent in row descriptors, DP, end cell, traceback, fuse, order

// arena layout: every array starts on a 16-byte boundary; sizes are functions of (ncap, ecap, lmax, stride)
struct PoaLay {
    u64 na, nb, nc, nd, rowof, ranka, rankb, meta, endval, spillreq, ea, enin, plist, alnrow, cur, kind, anchor, ncnt, wfd, expreq, blkexp, blka0, D, spill, total;
};
__host__ __device__ inline u64 al16(u64 x) { return (x + 15) & ~(u64)15; }
__host__ __device__ inline PoaLay poa_layout(u32 ncap, u32 ecap, u32 lmax, u32 stride) {
    PoaLay l; u64 o = 0;
    const u64 nr = (u64)ncap + 1;
    l.na = o; o = al16(o + 16ull * ncap);          // {pos_sum, pos_n, in_cnt, out_cnt}
    l.nb = o; o = al16(o + 16ull * ncap);          // tails of the first four in-edges
    l.nc = o; o = al16(o + 16ull * ncap);          // {in_head, in_tail, out_head, out_tail} edge ids
    l.nd = o; o = al16(o + 16ull * ncap);          // {u16 aligned[6], u16 al_cnt, u8 code, u8 pad}
    l.rowof = o; o = al16(o + 4ull * ncap);
    l.ranka = o; o = al16(o + 4ull * ncap);
    l.rankb = o; o = al16(o + 4ull * ncap);
    l.meta = o; o = al16(o + 4ull * PMETA * nr);
    l.endval = o; o = al16(o + 2ull * nr);
    l.spillreq = o; o = al16(o + nr);
    l.ea = o; o = al16(o + 16ull * ecap);          // {tail, head, weight, next_out}
    l.enin = o; o = al16(o + 4ull * ecap);
    l.plist = o; o = al16(o + 8ull * ecap);        // {row, lo | hi << 16} of rows with more than four predecessors
    l.alnrow = o; o = al16(o + 4ull * lmax);
    l.cur = o; o = al16(o + 4ull * lmax);
    l.kind = o; o = al16(o + lmax);
    l.anchor = o; o = al16(o + 4ull * lmax);
    l.ncnt = o; o = al16(o + 4ull * ((u64)lmax + 1));
    l.wfd = o; o = al16(o + 4ull * PWFD * nr);       // anti-diagonal engine: per-row descriptor (band, flags, predecessor slots)
    l.expreq = o; o = al16(o + nr);                  // ... rows a later 64-row block reads through the LDS export area
    l.blkexp = o; o = al16(o + 4ull * (nr / 64 + 2)); // ... export slots handed out per block
    l.blka0 = o; o = al16(o + 4ull * (nr / 64 + 2));  // ... first anti-diagonal of every block's sweep (the traceback finds a cell's back-pointer through it)
    // back-pointers, one byte per band cell.  Row engine / chunk pipeline: row-major, `stride` per row.  Anti-diagonal engine (round 6): TRIP-major -- block b of 64 rows owns
    // TRX * 64 bytes, [trip][lane] eight bytes each, so that the store of a trip is one contiguous 512-byte piece instead of eight bytes in each of 64 rows (TRX = the longest
    // sweep of a block: 1024 anti-diagonals, 2048 for the widest class = stride / 512 == 4)
    l.D = o; o = al16(o + std::max<u64>((u64)stride * nr, (u64)(stride >= 2048 ? 2048 : 1024) * 64 * (nr / 64 + 2)));
    l.spill = o; o = al16(o + 2ull * stride * PSPILL);
    l.total = o;
    return l;
}

struct NodeD { u16 al[PAL]; u16 alcnt; u8 code; u8 pad; };
static_assert(sizeof(NodeD) == 16, "NodeD is one 16-byte record");

__device__ __forceinline__ int col_of(u32 pos_sum, u32 pos_n) {                 // PoaGraph::col_of: the rounded mean position
    return pos_n ? (int)((2u * pos_sum + pos_n) / (2u * pos_n)) : 1;
}
// A value that came from a global load, handed on through a move: the compiler's wait-count bookkeeping then sees a plain VALU result.
// Without it every join of the rare HBM paths (spill rows, long predecessor lists) with the common path carried an s_waitcnt vmcnt(0),
// i.e. every row waited for the previous row's back-pointer store to reach HBM.
__device__ __forceinline__ int launder(int x) { asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(x)); return x; }
__device__ __forceinline__ int wave_prefix_max(int v, const int ident) {
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xF, 0xF, false));   // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xF, 0xF, false));   // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xF, 0xF, false));   // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xF, 0xF, false));   // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xA, 0xF, false));   // row_bcast:15
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xC, 0xF, false));   // row_bcast:31
    return v;
}
__device__ __forceinline__ u32 wave_incl_sum(u32 v) {
    #pragma unroll
    for (int s = 1; s < 64; s <<= 1) { const u32 o = __shfl_up(v, s); if ((int)(threadIdx.x & 63) >= s) v += o; }
    return v;
}
__device__ __forceinline__ u32 wave_incl_max(u32 v) {
    #pragma unroll
    for (int s = 1; s < 64; s <<= 1) { const u32 o = __shfl_up(v, s); if ((int)(threadIdx.x & 63) >= s) v = max(v, o); }
    return v;
}
// block-wide exclusive scans of one value per thread (512 threads); `tmp` holds 8 words
__device__ __forceinline__ u32 block_excl_sum(u32 v, volatile u32* tmp, u32* total) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 inc = wave_incl_sum(v);
    __syncthreads();
    if (lane == 63) tmp[w] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
    #pragma unroll
    for (int x = 0; x < PW; x++) { const u32 t = tmp[x]; if (x < w) base += t; tot += t; }
    *total = tot;
    return base + inc - v;
}
__device__ __forceinline__ u32 block_excl_max(u32 v, volatile u32* tmp) {         // running maximum of the values of the threads before this one (0 if none)
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 inc = wave_incl_max(v);
    __syncthreads();
    if (lane == 63) tmp[w] = inc;
    __syncthreads();
    u32 base = 0;
    #pragma unroll
    for (int x = 0; x < PW; x++) { const u32 t = tmp[x]; if (x < w) base = max(base, t); }
    u32 prev = __shfl_up(inc, 1); if (lane == 0) prev = 0;
    return max(base, prev);
}

__host__ __device__ inline u32 job_lmax_pad(u32 lmax) { return (lmax + 31) & ~15u; }
template <int C> struct PCfg {
    static constexpr int CW = 64 * C;                     // columns per chunk
    static constexpr int R = (C == 1) ? 32 : 16;          // ring rows per wave
    static constexpr int DMAX = R / 2;                    // predecessors closer than this come from the ring
    static constexpr int STRIDE = PW * CW;                // cells per stored row (back-pointers, spill rows)
    static constexpr int CSH = (C == 1) ? 6 : (C == 2 ? 7 : 8);
};

// LDS of the anti-diagonal engine (ENG = 2): per wave that takes blocks, an EXPORT area (ES rows of the block a later block reads: TRX anti-diagonals each)
// and a RING (the last TR anti-diagonals of each of its 64 rows; lane l keeps anti-diagonal a in slot (a + l) mod TR of its 64-byte row, so that the
// 64 stores of a step fall into different banks); 64 x 8 dump cells (ONE area for all waves: write-only) for the export stores of rows nobody reads; two constant cells (0 and the floor).
// C = 2: 48 + 24 + 1 KB + the sequence + 4 KB of static LDS = ~79 KB -- TWO workgroups per CU (128 registers x 8 waves each: the register file holds exactly two)
template <int C> struct WfCfg {
    static constexpr int NBW = (C == 1) ? 5 : (C == 2 ? 6 : 8);   // waves that take blocks (a block's sweep lasts ~(64 + drift + band) steps, a new one starts every ~(64 + drift))
    static constexpr int TRX = (C == 4) ? 2048 : 1024;      // >= the sweep of a block + the drift to the next block (rows and columns both advance: up to 128 per block) for the widest band of the class
    static constexpr int ES = 8192 / (TRX * 2);             // export rows per block: 8 KB per wave
    static constexpr int TR = 32;
    static constexpr u32 EXP = 0;                           // [NBW][ES][TRX] int16
    static constexpr u32 RING = EXP + (u32)NBW * ES * TRX * 2;   // [NBW][64][TR] int16
    static constexpr u32 DUMP = RING + (u32)NBW * 64 * TR * 2;   // [64][8] int16: a lane whose row nobody reads stores its eight cells of a trip here -- ONE area for all waves (nobody reads it; per wave it was 5 KB more, and with them a workgroup's LDS passed 80 KB: one workgroup per CU instead of two)
    static constexpr u32 CST = DUMP + 1024u;                     // {0, PNEG}
    static constexpr u32 SQ = CST + 16;
};

// two workgroups per CU for the default class (160 KB of LDS per CU, 4 KB of static LDS per workgroup with the alignment of the dynamic area): holds for sequences up to 2016 bases.
// (The intent since round 4; the areas added later had silently grown a workgroup past 80 KB until round 6 -- hence the assertion.)
static_assert(WfCfg<2>::SQ + 2048 + 256 + 4096 <= 80 * 1024, "k_poa_graph<2, 2>: a workgroup's LDS must leave room for a second one on the CU");
struct PoaShared {
    int blk_a0[PW], blk_a1[PW];            // anti-diagonal engine: first and last anti-diagonal of the block a wave is sweeping
    int done[PW];                          // rows finished per wave: relaxed workgroup-scope atomics (plain ds_read / ds_write; a volatile member became a flat load)
    u32 scan[PW];
    unsigned long long red[PW];
    int tiered[PW];
    u32 spill_cnt, plist_cnt, tie_cnt, status;
    u32 tie_rows[PTIE];
    int best_v, best_i, best_j, has_aln, fp, lp, tie;
    u32 n_nodes, n_edges, n_rows, total_new, in_total;
};

// ENG = 0: the chunk pipeline over eight waves described above (C = 1, 2, 4 cells per lane and chunk).
// ENG = 1 (round 4), the ROW ENGINE: ONE wave computes a whole graph row per step.  Lane l holds the C cells of the columns j with j mod W in
// [l C, l C + C), W = 64 C >= band width + C: ABSOLUTE columns modulo W, so a cell of every row -- the row before (kept in registers), an older row
// (LDS ring of the last 64 rows) or a far one (its copy in HBM) -- sits in the same lane and register slot as the cell above it; the diagonal
// neighbour of a lane's first cell is the last cell of the lane before it, one DPP wave rotation.  No cross-wave hand-off, no polling, no
// descriptor ping-pong: a row is ~25 dependent steps of ~6-way independent work instead of a 200-instruction chain per (row, chunk).  The insertion
// chain is a prefix maximum over the lanes in ROTATED order (the band starts in lane (lo / C) mod 64 and wraps): both arcs are scanned at once as the
// two int16 halves of one register (v_pk_max_i16 under six DPP steps).  Back-pointers are not stored at all: the rows' VALUES go to HBM (2 bytes per
// cell), and the traceback recomputes the moves of the 16 cells around the path for 64 rows at a time, lane = row, before walking them on the scalar unit.
template <int C, int ENG>
__global__ __launch_bounds__(PNT, 4) void k_poa_graph(const PoaGJob* __restrict__ jobs, u8* __restrict__ arenas, const u8* __restrict__ seqs, const u8* __restrict__ wts,
                                                   const u64* __restrict__ seq_off, const u32* __restrict__ seq_band, PoaGOut* __restrict__ outs) {
    typedef PCfg<C> K;
    constexpr bool ROWS = ENG == 1;
    constexpr bool WF = ENG == 2;
    constexpr int W = 64 * C, VR = 64;                                          // ROWS: columns per stored row, rows in the LDS ring
    // WF, the anti-diagonal engine: lane = graph row, 64 consecutive rows per wave and block, one anti-diagonal (row + column) per step
    constexpr int NBW = WfCfg<C>::NBW, TRX = WfCfg<C>::TRX, ES = WfCfg<C>::ES, TR = WfCfg<C>::TR, KR = TR - 3;
    constexpr int CW = K::CW, R = K::R, DMAX = WF ? (1 << 30) : (ROWS ? VR : K::DMAX), STRIDE = ROWS ? 2 * W : K::STRIDE, CSH = K::CSH;
    constexpr int IDENT = -(1 << 29);
    constexpr int MININT = -2147483647 - 1;
    extern __shared__ __attribute__((aligned(4096))) u8 lds_raw[];                                    // WF: the export areas are addressed as (offset & mask) | base
    __shared__ PoaShared S;                                                                          // a static LDS object: its volatile members compile to ds_read / ds_write (behind a generic pointer they became flat loads)
    int16_t* ring = reinterpret_cast<int16_t*>(lds_raw);                                              // ENG 0: [PW][R][CW]; ROWS: [VR][W]; WF: see WfCfg
    u8* sq = WF ? lds_raw + WfCfg<C>::SQ : reinterpret_cast<u8*>(ring + (ROWS ? VR * W : PW * R * CW));   // [lmax]
    u32* msk = reinterpret_cast<u32*>(sq + ((job_lmax_pad(jobs[blockIdx.x].lmax))));                  // ROWS: [lmax / C + 2] per block of C columns, byte n = the columns whose base has index n
    const PoaGJob job = jobs[blockIdx.x];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const PoaLay lay = poa_layout(job.ncap, job.ecap, job.lmax, STRIDE);
    u8* A = arenas + job.arena;
    uint4* NA = (uint4*)(A + lay.na); uint4* NB = (uint4*)(A + lay.nb); uint4* NC = (uint4*)(A + lay.nc); NodeD* ND = (NodeD*)(A + lay.nd);
    u32* rowof = (u32*)(A + lay.rowof); u32* rank_cur = (u32*)(A + lay.ranka); u32* rank_nxt = (u32*)(A + lay.rankb);
    u32* meta = (u32*)(A + lay.meta); int16_t* endval = (int16_t*)(A + lay.endval); u8* spillreq = A + lay.spillreq;
    uint4* EA = (uint4*)(A + lay.ea); u32* enin = (u32*)(A + lay.enin); uint2* plist = (uint2*)(A + lay.plist);
    int32_t* alnrow = (int32_t*)(A + lay.alnrow); u32* curv = (u32*)(A + lay.cur); u8* kindv = A + lay.kind; u32* anchor = (u32*)(A + lay.anchor); u32* ncnt = (u32*)(A + lay.ncnt);
    u8* D = A + lay.D; int16_t* spillH = (int16_t*)(A + lay.spill);
    u32* wfd = (u32*)(A + lay.wfd); u8* expreq = A + lay.expreq; u32* blkexp = (u32*)(A + lay.blkexp); int* blka0 = (int*)(A + lay.blka0);

    if (tid == 0) { S.n_nodes = 0; S.n_edges = 0; S.n_rows = 0; S.status = 0; }
    u32 stat_ties = 0, stat_rows = 0, stat_far = 0, stat_spins = 0, stat_tasks = 0;
    u64 tk[6] = {0, 0, 0, 0, 0, 0}; u64 t_last = wall_clock64();
    #define PG_TICK(x) do { const u64 t_ = wall_clock64(); tk[x] += t_ - t_last; t_last = t_; } while (0)
    __syncthreads();

    for (u32 r = 0; r < job.n_seqs; r++) {
        const u64 sbase = seq_off[job.seq_first + r];
        const int L = (int)(seq_off[job.seq_first + r + 1] - sbase);
        if (L == 0) continue;                                                   // PoaGraph::add_alignment returns at once
        const int bw = (int)seq_band[job.seq_first + r];
        const int N = (int)S.n_rows;                                            // rows = nodes
        const u32 n0 = S.n_nodes;
        const u8* sg = seqs + sbase; const u8* wg = wts + sbase;
        for (int x = tid; x < L; x += PNT) sq[x] = sg[x];
        if constexpr (ROWS) {                                                       // block b = columns b C .. b C + C - 1; column j is scored against base j - 1
            for (int b = tid; b <= L / C + 1; b += PNT) {
                u32 mk = 0;
                #pragma unroll
                for (int c = 0; c < C; c++) { const int j = b * C + c; if (j >= 1 && j <= L) mk |= 1u << (8 * ((sg[j - 1] >> 1) & 3) + c); }
                msk[b] = mk;
            }
        }
        if (tid == 0) { S.spill_cnt = 0; S.plist_cnt = 0; S.tie_cnt = 0; S.has_aln = 0; S.fp = 0; S.lp = -1; S.tie = 0; }
        if (tid < PW) S.done[tid] = 0;
        // ---- A1: row of every node, clear the per-row flags
        for (int i = tid; i < N; i += PNT) { rowof[rank_cur[i]] = (u32)(i + 1); spillreq[i + 1] = 0; if constexpr (WF) expreq[i + 1] = 0; }
        if constexpr (WF) { for (int x = tid; x <= N / 64 + 1; x += PNT) blkexp[x] = 0; if (tid < PW) S.done[tid] = -1; }
        __syncthreads();
        // ---- A2: row descriptors
        for (int i0 = tid; i0 < N; i0 += PNT) {
            const int i = i0 + 1;
            const u32 nd = rank_cur[i0];
            const uint4 a = NA[nd]; const uint4 b = NB[nd];
            const int c = col_of(a.x, a.y);
            const int lo = min(L, max(0, c - bw)), hi = min(L, c + bw);
            const u32 np = a.z;
            u32 m[PMETA];
            #pragma unroll
            for (int x = 0; x < PMETA; x++) m[x] = 0;
            m[0] = (u32)lo | ((u32)hi << 16);
            m[1] = (u32)ND[nd].code | ((a.w == 0 ? 1u : 0u) << 8) | (np << 16);
            m[10] = (u32)c;
            const u32 tails[4] = {b.x, b.y, b.z, b.w};
            const int cb = (lo >> CSH) << CSH;                                  // first column of the row's first chunk
            u32 need_left = 0;                                                  // the latest predecessor row whose band reaches left of that column (the wave of the first chunk waits for its neighbour to have finished it)
            bool slow = np > 2;                                                 // the lean path of the DP takes rows with at most two predecessors, both inside the LDS ring
            bool partial = false;                                               // row engine: a predecessor's band covers the row's candidate range only in part
            #pragma unroll
            for (int o = 0; o < 4; o++) {
                if ((u32)o < np) {
                    const u32 t = tails[o]; const u32 pr = rowof[t]; const uint4 ta = NA[t];
                    const int tc = col_of(ta.x, ta.y);
                    const u32 plo = (u32)min(L, max(0, tc - bw)), phi = (u32)min(L, tc + bw);
                    m[2 + (o >> 1)] |= pr << (16 * (o & 1));
                    m[4 + o] = plo | (phi << 16);
                    if (i - (int)pr >= DMAX) { spillreq[pr] = 1; if (o < 2) slow = true; }
                    if ((int)plo < cb) need_left = max(need_left, pr);
                    if constexpr (ROWS) {
                        // the lean row step reads a predecessor's cells as they are stored (PNEG outside its band): right only if the predecessor's band covers
                        // the row's candidate range on both sides and the two bands together span at most W columns (no column aliases another modulo W)
                        const int j0r = max(lo, 1);
                        if (o < 2 && max(hi, (int)phi) - min(j0r - 1, (int)plo) + 1 > W) slow = true;              // a column of one band aliases a column of the other modulo W
                        if (o < 2 && ((int)plo > j0r || (int)phi + 1 < hi)) partial = true;                          // the predecessor's band does not cover the row's range: range masks
                    }
                }
            }
            if (ROWS && np == 0) slow = true;                                       // the virtual source row has no stored cells
            if (np == 0) m[4] = (u32)L << 16;                                   // the virtual source row: columns 0 .. L
            if (np < 2) { m[2] = (m[2] & 0xFFFFu) | (m[2] << 16); m[5] = m[4]; } // a second predecessor that repeats the first: the DP fetches both unconditionally
            if (np > 4) {                                                       // the full list, in in-edge order
                const u32 st = atomicAdd(&S.plist_cnt, np);
                m[8] = st;
                if (st + np <= job.ecap) {
                    u32 e = NC[nd].x;
                    for (u32 o = 0; o < np && e != PNIL; o++) {
                        const u32 t = EA[e].x; const u32 pr = rowof[t]; const uint4 ta = NA[t];
                        const int tc = col_of(ta.x, ta.y);
                        const int plo = min(L, max(0, tc - bw));
                        plist[st + o] = make_uint2(pr, (u32)plo | ((u32)min(L, tc + bw) << 16));
                        if (i - (int)pr >= DMAX) spillreq[pr] = 1;
                        if (plo < cb) need_left = max(need_left, pr);
                        e = enin[e];
                    }
                } else S.status = 5;
                if (np > 63) S.status = 9;                                      // the back-pointer holds 6 bits of predecessor ordinal
            }
            m[1] |= (slow ? 1u : 0u) << 10;
            if (ROWS) m[1] |= (partial ? 1u : 0u) << 11;
            m[11] = need_left;
            u32* mp = meta + (size_t)i * PMETA;
            #pragma unroll
            for (int x = 0; x < PMETA; x += 4) *reinterpret_cast<uint4*>(mp + x) = make_uint4(m[x], m[x + 1], m[x + 2], m[x + 3]);
        }
        __syncthreads();
        // ---- A4 (anti-diagonal engine): where a row finds each predecessor.  Row i is lane (i - 1) mod 64 of block (i - 1) / 64; its predecessor p = i - k is
        //   the lane before (k = 1): a DPP move; a row of the same block, k <= KR: that row's LDS ring; a row of the block before: the export area of the wave
        //   that swept it, if the block has a slot left (ES per block); anything else (and a fifth LDS predecessor): the row's copy in HBM ("far")
        auto wf_pred = [&](const u32* mp, const int o) -> int { return (int)(o < 4 ? ((mp[2 + (o >> 1)] >> (16 * (o & 1))) & 0xFFFFu) : plist[mp[8] + (u32)o].x); };
        if constexpr (WF) {
            for (int i = tid + 1; i <= N; i += PNT) {
                const u32* mp = meta + (size_t)i * PMETA;
                const int np = (int)(mp[1] >> 16);
                if (np > 15) { S.status = 9; continue; }                           // four bits of predecessor ordinal in a back-pointer
                const int b = (i - 1) >> 6, l = (i - 1) & 63;
                int cnt = 0;
                for (int o = 0; o < np; o++) {
                    const int p = wf_pred(mp, o), k = i - p, pb = (p - 1) >> 6;
                    if (k == 1 && l >= 1) continue;
                    if (pb == b && k <= KR && cnt < 4) { cnt++; continue; }
                    if (pb == b - 1 && cnt < 4) { cnt++; expreq[p] = 1; continue; }
                    spillreq[p] = 1;
                }
            }
            __syncthreads();
            for (int i = tid + 1; i <= N; i += PNT) if (expreq[i]) {
                const u32 s = atomicAdd(&blkexp[(i - 1) >> 6], 1u);
                if (s < (u32)ES) expreq[i] = (u8)(2 + s); else { expreq[i] = 0; spillreq[i] = 1; }
            }
            __syncthreads();
        }
        // ---- A3: spill slots for the rows a far successor reads back and for the sink rows (tie inspection); the row engine keeps every row in HBM anyway
        for (int i = tid + 1; i <= N && !ROWS; i += PNT) {
            u32* mp = meta + (size_t)i * PMETA;
            const u32 m1 = mp[1];
            if (spillreq[i] || ((m1 >> 8) & 1)) {
                const u32 slot = atomicAdd(&S.spill_cnt, 1u);
                if (slot < (u32)PSPILL) { mp[9] = slot; mp[1] = m1 | (1u << 9); } else S.status = 4;
                if (spillreq[i]) stat_far++;
            }
        }
        __syncthreads();
        if constexpr (WF) {
            for (int i = tid + 1; i <= N; i += PNT) {
                const u32* mp = meta + (size_t)i * PMETA;
                const u32 m1 = mp[1];
                const int np = (int)(m1 >> 16);
                const int b = (i - 1) >> 6, l = (i - 1) & 63, wv = b % NBW, wp = (b + NBW - 1) % NBW;
                u32 d[PWFD];
                #pragma unroll
                for (int x = 0; x < PWFD; x++) d[x] = 0;
                d[0] = mp[0];
                int cnt = 0, nm = 0, nf = 0, dpp = 0;
                if (np == 0) { d[3] = WfCfg<C>::CST / 2; nm = 1; }                 // the virtual source row: the constant cell 0, ordinal 0
                for (int o = 0; o < np && o < 16; o++) {
                    const int p = wf_pred(mp, o), k = i - p, pb = (p - 1) >> 6;
                    if (k == 1 && l >= 1) { dpp = o + 1; continue; }
                    if (pb == b && k <= KR && cnt < 4) { cnt++; d[3 + nm++] = ((WfCfg<C>::RING + (u32)(wv * 64 + (l - k)) * TR * 2) / 2) | ((u32)k << 16) | (1u << 24) | ((u32)o << 28); continue; }
                    if (pb == b - 1 && cnt < 4) {
                        cnt++;
                        const u32 er = expreq[p];
                        if (er >= 2) { d[3 + nm++] = ((WfCfg<C>::EXP + (u32)(wp * ES + (int)(er - 2)) * TRX * 2) / 2) | ((u32)k << 16) | (2u << 24) | ((u32)o << 28); continue; }
                    }
                    if (nf < 4) d[7 + nf++] = (u32)p | ((u32)o << 16); else S.status = 12;
                }
                const u32 er = expreq[i];
                d[1] = (m1 & 0x3FFu) | ((u32)nm << 12) | ((u32)nf << 16) | ((er >= 2 ? er - 1 : 0u) << 20) | ((u32)dpp << 24);   // code, sink (bit 8), spill copy (bit 9)
                d[2] = mp[9];
                u32* dp = wfd + (size_t)i * PWFD;
                #pragma unroll
                for (int x = 0; x < PWFD; x += 4) *reinterpret_cast<uint4*>(dp + x) = make_uint4(d[x], d[x + 1], d[x + 2], d[x + 3]);
            }
            __syncthreads();
        }
        if (S.status) break;
        PG_TICK(0);
        // ---- B: the DP.  Every wave walks all rows; in row i it owns the chunk k = w (mod 8) of the band, if the band has one.
        int best_v = PNEG, best_i = 0, best_j = 0; bool multi = false;
        if constexpr (WF) {
        if (N > 0 && w < NBW) {
            // ---- B (anti-diagonal engine).  Wave w sweeps the blocks b = w, w + NBW, ...: lane l is row i = 64 b + 1 + l, and in the step of anti-diagonal A it
            // computes the cell of column j = A - i.  The cell to the left is the lane's own last value; the cells of a predecessor row p = i - k lie on the
            // anti-diagonals A - k (above) and A - k - 1 (diagonal): the lane before by one DPP move, anything else by ONE ds_read per step and predecessor
            // (the diagonal value is the one read a step earlier).  Every candidate carries its traceback priority in its low six bits -- 63 - o for the diagonal
            // of predecessor o, 47 - o for its vertical, 16 for the insertion -- so the maximum is value and back-pointer at once (first predecessor on the
            // diagonal, then on the vertical, then the insertion: the order of the walk in poa.hpp).  A lane outside its band writes the floor, so bands need
            // no test on the reading side.  Block b + 1 runs behind block b on the same anti-diagonals (it needs A - 1 of block b): progress words in LDS.
            typedef __attribute__((address_space(3))) int16_t lds_s16_t;
            typedef __attribute__((address_space(3))) u8 lds_u8_t;
            const u32 lds0 = (u32)(uintptr_t)((__attribute__((address_space(3))) u8*)lds_raw);
            auto lds_read_s16 = [](const u32 a) -> int { return (int)*(lds_s16_t*)(uintptr_t)a; };
            auto lds_write_s16 = [](const u32 a, const int v) { *(lds_s16_t*)(uintptr_t)a = (int16_t)v; };
            auto lds_read_u8 = [](const u32 a) -> int { return (int)*(lds_u8_t*)(uintptr_t)a; };
            constexpr int NEG64 = PNEG * 64, NOTAG = -(1 << 28);
            constexpr u32 NEGPAIR = ((u32)(u16)(int16_t)PNEG) * 0x10001u;
            if (tid == 0) { *reinterpret_cast<int16_t*>(lds_raw + WfCfg<C>::CST) = 0; *reinterpret_cast<int16_t*>(lds_raw + WfCfg<C>::CST + 2) = (int16_t)PNEG; }   // every wave of phase B waits for block 0 or is wave 0 itself: the cells are there before anybody reads them
            if ((lds0 & 4095u) != 0 && tid == 0) S.status = 13;
            u64 t_wait0 = 0;
            bool gave_up = false;
            auto wait_for = [&](const int wsrc, const int need, const bool brief = true) {   // until wave wsrc has published `need`; a wait that outlasts any real one ends the cluster with a status.  brief: a few steps of the block before (poll often); else a block's worth (poll rarely: the polls of an idle wave take issue slots from the wave that shares its SIMD)
                for (int spins = 0;; spins++) {
                    const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&S.done[wsrc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (v >= need) { if (spins) stat_spins += (u32)(wall_clock64() - t_wait0); break; }
                    if (!spins) t_wait0 = wall_clock64();
                    if (brief) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(12);
                    const u32 st = (u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&S.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (spins > (1 << 21) || st) { if (!st) S.status = 10; gave_up = true; break; }
                }
                asm volatile("" ::: "memory");
            };
            auto key_of = [](const int b, const int A) -> int { return (b << 16) | (A & 0xFFFF); };
            for (int b = w; b * 64 < N && !gave_up; b += NBW) {
                const int ib = b * 64, i = ib + 1 + lane;
                const bool valid = i <= N;
                const int wp = (w + NBW - 1) % NBW, wn = (w + 1) % NBW;
                // the lane's row
                const uint4* dq = reinterpret_cast<const uint4*>(wfd + (size_t)min(i, N) * PWFD);
                const uint4 q0 = dq[0], q1 = dq[1], q2 = dq[2];
                const int lo = valid ? (int)(q0.x & 0xFFFF) : 1, hi = valid ? (int)(q0.x >> 16) : 0;
                const int j0 = max(lo, 1), span = valid ? hi - j0 : -1;
                const u32 fl = valid ? q0.y : 0u;
                const int code = (int)(fl & 0xFF), sink = (int)((fl >> 8) & 1), spl = (int)((fl >> 9) & 1), nmem = (int)((fl >> 12) & 7), nfar = (int)((fl >> 16) & 7);
                const int eslot = (int)((fl >> 20) & 15) - 1, dppo = (int)((fl >> 24) & 31) - 1;
                // sweep range and what the block needs
                int a0 = valid ? i + j0 - 1 : 0x7FFFFFFF, a1 = valid ? i + hi + 7 : 0;     // seven steps past the band: the back-pointer bytes leave eight at a time
                int kx = 0;                                                                // the farthest predecessor read from the export area
                const u32 sw[4] = {q0.w, q1.x, q1.y, q1.z};
                #pragma unroll
                for (int s = 0; s < 4; s++) if (s < nmem && ((sw[s] >> 24) & 3) == 2) kx = max(kx, (int)((sw[s] >> 16) & 0xFF));
                #pragma unroll
                for (int s = 32; s >= 1; s >>= 1) { a0 = min(a0, __shfl_xor(a0, s)); a1 = max(a1, __shfl_xor(a1, s)); kx = max(kx, __shfl_xor(kx, s)); }
                const int A0 = __builtin_amdgcn_readfirstlane(a0) & ~7, A1 = __builtin_amdgcn_readfirstlane(a1), KX = __builtin_amdgcn_readfirstlane(kx);   // the sweep starts on a multiple of eight (eight back-pointer bytes per trip) and runs in trips of eight
                const int Ab = A0;
                const int ns = __ballot(nmem >= 4) ? 4 : (__ballot(nmem >= 3) ? 3 : (__ballot(nmem >= 2) ? 2 : 1));
                const bool any_far = __ballot(nfar > 0) != 0, any_spl = __ballot(spl != 0) != 0, any_lo0 = __ballot(valid && lo == 0) != 0;
                const bool any_end = __ballot(valid && (sink || hi == L)) != 0, any_L = __ballot(valid && hi == L) != 0;
                if (A1 - A0 + 10 > TRX) { S.status = 11; gave_up = true; break; }        // the export area holds TRX anti-diagonals of a row
                // the export area of this wave was read by block b - NBW + 1
                if (b >= NBW) wait_for(wn, key_of(b - NBW + 1, 0xFFFF), false);
                if (gave_up) break;
                {
                    uint4* rr = reinterpret_cast<uint4*>(lds_raw + WfCfg<C>::RING + (u32)(w * 64 + lane) * TR * 2);
                    uint4* ee = reinterpret_cast<uint4*>(lds_raw + WfCfg<C>::EXP + (u32)w * ES * TRX * 2 + (u32)lane * 128);
                    const uint4 ng = make_uint4(NEGPAIR, NEGPAIR, NEGPAIR, NEGPAIR);
                    #pragma unroll
                    for (int x = 0; x < TR * 2 / 16; x++) rr[x] = ng;
                    #pragma unroll
                    for (int x = 0; x < 8; x++) ee[x] = ng;
                }
                asm volatile("" ::: "memory");
                if (lane == 0) { S.blk_a0[w] = A0; S.blk_a1[w] = A1; blka0[b] = A0; }
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_store(&S.done[w], key_of(b, max(A0 - 1, 0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (A0 may be 0 for block 0: -1 would read as "finished")
                if (b > 0) {
                    wait_for(wp, key_of(b - 1, 0), false);                               // block b - 1 has cleared its areas and said where it sweeps
                    if (gave_up) break;
                    const int pa0 = __builtin_amdgcn_readfirstlane(*(volatile int*)&S.blk_a0[wp]), pa1 = __builtin_amdgcn_readfirstlane(*(volatile int*)&S.blk_a1[wp]);
                    // an anti-diagonal this block reads of an exported row must not share its slot with one the row was swept on (slot = anti-diagonal mod TRX)
                    if (A0 - KX - 1 <= pa1 - TRX || A1 - 1 - TRX >= pa0) { S.status = 11; gave_up = true; break; }
                }
                if (any_far) {                                                          // far predecessors are read from their copies in HBM: every earlier block is complete
                    for (int x = 1; x < NBW && !gave_up; x++) { const int bq = b - x; if (bq >= 0) wait_for(bq % NBW, key_of(bq, 0xFFFF)); }
                    if (gave_up) break;
                    if (lane == 0) stat_far++;
                }
                // predecessor slots: address = (xa & mask) | base, xa grows by one cell per step
                u32 xa[4], mk[4], bs[4]; int tg[4];
                #pragma unroll
                for (int s = 0; s < 4; s++) {
                    const u32 x = sw[s];
                    const int kind = s < nmem ? (int)((x >> 24) & 3) : 0, k = (int)((x >> 16) & 0xFF), o = (int)(x >> 28);
                    const u32 off = s < nmem ? (x & 0xFFFFu) * 2u : (WfCfg<C>::CST + 2);
                    bs[s] = lds0 + off;
                    mk[s] = kind == 1 ? (u32)(TR * 2 - 1) : (kind == 2 ? (u32)(TRX * 2 - 1) : 0u);
                    xa[s] = kind == 1 ? (u32)(A0 - 1 - k + (lane - k)) * 2u : (u32)(A0 - 1 - k) * 2u;     // ring rows are rotated by their lane
                    tg[s] = s < nmem ? 47 - o + 64 * SG : NOTAG;
                }
                const int tgD = dppo >= 0 ? 47 - dppo + 64 * SG : NOTAG;
                u32 xr = (u32)(A0 - 1 + lane) * 2u; const u32 rb = lds0 + WfCfg<C>::RING + (u32)(w * 64 + lane) * TR * 2;
                const u32 emk = eslot >= 0 ? (u32)(TRX * 2 - 1) : 0u, ebs = eslot >= 0 ? lds0 + WfCfg<C>::EXP + (u32)(w * ES + eslot) * TRX * 2 : lds0 + WfCfg<C>::DUMP + (u32)lane * 16;
                const int pre = (valid && lo == 0) ? 0 : NEG64;                        // the cell left of column 1 is free when the band starts in column 0
                int tj = A0 - 1 - i - j0;                                                // j - j0 of the step before the first
                const int sq_lo = (int)(lds0 + WfCfg<C>::SQ);
                int sa = sq_lo + (A0 - 2 - i);                                           // LDS address of base j - 1, the one column j is scored against
                int Hu = PNEG;
                u32 acc = 0;
                int bv = PNEG, bj = 0;
                const int db = (lo >> CSH) << CSH;
                u8* const Db = D + (size_t)b * ((size_t)TRX * 64);                        // this block's back-pointers: [trip][lane] eight bytes = the cells of anti-diagonals A0 + 8 trip .. + 7 of the lane's row (column = anti-diagonal - row)
                const long long srow = (long long)q0.z * STRIDE - db;                   // + j: this row's copy (spill slot q0.z)
                const int dlo = valid ? lo : 0x40000000; const u32 dspan = valid ? (u32)(hi + 7 - lo) : 0u, bspan = valid ? (u32)(hi - lo) : 0u;
                // the sweep, compiled for NS = 1 .. 4 LDS predecessors per lane; SP: the block has rows of the rare kinds (far predecessors, a band that starts in
                // column 0, rows with a copy in HBM, end cells) -- three blocks of four run the lean variant.  A lone wavefront issues one instruction per ~3 ns
                // whatever it depends on (measured: twenty more VALU instructions per step cost the same dependent or not), so the step is written for the
                // instruction COUNT: four anti-diagonals per loop trip (the back-pointer dword, the progress word and the wait for the block before are handled
                // once per trip and need no counters), and the diagonal candidate of a predecessor is its vertical candidate of the step before (same cell, same
                // priority constant: the score of the column is added to the maximum afterwards)
                auto sweep = [&](auto nsc, auto spc) {
                    constexpr int NS = decltype(nsc)::value; constexpr bool SP = decltype(spc)::value;
                    int pu[NS], puD = NEG64 + NOTAG;
                    #pragma unroll
                    for (int s = 0; s < NS; s++) pu[s] = NEG64 + NOTAG;
                    if (b > 0) { wait_for(wp, key_of(b - 1, Ab + 7), false); if (gave_up) return; }
                    // the LDS reads of a step are issued a step ahead (its cells were written at least a step before: k >= 2 in the ring, and block b - 1 is ahead by
                    // eight anti-diagonals), so a step does not wait for the LDS round trip
                    int rdn[NS], sbn;
                    auto issue = [&]() {
                        sa++; sbn = lds_read_u8((u32)sa);                  // the base column j is scored against.  No clamp: a step outside the row's band (inactive: its candidate is dropped) reads up to 70 bytes before / after the sequence -- the areas in front of it, and the 256 bytes poa_wf_lds adds behind it
                        #pragma unroll
                        for (int s = 0; s < NS; s++) { xa[s] += 2; rdn[s] = lds_read_s16((xa[s] & mk[s]) | bs[s]); }
                        asm volatile("" ::: "memory");
                    };
                    issue();
                    u32 eb = 0;                                                                     // the export cell of the trip's first anti-diagonal
                    auto step = [&](auto tc) {
                        constexpr int T = decltype(tc)::value;                                      // the step of the trip: the export store's offset is an immediate
                        tj++;
                        const bool active = (u32)tj <= (u32)span;
                        const int sb = sbn;
                        int rd[NS];
                        #pragma unroll
                        for (int s = 0; s < NS; s++) rd[s] = rdn[s];
                        issue();                                                                    // the next step's
                        const int upD = __builtin_amdgcn_update_dpp(0, Hu, 0x138, 0xF, 0xF, true);            // wave_shr:1 (lane 0 never has a DPP predecessor: its candidate carries NOTAG)
                        int dm = puD;
                        int um = (upD << 6) + tgD; puD = um;
                        #pragma unroll
                        for (int s = 0; s < NS; s++) { const int u = (rd[s] << 6) + tg[s]; dm = max(dm, pu[s]); pu[s] = u; um = max(um, u); }
                        if constexpr (SP) {
                            if (any_far && nfar > 0) {
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's own copies (a far row of the same block) have left
                                const u32 fw[4] = {q1.w, q2.x, q2.y, q2.z};
                                const int j = tj + j0;
                                for (int f = 0; f < nfar; f++) {
                                    const int p = (int)(fw[f] & 0xFFFF), o = (int)(fw[f] >> 16);
                                    const u32* mpp = meta + (size_t)p * PMETA;
                                    const u32 lh = mpp[0]; const int lop = (int)(lh & 0xFFFF), hip = (int)(lh >> 16);
                                    int16_t* src = spillH + (size_t)mpp[9] * STRIDE - ((lop >> CSH) << CSH);
                                    const int uv = (j >= lop && j <= hip) ? (int)__hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : PNEG;
                                    const int dv = (j - 1 >= lop && j - 1 <= hip) ? (int)__hip_atomic_load(src + j - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : PNEG;
                                    const int tt = 47 - o + 64 * SG;
                                    um = max(um, (uv << 6) + tt); dm = max(dm, (dv << 6) + tt);
                                }
                            }
                        }
                        const int sc = (sb == code) ? (SM * 64 + 16 - 64 * SG) : (SX * 64 + 16 - 64 * SG);
                        int m = max(max(um, (Hu << 6) + (64 * SG + 16)), NEG64 + 16);
                        m = max(m, dm + sc);
                        int inact = NEG64;
                        if constexpr (SP) { if (any_lo0) inact = tj < 0 ? pre : NEG64; }
                        const int cand = active ? m : inact;
                        Hu = cand >> 6;
                        // the cell: own ring row, export row (or the dump cell), back-pointer byte
                        xr += 2; lds_write_s16((xr & (u32)(TR * 2 - 1)) | rb, Hu);
                        lds_write_s16(eb + 2u * T, Hu);
                        acc = __builtin_amdgcn_alignbit((u32)cand, acc, 8);
                        if constexpr (SP) {
                            const int j = tj + j0;
                            if (any_spl) { if (spl && (u32)(j - dlo) <= bspan) spillH[srow + j] = (int16_t)Hu; }       // column 0 of a band that starts there included
                            if (any_end) {
                                if (any_L && hi == L && j == L && valid) endval[i] = (int16_t)Hu;
                                if ((u32)(j - dlo) <= bspan && (sink || j == L) && Hu > bv) { bv = Hu; bj = j; }
                            }
                        }
                    };
                    // round 5: EIGHT anti-diagonals per trip (four in round 4): the wait for the block before, the back-pointer store (a 64-bit address, a range test, one
                    // store instruction touching 64 rows), the progress word and the loop control were ~35 of a trip's ~135 instructions
                    for (int Ag = Ab; Ag <= A1; Ag += 8) {
                        if (b > 0) { wait_for(wp, key_of(b - 1, Ag + 7)); if (gave_up) break; }   // block b - 1 is through anti-diagonal Ag + 7: eight steps of reading its rows, a step ahead
                        // a trip starts on a multiple of eight anti-diagonals and an export row holds a multiple of eight: its eight cells are consecutive
                        eb = ((u32)Ag * 2u & emk) | ebs;
                        step(std::integral_constant<int, 0>()); step(std::integral_constant<int, 1>()); step(std::integral_constant<int, 2>()); step(std::integral_constant<int, 3>());
                        const u32 acc_lo = acc;
                        step(std::integral_constant<int, 4>()); step(std::integral_constant<int, 5>()); step(std::integral_constant<int, 6>()); step(std::integral_constant<int, 7>());
                        const int j = tj + j0;                                           // the column of anti-diagonal Ag + 7
                        if ((u32)(j - dlo) <= dspan) *reinterpret_cast<uint2*>(Db + ((size_t)((Ag - Ab) >> 3) * 64 + lane) * 8) = make_uint2(acc_lo, acc);   // trip-major: the lanes inside their bands are neighbours, their eight bytes too
                        if (lane == 0) __hip_atomic_store(&S.done[w], key_of(b, Ag + 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                };
                const bool special = any_far || any_lo0 || any_spl || any_end;
                __builtin_amdgcn_s_setprio(2);
                #define PG_SWEEP(NN) do { if (special) sweep(std::integral_constant<int, NN>(), std::true_type()); else sweep(std::integral_constant<int, NN>(), std::false_type()); } while (0)
                if (ns == 1) PG_SWEEP(1); else if (ns == 2) PG_SWEEP(2); else if (ns == 3) PG_SWEEP(3); else PG_SWEEP(4);
                #undef PG_SWEEP
                __builtin_amdgcn_s_setprio(0);
                if (gave_up) break;
                if (any_spl) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the copies have landed before anybody learns that the block is done
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_store(&S.done[w], key_of(b, 0xFFFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (any_end && valid && bv > PNEG) {                                     // this thread's end cell so far (phase C reduces: value, then row, then column)
                    if (bv > best_v) { best_v = bv; best_i = i; best_j = bj; multi = false; }
                    else if (bv == best_v && i != best_i) multi = true;                  // a thread's rows come in increasing order
                }
                stat_tasks++;
            }
        }
        } else
        if constexpr (ROWS) {
        if (N > 0 && w == 0) {
            // ---- B (row engine): wave 0 takes the rows one after the other; the other waves wait at the barrier below
            typedef short pk2 __attribute__((ext_vector_type(2)));
            int16_t* V = reinterpret_cast<int16_t*>(D);                            // [N + 1][W] values of every row, column j at index j mod W, PNEG outside the row's band
            __builtin_amdgcn_s_setprio(3);                                        // one latency-bound wave per cluster: it issues ~1 instruction per 7 cycles and must not queue behind the eight-wave kernels of other samples on its SIMD
            int prev[C];                                                          // the row before, this lane's cells
            #pragma unroll
            for (int c = 0; c < C; c++) prev[c] = PNEG;
            const int nblk = L / C + 1;
            auto unpack = [&](const u32* src, int (&out)[C]) {
                #pragma unroll
                for (int k2 = 0; k2 < C / 2; k2++) { const u32 x = src[k2]; out[2 * k2] = ((int)(x << 16)) >> 16; out[2 * k2 + 1] = ((int)x) >> 16; }
            };
            // A row's descriptor and everything that depends on it alone -- band, block of this lane, valid cells, match bits -- is PREPARED one row ahead
            // (software pipelining by hand): the scan of row i is a chain of dependent DPP steps, and a lone wave retires a dependent instruction every
            // ~8 cycles against ~4 for an independent one; the preparation of row i + 1 is independent work that fills those slots.
            struct RowPrep { int lo, hi, j0, code, sink, slow, partial, np, p0, p1, blo; u32 lh0, lh1; int jb; u32 vmask, mbits; };
            u32 mv[6];                                                            // descriptor words 0, 1, 2, 4, 5 of 64 rows, lane = row (the general step reads the rest from memory)
            auto prepare = [&](const int r) -> RowPrep {
                RowPrep q;
                if (((r - 1) & 63) == 0) {                                        // a new batch of 64 descriptors
                    const int row = min(r + lane, N);
                    const uint4* mp = reinterpret_cast<const uint4*>(meta + (size_t)row * PMETA);
                    const uint4 q0 = mp[0], q1 = mp[1];
                    mv[0] = q0.x; mv[1] = q0.y; mv[2] = q0.z; mv[3] = q1.x; mv[4] = q1.y;
                    #pragma unroll
                    for (int x = 0; x < 5; x++) asm volatile("v_mov_b32 %0, %1" : "=v"(mv[x]) : "v"(mv[x]));      // see the chunk pipeline: no pending load behind the readlanes
                }
                const int rr = (r - 1) & 63;
                const u32 m0 = __builtin_amdgcn_readlane(mv[0], rr), m1 = __builtin_amdgcn_readlane(mv[1], rr), m2 = __builtin_amdgcn_readlane(mv[2], rr);
                q.lh0 = __builtin_amdgcn_readlane(mv[3], rr); q.lh1 = __builtin_amdgcn_readlane(mv[4], rr);
                q.lo = (int)(m0 & 0xFFFF); q.hi = (int)(m0 >> 16);
                q.code = (int)(m1 & 0xFF); q.sink = (int)((m1 >> 8) & 1); q.slow = (int)((m1 >> 10) & 1); q.partial = (int)((m1 >> 11) & 1); q.np = (int)(m1 >> 16);
                q.p0 = (int)(m2 & 0xFFFF); q.p1 = (int)(m2 >> 16);
                q.j0 = max(q.lo, 1);
                q.blo = q.lo / C;
                const int blk = q.blo + ((lane - q.blo) & 63);                    // the block of C columns this lane holds for this row: block b lives in lane b mod 64
                q.jb = blk * C;
                const int va = min(max(q.j0 - q.jb, 0), C), vb = min(max(q.hi - q.jb + 1, 0), C);
                q.vmask = vb > va ? (((1u << (vb - va)) - 1u) << va) : 0u;        // cells of this lane inside [j0, hi]
                q.mbits = (msk[min(blk, nblk)] >> (8 * ((q.code >> 1) & 3))) & 0xFFu;   // cells whose base equals the node's letter
                return q;
            };
            RowPrep nxt = prepare(1);
            {
                for (int i = 1; i <= N; i++) {
                    const RowPrep cur = nxt;
                    const int lo = cur.lo, hi = cur.hi, j0 = cur.j0, sink = cur.sink, slow = cur.slow, np = cur.np, blo = cur.blo, jb = cur.jb;
                    const u32 vmask = cur.vmask, mbits = cur.mbits;
                    if (lane == 0) { stat_far += (u32)slow; stat_tasks += (u32)(np == 2); }   // row engine: `far_rows` counts the rows of the general step, tasks[0] the rows with two predecessors
                    // mm8[c] = 8 + max(PNEG, the best (mis)match or deletion candidate of the cell): the +8 turns the scores {3, -8} into {11, 0} = 11 x the match bit
                    int mm8[C];
                    if (!slow) {
                        // ---- the lean step: one or two predecessors, the row before (registers) or a row of the ring, whose stored cells are what the
                        // candidates need (PNEG outside a band; phase A checked that no column aliases another).  `partial`: a predecessor's band covers
                        // only part of the row's range [j0, hi] -- its candidates count on the columns [max(j0, lop), min(hi, hip + 1)] only
                        const int partial = cur.partial;
                        const int p0 = cur.p0, p1 = cur.p1;
                        int pv[C];
                        if (p0 == i - 1) {
                            #pragma unroll
                            for (int c = 0; c < C; c++) pv[c] = prev[c];
                        } else unpack(reinterpret_cast<const u32*>(ring + (p0 & (VR - 1)) * W + lane * C), pv);
                        int left = __builtin_amdgcn_update_dpp(0, pv[C - 1], 0x13C, 0xF, 0xF, false);       // wave_ror:1: the cell left of this lane's first one
                        #pragma unroll
                        for (int c = 0; c < C; c++) {
                            const int lv = c == 0 ? left : pv[c - 1];
                            mm8[c] = max(max(lv + 11 * (int)__builtin_amdgcn_ubfe(mbits, c, 1), pv[c] + (SG + 8)), PNEG + 8);
                        }
                        if (partial) {
                            const u32 lh0 = cur.lh0;
                            const int ra = max(j0, (int)(lh0 & 0xFFFF)), rb = min(hi, (int)(lh0 >> 16) + 1);
                            const int ia = min(max(ra - jb, 0), C), ibx = min(max(rb - jb + 1, 0), C);
                            const u32 im = ibx > ia ? (((1u << (ibx - ia)) - 1u) << ia) : 0u;
                            #pragma unroll
                            for (int c = 0; c < C; c++) { const int M = __builtin_amdgcn_sbfe((int)im, c, 1); mm8[c] = (mm8[c] & M) | ((PNEG + 8) & ~M); }
                        }
                        if (np == 2) {
                            if (p1 == i - 1) {
                                #pragma unroll
                                for (int c = 0; c < C; c++) pv[c] = prev[c];
                            } else unpack(reinterpret_cast<const u32*>(ring + (p1 & (VR - 1)) * W + lane * C), pv);
                            left = __builtin_amdgcn_update_dpp(0, pv[C - 1], 0x13C, 0xF, 0xF, false);
                            if (!partial) {
                                #pragma unroll
                                for (int c = 0; c < C; c++) {
                                    const int lv = c == 0 ? left : pv[c - 1];
                                    mm8[c] = max(max(lv + 11 * (int)__builtin_amdgcn_ubfe(mbits, c, 1), pv[c] + (SG + 8)), mm8[c]);
                                }
                            } else {
                                const u32 lh1 = cur.lh1;
                                const int ra = max(j0, (int)(lh1 & 0xFFFF)), rb = min(hi, (int)(lh1 >> 16) + 1);
                                const int ia = min(max(ra - jb, 0), C), ibx = min(max(rb - jb + 1, 0), C);
                                const u32 im = ibx > ia ? (((1u << (ibx - ia)) - 1u) << ia) : 0u;
                                #pragma unroll
                                for (int c = 0; c < C; c++) {
                                    const int lv = c == 0 ? left : pv[c - 1];
                                    const int cand = max(lv + 11 * (int)__builtin_amdgcn_ubfe(mbits, c, 1), pv[c] + (SG + 8));
                                    const int M = __builtin_amdgcn_sbfe((int)im, c, 1);
                                    mm8[c] = max(mm8[c], (cand & M) | (MININT & ~M));
                                }
                            }
                        }
                    } else {
                        // ---- the general step: any number of predecessors in in-edge order, the virtual source row, rows older than the ring (from HBM),
                        // bands whose columns alias modulo W: every candidate is tested against its predecessor's band and range
                        int dmax[C], umax[C];
                        #pragma unroll
                        for (int c = 0; c < C; c++) { dmax[c] = PNEG; umax[c] = PNEG; }
                        const int npe = np == 0 ? 1 : np;
                        for (int o = 0; o < npe; o++) {
                            int p; u32 lh;
                            const u32* mpi = meta + (size_t)i * PMETA;                   // the general step reads its descriptor from memory (the registers may hold the next batch)
                            if (o < 4) {
                                const u32 pp = (u32)__builtin_amdgcn_readfirstlane(launder((int)mpi[2 + (o >> 1)]));
                                p = (int)((pp >> (16 * (o & 1))) & 0xFFFF);
                                lh = (u32)__builtin_amdgcn_readfirstlane(launder((int)mpi[4 + o]));
                            } else {
                                const uint2 pl = plist[(u32)__builtin_amdgcn_readfirstlane(launder((int)mpi[8])) + (u32)o];
                                p = __builtin_amdgcn_readfirstlane(launder((int)pl.x)); lh = (u32)__builtin_amdgcn_readfirstlane(launder((int)pl.y));
                            }
                            if (np == 0) p = 0;
                            const int lop = (int)(lh & 0xFFFF), hip = (int)(lh >> 16);
                            int raw[C];
                            if (p == 0) {
                                #pragma unroll
                                for (int c = 0; c < C; c++) raw[c] = 0;                      // the virtual source row: 0 in its band [0, L]
                            } else if (p == i - 1) {
                                #pragma unroll
                                for (int c = 0; c < C; c++) raw[c] = prev[c];
                            } else if (i - p < VR) unpack(reinterpret_cast<const u32*>(ring + (p & (VR - 1)) * W + lane * C), raw);
                            else {
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the row's own store (>= 64 rows ago) has left the wave
                                const u32* src = reinterpret_cast<const u32*>(V + (size_t)p * W + lane * C);
                                u32 xw[C / 2];
                                #pragma unroll
                                for (int k2 = 0; k2 < C / 2; k2++) xw[k2] = __hip_atomic_load(src + k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // L2-served: this CU's L1 may hold the line from before the store
                                unpack(xw, raw);
                            }
                            int rawl = __builtin_amdgcn_update_dpp(0, raw[C - 1], 0x13C, 0xF, 0xF, false);
                            const int ra = max(j0, lop), rb = min(hi, hip + 1);
                            #pragma unroll
                            for (int c = 0; c < C; c++) {
                                const int j = jb + c;
                                const int lraw = c == 0 ? rawl : raw[c - 1];
                                const int lv = (j - 1 >= lop && j - 1 <= hip) ? lraw : PNEG;
                                const int uv = (j >= lop && j <= hip) ? raw[c] : PNEG;
                                const bool in = j >= ra && j <= rb;
                                const int d = in ? lv + (((mbits >> c) & 1u) ? SM : SX) : MININT, u = in ? uv + SG : MININT;
                                dmax[c] = max(dmax[c], d); umax[c] = max(umax[c], u);
                            }
                        }
                        #pragma unroll
                        for (int c = 0; c < C; c++) mm8[c] = max(dmax[c], umax[c]) + 8;
                    }
                    if (i < N) nxt = prepare(i + 1);                               // independent of this row's values: fills the issue slots of the scan below
                    // ---- insertion chain: prefix maximum of (candidate + 6 (j - lo)) over the band's columns in order, seeded by the cell left of j0
                    const int base6 = 6 * (jb - lo);
                    int run[C], vm[C]; int acc = -32768;
                    #pragma unroll
                    for (int c = 0; c < C; c++) {
                        vm[c] = __builtin_amdgcn_sbfe((int)vmask, c, 1);         // all ones for a cell inside [j0, hi]
                        const int tt = mm8[c] + base6 + (6 * c - 8);
                        acc = max(acc, (tt & vm[c]) | (-32768 & ~vm[c]));
                        run[c] = acc;
                    }
                    const int s0 = blo & 63;                                      // the band starts in lane s0 and runs through lane 63 into lanes 0 .. s0 - 1
                    const bool segA = lane >= s0;
                    int pkv = segA ? (int)(((u32)acc & 0xFFFFu) | 0x80000000u) : (int)(0x8000u | ((u32)acc << 16));   // {arc A, arc B} as two int16, the other half at -32768
                    auto pkmax = [](int a, int b) { pk2 x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4); pk2 z = __builtin_elementwise_max(x, y); int r; __builtin_memcpy(&r, &z, 4); return r; };
                    const int PKID = (int)0x80008000u;
                    pkv = pkmax(pkv, __builtin_amdgcn_update_dpp(PKID, pkv, 0x111, 0xF, 0xF, false));   // row_shr:1
                    pkv = pkmax(pkv, __builtin_amdgcn_update_dpp(PKID, pkv, 0x112, 0xF, 0xF, false));   // row_shr:2
                    pkv = pkmax(pkv, __builtin_amdgcn_update_dpp(PKID, pkv, 0x114, 0xF, 0xF, false));   // row_shr:4
                    pkv = pkmax(pkv, __builtin_amdgcn_update_dpp(PKID, pkv, 0x118, 0xF, 0xF, false));   // row_shr:8
                    pkv = pkmax(pkv, __builtin_amdgcn_update_dpp(PKID, pkv, 0x142, 0xA, 0xF, false));   // row_bcast:15
                    pkv = pkmax(pkv, __builtin_amdgcn_update_dpp(PKID, pkv, 0x143, 0xC, 0xF, false));   // row_bcast:31
                    const int exv = __builtin_amdgcn_update_dpp(PKID, pkv, 0x138, 0xF, 0xF, false);      // wave_shr:1: the lanes before this one (lane 0: none)
                    const int totA = ((int)((u32)__builtin_amdgcn_readlane(pkv, 63) << 16)) >> 16;       // all of arc A
                    const int exA = ((int)((u32)exv << 16)) >> 16, exB = exv >> 16;
                    const int first = (lo == 0 ? 0 : PNEG) + 6 * (j0 - 1 - lo);
                    const int ex = max(segA ? exA : max(exB, totA), first);
                    int vv[C];
                    #pragma unroll
                    for (int c = 0; c < C; c++) {
                        const int x = max(max(ex, run[c]) - base6 - 6 * c, PNEG);
                        vv[c] = (x & vm[c]) | (PNEG & ~vm[c]);
                    }
                    if (lo == 0 && jb == 0) vv[0] = 0;                           // column 0 of a band that starts there: the free left border
                    if (sink || hi == L) {                                        // end cells: any column of a sink row, column L of any row
                        #pragma unroll
                        for (int c = 0; c < C; c++) {
                            const int j = jb + c, v = vv[c];
                            if (j >= lo && j <= hi && (sink || j == L)) {
                                if (v > best_v) { best_v = v; best_i = i; best_j = j; multi = false; }
                                else if (v == best_v && i != best_i) multi = true;
                            }
                        }
                    }
                    // the row: registers for the next row, LDS ring for the next 63, HBM for far successors, the end-cell inspection and the traceback
                    u32 pkd[C / 2];
                    #pragma unroll
                    for (int c = 0; c < C; c++) prev[c] = vv[c];
                    #pragma unroll
                    for (int k2 = 0; k2 < C / 2; k2++) pkd[k2] = ((u32)vv[2 * k2] & 0xFFFFu) | ((u32)vv[2 * k2 + 1] << 16);
                    u32* rdst = reinterpret_cast<u32*>(ring + (i & (VR - 1)) * W + lane * C);
                    u32* gdst = reinterpret_cast<u32*>(V + (size_t)i * W + lane * C);
                    #pragma unroll
                    for (int k2 = 0; k2 < C / 2; k2++) { rdst[k2] = pkd[k2]; gdst[k2] = pkd[k2]; }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                     // the rows are in L2 before anybody inspects them (end cells, traceback)
        }
        } else
        if (N > 0) {
            int16_t* myring = ring + w * (R * CW);
            const int wl = (w + PW - 1) & (PW - 1), wr = (w + 1) & (PW - 1);
            const int16_t* lring = ring + wl * (R * CW);
            for (int ib = 0; ib < N; ib += 64) {
                u32 mv[PMETA];
                {
                    const int row = min(ib + 1 + lane, N);
                    const uint4* mp = reinterpret_cast<const uint4*>(meta + (size_t)row * PMETA);
                    const uint4 q0 = mp[0], q1 = mp[1], q2 = mp[2];
                    mv[0] = q0.x; mv[1] = q0.y; mv[2] = q0.z; mv[3] = q0.w; mv[4] = q1.x; mv[5] = q1.y; mv[6] = q1.z; mv[7] = q1.w; mv[8] = q2.x; mv[9] = q2.y; mv[10] = q2.z; mv[11] = q2.w;
                    // The row loop below stores (back-pointers) and reads these registers by v_readlane: as results of loads still "in flight" in the
                    // compiler's bookkeeping they made every row wait for its predecessor's HBM store (s_waitcnt vmcnt(1) before each readlane, 1.2 us
                    // per row).  One move each, after the loads have landed, gives the loop registers without a pending load.
                    #pragma unroll
                    for (int x = 0; x < PMETA; x++) asm volatile("v_mov_b32 %0, %1" : "=v"(mv[x]) : "v"(mv[x]));
                }
                const int rend = min(64, N - ib);
                for (int rr = 0; rr < rend; rr++) {
                    const int i = ib + 1 + rr;
                    const u32 m0 = __builtin_amdgcn_readlane(mv[0], rr);
                    const int lo = (int)(m0 & 0xFFFF), hi = (int)(m0 >> 16);
                    const int k0 = lo >> CSH, k1 = hi >> CSH;
                    const int k = k0 + ((w - k0) & (PW - 1));
                    if (k <= k1) {
                        stat_tasks++;
                        const u32 m1 = __builtin_amdgcn_readlane(mv[1], rr), m2 = __builtin_amdgcn_readlane(mv[2], rr);
                        const int code = (int)(m1 & 0xFF), sink = (int)((m1 >> 8) & 1), spl = (int)((m1 >> 9) & 1), slow = (int)((m1 >> 10) & 1), np = (int)(m1 >> 16);
                        const int jbase = k << CSH, jf = jbase + lane * C, j0 = max(lo, 1), dbase = k0 << CSH;
                        // the first two predecessors (nine rows of ten have no more) are fetched together with everything else the row reads from LDS;
                        // the descriptor repeats the first as the second when there is only one, and describes the virtual source row as row 0
                        const int p0 = (int)(m2 & 0xFFFF), p1 = (int)(m2 >> 16);
                        const u32 lh0 = __builtin_amdgcn_readlane(mv[4], rr), lh1 = __builtin_amdgcn_readlane(mv[5], rr);
                        const int lop0 = (int)(lh0 & 0xFFFF), hip0 = (int)(lh0 >> 16), lop1 = (int)(lh1 & 0xFFFF), hip1 = (int)(lh1 >> 16);
                        // what the left neighbour must have finished: this row when the chunk continues it (carry + boundary cells), else the latest
                        // predecessor row whose band reaches into the neighbour's chunk (its cell in column jbase - 1 feeds column jbase)
                        const int needL = (k > k0) ? i : (int)__builtin_amdgcn_readlane(mv[11], rr);
                        const int needR = i - (R - DMAX) - 1;                   // the ring slot this row overwrites must no longer be needed by the right neighbour
                        const int s0 = (p0 & (R - 1)) << CSH, s1 = (p1 & (R - 1)) << CSH, si = (i & (R - 1)) << CSH;
                        int own0[C], own1[C], svb[C], bnd0, bnd1, carry;
                        u64 t_wait0 = 0;
                        for (int spins = 0;; spins++) {
                            const int flv = __hip_atomic_load(&S.done[wl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const int frv = __hip_atomic_load(&S.done[wr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            asm volatile("" ::: "memory");                      // the data reads below stay behind the two flag reads (LDS serves a wave in order)
                            if constexpr (C == 1) { own0[0] = (int)myring[s0 + lane]; own1[0] = (int)myring[s1 + lane]; }
                            else if constexpr (C == 2) {
                                const int x0 = *reinterpret_cast<const int*>(myring + s0 + lane * 2), x1 = *reinterpret_cast<const int*>(myring + s1 + lane * 2);
                                own0[0] = (x0 << 16) >> 16; own0[1] = x0 >> 16; own1[0] = (x1 << 16) >> 16; own1[1] = x1 >> 16;
                            } else {
                                const int2 x0 = *reinterpret_cast<const int2*>(myring + s0 + lane * 4), x1 = *reinterpret_cast<const int2*>(myring + s1 + lane * 4);
                                own0[0] = (x0.x << 16) >> 16; own0[1] = x0.x >> 16; own0[2] = (x0.y << 16) >> 16; own0[3] = x0.y >> 16;
                                own1[0] = (x1.x << 16) >> 16; own1[1] = x1.x >> 16; own1[2] = (x1.y << 16) >> 16; own1[3] = x1.y >> 16;
                            }
                            bnd0 = (int)lring[s0 + CW - 1]; bnd1 = (int)lring[s1 + CW - 1]; carry = (int)lring[si + CW - 1];
                            #pragma unroll
                            for (int c = 0; c < C; c++) svb[c] = (int)sq[min(max(jf + c - 1, 0), L - 1)];
                            asm volatile("" ::: "memory");
                            const int fl = __builtin_amdgcn_readfirstlane(flv), fr = __builtin_amdgcn_readfirstlane(frv);
                            if (fl >= needL && fr >= needR) { if (spins) stat_spins += (u32)(wall_clock64() - t_wait0); break; }
                            if (!spins) t_wait0 = wall_clock64();
                            __builtin_amdgcn_s_sleep(1);
                            if (spins > (1 << 22)) { S.status = 10; break; }    // a wait that outlasts any real one ends the cluster with a status instead of hanging the device
                        }
                        int sc[C], dmax[C], umax[C], dd[C], du[C];
                        #pragma unroll
                        for (int c = 0; c < C; c++) sc[c] = (svb[c] == code) ? SM : SX;   // columns 0 and > L of a chunk are masked below: their score is never used
                        if (!slow) {
                            // ---- the lean path: one or two predecessor rows in the ring (or the virtual source row).  A predecessor contributes on
                            // the columns [max(j0, lo_p), min(hi, hi_p + 1)]; inside that range its cell (p, j-1) is missing only at j = lo_p and its cell
                            // (p, j) only at j = hi_p + 1 (band edges), where the floor stands in -- the same candidates as the generic path below.
                            if (p0 == 0) {
                                #pragma unroll
                                for (int c = 0; c < C; c++) { own0[c] = 0; own1[c] = 0; }
                                bnd0 = 0; bnd1 = 0;
                            }
                            int left0 = __builtin_amdgcn_update_dpp(0, own0[C - 1], 0x138, 0xF, 0xF, false);   // wave_shr:1
                            if (lane == 0) left0 = bnd0;
                            const int ra0 = max(j0, lop0), cnt0 = max(min(hi, hip0 + 1) - ra0 + 1, 0);
                            #pragma unroll
                            for (int c = 0; c < C; c++) {
                                const int j = jf + c;
                                const bool in = (u32)(j - ra0) < (u32)cnt0;
                                const int lv = (j == lop0) ? PNEG : (c == 0 ? left0 : own0[c == 0 ? 0 : c - 1]);
                                const int uv = (j == hip0 + 1) ? PNEG : own0[c];
                                dmax[c] = in ? max(lv + sc[c], PNEG) : PNEG; umax[c] = in ? max(uv + SG, PNEG) : PNEG;
                                dd[c] = 0; du[c] = 0;
                            }
                            if (np == 2) {
                                int left1 = __builtin_amdgcn_update_dpp(0, own1[C - 1], 0x138, 0xF, 0xF, false);
                                if (lane == 0) left1 = bnd1;
                                const int ra1 = max(j0, lop1), cnt1 = max(min(hi, hip1 + 1) - ra1 + 1, 0);
                                #pragma unroll
                                for (int c = 0; c < C; c++) {
                                    const int j = jf + c;
                                    const bool in = (u32)(j - ra1) < (u32)cnt1;
                                    const int lv = (j == lop1) ? PNEG : (c == 0 ? left1 : own1[c == 0 ? 0 : c - 1]);
                                    const int uv = (j == hip1 + 1) ? PNEG : own1[c];
                                    const int d = in ? lv + sc[c] : MININT, u = in ? uv + SG : MININT;
                                    const bool bd = d > dmax[c], bu = u > umax[c];
                                    dmax[c] = bd ? d : dmax[c]; dd[c] = bd ? 1 : 0;
                                    umax[c] = bu ? u : umax[c]; du[c] = bu ? 1 : 0;
                                }
                            }
                        } else {
                        // ---- the generic path: any number of predecessors, rows older than the ring from their HBM spill copy
                        #pragma unroll
                        for (int c = 0; c < C; c++) { dmax[c] = PNEG; umax[c] = PNEG; dd[c] = 0; du[c] = 0; }
                        const bool ring0 = p0 > 0 && i - p0 < DMAX, ring1 = np >= 2 && i - p1 < DMAX;
                        // candidates of one predecessor row: val[c] = cell (p, jf - 1 + c), PNEG outside the predecessor's band
                        auto apply = [&](const int (&val)[C + 1], const int lop, const int hip, const int ord) {
                            const int ra = max(j0, lop), rb = min(hi, hip + 1);
                            #pragma unroll
                            for (int c = 0; c < C; c++) {
                                const int j = jf + c;
                                const bool in = j >= ra && j <= rb;
                                const int d = in ? val[c] + sc[c] : MININT, u = in ? val[c + 1] + SG : MININT;
                                const bool bd = d > dmax[c], bu = u > umax[c];
                                dmax[c] = bd ? d : dmax[c]; dd[c] = bd ? ord : dd[c];
                                umax[c] = bu ? u : umax[c]; du[c] = bu ? ord : du[c];
                            }
                        };
                        auto from_ring = [&](const int (&own)[C], const int bnd, const int lop, const int hip, const int ord) {
                            int val[C + 1];
                            int left = __builtin_amdgcn_update_dpp(0, own[C - 1], 0x138, 0xF, 0xF, false);   // wave_shr:1
                            if (lane == 0) left = bnd;
                            val[0] = left;
                            #pragma unroll
                            for (int c = 0; c < C; c++) val[c + 1] = own[c];
                            #pragma unroll
                            for (int c = 0; c <= C; c++) { const int x = jf - 1 + c; val[c] = (x >= lop && x <= hip) ? val[c] : PNEG; }
                            apply(val, lop, hip, ord);
                        };
                        auto from_far = [&](const int p, const int lop, const int hip, const int ord) {          // virtual source row, or a row older than the ring
                            int val[C + 1];
                            if (p == 0) {
                                #pragma unroll
                                for (int c = 0; c <= C; c++) { const int x = jf - 1 + c; val[c] = (x >= 0 && x <= L) ? 0 : PNEG; }
                            } else if (i - p < DMAX) {
                                const int sp = (p & (R - 1)) * CW;
                                int own[C];
                                #pragma unroll
                                for (int c = 0; c < C; c++) own[c] = (int)myring[sp + lane * C + c];
                                from_ring(own, (int)lring[sp + CW - 1], lop, hip, ord);
                                return;
                            } else {
                                const u32 slot = (u32)__builtin_amdgcn_readfirstlane(launder((int)meta[(size_t)p * PMETA + 9]));
                                const int16_t* src = spillH + (size_t)slot * STRIDE;
                                const int pb = (lop >> CSH) << CSH;
                                #pragma unroll
                                for (int c = 0; c <= C; c++) { const int x = jf - 1 + c; val[c] = launder((int)src[min(max(x - pb, 0), STRIDE - 1)]); }
                                #pragma unroll
                                for (int c = 0; c <= C; c++) { const int x = jf - 1 + c; val[c] = (x >= lop && x <= hip) ? val[c] : PNEG; }
                            }
                            apply(val, lop, hip, ord);
                        };
                        if (ring0) from_ring(own0, bnd0, lop0, hip0, 0); else from_far(p0, lop0, hip0, 0);
                        if (np >= 2) { if (ring1) from_ring(own1, bnd1, lop1, hip1, 1); else from_far(p1, lop1, hip1, 1); }
                        if (np >= 3) {
                            const u32 m3 = __builtin_amdgcn_readlane(mv[3], rr);
                            { const u32 lh = __builtin_amdgcn_readlane(mv[6], rr); from_far((int)(m3 & 0xFFFF), (int)(lh & 0xFFFF), (int)(lh >> 16), 2); }
                            if (np >= 4) { const u32 lh = __builtin_amdgcn_readlane(mv[7], rr); from_far((int)(m3 >> 16), (int)(lh & 0xFFFF), (int)(lh >> 16), 3); }
                            if (np > 4) {
                                const u32 st = __builtin_amdgcn_readlane(mv[8], rr);
                                for (int o = 4; o < np; o++) {
                                    const uint2 pl = plist[st + o];
                                    const int p = __builtin_amdgcn_readfirstlane(launder((int)pl.x)); const u32 lh = (u32)__builtin_amdgcn_readfirstlane(launder((int)pl.y));
                                    from_far(p, (int)(lh & 0xFFFF), (int)(lh >> 16), o);
                                }
                            }
                        }
                        }
                        // insertion chain: prefix maximum of (candidate - j*G), seeded by the cell left of the chunk
                        int run[C]; int acc = IDENT;
                        #pragma unroll
                        for (int c = 0; c < C; c++) {
                            const int j = jf + c;
                            if (j >= j0 && j <= hi) { const int t = max(dmax[c], umax[c]) - j * SG; acc = max(acc, t); }
                            run[c] = acc;
                        }
                        const int incl = wave_prefix_max(acc, IDENT);
                        int excl = __builtin_amdgcn_update_dpp(IDENT, incl, 0x138, 0xF, 0xF, false);     // wave_shr:1 (lane 0 keeps IDENT)
                        const int first = (k == k0) ? ((lo == 0 ? 0 : PNEG) - (j0 - 1) * SG) : (carry - (jbase - 1) * SG);
                        excl = max(excl, first);
                        int vv[C], ee[C];
                        #pragma unroll
                        for (int c = 0; c < C; c++) {
                            const int j = jf + c;
                            int v = 0, e = 3;
                            if (j >= j0 && j <= hi) {
                                const int mm = max(excl, run[c]);
                                v = max(mm + j * SG, PNEG);
                                e = dmax[c] == v ? (0 | (dd[c] << 2)) : (umax[c] == v ? (1 | (du[c] << 2)) : 2);
                            }
                            vv[c] = v; ee[c] = e;
                        }
                        if (sink || hi == L) {                                      // end cells: any column of a sink row, column L of any row
                            #pragma unroll
                            for (int c = 0; c < C; c++) {
                                const int j = jf + c, v = vv[c];
                                if (j >= lo && j <= hi && (sink || j == L)) {
                                    if (v > best_v) { best_v = v; best_i = i; best_j = j; multi = false; }
                                    else if (v == best_v && i != best_i) multi = true;
                                }
                            }
                        }
                        // ring (this wave's chunk of the row), back-pointers, spill copy, the value in column L
                        if constexpr (C == 1) myring[si + lane] = (int16_t)vv[0];
                        else if constexpr (C == 2) *reinterpret_cast<u32*>(myring + si + lane * 2) = ((u32)vv[0] & 0xFFFFu) | ((u32)vv[1] << 16);
                        else *reinterpret_cast<uint2*>(myring + si + lane * 4) = make_uint2(((u32)vv[0] & 0xFFFFu) | ((u32)vv[1] << 16), ((u32)vv[2] & 0xFFFFu) | ((u32)vv[3] << 16));
                        if (jf <= hi && jf + C - 1 >= lo) {
                            u8* dp = D + (size_t)i * STRIDE + (jf - dbase);
                            if constexpr (C == 1) dp[0] = (u8)ee[0];
                            else if constexpr (C == 2) *reinterpret_cast<u16*>(dp) = (u16)((u32)ee[0] | ((u32)ee[1] << 8));
                            else *reinterpret_cast<u32*>(dp) = (u32)ee[0] | ((u32)ee[1] << 8) | ((u32)ee[2] << 16) | ((u32)ee[3] << 24);
                            if (hi == L && jf <= L && L < jf + C) endval[i] = (int16_t)vv[L - jf];
                        }
                        if (spl) {
                            const u32 slot = __builtin_amdgcn_readlane(mv[9], rr);
                            int16_t* sp = spillH + (size_t)slot * STRIDE + (jf - dbase);
                            #pragma unroll
                            for (int c = 0; c < C; c++) sp[c] = (int16_t)vv[c];
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the copy has landed before any wave learns that the row is done
                        }
                    }
                    asm volatile("" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&S.done[w], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        stat_rows += (u32)N;
        __syncthreads();
        if constexpr (ROWS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // this CU's L1 may still hold the value rows of the read before
        PG_TICK(1);
        // ---- C: the end cell: maximum value, then the smallest row, then the smallest column
        {
            unsigned long long key = ((unsigned long long)(u32)(best_v + 32768) << 32) | (unsigned long long)(0xFFFFFFFFu - (((u32)best_i << 13) | (u32)best_j));
            #pragma unroll
            for (int s = 32; s >= 1; s >>= 1) { const unsigned long long o = __shfl_xor(key, s); key = o > key ? o : key; }
            if (lane == 0) S.red[w] = key;
            __syncthreads();
            unsigned long long g = S.red[0];
            #pragma unroll
            for (int x = 1; x < PW; x++) { const unsigned long long o = S.red[x]; g = o > g ? o : g; }
            const int gv = (int)(u32)(g >> 32) - 32768;
            const u32 gij = 0xFFFFFFFFu - (u32)(g & 0xFFFFFFFFu);
            const int gi = (int)(gij >> 13), gj = (int)(gij & 0x1FFF);
            const int tied = __syncthreads_or((N > 0 && best_v == gv && (multi || best_i != gi)) ? 1 : 0);
            if (tid == 0) { S.best_v = gv; S.best_i = gi; S.best_j = gj; }
            if (tied && gv > PNEG / 2) {
                // every row that holds the maximum in an end cell: column L (endval) or anywhere in a sink row (spill copy)
                for (int i = tid + 1; i <= N; i += PNT) {
                    const u32* mp = meta + (size_t)i * PMETA;
                    const u32 m0 = mp[0], m1 = mp[1];
                    const int lo = (int)(m0 & 0xFFFF), hi = (int)(m0 >> 16);
                    bool hit = false;
                    if constexpr (ROWS) {
                        const int16_t* src = reinterpret_cast<const int16_t*>(D) + (size_t)i * W;
                        if ((m1 >> 8) & 1) { for (int j = lo; j <= hi && !hit; j++) hit = (int)src[j % W] == gv; }
                        else if (hi == L) hit = (int)src[L % W] == gv;
                    } else
                    if ((m1 >> 8) & 1) {
                        const int16_t* src = spillH + (size_t)mp[9] * STRIDE; const int pb = (lo >> CSH) << CSH;
                        for (int j = lo; j <= hi && !hit; j++) hit = (int)src[j - pb] == gv;
                    } else if (hi == L) hit = (int)endval[i] == gv;
                    if (hit) { const u32 x = atomicAdd(&S.tie_cnt, 1u); if (x < (u32)PTIE) S.tie_rows[x] = (u32)i; }
                }
                __syncthreads();
                if (tid == 0) {
                    // resolved iff the first tied row is an ancestor of every other one through predecessor links among the tied rows
                    const u32 nt = S.tie_cnt;
                    bool ok = nt <= (u32)PTIE;
                    if (ok) {
                        for (u32 a = 1; a < nt; a++) { const u32 x = S.tie_rows[a]; int b = (int)a - 1; while (b >= 0 && S.tie_rows[b] > x) { S.tie_rows[b + 1] = S.tie_rows[b]; b--; } S.tie_rows[b + 1] = x; }
                        u32 reach = 1;                                              // bit a: tie_rows[a] descends from tie_rows[0]
                        for (u32 a = 1; a < nt && ok; a++) {
                            const u32* mp = meta + (size_t)S.tie_rows[a] * PMETA;
                            const u32 np = mp[1] >> 16;
                            bool found = false;
                            for (u32 o = 0; o < np && !found; o++) {
                                const u32 p = o < 4 ? ((mp[2 + (o >> 1)] >> (16 * (o & 1))) & 0xFFFF) : plist[mp[8] + o].x;
                                for (u32 b = 0; b < a; b++) if (((reach >> b) & 1) && S.tie_rows[b] == p) { found = true; break; }
                            }
                            if (found) reach |= 1u << a; else ok = false;
                        }
                        if (ok && (nt == 0 || (int)S.tie_rows[0] != gi)) ok = false;
                    }
                    if (!ok) S.status = 6;
                }
                stat_ties++;
            }
            __syncthreads();
            if (S.status) break;
        }
        PG_TICK(2);
        // ---- D (row engine): the moves are not stored; for 64 rows at a time, lane = row, the wave recomputes the moves of the 16 cells around the
        // column the path is expected in from the value rows (the same candidates, the same priorities: first predecessor reaching the value on the
        // diagonal, then on the vertical, then the insertion) and then walks them on the scalar unit -- readlane for a row's window, scalar shifts
        // and compares for the step, one lane of a register per path entry: 64 entries leave in one coalesced store
        if (ROWS && w == 0 && N > 0 && S.best_v > PNEG / 2) {
            const int16_t* V = reinterpret_cast<const int16_t*>(D);
            int i = S.best_i, j = S.best_j;
            const int jend = j;
            bool stop = false;
            int pbuf = 0, pk = 0, ptop = j;                                         // lane x of pbuf = alnrow[ptop - 1 - x]
            while (i > 0 && j > 0 && !stop) {
                const int ib = i;
                const int row = max(ib - lane, 1);
                const uint4* mp4 = reinterpret_cast<const uint4*>(meta + (size_t)row * PMETA);
                const uint4 qa = mp4[0], qb = mp4[1], qc = mp4[2];
                u32 q2 = qa.z, q3 = qa.w, q8 = qc.x;
                const int lo = (int)(qa.x & 0xFFFF), hi = (int)(qa.x >> 16), j0 = max(lo, 1);
                const int code = (int)(qa.y & 0xFF), np = (int)(qa.y >> 16);
                const int dlt = j - (int)__builtin_amdgcn_readlane((int)qc.z, 0);
                int wstart = ((int)qc.z + dlt - 8) & ~1;                            // even: a dword of the value row holds two cells
                u32 cw[4] = {0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u};   // 16 move bytes, 3 = no cell
                if (ib - lane >= 1) {
                    int dm[16], um[16]; u32 dd[16], du[16];
                    #pragma unroll
                    for (int x = 0; x < 16; x++) { dm[x] = PNEG; um[x] = PNEG; dd[x] = 0; du[x] = 0; }
                    // own values of the 16 cells
                    int own[16];
                    {
                        const u32* vr = reinterpret_cast<const u32*>(V + (size_t)row * W);
                        const int h0 = ((wstart % W) + W) % W;                       // even
                        #pragma unroll
                        for (int k2 = 0; k2 < 8; k2++) { const u32 x = vr[((h0 + 2 * k2) % W) >> 1]; own[2 * k2] = ((int)(x << 16)) >> 16; own[2 * k2 + 1] = ((int)x) >> 16; }
                    }
                    const int npe = np == 0 ? 1 : np;
                    for (int o = 0; o < npe; o++) {
                        int pr; u32 lh;
                        if (o < 4) { pr = (int)(((o < 2 ? q2 : q3) >> (16 * (o & 1))) & 0xFFFF); lh = o == 0 ? qb.x : (o == 1 ? qb.y : (o == 2 ? qb.z : qb.w)); }
                        else { const uint2 pl = plist[q8 + (u32)o]; pr = (int)pl.x; lh = pl.y; }
                        if (np == 0) pr = 0;
                        const int lop = (int)(lh & 0xFFFF), hip = (int)(lh >> 16);
                        int pvv[18];                                                // columns wstart - 2 .. wstart + 15
                        if (pr == 0) {
                            #pragma unroll
                            for (int x = 0; x < 18; x++) pvv[x] = 0;
                        } else {
                            const u32* vr = reinterpret_cast<const u32*>(V + (size_t)pr * W);
                            const int h0 = (((wstart - 2) % W) + W) % W;
                            #pragma unroll
                            for (int k2 = 0; k2 < 9; k2++) { const u32 x = vr[((h0 + 2 * k2) % W) >> 1]; pvv[2 * k2] = ((int)(x << 16)) >> 16; pvv[2 * k2 + 1] = ((int)x) >> 16; }
                        }
                        const int ra = max(j0, lop), rb = min(hi, hip + 1);
                        #pragma unroll
                        for (int x = 0; x < 16; x++) {
                            const int jj = wstart + x;
                            const int lv = (jj - 1 >= lop && jj - 1 <= hip) ? pvv[x + 1] : PNEG;
                            const int uv = (jj >= lop && jj <= hip) ? pvv[x + 2] : PNEG;
                            const bool in = jj >= ra && jj <= rb;
                            const int sc = (jj >= 1 && jj <= L && (int)sq[min(max(jj - 1, 0), L - 1)] == code) ? SM : SX;
                            const int d = in ? lv + sc : MININT, u = in ? uv + SG : MININT;
                            if (d > dm[x]) { dm[x] = d; dd[x] = (u32)o; }
                            if (u > um[x]) { um[x] = u; du[x] = (u32)o; }
                        }
                    }
                    #pragma unroll
                    for (int x = 0; x < 16; x++) {
                        const int jj = wstart + x;
                        u32 e = 3;
                        if (jj >= j0 && jj <= hi) e = dm[x] == own[x] ? (0u | (dd[x] << 2)) : (um[x] == own[x] ? (1u | (du[x] << 2)) : 2u);
                        cw[x >> 2] = (cw[x >> 2] & ~(0xFFu << (8 * (x & 3)))) | (e << (8 * (x & 3)));
                    }
                }
                // ---- the walk over this batch of rows
                while (i > 0 && j > 0) {
                    const int rr = ib - i;
                    if (rr > 63) break;
                    const int o = j - (int)__builtin_amdgcn_readlane(wstart, rr);
                    if (o < 0 || o > 15) break;                                     // the path left the row's window: recompute from here
                    const u32 w0 = __builtin_amdgcn_readlane((int)cw[0], rr), w1 = __builtin_amdgcn_readlane((int)cw[1], rr), w2 = __builtin_amdgcn_readlane((int)cw[2], rr), w3 = __builtin_amdgcn_readlane((int)cw[3], rr);
                    const u32 wsel = o < 8 ? (o < 4 ? w0 : w1) : (o < 12 ? w2 : w3);
                    const u32 e = (wsel >> (8 * (o & 3))) & 0xFF;
                    const int mvv = (int)(e & 3), ord = (int)(e >> 2);
                    if (mvv == 3) { stop = true; break; }
                    int ent = -1, p = i;
                    if (mvv == 2) ent = 0;
                    else {
                        if (ord < 2) p = (int)((__builtin_amdgcn_readlane((int)q2, rr) >> (16 * ord)) & 0xFFFF);
                        else if (ord < 4) p = (int)((__builtin_amdgcn_readlane((int)q3, rr) >> (16 * (ord - 2))) & 0xFFFF);
                        else p = __builtin_amdgcn_readfirstlane((int)plist[__builtin_amdgcn_readlane((int)q8, rr) + (u32)ord].x);
                        if (mvv == 0) ent = i;
                    }
                    if (ent >= 0) {                                                  // a sequence position is consumed: alnrow[j - 1] = ent
                        pbuf = lane == pk ? ent : pbuf;                             // lane pk collects the entry (a compare and a select; v_writelane takes one scalar operand only)
                        pk++; j--;
                        if (pk == 64) { alnrow[ptop - 1 - lane] = pbuf; ptop = j; pk = 0; }
                    }
                    i = p;
                }
            }
            if (pk > 0 && lane < pk) alnrow[ptop - 1 - lane] = pbuf;
            if (lane == 0 && j < jend) { S.has_aln = 1; S.fp = j; S.lp = jend - 1; }
        }
        // ---- D: traceback (wave 0): back-pointers of 64 rows at a time, 16 cells around the column the path is expected in
        if (!ROWS && w == 0 && N > 0 && S.best_v > PNEG / 2) {
            int i = S.best_i, j = S.best_j;
            const int jend = j;
            bool stop = false;
            while (i > 0 && j > 0 && !stop) {
                const int ib = i;
                const int row = ib - lane;
                u32 q2 = 0, q3 = 0, q8 = 0, qcol = 0; uint4 dw = make_uint4(0, 0, 0, 0); int wstart = 0;
                const u32* mp = meta + (size_t)max(row, 1) * PMETA;
                const u32 q0 = mp[0];
                q2 = mp[2]; q3 = mp[3]; q8 = mp[8]; qcol = mp[10];
                const int dlt = j - (int)__builtin_amdgcn_readlane(qcol, 0);
                if (row >= 1) {
                    if constexpr (WF) {
                        // trip-major back-pointers: the sixteen cells of two consecutive trips of the row's block, around the anti-diagonal the path is expected on
                        const int bq = (row - 1) >> 6, lq = (row - 1) & 63, a0 = blka0[bq];
                        const int t0 = min(max((row + (int)qcol + dlt - 8 - a0) >> 3, 0), TRX / 8 - 2);
                        const uint2* gp = reinterpret_cast<const uint2*>(D + (size_t)bq * ((size_t)TRX * 64) + ((size_t)t0 * 64 + lq) * 8);
                        const uint2 g0 = gp[0], g1 = gp[64];
                        dw = make_uint4(g0.x, g0.y, g1.x, g1.y);
                        wstart = a0 + 8 * t0 - row;
                    } else {
                        const int db = ((int)(q0 & 0xFFFF) >> CSH) << CSH;
                        int off = ((int)qcol + dlt - 8 - db) & ~3;
                        off = min(max(off, 0), STRIDE - 16);
                        wstart = db + off;
                        dw = *reinterpret_cast<const uint4*>(D + (size_t)row * STRIDE + off);
                    }
                }
                // as in the DP: registers the walk reads by v_readlane must not look like loads in flight (the walk stores one path entry per step)
                asm volatile("v_mov_b32 %0, %1" : "=v"(q2) : "v"(q2)); asm volatile("v_mov_b32 %0, %1" : "=v"(q3) : "v"(q3)); asm volatile("v_mov_b32 %0, %1" : "=v"(q8) : "v"(q8));
                asm volatile("v_mov_b32 %0, %1" : "=v"(dw.x) : "v"(dw.x)); asm volatile("v_mov_b32 %0, %1" : "=v"(dw.y) : "v"(dw.y)); asm volatile("v_mov_b32 %0, %1" : "=v"(dw.z) : "v"(dw.z)); asm volatile("v_mov_b32 %0, %1" : "=v"(dw.w) : "v"(dw.w));
                asm volatile("v_mov_b32 %0, %1" : "=v"(wstart) : "v"(wstart));
                // The walk, a RUN at a time.  Most steps are diagonal moves to the row before (the order keeps a chain's nodes adjacent), so every lane decodes the
                // move of ITS row in the column the path has there if it comes down from (i, j) by such steps alone -- column j - (i - row) -- and one ballot
                // gives the length of the run that really is of that kind; its entries leave in one store.  The cell that ends the run (a skipped sibling or
                // inserted row, a vertical or horizontal move) is in the lane after the run, already decoded: one general step from that lane's registers.
                while (i > 0 && j > 0) {
                    const int rr = ib - i;
                    if (rr > 63) break;
                    const int cR = j - (lane - rr);
                    const int o = cR - wstart;
                    const bool inwin = (u32)o < 16u;
                    const u32 dsel = o < 8 ? (o < 4 ? dw.x : dw.y) : (o < 12 ? dw.z : dw.w);
                    const u32 e = (dsel >> (8 * (o & 3))) & 0xFF;
                    const u32 c6 = 63u - (e & 63u);                                  // WF: the byte's low six bits are the winning candidate's priority: 63 - o diagonal, 47 - o vertical, 16 insertion, 0 no cell
                    const int mvv = WF ? (int)(c6 >> 4) : (int)(e & 3), ord = WF ? (int)(c6 & 15) : (int)(e >> 2);
                    const int p4 = (int)(((ord < 2 ? q2 : q3) >> (16 * (ord & 1))) & 0xFFFFu);   // the predecessor row for ord < 4
                    const bool cell = lane >= rr && row >= 1 && cR >= 1;              // this lane's row and column exist on the way down
                    const bool chain = cell && inwin && mvv == 0 && ord < 4 && p4 == row - 1;
                    const unsigned long long bm = __ballot(chain) >> rr;
                    const int run = ~bm ? __builtin_ctzll(~bm) : 64;
                    const int B = rr + run;                                         // the lane of the cell that ends the run
                    // that cell, decoded where it lives: where the path goes next, whether it consumes a column, and the three cases the scalar side must look at
                    const int consume = (mvv == 0 || mvv == 2) ? 1 : 0;
                    const int fl = !inwin ? 1 : (mvv == 3 ? 2 : ((mvv < 2 && ord >= 4) ? 4 : 0));   // left the window | no cell | a predecessor beyond the four of the descriptor
                    const int pk = (mvv == 2 ? row : p4) | (consume << 16) | (fl << 17);
                    const bool st_end = lane == B && cell && fl == 0 && consume;
                    if ((lane >= rr && lane < B) || st_end) alnrow[cR - 1] = (st_end && mvv == 2) ? 0 : row;   // the run's entries and the ending cell's, one store
                    i -= run; j -= run;
                    if (B > 63 || i <= 0 || j <= 0) continue;                       // the next batch of rows, or the end of the path
                    const int pB = __builtin_amdgcn_readlane(pk, B);
                    if (pB >> 17) {
                        if ((pB >> 17) & 1) break;                                  // the path left the row's window: fetch again from here
                        if ((pB >> 17) & 2) { stop = true; break; }
                        const int mvB = __builtin_amdgcn_readlane(mvv, B), ordB = __builtin_amdgcn_readlane(ord, B);
                        const int pl = __builtin_amdgcn_readfirstlane((int)plist[__builtin_amdgcn_readlane(q8, B) + (u32)ordB].x);
                        if (mvB == 0) { if (lane == 0) alnrow[j - 1] = i; j--; }
                        i = pl;
                        continue;
                    }
                    j -= (pB >> 16) & 1;
                    i = pB & 0xFFFF;
                }
            }
            if (lane == 0 && j < jend) { S.has_aln = 1; S.fp = j; S.lp = jend - 1; }
        }
        __syncthreads();
        PG_TICK(3);
        // ---- E: fuse the path into the graph, position-parallel
        const int has = S.has_aln, fp = S.fp, lp = S.lp;
        const int ppt = (L + PNT - 1) / PNT;                                       // consecutive positions per thread
        const int pa = min(L, tid * ppt), pb = min(L, pa + ppt);
        u32 my_new = 0;
        for (int p = pa; p < pb; p++) {                                            // E1: same node / aligned sibling / new node
            u8 kd; u32 cu = PNIL, an = 0;
            if (!has || p < fp || p > lp) kd = 1;                                  // unaligned end: chain node
            else {
                const int i = alnrow[p];
                if (i == 0) kd = 2;                                                // inserted base
                else {
                    const u32 nd = rank_cur[i - 1];
                    const NodeD d = ND[nd];
                    const u8 letter = sq[p];
                    // the block of the path node ends at the largest row of its aligned set
                    u32 er = (u32)i;
                    for (u32 a = 0; a < d.alcnt; a++) er = max(er, rowof[d.al[a]]);
                    an = er;
                    if (d.code == letter) { kd = 0; cu = nd; }
                    else {
                        kd = 3;                                                    // new sibling unless an aligned node carries the letter
                        for (u32 a = 0; a < d.alcnt; a++) if (ND[d.al[a]].code == letter) { kd = 0; cu = d.al[a]; break; }
                        if (kd == 3) { cu = nd; if (d.alcnt >= PAL) S.status = 3; }
                    }
                }
            }
            kindv[p] = kd; curv[p] = cu; anchor[p] = an;
            if (kd == 2 || kd == 3) my_new++;
        }
        u32 in_total;
        const u32 in_excl = block_excl_sum(my_new, S.scan, &in_total);             // new nodes inside the aligned range before this thread's positions
        const u32 n_pre = has ? (u32)fp : (u32)L, n_suf = has ? (u32)(L - 1 - lp) : 0u;
        const u32 total_new = n_pre + n_suf + in_total;
        if (n0 + total_new > job.ncap || n0 + total_new > 65535u) { if (tid == 0) S.status = 1; }
        // running maximum of the anchors (block end rows) of the positions before this thread's
        u32 my_anchor = 0;
        for (int p = pa; p < pb; p++) my_anchor = max(my_anchor, anchor[p]);
        u32 run_anchor = block_excl_max(my_anchor, S.scan);
        __syncthreads();
        if (S.status) break;
        {                                                                          // E2 + E3: ids, new nodes, fused positions
            u32 run = in_excl;
            for (int p = pa; p < pb; p++) {
                const u8 kd = kindv[p]; const u32 cu = curv[p];
                u32 id, before;                                                    // node of this position; new positions before it
                if (kd == 1) {
                    if (!has || p < fp) { id = n0 + (u32)p; before = (u32)p; }
                    else { id = n0 + n_pre + (u32)(p - lp - 1); before = n_pre + in_total + (u32)(p - lp - 1); }
                } else { before = n_pre + run; id = (kd == 0) ? cu : n0 + n_pre + n_suf + run; if (kd != 0) run++; }
                run_anchor = max(run_anchor, anchor[p]);
                anchor[p] = run_anchor; ncnt[p] = before;
                const u8 letter = sq[p];
                if (kd == 0) { atomicAdd(&NA[id].x, (u32)p + 1u); atomicAdd(&NA[id].y, 1u); }   // note_position (one position per node and read; the atomics only spare a read-modify-write of the record)
                else {
                    NA[id] = make_uint4((u32)p + 1u, 1u, 0u, 0u);
                    NB[id] = make_uint4(PNIL, PNIL, PNIL, PNIL);
                    NC[id] = make_uint4(PNIL, PNIL, PNIL, PNIL);
                    NodeD nn; nn.alcnt = 0; nn.code = letter; nn.pad = 0;
                    #pragma unroll
                    for (int a = 0; a < PAL; a++) nn.al[a] = 0;
                    if (kd == 3) {                                                 // new sibling of the path node cu: aligned = aligned[cu] + [cu], and everybody learns about it
                        const NodeD d = ND[cu];
                        for (u32 a = 0; a < d.alcnt; a++) { nn.al[nn.alcnt++] = d.al[a]; NodeD* o = &ND[d.al[a]]; o->al[o->alcnt] = (u16)id; o->alcnt++; }
                        nn.al[nn.alcnt++] = (u16)cu;
                        NodeD* o = &ND[cu]; o->al[o->alcnt] = (u16)id; o->alcnt++;
                    }
                    ND[id] = nn;
                }
                curv[p] = id;
            }
            if (tid == 0) ncnt[L] = total_new;
        }
        __syncthreads();
        // E4: edges (node of p-1 -> node of p), weight w[p-1] + w[p]; a node is tail of one and head of one position per read
        for (int p = max(pa, 1); p < pb; p++) {
            const u32 tail = curv[p - 1], head = curv[p];
            const u32 wgt = (u32)wg[p - 1] + (u32)wg[p];
            u32 e = NC[tail].z; bool found = false;
            while (e != PNIL) { const uint4 ed = EA[e]; if (ed.y == head) { EA[e].z = ed.z + wgt; found = true; break; } e = ed.w; }
            if (!found) {
                const u32 ne = atomicAdd(&S.n_edges, 1u);
                if (ne >= job.ecap) { S.status = 2; continue; }
                EA[ne] = make_uint4(tail, head, wgt, PNIL); enin[ne] = PNIL;
                u32* tc = reinterpret_cast<u32*>(&NC[tail]); u32* hc = reinterpret_cast<u32*>(&NC[head]);
                const u32 ot = tc[3];
                if (ot == PNIL) tc[2] = ne; else EA[ot].w = ne;
                tc[3] = ne;
                const u32 it = hc[1];
                if (it == PNIL) hc[0] = ne; else enin[it] = ne;
                hc[1] = ne;
                u32* ta = reinterpret_cast<u32*>(&NA[tail]); u32* ha = reinterpret_cast<u32*>(&NA[head]);
                ta[3] = ta[3] + 1;
                const u32 ic = ha[2]; ha[2] = ic + 1;
                if (ic < 4) reinterpret_cast<u32*>(&NB[head])[ic] = tail;
            }
        }
        __syncthreads();
        if (S.status) break;
        PG_TICK(4);
        // E5: splice the new nodes into the order: an old row i moves behind the new positions anchored before it; a new position p
        // goes behind its anchor block, after the new positions before it
        for (int i = tid + 1; i <= N; i += PNT) {
            int l = 0, h = L;
            while (l < h) { const int m = (l + h) >> 1; if (anchor[m] >= (u32)i) h = m; else l = m + 1; }
            rank_nxt[(u32)(i - 1) + ncnt[l]] = rank_cur[i - 1];
        }
        for (int p = pa; p < pb; p++) if (kindv[p] != 0) rank_nxt[anchor[p] + ncnt[p]] = curv[p];
        __syncthreads();
        if (tid == 0) { S.n_nodes = n0 + total_new; S.n_rows = (u32)N + total_new; }
        { u32* t = rank_cur; rank_cur = rank_nxt; rank_nxt = t; }
        __syncthreads();
        PG_TICK(5);
    }
    __syncthreads();
    // the final order is not exported (PoaGraph::consensus sorts the graph the way spoa does); counters per cluster
    const u32 ties = stat_ties;
    u32 far_tot; block_excl_sum(stat_far, S.scan, &far_tot);
    if (lane == 0) { S.scan[w] = stat_spins; S.tie_rows[w] = stat_tasks; }
    __syncthreads();
    if (tid == 0) { PoaGOut o; for (int x = 0; x < PW; x++) { o.spins[x] = S.scan[x]; o.tasks[x] = S.tie_rows[x]; } o.status = (int32_t)S.status; o.n_nodes = S.n_nodes; o.n_edges = S.n_edges; o.ties = ties; o.rows_done = stat_rows; o.tie_reads = stat_ties; o.far_rows = far_tot; o.pad = 0; for (int x = 0; x < 6; x++) o.ticks[x] = tk[x]; outs[blockIdx.x] = o; }
}

// K12's inputs from the RESIDENT reads (svt_poa_graphs_submit_reads): sequence s of the launch = read read_idx[s] of the batch, reverse-complemented when rev[s];
// letters are the 2-bit codes decoded ("ACGT": what the host builds from the normalised bases, src/types.rs:92-101), weights the read's 4-bit quality bins
// decoded as the host does (bin * 3 + 33 for the four bases of a bin, src/alignment.rs:248-273; 33 without qualities), in the same orientation
__global__ void k_poa_gather(BatchView bv, const u8* __restrict__ qualbins, const u64* __restrict__ qb_off, const u32* __restrict__ read_idx, const u8* __restrict__ rev,
                             const u64* __restrict__ seq_off, u32 n_seqs, u8* __restrict__ o_seq, u8* __restrict__ o_wts) {
    const u32 s = blockIdx.x;
    if (s >= n_seqs) return;
    const u32 r = read_idx[s];
    const u64 o = seq_off[s]; const u32 len = (u32)(seq_off[s + 1] - o);
    const u32* w = bv.packed + bv.woff[r];
    const u8* qb = qualbins ? qualbins + qb_off[r] : nullptr;
    const bool rc = rev && rev[s];
    for (u32 i = threadIdx.x; i < len; i += blockDim.x) {
        const u32 p = rc ? len - 1 - i : i;
        u32 code = (w[p >> 4] >> (30 - 2 * (p & 15))) & 3u;
        if (rc) code = 3u - code;
        o_seq[o + i] = (u8)"ACGT"[code];
        const u32 bin = p >> 2;
        o_wts[o + i] = qb ? (u8)(((qb[bin >> 1] >> (4 * (bin & 1))) & 15u) * 3u + 33u) : (u8)33;
    }
}
// K12c: the consensus of every finished graph on the device (PoaGraph::consensus of poa.hpp = spoa's heaviest bundle with branch completion, on spoa's own
// order: the depth-first topological sort that keeps aligned nodes adjacent, PoaGraph::topological_sort).  Both are sequential walks over a graph of a few
// thousand nodes: the workgroup stages the graph in LDS as 16-bit lists (all threads), then ONE lane runs the two walks there -- ~15 dependent LDS
// accesses per node, a few milliseconds per cluster, all clusters side by side, off the host (which spent 20 ms of CPU per 100k-read step on
// import_graph + consensus).  A graph that does not fit the LDS gets cons_len = 0xFFFFFFFF: the caller fetches the graph and runs its own consensus().
// Ties are decided exactly as the host decides them: in-edge, out-edge and aligned lists are in creation order on both sides (phase E).
__global__ void __launch_bounds__(256) k_poa_consensus(const PoaGJob* __restrict__ jobs, const u8* __restrict__ arenas, PoaGOut* __restrict__ outs, u32 stride, u32 lds_bytes, u8* __restrict__ cons) {
    extern __shared__ __attribute__((aligned(16))) u8 cl_raw[];
    const PoaGJob job = jobs[blockIdx.x];
    PoaGOut* O = outs + blockIdx.x;
    if (O->status != 0) { if (threadIdx.x == 0) O->pad = 0xFFFFFFFFu; return; }
    const u32 n = O->n_nodes, ne = O->n_edges;
    u64 cbase = 0;
    for (u32 x = 0; x < blockIdx.x; x++) cbase += jobs[x].ncap;                 // this cluster's slot in `cons`: node capacities of the clusters before it
    const PoaLay lay = poa_layout(job.ncap, job.ecap, job.lmax, stride);
    const u8* A = arenas + job.arena;
    const uint4* NC = (const uint4*)(A + lay.nc); const NodeD* ND = (const NodeD*)(A + lay.nd);
    const uint4* EA = (const uint4*)(A + lay.ea); const u32* enin = (const u32*)(A + lay.enin);
    // LDS: per node {in_head, out_head} u16, aligned[6] u16, alcnt / code / mark / chk u8, rank / pred / pos u16, score i64; per edge {tail, head, next_in, next_out} u16, weight u32; stack u16
    const u32 stack_cap = min(ne + 6 * n + 16, 8192u);                          // the walk's stack stays short in practice (node ids follow creation order: a node's predecessors are mostly marked already); a deeper one ends with the flag
    size_t o = 0;
    auto carve = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
    const size_t o_score = carve(8ull * n), o_w = carve(4ull * ne), o_inh = carve(2ull * n), o_outh = carve(2ull * n), o_al = carve(12ull * n), o_rank = carve(2ull * n), o_pred = carve(2ull * n),
                 o_pos = carve(2ull * n), o_tail = carve(2ull * ne), o_head = carve(2ull * ne), o_nin = carve(2ull * ne), o_nout = carve(2ull * ne), o_stack = carve(2ull * stack_cap),
                 o_alc = carve(n), o_code = carve(n), o_mark = carve(n), o_chk = carve(n);
    if (n == 0 || n > 65534 || ne > 65534 || o > lds_bytes) { if (threadIdx.x == 0) O->pad = n == 0 ? 0u : 0xFFFFFFFFu; return; }
    long long* score = (long long*)(cl_raw + o_score); u32* wgt = (u32*)(cl_raw + o_w);
    u16* inh = (u16*)(cl_raw + o_inh); u16* outh = (u16*)(cl_raw + o_outh); u16* al = (u16*)(cl_raw + o_al); u16* rank = (u16*)(cl_raw + o_rank); u16* pred = (u16*)(cl_raw + o_pred);
    u16* pos = (u16*)(cl_raw + o_pos); u16* tail = (u16*)(cl_raw + o_tail); u16* head = (u16*)(cl_raw + o_head); u16* nin = (u16*)(cl_raw + o_nin); u16* nout = (u16*)(cl_raw + o_nout);
    u16* stack = (u16*)(cl_raw + o_stack);
    u8* alc = cl_raw + o_alc; u8* code = cl_raw + o_code; u8* mark = cl_raw + o_mark; u8* chk = cl_raw + o_chk;
    constexpr u16 NIL = 0xFFFF;
    for (u32 v = threadIdx.x; v < n; v += blockDim.x) {
        const uint4 nc = NC[v]; const NodeD nd = ND[v];
        inh[v] = nc.x == PNIL ? NIL : (u16)nc.x; outh[v] = nc.z == PNIL ? NIL : (u16)nc.z;
        alc[v] = (u8)nd.alcnt; code[v] = nd.code; mark[v] = 0; chk[v] = 0;
        for (int k = 0; k < PAL; k++) al[v * 6 + k] = nd.al[k];
        score[v] = 0; pred[v] = NIL;
    }
    for (u32 e = threadIdx.x; e < ne; e += blockDim.x) {
        const uint4 ed = EA[e];
        tail[e] = (u16)ed.x; head[e] = (u16)ed.y; wgt[e] = ed.z; nout[e] = ed.w == PNIL ? NIL : (u16)ed.w;
        const u32 ni = enin[e]; nin[e] = ni == PNIL ? NIL : (u16)ni;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    // ---- PoaGraph::topological_sort: depth-first, a node after its predecessors and directly followed by its aligned siblings
    u32 nr = 0, sp = 0; bool bad = false;
    for (u32 s = 0; s < n && !bad; s++) {
        if (mark[s]) continue;
        stack[sp++] = (u16)s;
        while (sp) {
            const u32 c = stack[sp - 1];
            bool valid = true;
            if (mark[c] != 2) {
                if (sp + 8 + 64 >= stack_cap) { bad = true; break; }            // in-degrees are below 64 (status 9 otherwise)
                for (u16 e = inh[c]; e != NIL; e = nin[e]) { const u32 t_ = tail[e]; if (mark[t_] != 2) { stack[sp++] = (u16)t_; valid = false; } }
                const u32 na = alc[c];
                if (na && !chk[c]) for (u32 k = 0; k < na; k++) { const u32 a = al[c * 6 + k]; if (mark[a] != 2) { stack[sp++] = (u16)a; chk[a] = 1; valid = false; } }
                if (valid) {
                    mark[c] = 2;
                    if (!chk[c]) { rank[nr++] = (u16)c; for (u32 k = 0; k < na; k++) rank[nr++] = al[c * 6 + k]; }
                } else mark[c] = 1;
            }
            if (valid) sp--;
        }
    }
    if (bad || nr != n) { O->pad = 0xFFFFFFFFu; return; }
    // ---- PoaGraph::consensus: heaviest bundle ...
    int mx = -1;
    for (u32 i = 0; i < n; i++) {
        const u32 v = rank[i];
        long long sv = 0; u32 pv = NIL;                                         // score[v] / pred[v] while the in-edges are relaxed
        for (u16 e = inh[v]; e != NIL; e = nin[e]) {
            const u32 t_ = tail[e];
            if (score[t_] < 0) continue;
            const long long w_ = (long long)wgt[e];
            if (sv < w_ || (sv == w_ && pv != NIL && score[pv] <= score[t_])) { sv = w_; pv = t_; }
        }
        if (pv != NIL) sv += score[pv];
        score[v] = sv; pred[v] = (u16)pv;
        if (mx < 0 || score[mx] < score[v]) mx = (int)v;
    }
    // ... and branch completion: extend the heaviest path to a sink
    for (u32 i = 0; i < n; i++) pos[rank[i]] = (u16)i;
    while (outh[mx] != NIL) {
        for (u16 e = outh[mx]; e != NIL; e = nout[e]) for (u16 e2 = inh[head[e]]; e2 != NIL; e2 = nin[e2]) if ((int)tail[e2] != mx) score[tail[e2]] = -1;
        int nmx = -1; long long best = 0;
        for (u32 i = (u32)pos[mx] + 1; i < n; i++) {
            const u32 v = rank[i];
            score[v] = -1; pred[v] = NIL;
            long long sv = -1; u32 pv = NIL;
            for (u16 e = inh[v]; e != NIL; e = nin[e]) {
                const u32 t_ = tail[e];
                if (score[t_] == -1) continue;
                const long long w_ = (long long)wgt[e];
                if (sv < w_ || (sv == w_ && pv != NIL && score[pv] <= score[t_])) { sv = w_; pv = t_; }
            }
            if (pv != NIL) { score[v] = sv + score[pv]; pred[v] = (u16)pv; if (nmx < 0 || best < score[v]) { nmx = (int)v; best = score[v]; } }
        }
        if (nmx < 0) break;
        mx = nmx;
    }
    u32 len = 0;
    for (u32 v = (u32)mx; v != NIL; v = pred[v]) len++;
    u8* dst = cons + cbase;
    u32 k = len;
    for (u32 v = (u32)mx; v != NIL; v = pred[v]) dst[--k] = code[v];
    O->pad = len;
}

// compact the final graphs: per cluster nodes [node_off[c], +n_nodes) and edges [edge_off[c], +n_edges)
__global__ void k_poa_graph_export(const PoaGJob* __restrict__ jobs, const u8* __restrict__ arenas, const PoaGOut* __restrict__ outs, const u64* __restrict__ node_off, const u64* __restrict__ edge_off,
                                   u32 stride, u8* __restrict__ o_code, u16* __restrict__ o_al, u32* __restrict__ o_edge) {
    const PoaGJob job = jobs[blockIdx.x];
    const PoaGOut o = outs[blockIdx.x];
    if (o.status != 0) return;
    const PoaLay lay = poa_layout(job.ncap, job.ecap, job.lmax, stride);
    const u8* A = arenas + job.arena;
    const NodeD* ND = (const NodeD*)(A + lay.nd); const uint4* EA = (const uint4*)(A + lay.ea);
    for (u32 v = threadIdx.x; v < o.n_nodes; v += blockDim.x) {
        const NodeD d = ND[v];
        o_code[node_off[blockIdx.x] + v] = d.code;
        u16* al = o_al + (node_off[blockIdx.x] + v) * 8;
        al[0] = d.alcnt;
        for (int a = 0; a < PAL; a++) al[1 + a] = d.al[a];
        al[7] = 0;
    }
    for (u32 e = threadIdx.x; e < o.n_edges; e += blockDim.x) {
        const uint4 ed = EA[e];
        u32* dst = o_edge + (edge_off[blockIdx.x] + e) * 3;
        dst[0] = ed.x; dst[1] = ed.y; dst[2] = ed.z;
    }
}

template <int C> size_t poa_graph_lds(u32 lmax) { return (size_t)PW * PCfg<C>::R * PCfg<C>::CW * 2 + job_lmax_pad(lmax); }
template <int C> size_t poa_wf_lds(u32 lmax) { return (size_t)WfCfg<C>::SQ + job_lmax_pad(lmax) + 256; }   // + 256: the sweep reads the base of columns up to 70 beyond the sequence for lanes outside their band
template <int C> size_t poa_rows_lds(u32 lmax) { return (size_t)64 * 64 * C * 2 + job_lmax_pad(lmax) + 4 * ((size_t)lmax / C + 4); }

}  // namespace

int launch_poa_consensus(svt_ctx* c, int C, u32 n_clusters, const void* d_jobs, const u8* d_arenas, void* d_outs, u8* d_cons) {
    if (n_clusters == 0) return SVT_OK;
    const u32 lds = 156 * 1024;                                                 // a workgroup may hold 160 KB on gfx950; one per CU is plenty for one lane of work each
    HIPCHK(c, hipFuncSetAttribute((const void*)k_poa_consensus, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   // per DEVICE in HIP, and contexts of several devices / threads share this code: set on every launch, as the DP launches do
    ProfScope ps(c, "k_poa_consensus", 0.0, (double)n_clusters);
    hipLaunchKernelGGL(k_poa_consensus, dim3(n_clusters), dim3(256), lds, c->stream, (const PoaGJob*)d_jobs, d_arenas, (PoaGOut*)d_outs, poa_graph_stride(C), lds, d_cons);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
int launch_poa_gather(svt_ctx* c, const svt_batch* B, const u32* d_read_idx, const u8* d_rev, const u64* d_seq_off, u32 n_seqs, u8* d_seq, u8* d_wts, double bytes) {
    if (n_seqs == 0) return SVT_OK;
    ProfScope ps(c, "k_poa_gather", bytes * 2.5, (double)n_seqs);
    hipLaunchKernelGGL(k_poa_gather, dim3(n_seqs), dim3(256), 0, c->stream, B->view(), B->seeds.valid ? B->seeds.qualbins : nullptr, B->seeds.valid ? B->seeds.qb_off : nullptr, d_read_idx, d_rev, d_seq_off, n_seqs, d_seq, d_wts);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

// C = 1, 2, 4: the chunk pipeline with C cells per lane and chunk; C = 106, 108: the row engine with 6 / 8 cells per lane (W = 384 / 512 columns per row)
u32 poa_graph_stride(int C) { return C >= 200 ? (u32)(PW * 64 * (C - 200)) : (C >= 100 ? (u32)(2 * 64 * (C - 100)) : (u32)(PW * 64 * C)); }
u64 poa_graph_arena_bytes(u32 ncap, u32 ecap, u32 lmax, int C) { return poa_layout(ncap, ecap, lmax, poa_graph_stride(C)).total; }
size_t poa_graph_job_bytes() { return sizeof(PoaGJob); }
size_t poa_graph_out_bytes() { return sizeof(PoaGOut); }
int poa_graph_max_band(int C) { return C >= 200 ? 160 * (C - 200) : (C >= 100 ? (64 * (C - 100) - (C - 100) - 1) / 2 : 160 * C); }   // chunk pipeline: a band of 2*bw+1 columns touches at most 7 of the 8 chunks; row engine: 2 bw + 1 <= W - C

int launch_poa_graph(svt_ctx* c, int C, u32 n_clusters, u32 lmax, const void* d_jobs, u8* d_arenas, const u8* d_seqs, const u8* d_wts, const u64* d_seq_off, const u32* d_band, void* d_outs, double cells) {
    if (n_clusters == 0) return SVT_OK;
    ProfScope ps(c, C >= 200 ? "k_poa_diag" : (C >= 100 ? "k_poa_rows" : "k_poa_graph"), cells * (C >= 100 && C < 200 ? 2.0 : 1.0), 0.0);   // bytes: one back-pointer byte (chunk pipeline) or one int16 value (row engine) per band cell; units (graph rows) are added by svt_poa_graphs_wait
    #define PG_LAUNCH(CC) do { \
        const size_t sh = poa_graph_lds<CC>(lmax); \
        HIPCHK(c, hipFuncSetAttribute((const void*)k_poa_graph<CC, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
        hipLaunchKernelGGL((k_poa_graph<CC, 0>), dim3(n_clusters), dim3(PNT), sh, c->stream, (const PoaGJob*)d_jobs, d_arenas, d_seqs, d_wts, d_seq_off, d_band, (PoaGOut*)d_outs); } while (0)
    #define PR_LAUNCH(CC) do { \
        const size_t sh = poa_rows_lds<CC>(lmax); \
        HIPCHK(c, hipFuncSetAttribute((const void*)k_poa_graph<CC, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
        hipLaunchKernelGGL((k_poa_graph<CC, 1>), dim3(n_clusters), dim3(PNT), sh, c->stream, (const PoaGJob*)d_jobs, d_arenas, d_seqs, d_wts, d_seq_off, d_band, (PoaGOut*)d_outs); } while (0)
    #define PD_LAUNCH(CC) do { \
        const size_t sh = poa_wf_lds<CC>(lmax); \
        HIPCHK(c, hipFuncSetAttribute((const void*)k_poa_graph<CC, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
        hipLaunchKernelGGL((k_poa_graph<CC, 2>), dim3(n_clusters), dim3(PNT), sh, c->stream, (const PoaGJob*)d_jobs, d_arenas, d_seqs, d_wts, d_seq_off, d_band, (PoaGOut*)d_outs); } while (0)
    if (C == 201) PD_LAUNCH(1); else if (C == 202) PD_LAUNCH(2); else if (C == 204) PD_LAUNCH(4); else
    if (C == 106) PR_LAUNCH(6); else if (C == 108) PR_LAUNCH(8); else if (C == 1) PG_LAUNCH(1); else if (C == 2) PG_LAUNCH(2); else PG_LAUNCH(4);
    #undef PG_LAUNCH
    #undef PR_LAUNCH
    #undef PD_LAUNCH
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
int launch_poa_graph_export(svt_ctx* c, int C, u32 n_clusters, const void* d_jobs, const u8* d_arenas, const void* d_outs, const u64* d_node_off, const u64* d_edge_off, u8* o_code, u16* o_al, u32* o_edge) {
    if (n_clusters == 0) return SVT_OK;
    hipLaunchKernelGGL(k_poa_graph_export, dim3(n_clusters), dim3(256), 0, c->stream, (const PoaGJob*)d_jobs, d_arenas, (const PoaGOut*)d_outs, d_node_off, d_edge_off, poa_graph_stride(C), o_code, o_al, o_edge);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}
