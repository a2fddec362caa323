// capi.hip -- implementation of include/savont_hip.h: context, HBM-resident batches, entry points.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: the library binds RCCL with dlopen (rccl_api below), it is not a link dependency
#include <ctime>
#include "svt_internal.hpp"

// ------------------------------------------------------------------------------------------------
// small infrastructure
// ------------------------------------------------------------------------------------------------
int svt_fail(svt_ctx* c, int code, const std::string& msg) {
    if (c && c->sh_failed && code == SVT_ERR_HIP) { c->err = "shard exchange failed (" + c->sh_fail_why + "); " + msg; return SVT_ERR_EXCHANGE; }   // the stream error behind an aborted collective
    if (c) c->err = msg;
    return code;
}

// two HIP events per launch are not free: at 100k reads a step is ~180 launches, and creating + recording their events was a tenth of the host CPU of a
// step (sampled at 2 CPUs per rank).  Level 2 of svt_profile_enable times only the kernels a roofline is quoted for -- the POA engine, the affine aligner's
// span and its forward pass -- and the events are reused from a per-context pool.
static bool prof_heavy(const char* name) { return !strncmp(name, "k_poa_", 6) || !strncmp(name, "k_align_affine", 14) || !strncmp(name, "k_align_end", 11); }
static hipEvent_t prof_event(svt_ctx* c) {
    if (!c->prof_events.empty()) { hipEvent_t e = c->prof_events.back(); c->prof_events.pop_back(); return e; }
    hipEvent_t e = nullptr; hipEventCreate(&e); return e;
}
ProfScope::ProfScope(svt_ctx* ctx, const char* name, double bytes, double units, hipStream_t on) : c(ctx), st(on ? on : ctx->stream) {
    if (!c->profiling()) return;
    if ((c->parent ? c->parent->prof_level : c->prof_level) == 2 && !prof_heavy(name)) return;
    for (size_t i = 0; i < c->prof_entries.size(); i++) if (c->prof_entries[i].name == name) idx = (int)i;
    if (idx < 0) { c->prof_entries.push_back(ProfEntry()); idx = (int)c->prof_entries.size() - 1; c->prof_entries[idx].name = name; }
    c->prof_entries[idx].launches++; c->prof_entries[idx].bytes += bytes; c->prof_entries[idx].units += units;
    a = prof_event(c); b = prof_event(c);
    hipEventRecord(a, st);
}
ProfScope::~ProfScope() {
    if (idx < 0) return;
    hipEventRecord(b, st);
    c->pending.push_back(PendingEvt{idx, a, b});
}
void prof_add_bytes(svt_ctx* c, const char* name, double bytes) {
    if (!c->profiling()) return;
    for (auto& e : c->prof_entries) if (e.name == name) { e.bytes += bytes; return; }
}
void prof_add_units(svt_ctx* c, const char* name, double units) {          // units known only after the launch (graph rows of K12)
    if (!c->profiling()) return;
    for (auto& e : c->prof_entries) if (e.name == name) { e.units += units; return; }
}
void prof_note_units(svt_ctx* c, const char* name, double units) {         // a line of its own without a launch: units a launch covers, split by kind (the band classes of K8a's one launch)
    if (!c->profiling()) return;
    svt_ctx* root = c;
    for (auto& e : root->prof_entries) if (e.name == name) { e.units += units; return; }
    ProfEntry e; e.name = name; e.units = units; root->prof_entries.push_back(e);
}
static void prof_drain_one(svt_ctx* c) {
    for (auto& p : c->pending) {
        hipEventSynchronize(p.b);
        float ms = 0; hipEventElapsedTime(&ms, p.a, p.b);
        c->prof_entries[p.idx].ms += ms;
        if (c->prof_events.size() < 1024) { c->prof_events.push_back(p.a); c->prof_events.push_back(p.b); } else { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    }
    c->pending.clear();
}
// the launches of a context's forks are folded into it (call only while the forks are idle)
static void prof_drain(svt_ctx* c) {
    prof_drain_one(c);
    for (svt_ctx* f : c->forks) {
        prof_drain_one(f);
        for (auto& e : f->prof_entries) {
            int idx = -1;
            for (size_t i = 0; i < c->prof_entries.size(); i++) if (c->prof_entries[i].name == e.name) idx = (int)i;
            if (idx < 0) { c->prof_entries.push_back(ProfEntry()); idx = (int)c->prof_entries.size() - 1; c->prof_entries[idx].name = e.name; }
            c->prof_entries[idx].launches += e.launches; c->prof_entries[idx].ms += e.ms; c->prof_entries[idx].bytes += e.bytes; c->prof_entries[idx].units += e.units;
        }
        f->prof_entries.clear();
    }
}

// bump allocator over one reusable device buffer (reset at the start of every API call that uses it)
struct Arena {
    svt_ctx* c; size_t used = 0; std::vector<std::pair<size_t, size_t>> req;
    explicit Arena(svt_ctx* ctx) : c(ctx) {}
};
// Wait for the context's stream: hipStreamSynchronize (spins on the host, lowest latency) by default; SAVONT_SYNC=block waits on a
// blocking event instead (interrupt, no spinning) -- for hosts where the ~24 waiting threads of Stage 3 are short of CPU time.
// Measured on the 16-CPU-quota boxes: no difference (cgroup cpu.stat shows no throttling during a run).
// Waiting for the stream.  hipStreamSynchronize spins: the host thread burns a core for as long as the kernels run -- and so does
// hipEventSynchronize on a hipEventBlockingSync event (tools/micro/sync_cost.hip: 200 ms of CPU for 200 ms of kernel either way; only
// hipSetDeviceFlags(hipDeviceScheduleBlockingSync) makes the runtime sleep, and with five pipelines on one device that setting hung the
// bench).  "sync_block" therefore polls: a short spin for the calls that return in microseconds, then hipStreamQuery between 40 us sleeps.
// With samples in flight the cores belong to the other samples' host work meanwhile; a lone sample pays <= 40 us per wait.
static hipError_t ctx_sync_wait(svt_ctx* c);
static hipError_t ctx_sync(svt_ctx* c) { const hipError_t e = ctx_sync_wait(c); c->pk_busy[0] = c->pk_busy[1] = false; return e; }
static void shard_comm_abort(svt_ctx* c, const char* why);
static hipError_t ctx_sync_wait(svt_ctx* c) {
    if (c->sh_inflight && c->sh_comm) {
        // a collective of the shard communicator is on this stream: a peer that never joins it (died, returned early, issued another collective) would keep the
        // kernel -- and this wait -- alive for ever.  Poll with a deadline ("shard_timeout_s"); past it the communicator is ABORTED (ncclCommAbort: the collective's
        // kernels see the flag and leave, here and -- through their own deadline -- on the peers) and the wait ends with an error that svt_fail reports as SVT_ERR_EXCHANGE.
        // The clock starts when everything queued BEFORE the collective has completed (sh_mark: an event recorded right in front of the first collective since the last wait):
        // kernels of this rank that precede it -- a 1 M-read step's Stage 3, say -- are not "no progress in a collective" (ADVICE r05).  What the deadline still covers beside a
        // dead peer is a peer that is merely LATE (its host still in the POA of its clusters): "shard_timeout_s" (default 180) must exceed the largest imbalance between ranks.
        auto t0 = std::chrono::steady_clock::now();
        bool started = c->sh_mark == nullptr || !c->sh_mark_set;
        const double limit = (double)std::max(1, c->opt().shard_timeout_s);
        long ns = 20000;
        for (int polls = 0;; polls++) {
            const hipError_t e = hipStreamQuery(c->stream);
            if (e != hipErrorNotReady) { c->sh_inflight = false; c->sh_mark_set = false; return e; }
            if (!started && hipEventQuery(c->sh_mark) != hipErrorNotReady) { started = true; t0 = std::chrono::steady_clock::now(); }
            if (started && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
                c->sh_mark_set = false;
                shard_comm_abort(c, "no progress in a grouped collective within shard_timeout_s: a peer rank did not join it");
                hipStreamSynchronize(c->stream);                    // the aborted kernels leave
                (void)hipGetLastError();
                return hipErrorLaunchTimeOut;
            }
            timespec ts{0, ns}; nanosleep(&ts, nullptr);
            if (polls >= 32 && ns < 1000000) ns *= 2;
        }
    }
    if (!c->opt().sync_block) return hipStreamSynchronize(c->stream);
    for (int spin = 0; spin < 64; spin++) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e != hipErrorNotReady) return e;
    }
    // a short wait costs at most 40 us of extra latency; a long one (the K12 launch runs 100-200 ms) backs off to 1 ms between polls -- every
    // hipStreamQuery is ~10 us of CPU, and at 40 us per poll a waiting pipeline burned a third of a core (0.2 CPU-s per step with K12)
    long ns = 40000;
    for (int polls = 0;; polls++) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e != hipErrorNotReady) return e;
        timespec ts{0, ns};
        nanosleep(&ts, nullptr);
        if (polls >= 32 && ns < 1000000) ns *= 2;
    }
}
// A copy to HOST memory the caller owns (pageable unless the caller pinned it).  The runtime makes such a copy wait for the stream on the calling thread, SPINNING -- for as
// long as the kernels queued before it run.  Under "sync_block" (samples in flight share the host cores) the stream is therefore waited for first, the polite way; what is
// left to spin through is the copy itself.  (Measured: the 4-byte overflow flag of the counting pass alone cost 4 % of a 2-CPU step this way.)
static hipError_t memcpy_d2h(svt_ctx* c, void* dst, const void* src, size_t bytes) {
    if (c->opt().sync_block) { const hipError_t e = ctx_sync_wait(c); if (e != hipSuccess) return e; }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
}
// Waiting for a launch that is known to run for tens of milliseconds (K12).  Under "sync_block" the first hipStreamQuery that finds the stream busy leaves a thread
// of the HIP runtime spinning on a core until the awaited kernel ends, however long this thread sleeps between polls (tools/thread_cpu.py: 80 ms of CPU per K12 launch
// on a thread that is not ours; profiles/r04_poa.md).  So no runtime call while waiting: a one-lane kernel behind the work sets a word in page-locked host memory and
// the host looks at it between sleeps.  Short waits keep ctx_sync: the extra launch and the sleep granularity cost them more than the query does.
__global__ void k_sync_word(volatile unsigned* w, unsigned v) { *w = v; __threadfence_system(); }
static hipError_t ctx_sync_long(svt_ctx* c) {
    if (!c->opt().sync_block) return ctx_sync(c);
    if (!c->sync_word) {
        if (hipHostMalloc((void**)&c->sync_word, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { c->sync_word = nullptr; (void)hipGetLastError(); return ctx_sync(c); }
        *c->sync_word = 0;
    }
    const unsigned v = ++c->sync_seq;
    hipLaunchKernelGGL(k_sync_word, dim3(1), dim3(1), 0, c->stream, (volatile unsigned*)c->sync_word, v);
    if (hipGetLastError() != hipSuccess) return ctx_sync(c);
    long ns = 50000; double waited = 0;
    while (*(volatile unsigned*)c->sync_word != v && waited < 20.0) {          // 20 s: a fault in the awaited kernel never sets the word; ctx_sync below reports it
        timespec ts{0, ns}; nanosleep(&ts, nullptr);
        waited += 1e-9 * (double)ns;
        if (ns < 400000) ns += 50000;
    }
    return ctx_sync(c);                                                        // the stream is idle: returns at once, with the launch's error if there was one
}
static bool ensure_scratch(svt_ctx* c, size_t bytes) {
    if (bytes <= c->scratch_bytes) return true;
    if (c->scratch) { ctx_sync(c); hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
    // hipFree / hipMalloc synchronise the whole device (7-10 ms each while other streams are busy): grow generously, and give the
    // forked contexts of the parallel stages a floor so that a fork that meets a larger group later does not stall everyone
    size_t want = bytes + bytes / 2 + (1 << 20);
    if (c->parent && want < ((size_t)64 << 20)) want = (size_t)64 << 20;
    if (hipMalloc(&c->scratch, want) != hipSuccess) { c->scratch = nullptr; return false; }
    c->scratch_bytes = want;
    return true;
}
// Pinned staging for the many small calls of the greedy stages is OPT-IN (SAVONT_PIN=1).  Copies from / to pinned memory go through the
// SDMA engines, and an SDMA engine that sat idle for >~10 ms (the host-only stretches between the stages) takes 17-26 ms to start its
// next copy on this platform (rocprofv3 memory-copy trace: the first Stage-2 call of a step stalled that long in every other step);
// copies from pageable memory are staged by the runtime and run as blit kernels, which start at once.
static bool ensure_pinned(svt_ctx* c, size_t bytes) {
    if (!c->opt().pin_staging) return false;
    if (bytes <= c->pin_bytes) return true;
    if (bytes > ((size_t)64 << 20)) return false;                                // large transfers keep the direct path
    if (c->pin) { ctx_sync(c); hipHostFree(c->pin); c->pin = nullptr; c->pin_bytes = 0; }
    size_t want = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 20);
    if (hipHostMalloc(&c->pin, want, hipHostMallocDefault) != hipSuccess) { c->pin = nullptr; return false; }
    c->pin_bytes = want;
    return true;
}
// pinned, device-visible host buffer that kernels of small calls read and write in place (zero copy)
static bool ensure_zero_copy(svt_ctx* c, size_t bytes) {
    if (!c->opt().zero_copy) return false;
    if (bytes <= c->zc_bytes) return true;
    if (c->zc) { ctx_sync(c); hipHostFree(c->zc); c->zc = nullptr; c->zc_bytes = 0; }
    size_t want = std::max<size_t>(bytes + bytes / 2, (size_t)4 << 20);
    if (hipHostMalloc(&c->zc, want, hipHostMallocDefault) != hipSuccess) { c->zc = nullptr; return false; }
    c->zc_bytes = want;
    return true;
}
void* svt_scratch(svt_ctx* c, size_t bytes) { return ensure_scratch(c, bytes) ? c->scratch : nullptr; }
// carve sub-buffers out of the scratch: sizes first, then pointers
struct Carve {
    std::vector<size_t> offs; size_t total = 0;
    size_t add(size_t bytes) { size_t o = total; offs.push_back(o); total += (bytes + 255) & ~(size_t)255; return offs.size() - 1; }
};
template <class T> static T* carve_ptr(svt_ctx* c, const Carve& cv, size_t id) { return (T*)((char*)c->scratch + cv.offs[id]); }

// Small host arrays travel together: an hipMemcpyAsync costs ~10 us of runtime work whatever its size, and a step of the pipeline issued ~200
// of them.  UpPack stages the arrays of one call -- whose destinations were carved next to each other -- in one host buffer laid out as the
// carve is and sends it with ONE copy; DownPack fetches carve regions next to each other with ONE copy and hands the pieces out after the
// sync.  Both own their staging memory: keep them alive until the stream has been synchronised.
// pinned staging buffer of direction `dir` (0 up, 1 down), at least `bytes`, not in flight; nullptr -> the pack falls back to pageable memory
static char* pack_stage(svt_ctx* c, int dir, size_t bytes) {
    if (bytes > ((size_t)256 << 20)) return nullptr;
    if (c->pk_busy[dir]) ctx_sync(c);                                              // a second pack of the same direction before the call's sync
    if (bytes > c->pk_bytes[dir]) {
        if (c->pk[dir]) { ctx_sync(c); hipHostFree(c->pk[dir]); c->pk[dir] = nullptr; c->pk_bytes[dir] = 0; }
        const size_t want = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 20);
        if (hipHostMalloc(&c->pk[dir], want, hipHostMallocDefault) != hipSuccess) { c->pk[dir] = nullptr; return nullptr; }
        c->pk_bytes[dir] = want;
    }
    c->pk_busy[dir] = true;
    return (char*)c->pk[dir];
}
struct UpPack {
    svt_ctx* c; const Carve& cv; size_t lo = (size_t)-1, hi = 0;
    struct Item { size_t off; const void* src; size_t bytes; };
    std::vector<Item> items; std::vector<char> stage;
    UpPack(svt_ctx* c_, const Carve& cv_) : c(c_), cv(cv_) {}
    void put(size_t id, const void* src, size_t bytes) { if (!bytes || !src) return; items.push_back({cv.offs[id], src, bytes}); lo = std::min(lo, cv.offs[id]); hi = std::max(hi, cv.offs[id] + bytes); }
    hipError_t send() {
        if (items.empty()) return hipSuccess;
        char* st = pack_stage(c, 0, hi - lo);                                      // pinned: one DMA, not a chain of staged blits
        if (!st) { if (items.size() == 1) return hipMemcpyAsync((char*)c->scratch + items[0].off, items[0].src, items[0].bytes, hipMemcpyHostToDevice, c->stream); stage.resize(hi - lo); st = stage.data(); }
        for (const Item& it : items) memcpy(st + (it.off - lo), it.src, it.bytes);
        return hipMemcpyAsync((char*)c->scratch + lo, st, hi - lo, hipMemcpyHostToDevice, c->stream);
    }
};
// A counter the host needs before it can go on (list lengths, cursors): a one-wave kernel stores it into pinned, device-visible host memory --
// no copy engine, no blit kernel -- and the host reads it after the sync.  Falls back to a copy when zero-copy I/O is off.
__global__ void k_peek(const u32* __restrict__ src, u32* __restrict__ dst, u32 n_words) { if (threadIdx.x < n_words) dst[threadIdx.x] = src[threadIdx.x]; }
static hipError_t peek(svt_ctx* c, const void* dsrc, void* hdst, size_t bytes) {        // bytes <= 256, a multiple of 4; synchronises the stream
    if (bytes <= 256 && ensure_zero_copy(c, 4096)) {
        u32* slot = (u32*)((char*)c->zc + c->zc_bytes - 256);                          // the tail of the zero-copy buffer: small calls fill it from the front
        hipLaunchKernelGGL(k_peek, dim3(1), dim3(64), 0, c->stream, (const u32*)dsrc, slot, (u32)(bytes / 4));
        const hipError_t e = ctx_sync(c);
        if (e == hipSuccess) memcpy(hdst, slot, bytes);
        return e;
    }
    const hipError_t e = memcpy_d2h(c, hdst, dsrc, bytes);
    return e != hipSuccess ? e : ctx_sync(c);
}
// a block the caller lays out itself (descriptors of one dmalloc'd allocation): filled in pinned memory, sent with one copy
struct StageUp {
    svt_ctx* c; char* p; std::vector<char> fb;
    StageUp(svt_ctx* c_, size_t bytes) : c(c_) { p = pack_stage(c, 0, bytes); if (!p) { fb.resize(bytes); p = fb.data(); } }
    hipError_t send(void* dst, size_t bytes) { return hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, c->stream); }
};
struct DownPack {
    svt_ctx* c; size_t lo = (size_t)-1, hi = 0;
    struct Item { const char* src; void* dst; size_t bytes; };
    std::vector<Item> items; std::vector<char> stage; const char* base = nullptr; char* st = nullptr; bool direct = false;
    explicit DownPack(svt_ctx* c_) : c(c_) {}
    void get(const void* dsrc, void* dst, size_t bytes) { if (!bytes || !dst) return; items.push_back({(const char*)dsrc, dst, bytes}); }
    // the regions must lie in ONE device allocation (the scratch, or one dmalloc block); what lies between them is fetched too
    hipError_t recv() {
        if (items.empty()) return hipSuccess;
        const char* a = items[0].src; const char* b = items[0].src + items[0].bytes;
        for (const Item& it : items) { a = std::min(a, it.src); b = std::max(b, it.src + it.bytes); }
        base = a; st = pack_stage(c, 1, (size_t)(b - a));
        if (!st) { if (items.size() == 1) { direct = true; return memcpy_d2h(c, items[0].dst, items[0].src, items[0].bytes); } stage.resize((size_t)(b - a)); st = stage.data(); }
        return hipMemcpyAsync(st, a, (size_t)(b - a), hipMemcpyDeviceToHost, c->stream);
    }
    void scatter() { if (!direct) for (const Item& it : items) memcpy(it.dst, st + (it.src - base), it.bytes); }   // after the sync
};

// Caching device allocator: hipMalloc/hipFree cost 0.1-1 ms each and a step of the pipeline would issue ~100 of them;
// freed blocks are parked and handed out again when a request of a similar size (<= 2x) arrives.
// Blocks are keyed by the device they were allocated on (a process may hold contexts on several GPUs, svt_create(device_id)).
struct PoolBlock { void* p; size_t bytes; int dev; };
static std::vector<PoolBlock>& pool_free() { static std::vector<PoolBlock> v; return v; }
static std::vector<PoolBlock>& pool_live() { static std::vector<PoolBlock> v; return v; }
static std::mutex& pool_mutex() { static std::mutex m; return m; }     // contexts of several host threads share the pool
static void* pool_alloc(size_t bytes) {
    std::lock_guard<std::mutex> lock(pool_mutex());
    bytes = (bytes + 255) & ~(size_t)255;
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    auto& fr = pool_free();
    size_t best = (size_t)-1;
    for (size_t i = 0; i < fr.size(); i++) if (fr[i].dev == dev && fr[i].bytes >= bytes && fr[i].bytes <= 2 * bytes + 4096 && (best == (size_t)-1 || fr[i].bytes < fr[best].bytes)) best = i;
    if (best != (size_t)-1) { PoolBlock b = fr[best]; fr[best] = fr.back(); fr.pop_back(); pool_live().push_back(b); return b.p; }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        for (size_t i = 0; i < fr.size();) { if (fr[i].dev == dev) { hipFree(fr[i].p); fr[i] = fr.back(); fr.pop_back(); } else i++; }   // release this device's parked blocks and retry once
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    }
    pool_live().push_back(PoolBlock{p, bytes, dev});
    return p;
}
static void pool_release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(pool_mutex());
    auto& lv = pool_live();
    for (size_t i = 0; i < lv.size(); i++) if (lv[i].p == p) { pool_free().push_back(lv[i]); lv[i] = lv.back(); lv.pop_back(); return; }
    hipFree(p);                                               // not ours (should not happen)
}
static void pool_trim() {                                       // the caller has made its device current (svt_destroy)
    std::lock_guard<std::mutex> lock(pool_mutex());
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return;
    auto& fr = pool_free();
    for (size_t i = 0; i < fr.size();) { if (fr[i].dev == dev) { hipFree(fr[i].p); fr[i] = fr.back(); fr.pop_back(); } else i++; }
}

template <class T> static int dmalloc(svt_ctx* c, T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    *p = (T*)pool_alloc(count * sizeof(T));
    if (!*p) return svt_fail(c, SVT_ERR_HIP, "device allocation failed (hipMalloc)");
    return SVT_OK;
}
#define TRY(x) do { int rc_ = (x); if (rc_ != SVT_OK) return rc_; } while (0)
static void dfree(void* p) { pool_release(p); }

static void free_seeds(SeedsDev& s) {
    dfree(s.meta_block);       // mini_base, qb_off, est_id, set_cnt, n_solid, mini_cnt, snp_cnt, est_valid, lsh_valid, status, snp_cursor live in it
    dfree(s.mini_pos); dfree(s.mini_kmer); dfree(s.mini_flags); dfree(s.set_kmer);
    dfree(s.snp_base); dfree(s.snp_pos); dfree(s.snp_kmer); dfree(s.snp_flags);
    dfree(s.lsh); dfree(s.qualbins);
    dfree(s.p_all); dfree(s.p_filt); dfree(s.allele); dfree(s.nz_cnt); dfree(s.nz_idx); dfree(s.nz_pa); dfree(s.nz_pf); dfree(s.nz_a);
    s = SeedsDev();
}

// CSR gather: fixed-capacity / cursor-ordered regions -> read-ordered compact arrays
__global__ void k_csr_gather(const u64* __restrict__ src_base, const u32* __restrict__ cnt, const u64* __restrict__ dst_off, u32 n,
                             const u32* __restrict__ spos, const u64* __restrict__ skm, const u8* __restrict__ sfl,
                             u32* __restrict__ dpos, u64* __restrict__ dkm, u8* __restrict__ dfl) {
    u32 r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    u64 sb = src_base[r], db = dst_off[r]; u32 m = cnt[r];
    for (u32 i = d_lane(); i < m; i += 64) {
        if (dpos) dpos[db + i] = spos[sb + i];
        if (dkm) dkm[db + i] = skm[sb + i];
        if (dfl) dfl[db + i] = sfl[sb + i];
    }
}
int launch_csr_gather(svt_ctx* c, const svt_batch* b, int which, const u64* d_dst_off, u32* d_pos, u64* d_kmer, u8* d_flags) {
    const SeedsDev& s = b->seeds;
    if (b->n == 0) return SVT_OK;
    if (which == 0) hipLaunchKernelGGL(k_csr_gather, dim3((b->n + 3) / 4), dim3(256), 0, c->stream, s.mini_base, s.mini_cnt, d_dst_off, b->n, s.mini_pos, s.mini_kmer, s.mini_flags, d_pos, d_kmer, d_flags);
    else hipLaunchKernelGGL(k_csr_gather, dim3((b->n + 3) / 4), dim3(256), 0, c->stream, s.snp_base, s.snp_cnt, d_dst_off, b->n, s.snp_pos, s.snp_kmer, s.snp_flags, d_pos, d_kmer, d_flags);
    HIPCHK(c, hipGetLastError());
    return SVT_OK;
}

extern "C" {

int svt_version(void) { return 100; }
// Samples in flight and the side streams of K8a are separate HIP streams; the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and
// a long kernel -- the 100-200 ms K12 launch -- blocks every stream that shares its queue (round 4: six pipelines with the device POA ran at 107 ms per
// step on 4 queues and 66 ms on 16).  The library asks for 16 unless the caller has set the variable; it only takes effect before the runtime initialises.
// setenv is not safe against getenv in other threads: it runs ONCE per process (std::call_once), only when the variable is unset, and the header tells hosts
// with threads of their own to export GPU_MAX_HW_QUEUES themselves before they start them (then nothing is written here).
static void want_hw_queues() { static std::once_flag once; std::call_once(once, [] { setenv("GPU_MAX_HW_QUEUES", "16", 0 /* never over a value the host set */); }); }
int svt_device_count(void) {
    want_hw_queues(); int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

int svt_create(int device_id, svt_ctx** out) {
    if (!out) return SVT_ERR_ARG;
    *out = nullptr;
    want_hw_queues();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return SVT_ERR_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return SVT_ERR_NODEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SVT_ERR_NODEVICE;     // gfx950 code objects only; no fallback
    if (hipSetDevice(device_id) != hipSuccess) return SVT_ERR_NODEVICE;
    svt_ctx* c = new svt_ctx();
    c->device = device_id;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SVT_ERR_HIP; }
    double pt[256];
    for (int x = 0; x < 256; x++) pt[x] = pow(10.0, -(double)x / 10.0);           // seeding.rs:811-812 (host libm, like the reference)
    if (hipMalloc((void**)&c->d_ptable, sizeof(pt)) != hipSuccess) { delete c; return SVT_ERR_HIP; }
    hipMemcpy(c->d_ptable, pt, sizeof(pt), hipMemcpyHostToDevice);
    *out = c;
    return SVT_OK;
}
static void shard_comm_drop(svt_ctx* c);
void svt_destroy(svt_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    ctx_sync(c);
    shard_comm_drop(c);
    for (int s = 0; s < svt_ctx::N_SIDE; s++) { if (c->side[s]) { hipStreamSynchronize(c->side[s]); hipStreamDestroy(c->side[s]); c->side[s] = nullptr; } if (c->side_done[s]) { hipEventDestroy(c->side_done[s]); c->side_done[s] = nullptr; } }
    if (c->side_go) { hipEventDestroy(c->side_go); c->side_go = nullptr; }
    if (c->parent) {                                              // a fork owns its stream and scratch only
        svt_ctx* p = c->parent;
        prof_drain(p);
        for (size_t i = 0; i < p->forks.size(); i++) if (p->forks[i] == c) { p->forks[i] = p->forks.back(); p->forks.pop_back(); break; }
        if (c->scratch) hipFree(c->scratch);
        if (c->pin) hipHostFree(c->pin);
        if (c->zc) hipHostFree(c->zc);
        for (int d = 0; d < 2; d++) if (c->pk[d]) hipHostFree(c->pk[d]);
        if (c->ev_block) hipEventDestroy(c->ev_block);
    if (c->sh_mark) hipEventDestroy(c->sh_mark);
        for (hipEvent_t e : c->prof_events) hipEventDestroy(e);
        if (c->sync_word) hipHostFree(c->sync_word);
        hipStreamDestroy(c->stream);
        delete c;
        return;
    }
    while (!c->forks.empty()) svt_destroy(c->forks.back());
    prof_drain(c);
    dfree(c->ht); dfree(c->snp_keys); dfree(c->snp_vals); dfree(c->d_hf); dfree(c->d_ptable); dfree(c->snp_occ); dfree(c->d_rank);
    dfree(c->tab_kmer); dfree(c->tab_rev); dfree(c->tab_fwd); dfree(c->tab_tmp);
    if (c->scratch) hipFree(c->scratch);
    if (c->pin) hipHostFree(c->pin);
    if (c->zc) hipHostFree(c->zc);
    for (int d = 0; d < 2; d++) if (c->pk[d]) hipHostFree(c->pk[d]);
    if (c->ev_block) hipEventDestroy(c->ev_block);
    if (c->sh_mark) hipEventDestroy(c->sh_mark);
    for (hipEvent_t e : c->prof_events) hipEventDestroy(e);
    if (c->sync_word) hipHostFree(c->sync_word);
    hipStreamDestroy(c->stream);
    pool_trim();
    delete c;
}
int svt_fork(svt_ctx* parent, svt_ctx** out) {
    if (!parent || !out) return SVT_ERR_ARG;
    *out = nullptr;
    if (parent->parent) return svt_fail(parent, SVT_ERR_ARG, "svt_fork: fork the root context, not a fork");
    hipSetDevice(parent->device);
    svt_ctx* c = new svt_ctx();
    c->device = parent->device; c->parent = parent;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return svt_fail(parent, SVT_ERR_HIP, "svt_fork: stream creation failed"); }
    parent->forks.push_back(c);
    svt_fork_refresh(c);
    *out = c;
    return SVT_OK;
}
// re-read the parent's shared tables (after svt_set_snpmers on the parent); the fork must be idle
int svt_fork_refresh(svt_ctx* c) {
    if (!c || !c->parent) return SVT_ERR_ARG;
    const svt_ctx* p = c->parent;
    c->k = p->k; c->snp_keys = p->snp_keys; c->snp_vals = p->snp_vals; c->snp_mask = p->snp_mask; c->d_hf = p->d_hf; c->n_hf = p->n_hf;
    c->n_sites = p->n_sites; c->words = p->words; c->site_order = p->site_order; c->d_ptable = p->d_ptable;
    c->snp_occ = p->snp_occ; c->snp_occ_mask = p->snp_occ_mask; c->d_rank = p->d_rank; c->rank_s = p->rank_s;
    return SVT_OK;
}
const char* svt_last_error(const svt_ctx* c) { return c ? c->err.c_str() : "null context"; }

// Kernel and copy-path selections are context state set through the ABI, not process environment.
static int* option_slot(SvtOptions& o, const char* key) {
    const std::string k = key ? key : "";
    if (k == "k8_kernel") return &o.k8_kernel;
    if (k == "k9_kernel") return &o.k9_kernel;
    if (k == "shard_seeds") return &o.shard_seeds;
    if (k == "count_kernel") return &o.count_kernel;
    if (k == "count_table_hint") return &o.count_table_hint;
    if (k == "consensus_dense") return &o.consensus_dense;
    if (k == "consensus_chunk") return &o.consensus_chunk;
    if (k == "pin_staging") return &o.pin_staging;
    if (k == "zero_copy") return &o.zero_copy;
    if (k == "sync_block") return &o.sync_block;
    if (k == "keep_ascii") return &o.keep_ascii;
    if (k == "seeds_hash") return &o.seeds_hash;
    if (k == "k9_window") return &o.k9_window;
    if (k == "k8a_queue") return &o.k8a_queue;
    if (k == "k8a_g16") return &o.k8a_g16;
    if (k == "k8a_pk16") return &o.k8a_pk16;
    if (k == "poa_rows") return &o.poa_rows;
    if (k == "shard_world1") return &o.shard_world1;
    if (k == "shard_timeout_s") return &o.shard_timeout_s;
    return nullptr;
}
int svt_set_option(svt_ctx* c, const char* key, int64_t value) {
    if (!c) return SVT_ERR_ARG;
    svt_ctx* root = c->parent ? c->parent : c;
    int* slot = option_slot(root->options, key);
    if (!slot) return svt_fail(c, SVT_ERR_ARG, std::string("svt_set_option: unknown option '") + (key ? key : "") + "'");
    const std::string k = key;
    const int64_t hi = k == "k9_kernel" ? 3 : k == "consensus_chunk" ? 65536 : k == "k9_window" ? 64 : k == "poa_rows" ? 2 : k == "shard_timeout_s" ? 86400 : k == "k8a_pk16" ? 3 : k == "count_kernel" ? 3 : 1;
    if (k == "k9_window" && value != 32 && value != 64) return svt_fail(c, SVT_ERR_ARG, "svt_set_option: k9_window is 32 or 64");
    if (value < 0 || value > hi) return svt_fail(c, SVT_ERR_ARG, "svt_set_option: value out of range for '" + k + "'");
    *slot = (int)value;
    return SVT_OK;
}
int svt_get_option(svt_ctx* c, const char* key, int64_t* value) {
    if (!c || !value) return SVT_ERR_ARG;
    svt_ctx* root = c->parent ? c->parent : c;
    // read-only counters of this context: K9 pairs walked with the windowed slab / walked again around their end diagonal / with the full slab
    if (key && !strcmp(key, "shard_exchanges")) { *value = (int64_t)c->sh_calls; return SVT_OK; }
    if (key && !strcmp(key, "shard_bytes")) { *value = (int64_t)c->sh_bytes; return SVT_OK; }
    if (key && !strcmp(key, "k9_pairs")) { *value = (int64_t)c->k9_pairs; return SVT_OK; }
    if (key && !strcmp(key, "k9_again_pairs")) { *value = (int64_t)c->k9_again_pairs; return SVT_OK; }
    if (key && !strcmp(key, "k9_redo_pairs")) { *value = (int64_t)c->k9_redo_pairs; return SVT_OK; }
    if (key && !strcmp(key, "k8a_packed_pairs")) { *value = (int64_t)c->k8a_packed; return SVT_OK; }
    if (key && !strcmp(key, "k8a_redo_pairs")) { *value = (int64_t)c->k8a_redo; return SVT_OK; }
    if (key && !strcmp(key, "poa_clusters")) { *value = (int64_t)c->poa_clusters; return SVT_OK; }
    if (key && !strcmp(key, "poa_handed_back")) { *value = (int64_t)c->poa_handed_back; return SVT_OK; }
    if (key && !strcmp(key, "poa_cons_device")) { *value = (int64_t)c->poa_cons_device; return SVT_OK; }
    int* slot = option_slot(root->options, key);
    if (!slot) return svt_fail(c, SVT_ERR_ARG, std::string("svt_get_option: unknown option '") + (key ? key : "") + "'");
    *value = *slot;
    return SVT_OK;
}

int svt_profile_enable(svt_ctx* c, int on) { c->prof = on != 0; c->prof_level = on == 2 ? 2 : 1; return SVT_OK; }
void svt_profile_reset(svt_ctx* c) { prof_drain(c); c->prof_entries.clear(); }
int svt_profile_count(svt_ctx* c) { prof_drain(c); return (int)c->prof_entries.size(); }
int svt_profile_get(svt_ctx* c, int idx, char* name_out, uint64_t* launches, double* ms, double* algo_bytes, double* units) {
    prof_drain(c);
    if (idx < 0 || idx >= (int)c->prof_entries.size()) return SVT_ERR_ARG;
    const ProfEntry& e = c->prof_entries[idx];
    if (name_out) { strncpy(name_out, e.name.c_str(), 63); name_out[63] = 0; }
    if (launches) *launches = e.launches; if (ms) *ms = e.ms; if (algo_bytes) *algo_bytes = e.bytes; if (units) *units = e.units;
    return SVT_OK;
}

// ---- batches ----------------------------------------------------------------------------------
int svt_batch_upload(svt_ctx* c, const uint8_t* seq, const uint8_t* qual, const uint64_t* offsets, uint32_t n, svt_batch** out) {
    if (!c || !out || (n && (!seq || !offsets))) return svt_fail(c, SVT_ERR_ARG, "svt_batch_upload: null argument");
    hipSetDevice(c->device);
    svt_batch* b = new svt_batch();
    b->n = n; b->has_qual = qual != nullptr;
    b->h_off.assign(offsets, offsets + n + 1);
    b->h_woff.resize(n + 1);
    u64 wo = 0;
    for (u32 i = 0; i < n; i++) {
        u64 len = offsets[i + 1] - offsets[i];
        if (len > 0xFFFFFF) { delete b; return svt_fail(c, SVT_ERR_ARG, "sequence longer than 16 Mbases"); }
        b->max_len = std::max<u32>(b->max_len, (u32)len);
        b->h_woff[i] = wo; wo += (len + 15) / 16 + SVT_PAD_WORDS;
    }
    b->h_woff[n] = wo;
    b->total_bases = offsets[n] - offsets[0]; b->total_words = wo;
    const u64 base0 = offsets[0];
    std::vector<u64> rel(n + 1);
    for (u32 i = 0; i <= n; i++) rel[i] = offsets[i] - base0;
    b->h_off = rel;
    u8* d_ascii = nullptr;
    // every failure below releases what was allocated so far (the batch owns its device arrays; d_ascii is a temporary)
    auto body = [&]() -> int {
        TRY(dmalloc(c, &b->d_packed, wo)); TRY(dmalloc(c, &b->d_nmask, wo)); TRY(dmalloc(c, &b->d_flags, n));
        if (b->total_bases <= ((u64)4 << 20)) {
            // a small batch (consensuses, ASVs: uploaded ~10 times per step): offsets, ASCII and qualities in ONE block, sent with one copy
            const size_t so = (size_t)(n + 1) * 8, sb = (size_t)((b->total_bases + 7) & ~(u64)7);
            const size_t total = 2 * so + sb + (qual ? sb + 64 : 0);         // + the slack behind the qualities (see below)
            TRY(dmalloc(c, &b->d_block, total));
            b->d_off = (u64*)b->d_block; b->d_woff = (u64*)(b->d_block + so); d_ascii = b->d_block + 2 * so; if (qual) b->d_qual = d_ascii + sb;
            StageUp stage(c, total);
            memcpy(stage.p, rel.data(), so); memcpy(stage.p + so, b->h_woff.data(), so);
            if (b->total_bases) { memcpy(stage.p + 2 * so, seq + base0, b->total_bases); if (qual) memcpy(stage.p + 2 * so + sb, qual + base0, b->total_bases); }
            HIPCHK(c, stage.send(b->d_block, total));
            const int rcp = launch_pack(c, b, d_ascii);
            ctx_sync(c);                                               // `stage` goes out of scope
            return rcp;
        }
        TRY(dmalloc(c, &b->d_off, n + 1)); TRY(dmalloc(c, &b->d_woff, n + 1));
        TRY(dmalloc(c, &d_ascii, b->total_bases));
        if (qual) TRY(dmalloc(c, &b->d_qual, b->total_bases + 64));        // 64 bytes of slack: the counting kernel reads a window's 64 quality bytes per read as four 16-byte loads
        HIPCHK(c, hipMemcpyAsync(b->d_off, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(b->d_woff, b->h_woff.data(), (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
        if (b->total_bases) HIPCHK(c, hipMemcpyAsync(d_ascii, seq + base0, b->total_bases, hipMemcpyHostToDevice, c->stream));
        if (qual && b->total_bases) HIPCHK(c, hipMemcpyAsync(b->d_qual, qual + base0, b->total_bases, hipMemcpyHostToDevice, c->stream));
        return launch_pack(c, b, d_ascii);
    };
    const int rc = body();
    ctx_sync(c);
    if (b->d_block) { if (rc == SVT_OK && c->opt().keep_ascii) b->d_ascii = d_ascii; }          // the ASCII bases live (and die) with the block
    else if (rc == SVT_OK && c->opt().keep_ascii) b->d_ascii = d_ascii; else dfree(d_ascii);
    if (rc != SVT_OK) { svt_batch_free(c, b); return rc; }
    *out = b;
    return SVT_OK;
}
// A view of reads [lo, hi) of `parent` (multi-GPU: the read block a rank counts): no copy -- the offsets stay absolute
// into the parent's arrays.  Only the kernels that read packed bases / qualities / flags may take a slice (svt_count_partial,
// svt_split_kmers_emit); seeds live on whole batches.  Free it with svt_batch_free before the parent.
int svt_batch_slice(svt_ctx* c, const svt_batch* parent, uint32_t lo, uint32_t hi, svt_batch** out) {
    if (!c || !parent || !out) return svt_fail(c, SVT_ERR_ARG, "svt_batch_slice: null argument");
    *out = nullptr;
    if (parent->slice_of) return svt_fail(c, SVT_ERR_ARG, "svt_batch_slice: slice the whole batch, not a slice");
    if (lo > hi || hi > parent->n) return svt_fail(c, SVT_ERR_ARG, "svt_batch_slice: range outside the batch");
    svt_batch* b = new svt_batch();
    b->slice_of = parent; b->n = hi - lo; b->has_qual = parent->has_qual;
    b->h_off.assign(parent->h_off.begin() + lo, parent->h_off.begin() + hi + 1);
    b->h_woff.assign(parent->h_woff.begin() + lo, parent->h_woff.begin() + hi + 1);
    b->total_bases = b->h_off.back() - b->h_off.front(); b->total_words = b->h_woff.back() - b->h_woff.front();
    for (u32 i = 0; i < b->n; i++) b->max_len = std::max<u32>(b->max_len, (u32)(b->h_off[i + 1] - b->h_off[i]));
    b->d_off = parent->d_off + lo; b->d_woff = parent->d_woff + lo; b->d_flags = parent->d_flags + lo;
    b->d_packed = parent->d_packed; b->d_nmask = parent->d_nmask; b->d_qual = parent->d_qual;
    *out = b;
    return SVT_OK;
}
void svt_batch_free(svt_ctx* c, svt_batch* b) {
    if (!b) return;
    if (c) { hipSetDevice(c->device); ctx_sync(c); }
    if (b->slice_of) { delete b; return; }
    if (b->d_block) dfree(b->d_block); else { dfree(b->d_off); dfree(b->d_woff); dfree(b->d_qual); dfree(b->d_ascii); }
    dfree(b->d_packed); dfree(b->d_nmask); dfree(b->d_flags);
    dfree(b->d_tag_qual); dfree(b->d_tag_hp);
    free_seeds(b->seeds);
    delete b;
}
// --use-hpc: per-base tags of a batch of homopolymer-compressed reads (src/alignment.rs:480): the minimum quality and the length of the
// run every base stands for.  K9 then reads a target's qualities from the tags and writes the run length into bits 56-63 of Base cells.
int svt_batch_set_tags(svt_ctx* c, svt_batch* b, const uint8_t* qual, const uint8_t* hp_len) {
    if (!c || !b || !qual || !hp_len) return svt_fail(c, SVT_ERR_ARG, "svt_batch_set_tags: null argument");
    if (b->slice_of) return svt_fail(c, SVT_ERR_ARG, "svt_batch_set_tags: tags live on whole batches");
    hipSetDevice(c->device);
    const size_t nb = std::max<size_t>(b->total_bases, 1);
    if (!b->d_tag_qual) TRY(dmalloc(c, &b->d_tag_qual, nb));
    if (!b->d_tag_hp) TRY(dmalloc(c, &b->d_tag_hp, nb));
    HIPCHK(c, hipMemcpyAsync(b->d_tag_qual, qual, b->total_bases, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b->d_tag_hp, hp_len, b->total_bases, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, ctx_sync(c));
    return SVT_OK;
}
// K0 again from the ASCII bases kept in HBM ("keep_ascii"): the 2-bit words, the non-ACGT mask and the per-read flags are rewritten
int svt_batch_repack(svt_ctx* c, svt_batch* b) {
    if (!c || !b) return svt_fail(c, SVT_ERR_ARG, "svt_batch_repack: null argument");
    if (!b->d_ascii) return svt_fail(c, SVT_ERR_STATE, "svt_batch_repack: the batch was uploaded without the keep_ascii option");
    hipSetDevice(c->device);
    TRY(launch_pack(c, b, b->d_ascii));
    HIPCHK(c, ctx_sync(c));
    return SVT_OK;
}

// Streaming-copy rate of this device's HBM, measured the way the guide's 6.29 TB/s figure was (float4 grid-stride copy):
// the denominator bench.py states next to the 8 TB/s spec.  bytes = size of each of the two buffers.
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
__global__ void k_copy16(const v4u_t* __restrict__ src, v4u_t* __restrict__ dst, u64 n) {   // non-temporal 16-byte loads / stores, 64 blocks per CU: the best of tools/micro/hbm_copy.hip
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
int svt_hbm_copy_peak(svt_ctx* c, uint64_t bytes, int iters, double* gb_per_s) {
    if (!c || !gb_per_s || bytes < 4096 || iters < 1) return svt_fail(c, SVT_ERR_ARG, "svt_hbm_copy_peak: bad argument");
    hipSetDevice(c->device);
    *gb_per_s = 0.0;
    bytes &= ~(uint64_t)4095;
    v4u_t *a = nullptr, *b = nullptr;
    int rc = [&]() -> int {
        TRY(dmalloc(c, (u8**)&a, bytes)); TRY(dmalloc(c, (u8**)&b, bytes));
        HIPCHK(c, hipMemsetAsync(a, 1, bytes, c->stream));
        hipEvent_t e0, e1;
        HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
        const u64 n = bytes / 16;
        const u32 blocks = 256 * 64;                               // 64 blocks of 256 threads per CU
        hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, c->stream, (const v4u_t*)a, b, n);      // warm-up
        double best = 0.0;
        for (int it = 0; it < iters; it++) {
            hipEventRecord(e0, c->stream);
            hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, c->stream, (const v4u_t*)a, b, n);
            hipEventRecord(e1, c->stream);
            if (hipEventSynchronize(e1) != hipSuccess) break;
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (ms > 0) best = std::max(best, 2.0 * (double)bytes / 1e9 / ((double)ms / 1e3));
        }
        hipEventDestroy(e0); hipEventDestroy(e1);
        HIPCHK(c, hipGetLastError());
        *gb_per_s = best;
        return SVT_OK;
    }();
    ctx_sync(c);
    dfree(a); dfree(b);
    return rc;
}
uint32_t svt_batch_size(const svt_batch* b) { return b ? b->n : 0; }
int svt_batch_fetch_packed(svt_ctx* c, const svt_batch* b, uint32_t read, uint32_t* words, uint16_t* nonacgt) {
    if (!b || read >= b->n) return svt_fail(c, SVT_ERR_ARG, "svt_batch_fetch_packed: bad read index");
    u64 len = b->h_off[read + 1] - b->h_off[read]; u64 nw = (len + 15) / 16;
    if (words) HIPCHK(c, hipMemcpy(words, b->d_packed + b->h_woff[read], nw * 4, hipMemcpyDeviceToHost));
    if (nonacgt) HIPCHK(c, hipMemcpy(nonacgt, b->d_nmask + b->h_woff[read], nw * 2, hipMemcpyDeviceToHost));
    return SVT_OK;
}

// ---- stage 1 ------------------------------------------------------------------------------------
static int check_k(svt_ctx* c, u32 k) {
    if (k < 3 || k > 31 || (k & 1) == 0) return svt_fail(c, SVT_ERR_ARG, "k must be odd and <= 31 (seeding.rs:988)");
    return SVT_OK;
}

int svt_split_kmers_emit(svt_ctx* c, const svt_batch* b, uint32_t k, uint8_t min_bq, const uint8_t* rc_flags,
                         const uint64_t* out_offsets, uint64_t* out, uint32_t* out_counts) {
    if (!c || !b || !out_offsets || !out || !out_counts) return svt_fail(c, SVT_ERR_ARG, "svt_split_kmers_emit: null argument");
    TRY(check_k(c, k));
    hipSetDevice(c->device);
    u32 n = b->n;
    u64 total = 0;
    for (u32 i = 0; i < n; i++) { u64 len = b->h_off[i + 1] - b->h_off[i]; u64 need = len >= k ? len - k + 1 : 0; total = std::max(total, out_offsets[i] + need); }
    Carve cv; size_t i_off = cv.add((n + 1) * 8), i_out = cv.add(total * 8), i_cnt = cv.add(n * 4), i_rc = cv.add(n);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u64* d_off = carve_ptr<u64>(c, cv, i_off); u64* d_out = carve_ptr<u64>(c, cv, i_out); u32* d_cnt = carve_ptr<u32>(c, cv, i_cnt); u8* d_rc = carve_ptr<u8>(c, cv, i_rc);
    HIPCHK(c, hipMemcpyAsync(d_off, out_offsets, n * 8, hipMemcpyHostToDevice, c->stream));
    if (rc_flags) HIPCHK(c, hipMemcpyAsync(d_rc, rc_flags, n, hipMemcpyHostToDevice, c->stream));
    TRY(launch_split_emit(c, b, k, min_bq, rc_flags ? d_rc : nullptr, d_off, d_out, d_cnt));
    HIPCHK(c, memcpy_d2h(c, out, d_out, total * 8));
    HIPCHK(c, memcpy_d2h(c, out_counts, d_cnt, n * 4));
    HIPCHK(c, ctx_sync(c));
    return SVT_OK;
}

static int ht_prepare_cap(svt_ctx* c, u64 cap) {
    if (c->ht_cap != cap) {
        dfree(c->ht); c->ht = nullptr; c->ht_cap = 0;
        TRY(dmalloc(c, &c->ht, cap));
        c->ht_cap = cap;
    }
    return launch_ht_init(c);
}
static int ht_prepare(svt_ctx* c, u64 want_entries) {          // capacity for `want_entries` DISTINCT keys at load <= 2/3
    u64 cap = 1024;
    while (cap < want_entries + want_entries / 2) cap <<= 1;
    return ht_prepare_cap(c, cap);
}

static int count_insert(svt_ctx* c, const svt_batch* b, u32 k, u8 min_bq, const u8* rc_flags) {
    TRY(check_k(c, k));
    hipSetDevice(c->device);
    u64 positions = 0;
    for (u32 i = 0; i < b->n; i++) { u64 len = b->h_off[i + 1] - b->h_off[i]; if (len >= k) positions += len - k + 1; }
    c->ht_positions = positions;
    // Sequencing reads repeat most k-mers; distinct k-mers are typically 10-25 % of the positions.  Start with a table
    // sized for positions/2.5 distinct keys and let the kernel report overflow (probe length > 4096): then retry with 4x.
    // The worst case (every k-mer distinct) ends at the always-safe size 1.5 * positions.
    u64 safe = 1024; while (safe < positions + positions / 2) safe <<= 1;
    u64 cap = 1024; while (cap < (positions * 2) / 5) cap <<= 1;
    if (cap > safe) cap = safe;
    // ... unless this context has just counted a batch of the same order of size (a pipeline's next sample): amplicon reads repeat their k-mers far more than
    // that rule assumes -- 2.5 M distinct keys in 150 M positions at 100k reads, a 1.07 GB table 4 % full, cleared and scanned every step and too large for the
    // 256 MB of Infinity Cache its atomics would otherwise hit.  The table then holds twice the keys of the batch before (scaled by the positions) at load <= 2/3;
    // a sample that needs more overflows and is counted again in a table four times the size, as before.
    if (c->opt().count_table_hint && c->ht_hint_distinct && c->ht_hint_positions && positions >= c->ht_hint_positions / 2 && positions <= c->ht_hint_positions * 2) {
        const u64 want = (u64)(2.0 * (double)c->ht_hint_distinct * (double)positions / (double)c->ht_hint_positions) + 4096;
        u64 hc = 1u << 16; while (hc < want + want / 2) hc <<= 1;
        if (hc < cap) cap = hc;
    }
    Carve cv; size_t irc = cv.add(b->n), iov = cv.add(4);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u8* d_rc = carve_ptr<u8>(c, cv, irc); u32* d_ov = carve_ptr<u32>(c, cv, iov);
    if (rc_flags) HIPCHK(c, hipMemcpyAsync(d_rc, rc_flags, b->n, hipMemcpyHostToDevice, c->stream));
    while (true) {
        TRY(ht_prepare_cap(c, cap));
        HIPCHK(c, hipMemsetAsync(d_ov, 0, 4, c->stream));
        TRY(launch_count_insert(c, b, k, min_bq, rc_flags ? d_rc : nullptr, d_ov));
        u32 ov = 0;
        HIPCHK(c, peek(c, d_ov, &ov, 4));                                        // (a copy into pageable memory made the runtime spin for the whole counting kernel: 4 % of a 2-CPU step)
        if (!ov) break;
        if (cap >= safe) return svt_fail(c, SVT_ERR_OVERFLOW, "k-mer table overflow at the safe capacity (should be impossible)");
        cap = std::min(cap * 4, safe);
    }
    c->ht_fresh = true;
    return SVT_OK;
}

// The kept entries -> canonical order (kernels_table.hip), resident in HBM; the two short selections Stage 1b needs are fetched at once.
static int table_finish(svt_ctx* c, u32 k, u64 kept, const u64* dk, const u32* dr, const u32* df) {
    c->tab_n = kept; c->grp_kmer.clear(); c->grp_rev.clear(); c->grp_fwd.clear(); c->heavy_kmer.clear(); c->heavy_rev.clear(); c->heavy_fwd.clear();
    c->cnt_kmer.clear(); c->cnt_rev.clear(); c->cnt_fwd.clear();
    if (kept == 0) { c->tab_valid = true; c->tab_on_host = true; return SVT_OK; }
    if (kept > c->tab_cap) {
        dfree(c->tab_kmer); dfree(c->tab_rev); dfree(c->tab_fwd); c->tab_kmer = nullptr; c->tab_rev = c->tab_fwd = nullptr; c->tab_cap = 0;
        const u64 cap = kept + kept / 4;
        TRY(dmalloc(c, &c->tab_kmer, cap)); TRY(dmalloc(c, &c->tab_rev, cap)); TRY(dmalloc(c, &c->tab_fwd, cap));
        c->tab_cap = cap;
    }
    size_t need_sort = 0, need_sel = 0;
    TRY(launch_table_sort(c, k, kept, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &need_sort));
    TRY(launch_table_select(c, k, kept, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &need_sel));
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_ka = 0, o_kb = o_ka + al(kept * 8), o_ia = o_kb + al(kept * 8), o_ib = o_ia + al(kept * 4), o_fg = o_ib + al(kept * 4), o_fh = o_fg + al(kept),
                 o_og = o_fh + al(kept), o_oh = o_og + al(kept * 4), o_cn = o_oh + al(kept * 4), o_tmp = o_cn + 256, total = o_tmp + al(std::max(need_sort, need_sel));
    if (total > c->tab_tmp_bytes) {
        dfree(c->tab_tmp); c->tab_tmp = nullptr; c->tab_tmp_bytes = 0;
        u8* p = nullptr; TRY(dmalloc(c, &p, total + total / 4)); c->tab_tmp = p; c->tab_tmp_bytes = total + total / 4;
    }
    char* base = (char*)c->tab_tmp;
    u32* d_cn = (u32*)(base + o_cn); u32* d_og = (u32*)(base + o_og); u32* d_oh = (u32*)(base + o_oh);
    TRY(launch_table_sort(c, k, kept, dk, dr, df, c->tab_kmer, c->tab_rev, c->tab_fwd, (u64*)(base + o_ka), (u64*)(base + o_kb), (u32*)(base + o_ia), (u32*)(base + o_ib),
                          base + o_tmp, c->tab_tmp_bytes - o_tmp, nullptr));
    TRY(launch_table_select(c, k, kept, c->tab_kmer, c->tab_rev, c->tab_fwd, (u8*)(base + o_fg), (u8*)(base + o_fh), d_og, d_oh, d_cn, base + o_tmp, c->tab_tmp_bytes - o_tmp, nullptr));
    u32 hc[2] = {0, 0};
    HIPCHK(c, peek(c, d_cn, hc, 8));
    c->tab_valid = true;
    // the selected entries, gathered into the (now free) key buffers: [grp | heavy] km, rev, fwd
    const u64 ng = hc[0], nh = hc[1], ns = ng + nh;
    if (ns) {
        // km | rev | fwd of the selected entries back to back in the key buffer when they fit (they are ~1 % of the table): one copy back
        u64* gk = (u64*)(base + o_ka); u32* gr = (u32*)(base + o_ia); u32* gf = (u32*)(base + o_ib);
        if (ns * 2 <= kept) { gr = (u32*)(gk + ns); gf = gr + ns; }
        TRY(launch_table_gather(c, d_og, ng, c->tab_kmer, c->tab_rev, c->tab_fwd, gk, gr, gf));
        TRY(launch_table_gather(c, d_oh, nh, c->tab_kmer, c->tab_rev, c->tab_fwd, gk + ng, gr + ng, gf + ng));
        c->grp_kmer.resize(ng); c->grp_rev.resize(ng); c->grp_fwd.resize(ng); c->heavy_kmer.resize(nh); c->heavy_rev.resize(nh); c->heavy_fwd.resize(nh);
        if (ns * 2 <= kept) {
            DownPack dn(c);
            dn.get(gk, c->grp_kmer.data(), ng * 8); dn.get(gr, c->grp_rev.data(), ng * 4); dn.get(gf, c->grp_fwd.data(), ng * 4);
            dn.get(gk + ng, c->heavy_kmer.data(), nh * 8); dn.get(gr + ng, c->heavy_rev.data(), nh * 4); dn.get(gf + ng, c->heavy_fwd.data(), nh * 4);
            HIPCHK(c, dn.recv());
            HIPCHK(c, ctx_sync(c));
            dn.scatter();
            return SVT_OK;
        }
        if (ng) {
            HIPCHK(c, memcpy_d2h(c, c->grp_kmer.data(), gk, ng * 8));
            HIPCHK(c, memcpy_d2h(c, c->grp_rev.data(), gr, ng * 4));
            HIPCHK(c, memcpy_d2h(c, c->grp_fwd.data(), gf, ng * 4));
        }
        if (nh) {
            HIPCHK(c, memcpy_d2h(c, c->heavy_kmer.data(), gk + ng, nh * 8));
            HIPCHK(c, memcpy_d2h(c, c->heavy_rev.data(), gr + ng, nh * 4));
            HIPCHK(c, memcpy_d2h(c, c->heavy_fwd.data(), gf + ng, nh * 4));
        }
        HIPCHK(c, ctx_sync(c));
    }
    return SVT_OK;
}
// host copy of the sorted table, on demand (tests, the multi-GPU merge path, callers of the B1 boundary that want the whole table)
static int table_to_host(svt_ctx* c) {
    if (!c->tab_valid || c->tab_on_host) return SVT_OK;
    hipSetDevice(c->device);
    c->cnt_kmer.resize(c->tab_n); c->cnt_rev.resize(c->tab_n); c->cnt_fwd.resize(c->tab_n);
    if (c->tab_n) {
        HIPCHK(c, memcpy_d2h(c, c->cnt_kmer.data(), c->tab_kmer, c->tab_n * 8));
        HIPCHK(c, memcpy_d2h(c, c->cnt_rev.data(), c->tab_rev, c->tab_n * 4));
        HIPCHK(c, memcpy_d2h(c, c->cnt_fwd.data(), c->tab_fwd, c->tab_n * 4));
        HIPCHK(c, ctx_sync(c));
    }
    c->tab_on_host = true;
    return SVT_OK;
}

// mode 0/1: filtered + sorted into ctx vectors; mode 2: everything, unsorted
static int count_collect(svt_ctx* c, u32 k, int mode, u64* n_distinct, u64* n_kept) {
    // ONE scan of the table.  Output capacity is a guaranteed bound: a kept k-mer has total count >= 3 (seq_parse.rs:36,41),
    // so kept <= positions/3; mode 2 (all entries) is bounded by the number of insertions / merged entries.
    u64 bound = (mode == 2) ? std::min<u64>(c->ht_cap, std::max<u64>(c->ht_positions, c->ht_distinct)) : c->ht_positions / 3 + 1;
    if (bound > c->ht_cap) bound = c->ht_cap;
    Carve cv; size_t ik = cv.add(bound * 8), ir = cv.add(bound * 4), iff = cv.add(bound * 4), icn = cv.add(16);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u64* dk = carve_ptr<u64>(c, cv, ik); u32* dr = carve_ptr<u32>(c, cv, ir); u32* df = carve_ptr<u32>(c, cv, iff); ull* d_cnt = carve_ptr<ull>(c, cv, icn);
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, 16, c->stream));
    TRY(launch_ht_compact(c, mode, dk, dr, df, d_cnt));
    ull h[2];
    HIPCHK(c, memcpy_d2h(c, h, d_cnt, 16));
    HIPCHK(c, ctx_sync(c));
    u64 kept = h[1];
    c->ht_distinct = h[0];
    if (c->ht_fresh) { c->ht_hint_distinct = h[0]; c->ht_hint_positions = c->ht_positions; c->ht_fresh = false; }   // the table of ONE counted batch (not a merged one): what the next batch's table is sized from
    if (kept > bound) return svt_fail(c, SVT_ERR_OVERFLOW, "count_collect: kept entries exceed the guaranteed bound");
    c->tab_valid = false; c->tab_on_host = false;
    if (mode == 2) {
        c->cnt_kmer.resize(kept); c->cnt_rev.resize(kept); c->cnt_fwd.resize(kept);
        if (kept) {
            HIPCHK(c, memcpy_d2h(c, c->cnt_kmer.data(), dk, kept * 8));
            HIPCHK(c, memcpy_d2h(c, c->cnt_rev.data(), dr, kept * 4));
            HIPCHK(c, memcpy_d2h(c, c->cnt_fwd.data(), df, kept * 4));
            HIPCHK(c, ctx_sync(c));
        }
    } else {
        TRY(table_finish(c, k, kept, dk, dr, df));
    }
    if (n_distinct) *n_distinct = c->ht_distinct;
    if (n_kept) *n_kept = kept;
    return SVT_OK;
}

int svt_count_split_kmers(svt_ctx* c, const svt_batch* b, uint32_t k, uint8_t min_bq, const uint8_t* rc_flags, int single_strand,
                          uint64_t* n_distinct, uint64_t* n_kept) {
    if (!c || !b) return svt_fail(c, SVT_ERR_ARG, "svt_count_split_kmers: null argument");
    TRY(count_insert(c, b, k, min_bq, rc_flags));
    return count_collect(c, k, single_strand ? 1 : 0, n_distinct, n_kept);
}
int svt_count_candidates_sizes(svt_ctx* c, uint64_t* n_table, uint64_t* n_group_entries, uint64_t* n_heavy) {
    if (!c || !c->tab_valid) return svt_fail(c, SVT_ERR_STATE, "svt_count_candidates_sizes: no sorted table (call svt_count_split_kmers / svt_count_finalize first)");
    if (n_table) *n_table = c->tab_n;
    if (n_group_entries) *n_group_entries = c->grp_kmer.size();
    if (n_heavy) *n_heavy = c->heavy_kmer.size();
    return SVT_OK;
}
int svt_count_candidates_fetch(svt_ctx* c, uint64_t* g_kmer, uint32_t* g_rev, uint32_t* g_fwd, uint64_t* h_kmer, uint32_t* h_rev, uint32_t* h_fwd) {
    if (!c || !c->tab_valid) return svt_fail(c, SVT_ERR_STATE, "svt_count_candidates_fetch: no sorted table");
    if (g_kmer) memcpy(g_kmer, c->grp_kmer.data(), c->grp_kmer.size() * 8);
    if (g_rev) memcpy(g_rev, c->grp_rev.data(), c->grp_rev.size() * 4);
    if (g_fwd) memcpy(g_fwd, c->grp_fwd.data(), c->grp_fwd.size() * 4);
    if (h_kmer) memcpy(h_kmer, c->heavy_kmer.data(), c->heavy_kmer.size() * 8);
    if (h_rev) memcpy(h_rev, c->heavy_rev.data(), c->heavy_rev.size() * 4);
    if (h_fwd) memcpy(h_fwd, c->heavy_fwd.data(), c->heavy_fwd.size() * 4);
    return SVT_OK;
}
int svt_count_fetch(svt_ctx* c, uint64_t* kmer, uint32_t* rev, uint32_t* fwd) {
    if (!c) return SVT_ERR_ARG;
    TRY(table_to_host(c));
    size_t n = c->cnt_kmer.size();
    if (kmer) memcpy(kmer, c->cnt_kmer.data(), n * 8);
    if (rev) memcpy(rev, c->cnt_rev.data(), n * 4);
    if (fwd) memcpy(fwd, c->cnt_fwd.data(), n * 4);
    return SVT_OK;
}
int svt_count_partial(svt_ctx* c, const svt_batch* b, uint32_t k, uint8_t min_bq, const uint8_t* rc_flags, uint64_t* n_distinct) {
    if (!c || !b) return svt_fail(c, SVT_ERR_ARG, "svt_count_partial: null argument");
    TRY(count_insert(c, b, k, min_bq, rc_flags));
    return count_collect(c, k, 2, n_distinct, nullptr);
}
int svt_count_export(svt_ctx* c, uint64_t* kmer, uint32_t* rev, uint32_t* fwd) { return svt_count_fetch(c, kmer, rev, fwd); }
int svt_count_merge(svt_ctx* c, const uint64_t* kmer, const uint32_t* rev, const uint32_t* fwd, uint64_t n) {
    if (!c || (n && (!kmer || !rev || !fwd))) return svt_fail(c, SVT_ERR_ARG, "svt_count_merge: null argument");
    hipSetDevice(c->device);
    if (!c->ht) return svt_fail(c, SVT_ERR_STATE, "svt_count_merge: no table (call svt_count_partial first)");
    // grow when the merged table could exceed half the capacity
    u64 need = c->ht_distinct + n;
    if (need + need / 2 > c->ht_cap) {
        std::vector<u64> k0 = c->cnt_kmer; std::vector<u32> r0 = c->cnt_rev, f0 = c->cnt_fwd;   // entries of the current table (mode 2 snapshot)
        TRY(ht_prepare(c, need));
        c->ht_distinct = 0;
        if (!k0.empty()) TRY(svt_count_merge(c, k0.data(), r0.data(), f0.data(), k0.size()));
    }
    Carve cv; size_t ik = cv.add(n * 8), ir = cv.add(n * 4), iff = cv.add(n * 4);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u64* dk = carve_ptr<u64>(c, cv, ik); u32* dr = carve_ptr<u32>(c, cv, ir); u32* df = carve_ptr<u32>(c, cv, iff);
    HIPCHK(c, hipMemcpyAsync(dk, kmer, n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dr, rev, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(df, fwd, n * 4, hipMemcpyHostToDevice, c->stream));
    TRY(launch_ht_merge(c, dk, dr, df, n));
    HIPCHK(c, ctx_sync(c));
    c->ht_fresh = false;
    c->ht_distinct += n;   // upper bound until the next collect
    c->ht_positions += n * 3;   // keeps count_collect's kept <= positions/3 bound valid for merged tables (n entries)
    return SVT_OK;
}
// ---- C1 on device-resident tables (RCCL all-gather of the partial tables, no host hop) -----------------------
// svt_count_partial_device: count the batch (a rank's read block) and return the number of distinct entries; nothing is copied out.
int svt_count_partial_device(svt_ctx* c, const svt_batch* b, uint32_t k, uint8_t min_bq, const uint8_t* rc_flags, uint64_t* n_distinct) {
    if (!c || !b || !n_distinct) return svt_fail(c, SVT_ERR_ARG, "svt_count_partial_device: null argument");
    TRY(count_insert(c, b, k, min_bq, rc_flags));
    Carve cv; size_t icn = cv.add(16);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    ull* d_cnt = carve_ptr<ull>(c, cv, icn);
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, 16, c->stream));
    TRY(launch_ht_compact(c, 2, nullptr, nullptr, nullptr, d_cnt));           // null outputs: only the occupied slots are counted
    ull h[2];
    HIPCHK(c, memcpy_d2h(c, h, d_cnt, 16));
    HIPCHK(c, ctx_sync(c));
    c->ht_distinct = h[0]; c->tab_valid = false; c->tab_on_host = false;
    if (c->ht_fresh) { c->ht_hint_distinct = h[0]; c->ht_hint_positions = c->ht_positions; c->ht_fresh = false; }   // a rank's partial table: the size hint for its next block
    *n_distinct = h[0];
    return SVT_OK;
}
// all entries of this context's table into caller-provided DEVICE buffers (e.g. torch tensors) of capacity `cap` (unfiltered, unsorted)
int svt_count_export_device(svt_ctx* c, uint64_t* d_kmer, uint32_t* d_rev, uint32_t* d_fwd, uint64_t cap, uint64_t* n) {
    if (!c || !c->ht || !n || (cap && (!d_kmer || !d_rev || !d_fwd))) return svt_fail(c, SVT_ERR_ARG, "svt_count_export_device: null argument or no table");
    hipSetDevice(c->device);
    if (cap < c->ht_distinct) return svt_fail(c, SVT_ERR_OVERFLOW, "svt_count_export_device: capacity below the table's distinct entries");
    Carve cv; size_t icn = cv.add(16);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    ull* d_cnt = carve_ptr<ull>(c, cv, icn);
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, 16, c->stream));
    TRY(launch_ht_compact(c, 2, d_kmer, d_rev, d_fwd, d_cnt));
    ull h[2];
    HIPCHK(c, memcpy_d2h(c, h, d_cnt, 16));
    HIPCHK(c, ctx_sync(c));
    *n = h[1];
    return SVT_OK;
}
// an empty table sized for `total_entries` merged entries (the sum of every rank's export is an upper bound of the distinct keys)
int svt_count_merge_begin(svt_ctx* c, uint64_t total_entries) {
    if (!c) return SVT_ERR_ARG;
    hipSetDevice(c->device);
    TRY(ht_prepare(c, std::max<u64>(total_entries, 1)));
    c->ht_fresh = false;
    c->ht_distinct = 0; c->ht_positions = 0; c->tab_valid = false; c->tab_on_host = false;
    c->cnt_kmer.clear(); c->cnt_rev.clear(); c->cnt_fwd.clear();
    HIPCHK(c, ctx_sync(c));
    return SVT_OK;
}
// add n entries that already live in device memory (another rank's export after the all-gather, or this rank's own)
int svt_count_merge_device(svt_ctx* c, const uint64_t* d_kmer, const uint32_t* d_rev, const uint32_t* d_fwd, uint64_t n) {
    if (!c || !c->ht || (n && (!d_kmer || !d_rev || !d_fwd))) return svt_fail(c, SVT_ERR_ARG, "svt_count_merge_device: null argument or no table (svt_count_merge_begin first)");
    hipSetDevice(c->device);
    if ((c->ht_distinct + n) + (c->ht_distinct + n) / 2 > c->ht_cap) return svt_fail(c, SVT_ERR_OVERFLOW, "svt_count_merge_device: table too small (svt_count_merge_begin with the total first)");
    if (n) TRY(launch_ht_merge(c, d_kmer, d_rev, d_fwd, n));
    HIPCHK(c, ctx_sync(c));
    c->ht_distinct += n;        // upper bound until svt_count_finalize
    c->ht_positions += n * 3;   // keeps count_collect's kept <= positions/3 bound valid for merged tables
    return SVT_OK;
}
int svt_count_finalize(svt_ctx* c, uint32_t k, int single_strand, uint64_t* n_distinct, uint64_t* n_kept) {
    if (!c || !c->ht) return svt_fail(c, SVT_ERR_STATE, "svt_count_finalize: no table");
    TRY(check_k(c, k));
    hipSetDevice(c->device);
    return count_collect(c, k, single_strand ? 1 : 0, n_distinct, n_kept);
}

// ---- SNPmer table --------------------------------------------------------------------------------
int svt_set_snpmers(svt_ctx* c, uint32_t k, const uint64_t* split, const uint8_t* mid0, const uint8_t* mid1, const uint32_t* site_weight,
                    uint32_t n_sites, const uint64_t* high_freq, uint32_t n_hf) {
    if (!c || (n_sites && (!split || !mid0 || !mid1)) || (n_hf && !high_freq)) return svt_fail(c, SVT_ERR_ARG, "svt_set_snpmers: null argument");
    TRY(check_k(c, k));
    if (n_sites > SVT_MAX_SNPMER_SITES) return svt_fail(c, SVT_ERR_ARG, "svt_set_snpmers: " + std::to_string(n_sites) + " SNPmer sites; more than " + std::to_string(SVT_MAX_SNPMER_SITES) + " are not supported");
    hipSetDevice(c->device);
    ctx_sync(c);
    dfree(c->snp_keys); dfree(c->snp_vals); dfree(c->d_hf); dfree(c->snp_occ); c->snp_keys = nullptr; c->snp_vals = nullptr; c->d_hf = nullptr; c->snp_occ = nullptr;
    u32 cap = 16; while (cap < 8 * std::max<u32>(n_sites, 1)) cap <<= 1;
    std::vector<u64> keys(cap, SVT_EMPTY_KEY); std::vector<u32> vals(cap, 0);
    // internal bit position of a site = its rank by DESCENDING weight (ties: caller order).  True variant sites carry the
    // coverage of the sample, spurious ones a handful of reads, so the bits a read actually sets cluster in the first few
    // 64-bit words and the sparse rows of K6 stay short.  Any fixed bijection gives the same match/mismatch counts.
    c->site_order.resize(n_sites);
    for (u32 i = 0; i < n_sites; i++) c->site_order[i] = i;
    if (site_weight) std::stable_sort(c->site_order.begin(), c->site_order.end(), [&](u32 a, u32 b) { return site_weight[a] > site_weight[b]; });
    std::vector<u32> rank(n_sites);
    for (u32 r = 0; r < n_sites; r++) rank[c->site_order[r]] = r;
    for (u32 i = 0; i < n_sites; i++) {
        for (int al = 0; al < 2; al++) {
            u8 mid = al ? mid1[i] : mid0[i], other = al ? mid0[i] : mid1[i];
            u64 km = split[i] | ((u64)mid << (k - 1));                       // kmer_comp.rs:74-75
            u32 bit = mid > other ? 1u : 0u;                                 // allele bit = larger mid base
            u32 h = snp_slot_hash(km) & (cap - 1);
            while (keys[h] != SVT_EMPTY_KEY && keys[h] != km) h = (h + 1) & (cap - 1);
            keys[h] = km; vals[h] = (rank[i] << 1) | bit;
        }
    }
    TRY(dmalloc(c, &c->snp_keys, cap)); TRY(dmalloc(c, &c->snp_vals, cap));
    {   // occupancy of the slots, folded to at most 2^18 bits (K3 keeps it in LDS: a probe whose first slot is empty never leaves the CU)
        const u32 bits = std::min<u32>(cap, 1u << 18);
        std::vector<u32> occ(bits / 32 ? bits / 32 : 1, 0);
        for (u32 h = 0; h < cap; h++) if (keys[h] != SVT_EMPTY_KEY) { const u32 b = h & (bits - 1); occ[b >> 5] |= 1u << (b & 31); }
        TRY(dmalloc(c, &c->snp_occ, occ.size()));
        HIPCHK(c, hipMemcpy(c->snp_occ, occ.data(), occ.size() * 4, hipMemcpyHostToDevice));
        c->snp_occ_mask = std::max<u32>(bits, 32) - 1;
    }
    HIPCHK(c, hipMemcpy(c->snp_keys, keys.data(), (size_t)cap * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->snp_vals, vals.data(), (size_t)cap * 4, hipMemcpyHostToDevice));
    std::vector<u64> hf(high_freq, high_freq + n_hf);
    std::sort(hf.begin(), hf.end());
    TRY(dmalloc(c, &c->d_hf, n_hf));
    if (n_hf) HIPCHK(c, hipMemcpy(c->d_hf, hf.data(), (size_t)n_hf * 8, hipMemcpyHostToDevice));
    c->snp_mask = cap - 1; c->n_hf = n_hf; c->n_sites = n_sites; c->words = (n_sites + 63) / 64; c->k = k;
    return SVT_OK;
}

int svt_host_pin(svt_ctx* c, void* ptr, uint64_t bytes) {
    if (!c || !ptr || !bytes) return SVT_ERR_ARG;
    hipSetDevice(c->device);
    if (hipHostRegister(ptr, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return svt_fail(c, SVT_ERR_HIP, "svt_host_pin: hipHostRegister failed"); }
    return SVT_OK;
}
int svt_host_unpin(svt_ctx* c, void* ptr) {
    if (!c || !ptr) return SVT_ERR_ARG;
    hipSetDevice(c->device); ctx_sync(c);
    if (hipHostUnregister(ptr) != hipSuccess) { (void)hipGetLastError(); return svt_fail(c, SVT_ERR_HIP, "svt_host_unpin: hipHostUnregister failed"); }
    return SVT_OK;
}

// ---- multi-GPU tile sharding -------------------------------------------------------------------------
// RCCL (backend "nccl" on ROCm) bound at run time: the process usually holds one already (torch ships librccl.so.1; the loader returns the
// loaded copy for the same SONAME), a Rust caller gets /opt/rocm's.  A box without RCCL still loads the library; svt_set_shard_comm then fails.
namespace {
struct RcclApi {
    void* lib = nullptr; bool tried = false;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr; decltype(&ncclCommInitRank) CommInitRank = nullptr; decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr; decltype(&ncclGroupEnd) GroupEnd = nullptr; decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr; decltype(&ncclCommAbort) CommAbort = nullptr;
};
RcclApi g_rccl; std::mutex g_rccl_mu;
const RcclApi* rccl_api() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.tried) return g_rccl.lib ? &g_rccl : nullptr;
    g_rccl.tried = true;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) return nullptr;
    auto sym = [&](const char* n) { return dlsym(h, n); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId"); g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy"); g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd"); g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString"); g_rccl.CommAbort = (decltype(g_rccl.CommAbort))sym("ncclCommAbort");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.Broadcast) { dlclose(h); return nullptr; }
    g_rccl.lib = h;
    return &g_rccl;
}
}  // namespace
static_assert(SVT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "svt_shard_comm_id hands out an ncclUniqueId");

static void shard_comm_drop(svt_ctx* c) {
    if (!c->sh_comm) return;
    if (const RcclApi* R = rccl_api()) { hipSetDevice(c->device); ctx_sync(c); if (c->sh_comm) R->CommDestroy((ncclComm_t)c->sh_comm); }
    c->sh_comm = nullptr;
}
// A rank that cannot go on inside a sharded step -- a collective that timed out, an error between the Start and the End of a group, a failure the peers do not
// share -- ABORTS its communicator: its queued collectives leave the stream, and the peers' collectives stop making progress and end at their own deadline instead
// of waiting for ever.  Every later exchange on this context fails at once with SVT_ERR_EXCHANGE (sh_failed) until the caller sets a new communicator.
static void shard_comm_abort(svt_ctx* c, const char* why) {
    if (c->sh_failed) return;
    c->sh_failed = true; c->sh_fail_why = why ? why : "aborted"; c->sh_inflight = false;
    if (c->sh_comm) {
        const RcclApi* R = rccl_api();
        if (R && R->CommAbort) R->CommAbort((ncclComm_t)c->sh_comm);
        c->sh_comm = nullptr;                                      // aborted handles are not destroyed again
    }
}
int svt_shard_abort(svt_ctx* c, const char* why) {
    if (!c) return SVT_ERR_ARG;
    hipSetDevice(c->device);
    shard_comm_abort(c, why ? why : "svt_shard_abort");
    return SVT_OK;
}
int svt_set_shard(svt_ctx* c, uint32_t rank, uint32_t world, svt_exchange_fn exchange, void* user) {
    if (!c) return SVT_ERR_ARG;
    if (world > 32 || (world > 1 && rank >= world)) return svt_fail(c, SVT_ERR_ARG, "svt_set_shard: rank / world out of range (world <= 32)");
    shard_comm_drop(c);
    c->sh_failed = false; c->sh_fail_why.clear(); c->sh_inflight = false;
    if (world <= 1 || !exchange) { c->sh_rank = 0; c->sh_world = 1; c->sh_fn = nullptr; c->sh_user = nullptr; return SVT_OK; }
    c->sh_rank = rank; c->sh_world = world; c->sh_fn = exchange; c->sh_user = user;
    return SVT_OK;
}
int svt_shard_comm_id(uint8_t* id) {
    if (!id) return SVT_ERR_ARG;
    const RcclApi* R = rccl_api();
    if (!R) return SVT_ERR_STATE;
    ncclUniqueId u;
    if (R->GetUniqueId(&u) != ncclSuccess) return SVT_ERR_EXCHANGE;
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return SVT_OK;
}
int svt_set_shard_comm(svt_ctx* c, uint32_t rank, uint32_t world, const uint8_t* id) {
    if (!c) return SVT_ERR_ARG;
    if (world == 0 || world > 32 || rank >= world || !id) return svt_fail(c, SVT_ERR_ARG, "svt_set_shard_comm: rank / world out of range (1 <= world <= 32) or no id");
    const RcclApi* R = rccl_api();
    if (!R) return svt_fail(c, SVT_ERR_STATE, "svt_set_shard_comm: librccl.so.1 could not be loaded");
    hipSetDevice(c->device);
    shard_comm_drop(c);
    ncclUniqueId u; memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = R->CommInitRank(&comm, (int)world, u, (int)rank);
    if (r != ncclSuccess) return svt_fail(c, SVT_ERR_EXCHANGE, std::string("svt_set_shard_comm: ncclCommInitRank: ") + (R->GetErrorString ? R->GetErrorString(r) : "failed"));
    c->sh_comm = comm; c->sh_rank = rank; c->sh_world = world; c->sh_fn = nullptr; c->sh_user = nullptr;
    c->sh_failed = false; c->sh_fail_why.clear(); c->sh_inflight = false;
    return SVT_OK;
}
// "shard_world1" (a test option): a ONE-rank RCCL communicator runs the sharded code paths -- the rank's slice is everything, every exchange is a
// grouped broadcast from rank 0 to itself -- so that a one-GPU box exercises the RCCL calls on the library's stream
static inline bool shard_on(const svt_ctx* c) { return (c->sh_fn != nullptr || c->sh_comm != nullptr || c->sh_failed) && (c->sh_world > 1 || ((c->sh_comm != nullptr || c->sh_failed) && c->opt().shard_world1)); }
static inline bool sharded(const svt_ctx* c) { return shard_on(c) && !c->sh_paused; }
// contiguous split of n items: rank r owns [n r / W, n (r + 1) / W)
static inline u64 shard_lo(u64 n, u32 r, u32 W) { return n * r / W; }
// in-place all-gather-v of a device array split at elem_off (in elements).
//   RCCL communicator: ONE grouped collective -- a broadcast per non-empty slice, rooted at its owner, between ncclGroupStart / ncclGroupEnd -- on the
//     context's own stream: it is ordered behind the kernels that produced this rank's slice and before whatever the caller enqueues next; the
//     host does not wait here (callers that read the result on the host sync as they would after any kernel).
//   hook: the stream is synchronised first (the hook works outside it), then the hook runs to completion.
// an event in front of the first collective since the last completed wait: the deadline of that wait starts when the event has completed (ctx_sync_wait)
static void shard_mark(svt_ctx* c) {
    if (c->sh_inflight || c->sh_mark_set) return;
    if (!c->sh_mark && hipEventCreateWithFlags(&c->sh_mark, hipEventDisableTiming) != hipSuccess) { c->sh_mark = nullptr; (void)hipGetLastError(); return; }
    if (hipEventRecord(c->sh_mark, c->stream) == hipSuccess) c->sh_mark_set = true; else (void)hipGetLastError();
}
static int shard_exchange(svt_ctx* c, void* dev_base, u64 elem_bytes, const u64* elem_off) {
    if (c->sh_failed) return svt_fail(c, SVT_ERR_EXCHANGE, "shard exchange: the communicator of this context was aborted (" + c->sh_fail_why + ")");
    if (!c->sh_comm) HIPCHK(c, ctx_sync(c));
    if (elem_off[c->sh_world] == elem_off[0]) return SVT_OK;
    if (c->sh_depth == 0) c->sh_calls++;
    c->sh_bytes += (elem_off[c->sh_world] - elem_off[0]) * elem_bytes;
    if (c->sh_comm) {
        const RcclApi* R = rccl_api();
        if (c->sh_depth == 0) shard_mark(c);
        ncclResult_t r = R->GroupStart();
        for (u32 q = 0; q < c->sh_world && r == ncclSuccess; q++) {
            const u64 n = (elem_off[q + 1] - elem_off[q]) * elem_bytes;
            if (n == 0) continue;
            u8* at = (u8*)dev_base + elem_off[q] * elem_bytes;
            r = R->Broadcast(at, at, (size_t)n, ncclUint8, (int)q, (ncclComm_t)c->sh_comm, c->stream);
        }
        const ncclResult_t e = R->GroupEnd();
        if (r == ncclSuccess) r = e;
        c->sh_inflight = true;                                     // the next wait on this stream has a deadline (ctx_sync_wait)
        if (r != ncclSuccess) {
            const std::string why = std::string("RCCL grouped broadcast: ") + (R->GetErrorString ? R->GetErrorString(r) : "failed");
            shard_comm_abort(c, why.c_str());
            return svt_fail(c, SVT_ERR_EXCHANGE, "shard exchange (" + why + ")");
        }
        return SVT_OK;
    }
    if (c->sh_fn(c->sh_user, dev_base, elem_bytes, elem_off) != 0) { shard_comm_abort(c, "the shard exchange hook failed"); return svt_fail(c, SVT_ERR_EXCHANGE, "the shard exchange hook failed"); }
    return SVT_OK;
}
// Several arrays that travel together (the seed arrays of a read block, the two count arrays of K5): with an RCCL communicator they become ONE
// grouped collective (NCCL groups nest: the inner Start / End pairs of shard_exchange only count), counted as one exchange; with a hook every array
// is its own call, as before.
struct ShardGroup {
    svt_ctx* c; bool open = false;
    explicit ShardGroup(svt_ctx* c_) : c(c_) { if (c->sh_comm) { shard_mark(c); if (rccl_api()->GroupStart() == ncclSuccess) { open = true; c->sh_calls++; c->sh_depth++; } } }
    int close() {
        if (!open) return SVT_OK;
        open = false; c->sh_depth--;
        if (rccl_api()->GroupEnd() == ncclSuccess) return SVT_OK;
        shard_comm_abort(c, "RCCL group end failed");            // as the destructor path and shard_exchange do: the peers must not wait out their whole deadline
        return svt_fail(c, SVT_ERR_EXCHANGE, "shard exchange (RCCL group end) failed");
    }
    // left open: a TRY returned between Start and End.  The group must be ended (RCCL keeps the nesting depth per thread), but what it would enqueue is a PARTIAL
    // collective the peers' full one never matches: the communicator is aborted right behind it, so nobody waits on the mismatch
    ~ShardGroup() { if (open) { c->sh_depth--; rccl_api()->GroupEnd(); shard_comm_abort(c, "a rank-local error between the start and the end of a grouped exchange"); hipStreamSynchronize(c->stream); (void)hipGetLastError(); } }
};
static int shard_exchange_even(svt_ctx* c, void* dev_base, u64 elem_bytes, u64 n) {      // the split of shard_lo
    u64 off[33];
    for (u32 r = 0; r <= c->sh_world; r++) off[r] = shard_lo(n, r, c->sh_world);
    return shard_exchange(c, dev_base, elem_bytes, off);
}

// every rank contributes ONE region [start[r], start[r] + count[r]) of a device array (regions that do not tile the array: the SNPmer lists a
// rank allocates from its own cursor range): one hook call per source rank, in which only that rank's slice is non-empty
static int shard_exchange_regions(svt_ctx* c, void* dev_base, u64 elem_bytes, const u64* start, const u64* count) {
    for (u32 q = 0; q < c->sh_world; q++) {
        if (count[q] == 0) continue;
        u64 off[33];
        for (u32 r = 0; r <= c->sh_world; r++) off[r] = r <= q ? start[q] : start[q] + count[q];
        TRY(shard_exchange(c, dev_base, elem_bytes, off));
    }
    return SVT_OK;
}
// one u64 per rank -> all of them on every rank (counts, flags)
static int shard_allgather_u64(svt_ctx* c, u64 mine, u64* all) {
    ull* dcx = nullptr;
    TRY(dmalloc(c, &dcx, 32));
    int rc = [&]() -> int {
        HIPCHK(c, hipMemcpyAsync(dcx + c->sh_rank, &mine, 8, hipMemcpyHostToDevice, c->stream));
        u64 one[33]; for (u32 r = 0; r <= c->sh_world; r++) one[r] = r;
        TRY(shard_exchange(c, dcx, 8, one));
        HIPCHK(c, peek(c, dcx, all, 8 * c->sh_world));
        return SVT_OK;
    }();
    dfree(dcx);
    return rc;
}

int svt_shard_info(const svt_ctx* c, uint32_t* rank, uint32_t* world) {
    if (!c) return SVT_ERR_ARG;
    const bool on = shard_on(c);
    if (rank) *rank = on ? c->sh_rank : 0;
    if (world) *world = on ? c->sh_world : 1;
    return SVT_OK;
}
int svt_shard_pause(svt_ctx* c, int on) { if (!c) return SVT_ERR_ARG; const int was = c->sh_paused ? 1 : 0; c->sh_paused = on != 0; return was; }
int svt_shard_allgather_u64(svt_ctx* c, uint64_t mine, uint64_t* all) {
    if (!c || !all) return SVT_ERR_ARG;
    if (!shard_on(c)) { all[0] = mine; return SVT_OK; }
    hipSetDevice(c->device);
    return shard_allgather_u64(c, mine, all);
}
int svt_shard_allgatherv(svt_ctx* c, const void* mine, const uint64_t* bytes, void* all) {
    if (!c || !bytes || !all) return SVT_ERR_ARG;
    if (!shard_on(c)) { if (bytes[0]) memcpy(all, mine, bytes[0]); return SVT_OK; }
    hipSetDevice(c->device);
    u64 off[33]; off[0] = 0;
    for (u32 r = 0; r < c->sh_world; r++) off[r + 1] = off[r] + bytes[r];
    if (off[c->sh_world] == 0) return SVT_OK;
    u8* d = nullptr;
    TRY(dmalloc(c, &d, off[c->sh_world]));
    int rc = [&]() -> int {
        if (bytes[c->sh_rank]) HIPCHK(c, hipMemcpyAsync(d + off[c->sh_rank], mine, bytes[c->sh_rank], hipMemcpyHostToDevice, c->stream));
        TRY(shard_exchange(c, d, 1, off));
        HIPCHK(c, memcpy_d2h(c, all, d, off[c->sh_world]));
        HIPCHK(c, ctx_sync(c));
        return SVT_OK;
    }();
    dfree(d);
    return rc;
}
// C1 in one call (src/seq_parse.rs:434-487: the reference's consumer threads each own the k-mers with kmer % threads == t; here every rank owns
// the k-mers of its READ BLOCK and the partial tables meet by an all-gather).  This rank has counted its block (svt_count_partial_device); the
// library gathers every rank's entries -- one u64 per rank for the sizes, then the three entry arrays (k-mer, rev, fwd) as ONE grouped
// collective on device memory -- re-creates the table for the total, adds all entries in one launch (sums: the order is immaterial) and
// filters / sorts as svt_count_finalize does.  Every rank ends with the identical table.  Works while the tile slicing is paused (the caller deals
// the read blocks out itself).  Without a shard: svt_count_finalize.
int svt_count_shard_merge(svt_ctx* c, uint32_t k, int single_strand, uint64_t* n_distinct, uint64_t* n_kept) {
    if (!c || !c->ht) return svt_fail(c, SVT_ERR_STATE, "svt_count_shard_merge: no table (svt_count_partial_device first)");
    TRY(check_k(c, k));
    hipSetDevice(c->device);
    if (!shard_on(c)) return count_collect(c, k, single_strand ? 1 : 0, n_distinct, n_kept);
    const u32 Wd = c->sh_world;
    u64 cnt[32] = {0}, off[33]; off[0] = 0;
    TRY(shard_allgather_u64(c, c->ht_distinct, cnt));
    for (u32 r = 0; r < Wd; r++) off[r + 1] = off[r] + cnt[r];
    const u64 total = off[Wd];
    u64* gk = nullptr; u32* gr = nullptr; u32* gf = nullptr; ull* d_cnt = nullptr;
    TRY(dmalloc(c, &gk, total + 1)); TRY(dmalloc(c, &gr, total + 1)); TRY(dmalloc(c, &gf, total + 1)); TRY(dmalloc(c, &d_cnt, 2));
    int rc = [&]() -> int {
        HIPCHK(c, hipMemsetAsync(d_cnt, 0, 16, c->stream));
        const u64 o = off[c->sh_rank];
        TRY(launch_ht_compact(c, 2, gk + o, gr + o, gf + o, d_cnt));         // this rank's entries into its slice
        ull h[2] = {0, 0};
        HIPCHK(c, peek(c, d_cnt, h, 16));
        // a rank whose table changed since svt_count_partial_device must not leave alone: the peers would wait for it in the grouped exchange below.  The ranks
        // agree on the check (one more word per rank) and leave together
        u64 bad[32] = {0};
        TRY(shard_allgather_u64(c, h[1] != cnt[c->sh_rank] ? 1 : 0, bad));
        if (h[1] != cnt[c->sh_rank]) return svt_fail(c, SVT_ERR_STATE, "svt_count_shard_merge: the table changed since svt_count_partial_device");
        for (u32 r = 0; r < Wd; r++) if (bad[r]) return svt_fail(c, SVT_ERR_STATE, "svt_count_shard_merge: the table of rank " + std::to_string(r) + " changed since its svt_count_partial_device");
        { ShardGroup grp(c); TRY(shard_exchange(c, gk, 8, off)); TRY(shard_exchange(c, gr, 4, off)); TRY(shard_exchange(c, gf, 4, off)); TRY(grp.close()); }
        TRY(ht_prepare(c, std::max<u64>(total, 1)));
        c->ht_fresh = false;
        c->ht_distinct = 0; c->ht_positions = 0; c->tab_valid = false; c->tab_on_host = false;
        c->cnt_kmer.clear(); c->cnt_rev.clear(); c->cnt_fwd.clear();
        if (total) TRY(launch_ht_merge(c, gk, gr, gf, total));
        c->ht_distinct = total; c->ht_positions = total * 3;                  // upper bounds until the collect below (as svt_count_merge_device keeps them)
        HIPCHK(c, ctx_sync(c));
        return SVT_OK;
    }();
    dfree(gk); dfree(gr); dfree(gf); dfree(d_cnt);
    if (rc != SVT_OK) return rc;
    return count_collect(c, k, single_strand ? 1 : 0, n_distinct, n_kept);
}

// the dense rows of a batch whose seeds were extracted rank-sliced, gathered on first use by a path that reads them (every rank takes the same path)
static int ensure_dense_rows(svt_ctx* c, const svt_batch* b) {
    SeedsDev& s = const_cast<svt_batch*>(b)->seeds;
    if (!s.rows_partial) return SVT_OK;
    if (!sharded(c)) return svt_fail(c, SVT_ERR_STATE, "the SNPmer rows of this batch are partial and the shard is gone");
    u64 roff[33];
    for (u32 r = 0; r <= c->sh_world; r++) roff[r] = shard_lo(b->n, r, c->sh_world);
    { ShardGroup grp(c); TRY(shard_exchange(c, s.p_all, 8 * (u64)s.words, roff)); TRY(shard_exchange(c, s.p_filt, 8 * (u64)s.words, roff)); TRY(shard_exchange(c, s.allele, 8 * (u64)s.words, roff)); TRY(grp.close()); }
    s.rows_partial = false;
    return SVT_OK;
}

// ---- seeds ---------------------------------------------------------------------------------------
int svt_extract_seeds(svt_ctx* c, svt_batch* b, uint32_t k, uint32_t cpar, uint8_t min_bq, int use_qual) {
    if (!c || !b) return svt_fail(c, SVT_ERR_ARG, "svt_extract_seeds: null argument");
    TRY(check_k(c, k));
    if (cpar < 1 || cpar > 17 || cpar > k) return svt_fail(c, SVT_ERR_ARG, "c must be in 1..min(k,17)");
    if (k > 23) return svt_fail(c, SVT_ERR_ARG, "seed extraction needs k <= 23 (Kmer48, src/cli.rs:152)");
    if (!c->snp_keys || c->k != k) return svt_fail(c, SVT_ERR_STATE, "svt_extract_seeds: call svt_set_snpmers with the same k first");
    hipSetDevice(c->device);
    ctx_sync(c);
    free_seeds(b->seeds);
    SeedsDev& s = b->seeds;
    const u32 n = b->n;
    const u32 sl = k - cpar + 1, win = cpar, midw = (k - sl) / 2;
    if (sl <= 7 && !c->parent && (c->rank_s != sl || !c->d_rank)) {
        // mm_hash64 (src/seeding.rs:18-28) is a bijection and the syncmer test only compares its values (:527-537): the RANK of the hash of the canonical s-mer among all 4^s
        // forward s-mers decides exactly as the hash does.  rank[forward s-mer] as u16, 32 KB at s = 7
        auto mm = [](u64 key) { key = (~key) + (key << 21); key ^= key >> 24; key = (key + (key << 3)) + (key << 8); key ^= key >> 14; key = (key + (key << 2)) + (key << 4); key ^= key >> 28; key += key << 31; return key; };
        const u32 nt = 1u << (2 * sl);
        std::vector<std::pair<u64, u32>> hs; hs.reserve(nt);
        std::vector<u32> canon_of(nt);
        for (u32 v = 0; v < nt; v++) {
            u32 rc = 0; for (u32 j = 0; j < sl; j++) rc |= (3u - ((v >> (2 * j)) & 3u)) << (2 * (sl - 1 - j));
            canon_of[v] = std::min(v, rc);
            if (canon_of[v] == v) hs.push_back({mm((u64)v), v});
        }
        std::sort(hs.begin(), hs.end());
        std::vector<u16> rank_of_canon(nt, 0), table(std::max<u32>(nt, 4), 0);
        for (size_t i = 0; i < hs.size(); i++) rank_of_canon[hs[i].second] = (u16)i;
        for (u32 v = 0; v < nt; v++) table[v] = rank_of_canon[canon_of[v]];
        dfree(c->d_rank); c->d_rank = nullptr;
        TRY(dmalloc(c, &c->d_rank, table.size()));
        HIPCHK(c, hipMemcpy(c->d_rank, table.data(), table.size() * 2, hipMemcpyHostToDevice));
        c->rank_s = sl;
    }
    const u32 spacing = std::min(midw, win - 1 - midw) + 1;                  // two accepted syncmers are >= spacing apart
    std::vector<u64> mbase(n + 1), qoff(n + 1);
    u64 mc = 0, qb = 0; u32 maxm = 1;
    for (u32 i = 0; i < n; i++) {
        u64 len = b->h_off[i + 1] - b->h_off[i];
        u32 cap = len >= k ? (u32)((len - k + 1 + spacing - 1) / spacing + 1) : 0;
        mbase[i] = mc; mc += cap; maxm = std::max(maxm, cap);
        qoff[i] = qb; if (use_qual && b->has_qual) qb += ((len + 3) / 4 + 1) / 2;
    }
    mbase[n] = mc; qoff[n] = qb;
    s.k = k; s.c = cpar; s.mini_cap = mc; s.words = c->words; s.qb_bytes = qb;
    {   // the per-read records: what the host sends (region starts) and what it reads back after the kernel, each group contiguous
        size_t off = 0; auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 15) & ~(size_t)15; return o; };
        const size_t o_mb = take((size_t)(n + 1) * 8), o_qo = take((size_t)(n + 1) * 8);
        const size_t o_est = take((size_t)n * 8), o_setc = take((size_t)n * 4), o_nsol = take((size_t)n * 4), o_mcnt = take((size_t)n * 4), o_scnt = take((size_t)n * 4);
        const size_t o_ev = take(n), o_lv = take(n), o_st = take(n), o_cur = take(16);
        TRY(dmalloc(c, &s.meta_block, off));
        u8* mb = s.meta_block;
        s.mini_base = (u64*)(mb + o_mb); s.qb_off = (u64*)(mb + o_qo); s.est_id = (double*)(mb + o_est); s.set_cnt = (u32*)(mb + o_setc); s.n_solid = (u32*)(mb + o_nsol);
        s.mini_cnt = (u32*)(mb + o_mcnt); s.snp_cnt = (u32*)(mb + o_scnt); s.est_valid = mb + o_ev; s.lsh_valid = mb + o_lv; s.status = mb + o_st; s.snp_cursor = (ull*)(mb + o_cur);
        s.meta_fetch_off = o_est; s.meta_fetch_bytes = o_cur + 8 - o_est;
        StageUp stage(c, o_est);
        memcpy(stage.p + o_mb, mbase.data(), (size_t)(n + 1) * 8); memcpy(stage.p + o_qo, qoff.data(), (size_t)(n + 1) * 8);
        HIPCHK(c, stage.send(mb, o_est));
    }
    TRY(dmalloc(c, &s.mini_pos, mc)); TRY(dmalloc(c, &s.mini_kmer, mc)); TRY(dmalloc(c, &s.mini_flags, mc)); TRY(dmalloc(c, &s.set_kmer, mc));
    TRY(dmalloc(c, &s.snp_base, n)); TRY(dmalloc(c, &s.lsh, (u64)n * SVT_LSH_TABLES));
    if (qb) TRY(dmalloc(c, &s.qualbins, qb));
    u64 snp_cap = (u64)n * 64 + 4096 + (u64)std::min<u32>(n + 16, 8192) * 512;   // + what the waves of the rank-table K3 may leave unused of their last piece (kernels_seeds.hip: SEEDS_CHUNK)
    u32 maxs = 256;
    std::vector<u8> status(n);
    // svt_set_shard: this rank extracts the seeds of its read block only (K3, K4 and the bitset rows are per read), allocating its SNPmer lists
    // from its own range of the shared arrays; everything the other stages read is completed by the exchanges at the end
    const bool sh = sharded(c) && c->opt().shard_seeds && n >= 4096;
    const u32 Wd = sh ? c->sh_world : 1, rk = sh ? c->sh_rank : 0;
    const u32 r_lo = sh ? (u32)shard_lo(n, rk, Wd) : 0, r_hi = sh ? (u32)shard_lo(n, rk + 1, Wd) : n;
    u64 used[32] = {0}, sbase[32] = {0};
    for (int attempt = 0; attempt < 6; attempt++) {
        dfree(s.snp_pos); dfree(s.snp_kmer); dfree(s.snp_flags); s.snp_pos = nullptr; s.snp_kmer = nullptr; s.snp_flags = nullptr;
        TRY(dmalloc(c, &s.snp_pos, snp_cap)); TRY(dmalloc(c, &s.snp_kmer, snp_cap)); TRY(dmalloc(c, &s.snp_flags, snp_cap));
        s.snp_cap = snp_cap;
        const u64 share = snp_cap / Wd;
        for (u32 r = 0; r < Wd; r++) sbase[r] = (u64)r * share;
        const ull cur0[1] = {sbase[rk]};
        HIPCHK(c, hipMemcpyAsync(s.snp_cursor, cur0, 8, hipMemcpyHostToDevice, c->stream));
        size_t lds = 80 * 8 + (size_t)maxs * 12;
        if (lds > 160 * 1024) return svt_fail(c, SVT_ERR_ARG, "read too long for the seed kernel's LDS buffers");
        TRY(launch_seeds(c, b, k, cpar, min_bq, use_qual, maxm, maxs, r_lo, r_hi));
        ull cursor = 0;
        DownPack dn(c); dn.get(s.status, status.data(), n); dn.get(s.snp_cursor, &cursor, 8);          // neighbours in the meta block
        HIPCHK(c, dn.recv());
        HIPCHK(c, ctx_sync(c));
        dn.scatter();
        bool local_over = false;
        for (u32 i = r_lo; i < r_hi; i++) if (status[i] == 2) local_over = true;
        u64 need = cursor - cur0[0];                                                  // entries this rank's reads asked for
        used[rk] = need;
        if (sh) {   // the ranks decide together: the same capacity and the same layout everywhere
            u64 all[32];
            TRY(shard_allgather_u64(c, need | ((u64)local_over << 63), all));
            need = 0; local_over = false;
            for (u32 r = 0; r < Wd; r++) { used[r] = all[r] & ~((u64)1 << 63); need = std::max(need, used[r]); local_over |= (all[r] >> 63) != 0; }
        }
        if (need <= share && !local_over) break;
        if (need > share) snp_cap = (need + 4096) * Wd;
        else { maxs *= 4; if (maxs > 8192) return svt_fail(c, SVT_ERR_OVERFLOW, "a read has more than 8192 raw SNPmer hits"); }
        if (attempt == 5) return svt_fail(c, SVT_ERR_OVERFLOW, "SNPmer output buffers kept overflowing");
    }
    s.max_set = maxm;
    u32 np2 = 64; while (np2 < maxm) np2 <<= 1;
    TRY(launch_lsh_sets(c, b, np2, r_lo, r_hi));
    if (c->words) {
        TRY(dmalloc(c, &s.p_all, (u64)n * c->words)); TRY(dmalloc(c, &s.p_filt, (u64)n * c->words)); TRY(dmalloc(c, &s.allele, (u64)n * c->words));
        TRY(dmalloc(c, &s.nz_cnt, n)); TRY(dmalloc(c, &s.nz_idx, s.snp_cap)); TRY(dmalloc(c, &s.nz_pa, s.snp_cap)); TRY(dmalloc(c, &s.nz_pf, s.snp_cap)); TRY(dmalloc(c, &s.nz_a, s.snp_cap));
        TRY(launch_snp_bits(c, b, r_lo, r_hi));
    }
    if (sh) {
        u64 roff[33], moff[33], qo[33];
        for (u32 r = 0; r <= Wd; r++) { const u64 x = shard_lo(n, r, Wd); roff[r] = x; moff[r] = mbase[x]; qo[r] = qoff[x]; }
        ShardGroup grp(c);                                      // every array below in one grouped collective (RCCL)
        auto per_read = [&](void* p, u64 eb) -> int { return p ? shard_exchange(c, p, eb, roff) : SVT_OK; };
        // per-read records
        TRY(per_read(s.est_id, 8)); TRY(per_read(s.set_cnt, 4)); TRY(per_read(s.n_solid, 4)); TRY(per_read(s.mini_cnt, 4)); TRY(per_read(s.snp_cnt, 4));
        TRY(per_read(s.est_valid, 1)); TRY(per_read(s.lsh_valid, 1)); TRY(per_read(s.status, 1)); TRY(per_read(s.snp_base, 8)); TRY(per_read(s.lsh, 8 * SVT_LSH_TABLES));
        if (c->words) { TRY(per_read(s.nz_cnt, 4)); s.rows_partial = true; }    // the dense rows (3 x words x 8 B per read: 7 KB at 18k sites) stay on their owners: K6 reads the sparse form
        // the fixed-capacity minimizer regions and the quality bins follow the reads
        TRY(shard_exchange(c, s.set_kmer, 8, moff));            // K5 / K7 read the sorted sets; the raw minimizer lists (13 B per slot) stay partial
        s.mini_partial = true;
        if (qb) TRY(shard_exchange(c, s.qualbins, 1, qo));
        // the SNPmer lists: one region per rank
        TRY(shard_exchange_regions(c, s.snp_pos, 4, sbase, used)); TRY(shard_exchange_regions(c, s.snp_kmer, 8, sbase, used)); TRY(shard_exchange_regions(c, s.snp_flags, 1, sbase, used));
        if (c->words) { TRY(shard_exchange_regions(c, s.nz_idx, 4, sbase, used)); TRY(shard_exchange_regions(c, s.nz_pa, 8, sbase, used)); TRY(shard_exchange_regions(c, s.nz_pf, 8, sbase, used));
                        TRY(shard_exchange_regions(c, s.nz_a, 8, sbase, used)); }
        TRY(grp.close());
    }
    HIPCHK(c, ctx_sync(c));
    s.valid = true;
    return SVT_OK;
}

// ---- Stage 1c on the device: the intake filters and the read order (src/kmer_comp.rs:117,185,233,248 and src/main.rs:538) ----
int svt_twin_order(svt_ctx* c, const svt_batch* b, uint32_t min_len, uint32_t max_len, uint32_t cpar, double cutoff, uint32_t* n_kept, uint32_t* order, uint64_t* est_key) {
    if (!c || !b || !n_kept || !order || !est_key || cpar == 0) return svt_fail(c, SVT_ERR_ARG, "svt_twin_order: null argument");
    if (!b->seeds.valid) return svt_fail(c, SVT_ERR_STATE, "svt_twin_order: no seeds (call svt_extract_seeds)");
    hipSetDevice(c->device);
    const u32 n = b->n;
    *n_kept = 0;
    if (n == 0) return SVT_OK;
    size_t need = 0;
    TRY(launch_twin_order(c, b, min_len, max_len, cpar, cutoff, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, &need, 0));
    Carve cv; const size_t ifl = cv.add(n), ika = cv.add((size_t)n * 8), iia = cv.add((size_t)n * 4), iib = cv.add((size_t)n * 4), ikx = cv.add((size_t)n * 8), iky = cv.add((size_t)n * 8),
                         icn = cv.add(16), itm = cv.add(need + 16);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u8* dfl = carve_ptr<u8>(c, cv, ifl); u64* dka = carve_ptr<u64>(c, cv, ika); u32* dia = carve_ptr<u32>(c, cv, iia); u32* dib = carve_ptr<u32>(c, cv, iib);
    u64* dkx = carve_ptr<u64>(c, cv, ikx); u64* dky = carve_ptr<u64>(c, cv, iky); u32* dcn = carve_ptr<u32>(c, cv, icn); void* dtm = carve_ptr<char>(c, cv, itm);
    TRY(launch_twin_order(c, b, min_len, max_len, cpar, cutoff, dfl, dka, dia, dib, dkx, dky, dcn, 0, dtm, need + 16, nullptr, 0));
    u32 kept = 0;
    HIPCHK(c, peek(c, dcn, &kept, 4));
    if (kept > n) return svt_fail(c, SVT_ERR_STATE, "svt_twin_order: selection count out of range");
    *n_kept = kept;
    if (kept == 0) return SVT_OK;
    TRY(launch_twin_order(c, b, min_len, max_len, cpar, cutoff, dfl, dka, dia, dib, dkx, dky, dcn, kept, dtm, need + 16, nullptr, 1));
    HIPCHK(c, memcpy_d2h(c, order, dib, (size_t)kept * 4));
    HIPCHK(c, memcpy_d2h(c, est_key, dky, (size_t)kept * 8));
    HIPCHK(c, ctx_sync(c));
    return SVT_OK;
}
int svt_twin_gather(svt_ctx* c, const svt_batch* b, const uint32_t* order, uint32_t n, uint32_t* length, uint32_t* n_mini, uint32_t* n_unique, uint32_t* n_snp_filtered,
                    double* est_id, uint8_t* est_valid, uint64_t* lsh, uint8_t* lsh_valid) {
    if (!c || !b || (n && (!order || !length || !n_mini || !n_unique || !n_snp_filtered || !est_id || !est_valid || !lsh_valid))) return svt_fail(c, SVT_ERR_ARG, "svt_twin_gather: null argument");
    if (!b->seeds.valid) return svt_fail(c, SVT_ERR_STATE, "svt_twin_gather: no seeds (call svt_extract_seeds)");
    if (n == 0) return SVT_OK;
    for (u32 t = 0; t < n; t++) if (order[t] >= b->n) return svt_fail(c, SVT_ERR_ARG, "svt_twin_gather: read index outside the batch");
    hipSetDevice(c->device);
    Carve cv; const size_t io = cv.add((size_t)n * 4), il = cv.add((size_t)n * 4), im = cv.add((size_t)n * 4), iu = cv.add((size_t)n * 4), is = cv.add((size_t)n * 4), ie = cv.add((size_t)n * 8),
                         iv = cv.add(n), iw = cv.add(n), ih = cv.add(lsh ? (size_t)n * SVT_LSH_TABLES * 8 : 16);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u32* dord = carve_ptr<u32>(c, cv, io); u32* dl = carve_ptr<u32>(c, cv, il); u32* dm = carve_ptr<u32>(c, cv, im); u32* du = carve_ptr<u32>(c, cv, iu); u32* ds = carve_ptr<u32>(c, cv, is);
    double* de = carve_ptr<double>(c, cv, ie); u8* dv = carve_ptr<u8>(c, cv, iv); u8* dw = carve_ptr<u8>(c, cv, iw); u64* dh = carve_ptr<u64>(c, cv, ih);
    HIPCHK(c, hipMemcpyAsync(dord, order, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    TRY(launch_twin_gather(c, b, dord, n, dl, dm, du, ds, de, dv, dw, lsh ? dh : nullptr));
    DownPack dn(c); dn.get(dl, length, (size_t)n * 4); dn.get(dm, n_mini, (size_t)n * 4); dn.get(du, n_unique, (size_t)n * 4); dn.get(ds, n_snp_filtered, (size_t)n * 4);
    dn.get(de, est_id, (size_t)n * 8); dn.get(dv, est_valid, n); dn.get(dw, lsh_valid, n);
    HIPCHK(c, dn.recv());
    if (lsh) HIPCHK(c, memcpy_d2h(c, lsh, dh, (size_t)n * SVT_LSH_TABLES * 8));
    HIPCHK(c, ctx_sync(c));
    dn.scatter();
    return SVT_OK;
}

static int fetch_counts(svt_ctx* c, const svt_batch* b, std::vector<u32>& mc, std::vector<u32>& sc) {
    mc.resize(b->n); sc.resize(b->n);
    if (b->n == 0) return SVT_OK;
    DownPack dn(c); dn.get(b->seeds.mini_cnt, mc.data(), (size_t)b->n * 4); dn.get(b->seeds.snp_cnt, sc.data(), (size_t)b->n * 4);
    HIPCHK(c, dn.recv());
    HIPCHK(c, ctx_sync(c));
    dn.scatter();
    return SVT_OK;
}

int svt_seeds_sizes(svt_ctx* c, const svt_batch* b, uint64_t* n_mini, uint64_t* n_snp, uint64_t* n_qb) {
    if (!c || !b || !b->seeds.valid) return svt_fail(c, SVT_ERR_STATE, "svt_seeds_sizes: no seeds (call svt_extract_seeds)");
    hipSetDevice(c->device);
    std::vector<u32> mc, sc; TRY(fetch_counts(c, b, mc, sc));
    u64 a = 0, s2 = 0; for (u32 i = 0; i < b->n; i++) { a += mc[i]; s2 += sc[i]; }
    if (n_mini) *n_mini = a; if (n_snp) *n_snp = s2; if (n_qb) *n_qb = b->seeds.qb_bytes;
    return SVT_OK;
}

int svt_qualbin_mean(svt_ctx* c, const svt_batch* b, const double* table16, double* mean) {
    if (!c || !b || !table16 || !mean) return svt_fail(c, SVT_ERR_ARG, "svt_qualbin_mean: null argument");
    if (!b->seeds.valid || !b->seeds.qualbins) return svt_fail(c, SVT_ERR_STATE, "svt_qualbin_mean: no quality bins (svt_extract_seeds with use_qual = 1 first)");
    if (b->n == 0) return SVT_OK;
    hipSetDevice(c->device);
    Carve cv; size_t it = cv.add(16 * 8), io = cv.add((size_t)b->n * 8);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    double* dt = carve_ptr<double>(c, cv, it); double* dout = carve_ptr<double>(c, cv, io);
    HIPCHK(c, hipMemcpyAsync(dt, table16, 16 * 8, hipMemcpyHostToDevice, c->stream));
    TRY(launch_qualbin_mean(c, b, dt, dout));
    HIPCHK(c, memcpy_d2h(c, mean, dout, (size_t)b->n * 8));
    HIPCHK(c, ctx_sync(c));
    return SVT_OK;
}

int svt_seeds_fetch(svt_ctx* c, const svt_batch* b, const svt_seeds_out* o) {
    if (!c || !b || !o || !b->seeds.valid) return svt_fail(c, SVT_ERR_STATE, "svt_seeds_fetch: no seeds");
    hipSetDevice(c->device);
    const SeedsDev& s = b->seeds; const u32 n = b->n;
    std::vector<u32> mc(n), sc(n);
    if (n) {   // the counts and every per-read record the caller asked for: one copy of the meta block's fetch region
        DownPack dn(c);
        dn.get(s.mini_cnt, mc.data(), (size_t)n * 4); dn.get(s.snp_cnt, sc.data(), (size_t)n * 4);
        dn.get(s.est_id, o->est_id, (size_t)n * 8); dn.get(s.est_valid, o->est_valid, n); dn.get(s.lsh_valid, o->lsh_valid, n);
        dn.get(s.set_cnt, o->n_unique, (size_t)n * 4); dn.get(s.n_solid, o->n_solid, (size_t)n * 4); dn.get(s.status, o->status, n);
        HIPCHK(c, dn.recv());
        HIPCHK(c, ctx_sync(c));
        dn.scatter();
        if (o->lsh) { DownPack dl(c); dl.get(s.lsh, o->lsh, (size_t)n * SVT_LSH_TABLES * 8); HIPCHK(c, dl.recv()); HIPCHK(c, ctx_sync(c)); dl.scatter(); }
    }
    std::vector<u64> moff(n + 1), soff(n + 1);
    u64 a = 0, s2 = 0;
    for (u32 i = 0; i < n; i++) { moff[i] = a; a += mc[i]; soff[i] = s2; s2 += sc[i]; }
    moff[n] = a; soff[n] = s2;
    if (o->mini_off) memcpy(o->mini_off, moff.data(), (n + 1) * 8);
    if (o->snp_off) memcpy(o->snp_off, soff.data(), (n + 1) * 8);
    if (s.mini_partial && (o->mini_pos || o->mini_kmer || o->mini_flags)) {
        // the raw minimizer lists were left on their owners (svt_set_shard): gather them now -- every rank makes this call
        if (!sharded(c)) return svt_fail(c, SVT_ERR_STATE, "svt_seeds_fetch: the minimizer lists are partial and the shard is gone");
        std::vector<u64> mb(n + 1);
        HIPCHK(c, hipMemcpy(mb.data(), s.mini_base, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost));
        u64 moff2[33];
        for (u32 r = 0; r <= c->sh_world; r++) moff2[r] = mb[shard_lo(n, r, c->sh_world)];
        { ShardGroup grp(c); TRY(shard_exchange(c, s.mini_pos, 4, moff2)); TRY(shard_exchange(c, s.mini_kmer, 8, moff2)); TRY(shard_exchange(c, s.mini_flags, 1, moff2)); TRY(grp.close()); }
        const_cast<SeedsDev&>(s).mini_partial = false;
    }
    for (int which = 0; which < 2; which++) {
        u64 tot = which ? s2 : a;
        u32* hp = which ? o->snp_pos : o->mini_pos; u64* hk = which ? o->snp_kmer : o->mini_kmer; u8* hf = which ? o->snp_flags : o->mini_flags;
        if ((!hp && !hk && !hf) || tot == 0) continue;
        Carve cv; size_t io = cv.add((n + 1) * 8), ip = cv.add(tot * 4), ik = cv.add(tot * 8), iff = cv.add(tot);
        if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
        u64* d_off = carve_ptr<u64>(c, cv, io); u32* dp = carve_ptr<u32>(c, cv, ip); u64* dk = carve_ptr<u64>(c, cv, ik); u8* df = carve_ptr<u8>(c, cv, iff);
        HIPCHK(c, hipMemcpyAsync(d_off, which ? soff.data() : moff.data(), (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
        TRY(launch_csr_gather(c, b, which, d_off, dp, dk, df));
        DownPack dn(c);
        if (hp && hk) { dn.get(dp, hp, tot * 4); dn.get(dk, hk, tot * 8); dn.get(df, hf, tot); }       // everything: the three arrays are neighbours in the scratch
        else { if (hp) HIPCHK(c, memcpy_d2h(c, hp, dp, tot * 4)); if (hk) HIPCHK(c, memcpy_d2h(c, hk, dk, tot * 8));
               if (hf) HIPCHK(c, memcpy_d2h(c, hf, df, tot)); }
        HIPCHK(c, dn.recv());
        HIPCHK(c, ctx_sync(c));
        dn.scatter();
    }
    if (n) {
        if (o->qualbin_off) HIPCHK(c, hipMemcpy(o->qualbin_off, s.qb_off, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost));
        if (o->qualbins && s.qb_bytes) HIPCHK(c, hipMemcpy(o->qualbins, s.qualbins, s.qb_bytes, hipMemcpyDeviceToHost));
    }
    return SVT_OK;
}

// ---- Stage 2 candidate lists (K5c) -----------------------------------------------------------------
int svt_lsh_candidates(svt_ctx* c, const svt_batch* b, const uint32_t* q_idx, uint32_t n_q, const uint32_t* r_idx, uint32_t n_ref, const uint32_t* ref_limit,
                       uint32_t mode, uint32_t top_n, uint32_t cap, uint32_t capacity, uint32_t* out_cnt, uint32_t* out_off, uint32_t* out, uint32_t* n_out) {
    if (!c || !b || (n_q && (!q_idx || !out_cnt || !out_off || (capacity && !out))) || (n_ref && !r_idx)) return svt_fail(c, SVT_ERR_ARG, "svt_lsh_candidates: null argument");
    if (!b->seeds.valid || !b->seeds.lsh) return svt_fail(c, SVT_ERR_STATE, "svt_lsh_candidates: seeds missing");
    if (mode > 1 || cap == 0 || cap > 256) return svt_fail(c, SVT_ERR_ARG, "svt_lsh_candidates: mode is 0 or 1, cap 1..256");
    if (n_out) *n_out = 0;
    if (n_q == 0) return SVT_OK;
    for (u32 i = 0; i < n_q; i++) if (q_idx[i] >= b->n) return svt_fail(c, SVT_ERR_ARG, "svt_lsh_candidates: query index out of range");
    for (u32 j = 0; j < n_ref; j++) if (r_idx[j] >= b->n) return svt_fail(c, SVT_ERR_ARG, "svt_lsh_candidates: reference index out of range");
    hipSetDevice(c->device);
    if (n_ref == 0) { memset(out_cnt, 0, (size_t)n_q * 4); memset(out_off, 0, (size_t)n_q * 4); return SVT_OK; }
    Carve cv; const size_t iq = cv.add((size_t)n_q * 4), il = cv.add((size_t)n_q * 4), ir = cv.add((size_t)n_ref * 4), ic = cv.add((size_t)n_q * 8 + 8), io = cv.add((size_t)capacity * 8 + 8);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u32* dq = carve_ptr<u32>(c, cv, iq); u32* dl = carve_ptr<u32>(c, cv, il); u32* dr = carve_ptr<u32>(c, cv, ir); u32* dc = carve_ptr<u32>(c, cv, ic); u32* dout = carve_ptr<u32>(c, cv, io);
    u32* doff = dc + n_q; u32* dcur = dc + 2 * (size_t)n_q;                    // counts | offsets | cursor: fetched together
    UpPack up(c, cv); up.put(iq, q_idx, (size_t)n_q * 4); if (ref_limit) up.put(il, ref_limit, (size_t)n_q * 4); up.put(ir, r_idx, (size_t)n_ref * 4);
    HIPCHK(c, up.send());
    TRY(launch_lsh_candidates(c, b, dq, n_q, dr, n_ref, ref_limit ? dl : nullptr, mode, top_n, cap, capacity, dcur, dc, doff, dout));
    u32 cur = 0;
    { DownPack dn(c); dn.get(dc, out_cnt, (size_t)n_q * 4); dn.get(doff, out_off, (size_t)n_q * 4); dn.get(dcur, &cur, 4); HIPCHK(c, dn.recv()); HIPCHK(c, ctx_sync(c)); dn.scatter(); }
    const u32 used = std::min(cur, capacity);                                  // lists handed out beyond the capacity were flagged and not written
    if (used) { DownPack dn(c); dn.get(dout, out, (size_t)used * 8); HIPCHK(c, dn.recv()); HIPCHK(c, ctx_sync(c)); dn.scatter(); }
    if (n_out) *n_out = used;
    return SVT_OK;
}

// ---- K5/K7 -----------------------------------------------------------------------------------------
int svt_minimizer_shared_counts(svt_ctx* c, const svt_batch* A, const svt_batch* B, const uint32_t* a_idx, const uint32_t* b_idx, uint64_t n_pairs,
                                uint32_t* shared, uint32_t* same_strand) {
    if (!c || !A || !B || (n_pairs && (!a_idx || !b_idx || !shared))) return svt_fail(c, SVT_ERR_ARG, "svt_minimizer_shared_counts: null argument");
    if (!A->seeds.valid || !B->seeds.valid) return svt_fail(c, SVT_ERR_STATE, "svt_minimizer_shared_counts: seeds missing");
    if (n_pairs == 0) return SVT_OK;
    if (n_pairs > 0x7FFFFFFFull) return svt_fail(c, SVT_ERR_ARG, "too many pairs in one call");
    hipSetDevice(c->device);
    Carve cv; size_t iab = cv.add(n_pairs * 8), ism = cv.add(n_pairs * 8);       // {a, b} and {shared, same} contiguous: one DMA each way
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u32* da = carve_ptr<u32>(c, cv, iab); u32* db = da + n_pairs; u32* ds = carve_ptr<u32>(c, cv, ism); u32* dm = ds + n_pairs;
    if (sharded(c) && n_pairs >= 4096) {
        // this rank's slice of the pairs; the two count arrays are completed by the exchange (tile rows by block + all-gather)
        const u64 lo = shard_lo(n_pairs, c->sh_rank, c->sh_world), hi = shard_lo(n_pairs, c->sh_rank + 1, c->sh_world);
        StageUp st(c, n_pairs * 8);
        memcpy(st.p, a_idx, n_pairs * 4); memcpy(st.p + n_pairs * 4, b_idx, n_pairs * 4);
        HIPCHK(c, st.send(da, n_pairs * 8));
        if (hi > lo) TRY(launch_set_intersect(c, A, B, da + lo, db + lo, hi - lo, ds + lo, dm + lo));
        { ShardGroup grp(c); TRY(shard_exchange_even(c, ds, 4, n_pairs)); TRY(shard_exchange_even(c, dm, 4, n_pairs)); TRY(grp.close()); }
        DownPack dn(c); dn.get(ds, shared, n_pairs * 4); dn.get(dm, same_strand, n_pairs * 4);
        HIPCHK(c, dn.recv());
        HIPCHK(c, ctx_sync(c));
        dn.scatter();
        return SVT_OK;
    }
    if (n_pairs <= ((u64)1 << 20) && ensure_zero_copy(c, n_pairs * 16)) {
        // small call: the kernel reads the pair indices from, and writes its counts to, pinned host memory directly -- no copy engine
        // in the path (the first SDMA copy after the device sat idle for ~10 ms, which is where Stage 2 starts, took 20-30 ms in every
        // third or fourth step: rocprofv3 --memory-copy-trace), one launch and one wait
        u32* up = (u32*)c->zc; u32* down = up + 2 * n_pairs;
        memcpy(up, a_idx, n_pairs * 4); memcpy(up + n_pairs, b_idx, n_pairs * 4);
        TRY(launch_set_intersect(c, A, B, up, up + n_pairs, n_pairs, down, down + n_pairs));
        HIPCHK(c, ctx_sync(c));
        memcpy(shared, down, n_pairs * 4);
        if (same_strand) memcpy(same_strand, down + n_pairs, n_pairs * 4);
        return SVT_OK;
    }
    if (ensure_pinned(c, n_pairs * 16)) {
        u32* up = (u32*)c->pin; u32* down = up + 2 * n_pairs;
        memcpy(up, a_idx, n_pairs * 4); memcpy(up + n_pairs, b_idx, n_pairs * 4);
        HIPCHK(c, hipMemcpyAsync(da, up, n_pairs * 8, hipMemcpyHostToDevice, c->stream));
        TRY(launch_set_intersect(c, A, B, da, db, n_pairs, ds, dm));
        HIPCHK(c, memcpy_d2h(c, down, ds, n_pairs * (same_strand ? 8 : 4)));
        HIPCHK(c, ctx_sync(c));
        memcpy(shared, down, n_pairs * 4);
        if (same_strand) memcpy(same_strand, down + n_pairs, n_pairs * 4);
        return SVT_OK;
    }
    HIPCHK(c, hipMemcpyAsync(da, a_idx, n_pairs * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(db, b_idx, n_pairs * 4, hipMemcpyHostToDevice, c->stream));
    TRY(launch_set_intersect(c, A, B, da, db, n_pairs, ds, dm));
    HIPCHK(c, memcpy_d2h(c, shared, ds, n_pairs * 4));
    if (same_strand) HIPCHK(c, memcpy_d2h(c, same_strand, dm, n_pairs * 4));
    HIPCHK(c, ctx_sync(c));
    return SVT_OK;
}

// ---- K6 ---------------------------------------------------------------------------------------------
uint32_t svt_snpmer_words(const svt_ctx* c) { return c ? c->words : 0; }
int svt_snpmer_site_order(const svt_ctx* c, uint32_t* order) {
    if (!c || !order) return SVT_ERR_ARG;
    memcpy(order, c->site_order.data(), c->site_order.size() * 4);
    return SVT_OK;
}
int svt_snpmer_bits_fetch(svt_ctx* c, const svt_batch* b, uint64_t* p_all, uint64_t* p_filt, uint64_t* allele) {
    if (!c || !b || !b->seeds.valid) return svt_fail(c, SVT_ERR_STATE, "svt_snpmer_bits_fetch: seeds missing");
    hipSetDevice(c->device);
    size_t bytes = (size_t)b->n * b->seeds.words * 8;
    if (bytes == 0) return SVT_OK;
    TRY(ensure_dense_rows(c, b));
    if (p_all) HIPCHK(c, hipMemcpy(p_all, b->seeds.p_all, bytes, hipMemcpyDeviceToHost));
    if (p_filt) HIPCHK(c, hipMemcpy(p_filt, b->seeds.p_filt, bytes, hipMemcpyDeviceToHost));
    if (allele) HIPCHK(c, hipMemcpy(allele, b->seeds.allele, bytes, hipMemcpyDeviceToHost));
    return SVT_OK;
}
int svt_bitset_upload(svt_ctx* c, const uint64_t* presence, const uint64_t* allele, uint32_t n_rows, svt_bitset** out) {
    if (!c || !out || (n_rows && (!presence || !allele))) return svt_fail(c, SVT_ERR_ARG, "svt_bitset_upload: null argument");
    hipSetDevice(c->device);
    svt_bitset* s = new svt_bitset(); s->n_rows = n_rows; s->words = c->words;
    size_t cnt = (size_t)n_rows * c->words;
    TRY(dmalloc(c, &s->p, 2 * cnt)); s->a = s->p + cnt;                           // presence and allele rows in one block
    if (cnt) { HIPCHK(c, hipMemcpy(s->p, presence, cnt * 8, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(s->a, allele, cnt * 8, hipMemcpyHostToDevice)); }
    *out = s;
    return SVT_OK;
}
void svt_bitset_free(svt_ctx* c, svt_bitset* s) {
    if (!s) return;
    if (c) { hipSetDevice(c->device); ctx_sync(c); }
    dfree(s->p); delete s;                                                        // s->a lies in the same block
}

static const u64* view_ptr(const svt_batch* b, int view) { return view == SVT_VIEW_FILTERED ? b->seeds.p_filt : b->seeds.p_all; }

struct SegDescHost { u32 row_begin, n_rows, col_begin, n_reps; };
struct SegTileHost { u32 seg, row0; };
// rows_out: the entries come back row by row (out_row = n_rows + 1 offsets, out_col / out_mm in row order) instead of as unordered triples
static int compat_lists_seg_impl(svt_ctx* c, const svt_batch* R, int view, const uint32_t* row_idx, uint32_t n_rows, const uint32_t* seg_row_off,
                                 const uint32_t* col_idx, const uint32_t* seg_col_off, uint32_t n_seg, int filter, bool rows_out,
                                 uint32_t* out_row, uint32_t* out_col, uint32_t* out_mm, uint64_t cap, uint64_t* n_out) {
    if (!c || !R || !n_out || (n_rows && (!row_idx || !seg_row_off || !col_idx || !seg_col_off)) || (rows_out && !out_row)) return svt_fail(c, SVT_ERR_ARG, "svt_snpmer_compat_lists_seg: null argument");
    if (rows_out) memset(out_row, 0, ((size_t)n_rows + 1) * 4);
    if (rows_out && cap > 0xFFFFFFFFull) return svt_fail(c, SVT_ERR_ARG, "svt_snpmer_compat_rows_seg: the row offsets are 32-bit");
    if (!R->seeds.valid) return svt_fail(c, SVT_ERR_STATE, "svt_snpmer_compat_lists_seg: seeds missing");
    if (R->seeds.words != c->words) return svt_fail(c, SVT_ERR_STATE, "svt_snpmer_compat_lists_seg: the seeds were extracted with a different SNPmer table than the context holds now");
    *n_out = 0;
    const u32 W = c->words;
    if (n_rows == 0 || n_seg == 0 || W == 0) return SVT_OK;
    if (sizeof(SegDescHost) != seg_desc_bytes() || sizeof(SegTileHost) != seg_tile_bytes()) return svt_fail(c, SVT_ERR_STATE, "svt_snpmer_compat_lists_seg: record layouts differ");
    const int RT = compat_seg_rt(W);
    if (RT == 0) return svt_fail(c, SVT_ERR_TOOWIDE, "svt_snpmer_compat_lists_seg: the SNPmer rows do not fit the LDS tile");
    if (seg_row_off[n_seg] != n_rows) return svt_fail(c, SVT_ERR_ARG, "svt_snpmer_compat_lists_seg: seg_row_off does not end at n_rows");
    hipSetDevice(c->device);
    const u32 n_cols = seg_col_off[n_seg];
    std::vector<SegDescHost> segs(n_seg); std::vector<SegTileHost> tiles; std::vector<u32> row_seg(n_rows);
    u32 max_reps = 0;
    for (u32 s = 0; s < n_seg; s++) {
        const u32 nr = seg_row_off[s + 1] - seg_row_off[s], ncs = seg_col_off[s + 1] - seg_col_off[s];
        if (ncs < nr) return svt_fail(c, SVT_ERR_ARG, "svt_snpmer_compat_lists_seg: a segment's columns must end with its rows");
        segs[s] = SegDescHost{seg_row_off[s], nr, seg_col_off[s], ncs - nr};
        max_reps = std::max(max_reps, ncs - nr);
        for (u32 r = 0; r < nr; r += (u32)RT) tiles.push_back(SegTileHost{s, seg_row_off[s] + r});
        for (u32 r = 0; r < nr; r++) row_seg[seg_row_off[s] + r] = s;
    }
    // one upload: {row_idx | row_seg | col_idx | segs | tiles}
    const size_t w_rows = n_rows, w_cols = n_cols, w_seg = (size_t)n_seg * 4, w_tile = tiles.size() * 2;
    const size_t n_in = w_rows * 2 + w_cols + w_seg + w_tile;
    Carve cv;
    const size_t iin = cv.add(n_in * 4), ior = cv.add(cap * 12), icn = cv.add(16 + (size_t)n_seg * 4 + (size_t)n_rows * (rows_out ? 16 : 8));
    const size_t ioff = rows_out ? cv.add(((size_t)n_rows + 1) * 4) : 0, ipr = rows_out ? cv.add(cap * 8) : 0;
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u32* din = carve_ptr<u32>(c, cv, iin);
    u32* dri = din; u32* drs = dri + w_rows; u32* dci = drs + w_rows; u32* dsg = dci + w_cols; u32* dtl = dsg + w_seg;
    u32* dor = carve_ptr<u32>(c, cv, ior); u32* doc = dor + 1; u32* dom = dor + 2;       // the kernel writes (row, col, mm) records: entry d at [3 d]
    // counter | per-segment list lengths | per-row flags | (rows_out: records per row | fill cursors) | device-made column list
    ull* dcn = carve_ptr<ull>(c, cv, icn); u32* dnsel = (u32*)(dcn + 2); u32* dhas = dnsel + n_seg; u32* drc = dhas + n_rows; u32* dcur = drc + (rows_out ? n_rows : 0); u32* dsel = dcur + (rows_out ? n_rows : 0);
    u32* doff = rows_out ? carve_ptr<u32>(c, cv, ioff) : nullptr; u32* dpr = rows_out ? carve_ptr<u32>(c, cv, ipr) : nullptr;
    std::vector<u32> up(n_in);
    memcpy(up.data(), row_idx, w_rows * 4); memcpy(up.data() + w_rows, row_seg.data(), w_rows * 4); memcpy(up.data() + 2 * w_rows, col_idx, w_cols * 4);
    memcpy(up.data() + 2 * w_rows + w_cols, segs.data(), w_seg * 4); memcpy(up.data() + 2 * w_rows + w_cols + w_seg, tiles.data(), w_tile * 4);
    HIPCHK(c, hipMemcpyAsync(din, up.data(), n_in * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(dcn, 0, 16 + (size_t)n_seg * 4 + (size_t)n_rows * (rows_out ? 12 : 4), c->stream));       // counter, per-segment list lengths, per-row flags (and counts, cursors)
    ull cnt = 0;
    DownPack dn_rc(c);                                                          // rows_out: the records per row, into out_row[1 ..]
    if (sharded(c) && tiles.size() >= 2 * c->sh_world) {
        // this rank's contiguous slice of the row tiles (tiles are in row order).  Phase 0 flags the rows that met a representative: the flags
        // of all rows are completed by an exchange before every rank derives the (identical) unflagged-column lists; phase 1 again runs the own
        // tiles; the records of all ranks are then gathered behind one another (their order is immaterial to the caller).
        const u32 NT = (u32)tiles.size(), Wd = c->sh_world;
        const u32 t_lo = (u32)shard_lo(NT, c->sh_rank, Wd), t_hi = (u32)shard_lo(NT, c->sh_rank + 1, Wd);
        u64 row_off[33];
        for (u32 r = 0; r <= Wd; r++) { const u32 t = (u32)shard_lo(NT, r, Wd); row_off[r] = t < NT ? tiles[t].row0 : n_rows; }
        u32* dtmp = nullptr; ull* dcx = nullptr;
        TRY(dmalloc(c, &dtmp, cap * 3)); TRY(dmalloc(c, &dcx, 32));
        int rc = [&]() -> int {
            if (t_hi > t_lo) TRY(launch_compat_lists_seg(c, R->seeds, view, dri, n_rows, dtl + 2 * (size_t)t_lo, t_hi - t_lo, dsg, max_reps, R->seeds, view, dci, n_cols, W, filter, 0, dor, doc, dom, cap, dcn, dhas, dsel, dnsel));
            TRY(shard_exchange(c, dhas, 4, row_off));
            TRY(launch_unflagged_cols_seg(c, dhas, drs, dsg, n_rows, dsel, dnsel));
            if (t_hi > t_lo) TRY(launch_compat_lists_seg(c, R->seeds, view, dri, n_rows, dtl + 2 * (size_t)t_lo, t_hi - t_lo, dsg, max_reps, R->seeds, view, dci, n_cols, W, filter, 1, dor, doc, dom, cap, dcn, dhas, dsel, dnsel));
            ull mine = 0;
            HIPCHK(c, peek(c, dcn, &mine, 8));
            if (mine > cap) mine = cap + 1;                            // overflow is reported through the total below
            ull counts[32] = {0};
            HIPCHK(c, hipMemcpyAsync(dcx + c->sh_rank, &mine, 8, hipMemcpyHostToDevice, c->stream));
            u64 one[33]; for (u32 r = 0; r <= Wd; r++) one[r] = r;
            TRY(shard_exchange(c, dcx, 8, one));
            HIPCHK(c, peek(c, dcx, counts, 8 * Wd));
            u64 off[33]; off[0] = 0;
            for (u32 r = 0; r < Wd; r++) off[r + 1] = off[r] + counts[r];
            cnt = off[Wd];
            if (cnt > cap) return SVT_OK;
            if (mine) HIPCHK(c, hipMemcpyAsync(dtmp + 3 * off[c->sh_rank], dor, mine * 12, hipMemcpyDeviceToDevice, c->stream));
            TRY(shard_exchange(c, dtmp, 12, off));
            if (cnt) HIPCHK(c, hipMemcpyAsync(dor, dtmp, cnt * 12, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, ctx_sync(c));
            return SVT_OK;
        }();
        dfree(dtmp); dfree(dcx);
        if (rc != SVT_OK) return rc;
        if (rows_out && cnt && cnt <= cap) {
            TRY(launch_rec_row_count(c, dor, nullptr, cnt, cap, drc));
            dn_rc.get(drc, out_row + 1, (size_t)n_rows * 4);
            HIPCHK(c, dn_rc.recv());
            HIPCHK(c, ctx_sync(c));
            dn_rc.scatter();
        }
    } else {
    TRY(launch_compat_lists_seg(c, R->seeds, view, dri, n_rows, dtl, (u32)tiles.size(), dsg, max_reps, R->seeds, view, dci, n_cols, W, filter, 0, dor, doc, dom, cap, dcn, dhas, dsel, dnsel));
    TRY(launch_unflagged_cols_seg(c, dhas, drs, dsg, n_rows, dsel, dnsel));
    TRY(launch_compat_lists_seg(c, R->seeds, view, dri, n_rows, dtl, (u32)tiles.size(), dsg, max_reps, R->seeds, view, dci, n_cols, W, filter, 1, dor, doc, dom, cap, dcn, dhas, dsel, dnsel));
    if (rows_out) {                                                             // counted while the total is still on the device; the counts ride with the wait for it
        TRY(launch_rec_row_count(c, dor, dcn, 0, cap, drc));
        dn_rc.get(drc, out_row + 1, (size_t)n_rows * 4);
        HIPCHK(c, dn_rc.recv());
    }
    HIPCHK(c, peek(c, dcn, &cnt, 8));
    if (rows_out) dn_rc.scatter();
    }
    *n_out = cnt;
    prof_add_bytes(c, "k_compat_lists", 12.0 * (double)std::min<u64>(cnt, cap));
    if (cnt > cap) { if (rows_out) memset(out_row, 0, ((size_t)n_rows + 1) * 4); return svt_fail(c, SVT_ERR_OVERFLOW, "svt_snpmer_compat_lists_seg: output capacity too small"); }
    if (rows_out) {
        if (cnt == 0) { memset(out_row, 0, ((size_t)n_rows + 1) * 4); return SVT_OK; }
        for (u32 r = 0; r < n_rows; r++) out_row[r + 1] += out_row[r];           // counts -> offsets
        if (out_row[n_rows] != cnt) return svt_fail(c, SVT_ERR_STATE, "svt_snpmer_compat_rows_seg: the rows' counts do not add up to the records");
        UpPack up_off(c, cv); up_off.put(ioff, out_row, ((size_t)n_rows + 1) * 4);
        HIPCHK(c, up_off.send());
        TRY(launch_rec_fill(c, dor, cnt, doff, dcur, dpr, dpr + cnt));
        DownPack dn(c); dn.get(dpr, out_col, cnt * 4); dn.get(dpr + cnt, out_mm, cnt * 4);
        HIPCHK(c, dn.recv());
        HIPCHK(c, ctx_sync(c));
        dn.scatter();
        return SVT_OK;
    }
    if (cnt) {
        std::vector<u32> rec(cnt * 3);
        DownPack dn(c); dn.get(dor, rec.data(), cnt * 12);
        HIPCHK(c, dn.recv());
        HIPCHK(c, ctx_sync(c));
        dn.scatter();
        for (u64 i = 0; i < cnt; i++) { out_row[i] = rec[3 * i]; out_col[i] = rec[3 * i + 1]; out_mm[i] = rec[3 * i + 2]; }
    }
    return SVT_OK;
}

int svt_snpmer_compat_lists_seg(svt_ctx* c, const svt_batch* R, int view, const uint32_t* row_idx, uint32_t n_rows, const uint32_t* seg_row_off,
                                const uint32_t* col_idx, const uint32_t* seg_col_off, uint32_t n_seg, int filter,
                                uint32_t* out_row, uint32_t* out_col, uint32_t* out_mm, uint64_t cap, uint64_t* n_out) {
    return compat_lists_seg_impl(c, R, view, row_idx, n_rows, seg_row_off, col_idx, seg_col_off, n_seg, filter, false, out_row, out_col, out_mm, cap, n_out);
}
int svt_snpmer_compat_rows_seg(svt_ctx* c, const svt_batch* R, int view, const uint32_t* row_idx, uint32_t n_rows, const uint32_t* seg_row_off,
                               const uint32_t* col_idx, const uint32_t* seg_col_off, uint32_t n_seg, int filter,
                               uint32_t* out_off, uint32_t* out_col, uint32_t* out_mm, uint64_t cap, uint64_t* n_out) {
    return compat_lists_seg_impl(c, R, view, row_idx, n_rows, seg_row_off, col_idx, seg_col_off, n_seg, filter, true, out_off, out_col, out_mm, cap, n_out);
}

int svt_snpmer_compat_lists(svt_ctx* c, const svt_batch* R, int row_view, const uint32_t* row_idx, uint32_t n_rows,
                            const svt_batch* C, int col_view, const svt_bitset* S, const uint32_t* col_idx, uint32_t n_cols,
                            int filter, int triangular, uint32_t tri_base, const uint32_t* row_max_mismatch,
                            uint32_t* out_row, uint32_t* out_col, uint32_t* out_mm, uint64_t cap, uint64_t* n_out) {
    if (!c || !R || !n_out || (n_rows && !row_idx) || (!C && !S)) return svt_fail(c, SVT_ERR_ARG, "svt_snpmer_compat_lists: null argument");
    if (!R->seeds.valid || (C && !C->seeds.valid)) return svt_fail(c, SVT_ERR_STATE, "svt_snpmer_compat_lists: seeds missing");
    if (R->seeds.words != c->words || (C && C->seeds.words != c->words)) return svt_fail(c, SVT_ERR_STATE, "svt_snpmer_compat_lists: the seeds were extracted with a different SNPmer table than the context holds now");
    *n_out = 0;
    const u32 W = c->words;
    if (n_rows == 0 || n_cols == 0 || W == 0) return SVT_OK;
    if (C && !col_idx) return svt_fail(c, SVT_ERR_ARG, "svt_snpmer_compat_lists: col_idx required with a column batch");
    hipSetDevice(c->device);
    Carve cv;
    const size_t n_in = (size_t)n_rows * 2 + n_cols;                             // {row_idx | row_max | col_idx} contiguous: one upload
    size_t iin = cv.add(n_in * 4), icp = cv.add((size_t)n_cols * W * 16);
    size_t ior = cv.add(cap * 4), ioc = cv.add(cap * 4), iom = cv.add(cap * 4), icn = cv.add(16 + (size_t)n_rows * 8);   // counter | count of unflagged in-tile columns, the per-row "has a compatible column" flags, the list of unflagged in-tile columns
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u32* dri = carve_ptr<u32>(c, cv, iin); u32* drm = dri + n_rows; u32* dci = drm + n_rows; ulonglong2* dcp = carve_ptr<ulonglong2>(c, cv, icp);
    u32* dor = carve_ptr<u32>(c, cv, ior); u32* doc = carve_ptr<u32>(c, cv, ioc); u32* dom = carve_ptr<u32>(c, cv, iom); ull* dcn = carve_ptr<ull>(c, cv, icn);
    u32* dnsel = (u32*)(dcn + 1); u32* dhas = (u32*)(dcn + 2); u32* dsel = dhas + n_rows;
    // triangular == 2 (SVT_TRI_NEW_ONLY): of the in-tile columns (>= tri_base) only those whose own row has NO compatible column among
    // the first tri_base columns are reported -- two launches, the first over the old columns sets the per-row flag the second reads
    const bool split = C && triangular == 2 && tri_base > 0 && tri_base < n_cols && (u64)n_cols - tri_base <= n_rows;
    if (triangular == 2 && !split) triangular = 1;
    const bool pinned = ensure_pinned(c, n_in * 4 + 64);
    if (pinned) {
        u32* up = (u32*)c->pin;
        memcpy(up, row_idx, (size_t)n_rows * 4);
        if (row_max_mismatch) memcpy(up + n_rows, row_max_mismatch, (size_t)n_rows * 4);
        if (col_idx) memcpy(up + 2 * (size_t)n_rows, col_idx, (size_t)n_cols * 4);
        HIPCHK(c, hipMemcpyAsync(dri, up, n_in * 4, hipMemcpyHostToDevice, c->stream));
    } else {
        HIPCHK(c, hipMemcpyAsync(dri, row_idx, (size_t)n_rows * 4, hipMemcpyHostToDevice, c->stream));
        if (col_idx) HIPCHK(c, hipMemcpyAsync(dci, col_idx, (size_t)n_cols * 4, hipMemcpyHostToDevice, c->stream));
        if (row_max_mismatch) HIPCHK(c, hipMemcpyAsync(drm, row_max_mismatch, (size_t)n_rows * 4, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipMemsetAsync(dcn, 0, 16 + (split ? (size_t)n_rows * 4 : 0), c->stream));
    int cs = 1;                                                                  // columns of a batch: column-sparse kernel (no gather) when the dense rows fit LDS
    if (C && split) {
        cs = launch_compat_lists_cs(c, R->seeds, row_view, dri, n_rows, C->seeds, col_view, dci, tri_base, W, filter, 0, tri_base, row_max_mismatch ? drm : nullptr, dor, doc, dom, cap, dcn, 0, dhas);
        if (cs < 0) return cs;
        if (cs == 0) {
            TRY(launch_unflagged_cols(c, dhas, n_cols - tri_base, tri_base, dsel, dnsel));
            cs = launch_compat_lists_cs(c, R->seeds, row_view, dri, n_rows, C->seeds, col_view, dci, n_cols, W, filter, 1, tri_base, row_max_mismatch ? drm : nullptr, dor, doc, dom, cap, dcn, tri_base, nullptr, dsel, dnsel);
            if (cs < 0) return cs;
        }
        else triangular = 1;                                                     // dense rows do not fit LDS: the plain triangular lists (a superset) from the dense-column kernels
    } else if (C) { cs = launch_compat_lists_cs(c, R->seeds, row_view, dri, n_rows, C->seeds, col_view, dci, n_cols, W, filter, triangular, tri_base, row_max_mismatch ? drm : nullptr, dor, doc, dom, cap, dcn); if (cs < 0) return cs; }
    if (cs == 1) {
        if (C) { TRY(ensure_dense_rows(c, C)); TRY(launch_gather_cols_t(c, view_ptr(C, col_view), C->seeds.allele, dci, n_cols, W, dcp)); }
        else {
            if (!col_idx && n_cols != S->n_rows) return svt_fail(c, SVT_ERR_ARG, "svt_snpmer_compat_lists: n_cols != bitset rows");
            TRY(launch_gather_cols_t(c, S->p, S->a, col_idx ? dci : nullptr, n_cols, W, dcp));
        }
        TRY(launch_compat_lists(c, R->seeds, row_view, dri, n_rows, dcp, n_cols, W, filter, triangular, tri_base, row_max_mismatch ? drm : nullptr, dor, doc, dom, cap, dcn));
    }
    ull cnt = 0;
    ull* hcnt = pinned ? (ull*)((char*)c->pin + ((n_in * 4 + 15) & ~(size_t)15)) : &cnt;   // the pinned upload area is consumed once the kernels ran in stream order
    HIPCHK(c, memcpy_d2h(c, hcnt, dcn, 8));
    HIPCHK(c, ctx_sync(c));
    cnt = *hcnt;
    *n_out = cnt;
    prof_add_bytes(c, "k_compat_lists", 12.0 * (double)std::min<u64>(cnt, cap));
    if (cnt > cap) return svt_fail(c, SVT_ERR_OVERFLOW, "svt_snpmer_compat_lists: output capacity too small");
    if (cnt) {
        if (ensure_pinned(c, cnt * 12)) {
            u32* down = (u32*)c->pin;
            HIPCHK(c, memcpy_d2h(c, down, dor, cnt * 4));
            HIPCHK(c, memcpy_d2h(c, down + cnt, doc, cnt * 4));
            HIPCHK(c, memcpy_d2h(c, down + 2 * cnt, dom, cnt * 4));
            HIPCHK(c, ctx_sync(c));
            memcpy(out_row, down, cnt * 4); memcpy(out_col, down + cnt, cnt * 4); memcpy(out_mm, down + 2 * cnt, cnt * 4);
        } else {
            HIPCHK(c, memcpy_d2h(c, out_row, dor, cnt * 4));
            HIPCHK(c, memcpy_d2h(c, out_col, doc, cnt * 4));
            HIPCHK(c, memcpy_d2h(c, out_mm, dom, cnt * 4));
            HIPCHK(c, ctx_sync(c));
        }
    }
    return SVT_OK;
}

// a12-a14 fused (src/alignment.rs:1786-1846): K6 overlap lists -> K7 on every candidate pair -> the f64 filters and the per-read
// lowest-mismatch ties, all on device-resident lists; only the ties come back.
int svt_read_asv_ties(svt_ctx* c, const svt_batch* R, const uint32_t* row_idx, uint32_t n_rows, const svt_batch* A, uint32_t n_asvs,
                      const uint32_t* row_max_mismatch, double min_frac, double c_param,
                      uint32_t* tie_row, uint32_t* tie_col, uint8_t* tie_rev, uint32_t* tie_mismatches, uint64_t cap, uint64_t* n_ties, uint64_t* n_candidates) {
    if (!c || !R || !A || !n_ties || (n_rows && !row_idx) || (cap && (!tie_row || !tie_col || !tie_rev))) return svt_fail(c, SVT_ERR_ARG, "svt_read_asv_ties: null argument");
    if (!R->seeds.valid || !A->seeds.valid) return svt_fail(c, SVT_ERR_STATE, "svt_read_asv_ties: seeds missing");
    if (R->seeds.words != c->words || A->seeds.words != c->words) return svt_fail(c, SVT_ERR_STATE, "svt_read_asv_ties: the seeds were extracted with a different SNPmer table than the context holds now");
    *n_ties = 0; if (n_candidates) *n_candidates = 0;
    const u32 W = c->words;
    if (n_rows == 0 || n_asvs == 0 || W == 0) return SVT_OK;
    if (n_asvs > A->n) return svt_fail(c, SVT_ERR_ARG, "svt_read_asv_ties: n_asvs exceeds the batch");
    hipSetDevice(c->device);
    static u64 pair_hint = 0;                                                     // last call's candidate count: avoids the overflow retry of the first pass
    u64 pcap = std::max<u64>(std::max<u64>(4096, (u64)n_rows * 8), pair_hint + pair_hint / 8);
    for (int attempt = 0; attempt < 3; attempt++) {
        Carve cv;
        size_t iri = cv.add((size_t)n_rows * 4), irm = cv.add((size_t)n_rows * 4), ici = cv.add((size_t)n_asvs * 4), icp = cv.add((size_t)n_asvs * W * 16);
        size_t ior = cv.add(pcap * 4), ioc = cv.add(pcap * 4), iom = cv.add(pcap * 4), icn = cv.add(64), iai = cv.add(pcap * 4), ish = cv.add(pcap * 4), isa = cv.add(pcap * 4),
               ikp = cv.add(pcap), ilw = cv.add((size_t)n_rows * 4), itr = cv.add(cap * 4), itc = cv.add(cap * 4), itv = cv.add(cap), itm = cv.add(tie_mismatches ? cap * 4 : 4),
               isr = cv.add(pcap * 4), isc = cv.add(pcap * 4), ism2 = cv.add(pcap * 4), irn = cv.add((size_t)n_rows * 4), idn = cv.add(n_rows);
        if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
        u32* dri = carve_ptr<u32>(c, cv, iri); u32* drm = carve_ptr<u32>(c, cv, irm); u32* dci = carve_ptr<u32>(c, cv, ici); ulonglong2* dcp = carve_ptr<ulonglong2>(c, cv, icp);
        u32* dor = carve_ptr<u32>(c, cv, ior); u32* doc = carve_ptr<u32>(c, cv, ioc); u32* dom = carve_ptr<u32>(c, cv, iom); ull* dcn = carve_ptr<ull>(c, cv, icn);
        u32* dai = carve_ptr<u32>(c, cv, iai); u32* dsh = carve_ptr<u32>(c, cv, ish); u32* dsa = carve_ptr<u32>(c, cv, isa); u8* dkp = carve_ptr<u8>(c, cv, ikp);
        u32* dlw = carve_ptr<u32>(c, cv, ilw); u32* dtr = carve_ptr<u32>(c, cv, itr); u32* dtc = carve_ptr<u32>(c, cv, itc); u8* dtv = carve_ptr<u8>(c, cv, itv);
        u32* dtm = tie_mismatches ? carve_ptr<u32>(c, cv, itm) : nullptr;
        u32* dsr = carve_ptr<u32>(c, cv, isr); u32* dsc = carve_ptr<u32>(c, cv, isc); u32* dsm = carve_ptr<u32>(c, cv, ism2); u32* drmin = carve_ptr<u32>(c, cv, irn); u8* ddone = carve_ptr<u8>(c, cv, idn);
        std::vector<u32> cols(n_asvs); for (u32 i = 0; i < n_asvs; i++) cols[i] = i;
        UpPack up(c, cv); up.put(iri, row_idx, (size_t)n_rows * 4); up.put(irm, row_max_mismatch, (size_t)n_rows * 4); up.put(ici, cols.data(), (size_t)n_asvs * 4);
        HIPCHK(c, up.send());
        HIPCHK(c, hipMemsetAsync(dcn, 0, 64, c->stream));
        int cs = launch_compat_lists_cs(c, R->seeds, SVT_VIEW_ALL, dri, n_rows, A->seeds, SVT_VIEW_ALL, dci, n_asvs, W, SVT_LIST_OVERLAP, 0, 0, row_max_mismatch ? drm : nullptr, dor, doc, dom, pcap, dcn);
        if (cs < 0) return cs;
        if (cs == 1) {
            TRY(ensure_dense_rows(c, A));
            TRY(launch_gather_cols_t(c, view_ptr(A, SVT_VIEW_ALL), A->seeds.allele, dci, n_asvs, W, dcp));
            TRY(launch_compat_lists(c, R->seeds, SVT_VIEW_ALL, dri, n_rows, dcp, n_asvs, W, SVT_LIST_OVERLAP, 0, 0, row_max_mismatch ? drm : nullptr, dor, doc, dom, pcap, dcn));
        }
        ull cnt = 0;
        HIPCHK(c, peek(c, dcn, &cnt, 8));
        pair_hint = cnt;
        if (n_candidates) *n_candidates = cnt;
        if (cnt > pcap) { pcap = cnt + cnt / 16 + 1024; continue; }
        if (cnt == 0) return SVT_OK;
        // phase 1: the candidates at every read's lowest mismatch count; phase 2: the rest, only for reads phase 1 did not settle
        for (int phase = 0; phase < 2; phase++) {
            HIPCHK(c, hipMemsetAsync(dcn + 2 + phase, 0, 8, c->stream));
            if (phase == 0) { HIPCHK(c, hipMemsetAsync(drmin, 0xFF, (size_t)n_rows * 4, c->stream)); HIPCHK(c, hipMemsetAsync(ddone, 0, n_rows, c->stream)); }
            TRY(launch_candidate_select(c, dor, doc, dom, cnt, drmin, ddone, phase, dsr, dsc, dsm, dcn + 2 + phase));
            ull ns = 0;
            HIPCHK(c, peek(c, dcn + 2 + phase, &ns, 8));
            if (ns == 0) continue;
            HIPCHK(c, hipMemsetAsync(dlw, 0xFF, (size_t)n_rows * 4, c->stream));
            TRY(launch_tie_passes(c, dri, dsr, dsc, dsm, ns, dai, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr));
            TRY(launch_set_intersect(c, R, A, dai, dsc, ns, dsh, dsa));
            TRY(launch_tie_passes(c, dri, dsr, dsc, dsm, ns, dai, 1, dsh, dsa, R->seeds.set_cnt, A->seeds.set_cnt, min_frac, c_param, dlw, dkp, dtr, dtc, dtv, dtm, cap, dcn + 1, ddone));
        }
        ull nt = 0;
        HIPCHK(c, peek(c, dcn + 1, &nt, 8));
        *n_ties = nt;
        if (nt > cap) return svt_fail(c, SVT_ERR_OVERFLOW, "svt_read_asv_ties: output capacity too small");
        if (nt) {
            DownPack dn(c); dn.get(dtr, tie_row, nt * 4); dn.get(dtc, tie_col, nt * 4); dn.get(dtv, tie_rev, nt); if (tie_mismatches) dn.get(dtm, tie_mismatches, nt * 4);
            HIPCHK(c, dn.recv());                                   // the four lists are neighbours in the scratch: one copy
            HIPCHK(c, ctx_sync(c));
            dn.scatter();
        }
        return SVT_OK;
    }
    return svt_fail(c, SVT_ERR_STATE, "svt_read_asv_ties: candidate list kept growing");
}

int svt_snpmer_consensus(svt_ctx* c, const svt_batch* R, const uint64_t* cl_off, const uint32_t* members, uint32_t n_clusters,
                         uint64_t* presence, uint64_t* allele, svt_bitset** out_set) {
    if (!c || !R || (n_clusters && (!cl_off || !members))) return svt_fail(c, SVT_ERR_ARG, "svt_snpmer_consensus: null argument");
    if (!R->seeds.valid) return svt_fail(c, SVT_ERR_STATE, "svt_snpmer_consensus: seeds missing");
    if (R->seeds.words != c->words) return svt_fail(c, SVT_ERR_STATE, "svt_snpmer_consensus: the seeds were extracted with a different SNPmer table than the context holds now");
    hipSetDevice(c->device);
    const u32 W = c->words;
    svt_bitset* s = new svt_bitset(); s->n_rows = n_clusters; s->words = W;
    const size_t cnt = (size_t)n_clusters * W;
    TRY(dmalloc(c, &s->p, 2 * cnt)); s->a = s->p + cnt;                           // presence and allele rows in one block: one copy back
    if (cnt) {
        const u64 nmem = cl_off[n_clusters];
        const size_t cbytes = c->opt().consensus_dense ? 0 : consensus_counter_bytes(n_clusters, W);   // per-site counters of the sparse-row kernel (0: dense-row kernel)
        u64 max_cluster = 0;
        for (u32 i = 0; i < n_clusters; i++) max_cluster = std::max<u64>(max_cluster, cl_off[i + 1] - cl_off[i]);
        Carve cv; size_t io = cv.add((size_t)(n_clusters + 1) * 8), im = cv.add((size_t)nmem * 4), ic = cv.add(cbytes);
        if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
        u64* doff = carve_ptr<u64>(c, cv, io); u32* dmem = carve_ptr<u32>(c, cv, im);
        UpPack up(c, cv); up.put(io, cl_off, (size_t)(n_clusters + 1) * 8); up.put(im, members, (size_t)nmem * 4);
        HIPCHK(c, up.send());
        if (!cbytes) TRY(ensure_dense_rows(c, R));                 // the dense-row consensus kernel (option consensus_dense)
        TRY(launch_consensus(c, R->seeds, doff, dmem, n_clusters, nmem, W, s->p, s->a, cbytes ? carve_ptr<ull>(c, cv, ic) : nullptr, max_cluster));
        DownPack dn(c); dn.get(s->p, presence, cnt * 8); dn.get(s->a, allele, cnt * 8);
        HIPCHK(c, dn.recv());
        HIPCHK(c, ctx_sync(c));
        dn.scatter();
    }
    if (out_set) *out_set = s; else svt_bitset_free(c, s);
    return SVT_OK;
}

int svt_snpmer_best_column(svt_ctx* c, const svt_batch* R, int row_view, const uint32_t* row_idx, uint32_t n_rows, const svt_bitset* S,
                           const uint32_t* col_lo, const uint32_t* col_hi, uint32_t* best_col, uint32_t* best_score) {
    if (!c || !R || !S || (n_rows && (!row_idx || !best_col))) return svt_fail(c, SVT_ERR_ARG, "svt_snpmer_best_column: null argument");
    if (!R->seeds.valid) return svt_fail(c, SVT_ERR_STATE, "svt_snpmer_best_column: seeds missing");
    if (R->seeds.words != c->words || S->words != c->words) return svt_fail(c, SVT_ERR_STATE, "svt_snpmer_best_column: rows / columns were built with a different SNPmer table than the context holds now");
    if (n_rows == 0) return SVT_OK;
    const u32 W = c->words, NC = S->n_rows;
    hipSetDevice(c->device);
    if (W == 0 || NC == 0) { for (u32 i = 0; i < n_rows; i++) { best_col[i] = col_lo ? col_lo[i] : 0; if (best_score) best_score[i] = 0xFFFF; } return SVT_OK; }
    Carve cv; size_t iri = cv.add((size_t)n_rows * 4), ilo = cv.add((size_t)n_rows * 4), ihi = cv.add((size_t)n_rows * 4), icp = cv.add((size_t)NC * W * 16),
              ibc = cv.add((size_t)n_rows * 4), ibs = cv.add((size_t)n_rows * 4);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u32* dri = carve_ptr<u32>(c, cv, iri); u32* dlo = carve_ptr<u32>(c, cv, ilo); u32* dhi = carve_ptr<u32>(c, cv, ihi);
    ulonglong2* dcp = carve_ptr<ulonglong2>(c, cv, icp); u32* dbc = carve_ptr<u32>(c, cv, ibc); u32* dbs = carve_ptr<u32>(c, cv, ibs);
    UpPack up(c, cv); up.put(iri, row_idx, (size_t)n_rows * 4); up.put(ilo, col_lo, (size_t)n_rows * 4); up.put(ihi, col_hi, (size_t)n_rows * 4);
    HIPCHK(c, up.send());
    TRY(launch_gather_cols_t(c, S->p, S->a, nullptr, NC, W, dcp));
    TRY(launch_best_column(c, R->seeds, row_view, dri, n_rows, dcp, NC, W, col_lo ? dlo : nullptr, col_hi ? dhi : nullptr, dbc, dbs));
    DownPack dn(c); dn.get(dbc, best_col, (size_t)n_rows * 4); dn.get(dbs, best_score, (size_t)n_rows * 4);
    HIPCHK(c, dn.recv());
    HIPCHK(c, ctx_sync(c));
    dn.scatter();
    return SVT_OK;
}

// ---- K8 ---------------------------------------------------------------------------------------------
// K8a for pairs whose bands are known (wa): class of every pair (kernels_affine.hip), pairs of a class ordered by length, tasks of up to G neighbours, tasks in
// falling cost, ONE launch that draws them from a counter.  dtasks: device space for n_pairs + 2 8-byte words (the tasks; the counter in front).
// "k8a_queue" = 0: round 4's launch per class on side streams (kept for comparison and for the per-class ISA counts).
static int affine_launches(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx, const uint32_t* t_idx, uint64_t n_pairs, const u32* wa,
                           const u32* dq, const u32* dt, const u8* dr, u32* db, u32* dsel, int32_t* dn, int32_t* dsc, u64* dtasks, u32* dredo, const u8* packed_ok = nullptr) {
    const u32 lds_words = (Q->max_len + 15) / 16 + 2 + (T->max_len + 15) / 16 + 2;
    const bool queue = c->opt().k8a_queue != 0;
    const bool pk16 = queue && c->opt().k8a_pk16 != 0 && dredo != nullptr;      // the packed cell (kernels_affine.hip: aff16_pairs): classes AFF_NCLS .. AFF_NCLS + 6, 32 pairs per task
    constexpr int NC = AFF_NCLS + AFF16_NCLS;
    // class + length key of every pair; a counting sort by (class, falling length in 8-base steps) puts neighbours in length into one wave: the groups of a wave share
    // its loops (latest entry, earliest exit, longest pair)
    constexpr u32 LB = 2048;                                        // length buckets: (n + m) / 16, sequences are <= 16000 bases
    int cls_tab[512];
    for (u32 w = 0; w < 512; w++) cls_tab[w] = affine_class_of(w, lds_words, c->opt().k8a_g16 ? 16 : 8);
    std::vector<u32> key(n_pairs), steps(n_pairs);
    std::vector<u32> cnt((size_t)NC * LB + 1, 0);
    double bytes[NC] = {0}, cells[NC] = {0}; u64 n_cls[NC] = {0}; u64 n_packed = 0;
    for (u64 i = 0; i < n_pairs; i++) {
        const u32 w = wa[i];
        int cls = cls_tab[w];
        const u64 lq = Q->h_off[q_idx[i] + 1] - Q->h_off[q_idx[i]], lt = T->h_off[t_idx[i] + 1] - T->h_off[t_idx[i]];
        // bands <= 63 can go through the packed cell; it pays where the certificate (score >= n + m - 254) will hold: the caller says so per pair when it knows the pair's
        // unit-cost distance (packed_ok), else: bands <= 39 between sequences whose lengths differ by <= 64
        const int pkm = c->opt().k8a_pk16;
        if (pk16 && w <= (pkm == 1 ? 63u : 39u) && ((packed_ok && pkm != 3) ? packed_ok[i] != 0 : (w <= 39 && (lq > lt ? lq - lt : lt - lq) <= 64))) { const int c16 = affine16_class_of(w); if (c16 >= 0) { cls = c16; n_packed++; } }
        steps[i] = (u32)((lq + lt) / 2 + 1);
        key[i] = (u32)cls * LB + (LB - 1 - std::min<u32>((u32)((lq + lt) >> 4), LB - 1));
        cnt[key[i] + 1]++; n_cls[cls]++;
        bytes[cls] += (double)((lq + 3) / 4 + (lt + 3) / 4 + 24);
        cells[cls] += (double)lq * (double)(2 * w + 1);
    }
    if (!queue) {      // a class with few pairs is a launch that is all tail: its pairs ride along in the next wider class of the same lane split that runs anyway
        for (int cls = 0; cls + 1 < AFF_NCLS; cls++) {
            if (!n_cls[cls] || n_cls[cls] >= 4096) continue;
            int up = -1;
            for (int x = cls + 1; x < AFF_NCLS && x <= cls + 2; x++) if (n_cls[x] && AFF_G[x] == AFF_G[cls]) { up = x; break; }
            if (up < 0) continue;
            for (u64 i = 0; i < n_pairs; i++) if (key[i] / LB == (u32)cls) { cnt[key[i] + 1]--; key[i] += (u32)(up - cls) * LB; cnt[key[i] + 1]++; }
            n_cls[up] += n_cls[cls]; bytes[up] += bytes[cls]; cells[up] += cells[cls]; n_cls[cls] = 0; bytes[cls] = 0; cells[cls] = 0;
        }
    }
    for (size_t k = 1; k < cnt.size(); k++) cnt[k] += cnt[k - 1];
    std::vector<u32> all(n_pairs);
    { std::vector<u32> at(cnt.begin(), cnt.end() - 1); for (u64 i = 0; i < n_pairs; i++) all[at[key[i]]++] = (u32)i; }
    u64 so_of[NC + 1]; for (int cls = 0; cls <= NC; cls++) so_of[cls] = cnt[(size_t)cls * LB];
    double span_bytes = 0, span_cells = 0; for (int cls = 0; cls < NC; cls++) { span_bytes += bytes[cls]; span_cells += cells[cls]; }
    struct Task { u32 first, cc; };
    std::vector<Task> tasks; int max_g = 1; u32 n_packed_tasks = 0;
    if (queue) {
        std::vector<std::pair<double, Task>> tk;
        for (int cls = 0; cls < NC; cls++) {
            const u32 G = cls < AFF_NCLS ? (u32)AFF_G[cls] : 128u / (u32)AFF16_LG[cls - AFF_NCLS];              // the packed cell: 32 or 16 pairs per wave, no LDS
            if (cls < AFF_NCLS && so_of[cls + 1] > so_of[cls]) max_g = std::max(max_g, (int)G);
            if (cls >= AFF_NCLS && so_of[cls + 1] > so_of[cls]) { max_g = std::max(max_g, 8); n_packed_tasks += (u32)((so_of[cls + 1] - so_of[cls] + G - 1) / G); }   // a wave may rerun eight of the packed cell's pairs through the 32-bit cell: LDS for eight
            for (u64 p = so_of[cls]; p < so_of[cls + 1]; p += G) {
                const u32 n = (u32)std::min<u64>(G, so_of[cls + 1] - p);
                tk.push_back({affine_task_cost(cls, steps[all[p]] + 8), Task{(u32)p, n | ((u32)cls << 8)}});    // the first pair of a task is its longest (within a bucket's 16 bases)
            }
        }
        // falling cost -- the packed cell's tasks first: the pairs they give no certificate for are rerun by the waves between their later tasks (a full 32-bit task each
        // eight of them), which must not start when the queue is already empty
        std::stable_sort(tk.begin(), tk.end(), [](const auto& a, const auto& b) { const bool pa = (a.second.cc >> 8) >= AFF_NCLS, pb = (b.second.cc >> 8) >= AFF_NCLS; return pa != pb ? pa : a.first > b.first; });
        tasks.reserve(tk.size()); for (auto& t : tk) tasks.push_back(t.second);
        if (c->profiling()) for (int cls = 0; cls < NC; cls++) if (cells[cls] > 0) prof_note_units(c, (std::string(affine_class_name(cls)) + "_cells").c_str(), cells[cls]);
    }
    const size_t gap = (char*)dsel - (char*)db;                     // the callers carve the list right after the bands, and the tasks after the list: one copy for all
    const size_t gap2 = (char*)dtasks - (char*)db, tbytes = queue ? 8 + tasks.size() * 8 : 0;
    if ((char*)dsel >= (char*)(db + n_pairs) && gap <= n_pairs * 4 + 4096 && (char*)dtasks >= (char*)(dsel + n_pairs) && gap2 <= n_pairs * 8 + 8192) {
        StageUp st(c, gap2 + tbytes);
        memcpy(st.p, wa, n_pairs * 4); memcpy(st.p + gap, all.data(), n_pairs * 4);
        if (queue) { memset(st.p + gap2, 0, 8); memcpy(st.p + gap2 + 8, tasks.data(), tasks.size() * 8); }
        HIPCHK(c, st.send(db, gap2 + tbytes));
    } else {
        HIPCHK(c, hipMemcpyAsync(db, wa, n_pairs * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(dsel, all.data(), n_pairs * 4, hipMemcpyHostToDevice, c->stream));
        if (queue) { HIPCHK(c, hipMemsetAsync(dtasks, 0, 8, c->stream)); HIPCHK(c, hipMemcpyAsync(dtasks + 1, tasks.data(), tasks.size() * 8, hipMemcpyHostToDevice, c->stream)); }
    }
    if (queue) {
        TRY(launch_align_affine_queue(c, Q, T, dq, dt, dr, db, dsel, dtasks + 1, (u32)tasks.size(), (u32*)dtasks, max_g, dn, dsc, Q->max_len, T->max_len, span_bytes, span_cells, dredo, n_packed, n_packed_tasks));
        HIPCHK(c, ctx_sync(c));                                    // `all` / `tasks` are pageable: the copy has read them before they go
        if (n_packed) { u32 nr = 0; HIPCHK(c, peek(c, dredo, &nr, 4)); c->k8a_packed += n_packed; c->k8a_redo += nr; }   // svt_get_option "k8a_packed_pairs" / "k8a_redo_pairs"
        return SVT_OK;
    }
    // round 4: one launch per band class, every class on a stream of its own (the context's stream + up to seven side streams; the context's stream then waits for
    // them), the classes with the long waves first.  Whatever happens in the loop, the context's stream waits for every side stream that got work before this returns:
    // the kernels read the context's scratch buffer, which the next call may free or reuse behind a sync of c->stream alone.
    int order[AFF_NCLS], n_run = 0;
    for (int cls = 0; cls < AFF_NCLS; cls++) if (n_cls[cls]) order[n_run++] = cls;
    std::sort(order, order + n_run, [&](int a, int b) { return AFF_P[a] != AFF_P[b] ? AFF_P[a] > AFF_P[b] : cells[a] * AFF_P[a] / AFF_G[a] > cells[b] * AFF_P[b] / AFF_G[b]; });
    const bool multi = n_run > 1;
    ProfScope span(c, "k_align_affine_span", span_bytes, span_cells);     // first class launch .. last one done, on the context's stream: the classes' own event spans overlap each other
    constexpr int NS = svt_ctx::N_SIDE;
    if (multi && !c->side_go) {
        HIPCHK(c, hipEventCreateWithFlags(&c->side_go, hipEventDisableTiming));
        for (int s = 0; s < NS; s++) { HIPCHK(c, hipStreamCreateWithFlags(&c->side[s], hipStreamNonBlocking)); HIPCHK(c, hipEventCreateWithFlags(&c->side_done[s], hipEventDisableTiming)); }
    }
    if (multi) HIPCHK(c, hipEventRecord(c->side_go, c->stream));            // the uploads above are ordered before every class launch
    bool used[NS] = {};
    struct JoinSides {
        svt_ctx* c; bool* used; bool done = false;
        void run() { if (done) return; done = true; for (int s = 0; s < NS; s++) if (used[s]) { if (hipEventRecord(c->side_done[s], c->side[s]) == hipSuccess) hipStreamWaitEvent(c->stream, c->side_done[s], 0); else hipStreamSynchronize(c->side[s]); } }
        ~JoinSides() { run(); }
    } join{c, used};
    for (int x = 0; x < n_run; x++) {
        const int cls = order[x], lane = (n_run - 1 - x) % (NS + 1);        // the last (narrowest, largest) class on the context's stream, the others on side streams
        hipStream_t on = lane == 0 ? c->stream : c->side[lane - 1];
        if (lane != 0 && !used[lane - 1]) { HIPCHK(c, hipStreamWaitEvent(on, c->side_go, 0)); used[lane - 1] = true; }
        TRY(launch_align_affine(c, on, Q, T, dq, dt, dr, db, dsel + so_of[cls], so_of[cls + 1] - so_of[cls], cls, dn, dsc, Q->max_len, T->max_len, bytes[cls], cells[cls]));
    }
    join.run();                                                    // joined here, so that the sync below covers the side streams
    HIPCHK(c, ctx_sync(c));                                        // `all` is pageable: the copy has read it before it goes
    return SVT_OK;
}

static int align_nm_run(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx, const uint32_t* t_idx, const uint8_t* reverse,
                        const uint32_t* band, uint64_t n_pairs, int32_t* nm, int32_t* score, bool affine) {
    if (!c || !Q || !T || (n_pairs && (!q_idx || !t_idx || !band || !nm))) return svt_fail(c, SVT_ERR_ARG, "svt_align_nm: null argument");
    if (n_pairs == 0) return SVT_OK;
    if (n_pairs > 0x7FFFFFFFull) return svt_fail(c, SVT_ERR_ARG, "too many pairs in one call");
    if (Q->max_len > 16000 || T->max_len > 16000) return svt_fail(c, SVT_ERR_ARG, "svt_align_nm: sequences longer than 16000 bases are not supported");
    hipSetDevice(c->device);
    std::vector<u32> sel[3]; double bytes[3] = {0, 0, 0}, cells[3] = {0, 0, 0};
    for (u64 i = 0; i < n_pairs; i++) {
        if (q_idx[i] >= Q->n || t_idx[i] >= T->n) return svt_fail(c, SVT_ERR_ARG, "svt_align_nm: index out of range");
        u32 w = band[i];
        if (w > 511) return svt_fail(c, SVT_ERR_ARG, "svt_align_nm: band > 511");
        int cls = w <= 127 ? 0 : (w <= 255 ? 1 : 2);
        sel[cls].push_back((u32)i);
        u64 lq = Q->h_off[q_idx[i] + 1] - Q->h_off[q_idx[i]], lt = T->h_off[t_idx[i] + 1] - T->h_off[t_idx[i]];
        bytes[cls] += (double)((lq + 3) / 4 + (lt + 3) / 4 + 24);            // SURVEY 8d K8 algorithmic bytes
        cells[cls] += (double)lq * (double)(2 * w + 1);                       // DP cells inside the band (profile "units")
    }
    Carve cv; size_t iq = cv.add(n_pairs * 4), it = cv.add(n_pairs * 4), ir = cv.add(n_pairs), ib = cv.add(n_pairs * 4), is = cv.add(n_pairs * 4), itk = cv.add((n_pairs + 2) * 8), in_ = cv.add(n_pairs * 4);
    size_t isc = cv.add(n_pairs * 4);          // bands | class-ordered list | K8a tasks: neighbours, one upload
    const size_t ird = cv.add((n_pairs + 8) * 4);   // K8a's packed cell: count + the pairs it gave no certificate for
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u32* dq = carve_ptr<u32>(c, cv, iq); u32* dt = carve_ptr<u32>(c, cv, it); u8* dr = carve_ptr<u8>(c, cv, ir); u32* db = carve_ptr<u32>(c, cv, ib);
    int32_t* dn = carve_ptr<int32_t>(c, cv, in_); u32* dsel = carve_ptr<u32>(c, cv, is); int32_t* dsc = carve_ptr<int32_t>(c, cv, isc); u64* dtk = carve_ptr<u64>(c, cv, itk);
    u32* drd = carve_ptr<u32>(c, cv, ird);
    UpPack up(c, cv); up.put(iq, q_idx, n_pairs * 4); up.put(it, t_idx, n_pairs * 4); up.put(ir, reverse, n_pairs); up.put(ib, band, n_pairs * 4);
    HIPCHK(c, up.send());
    u64 so = 0;
    if (affine) TRY(affine_launches(c, Q, T, q_idx, t_idx, n_pairs, band, dq, dt, reverse ? dr : nullptr, db, dsel, dn, dsc, dtk, drd));
    for (int cls = 0; cls < 3 && !affine; cls++) {
        if (sel[cls].empty()) continue;
        HIPCHK(c, hipMemcpyAsync(dsel + so, sel[cls].data(), sel[cls].size() * 4, hipMemcpyHostToDevice, c->stream));
        const bool wavefront = c->opt().k8_kernel == 1;                                  // the anti-diagonal kernel (K9 without traceback)
        if (wavefront) TRY(launch_align(c, Q, T, dq, dt, reverse ? dr : nullptr, db, dsel + so, sel[cls].size(), cls == 0 ? 1 : (cls == 1 ? 2 : 4), dn, Q->max_len, T->max_len, bytes[cls], cells[cls]));
        else TRY(launch_align_bp(c, Q, T, dq, dt, reverse ? dr : nullptr, db, dsel + so, sel[cls].size(), cls == 0 ? 1 : (cls == 1 ? 2 : 4), dn, bytes[cls], cells[cls]));
        so += sel[cls].size();
    }
    DownPack dn_(c); dn_.get(dn, nm, n_pairs * 4); if (affine) dn_.get(dsc, score, n_pairs * 4);
    HIPCHK(c, dn_.recv());
    HIPCHK(c, ctx_sync(c));
    dn_.scatter();
    return SVT_OK;
}
int svt_align_nm(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx, const uint32_t* t_idx, const uint8_t* reverse,
                 const uint32_t* band, uint64_t n_pairs, int32_t* nm) {
    return align_nm_run(c, Q, T, q_idx, t_idx, reverse, band, n_pairs, nm, nullptr, false);
}
// K8a near the unit-cost optimum: the forward pass of the bit-parallel K9 gives every pair its unit-cost distance d and the diagonal e of
// its end cell; an overlap alignment of cost d ending on e stays within |j - i| <= |e| + d, and the affine DP runs in
// |j - i| <= min(band, |e| + d + 8) -- four pairs per wavefront up to 47, two up to 95 (kernels_affine.hip).  The forward pass runs inside min(band, 255).
int svt_align_nm_affine_near(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx, const uint32_t* t_idx, const uint8_t* reverse,
                             const uint32_t* band, uint64_t n_pairs, int32_t* nm, int32_t* score, uint32_t* band_used) {
    if (!c || !Q || !T || (n_pairs && (!q_idx || !t_idx || !band || !nm))) return svt_fail(c, SVT_ERR_ARG, "svt_align_nm_affine_near: null argument");
    if (n_pairs == 0) return SVT_OK;
    if (n_pairs > 0x7FFFFFFFull) return svt_fail(c, SVT_ERR_ARG, "too many pairs in one call");
    if (Q->max_len > 16000 || T->max_len > 16000) return svt_fail(c, SVT_ERR_ARG, "svt_align_nm_affine_near: sequences longer than 16000 bases are not supported");
    hipSetDevice(c->device);
    std::vector<u32> sel[2], wf(n_pairs); double fcells[2] = {0, 0};   // forward pass: bands <= 127 / <= 255 (wider ones clipped to 255), and the band cells it covers (profile units)
    for (u64 i = 0; i < n_pairs; i++) {
        if (q_idx[i] >= Q->n || t_idx[i] >= T->n) return svt_fail(c, SVT_ERR_ARG, "svt_align_nm_affine_near: index out of range");
        if (band[i] > 511) return svt_fail(c, SVT_ERR_ARG, "svt_align_nm_affine_near: band > 511");
        wf[i] = std::min<u32>(band[i], 255);                     // the forward pass carries at most 511 band cells: wider bands are walked in their inner 255
        { const int cl = wf[i] <= 127 ? 0 : 1; sel[cl].push_back((u32)i); fcells[cl] += (double)(Q->h_off[q_idx[i] + 1] - Q->h_off[q_idx[i]]) * (double)(2 * wf[i] + 1); }
    }
    const u64 nk = sel[0].size() + sel[1].size();
    Carve cv; size_t iq = cv.add(n_pairs * 4), it = cv.add(n_pairs * 4), ir = cv.add(n_pairs), ib = cv.add(n_pairs * 4), is = cv.add(n_pairs * 4), itk = cv.add((n_pairs + 2) * 8), in_ = cv.add(n_pairs * 4);
    size_t isc = cv.add(n_pairs * 4), ik = cv.add(nk * 8);
    const size_t ird = cv.add((n_pairs + 8) * 4);   // K8a's packed cell: count + the pairs it gave no certificate for
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u32* drd = carve_ptr<u32>(c, cv, ird);
    u32* dq = carve_ptr<u32>(c, cv, iq); u32* dt = carve_ptr<u32>(c, cv, it); u8* dr = carve_ptr<u8>(c, cv, ir); u32* db = carve_ptr<u32>(c, cv, ib);
    int32_t* dn = carve_ptr<int32_t>(c, cv, in_); u32* dsel = carve_ptr<u32>(c, cv, is); int32_t* dsc = carve_ptr<int32_t>(c, cv, isc); u64* dkeys = carve_ptr<u64>(c, cv, ik); u64* dtk = carve_ptr<u64>(c, cv, itk);
    std::vector<u32> wa(band, band + n_pairs);
    std::vector<u8> pk_ok(n_pairs, 0);                            // K8a's packed cell is worth a try (its certificate will very likely hold)
    std::vector<u32> all(sel[0]); all.insert(all.end(), sel[1].begin(), sel[1].end());
    UpPack up(c, cv); up.put(iq, q_idx, n_pairs * 4); up.put(it, t_idx, n_pairs * 4); up.put(ir, reverse, n_pairs); up.put(ib, wf.data(), n_pairs * 4); up.put(is, all.data(), nk * 4);
    HIPCHK(c, up.send());
    if (nk) {
        u64 so = 0;
        for (int cls = 0; cls < 2; cls++) {
            if (sel[cls].empty()) continue;
            TRY(launch_align_tb_bp(c, Q, T, dq, dt, reverse ? dr : nullptr, db, dsel + so, sel[cls].size(), cls == 0 ? 1 : 2, dn, T->max_len, nullptr, nullptr, nullptr, nullptr, 3, dkeys + so, nullptr, nullptr, fcells[cls]));
            so += sel[cls].size();
        }
        std::vector<u64> hk(nk);
        DownPack dk(c); dk.get(dkeys, hk.data(), nk * 8);             // through the page-locked staging buffer: a copy into pageable memory queued behind the kernel spins on a core while it runs (2.4 ms per call)
        HIPCHK(c, dk.recv());
        HIPCHK(c, ctx_sync(c));
        dk.scatter();
        for (u64 g = 0; g < nk; g++) {
            if (hk[g] == ~0ull) continue;                          // no end cell inside the band: the band stays
            const u64 d = hk[g] >> 40; const int e = (int)(hk[g] & 0xFFFFF) - 2048;
            const u32 i = all[g];
            wa[i] = (u32)std::min<u64>(band[i], (u64)(e < 0 ? -e : e) + d + 8);
            // the packed cell certifies a result whose score is >= n + m - 254: an overlap of unit cost d on diagonal e scores about 2 min(n, m) - 6 d - 2 |e| locally
            const u64 lq = Q->h_off[q_idx[i] + 1] - Q->h_off[q_idx[i]], lt = T->h_off[t_idx[i] + 1] - T->h_off[t_idx[i]];
            pk_ok[i] = 6 * d + 2 * (u64)(e < 0 ? -e : e) + (lq > lt ? lq - lt : lt - lq) <= 236 ? 1 : 0;
        }
    }
    TRY(affine_launches(c, Q, T, q_idx, t_idx, n_pairs, wa.data(), dq, dt, reverse ? dr : nullptr, db, dsel, dn, dsc, dtk, drd, pk_ok.data()));
    DownPack dn_(c); dn_.get(dn, nm, n_pairs * 4); dn_.get(dsc, score, n_pairs * 4);
    HIPCHK(c, dn_.recv());
    HIPCHK(c, ctx_sync(c));
    dn_.scatter();
    if (band_used) memcpy(band_used, wa.data(), n_pairs * 4);
    return SVT_OK;
}

// K8a: same pairs and bands, minimap2-style nm of the best local two-piece-affine alignment (kernels_affine.hip)
int svt_align_nm_affine(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx, const uint32_t* t_idx, const uint8_t* reverse,
                        const uint32_t* band, uint64_t n_pairs, int32_t* nm, int32_t* score) {
    return align_nm_run(c, Q, T, q_idx, t_idx, reverse, band, n_pairs, nm, score, true);
}

// K9 for all pairs; rows land in d_cells (device, total = cell_off[n_pairs] u64) at cell_off[pair]; span / nm go to the host
static int pileup_run(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx, const uint32_t* t_idx, const uint8_t* reverse,
                      const uint32_t* band, uint64_t n_pairs, const uint64_t* cell_off, u64* d_cells, uint32_t* span, int32_t* nm, const char* who) {
    if (Q->max_len > 16000 || T->max_len > 16000) return svt_fail(c, SVT_ERR_ARG, std::string(who) + ": sequences longer than 16000 bases are not supported");
    std::vector<u32> sel[4];                                                         // bands <= 127 | <= 255 | <= 383 | <= 511
    for (u64 i = 0; i < n_pairs; i++) {
        if (q_idx[i] >= Q->n || t_idx[i] >= T->n) return svt_fail(c, SVT_ERR_ARG, std::string(who) + ": index out of range");
        if (band[i] > 511) return svt_fail(c, SVT_ERR_ARG, std::string(who) + ": band > 511");
        if (cell_off[i + 1] - cell_off[i] < Q->h_off[q_idx[i] + 1] - Q->h_off[q_idx[i]]) return svt_fail(c, SVT_ERR_ARG, std::string(who) + ": cell_off must follow the query lengths");
        sel[band[i] <= 127 ? 0 : (band[i] <= 255 ? 1 : (band[i] <= 383 ? 2 : 3))].push_back((u32)i);
    }
    // the bit-parallel kernel walks 64 pairs per wave and stores one direction window per pair-column [column][lane]: neighbours in target length share a slab, so a
    // slab's last columns are not rows of a few long stragglers (partial 512-byte rows: PMC WRITE_SIZE 32 KB per pair in the bench's call against 26 KB for equal lengths)
    for (int cls = 0; cls < 3; cls++) {
        if (sel[cls].size() < 128) continue;
        std::vector<u32> cnt(16002, 0), out(sel[cls].size());                      // a counting sort by falling target length (<= 16000), stable in input order: O(n), no comparisons
        auto key = [&](u32 i) -> u32 { return 16000u - (u32)std::min<u64>(16000, T->h_off[t_idx[i] + 1] - T->h_off[t_idx[i]]); };
        for (u32 i : sel[cls]) cnt[key(i) + 1]++;
        for (size_t k = 1; k < cnt.size(); k++) cnt[k] += cnt[k - 1];
        for (u32 i : sel[cls]) out[cnt[key(i)]++] = i;
        sel[cls].swap(out);
    }
    // pair descriptors stay resident for all chunks: {q | t | band | nm | reverse} in one block, sent with one copy
    u32* dblock = nullptr;
    const size_t wr = (n_pairs + 3) / 4;                                           // reverse flags, in u32 units
    int rc = dmalloc(c, &dblock, n_pairs * 4 + wr);
    u32* dq = dblock; u32* dt = dblock + n_pairs; u32* db = dblock + 2 * n_pairs; u8* dr = (u8*)(dblock + 3 * n_pairs); int32_t* dn = (int32_t*)(dblock + 3 * n_pairs + wr);
    if (rc == SVT_OK) {
        StageUp stage(c, (3 * n_pairs + wr) * 4);
        memcpy(stage.p, q_idx, n_pairs * 4); memcpy(stage.p + n_pairs * 4, t_idx, n_pairs * 4); memcpy(stage.p + 2 * n_pairs * 4, band, n_pairs * 4);
        if (reverse) memcpy(stage.p + 3 * n_pairs * 4, reverse, n_pairs);
        if (stage.send(dblock, (3 * n_pairs + wr) * 4) != hipSuccess) rc = svt_fail(c, SVT_ERR_HIP, std::string(who) + ": descriptor upload failed");
    }
    for (int cls = 0; cls < 4 && rc == SVT_OK; cls++) {
        const int rclass_bp = cls + 1;                                                 // lane-per-pair kernel: 8 / 16 / 24 words of band rows (cls 3, bands 384-511: none)
        const int k9 = c->opt().k9_kernel;                                             // svt_set_option("k9_kernel") pins the kernel (tests, profiling)
        const bool wavefront = k9 == 1, force_bp = k9 >= 2, full_slab = k9 == 3;
        // bit-parallel K9 (one pair per lane) for bands up to 255 once there are enough pairs to fill the chip with waves: a lone wave
        // needs ~2.4 ms for a 1.5 kb pair, the block-per-pair anti-diagonal kernel ~1 ms, and the two cross at ~6k pairs
        const bool bp = cls < 3 && !wavefront && (force_bp || sel[cls].size() >= 6000);
        const int rclass = bp ? rclass_bp : (cls == 0 ? 1 : (cls == 1 ? 2 : 4));        // wave-per-pair kernel: 4 / 8 / 16 diagonals per lane
        const u64 stride = bp ? align_tb_dwords_bp(rclass, T->max_len, full_slab) : align_tb_dwords(rclass, Q->max_len, T->max_len);
        const u64 stride_full = bp ? align_tb_dwords_bp(rclass, T->max_len, true) : 0;
        u64 chunk = std::max<u64>(1, (u64)(6ull << 30) / (stride * 4));                // traceback slabs: <= 6 GiB per launch
        if (bp) chunk = std::max<u64>(64, chunk & ~(u64)63);                           // slabs are laid out per 64 pairs
        for (u64 lo = 0; lo < sel[cls].size() && rc == SVT_OK; lo += chunk) {
            const u64 ns = std::min<u64>(chunk, sel[cls].size() - lo), ns64 = (ns + 63) & ~(u64)63;
            const bool windowed = bp && !full_slab;
            const u64 redo_cap = windowed ? std::min<u64>(ns64, 4096) : 0;              // pairs per launch of the full-slab pass over drifted walks
            std::vector<u64> loff(ns);
            for (u64 i = 0; i < ns; i++) loff[i] = cell_off[sel[cls][lo + i]];         // absolute row starts
            Carve cv; size_t is = cv.add(ns * 4), io = cv.add(ns * 8), isp = cv.add(ns * 16), itb = cv.add(ns64 * stride * 4);
            size_t ik = cv.add(windowed ? ns * 8 : 0), ir = cv.add(windowed ? (ns + 1) * 4 : 0), ir2 = cv.add(windowed ? (ns + 1) * 4 : 0), itf = cv.add(redo_cap * stride_full * 4);
            if (!ensure_scratch(c, cv.total)) { rc = svt_fail(c, SVT_ERR_HIP, "scratch allocation failed"); break; }
            u32* dsel = carve_ptr<u32>(c, cv, is); u64* doff = carve_ptr<u64>(c, cv, io); u32* dspan = carve_ptr<u32>(c, cv, isp); u32* dtb = carve_ptr<u32>(c, cv, itb);
            UpPack up(c, cv); up.put(is, sel[cls].data() + lo, ns * 4); up.put(io, loff.data(), ns * 8);
            if (up.send() != hipSuccess) { rc = svt_fail(c, SVT_ERR_HIP, std::string(who) + ": upload failed"); break; }
            if (windowed) {
                // one 64-bit window of direction bits per column around the line (0,0)-(n,m); walks that leave it run again around their end
                // diagonal, and what still drifts once more with the full slab (kernels_align.hip, k_align_bp_tb)
                u64* dkeys = carve_ptr<u64>(c, cv, ik); u32* dredo = carve_ptr<u32>(c, cv, ir); u32* dredo2 = carve_ptr<u32>(c, cv, ir2); u32* dtbf = carve_ptr<u32>(c, cv, itf);
                auto count_of = [&](u32* d, u32& n) -> bool { return peek(c, d, &n, 4) == hipSuccess; };
                hipMemsetAsync(dredo, 0, 4, c->stream); hipMemsetAsync(dredo2, 0, 4, c->stream);
                // round 4: the first pass keeps 32 bits per column (option "k9_window"): half the slab traffic; the pass around the end diagonal keeps 64 (the slab is sized for that)
                rc = launch_align_tb_bp(c, Q, T, dq, dt, reverse ? dr : nullptr, db, dsel, ns, rclass, dn, T->max_len, dtb, d_cells, doff, dspan, 1, dkeys, dredo, nullptr, 0.0, c->opt().k9_window == 32 ? 32 : 64);
                if (rc != SVT_OK) break;
                u32 n_again = 0, n_full = 0;
                if (!count_of(dredo, n_again)) { rc = svt_fail(c, SVT_ERR_HIP, std::string(who) + ": copy back failed"); break; }
                if (n_again) {
                    rc = launch_align_tb_bp(c, Q, T, dq, dt, reverse ? dr : nullptr, db, dsel, n_again, rclass, dn, T->max_len, dtb, d_cells, doff, dspan, 2, dkeys, dredo2, dredo + 1);
                    if (rc != SVT_OK) break;
                    if (!count_of(dredo2, n_full)) { rc = svt_fail(c, SVT_ERR_HIP, std::string(who) + ": copy back failed"); break; }
                }
                c->k9_pairs += ns; c->k9_again_pairs += n_again; c->k9_redo_pairs += n_full;
                for (u64 r = 0; r < n_full && rc == SVT_OK; r += redo_cap)
                    rc = launch_align_tb_bp(c, Q, T, dq, dt, reverse ? dr : nullptr, db, dsel, std::min<u64>(redo_cap, n_full - r), rclass, dn, T->max_len, dtbf, d_cells, doff, dspan, 0, nullptr, nullptr, dredo2 + 1 + r);
            } else
            rc = bp ? launch_align_tb_bp(c, Q, T, dq, dt, reverse ? dr : nullptr, db, dsel, ns, rclass, dn, T->max_len, dtb, d_cells, doff, dspan, 0, nullptr, nullptr, nullptr)
                    : launch_align_tb(c, Q, T, dq, dt, reverse ? dr : nullptr, db, dsel, ns, rclass, dn, Q->max_len, T->max_len, dtb, d_cells, doff, dspan);
            if (rc != SVT_OK) break;
            std::vector<u32> hs(ns * 4);
            DownPack dsp(c); dsp.get(dspan, hs.data(), ns * 16);
            if (dsp.recv() != hipSuccess || ctx_sync(c) != hipSuccess) { rc = svt_fail(c, SVT_ERR_HIP, std::string(who) + ": copy back failed"); break; }
            dsp.scatter();
            if (span) for (u64 i = 0; i < ns; i++) memcpy(span + (u64)sel[cls][lo + i] * 4, hs.data() + i * 4, 16);
        }
    }
    if (rc == SVT_OK) { DownPack dnm(c); dnm.get(dn, nm, n_pairs * 4); if (dnm.recv() != hipSuccess || ctx_sync(c) != hipSuccess) rc = svt_fail(c, SVT_ERR_HIP, std::string(who) + ": nm copy failed"); else dnm.scatter(); }
    ctx_sync(c);
    dfree(dblock);
    return rc;
}

int svt_align_pileup(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx, const uint32_t* t_idx, const uint8_t* reverse,
                     const uint32_t* band, uint64_t n_pairs, const uint64_t* cell_off, uint64_t* cells, uint32_t* span, int32_t* nm) {
    if (!c || !Q || !T || (n_pairs && (!q_idx || !t_idx || !band || !nm || !cell_off || !cells || !span))) return svt_fail(c, SVT_ERR_ARG, "svt_align_pileup: null argument");
    if (n_pairs == 0) return SVT_OK;
    for (u64 i = 0; i < n_pairs; i++)
        if (q_idx[i] < Q->n && cell_off[i + 1] - cell_off[i] != Q->h_off[q_idx[i] + 1] - Q->h_off[q_idx[i]]) return svt_fail(c, SVT_ERR_ARG, "svt_align_pileup: cell_off must follow the query lengths");
    hipSetDevice(c->device);
    u64* dcells = nullptr;
    TRY(dmalloc(c, &dcells, cell_off[n_pairs]));
    int rc = pileup_run(c, Q, T, q_idx, t_idx, reverse, band, n_pairs, cell_off, dcells, span, nm, "svt_align_pileup");
    if (rc == SVT_OK && cell_off[n_pairs]) {                                     // through the pinned staging buffer and a polite wait (a plain hipMemcpy of ~20 MB into pageable memory spun for 2 % of a 2-CPU step)
        DownPack dc(c); dc.get(dcells, cells, cell_off[n_pairs] * 8);
        if (dc.recv() != hipSuccess || ctx_sync(c) != hipSuccess) rc = svt_fail(c, SVT_ERR_HIP, "svt_align_pileup: copy back failed"); else dc.scatter();
    }
    dfree(dcells);
    return rc;
}

// ---- K10: device-resident pile-ups + per-column statistics ------------------------------------------------------------
struct svt_pileup {
    u64 n_pairs = 0, n_cells = 0; u32 n_groups = 0; u64 n_cols = 0; u32 n_tiles = 0;
    const svt_batch* Q = nullptr;
    u64* d_cells = nullptr; u64* d_cell_off = nullptr; u32* d_pair_q = nullptr; u64* d_grp_off = nullptr; u64* d_col_off = nullptr; void* d_tiles = nullptr;
    std::vector<u64> h_cell_off; std::vector<u32> h_pair_q; u64 n_rows_cells = 0;   // row starts in HBM (64-byte aligned), the query of every pair, cells without the padding
};
void svt_pileup_free(svt_ctx*, svt_pileup* p) {
    if (!p) return;
    dfree(p->d_cells); dfree(p->d_cell_off);                       // grp_off, col_off, tiles and pair_q live in the cell_off block
    delete p;
}
int svt_pileup_create(svt_ctx* c, const svt_batch* Q, const svt_batch* T, const uint32_t* q_idx, const uint32_t* t_idx, const uint8_t* reverse, const uint32_t* band,
                      uint64_t n_pairs, const uint64_t* grp_off, uint32_t n_groups, svt_pileup** out, uint32_t* span, int32_t* nm) {
    if (out) *out = nullptr;
    if (!c || !Q || !T || !out || !grp_off || (n_pairs && (!q_idx || !t_idx || !band || !nm))) return svt_fail(c, SVT_ERR_ARG, "svt_pileup_create: null argument");
    if (grp_off[0] != 0 || grp_off[n_groups] != n_pairs) return svt_fail(c, SVT_ERR_ARG, "svt_pileup_create: grp_off must cover [0, n_pairs)");
    hipSetDevice(c->device);
    struct Tile { u32 group, col0; };
    std::vector<u64> cell_off(n_pairs + 1, 0), col_off(n_groups + 1, 0); std::vector<Tile> tiles;
    for (u32 g = 0; g < n_groups; g++) {
        if (grp_off[g + 1] < grp_off[g]) return svt_fail(c, SVT_ERR_ARG, "svt_pileup_create: grp_off not monotone");
        u64 len = 0;
        for (u64 i = grp_off[g]; i < grp_off[g + 1]; i++) {
            if (q_idx[i] >= Q->n) return svt_fail(c, SVT_ERR_ARG, "svt_pileup_create: index out of range");
            if (q_idx[i] != q_idx[grp_off[g]]) return svt_fail(c, SVT_ERR_ARG, "svt_pileup_create: the pairs of one group must share the query");
            len = Q->h_off[q_idx[i] + 1] - Q->h_off[q_idx[i]];
            cell_off[i + 1] = cell_off[i] + ((len + 7) & ~(u64)7);                      // rows start on 64-byte lines: K9 stores a row in 64-byte runs (svt_pileup_fetch hands these offsets out)
        }
        col_off[g + 1] = col_off[g] + len;                                           // an empty group has no columns
        for (u64 c0 = 0; c0 < len; c0 += 256) tiles.push_back(Tile{g, (u32)c0});
    }
    svt_pileup* p = new svt_pileup();
    p->n_pairs = n_pairs; p->n_cells = cell_off[n_pairs]; p->n_groups = n_groups; p->n_cols = col_off[n_groups]; p->n_tiles = (u32)tiles.size(); p->Q = Q;
    p->h_cell_off = cell_off; p->h_pair_q.assign(q_idx, q_idx + n_pairs);
    for (u64 i = 0; i < n_pairs; i++) p->n_rows_cells += Q->h_off[q_idx[i] + 1] - Q->h_off[q_idx[i]];
    int rc = SVT_OK;
    // descriptors {cell_off | grp_off | col_off | tiles | pair_q} in one block: one copy
    const size_t w_co = n_pairs + 1, w_go = (size_t)n_groups + 1, w_ti = tiles.size();
    const size_t n64 = w_co + 2 * w_go + w_ti + (n_pairs + 1) / 2;
    if ((rc = dmalloc(c, &p->d_cells, p->n_cells)) != SVT_OK || (rc = dmalloc(c, &p->d_cell_off, n64)) != SVT_OK) { svt_pileup_free(c, p); return rc; }
    p->d_grp_off = p->d_cell_off + w_co; p->d_col_off = p->d_grp_off + w_go; p->d_tiles = (void*)(p->d_col_off + w_go); p->d_pair_q = (u32*)(p->d_col_off + w_go + w_ti);
    if (n_pairs) rc = pileup_run(c, Q, T, q_idx, t_idx, reverse, band, n_pairs, cell_off.data(), p->d_cells, span, nm, "svt_pileup_create");
    if (rc == SVT_OK) {
        StageUp stg(c, n64 * 8); u64* stage = (u64*)stg.p;
        memcpy(stage, cell_off.data(), w_co * 8); memcpy(stage + w_co, grp_off, w_go * 8); memcpy(stage + w_co + w_go, col_off.data(), w_go * 8);
        if (w_ti) memcpy(stage + w_co + 2 * w_go, tiles.data(), w_ti * sizeof(Tile));
        if (n_pairs) memcpy(stage + w_co + 2 * w_go + w_ti, q_idx, n_pairs * 4);
        if (stg.send(p->d_cell_off, n64 * 8) != hipSuccess || ctx_sync(c) != hipSuccess) rc = svt_fail(c, SVT_ERR_HIP, "svt_pileup_create: descriptor upload failed");
    }
    if (rc != SVT_OK) { svt_pileup_free(c, p); return rc; }
    *out = p;
    return SVT_OK;
}
uint64_t svt_pileup_cells(const svt_pileup* p) { return p ? p->n_rows_cells : 0; }
uint64_t svt_pileup_columns(const svt_pileup* p) { return p ? p->n_cols : 0; }
int svt_pileup_fetch(svt_ctx* c, const svt_pileup* p, uint64_t* cells, uint64_t* cell_off) {
    if (!c || !p) return svt_fail(c, SVT_ERR_ARG, "svt_pileup_fetch: null argument");
    hipSetDevice(c->device);
    // the rows sit on 64-byte lines in HBM (svt_pileup_create); the hook hands them out back to back, as svt_align_pileup lays them out
    std::vector<u64> tight(p->n_pairs + 1, 0);
    for (u64 i = 0; i < p->n_pairs; i++) tight[i + 1] = tight[i] + p->Q->h_off[p->h_pair_q[i] + 1] - p->Q->h_off[p->h_pair_q[i]];
    if (cell_off) memcpy(cell_off, tight.data(), (p->n_pairs + 1) * 8);
    if (cells && p->n_cells) {
        std::vector<u64> padded(p->n_cells);
        HIPCHK(c, hipMemcpy(padded.data(), p->d_cells, p->n_cells * 8, hipMemcpyDeviceToHost));
        for (u64 i = 0; i < p->n_pairs; i++) memcpy(cells + tight[i], padded.data() + p->h_cell_off[i], (tight[i + 1] - tight[i]) * 8);
    }
    return SVT_OK;
}
int svt_pileup_stats(svt_ctx* c, const svt_pileup* p, const uint8_t* grp_selected, uint32_t* depth, uint32_t* err, uint64_t* qual_total, uint64_t* qual_err) {
    if (!c || !p || !depth || !err || !qual_total || !qual_err) return svt_fail(c, SVT_ERR_ARG, "svt_pileup_stats: null argument");
    hipSetDevice(c->device);
    memset(qual_total, 0, 256 * 8); memset(qual_err, 0, 256 * 8);
    if (p->n_cols == 0) return SVT_OK;
    Carve cv; size_t id = cv.add(p->n_cols * 4), ie = cv.add(p->n_cols * 4), it = cv.add(256 * 8), ir = cv.add(256 * 8), is = cv.add((size_t)p->n_groups + 1);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u32* dd = carve_ptr<u32>(c, cv, id); u32* de = carve_ptr<u32>(c, cv, ie); ull* dt = carve_ptr<ull>(c, cv, it); ull* dr = carve_ptr<ull>(c, cv, ir); u8* ds = carve_ptr<u8>(c, cv, is);
    HIPCHK(c, hipMemsetAsync(dt, 0, 256 * 8, c->stream)); HIPCHK(c, hipMemsetAsync(dr, 0, 256 * 8, c->stream));
    if (grp_selected) HIPCHK(c, hipMemcpyAsync(ds, grp_selected, p->n_groups, hipMemcpyHostToDevice, c->stream));
    TRY(launch_pileup_stats(c, p->Q, p->d_cells, p->d_cell_off, p->d_pair_q, p->d_grp_off, p->d_col_off, grp_selected ? ds : nullptr, p->d_tiles, p->n_tiles, p->n_cells, dd, de, dt, dr));
    HIPCHK(c, memcpy_d2h(c, depth, dd, p->n_cols * 4));
    HIPCHK(c, memcpy_d2h(c, err, de, p->n_cols * 4));
    HIPCHK(c, memcpy_d2h(c, qual_total, dt, 256 * 8));
    HIPCHK(c, memcpy_d2h(c, qual_err, dr, 256 * 8));
    HIPCHK(c, ctx_sync(c));
    return SVT_OK;
}
int svt_pileup_hp_median(svt_ctx* c, const svt_pileup* p, uint8_t* median) {
    if (!c || !p || !median) return svt_fail(c, SVT_ERR_ARG, "svt_pileup_hp_median: null argument");
    hipSetDevice(c->device);
    if (p->n_cols == 0) return SVT_OK;
    Carve cv; size_t im = cv.add(p->n_cols);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    u8* dm = carve_ptr<u8>(c, cv, im);
    TRY(launch_pileup_hp_median(c, p->d_cells, p->d_cell_off, p->d_grp_off, p->d_col_off, p->d_tiles, p->n_tiles, p->n_cells, dm));
    HIPCHK(c, memcpy_d2h(c, median, dm, p->n_cols));
    HIPCHK(c, ctx_sync(c));
    return SVT_OK;
}
int svt_pileup_loglik(svt_ctx* c, const svt_pileup* p, const double* ln_table, double ln_indel_err, double ln_indel_acc, double* lr, double* ln) {
    if (!c || !p || !ln_table || !lr || !ln) return svt_fail(c, SVT_ERR_ARG, "svt_pileup_loglik: null argument");
    hipSetDevice(c->device);
    if (p->n_cols == 0) return SVT_OK;
    Carve cv; size_t il = cv.add(p->n_cols * 8), in = cv.add(p->n_cols * 8), it = cv.add(512 * 8);
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    double* dl = carve_ptr<double>(c, cv, il); double* dn = carve_ptr<double>(c, cv, in); double* dt = carve_ptr<double>(c, cv, it);
    HIPCHK(c, hipMemcpyAsync(dt, ln_table, 512 * 8, hipMemcpyHostToDevice, c->stream));
    TRY(launch_pileup_loglik(c, p->Q, p->d_cells, p->d_cell_off, p->d_pair_q, p->d_grp_off, p->d_col_off, p->d_tiles, p->n_tiles, p->n_cells, dt, ln_indel_err, ln_indel_acc, dl, dn));
    HIPCHK(c, memcpy_d2h(c, lr, dl, p->n_cols * 8));
    HIPCHK(c, memcpy_d2h(c, ln, dn, p->n_cols * 8));
    HIPCHK(c, ctx_sync(c));
    return SVT_OK;
}

// ---- K12: Stage-4a POA with device-resident graphs ---------------------------------------------------------------------
struct PoaGJobHost { u64 arena, seq_first; u32 n_seqs, ncap, ecap, lmax; };
// sequences and weights either from the caller's arrays (seq, weights) or gathered on the device from a resident batch (B, read_idx, reverse); seq_off is the host's either way
static int poa_submit_impl(svt_ctx* c, uint32_t n_clusters, const uint64_t* cl_off, const uint64_t* seq_off, const uint8_t* seq, const uint8_t* weights,
                           const svt_batch* B, const uint32_t* read_idx, const uint8_t* reverse, const uint32_t* seq_band) {
    c->poa_last.valid = false; c->poa_last.pending = false;
    if (n_clusters == 0) { c->poa_last.pending = true; c->poa_last.n_clusters = 0; return SVT_OK; }
    if (sizeof(PoaGJobHost) != poa_graph_job_bytes() || sizeof(svt_poa_result) != poa_graph_out_bytes()) return svt_fail(c, SVT_ERR_STATE, "svt_poa_graphs: record layouts differ");
    hipSetDevice(c->device);
    const u64 n_seqs = cl_off[n_clusters];
    // one kernel width for the launch: the widest band decides (a band of 2 bw + 1 columns must touch at most 7 of the 8 chunks of 64 C columns)
    u32 max_bw = 0, lmax_all = 0; double cells = 0;
    for (u64 s = 0; s < n_seqs; s++) {
        const u64 len = seq_off[s + 1] - seq_off[s];
        if (len > 5440) return svt_fail(c, SVT_ERR_ARG, "svt_poa_graphs: sequence longer than 5440 bases (16-bit cells)");
        if (len) { max_bw = std::max(max_bw, seq_band[s]); lmax_all = std::max<u32>(lmax_all, (u32)len); }
        cells += (double)len * (2.0 * seq_band[s] + 1.0);
    }
    int C = max_bw <= (u32)poa_graph_max_band(1) ? 1 : (max_bw <= (u32)poa_graph_max_band(2) ? 2 : 4);
    if (max_bw > (u32)poa_graph_max_band(4)) return svt_fail(c, SVT_ERR_ARG, "svt_poa_graphs: band half-width above 640 columns");
    // the row engine (one wave per cluster, a graph row per step; kernels_poa_graph.hip ENG = 1) takes the launch when every band fits its 384 / 512 columns
    // and every base is one of ACGT (its match masks are indexed by two bits of the letter); option "poa_rows" 0 keeps the chunk pipeline
    if (c->opt().poa_rows == 2) C += 200;                                          // the anti-diagonal engine (ENG = 2): lane = graph row, same band classes and back-pointer rows as the chunk pipeline
    else if (c->opt().poa_rows && max_bw <= (u32)poa_graph_max_band(108)) {
        bool acgt = true;
        if (seq) { const u64 nb = seq_off[n_seqs]; for (u64 x = 0; x < nb && acgt; x++) { const u8 b = seq[x]; acgt = b == 'A' || b == 'C' || b == 'G' || b == 'T'; } }   // gathered sequences are decoded 2-bit codes: always ACGT
        if (acgt) C = max_bw <= (u32)poa_graph_max_band(106) ? 106 : 108;
    }
    std::vector<PoaGJobHost> jobs(n_clusters);
    u64 arena = 0, sum_ncap = 0, sum_ecap = 0;
    for (u32 j = 0; j < n_clusters; j++) {
        u32 lmax = 1;
        for (u64 s = cl_off[j]; s < cl_off[j + 1]; s++) lmax = std::max<u32>(lmax, (u32)(seq_off[s + 1] - seq_off[s]));
        const u32 ncap = std::min<u32>(65535u, 4u * lmax + 2048u), ecap = 2u * ncap;
        jobs[j] = PoaGJobHost{arena, cl_off[j], (u32)(cl_off[j + 1] - cl_off[j]), ncap, ecap, lmax};
        arena += (poa_graph_arena_bytes(ncap, ecap, lmax, C) + 255) & ~(u64)255;
        sum_ncap += ncap; sum_ecap += ecap;
    }
    const u64 n_bytes = seq_off[n_seqs];
    Carve cv;
    const size_t ij = cv.add(n_clusters * sizeof(PoaGJobHost)), io = cv.add(n_clusters * sizeof(svt_poa_result)), iso = cv.add((n_seqs + 1) * 8), ib = cv.add(n_seqs * 4 + 4),
                 iri = cv.add(n_seqs * 4 + 4), irv = cv.add(n_seqs + 16),
                 is = cv.add(n_bytes + 16), iw = cv.add(n_bytes + 16), ino = cv.add((n_clusters + 1) * 8), ieo = cv.add((n_clusters + 1) * 8), ia = cv.add(arena + 256),
                 ic = cv.add(sum_ncap + 16), il = cv.add(sum_ncap * 16 + 16), ie = cv.add(sum_ecap * 12 + 16),   // the compacted graphs, sized by the capacities
                 icn = cv.add(sum_ncap + 16);                                                                          // the consensuses (K12c): cluster j at the sum of the node capacities before it
    if (!ensure_scratch(c, cv.total)) return svt_fail(c, SVT_ERR_HIP, "scratch allocation failed");
    void* dj = carve_ptr<char>(c, cv, ij); void* dout = carve_ptr<char>(c, cv, io); u64* dso = carve_ptr<u64>(c, cv, iso); u32* db = carve_ptr<u32>(c, cv, ib);
    u32* dri = carve_ptr<u32>(c, cv, iri); u8* drv = carve_ptr<u8>(c, cv, irv);
    u8* ds = carve_ptr<u8>(c, cv, is); u8* dw = carve_ptr<u8>(c, cv, iw); u8* da = carve_ptr<u8>(c, cv, ia);
    HIPCHK(c, hipMemcpyAsync(dj, jobs.data(), n_clusters * sizeof(PoaGJobHost), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dso, seq_off, (n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(db, seq_band, n_seqs * 4, hipMemcpyHostToDevice, c->stream));
    if (seq) {
        HIPCHK(c, hipMemcpyAsync(ds, seq, n_bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(dw, weights, n_bytes, hipMemcpyHostToDevice, c->stream));
    } else {
        HIPCHK(c, hipMemcpyAsync(dri, read_idx, n_seqs * 4, hipMemcpyHostToDevice, c->stream));
        if (reverse) HIPCHK(c, hipMemcpyAsync(drv, reverse, n_seqs, hipMemcpyHostToDevice, c->stream));
        TRY(launch_poa_gather(c, B, dri, reverse ? drv : nullptr, dso, (u32)n_seqs, ds, dw, (double)n_bytes));
    }
    TRY(launch_poa_graph(c, C, n_clusters, lmax_all, dj, da, ds, dw, dso, db, dout, cells));
    TRY(launch_poa_consensus(c, C, n_clusters, dj, da, dout, carve_ptr<u8>(c, cv, icn)));      // behind the graphs on the same stream: svt_poa_graphs_wait finds the lengths in the results
    auto& P = c->poa_last;
    P.off_cons = cv.offs[icn]; P.cons_slot.assign(n_clusters + 1, 0);
    for (u32 j = 0; j < n_clusters; j++) P.cons_slot[j + 1] = P.cons_slot[j] + jobs[j].ncap;
    P.pending = true; P.n_clusters = n_clusters; P.C = C; P.off_jobs = cv.offs[ij]; P.off_outs = cv.offs[io]; P.off_noff = cv.offs[ino]; P.off_eoff = cv.offs[ieo]; P.off_arena = cv.offs[ia];
    P.off_code = cv.offs[ic]; P.off_al = cv.offs[il]; P.off_edge = cv.offs[ie];
    return SVT_OK;
}
int svt_poa_graphs_submit(svt_ctx* c, uint32_t n_clusters, const uint64_t* cl_off, const uint64_t* seq_off, const uint8_t* seq, const uint8_t* weights, const uint32_t* seq_band) {
    if (!c || (n_clusters && (!cl_off || !seq_off || !seq || !weights || !seq_band))) return svt_fail(c, SVT_ERR_ARG, "svt_poa_graphs_submit: null argument");
    return poa_submit_impl(c, n_clusters, cl_off, seq_off, seq, weights, nullptr, nullptr, nullptr, seq_band);
}
int svt_poa_graphs_submit_reads(svt_ctx* c, const svt_batch* b, uint32_t n_clusters, const uint64_t* cl_off, const uint32_t* read_idx, const uint8_t* reverse, const uint32_t* seq_band) {
    if (!c || !b || (n_clusters && (!cl_off || !read_idx || !seq_band))) return svt_fail(c, SVT_ERR_ARG, "svt_poa_graphs_submit_reads: null argument");
    const u64 n_seqs = n_clusters ? cl_off[n_clusters] : 0;
    std::vector<u64> seq_off(n_seqs + 1, 0);
    for (u64 s = 0; s < n_seqs; s++) {
        if (read_idx[s] >= b->n) return svt_fail(c, SVT_ERR_ARG, "svt_poa_graphs_submit_reads: read index out of range");
        seq_off[s + 1] = seq_off[s] + (b->h_off[read_idx[s] + 1] - b->h_off[read_idx[s]]);
    }
    return poa_submit_impl(c, n_clusters, cl_off, seq_off.data(), nullptr, nullptr, b, read_idx, reverse, seq_band);
}
int svt_poa_graphs_wait(svt_ctx* c, svt_poa_result* res, uint64_t* node_off, uint64_t* edge_off) {
    if (!c || !c->poa_last.pending) return svt_fail(c, SVT_ERR_STATE, "svt_poa_graphs_wait: no svt_poa_graphs_submit on this context");
    auto& P = c->poa_last;
    P.pending = false;
    const u32 n_clusters = P.n_clusters;
    if (n_clusters == 0) return SVT_OK;
    if (!res || !node_off || !edge_off) return svt_fail(c, SVT_ERR_ARG, "svt_poa_graphs_wait: null argument");
    hipSetDevice(c->device);
    char* base = (char*)c->scratch;
    // wait first, copy afterwards: a copy into pageable memory queued behind the launch makes the runtime spin on a core for as long as the kernel runs (0.2 CPU-s per step)
    HIPCHK(c, ctx_sync_long(c));
    HIPCHK(c, memcpy_d2h(c, res, base + P.off_outs, n_clusters * sizeof(svt_poa_result)));
    HIPCHK(c, ctx_sync(c));
    node_off[0] = 0; edge_off[0] = 0;
    { double rows = 0; for (u32 j = 0; j < n_clusters; j++) rows = std::max(rows, (double)res[j].rows_done); prof_add_units(c, P.C >= 200 ? "k_poa_diag" : (P.C >= 100 ? "k_poa_rows" : "k_poa_graph"), rows); }   // profile units of K12 = graph rows of the launch's LONGEST chain (a cluster's rows are one dependent chain; the clusters run side by side)
    c->poa_clusters += n_clusters;
    for (u32 j = 0; j < n_clusters; j++) {
        const bool ok = res[j].status == 0;
        if (!ok) c->poa_handed_back++; else if (res[j].cons_len != 0xFFFFFFFFu) c->poa_cons_device++;
        node_off[j + 1] = node_off[j] + (ok ? res[j].n_nodes : 0); edge_off[j + 1] = edge_off[j] + (ok ? res[j].n_edges : 0);
    }
    HIPCHK(c, hipMemcpyAsync(base + P.off_noff, node_off, (n_clusters + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(base + P.off_eoff, edge_off, (n_clusters + 1) * 8, hipMemcpyHostToDevice, c->stream));
    TRY(launch_poa_graph_export(c, P.C, n_clusters, base + P.off_jobs, (const u8*)(base + P.off_arena), base + P.off_outs, (const u64*)(base + P.off_noff), (const u64*)(base + P.off_eoff),
                                (u8*)(base + P.off_code), (u16*)(base + P.off_al), (u32*)(base + P.off_edge)));
    P.valid = true; P.n_nodes = node_off[n_clusters]; P.n_edges = edge_off[n_clusters];
    return SVT_OK;
}
int svt_poa_graphs(svt_ctx* c, uint32_t n_clusters, const uint64_t* cl_off, const uint64_t* seq_off, const uint8_t* seq, const uint8_t* weights,
                   const uint32_t* seq_band, svt_poa_result* res, uint64_t* node_off, uint64_t* edge_off) {
    if (!c || (n_clusters && (!res || !node_off || !edge_off))) return svt_fail(c, SVT_ERR_ARG, "svt_poa_graphs: null argument");
    TRY(svt_poa_graphs_submit(c, n_clusters, cl_off, seq_off, seq, weights, seq_band));
    return svt_poa_graphs_wait(c, res, node_off, edge_off);
}
int svt_poa_consensus_fetch(svt_ctx* c, const svt_poa_result* res, uint64_t* cons_off, uint8_t* cons) {
    if (!c || !c->poa_last.valid) return svt_fail(c, SVT_ERR_STATE, "svt_poa_consensus_fetch: no svt_poa_graphs result on this context");
    const auto& p = c->poa_last;
    if (p.n_clusters && (!res || !cons_off)) return svt_fail(c, SVT_ERR_ARG, "svt_poa_consensus_fetch: null argument");
    hipSetDevice(c->device);
    cons_off[0] = 0;
    for (u32 j = 0; j < p.n_clusters; j++) {
        const u64 len = (res[j].status == 0 && res[j].cons_len != 0xFFFFFFFFu) ? res[j].cons_len : 0;
        if (len > p.cons_slot[j + 1] - p.cons_slot[j]) return svt_fail(c, SVT_ERR_ARG, "svt_poa_consensus_fetch: cons_len of a cluster exceeds what the device wrote for it (results of another run?)");
        cons_off[j + 1] = cons_off[j] + len;
    }
    if (cons_off[p.n_clusters] == 0) return SVT_OK;
    if (!cons) return svt_fail(c, SVT_ERR_ARG, "svt_poa_consensus_fetch: null argument");
    const u64 span = p.cons_slot[p.n_clusters];
    DownPack dn(c); std::vector<u8> all(span);
    dn.get((char*)c->scratch + p.off_cons, all.data(), span);
    HIPCHK(c, dn.recv());
    HIPCHK(c, ctx_sync(c));
    dn.scatter();
    for (u32 j = 0; j < p.n_clusters; j++) if (cons_off[j + 1] > cons_off[j]) memcpy(cons + cons_off[j], all.data() + p.cons_slot[j], cons_off[j + 1] - cons_off[j]);
    return SVT_OK;
}
int svt_poa_graphs_fetch(svt_ctx* c, uint8_t* code, uint16_t* aligned, uint32_t* edges) {
    if (!c || !c->poa_last.valid) return svt_fail(c, SVT_ERR_STATE, "svt_poa_graphs_fetch: no svt_poa_graphs result on this context");
    hipSetDevice(c->device);
    const auto& p = c->poa_last;
    if (!code && !aligned && !edges) { c->poa_last.valid = false; return SVT_OK; }   // the caller took the consensuses (svt_poa_consensus_fetch) and does not need the graphs: the result ends here
    if (p.n_nodes && (!code || !aligned)) return svt_fail(c, SVT_ERR_ARG, "svt_poa_graphs_fetch: null argument");
    if (p.n_edges && !edges) return svt_fail(c, SVT_ERR_ARG, "svt_poa_graphs_fetch: null argument");
    if (p.n_nodes) {
        HIPCHK(c, memcpy_d2h(c, code, (char*)c->scratch + p.off_code, p.n_nodes));
        HIPCHK(c, memcpy_d2h(c, aligned, (char*)c->scratch + p.off_al, p.n_nodes * 16));
    }
    if (p.n_edges) HIPCHK(c, memcpy_d2h(c, edges, (char*)c->scratch + p.off_edge, p.n_edges * 12));
    HIPCHK(c, ctx_sync(c));
    c->poa_last.valid = false;
    return SVT_OK;
}

}  // extern "C"
