// sampler.hpp -- a poor man's CPU profiler for the host library (development tool, SAVONT_SAMPLE=<file>): every thread that enters the library
// arms a timer on ITS OWN CPU clock (1 ms of thread CPU -> SIGPROF to that thread), the handler records the interrupted program counter, and
// at exit the counts per (module, offset) are written to the file -- symbolise with `llvm-symbolizer --obj=libsavont_asv.so 0x<offset>` (tools/symbolize_samples.py).  No effect unless the variable is set.
#pragma once
#include <atomic>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <dlfcn.h>
#include <link.h>
#include <pthread.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <ctime>
#include <ucontext.h>
#include <unistd.h>

namespace savont {
namespace sampler {
constexpr size_t CAP = 1 << 18, STK = 4, SCAN = 16384;   // samples kept; callers kept per sample; stack words looked at per sample
inline std::atomic<size_t> g_n{0};
inline void** g_pc = nullptr;
inline void** g_stk = nullptr;
inline const char* g_path = nullptr;
inline size_t g_text[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};   // executable ranges of libsavont_*.so (start_once)
inline thread_local size_t t_stack_top = 0;
inline void on_prof(int, siginfo_t*, void* uc) {
#if defined(__x86_64__)
    const size_t i = g_n.fetch_add(1, std::memory_order_relaxed);
    if (i < CAP) {
        g_pc[i] = (void*)((ucontext_t*)uc)->uc_mcontext.gregs[REG_RIP];
        // the words of the stack that point into the library's code, innermost first: the callers of a sample taken in libc / the HIP runtime (a heuristic -- a stale
        // return address of an earlier call can sit there -- good enough to tell which stage an allocation, a copy or a runtime call belongs to)
        void* const* sp = (void* const*)((ucontext_t*)uc)->uc_mcontext.gregs[REG_RSP];
        const size_t top = t_stack_top;                                       // end of this thread's stack (arm_thread): nothing is read beyond it
        size_t room = top > (size_t)sp ? (top - (size_t)sp) / sizeof(void*) : 0;
        if (room > SCAN) room = SCAN;
        size_t found = 0;
        for (size_t k = 0; k < room && found < STK; k++) {
            const size_t w = (size_t)sp[k];
            for (int m = 0; m < 4; m++) if (w >= g_text[m][0] && w < g_text[m][1]) { g_stk[i * STK + found++] = (void*)w; break; }
        }
    }
#endif
}
inline void dump() {
    if (!g_path) return;
    signal(SIGPROF, SIG_IGN);
    FILE* f = fopen(g_path, "w");
    if (!f) return;
    const size_t n = g_n.load() < CAP ? g_n.load() : CAP;
    for (size_t i = 0; i < n; i++) {
        Dl_info di; memset(&di, 0, sizeof di);
        if (dladdr(g_pc[i], &di) && di.dli_fname) {
            const char* b = strrchr(di.dli_fname, '/'); b = b ? b + 1 : di.dli_fname;
            fprintf(f, "%s\t0x%zx\t%s", b, (size_t)((char*)g_pc[i] - (char*)di.dli_fbase), di.dli_sname ? di.dli_sname : "?");
        } else fprintf(f, "?\t%p\t?", g_pc[i]);
        // the words of the stack that point into this library, innermost first: the callers (a heuristic -- a stale return address of an earlier call can sit
        // there -- good enough to tell which stage an allocation, a copy or a runtime call belongs to)
        for (size_t k = 0; k < STK; k++) {
            Dl_info dc; memset(&dc, 0, sizeof dc);
            void* w = g_stk[i * STK + k];
            if (!w || !dladdr(w, &dc) || !dc.dli_fname) continue;
            const char* b = strrchr(dc.dli_fname, '/'); b = b ? b + 1 : dc.dli_fname;
            fprintf(f, "\t%s+0x%zx", b, (size_t)((char*)w - 1 - (char*)dc.dli_fbase));
        }
        fputc('\n', f);
    }
    fclose(f);
}
inline void arm_thread() {                       // idempotent per thread; a no-op unless the sampler is on
    static thread_local bool armed = false;
    if (armed || !g_path) return;
    armed = true;
    { pthread_attr_t at; void* lo = nullptr; size_t sz = 0;
      if (pthread_getattr_np(pthread_self(), &at) == 0) { if (pthread_attr_getstack(&at, &lo, &sz) == 0) t_stack_top = (size_t)lo + sz; pthread_attr_destroy(&at); } }
    struct sigevent sev; memset(&sev, 0, sizeof sev);
    sev.sigev_notify = SIGEV_THREAD_ID; sev.sigev_signo = SIGPROF;
#ifndef sigev_notify_thread_id
#define sigev_notify_thread_id _sigev_un._tid
#endif
    sev.sigev_notify_thread_id = (pid_t)syscall(SYS_gettid);
    timer_t t;
    if (timer_create(CLOCK_THREAD_CPUTIME_ID, &sev, &t) != 0) return;
    struct itimerspec its; its.it_interval.tv_sec = 0; its.it_interval.tv_nsec = 997000; its.it_value = its.it_interval;
    timer_settime(t, 0, &its, nullptr);
}
inline void start_once() {
    static bool started = false;
    if (started) return;
    started = true;
    g_path = getenv("SAVONT_SAMPLE");
    if (!g_path || !*g_path) { g_path = nullptr; return; }
    g_pc = (void**)calloc(CAP, sizeof(void*)); g_stk = (void**)calloc(CAP * STK, sizeof(void*));
    dl_iterate_phdr(+[](struct dl_phdr_info* info, size_t, void*) -> int {
        if (!info->dlpi_name || !strstr(info->dlpi_name, "libsavont")) return 0;
        for (int m = 0; m < 4; m++) if (!g_text[m][1]) {
            size_t lo = (size_t)-1, hi = 0;
            for (int h = 0; h < info->dlpi_phnum; h++) { const auto& ph = info->dlpi_phdr[h]; if (ph.p_type == PT_LOAD && (ph.p_flags & PF_X)) { lo = std::min(lo, (size_t)(info->dlpi_addr + ph.p_vaddr)); hi = std::max(hi, (size_t)(info->dlpi_addr + ph.p_vaddr + ph.p_memsz)); } }
            if (hi) { g_text[m][0] = lo; g_text[m][1] = hi; }
            break;
        }
        return 0;
    }, nullptr);
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, nullptr);
    atexit(dump);
}
}  // namespace sampler
}  // namespace savont
