// sampler.hpp -- a poor man's CPU profiler for the host library (development tool, SAVONT_SAMPLE=<file>): every thread that enters the library
// arms a timer on ITS OWN CPU clock (1 ms of thread CPU -> SIGPROF to that thread), the handler records the interrupted program counter, and
// at exit the counts per (module, offset) are written to the file -- symbolise with `llvm-symbolizer --obj=libsavont_asv.so 0x<offset>` (tools/symbolize_samples.py).  No effect unless the variable is set.
#pragma once
#include <atomic>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <ctime>
#include <ucontext.h>
#include <unistd.h>

namespace savont {
namespace sampler {
constexpr size_t CAP = 1 << 20;
inline std::atomic<size_t> g_n{0};
inline void** g_pc = nullptr;
inline const char* g_path = nullptr;
inline void on_prof(int, siginfo_t*, void* uc) {
#if defined(__x86_64__)
    const size_t i = g_n.fetch_add(1, std::memory_order_relaxed);
    if (i < CAP) g_pc[i] = (void*)((ucontext_t*)uc)->uc_mcontext.gregs[REG_RIP];
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
        if (dladdr(g_pc[i], &di) && di.dli_fname) fprintf(f, "%s\t0x%zx\t%s\n", di.dli_fname, (size_t)((char*)g_pc[i] - (char*)di.dli_fbase), di.dli_sname ? di.dli_sname : "?");
        else fprintf(f, "?\t%p\t?\n", g_pc[i]);
    }
    fclose(f);
}
inline void arm_thread() {                       // idempotent per thread; a no-op unless the sampler is on
    static thread_local bool armed = false;
    if (armed || !g_path) return;
    armed = true;
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
    g_pc = (void**)calloc(CAP, sizeof(void*));
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, nullptr);
    atexit(dump);
}
}  // namespace sampler
}  // namespace savont
