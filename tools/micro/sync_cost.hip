// build: /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/sync_cost.hip -o tools/micro/_sync_cost   (run on a GPU box: gpurun -- tools/micro/_sync_cost)
// CPU time the threads of a process burn while the host waits for a kernel, by the way it waits: hipStreamSynchronize, a hipEventBlockingSync event,
// hipStreamQuery between sleeps, a plain sleep followed by hipStreamSynchronize, and a flag in pinned host memory that a one-lane kernel behind the
// work sets (no runtime call while waiting).  Per mode: wall, process CPU, and the CPU of every thread that used any (tid: user + system ms).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <chrono>
#include <dirent.h>
#include <map>
#include <string>
#include <unistd.h>
__global__ void spin(long long cycles, int* out) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} if (out) *out = 1; }
// the same wait with 256 B of private (scratch) memory per lane: a dynamically indexed local array
__global__ void spin_scratch(long long cycles, int* out) {
    volatile int a[64];
    for (int i = 0; i < 64; i++) a[i] = i * threadIdx.x;
    const long long t0 = wall_clock64(); int x = 0;
    while (wall_clock64() - t0 < cycles) { x = a[(x + threadIdx.x) & 63] + 1; }
    if (out) *out = x;
}
__global__ void set_flag(volatile unsigned* flag, unsigned v) { *flag = v; __threadfence_system(); }
static double cpu() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::map<int, std::pair<double, double>> threads() {
    std::map<int, std::pair<double, double>> out; const double tick = (double)sysconf(_SC_CLK_TCK);
    DIR* d = opendir("/proc/self/task"); if (!d) return out;
    while (dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        char p[64]; snprintf(p, sizeof p, "/proc/self/task/%s/stat", e->d_name);
        FILE* f = fopen(p, "r"); if (!f) continue;
        char buf[1024]; const size_t n = fread(buf, 1, sizeof buf - 1, f); buf[n] = 0; fclose(f);
        const char* r = strrchr(buf, ')'); if (!r) continue;
        unsigned long ut = 0, st = 0; char state;
        sscanf(r + 2, "%c %*d %*d %*d %*d %*d %*u %*u %*u %*u %*u %lu %lu", &state, &ut, &st);
        out[atoi(e->d_name)] = {ut / tick * 1e3, st / tick * 1e3};
    }
    closedir(d); return out;
}
int main(int argc, char** argv) {
    const int flags = argc > 1 ? atoi(argv[1]) : 0;
    if (flags) printf("hipSetDeviceFlags(BlockingSync) -> %d\n", (int)hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming);
    int* d; hipMalloc(&d, 4);
    unsigned* flag; hipHostMalloc(&flag, 64, hipHostMallocMapped | hipHostMallocCoherent); *flag = 0;
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000LL, d); hipStreamSynchronize(s);
    const char* names[] = {"hipStreamSynchronize", "blocking event", "query + 40 us sleeps", "sleep 25 ms, then synchronize", "pinned flag + 200 us sleeps", "query + 1 ms sleeps"};
    unsigned seq = 0;
    for (int mode = 0; mode < 6; mode++) {
        const auto t0 = threads(); const double c0 = cpu(), w0 = wall();
        for (int i = 0; i < 10; i++) {
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 2000000LL, d);       // 20 ms at 100 MHz
            if (mode == 0) hipStreamSynchronize(s);
            else if (mode == 1) { hipEventRecord(ev, s); hipEventSynchronize(ev); }
            else if (mode == 2) { while (hipStreamQuery(s) == hipErrorNotReady) { timespec ts{0, 40000}; nanosleep(&ts, nullptr); } }
            else if (mode == 3) { timespec ts{0, 25000000}; nanosleep(&ts, nullptr); hipStreamSynchronize(s); }
            else if (mode == 4) { ++seq; hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, s, flag, seq); while (*(volatile unsigned*)flag != seq) { timespec ts{0, 200000}; nanosleep(&ts, nullptr); } }
            else { while (hipStreamQuery(s) == hipErrorNotReady) { timespec ts{0, 1000000}; nanosleep(&ts, nullptr); } }
        }
        const double w1 = wall(), c1 = cpu(); const auto t1 = threads();
        printf("%-32s wall %6.1f ms, process cpu %6.1f ms; threads:", names[mode], (w1 - w0) * 1e3, (c1 - c0) * 1e3);
        for (auto& kv : t1) { auto it = t0.find(kv.first); const double u = kv.second.first - (it == t0.end() ? 0 : it->second.first), sy = kv.second.second - (it == t0.end() ? 0 : it->second.second); if (u + sy >= 10) printf(" %d: %.0f + %.0f", kv.first, u, sy); }
        printf("  (main is %d)\n", (int)getpid());
    }
    for (int big = 0; big < 2; big++) {
        const auto t0 = threads(); const double c0 = cpu(), w0 = wall();
        for (int i = 0; i < 10; i++) {
            hipLaunchKernelGGL(spin_scratch, dim3(big ? 2048 : 1), dim3(big ? 512 : 64), 0, s, 2000000LL, d);
            timespec ts{0, 25000000}; nanosleep(&ts, nullptr); hipStreamSynchronize(s);
        }
        const double w1 = wall(), c1 = cpu(); const auto t1 = threads();
        printf("%-32s wall %6.1f ms, process cpu %6.1f ms; threads:", big ? "scratch kernel, 2048 x 512" : "scratch kernel, 1 wave", (w1 - w0) * 1e3, (c1 - c0) * 1e3);
        for (auto& kv : t1) { auto it = t0.find(kv.first); const double u = kv.second.first - (it == t0.end() ? 0 : it->second.first), sy = kv.second.second - (it == t0.end() ? 0 : it->second.second); if (u + sy >= 10) printf(" %d: %.0f + %.0f", kv.first, u, sy); }
        printf("  (main is %d)\n", (int)getpid());
    }
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    return 0;
}
