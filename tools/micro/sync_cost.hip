// build: /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/sync_cost.hip -o build_tmp/sync_cost   (run on a GPU box: gpurun -- ./build_tmp/sync_cost)
// CPU time a host thread burns while it waits for a kernel: hipStreamSynchronize (default flags), a hipEventBlockingSync event, and both
// after hipSetDeviceFlags(hipDeviceScheduleBlockingSync).   hipcc --offload-arch=gfx950 -O2 tools/micro/sync_cost.hip -o sync_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ctime>
#include <chrono>
__global__ void spin(long long cycles, int* out) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} if (out) *out = 1; }
static double cpu() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int flags = argc > 1 ? atoi(argv[1]) : 0;
    if (flags) printf("hipSetDeviceFlags(BlockingSync) -> %d\n", (int)hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming);
    int* d; hipMalloc(&d, 4);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000LL, d); hipStreamSynchronize(s);
    const long long cyc = 100000000LL * 20 / 100;   // wall_clock64 runs at 100 MHz: 20 ms... x10 below
    for (int mode = 0; mode < 3; mode++) {
        double c0 = cpu(), w0 = wall();
        for (int i = 0; i < 10; i++) {
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 2000000LL, d);       // 20 ms at 100 MHz
            if (mode == 0) hipStreamSynchronize(s); else if (mode == 1) { hipEventRecord(ev, s); hipEventSynchronize(ev); }
            else { while (hipStreamQuery(s) == hipErrorNotReady) { timespec ts{0, 40000}; nanosleep(&ts, nullptr); } }
        }
        printf("%s: wall %.1f ms, process cpu %.1f ms\n", mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "blocking event      " : "query + 40 us sleeps", (wall() - w0) * 1e3, (cpu() - c0) * 1e3);
    }
    (void)cyc;
    return 0;
}
