// worker_pool.hpp -- persistent host worker pool shared by the stages (the reference uses its rayon pool for the same loops).
// run(n, f) calls f(0..n-1) on the workers plus the calling thread, dynamic scheduling in index order; the first exception
// thrown by a task is rethrown in the caller.  Several threads may call run() at the same time (pipelines of several samples in
// flight on one GPU, bench.py --in-flight; nested calls from inside a task): every call is its own job in a FIFO list, idle workers
// take indices from the oldest job that still has some, and a caller always works on its own job, so no call waits for another.
// SAVONT_THREADS overrides the size (default: the CPUs this process may use).
#pragma once
#include <pthread.h>
#include <sched.h>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace savont {

class WorkerPool {
public:
    static std::atomic<void (*)()>& thread_hook() { static std::atomic<void (*)()> h{nullptr}; return h; }   // read once by every worker when it starts: set before the first get() to reach all of them
    static WorkerPool& get() { static WorkerPool* p = new WorkerPool(); return *p; }
    size_t threads() const { return workers_.size() + 1; }                        // the CPUs of this process's share: the callers' thread counts as one   // never destroyed: workers are detached
    size_t size() const { return workers_.size() + 1; }
    void run(size_t n, const std::function<void(size_t)>& f) {
        if (n == 0) return;
        if (workers_.empty() || n == 1) { for (size_t i = 0; i < n; i++) f(i); return; }
        std::shared_ptr<Job> job = std::make_shared<Job>();
        job->f = &f; job->n = n;
        { std::lock_guard<std::mutex> l(m_); jobs_.push_back(job); }
        cv_.notify_all();
        for (size_t i; (i = job->next.fetch_add(1)) < n;) execute(*job, i);       // the caller helps with its own job
        {
            std::unique_lock<std::mutex> l(m_);
            retire(job);
            done_.wait(l, [&] { return job->done.load() >= n; });
        }
        if (job->err) std::rethrow_exception(job->err);
    }
private:
    struct Job {
        const std::function<void(size_t)>* f = nullptr; size_t n = 0;
        std::atomic<size_t> next{0}, done{0};
        std::exception_ptr err; std::mutex em;
    };
    WorkerPool() {
        // CPUs this process may use: hardware threads, capped by the container's CPU quota, shared between the ranks of a node
        // (torchrun sets LOCAL_WORLD_SIZE).  One thread per CPU: the step is CPU-bound (Stage-4a POA) and several samples are in flight, so
        // oversubscription only adds context switches (measured at 16 CPUs, 4-5 samples in flight: 1.0-1.1 CPU-s per step with 14-16 threads,
        // 1.3-1.4 with 24; 64-72 ms per step against 82-88).  SAVONT_THREADS overrides.
        double avail = (double)std::max(1u, std::thread::hardware_concurrency());
        { cpu_set_t set; CPU_ZERO(&set); if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) avail = std::min(avail, (double)CPU_COUNT(&set)); }   // taskset / a launcher's CPU binding
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long quota = 0, period = 0; char q[32] = {0};
            if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm' && period > 0 && (quota = atoll(q)) > 0) avail = std::min(avail, (double)quota / (double)period);
            fclose(f);
        }
        if (const char* e = getenv("LOCAL_WORLD_SIZE")) avail /= (double)std::max(1, atoi(e));
        unsigned T = (unsigned)std::min(64.0, std::max(2.0, avail + 0.5));
        if (const char* e = getenv("SAVONT_THREADS")) T = (unsigned)std::max(1, atoi(e));
        for (unsigned t = 1; t < T; t++) workers_.emplace_back([this] { loop(); });
        for (auto& w : workers_) w.detach();
    }
    void execute(Job& j, size_t i) {
        try { (*j.f)(i); }
        catch (...) {
            { std::lock_guard<std::mutex> l(j.em); if (!j.err) j.err = std::current_exception(); }
            // fail fast: the indices nobody has taken yet are skipped and counted as done (after a HIP failure every further task would issue work on a broken context)
            const size_t taken = j.next.exchange(j.n);
            if (taken < j.n) j.done.fetch_add(j.n - taken);
        }
        if (j.done.fetch_add(1) + 1 >= j.n) { std::lock_guard<std::mutex> l(m_); done_.notify_all(); }
    }
    void retire(const std::shared_ptr<Job>& job) {                               // m_ held: a job whose indices are all taken leaves the list
        for (auto it = jobs_.begin(); it != jobs_.end(); ++it) if (it->get() == job.get()) { jobs_.erase(it); break; }
    }
    void loop() {
        pthread_setname_np(pthread_self(), "svt-pool");                          // tools/thread_cpu.py tells the pool from the HIP runtime's threads by name
        if (void (*h)() = thread_hook().load()) h();                               // e.g. the development sampler arms its per-thread timer
        for (;;) {
            std::shared_ptr<Job> job; size_t i = 0;
            {
                std::unique_lock<std::mutex> l(m_);
                for (;;) {
                    while (!jobs_.empty() && jobs_.front()->next.load() >= jobs_.front()->n) jobs_.pop_front();
                    for (auto& j : jobs_) if (j->next.load() < j->n) { job = j; break; }
                    if (job) break;
                    cv_.wait(l);
                }
            }
            while ((i = job->next.fetch_add(1)) < job->n) execute(*job, i);       // stay on the job while it has indices (no lock per task)
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_; std::condition_variable cv_, done_;
    std::deque<std::shared_ptr<Job>> jobs_;
};
template <class F> inline void par_for(size_t n, F f) { WorkerPool::get().run(n, std::function<void(size_t)>(f)); }

}  // namespace savont
