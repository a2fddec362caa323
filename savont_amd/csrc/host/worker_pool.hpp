// worker_pool.hpp -- persistent host worker pool shared by the stages (the reference uses its rayon pool for the same loops).
// run(n, f) calls f(0..n-1) on the workers plus the calling thread, dynamic scheduling in index order; the first exception
// thrown by a task is rethrown in the caller.  SAVONT_THREADS overrides the size (default min(32, hardware threads)).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace savont {

class WorkerPool {
public:
    static WorkerPool& get() { static WorkerPool* p = new WorkerPool(); return *p; }   // never destroyed: workers are detached
    size_t size() const { return workers_.size() + 1; }
    void run(size_t n, const std::function<void(size_t)>& f) {
        if (n == 0) return;
        if (workers_.empty() || n == 1 || busy_.exchange(true)) { for (size_t i = 0; i < n; i++) f(i); return; }   // nested / concurrent use: inline
        {
            std::lock_guard<std::mutex> l(m_);
            fn_ = &f; n_ = n; next_.store(0); pending_ = workers_.size(); err_ = nullptr; gen_++;
        }
        cv_.notify_all();
        work();                                                                   // the caller helps
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> l(m_);
            done_.wait(l, [&] { return pending_ == 0; });
            fn_ = nullptr; err = err_; err_ = nullptr;
        }
        busy_.store(false);
        if (err) std::rethrow_exception(err);
    }
private:
    WorkerPool() {
        // CPUs this process may use: hardware threads, capped by the container's CPU quota, shared between the ranks of a node
        // (torchrun sets LOCAL_WORLD_SIZE).  1.5x oversubscription: tasks also wait on the GPU.  SAVONT_THREADS overrides.
        double avail = (double)std::max(1u, std::thread::hardware_concurrency());
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long quota = 0, period = 0; char q[32] = {0};
            if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm' && period > 0 && (quota = atoll(q)) > 0) avail = std::min(avail, (double)quota / (double)period);
            fclose(f);
        }
        if (const char* e = getenv("LOCAL_WORLD_SIZE")) avail /= (double)std::max(1, atoi(e));
        unsigned T = (unsigned)std::min(32.0, std::max(2.0, 1.5 * avail + 0.5));
        if (const char* e = getenv("SAVONT_THREADS")) T = (unsigned)std::max(1, atoi(e));
        for (unsigned t = 1; t < T; t++) workers_.emplace_back([this] { loop(); });
        for (auto& w : workers_) w.detach();
    }
    void work() {
        for (size_t i; (i = next_.fetch_add(1)) < n_;) {
            try { (*fn_)(i); }
            catch (...) { std::lock_guard<std::mutex> l(m_); if (!err_) err_ = std::current_exception(); next_.store(n_); }
        }
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return gen_ != seen; }); seen = gen_; }
            work();
            { std::lock_guard<std::mutex> l(m_); if (--pending_ == 0) done_.notify_all(); }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_; std::condition_variable cv_, done_;
    const std::function<void(size_t)>* fn_ = nullptr; size_t n_ = 0; std::atomic<size_t> next_{0}; size_t pending_ = 0; unsigned long long gen_ = 0;
    std::exception_ptr err_; std::atomic<bool> busy_{false};
};
template <class F> inline void par_for(size_t n, F f) { WorkerPool::get().run(n, std::function<void(size_t)>(f)); }

}  // namespace savont
