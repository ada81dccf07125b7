// Host-side spinning thread pool of the library (frx_api.cpp: host-vector L-BFGS, set-up and initial guess).  Host code only.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

namespace frx {

// ---- a tiny spinning thread pool (one evaluation round is ~100 us: no condvars) ----
// Work is STATICALLY partitioned: item i always runs on worker i % n, and every worker is pinned to its
// own CPU.  Each candidate's L-BFGS state (1.5 MB of (s, y) history at mem_size 128) is allocated, first
// touched and then always updated by the same core, so it stays in that core's cache hierarchy.
class SpinPool {
public:
    explicit SpinPool(int nthreads) : n_(std::max(1, nthreads)) {
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        std::vector<int> cpus;
        if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
            for (int c = 0; c < CPU_SETSIZE; c++)
                if (CPU_ISSET(c, &allowed)) cpus.push_back(c);
        // Every live pool of the process pins its workers onto its OWN range of the allowed CPUs: pool k (the lowest free index, taken at
        // construction, returned at destruction) uses CPUs [k n, (k + 1) n) of the list - several host threads driving separate handles
        // (frx_multi: one per device) neither share cores nor depend on which pool was created first.  A pool whose range does not fit
        // the allowed set runs unpinned.  The CALLER's thread is never pinned: its affinity is the caller's business.  FRX_PIN=0 disables
        // pinning altogether; FRX_PIN_OFFSET / FRX_PIN_STRIDE choose which allowed CPUs the ranges are cut from.
        const char *pe = std::getenv("FRX_PIN"), *po = std::getenv("FRX_PIN_OFFSET"), *ps = std::getenv("FRX_PIN_STRIDE");
        const int off = po ? std::atoi(po) : 0, stride = std::max(1, ps ? std::atoi(ps) : 1);
        live_pools().fetch_add(1, std::memory_order_acq_rel);
        {
            std::lock_guard<std::mutex> g(slot_lock());
            std::vector<char> &used = slots();
            size_t k = 0;
            while (k < used.size() && used[k]) k++;
            if (k == used.size()) used.push_back(0);
            used[k] = 1; slot_ = (int)k;
        }
        const long first = (long)off + (long)slot_ * n_ * stride, last = first + (long)(n_ - 1) * stride;
        const bool do_pin = !(pe && pe[0] == '0') && n_ > 1 && last < (long)cpus.size();
        for (int t = 1; t < n_; t++) {
            workers_.emplace_back([this, t] { loop(t); });
            if (do_pin) pin(workers_.back().native_handle(), cpus[first + (long)t * stride]);
        }
    }
    ~SpinPool() {
        stop_.store(true, std::memory_order_release);
        for (auto &w : workers_) w.join();
        live_pools().fetch_sub(1, std::memory_order_acq_rel);
        std::lock_guard<std::mutex> g(slot_lock());
        if (slot_ >= 0 && slot_ < (int)slots().size()) slots()[slot_] = 0;
    }
    int size() const { return n_; }
    // fn(i) for i in [0, count): worker t takes i = t, t + n, t + 2n, ...
    template <class F> void run(int count, F &&fn) {
        if (count <= 0) return;
        if (n_ == 1) { for (int i = 0; i < count; i++) fn(i); return; }
        fn_ = [&fn](int i) { fn(i); };
        count_ = count;
        pending_.store(n_ - 1, std::memory_order_relaxed);
        epoch_.fetch_add(1, std::memory_order_release);
        for (int i = 0; i < count; i += n_) fn_(i);
        while (pending_.load(std::memory_order_acquire) > 0) cpu_relax();
    }
private:
    static void cpu_relax() { __builtin_ia32_pause(); }
    static void pin(pthread_t th, int cpu) {
        cpu_set_t m;
        CPU_ZERO(&m);
        CPU_SET(cpu, &m);
        pthread_setaffinity_np(th, sizeof(m), &m);
    }
    void loop(int t) {
        unsigned seen = 0;
        int idle = 0;
        while (!stop_.load(std::memory_order_acquire)) {
            unsigned e = epoch_.load(std::memory_order_acquire);
            if (e != seen) {
                seen = e;
                for (int i = t; i < count_; i += n_) fn_(i);
                pending_.fetch_sub(1, std::memory_order_release);
                idle = 0;
            } else if (++idle > (live_pools().load(std::memory_order_relaxed) > 1 ? 2000 : 200000)) { std::this_thread::yield(); idle = 0; }
            else cpu_relax();
        }
    }
    int n_, slot_ = -1;
    static std::mutex &slot_lock() { static std::mutex m; return m; }
    static std::vector<char> &slots() { static std::vector<char> v; return v; }
    std::vector<std::thread> workers_;
    std::function<void(int)> fn_;
    int count_ = 0;
    std::atomic<int> pending_{0};
    std::atomic<unsigned> epoch_{0};
    std::atomic<bool> stop_{false};
    static std::atomic<int> &live_pools() { static std::atomic<int> n{0}; return n; }
};

// ---- the set-up pool: sleeping workers that outlive the call ----
// SpinPool pays for its threads on every construction (measured: 7-8 ms for 2 ... 32 pinned workers, box after box - more than the work of
// frx_initial_guess for 32 candidates, 0.8 ms on ONE core, which round 3 ran on such a pool: 20.9 ms).  Set-up work (H->V enumeration of a batch's
// polytopes, the waypoint solves of large batches) is short, rare and embarrassingly parallel: it goes to ONE process-wide pool whose workers
// are created on first use, sleep on a condition variable in between (no spinning, no pinning: they cost nothing while a plan runs) and are
// joined when the library is unloaded.  Jobs are serialised (several host threads - one per device in frx_multi - take turns); tasks are handed
// out dynamically from a counter, so results must not depend on which worker ran what - every caller writes task-indexed output.
class TaskPool {
public:
    static TaskPool &get() { static TaskPool p; return p; }
    enum { MAX_WORKERS = 31 };
    // fn(task, worker) for every task in [0, ntasks) on up to `threads` threads (the caller is worker 0 and takes part); worker < threads
    template <class F> void run(int ntasks, int threads, F &&fn) {
        if (ntasks <= 0) return;
        threads = std::max(1, std::min({threads, ntasks, (int)MAX_WORKERS + 1}));
        if (threads == 1) { for (int i = 0; i < ntasks; i++) fn(i, 0); return; }
        std::lock_guard<std::mutex> job(job_lock_);
        grow(threads - 1);
        std::function<void(int, int)> f = [&fn](int i, int w) { fn(i, w); };
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &f; ntasks_ = ntasks; next_.store(0, std::memory_order_relaxed);
            want_ = threads - 1; joined_ = 0; running_ = 0; gen_++;
        }
        cv_.notify_all();
        for (int i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < ntasks;) f(i, 0);
        std::unique_lock<std::mutex> g(m_);
        want_ = joined_;                                                     // nobody else may join this job any more ...
        done_.wait(g, [this] { return running_ == 0; });                     // ... and those who did have finished
        fn_ = nullptr;
    }
private:
    //  fork safety: a forked child has this object but none of its threads - joining them in the static destructor would never return.
    // The child forgets them (the std::thread objects are leaked on purpose: they name threads that do not exist there) and starts with an empty pool.
    TaskPool() { pthread_atfork(nullptr, nullptr, &TaskPool::after_fork_in_child); }
    static void after_fork_in_child() {
        TaskPool &p = get();
        new std::vector<std::thread>(std::move(p.th_));                       // never destroyed
        p.th_.clear();
        p.fn_ = nullptr; p.ntasks_ = 0; p.want_ = 0; p.joined_ = 0; p.running_ = 0;
    }
    ~TaskPool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void grow(int n) { while ((int)th_.size() < std::min(n, (int)MAX_WORKERS)) { const unsigned seen = gen_; th_.emplace_back([this, seen] { loop(seen); }); } }
    void loop(unsigned seen) {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [&] { return stop_ || (gen_ != seen && joined_ < want_); });
            if (stop_) return;
            seen = gen_;
            const int w = ++joined_;                                         // worker index of this job: 1 .. want
            running_++;
            std::function<void(int, int)> *f = fn_;
            const int nt = ntasks_;
            g.unlock();
            for (int i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < nt;) (*f)(i, w);
            g.lock();
            if (--running_ == 0) done_.notify_all();
        }
    }
    std::mutex job_lock_, m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    std::function<void(int, int)> *fn_ = nullptr;
    std::atomic<int> next_{0};
    int ntasks_ = 0, want_ = 0, joined_ = 0, running_ = 0;
    unsigned gen_ = 0;
    bool stop_ = false;
};

// how many threads `tasks` independent pieces of set-up work of ~`us_per_task` microseconds each are worth: a sleeping worker takes ~50 us to
// get going, and the boxes' cores are shared - at most 32, and only when every thread gets ~500 us of work (the initial guess of the headline
// batch, 0.8 ms in all, stays on the caller's thread; the H->V enumeration of its 4064 polytopes, 17 ms, is spread).  FRX_SETUP_THREADS overrides.
inline int setup_threads(long tasks, double us_per_task) {
    if (const char *e = std::getenv("FRX_SETUP_THREADS")) { const int v = std::atoi(e); if (v >= 1) return std::min(v, (int)TaskPool::MAX_WORKERS + 1); }
    const long hw = std::max(1u, std::thread::hardware_concurrency());
    return (int)std::max(1L, std::min({hw, (long)TaskPool::MAX_WORKERS + 1, (long)(tasks * us_per_task / 500.0)}));
}

} // namespace frx
