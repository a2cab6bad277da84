// host_pool.h — the host worker pool behind parallel_for (internal.h).  Definitions, not declarations: included by exactly one
// translation unit of a library (glprover.hip for libglprover.so; host_selftest.cpp for the CPU test build).
#pragma once
#include <sched.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>

namespace glpi {

// One process-wide set of workers, created on first use and never torn down (the threads sleep on a condition variable and end
// with the process).  A job is (fn, arg, n items); items are handed out by compare-and-swap on a counter tagged with the job's
// generation, so a worker that wakes up late for job g can never take an item of job g + 1 with the stale fn of job g.  The
// caller works too and returns when all n items are done.
namespace {
struct HostPool {
    std::mutex run_mu;  // one job at a time; a caller that finds it taken runs its items itself
    std::mutex mu;
    std::condition_variable cv;
    uint64_t gen = 0;            // under mu
    size_t n = 0;                // under mu
    void (*fn)(void*, size_t) = nullptr;
    void* arg = nullptr;
    std::atomic<uint64_t> next{0};  // (generation << 32) | next item
    std::atomic<size_t> done{0};
    unsigned workers = 0;
    pid_t pid = 0;

    // items of generation g until none is left (or the generation has moved on)
    void drain(uint64_t g, size_t n_items, void (*f)(void*, size_t), void* a) {
        for (;;) {
            uint64_t cur = next.load(std::memory_order_acquire);
            for (;;) {
                if ((cur >> 32) != (g & 0xFFFFFFFFu) || (cur & 0xFFFFFFFFu) >= n_items) return;
                if (next.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel)) break;
            }
            f(a, (size_t)(cur & 0xFFFFFFFFu));
            done.fetch_add(1, std::memory_order_release);
        }
    }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            uint64_t g;
            size_t n_items;
            void (*f)(void*, size_t);
            void* a;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return gen != seen; });
                seen = g = gen;
                n_items = n;
                f = fn;
                a = arg;
            }
            drain(g, n_items, f, a);
        }
    }
};
HostPool* g_pool = nullptr;
std::once_flag g_pool_once;
}  // namespace

// Worker threads of this PROCESS, the caller included.  One process per GPU: eight ranks on one host must not
// each start a thread per hardware thread, so the width is, in this order,
//   GLP_HOST_THREADS                      when set (clamped to 1 .. 32);
//   the CPUs this process may run on      (sched_getaffinity) when the launcher has pinned the rank to a subset of the host
//                                          (bench.py pins rank r to the r-th slice of the cores before creating its context);
//   hardware threads / LOCAL_WORLD_SIZE   when torch.distributed.run / the launcher exported LOCAL_WORLD_SIZE > 1, floor 2;
//   hardware threads                      otherwise;
// never more than 32 (the per-proof host work of a batch stops scaling well before that), and never more than the CPUs the
// cgroup's quota grants (round 5: the GPU boxes show 256 logical CPUs and allow 16 CPUs of run time — cpu.max "1600000 100000";
// threads beyond the quota are throttled, not run), floor 2.
inline unsigned cgroup_cpu_quota() {  // CPUs of run time per period, rounded up; 0 = no quota / unknown
    auto ratio = [](long long q, long long per) -> unsigned { return q > 0 && per > 0 ? (unsigned)((q + per - 1) / per) : 0u; };
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "max 100000" or "1600000 100000"
        char q[32] = {0};
        long long per = 0;
        const int k = fscanf(f, "%31s %lld", q, &per);
        fclose(f);
        if (k == 2 && q[0] >= '0' && q[0] <= '9') return ratio(atoll(q), per);
        return 0;
    }
    long long q = -1, per = 0;  // cgroup v1
    if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(f, "%lld", &q) != 1) q = -1;
        fclose(f);
    }
    if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (fscanf(f, "%lld", &per) != 1) per = 0;
        fclose(f);
    }
    return ratio(q, per);
}

unsigned host_pool_width() {
    static const unsigned hw = [] {
        if (const char* e = getenv("GLP_HOST_THREADS")) {
            const int h = atoi(e);
            return h <= 0 ? 1u : (h > 32 ? 32u : (unsigned)h);
        }
        long onl = sysconf(_SC_NPROCESSORS_ONLN);  // the host's CPUs (hardware_concurrency may or may not honour the affinity mask)
        unsigned all = onl > 0 ? (unsigned)onl : std::thread::hardware_concurrency();
        if (all == 0) all = 1;
        unsigned h = all;
        cpu_set_t set;
        CPU_ZERO(&set);
        const int allowed = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : 0;
        unsigned lws = 1;  // ranks of this job on this host
        if (const char* l = getenv("LOCAL_WORLD_SIZE"))
            if (atoi(l) > 1) lws = (unsigned)atoi(l);
        if (allowed > 0 && (unsigned)allowed < all) {
            h = (unsigned)allowed;  // pinned: the slice is ours alone
        } else if (lws > 1) {
            h = all / lws;
            if (h < 2) h = 2;
        }
        // the quota is the whole cgroup's (every rank of the job lives in it): a rank's share is quota / LOCAL_WORLD_SIZE — eight
        // ranks each sizing their pool to the full 16-CPU grant would be 128 runnable threads against 16 CPUs of run time
        if (const unsigned quota = cgroup_cpu_quota()) {
            const unsigned share = quota / lws < 2 ? 2u : quota / lws;
            if (share < h) h = share;
        }
        return h == 0 ? 1u : (h > 32 ? 32u : h);
    }();
    return hw;
}

void host_pool_run(size_t n_items, size_t width, void (*fn)(void*, size_t), void* arg) {
    std::call_once(g_pool_once, [] {
        HostPool* p = new HostPool;
        p->pid = getpid();
        p->workers = host_pool_width() - 1;
        for (unsigned i = 0; i < p->workers; ++i) std::thread([p] { p->worker(); }).detach();
        g_pool = p;
    });
    HostPool* p = g_pool;
    // no workers, a forked child (the threads did not come along), more items than the tag leaves room for, or the pool is busy
    // with another context's job: the caller does it alone
    if (width <= 1 || p->workers == 0 || p->pid != getpid() || n_items >= 0xFFFFFFFFu || !p->run_mu.try_lock()) {
        for (size_t k = 0; k < n_items; ++k) fn(arg, k);
        return;
    }
    uint64_t g;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        g = ++p->gen;
        p->n = n_items;
        p->fn = fn;
        p->arg = arg;
        p->done.store(0, std::memory_order_relaxed);
        p->next.store((g & 0xFFFFFFFFu) << 32, std::memory_order_release);
    }
    p->cv.notify_all();
    p->drain(g, n_items, fn, arg);
    while (p->done.load(std::memory_order_acquire) < n_items) std::this_thread::yield();  // items in flight on workers: short
    p->run_mu.unlock();
}


}  // namespace glpi
