// par.h -- minimal host parallel-for on std::thread (the library deliberately
// does not link an OpenMP runtime: it is loaded into processes, e.g. Python
// with torch, that already carry one).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include "knobs.h"
#ifdef __linux__
#include <sched.h>
#include <sys/mman.h>
#endif
#include <new>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

namespace crp {

// CPUs this process may actually use: the affinity mask, cut by a cgroup CPU quota when there is one (a container given
// 16 CPUs' worth of a 256-thread host reports 256 from hardware_concurrency(); 64 threads on 16 CPUs' worth of time made
// the format builders slower, not faster).
inline int usable_cpus()
{
    int n = (int) std::thread::hardware_concurrency();
#ifdef __linux__
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = CPU_COUNT(&set);
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r"))
    {
        long long quota = 0, period = 0;
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
            n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
        fclose(f);
    }
    else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))
    {
        long long quota = 0, period = 100000;
        if (fscanf(g, "%lld", &quota) == 1 && quota > 0)
        {
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
            n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
        }
        fclose(g);
    }
#endif
    return n > 0 ? n : 1;
}

inline int host_threads()
{
    int n = knobs().num_threads;
    if (n <= 0)
    {
        static const int cpus = usable_cpus();
        n = cpus;
    }
    if (n <= 0) n = 1;
    return std::min(n, 64);
}

// fn(begin, end, thread_id) over [0, n) in dynamically claimed chunks.
template <typename F>
void parallel_chunks(long long n, long long chunk, F fn)
{
    if (n <= 0) return;
    if (chunk < 1) chunk = 1;
    const long long nchunks = (n + chunk - 1) / chunk;
    const int nthr = (int) std::min<long long>(host_threads(), nchunks);
    if (nthr <= 1)
    {
        fn(0LL, n, 0);
        return;
    }
    std::atomic<long long> next(0);
    auto worker = [&](int tid) {
        for (;;)
        {
            const long long c = next.fetch_add(1);
            if (c >= nchunks) break;
            const long long b = c * chunk;
            fn(b, std::min(n, b + chunk), tid);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t) nthr - 1);
    for (int t = 1; t < nthr; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto &th : pool) th.join();
}

// std::vector whose resize() leaves new elements uninitialised (for arrays of hundreds of MB that are filled by
// parallel_fill / by all threads afterwards: a value-initialising resize touches every page from ONE thread first)
// Blocks of 4 MiB and more are aligned to 2 MiB and advised MADV_HUGEPAGE: first-touch page faults take a process-wide lock, and
// with 4 KiB pages the builders' tens of gigabytes of fresh memory were faulted in at about the same speed by 4 threads and by 16
// (profiles/r04_build_time.txt).
template <typename T>
struct default_init_allocator : std::allocator<T>
{
    template <typename U> struct rebind { typedef default_init_allocator<U> other; };
    default_init_allocator() = default;
    template <typename U> default_init_allocator(const default_init_allocator<U> &) {}
    static constexpr size_t HUGE_MIN = (size_t) 4 << 20, HUGE_ALIGN = (size_t) 2 << 20;
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
#ifdef __linux__
        if (bytes >= HUGE_MIN)
        {
            void *p = nullptr;
            if (posix_memalign(&p, HUGE_ALIGN, (bytes + HUGE_ALIGN - 1) / HUGE_ALIGN * HUGE_ALIGN) != 0 || p == nullptr) throw std::bad_alloc();
            (void) madvise(p, (bytes + HUGE_ALIGN - 1) / HUGE_ALIGN * HUGE_ALIGN, MADV_HUGEPAGE);
            return static_cast<T *>(p);
        }
#endif
        return static_cast<T *>(::operator new(bytes));
    }
    void deallocate(T *p, size_t n)
    {
#ifdef __linux__
        if (n * sizeof(T) >= HUGE_MIN) { free(p); return; }
#endif
        ::operator delete(p);
    }
    template <typename U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <typename T> using big_vector = std::vector<T, default_init_allocator<T>>;

// Owns a heap object and hands it to a detached thread on scope exit: the builders' temporaries are tens of gigabytes on the
// largest inputs, and returning them to the system took seconds of the first product's time.
template <typename T>
struct released_async
{
    T *p;
    released_async() : p(new T) {}
    released_async(const released_async &) = delete;
    released_async &operator=(const released_async &) = delete;
    ~released_async()
    {
        T *q = p;
        p = nullptr;
        if (knobs().sync_release) { delete q; return; }      // (leak checkers: nothing may outlive main)
        try { std::thread([q] { delete q; }).detach(); }
        catch (...) { delete q; }
    }
    T &operator*() { return *p; }
    T *operator->() { return p; }
};

template <typename V, typename T>
void parallel_fill(V &v, size_t n, T value)
{
    v.resize(n);
    auto *p = v.data();
    parallel_chunks((long long) n, 1 << 20, [&](long long b, long long e, int) { std::fill(p + b, p + e, value); });
}

}  // namespace crp
