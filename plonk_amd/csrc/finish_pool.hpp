// Host helper threads for the arithmetic between two device phases of a proof (no HIP): fetch_commitments (prover.hip)
// turns the bit sums of a commitment group into compressed commitments — a Horner chain of ~20 doublings and additions in
// 64-bit-limb arithmetic per commitment, 13 us each on the bench host, with the device idle and the transcript waiting.
// The chains of a group are independent: a group of four costs 52 us on one thread and 13 on four.
//
// Round 4 tried a pool that parked its workers on a condition variable and measured no gain: waking a thread costs what a
// chain takes.  Here the workers are woken EARLY — `arm()` when the host is about to block in the stream synchronisation
// that precedes the arithmetic, i.e. hundreds of microseconds before the work exists — and then spin on an atomic until
// the job is posted (or the arming is withdrawn); between commitment groups they sleep.  What it costs: up to `workers`
// host threads spinning while the device runs a commitment group's tail.  plonk_gpu_config has no field for it;
// PLONK_HOST_THREADS=0 switches it off, the default is 3 workers when the process may run on at least 8 CPUs (sched_getaffinity), else none.
// Included by prover.hip and by the CPU test harness (tests/csrc/host_arith.cpp).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <thread>

#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define PLONK_CPU_RELAX() _mm_pause()
#else
#define PLONK_CPU_RELAX() std::this_thread::yield()
#endif

namespace plonk {

class FinishPool {
 public:
  typedef void (*TaskFn)(void* arg, int index);
  explicit FinishPool(int workers) : nth_(0) {
    const int want = workers < 0 ? 0 : (workers > MAXW ? MAXW : workers);
    for (int i = 0; i < want; ++i) {
      try {
        th_[i] = std::thread([this] { worker(); });
      } catch (...) {   // the process is out of threads: work with the ones that started (none: everything runs on the caller)
        break;
      }
      nth_ = i + 1;
    }
  }
  ~FinishPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    post(nullptr, nullptr, 0);   // releases a spinning worker
    cv_.notify_all();
    for (int i = 0; i < nth_; ++i) th_[i].join();
  }
  FinishPool(const FinishPool&) = delete;
  FinishPool& operator=(const FinishPool&) = delete;
  int workers() const { return nth_; }
  // The caller is about to wait for the device and will call run() afterwards: wake the workers now.  Every arm() MUST be
  // followed by a run() (count 0 withdraws it) — Armed below does that on every path.
  void arm() {
    if (!nth_) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      main_arm_ = ++arm_gen_;
    }
    cv_.notify_all();
  }
  // fn(arg, i) for i in [0, count) on the workers and the calling thread; returns when all of them are done.  count <= 255.
  void run(TaskFn fn, void* arg, int count) {
    if (!nth_) {
      for (int i = 0; i < count; ++i) fn(arg, i);
      return;
    }
    const uint64_t gen = post(fn, arg, count);
    take(gen);
    while (done_.load(std::memory_order_acquire) < count) PLONK_CPU_RELAX();
  }

 private:
  static constexpr int MAXW = 7;
  // The job word: generation << 32 | count << 16 | next index.  Indices are claimed by compare-and-swap on the WHOLE word, so
  // a worker that is still leaving job g can neither take nor skip an index of job g + 1 (a plain fetch_add on a shared
  // counter could do both: run a task twice, or let run() return while one is still running).
  uint64_t post(TaskFn fn, void* arg, int count) {
    fn_ = fn;
    arg_ = arg;
    done_.store(0, std::memory_order_relaxed);
    const uint64_t gen = job_gen_.load(std::memory_order_relaxed) + 1;
    job_arm_.store(main_arm_, std::memory_order_relaxed);
    word_.store(gen << 32 | (uint64_t)(count & 0xff) << 16, std::memory_order_release);
    job_gen_.store(gen, std::memory_order_release);
    return gen;
  }
  void take(uint64_t gen) {   // run tasks of job `gen` until none is left (or the job is no longer current)
    for (;;) {
      uint64_t w = word_.load(std::memory_order_acquire);
      const uint32_t idx = (uint32_t)(w & 0xffff), count = (uint32_t)((w >> 16) & 0xff);
      if ((w >> 32) != (gen & 0xffffffffu) || idx >= count) return;
      if (!word_.compare_exchange_weak(w, w + 1, std::memory_order_acq_rel)) continue;
      fn_(arg_, (int)idx);
      done_.fetch_add(1, std::memory_order_acq_rel);
    }
  }
  void worker() {
    uint64_t seen_arm = 0, seen_job = job_gen_.load(std::memory_order_acquire);
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || arm_gen_ != seen_arm; });
        if (stop_) return;
        seen_arm = arm_gen_;
      }
      // armed: the job follows within a device phase.  Bounded spin — a caller that never posts (it must not) costs 2 s of one
      // core, not a core for ever.
      const auto t0 = std::chrono::steady_clock::now();
      for (uint32_t spin = 0;; ++spin) {
        const uint64_t j = job_gen_.load(std::memory_order_acquire);
        if (j != seen_job) {
          seen_job = j;
          if (job_arm_.load(std::memory_order_relaxed) < seen_arm) continue;   // a job of an earlier arming (this thread woke late): keep waiting for ours
          take(j);
          break;
        }
        PLONK_CPU_RELAX();
        if ((spin & 0xfffff) == 0xfffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
      }
    }
  }
  int nth_;
  std::thread th_[MAXW];
  std::mutex mu_;
  std::condition_variable cv_;
  bool stop_ = false;
  uint64_t arm_gen_ = 0;    // guarded by mu_
  uint64_t main_arm_ = 0;   // the posting thread's copy of arm_gen_
  std::atomic<uint64_t> job_gen_{0}, job_arm_{0}, word_{0};
  std::atomic<int> done_{0};
  TaskFn fn_ = nullptr;
  void* arg_ = nullptr;
};

// arm() now, run() exactly once later — with the work, or empty from the destructor on an early return
class Armed {
 public:
  explicit Armed(FinishPool* p) : p_(p) { if (p_) p_->arm(); }
  ~Armed() { if (p_) p_->run(nullptr, nullptr, 0); }
  void run(FinishPool::TaskFn fn, void* arg, int count) {
    if (p_) { p_->run(fn, arg, count); p_ = nullptr; }
    else for (int i = 0; i < count; ++i) fn(arg, i);
  }
 private:
  FinishPool* p_;
};

}  // namespace plonk
