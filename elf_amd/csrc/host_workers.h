// Process-wide pool of host worker threads, created at the first use and parked on a condition variable in between (plain C++, no
// HIP): the per-game host work of a move boundary (selfplay_host.hip) and the trainer's queue draws (record_host.cpp) spread over it.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <unistd.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

class HostWorkers {
 public:
  static HostWorkers& get() { static HostWorkers w; return w; }
  // fn(i) for i in [0, n), strided over `nt` participants (the caller is participant 0); returns when all of them are done
  void run(size_t n, unsigned nt, const std::function<void(size_t)>& fn) {
    std::unique_lock<std::mutex> call(call_mu_);          // one parallel region at a time (contexts on several host threads)
    if (owner_ != getpid()) {                              // a forked child inherits the object but not the threads: start over
      (void)new std::vector<std::thread>(std::move(th_));   // abandoned, never destroyed: the handles name threads of the parent
      th_.clear();
      owner_ = getpid();
    }
    ensure(nt - 1);
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn; n_ = n; nt_ = nt; pending_ = nt - 1; ++epoch_;
    }
    cv_.notify_all();
    for (size_t i = 0; i < n; i += nt) fn(i);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }
  ~HostWorkers() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
 private:
  void ensure(unsigned k) {
    while (th_.size() < k) {
      const unsigned id = (unsigned)th_.size() + 1;        // participant index of this worker
      // a worker created during a later region must not mistake that region for a new epoch before it is asked to run
      const uint64_t seen0 = epoch_;
      th_.emplace_back([this, id, seen0] {
        uint64_t seen = seen0;
        for (;;) {
          const std::function<void(size_t)>* fn; size_t n; unsigned nt;
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
            if (stop_) return;
            seen = epoch_; fn = fn_; n = n_; nt = nt_;
          }
          if (id < nt) {
            for (size_t i = id; i < n; i += nt) (*fn)(i);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_one();
          }
        }
      });
    }
  }
  pid_t owner_ = getpid();
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> th_;
  const std::function<void(size_t)>* fn_ = nullptr;
  size_t n_ = 0;
  unsigned nt_ = 0, pending_ = 0;
  uint64_t epoch_ = 0;
  bool stop_ = false;
};


// how many participants a parallel region may use: the machine's cores, ELF_AMD_HOST_THREADS if set (one rank of several on a node),
// at most `cap`
inline unsigned host_worker_count(unsigned cap) {
  unsigned nt = std::thread::hardware_concurrency();
  if (const char* e = getenv("ELF_AMD_HOST_THREADS")) { const int v = atoi(e); if (v > 0) nt = (unsigned)v; }
  if (nt > cap) nt = cap;
  return nt < 1 ? 1 : nt;
}
