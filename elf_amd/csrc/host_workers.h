// Process-wide pool of host worker threads, created at the first use and parked on a condition variable in between (plain C++, no
// HIP): the per-game host work of a move boundary (selfplay_host.hip) and the trainer's queue draws (record_host.cpp) spread over it.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <unistd.h>

#include <pthread.h>

#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

class HostWorkers {
 public:
  // One pool per PROCESS, never destroyed: the object is leaked on purpose (no join at exit -- a forked child that leaves through
  // exit() would otherwise join handles that name the parent's threads), and a child process gets a fresh object from the atfork
  // handler (the parent's mutexes may have been held by threads that do not exist in the child).
  static HostWorkers& get() {
    static std::once_flag once;
    std::call_once(once, [] {
      slot() = new HostWorkers();
      pthread_atfork(nullptr, nullptr, [] { slot() = new HostWorkers(); });   // child: abandon the inherited object, threads and locks
    });
    return *slot();
  }
  // fn(i) for i in [0, n), strided over `nt` participants (the caller is participant 0); returns when all of them are done.  An
  // exception thrown by fn on a worker is carried to the caller and rethrown here once the region has drained.
  void run(size_t n, unsigned nt, const std::function<void(size_t)>& fn) {
    std::unique_lock<std::mutex> call(call_mu_);          // one parallel region at a time (contexts on several host threads)
    ensure(nt - 1);
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn; n_ = n; nt_ = nt; pending_ = nt - 1; ++epoch_; failed_ = nullptr;
    }
    cv_.notify_all();
    std::exception_ptr mine;
    try {
      for (size_t i = 0; i < n; i += nt) fn(i);
    } catch (...) {
      mine = std::current_exception();
    }
    std::exception_ptr theirs;
    {
      std::unique_lock<std::mutex> lk(mu_);
      done_.wait(lk, [&] { return pending_ == 0; });
      fn_ = nullptr;
      theirs = failed_;
      failed_ = nullptr;
    }
    call.unlock();
    if (mine) std::rethrow_exception(mine);
    if (theirs) std::rethrow_exception(theirs);
  }
 private:
  HostWorkers() = default;
  ~HostWorkers() = delete;                                 // leaked by design (see get())
  static HostWorkers*& slot() { static HostWorkers* p = nullptr; return p; }
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
            cv_.wait(lk, [&] { return epoch_ != seen; });
            seen = epoch_; fn = fn_; n = n_; nt = nt_;
          }
          if (id < nt) {
            std::exception_ptr ex;
            try {
              for (size_t i = id; i < n; i += nt) (*fn)(i);
            } catch (...) {
              ex = std::current_exception();
            }
            std::lock_guard<std::mutex> lk(mu_);
            if (ex && !failed_) failed_ = ex;
            if (--pending_ == 0) done_.notify_one();
          }
        }
      });
      th_.back().detach();                                 // never joined: the pool lives as long as the process
    }
  }
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> th_;
  const std::function<void(size_t)>* fn_ = nullptr;
  size_t n_ = 0;
  unsigned nt_ = 0, pending_ = 0;
  uint64_t epoch_ = 0;
  std::exception_ptr failed_;
};


// how many participants a parallel region may use: the machine's cores, ELF_AMD_HOST_THREADS if set (one rank of several on a node),
// at most `cap`
inline unsigned host_worker_count(unsigned cap) {
  unsigned nt = std::thread::hardware_concurrency();
  if (const char* e = getenv("ELF_AMD_HOST_THREADS")) { const int v = atoi(e); if (v > 0) nt = (unsigned)v; }
  if (nt > cap) nt = cap;
  return nt < 1 ? 1 : nt;
}
