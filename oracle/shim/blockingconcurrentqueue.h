#pragma once
#include <cassert>
#include <thread>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <chrono>
namespace moodycamel {
template <typename T> class BlockingConcurrentQueue {
 public:
  void enqueue(const T& v) { { std::lock_guard<std::mutex> l(m_); q_.push_back(v);} cv_.notify_one(); }
  void wait_dequeue(T& v) { std::unique_lock<std::mutex> l(m_); cv_.wait(l,[&]{return !q_.empty();}); v=q_.front(); q_.pop_front(); }
  template <typename D> bool wait_dequeue_timed(T& v, D d) { std::unique_lock<std::mutex> l(m_); if(!cv_.wait_for(l,d,[&]{return !q_.empty();})) return false; v=q_.front(); q_.pop_front(); return true; }
 private:
  std::mutex m_; std::condition_variable cv_; std::deque<T> q_;
};
}
