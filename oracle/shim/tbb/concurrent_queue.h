#pragma once
#include <deque>
#include <mutex>
namespace tbb { template <typename T> class concurrent_queue { public: void push(const T& v){std::lock_guard<std::mutex> l(m_); q_.push_back(v);} bool try_pop(T& v){std::lock_guard<std::mutex> l(m_); if(q_.empty()) return false; v=q_.front(); q_.pop_front(); return true;} private: std::mutex m_; std::deque<T> q_; }; }
