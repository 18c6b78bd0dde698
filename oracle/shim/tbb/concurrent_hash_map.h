#pragma once
#include <functional>
#include <mutex>
#include <unordered_map>
#include <memory>
namespace tbb {
namespace interface5 { template <typename T> inline size_t tbb_hasher(const T& t) { return std::hash<T>()(t); } }
template <typename K, typename V>
class concurrent_hash_map {
  struct H { size_t operator()(const K& k) const { return interface5::tbb_hasher<K>(k); } };
  using Map = std::unordered_map<K, V, H>;
 public:
  using value_type = typename Map::value_type;
  class const_accessor { public: const value_type* p = nullptr; const value_type& operator*() const {return *p;} const value_type* operator->() const {return p;} };
  class accessor { public: value_type* p = nullptr; value_type& operator*() const {return *p;} value_type* operator->() const {return p;} };
  bool insert(accessor& a, const K& k) { std::lock_guard<std::mutex> l(m_); auto it = map_.find(k); bool fresh = false; if (it == map_.end()) { it = map_.emplace(std::piecewise_construct, std::forward_as_tuple(k), std::forward_as_tuple()).first; fresh = true; } a.p = &*it; return fresh; }
  bool find(accessor& a, const K& k) const { std::lock_guard<std::mutex> l(m_); auto it = const_cast<Map&>(map_).find(k); if (it == map_.end()) return false; a.p = &*it; return true; }
  bool find(const_accessor& a, const K& k) const { std::lock_guard<std::mutex> l(m_); auto it = map_.find(k); if (it == map_.end()) return false; a.p = &*it; return true; }
  Map& range() { return map_; }
 private:
  mutable std::mutex m_; Map map_;
};
}
