// TEST INFRASTRUCTURE (oracle only).  Boost is not installed in this image, and the reference's graph search
// (src/path_searching/include/jps_planner/jps_planner/graph_search.h:10,48-49) keeps its OPEN list in a
// boost::heap::d_ary_heap<StatePtr, mutable_<true>, arity<2>, compare<compare_state<StatePtr>>>.  This header restates the
// published algorithm of that container (Boost.Heap, d_ary_heap.hpp: push = append + sift-up; pop = swap root with last,
// shrink, sift-down; sift-up swaps while cmp(parent, child); sift-down picks the top child with std::max_element over the
// children and swaps while !cmp(child, node); increase(handle) = sift-up) with just the members the reference calls, so that
// graph_search.cpp compiles where it lies and keeps its tie order.  It is a restatement from the documentation, not a copy
// of Boost; the parity it gives is "reference search + restated heap".
#pragma once
#include <algorithm>
#include <cstddef>
#include <list>
#include <utility>
#include <vector>

namespace boost { namespace heap {

template <bool B> struct mutable_ {};
template <unsigned D> struct arity { static constexpr unsigned value = D; };
template <class C> struct compare { typedef C type; };

template <class T, class Mut, class Arity, class Cmp>
class d_ary_heap {
  typedef typename Cmp::type value_compare;
  static constexpr std::size_t D = Arity::value;
  typedef std::list<std::pair<T, std::size_t>> object_list;
  typedef typename object_list::iterator node;

 public:
  struct handle_type {
    node it;
    handle_type() : it() {}
    explicit handle_type(node n) : it(n) {}
  };

  bool empty() const { return q_.empty(); }
  std::size_t size() const { return q_.size(); }
  void clear() { q_.clear(); objects_.clear(); }
  const T &top() const { return q_.front()->first; }

  handle_type push(const T &v) {
    objects_.push_front(std::make_pair(v, std::size_t(0)));
    node n = objects_.begin();
    q_.push_back(n);
    n->second = q_.size() - 1;
    siftup(q_.size() - 1);
    return handle_type(n);
  }

  void pop() {
    node gone = q_.front();
    std::swap(q_.front(), q_.back());
    q_.pop_back();
    objects_.erase(gone);
    if (q_.empty()) return;
    q_[0]->second = 0;
    siftdown(0);
  }

  void increase(handle_type h) { siftup(h.it->second); }

 private:
  bool less(node a, node b) const { return cmp_(a->first, b->first); }

  void swap_nodes(std::size_t a, std::size_t b) {
    std::swap(q_[a], q_[b]);
    q_[a]->second = a;
    q_[b]->second = b;
  }

  void siftup(std::size_t index) {
    while (index != 0) {
      std::size_t parent = (index - 1) / D;
      if (less(q_[parent], q_[index])) {
        swap_nodes(parent, index);
        index = parent;
      } else
        return;
    }
  }

  void siftdown(std::size_t index) {
    while (index * D + 1 < q_.size()) {
      std::size_t first = index * D + 1;
      std::size_t last = std::min(first + D, q_.size());
      auto top_child = std::max_element(q_.begin() + first, q_.begin() + last,
                                        [this](node a, node b) { return less(a, b); });
      std::size_t c = std::size_t(top_child - q_.begin());
      if (!less(q_[c], q_[index])) {
        swap_nodes(c, index);
        index = c;
      } else
        return;
    }
  }

  object_list objects_;
  std::vector<node> q_;
  value_compare cmp_;
};

}}  // namespace boost::heap
