// Host-only: the iteration order of std::unordered_map<size_t, T> after inserting n DISTINCT keys
// in a given order -- the order the reference emits subsampled points in
// (geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:44-47).
//
// libstdc++'s hashtable (bits/hashtable.h, GCC 11) keeps ONE singly linked list of all nodes plus,
// per bucket, a pointer to the node BEFORE the bucket's first node:
//   _M_insert_bucket_begin  (:1888-1912)  non-empty bucket: link after the bucket's before-node;
//                                         empty bucket: new global head, and the bucket that used to
//                                         own the head now has the new node as its before-node
//   _M_rehash_aux(unique)   (:2380-2411)  relink every node, in list order, by the same two rules
//   _M_insert_unique_node   (:2010-2031)  ask _Prime_rehash_policy::_M_need_rehash first
// std::hash<size_t> is the identity and not cached, so bucket(node) = key % bucket_count.
// This file replays exactly those steps on flat arrays (no node allocations) and takes the bucket
// counts from the REAL policy object of the libstdc++ this library is built against, so the order
// is the container's by construction; tests/test_hash_order.py compares it with the container.
#include <unordered_map>  // brings std::__detail::_Prime_rehash_policy
#include <vector>

#include "common.hpp"

namespace gr {

void unordered_map_order(const uint64_t* keys, int64_t n, int32_t base, int32_t* perm_out) {
  constexpr int32_t NIL = -1, HEAD = -2, EMPTY = -3;
  std::__detail::_Prime_rehash_policy policy;
  std::size_t bkt_count = 1;
  std::vector<int32_t> bucket(1, EMPTY);  // "before" node of each bucket (HEAD = the list head slot)
  std::vector<int32_t> next(static_cast<size_t>(n > 0 ? n : 1));
  int32_t head = NIL;
  constexpr int64_t AHEAD = 12;  // the bucket array is hit at random: ask for the line a few inserts early
  for (int64_t i = 0; i < n; ++i) {
    if (i + AHEAD < n) __builtin_prefetch(&bucket[static_cast<std::size_t>(keys[i + AHEAD]) % bkt_count], 1, 1);
    const auto need = policy._M_need_rehash(bkt_count, static_cast<std::size_t>(i), 1);
    if (need.first) {
      const std::size_t nb = need.second;
      std::vector<int32_t> nbucket(nb, EMPTY);
      int32_t p = head;
      head = NIL;
      std::size_t bbegin_bkt = 0;
      while (p != NIL) {
        const int32_t nx = next[p];
        const std::size_t b = static_cast<std::size_t>(keys[p]) % nb;
        if (nbucket[b] == EMPTY) {
          next[p] = head;
          head = p;
          nbucket[b] = HEAD;
          if (next[p] != NIL) nbucket[bbegin_bkt] = p;
          bbegin_bkt = b;
        } else if (nbucket[b] == HEAD) {
          next[p] = head;
          head = p;
        } else {
          next[p] = next[nbucket[b]];
          next[nbucket[b]] = p;
        }
        p = nx;
      }
      bucket.swap(nbucket);
      bkt_count = nb;
    }
    const std::size_t b = static_cast<std::size_t>(keys[i]) % bkt_count;
    const int32_t node = static_cast<int32_t>(i);
    if (bucket[b] == EMPTY) {
      next[node] = head;
      head = node;
      if (next[node] != NIL) bucket[static_cast<std::size_t>(keys[next[node]]) % bkt_count] = node;
      bucket[b] = HEAD;
    } else if (bucket[b] == HEAD) {
      next[node] = head;
      head = node;
    } else {
      next[node] = next[bucket[b]];
      next[bucket[b]] = node;
    }
  }
  int64_t j = 0;
  for (int32_t p = head; p != NIL; p = next[p]) perm_out[j++] = base + p;
}

}  // namespace gr

// Host-only helper exposed for tests (no GPU involved): perm[j] = index (into keys) of the j-th
// element an std::unordered_map<size_t,...> would iterate after inserting keys[0..n) in order.
extern "C" int gr_host_unordered_map_order(const uint64_t* h_keys, int64_t n, int32_t* h_perm) {
  GR_REQUIRE(n >= 0 && n < (1ll << 31) - 1 && (n == 0 || (h_keys && h_perm)), "bad arguments");
  gr::unordered_map_order(h_keys, n, 0, h_perm);
  return GR_OK;
}
