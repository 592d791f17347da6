// TEST INFRASTRUCTURE (oracle): libstdc++'s std::nth_element written out.
//
// cv::KeyPointsFilter::retainBest (OpenCV 3.2 modules/features2d/src/keypoint.cpp) is
//     std::nth_element(begin, begin + n, end, KeypointResponseGreater());  ... std::partition(begin + n, end, ...); resize(new_end)
// and ORBextractor.cpp:692-694 / :708-709 cut the vector to n right afterwards, so what the reference keeps - WHICH of the tied
// key points and in WHAT order they are pushed into the level's list - is the first n entries of the permutation nth_element
// leaves behind (the partition only touches [n, end)).  That permutation is not specified by the C++ standard, but it is a fixed
// function of the input for a given standard library, and the reference is built with GCC: libstdc++'s introselect
// (bits/stl_algo.h: __introselect, __unguarded_partition_pivot, __move_median_to_first, __unguarded_partition,
// __insertion_sort, __heap_select; bits/stl_heap.h: __make_heap, __pop_heap, __adjust_heap, __push_heap - the same text from
// GCC 4.9 to 14, so also the GCC 7 of the reference's Ubuntu 18.04 CI).  This header restates it step by step on plain arrays;
// tests/cpp_stl_nth.cpp holds it to this machine's std::nth_element on random inputs with heavy ties, on every input of a
// small alphabet up to length 9, and on median-of-three killers that reach the heap-select branch.
// The HIP path (csrc/orb.hip: introselect_wave) computes the same permutation on the device; the parity tests compare the two.
#pragma once
#include <cstddef>
#include <utility>

namespace stl_nth {

template <class T, class Comp>
inline void push_heap_(T* first, ptrdiff_t hole, ptrdiff_t top, T value, Comp comp) {
    ptrdiff_t parent = (hole - 1) / 2;
    while (hole > top && comp(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

template <class T, class Comp>
inline void adjust_heap_(T* first, ptrdiff_t hole, ptrdiff_t len, T value, Comp comp) {
    const ptrdiff_t top = hole;
    ptrdiff_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (comp(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap_(first, hole, top, value, comp);
}

template <class T, class Comp>
inline void heap_select_(T* first, T* middle, T* last, Comp comp) {
    const ptrdiff_t len = middle - first;
    if (len >= 2)
        for (ptrdiff_t parent = (len - 2) / 2;; --parent) {   // __make_heap
            T v = first[parent];
            adjust_heap_(first, parent, len, v, comp);
            if (parent == 0) break;
        }
    for (T* i = middle; i < last; ++i)
        if (comp(*i, *first)) {                                // __pop_heap(first, middle, i)
            T v = *i;
            *i = *first;
            adjust_heap_(first, (ptrdiff_t)0, len, v, comp);
        }
}

template <class T, class Comp>
inline void move_median_to_first_(T* result, T* a, T* b, T* c, Comp comp) {
    if (comp(*a, *b)) {
        if (comp(*b, *c)) std::swap(*result, *b);
        else if (comp(*a, *c)) std::swap(*result, *c);
        else std::swap(*result, *a);
    } else if (comp(*a, *c)) std::swap(*result, *a);
    else if (comp(*b, *c)) std::swap(*result, *c);
    else std::swap(*result, *b);
}

template <class T, class Comp>
inline T* unguarded_partition_(T* first, T* last, T* pivot, Comp comp) {
    while (true) {
        while (comp(*first, *pivot)) ++first;
        --last;
        while (comp(*pivot, *last)) --last;
        if (!(first < last)) return first;
        std::swap(*first, *last);
        ++first;
    }
}

template <class T, class Comp>
inline void insertion_sort_(T* first, T* last, Comp comp) {
    if (first == last) return;
    for (T* i = first + 1; i != last; ++i) {
        T v = *i;
        if (comp(v, *first)) {
            for (T* p = i; p != first; --p) *p = *(p - 1);   // move_backward(first, i, i + 1)
            *first = v;
        } else {                                              // __unguarded_linear_insert
            T* hole = i;
            T* next = i - 1;
            while (comp(v, *next)) {
                *hole = *next;
                hole = next;
                --next;
            }
            *hole = v;
        }
    }
}

inline long& heap_selects() { static thread_local long n = 0; return n; }   // how often the depth limit was reached (tests)

inline int lg_(ptrdiff_t n) {   // std::__lg: floor(log2(n)), n > 0
    int k = 0;
    while (n > 1) { n >>= 1; ++k; }
    return k;
}

// std::nth_element(first, nth, last, comp).  depth_limit < 0: the library's own 2 * lg(n).
template <class T, class Comp>
inline void nth_element(T* first, T* nth, T* last, Comp comp, long depth_limit = -1) {
    if (first == last || nth == last) return;
    if (depth_limit < 0) depth_limit = 2L * lg_(last - first);
    while (last - first > 3) {
        if (depth_limit == 0) {
            ++heap_selects();
            heap_select_(first, nth + 1, last, comp);
            std::swap(*first, *nth);
            return;
        }
        --depth_limit;
        T* mid = first + (last - first) / 2;
        move_median_to_first_(first, first + 1, mid, last - 1, comp);
        T* cut = unguarded_partition_(first + 1, last, first, comp);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    insertion_sort_(first, last, comp);
}

}  // namespace stl_nth
