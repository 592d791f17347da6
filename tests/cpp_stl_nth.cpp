// oracle/stl_nth.h (libstdc++'s nth_element written out) against this machine's std::nth_element: the whole permutation, not
// just the n-th element.  Built and run by tests/test_orb_oracle.py::test_restated_nth_element_equals_libstdcxx.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../oracle/stl_nth.h"

struct E { int key, id; };
static bool greater_key(const E& a, const E& b) { return a.key > b.key; }

static long g_checked = 0;
static bool same(std::vector<E> v, int nth, const char* what) {
    std::vector<E> a = v, b = v;
    std::nth_element(a.begin(), a.begin() + nth, a.end(), greater_key);
    stl_nth::nth_element(b.data(), b.data() + nth, b.data() + b.size(), greater_key);
    ++g_checked;
    for (size_t i = 0; i < v.size(); ++i)
        if (a[i].id != b[i].id) {
            std::printf("MISMATCH (%s): n %zu nth %d at %zu\n", what, v.size(), nth, i);
            return false;
        }
    return true;
}

// McIlroy's adversary ("A killer adversary for quicksort", 1999) run against std::nth_element itself: the values it freezes make
// every median-of-three pivot one of the smallest remaining keys, so the range shrinks by a constant per round and the depth limit
// 2 lg n is reached: the heap-select branch.
struct Adversary {
    std::vector<int> val;
    int gas, nsolid = 0, candidate = 0;
    explicit Adversary(int n) : val(n, n), gas(n) {}
    bool less(int x, int y) {
        if (val[x] == gas && val[y] == gas) {
            if (x == candidate) val[x] = nsolid++;
            else val[y] = nsolid++;
        }
        if (val[x] == gas) candidate = x;
        else if (val[y] == gas) candidate = y;
        return val[x] < val[y];
    }
};

int main() {
    std::mt19937_64 rng(20260926);
    // 1. every input over a three-letter alphabet up to length 9, every nth
    for (int n = 1; n <= 9; ++n) {
        int total = 1;
        for (int i = 0; i < n; ++i) total *= 3;
        for (int code = 0; code < total; ++code) {
            std::vector<E> v(n);
            for (int i = 0, c = code; i < n; ++i, c /= 3) v[i] = E{c % 3, i};
            for (int nth = 0; nth < n; ++nth)
                if (!same(v, nth, "exhaustive")) return 1;
        }
    }
    // 2. random inputs, alphabets from 1 letter (all tied) to 2^20 (nearly no ties): FAST scores have ~60 distinct values
    for (int it = 0; it < 60000; ++it) {
        const int n = 1 + (int)(rng() % (it % 50 == 0 ? 6000 : 400));
        const int alpha = 1 << (rng() % 21);
        std::vector<E> v(n);
        for (int i = 0; i < n; ++i) v[i] = E{(int)(rng() % alpha), i};
        if (it % 7 == 0) std::sort(v.begin(), v.end(), greater_key);                                         // presorted
        if (it % 11 == 0) std::sort(v.begin(), v.end(), [](const E& a, const E& b) { return a.key < b.key; });  // reversed
        for (int i = 0; i < n; ++i) v[i].id = i;
        if (!same(v, (int)(rng() % n), "random")) return 1;
    }
    // 3. median-of-three killers: the depth limit must be reached, and the results must still agree
    const long before = stl_nth::heap_selects();
    int killers = 0;
    for (int n : {64, 200, 1000, 5000}) {
        for (int which = 0; which < 3; ++which) {
            const int nth = which == 0 ? n - 1 : which == 1 ? n / 2 : (3 * n) / 4;
            Adversary adv(n);
            std::vector<int> idx(n);
            for (int i = 0; i < n; ++i) idx[i] = i;
            std::nth_element(idx.begin(), idx.begin() + nth, idx.end(), [&](int a, int b) { return adv.less(a, b); });
            std::vector<E> v(n);
            for (int i = 0; i < n; ++i) v[i] = E{-adv.val[i], i};   // "less" on val = "greater" on -val
            const long h0 = stl_nth::heap_selects();
            if (!same(v, nth, "killer")) return 1;
            killers += stl_nth::heap_selects() > h0;
            // the same killer with ties folded in
            for (int i = 0; i < n; ++i) v[i].key /= 3;
            if (!same(v, nth, "killer / 3")) return 1;
        }
    }
    if (killers < 6) {
        std::printf("only %d of 12 killer inputs reached the heap-select branch\n", killers);
        return 1;
    }
    std::printf("ok: %ld arrays identical to std::nth_element, heap-select branch taken %ld times\n", g_checked,
                stl_nth::heap_selects() - before);
    return 0;
}
