// Host side shared by tools/chol64_solve.hip (the 64-wide prototype) and tools/chol32_emu.cpp (the shipped 32-wide kernel under the
// wave emulator): the task plan over a tile pattern in the format of csrc/ba.hip's solve_plan_build, and a random SPD system with
// that pattern.  Experiments / test harness only - nothing in the product includes this.
#pragma once
#include <algorithm>
#include <cstdio>
#include <vector>

struct Plan {
    std::vector<int4> tasks;
    std::vector<int> deps;
};

// The plan over a tile pattern.  P[i][j] (i >= j; i < nt tile rows, j < nbc block columns): tile (i, j) of the lower triangle is
// non-zero; the tile row of the rhs row is non-zero everywhere.  Symbolic fill on tiles, then the tasks by block column -
// dependencies point to earlier tasks only - and the x tasks last.
inline Plan plan_from_pattern(std::vector<std::vector<char>> P, int nt, int nbc, int* chain_len) {
    for (int m = 0; m < nbc; ++m)                 // fill: L(i, j) when L(i, m) and L(j, m) for some m < j
        for (int j = m + 1; j < nbc; ++j)
            if (P[j][m])
                for (int i = j; i < nt; ++i)
                    if (P[i][m]) P[i][j] = 1;
    // R = L^-T by tile rows: R(r, r) always; R(r, j), j > r, when R(r, m) and L(j, m) for some r <= m < j
    std::vector<std::vector<char>> R(nbc, std::vector<char>(nbc, 0));
    for (int r = 0; r < nbc; ++r) {
        R[r][r] = 1;
        for (int j = r + 1; j < nbc; ++j)
            for (int m = r; m < j; ++m)
                if (R[r][m] && P[j][m]) { R[r][j] = 1; break; }
    }
    Plan p;
    auto add = [&](int kind, int i, int j, const std::vector<int>& d) {
        int4 t;
        t.x = i | (kind << 16); t.y = j; t.z = (int)p.deps.size();
        p.deps.insert(p.deps.end(), d.begin(), d.end());
        t.w = (int)p.deps.size();
        p.tasks.push_back(t);
    };
    std::vector<int> depth(nbc, 1);               // block columns on the longest dependency chain ending in column j
    for (int j = 0; j < nbc; ++j) {
        for (int m = 0; m < j; ++m)
            if (P[j][m]) depth[j] = std::max(depth[j], depth[m] + 1);
        for (int i = j; i < nt; ++i) {            // the diagonal task, then the L tiles of the column
            if (i > j && !P[i][j]) continue;
            std::vector<int> d;
            for (int m = 0; m < j; ++m)
                if (P[j][m]) d.push_back(m | (i != j && P[i][m] ? 1 << 15 : 0));      // (the diagonal task's T is the identity: no T products)
            add(0, i, j, d);
        }
        for (int r = 0; r < j; ++r) {
            if (!R[r][j]) continue;
            std::vector<int> d;
            for (int m = 0; m < j; ++m)
                if (P[j][m]) d.push_back(m | (m >= r && R[r][m] ? 1 << 15 : 0));
            add(1, r, j, d);
        }
    }
    for (int r = 0; r < nbc; ++r) {
        std::vector<int> d;
        for (int j = r; j < nbc; ++j)
            if (R[r][j]) d.push_back(j);
        add(2, r, 0, d);
    }
    *chain_len = *std::max_element(depth.begin(), depth.end());
    return p;
}

// dense, or (arc > 0) two uncoupled arcs of `arc` tiles and a separator of `sep` tiles that couples to both - the nested-dissection
// shape the product's plan gives a window.  A: (n + 1) x (n + 1) augmented in ld x ld, row n = the right-hand side; SPD = G G^T + a
// diagonal, G with the pattern's block structure (arc rows use their own factor columns only: A(arc 0, arc 1) = 0 exactly)
struct CholSystem {
    int n, ld, nt, nbc, chain;
    std::vector<double> A, b;
    Plan plan;
};
inline CholSystem chol_system(int NB, int n_dense, int arc, int sep) {
    CholSystem S;
    const bool nd = arc > 0;
    const int n = S.n = nd ? NB * (2 * arc + sep) : n_dense;
    const int ld = S.ld = ((n + 1 + NB - 1) / NB) * NB, nt = S.nt = ld / NB, nbc = S.nbc = (n + NB - 1) / NB;
    std::vector<std::vector<char>> P(nt, std::vector<char>(nbc, 1));
    if (nd)
        for (int i = 0; i < nbc; ++i)
            for (int j = 0; j < nbc; ++j) {
                const int gi = i < arc ? 0 : i < 2 * arc ? 1 : 2, gj = j < arc ? 0 : j < 2 * arc ? 1 : 2;
                P[i][j] = gi == gj || gi == 2 || gj == 2;
            }
    S.A.assign((size_t)ld * ld, 0.0);
    S.b.assign(n, 0.0);
    std::vector<double> G((size_t)n * n, 0.0);
    unsigned long long s = 88172645463325252ull;
    auto unit = [&]() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return (double)((s * 2685821657736338717ull) >> 11) * (1.0 / 9007199254740992.0); };
    auto group = [&](int r) { return !nd ? 2 : r < NB * arc ? 0 : r < 2 * NB * arc ? 1 : 2; };
    for (int r = 0; r < n; ++r)
        for (int k = 0; k < n; ++k)
            if (group(r) == 2 || group(k) == group(r)) G[(size_t)r * n + k] = unit() - 0.5;
    for (int r = 0; r < n; ++r) {
        for (int c = 0; c <= r; ++c) {
            double a = r == c ? 0.5 * n : 0.0;
            for (int k = 0; k < n; ++k) a += G[(size_t)r * n + k] * G[(size_t)c * n + k];
            S.A[(size_t)r * ld + c] = a; S.A[(size_t)c * ld + r] = a;
        }
        S.b[r] = 10.0 * (unit() - 0.5);
        S.A[(size_t)n * ld + r] = S.b[r];
    }
    if (nd)
        for (int r = NB * arc; r < 2 * NB * arc; ++r)
            for (int c = 0; c < NB * arc; ++c)
                if (S.A[(size_t)r * ld + c] != 0.0) { std::printf("the arcs are coupled?\n"); std::abort(); }
    S.plan = plan_from_pattern(P, nt, nbc, &S.chain);
    return S;
}
// a system and ITS PLAN from a file: int32 {n, ld, tasks, dependency entries}, the tasks (4 int32 each), the dependency entries, A (ld x ld
// doubles, the right-hand side in row n) - what tests/test_wave_protocol_probe.py writes from the product's own plan code
inline bool chol_system_from_file(int NB, const char* path, CholSystem& S) {
    FILE* f = std::fopen(path, "rb");
    int hdr[4];
    if (!f || std::fread(hdr, 4, 4, f) != 4) return false;
    S.n = hdr[0]; S.ld = hdr[1]; S.nt = S.ld / NB; S.nbc = (S.n + NB - 1) / NB; S.chain = 0;
    S.plan.tasks.resize(hdr[2]); S.plan.deps.resize(hdr[3]); S.A.resize((size_t)S.ld * S.ld); S.b.resize(S.n);
    bool ok = std::fread(S.plan.tasks.data(), sizeof(int4), hdr[2], f) == (size_t)hdr[2];
    ok = ok && std::fread(S.plan.deps.data(), 4, hdr[3], f) == (size_t)hdr[3];
    ok = ok && std::fread(S.A.data(), 8, S.A.size(), f) == S.A.size();
    std::fclose(f);
    for (int c = 0; ok && c < S.n; ++c) S.b[c] = S.A[(size_t)S.n * S.ld + c];
    return ok;
}
// |A x - b|_inf / |b|_inf
inline double chol_residual(const CholSystem& S, const std::vector<double>& x) {
    double worst = 0.0, scale = 0.0;
    for (int r = 0; r < S.n; ++r) {
        double a = -S.b[r];
        for (int c = 0; c < S.n; ++c) a += S.A[(size_t)r * S.ld + c] * x[c];
        if (!(std::fabs(a) <= worst)) worst = std::fabs(a);     // (a NaN stays)
        scale = std::fmax(scale, std::fabs(S.b[r]));
    }
    return worst / scale;
}
