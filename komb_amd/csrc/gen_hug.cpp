// gen_hug.cpp -- synthetic power-law "hybrid unitig graph" workload (host only).
//
// KOMB builds its graph as a union of small cliques: every read (pair) links
// all unitigs it maps to (src/graph.cpp:310-352 expands each clique into all
// i<j pairs).  The generator mirrors that: n_cliques cliques of size
// min(1+Geom(0.45), 6) whose members are drawn i.i.d. from a power-law
// popularity w_i ~ (i+1)^(-1/(alpha-1)), then scattered over the id space by a
// seeded bijection.  Loops / duplicates are left in, exactly like the raw
// `edges` vector the reference hands to igraph_simplify (src/graph.cpp:438).
//
// Counter-based RNG (splitmix64 keyed by seed and clique index), so the output
// does not depend on the number of OpenMP threads.
#include "komb_accel.h"

#include <cmath>
#include <cstdint>
#include <vector>
#include <omp.h>

namespace {

inline uint64_t splitmix64(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t mix(uint64_t x)
{
    uint64_t s = x;
    return splitmix64(s);
}
inline double u01(uint64_t &s) { return (double)(splitmix64(s) >> 11) * 0x1.0p-53; }

// seeded bijection on [0,n): 4-round Feistel on ceil(log2 n) bits + cycle walking
struct Perm {
    uint64_t n, seed; int lb, rb; uint64_t lmask, rmask;
    Perm(uint64_t n_, uint64_t seed_) : n(n_), seed(seed_)
    {
        int bits = 1;
        while ((1ull << bits) < n) ++bits;
        if (bits < 2) bits = 2;
        lb = bits / 2; rb = bits - lb;
        lmask = (1ull << lb) - 1; rmask = (1ull << rb) - 1;
    }
    uint64_t once(uint64_t x) const
    {
        uint64_t l = x >> rb, r = x & rmask;           // l: lb bits, r: rb bits
        for (int round = 0; round < 4; ++round) {
            if ((round & 1) == 0) l = (l ^ mix(r + seed * 0x9E37 + round)) & lmask;
            else                  r = (r ^ mix(l + seed * 0x79B9 + round)) & rmask;
        }
        return (l << rb) | r;
    }
    uint64_t operator()(uint64_t x) const
    {
        do { x = once(x); } while (x >= n);
        return x;
    }
};

inline int clique_size(uint64_t &s)
{
    int k = 2;                                        // 1 + Geom(0.45), Geom >= 1
    while (k < 6 && u01(s) >= 0.45) ++k;
    return k;
}

} // namespace

extern "C" int64_t komb_gen_hug_edges(int64_t nv, int64_t n_cliques, double alpha,
                                      uint64_t seed, int64_t *uv)
{
    if (nv < 2 || n_cliques < 0 || !(alpha > 1.0)) return -1;
    const double gamma = 1.0 / (alpha - 1.0);
    const double om = 1.0 - gamma;                    // alpha > 2  =>  0 < om < 1
    const double top = (om != 0.0) ? std::pow((double)nv + 1.0, om) - 1.0 : std::log((double)nv + 1.0);
    const Perm perm((uint64_t)nv, seed);

    // pass 1: pairs per clique (size depends only on the clique's own stream)
    std::vector<int64_t> off((size_t)n_cliques + 1, 0);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_cliques; ++r) {
        uint64_t s = mix(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(r + 1)));
        int k = clique_size(s);
        off[(size_t)r + 1] = (int64_t)k * (k - 1) / 2;
    }
    for (int64_t r = 0; r < n_cliques; ++r) off[(size_t)r + 1] += off[(size_t)r];
    const int64_t n_raw = off[(size_t)n_cliques];
    if (!uv) return n_raw;

#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_cliques; ++r) {
        uint64_t s = mix(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(r + 1)));
        int k = clique_size(s);
        int64_t mem[6];
        for (int i = 0; i < k; ++i) {
            double u = u01(s);
            double x = (om != 0.0) ? std::pow(1.0 + u * top, 1.0 / om) : std::exp(u * top);
            int64_t rank = (int64_t)x - 1;
            if (rank < 0) rank = 0;
            if (rank >= nv) rank = nv - 1;
            mem[i] = (int64_t)perm((uint64_t)rank);
        }
        int64_t o = off[(size_t)r];
        for (int i = 0; i < k; ++i)
            for (int j = i + 1; j < k; ++j) { uv[2 * o] = mem[i]; uv[2 * o + 1] = mem[j]; ++o; }
    }
    return n_raw;
}
