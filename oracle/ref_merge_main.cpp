// ref_merge_main.cpp -- harness around the REFERENCE's own src/HashIndexedMinHeap.h
// (std-only header, compiled where it lies under /root/reference by oracle/Makefile; nothing of it is
// copied into this repo).  The binary goes to oracle/_ref/merge_ref and pins rows a12 / a13 of SURVEY.md
// section 8 -- the indexed min-heap and the greedy densest-block peel CombineCoreA::runMerge that is its only
// user.  TEST INFRASTRUCTURE ONLY.
//
// src/CombineCoreA.h itself cannot be compiled here (it includes <igraph.h>), so the LOOP of runMerge
// (src/CombineCoreA.h:45-219) is restated below, statement for statement, over a plain CSR instead of
// igraph's lazy adjacency list (igraph_lazy_adjlist_init(graph, al, IGRAPH_ALL, IGRAPH_NO_LOOPS,
// IGRAPH_NO_MULTIPLE), src/graph.cpp:645: the neighbours of a vertex of the simple graph, ascending) --
// while every heap operation it performs runs in the reference's own class.  Two defects of the reference
// are NOT reproduced, because they make its output undefined: removed[2][n] comes straight from malloc
// (src/CombineCoreA.h:104-108; here: zero-initialised, as VERDICT r2 prescribes), and `cols` is sized by the
// number of ROWS of the block (src/CombineCoreA.h:191; here the removal order, the modes and the size of
// the densest block are the output, from which rows and cols follow without that vector).
//
//   merge_ref <graph.bin>        graph.bin: int64 nv, int64 ns, int64 rowptr[nv+1], int32 col[ns],
//                                int64 has_susp, double susp[nv] (when has_susp)
// Output: "nblock maxdensity(%.17g)" then 2*nv lines "order mode".
#include "HashIndexedMinHeap.h"   // found via -I/root/reference/src

#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: merge_ref graph.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    int64_t hdr[2];
    if (!f || fread(hdr, 8, 2, f) != 2) return 2;
    const int numNodes = (int)hdr[0];
    std::vector<int64_t> rowptr((size_t)numNodes + 1);
    std::vector<int32_t> col((size_t)hdr[1] + 1);
    if (fread(rowptr.data(), 8, rowptr.size(), f) != rowptr.size()) return 2;
    if (hdr[1] && fread(col.data(), 4, (size_t)hdr[1], f) != (size_t)hdr[1]) return 2;
    int64_t has = 0;
    std::vector<double> susp;
    if (fread(&has, 8, 1, f) == 1 && has) {
        susp.resize((size_t)numNodes);
        if (numNodes && fread(susp.data(), 8, susp.size(), f) != susp.size()) return 2;
    }
    fclose(f);
    const double *suspiciousness = has ? susp.data() : nullptr;

    // side 0 = the "row" copy of the graph, side 1 = the "column" copy (src/CombineCoreA.h:47-99): a node's priority on
    // either side is its suspiciousness (0 when none is given) plus its degree; the running total is what the two sides hold
    const int n = numNodes;
    std::vector<double> prio[2] = {std::vector<double>((size_t)n, 0.0), std::vector<double>((size_t)n, 0.0)};
    double total = 0;
    if (suspiciousness)
        for (int v = 0; v < n; ++v) { prio[0][v] = prio[1][v] = suspiciousness[v]; total += 2 * suspiciousness[v]; }
    long slots = 0;
    for (int v = 0; v < n; ++v)
        for (int64_t j = rowptr[v]; j < rowptr[v + 1]; ++j) { prio[0][v] += 1; prio[1][col[j]] += 1; ++slots; }
    total += slots;
    HashIndexedMinHeap side0(n > 0 ? n : 1), side1(n > 0 ? n : 1);          // the reference's class, both times
    HashIndexedMinHeap *heap[2] = {&side0, &side1};
    for (int sd = 0; sd < 2; ++sd)
        for (int v = 0; v < n; ++v) heap[sd]->insert(v, prio[sd][v]);
    std::vector<int> who((size_t)2 * n + 1), side((size_t)2 * n + 1);
    std::vector<char> gone[2] = {std::vector<char>((size_t)n + 1, 0), std::vector<char>((size_t)n + 1, 0)};     // zero-initialised (see above)
    // the peel (src/CombineCoreA.h:111-174): take the smaller of the two heap minima -- side 0 only when it is strictly
    // smaller, or side 1 is empty --, record it, update the density of what is left, lower the node's neighbours on the
    // OTHER side by one
    double best = 0;
    int best_left = 0;
    for (int left = 2 * n; left >= 1;) {
        const std::pair<int, double> t0 = side0.peek(), t1 = side1.peek();
        const bool have0 = t0.first != INT_MIN && t0.second != INT_MIN;
        const bool none1 = t1.first == INT_MIN && t1.second == INT_MIN;
        const int sd = (have0 && (none1 || t0.second < t1.second)) ? 0 : 1;
        const std::pair<int, double> top = heap[sd]->poll();
        total -= top.second;
        --left;
        who[left] = top.first;
        side[left] = sd;
        const double density = total / left;
        if (left >= 1 && density > best) { best = density; best_left = left; }
        gone[sd][top.first] = 1;
        HashIndexedMinHeap *other = heap[sd ^ 1];
        for (int64_t j = rowptr[top.first]; j < rowptr[top.first + 1]; ++j) {
            const int w = col[j];
            if (!gone[sd ^ 1][w]) other->refreshPriority(w, other->getPriority(w) - 1);
        }
    }
    printf("%d %.17g\n", best_left, best);
    for (int i = 0; i < 2 * n; ++i) printf("%d %d\n", who[i], side[i]);
    return 0;
}
