// sam_port.cpp -- TEST INFRASTRUCTURE ONLY (a checker and a timed CPU baseline; never linked into the
// product).  CPU restatement of how the reference turns two SAM files into its edge list, with the
// reference's own choice of data structures, so that komb2's sort-based host pipeline (row N1 of the
// hot-path table) can be checked and timed against it at sizes the Python restatement cannot reach:
//
//   readSAM        src/graph.cpp:166-257  whole file in memory; OpenMP static byte chunks; a thread parses
//                                         the lines between two newlines of ITS chunk (so the line
//                                         straddling a chunk boundary is dropped); '@' lines skipped;
//                                         field 0 = read, field 2 = unitig, '*' skipped; key =
//                                         read.substr(1, read.find('/')); per-thread
//                                         unordered_map<string, unordered_set<string>>, reduced serially,
//                                         vertex ids handed out in reduction order
//   getEdgeInfo    src/graph.cpp:259-285  mate maps merged per read key
//   generateGraph  src/graph.cpp:287-393  every read's unitig set expands to all pairs; per-thread
//                                         unordered_set of seen (vid,vid) pairs; per-thread edge vectors
//                                         concatenated
//
// Usage: sam_port <threads> <r1.sam> <r2.sam> <out_pairs.txt>
// Writes one "nameA\tnameB" line per emitted pair (raw, like edgelist.txt but by NAME, because vertex
// numbers depend on hash iteration order) and prints the stage times.  Parity pinned by
// tests/test_komb2_host.py against tests/samgraph.py and komb2 on the same files.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <omp.h>

using umapset = std::unordered_map<std::string, std::unordered_set<std::string>>;
using clk = std::chrono::steady_clock;
static double since(clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); }

struct PairHash {
    size_t operator()(const std::pair<long, long> &p) const { return std::hash<long>()(p.first) ^ (std::hash<long>()(p.second) << 1); }
};

static void read_sam(const char *path, int threads, umapset &umap, std::unordered_map<std::string, long> &vid_of, long &next_vid)
{
    FILE *f = fopen(path, "r");
    if (!f) { fprintf(stderr, "File %s could not be opened. Exiting...\n", path); exit(EXIT_FAILURE); }
    fseek(f, 0, SEEK_END);
    const uint64_t size = (uint64_t)ftell(f);
    rewind(f);
    char *buf = (char *)malloc(size + 1);
    if (!buf || fread(buf, 1, size, f) != size) { fprintf(stderr, "read error on %s\n", path); exit(EXIT_FAILURE); }
    buf[size] = 0;
    fclose(f);

    std::vector<umapset> local((size_t)threads);
    std::vector<std::vector<uint64_t>> position((size_t)threads);
    position[0].push_back(0);
#pragma omp parallel num_threads(threads)
    {
        const int t = omp_get_thread_num();
#pragma omp for schedule(static)
        for (uint64_t i = 0; i < size; ++i)
            if (buf[i] == '\n' || buf[i] == '\0') position[(size_t)t].push_back(i);
        const auto &pos = position[(size_t)t];
        for (size_t i = 1; i < pos.size(); ++i) {
            uint64_t start = pos[i - 1] + 1;
            if (start == 1) start = 0;
            char *line = buf + start;
            if (line[0] == '@') continue;
            buf[pos[i]] = 0;                              // (the reference lets strtok_r run to the next tab; same fields)
            char *save = nullptr;
            char *tok = strtok_r(line, "\t", &save);
            if (!tok) continue;
            std::string read(tok);
            int count = 0;
            while (tok && count < 2) { tok = strtok_r(nullptr, "\t", &save); ++count; }
            if (!tok) continue;
            std::string unitig(tok);
            if (unitig != "*") local[(size_t)t][read.substr(1, read.find('/'))].insert(unitig);
        }
    }
    for (int t = 0; t < threads; ++t)
        for (auto &kv : local[(size_t)t])
            for (auto &u : kv.second) {
                umap[kv.first].insert(u);
                if (vid_of.find(u) == vid_of.end()) vid_of[u] = next_vid++;
            }
    free(buf);
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: sam_port <threads> <r1.sam> <r2.sam> <out_pairs.txt>\n"); return 2; }
    const int threads = std::max(1, atoi(argv[1]));
    umapset umap1, umap2;
    std::unordered_map<std::string, long> vid_of;
    long next_vid = 0;
    auto t0 = clk::now();
    read_sam(argv[2], threads, umap1, vid_of, next_vid);
    read_sam(argv[3], threads, umap2, vid_of, next_vid);
    const double t_sam = since(t0);

    t0 = clk::now();
    for (auto &kv : umap1) {                               // getEdgeInfo
        auto it2 = umap2.find(kv.first);
        if (it2 != umap2.end()) { kv.second.merge(it2->second); umap2.erase(it2); }
    }
    umap1.insert(umap2.begin(), umap2.end());
    umapset().swap(umap2);
    const double t_merge = since(t0);

    t0 = clk::now();
    std::vector<std::vector<std::string>> cliques;          // generateGraph
    cliques.reserve(umap1.size());
    for (auto &kv : umap1) cliques.emplace_back(kv.second.begin(), kv.second.end());
    umapset().swap(umap1);
    std::vector<std::string> name((size_t)next_vid);
    for (auto &kv : vid_of) name[(size_t)kv.second] = kv.first;
    std::vector<std::vector<long>> edges((size_t)threads);
    std::vector<std::unordered_set<std::pair<long, long>, PairHash>> seen((size_t)threads);
#pragma omp parallel num_threads(threads)
    {
        const int t = omp_get_thread_num();
#pragma omp for schedule(static)
        for (size_t c = 0; c < cliques.size(); ++c)
            for (size_t i = 0; i < cliques[c].size(); ++i)
                for (size_t j = i + 1; j < cliques[c].size(); ++j) {
                    const std::pair<long, long> e(vid_of[cliques[c][i]], vid_of[cliques[c][j]]);
                    if (seen[(size_t)t].insert(e).second) { edges[(size_t)t].push_back(e.first); edges[(size_t)t].push_back(e.second); }
                }
    }
    const double t_expand = since(t0);

    t0 = clk::now();
    FILE *out = fopen(argv[4], "w");
    if (!out) { fprintf(stderr, "cannot write %s\n", argv[4]); return 1; }
    uint64_t n_pairs = 0;
    for (auto &ev : edges)
        for (size_t i = 0; i + 1 < ev.size(); i += 2, ++n_pairs)
            fprintf(out, "%s\t%s\n", name[(size_t)ev[i]].c_str(), name[(size_t)ev[i + 1]].c_str());
    fclose(out);
    printf("threads %d vertices %ld cliques %zu raw_pairs %llu\n", threads, next_vid, cliques.size(), (unsigned long long)n_pairs);
    printf("seconds: readSAM %.3f getEdgeInfo %.3f generateGraph %.3f write %.3f\n", t_sam, t_merge, t_expand, since(t0));
    return 0;
}
