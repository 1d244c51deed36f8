// komb2.cpp -- drop-in `komb2` for KOMB.py (reference CLI: src/komb2.cpp:16-150).
//
// Same flags, same stdout lines, same files in -o (edgelist.txt, kcore.tsv,
// CoreA_anomaly.txt, optionally truss_unitigs.fasta), same exit codes, so
// KOMB.py's RunKOMB (KOMB.py:435-462) drives it unchanged.  The host builds the
// unitig graph from the two SAM files the way the reference does
// (src/graph.cpp:166-393) and hands the raw vertex pairs to the MI355X library
// through the C ABI of include/komb_accel.h; all decomposition arithmetic
// (simplify, degree, coreness, trussness, CoreA ranks) runs in HIP kernels.
// There is no CPU fallback: without a GPU the program exits non-zero.
//
// Environment knobs (KOMB.py passes a fixed argv, so extras are env vars):
//   KOMB_V1_OUTPUTS=1   re-enable combineFile and splitAnomalousUnitigs (commented out at
//                       src/komb2.cpp:126,139): combined.fasta, top_/low_scoring_anomalous_unitigs.txt,
//                       reference quirks kept; "fixed" = per-vertex decision and unitig names
//   KOMB_V1_ONLY=1      only those two stages, from the kcore.tsv + CoreA_anomaly.txt already in -o
//   KOMB_TRUSS=1        re-enable the runTruss stage the reference has commented
//                       out at src/graph.cpp:478 (writes truss_unitigs.fasta)
//   KOMB_STRICT_SAM=1   parse every SAM line (the reference drops the line that
//                       straddles each OpenMP byte-chunk boundary, see readSAM)
//   KOMB_DEVICE=<n>     HIP device ordinal (default 0)
//   KOMB_HOST_TIMES=1   stage times of the SAM -> edge list pipeline on stderr
//   KOMB_STOP_AFTER_EDGES=1  write edgelist.txt + vertex_names.txt and stop
//                       before touching the device (host-logic tests)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <ctime>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <omp.h>
#include <parallel/algorithm>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "komb_accel.h"

namespace {

using clk = std::chrono::steady_clock;
double since(clk::time_point t0) { return std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count() / 1000000.0; }

// The HIP context is created on a second thread beside the SAM parse.  A fatal exit on the main thread first waits for
// that thread (and drops the context): exit() runs the static destructors, and the HIP runtime must not be torn down
// under a thread that is still initialising it.
std::future<komb_ctx *> *g_ctx_early = nullptr;
[[noreturn]] void leave(int code)
{
    if (g_ctx_early && g_ctx_early->valid()) {
        komb_ctx *c = g_ctx_early->get();
        if (c) komb_destroy(c);
    }
    exit(code);
}

[[noreturn]] void file_not_found(const std::string &path)
{
    // src/graph.cpp:58-61
    fprintf(stderr, "File %s could not be opened. Exiting...\n", path.c_str());
    leave(EXIT_FAILURE);
}

[[noreturn]] void parse_error(const char *prog, const std::string &what)
{
    // TCLAP StdOutput::failure (external/tclap/StdOutput.h:131-153): message on stderr, exit(1)
    fprintf(stderr, "PARSE ERROR: %s\n\nBrief USAGE: \n   %s  -i <string> -j <string> -u <string> [-l <int>] [-t <int>] [-o <string>] [-f] [--] [--version] [-h]\n\nFor complete USAGE and HELP type: \n   %s --help\n\n",
            what.c_str(), prog, prog);
    exit(1);
}

void usage(const char *prog)
{
    printf("\nUSAGE: \n\n   %s  -i <string> -j <string> -u <string> [-l <int>] [-t <int>] [-o <string>] [-f]\n"
           "          [--] [--version] [-h]\n\nWhere: \n\n"
           "   -i <string>,  --input <string>\n     (required)  Input SAM file [Default: alingment1.sam]\n\n"
           "   -j <string>,  --input2 <string>\n     (required)  Second input SAM file [Default: alignment2.sam]\n\n"
           "   -u <string>,  --input-unitigs <string>\n     (required)  FASTA file containing unitigs [Default: unitigs.fa]\n\n"
           "   -l <int>,  --readlen <int>\n     Read Length (can be average) [Default: 151]\n\n"
           "   -t <int>,  --threads <int>\n     Number of Threads [Default: Max]\n\n"
           "   -o <string>,  --output <string>\n     Output directory [Default: output_yyyymmdd_hhmmss]\n\n"
           "   -f,  --fulgor\n     Use Fulgor pseudoalignments instead of SAM files\n\n"
           "   --,  --ignore_rest\n     Ignores the rest of the labeled arguments following this flag.\n\n"
           "   --version\n     Displays version information and exits.\n\n"
           "   -h,  --help\n     Displays usage information and exits.\n\n\n"
           "   KOMB: Taxonomy-oblivious characterization of metagenome dynamics\n\n", prog);
}

struct Args {
    std::string input, input2, unitigs, outdir;
    int readlen = 151, threads = 1;
    bool fulgor = false;
};

Args parse_args(int argc, const char **argv)
{
    Args a;
    a.threads = omp_get_max_threads();                     // src/komb2.cpp:24
    time_t now = time(nullptr);
    tm *t = localtime(&now);                               // src/komb2.cpp:27-32: unpadded fields
    a.outdir = "output_" + std::to_string(1900 + t->tm_year) + std::to_string(1 + t->tm_mon) + std::to_string(t->tm_mday) +
               "_" + std::to_string(t->tm_hour) + std::to_string(t->tm_min) + std::to_string(t->tm_sec);
    const char *prog = "komb2";
    bool have_i = false, have_j = false, have_u = false;
    auto need_value = [&](int &k, const std::string &flag) -> std::string {
        if (k + 1 >= argc) parse_error(prog, "Argument: " + flag + "\n             Missing a value for this argument!");
        return argv[++k];
    };
    auto to_int = [&](const std::string &v, const std::string &flag) -> int {
        char *end = nullptr;
        long x = strtol(v.c_str(), &end, 10);
        if (v.empty() || *end != '\0') parse_error(prog, "Argument: " + flag + "\n             Couldn't read argument value from string '" + v + "'");
        return (int)x;
    };
    for (int k = 1; k < argc; ++k) {
        std::string s = argv[k];
        if (s == "--") break;                              // --ignore_rest
        if (s == "-h" || s == "--help") { usage(prog); exit(0); }            // HelpVisitor.h:70
        if (s == "--version") { printf("\n%s  version: 2.0\n\n", prog); exit(0); } // VersionVisitor.h:74
        if (s == "-i" || s == "--input") { a.input = need_value(k, "-i (--input)"); have_i = true; }
        else if (s == "-j" || s == "--input2") { a.input2 = need_value(k, "-j (--input2)"); have_j = true; }
        else if (s == "-u" || s == "--input-unitigs") { a.unitigs = need_value(k, "-u (--input-unitigs)"); have_u = true; }
        else if (s == "-l" || s == "--readlen") a.readlen = to_int(need_value(k, "-l (--readlen)"), "-l (--readlen)");   // "-l -1" is a value
        else if (s == "-t" || s == "--threads") a.threads = to_int(need_value(k, "-t (--threads)"), "-t (--threads)");
        else if (s == "-o" || s == "--output") a.outdir = need_value(k, "-o (--output)");
        else if (s == "-f" || s == "--fulgor") a.fulgor = true;
        else parse_error(prog, "Argument: " + s + "\n             Couldn't find match for argument");
    }
    if (!have_i) parse_error(prog, "Required argument missing: input");
    if (!have_j) parse_error(prog, "Required argument missing: input2");
    if (!have_u) parse_error(prog, "Required argument missing: input-unitigs");
    if (a.threads < 1) a.threads = 1;
    return a;
}

// ---------------------------------------------------------------------- SAM
// SAM -> graph pipeline (reference: readSAM src/graph.cpp:166-257, getEdgeInfo :259-285,
// generateGraph :287-393).  Same observable result -- vertex = unitig name seen on a parsed
// line, edge = two unitigs sharing a read key in either file -- built without per-read string
// hash sets: both files are mmap'ed, every thread parses the lines of its own byte chunk into
// (read-key view, unitig id) records, records are sorted by read key (64-bit hash first,
// bytes on a tie, so grouping is exact), each run of equal keys is one clique, and the
// expanded pairs are deduplicated by sort + unique.
struct Mapped {
    const char *data = nullptr;
    size_t size = 0;
    std::string fallback;                                  // used when mmap is not possible
    void open(const std::string &path)
    {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) file_not_found(path);
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); file_not_found(path); }
        size = (size_t)st.st_size;
        if (size == 0) { ::close(fd); data = ""; return; }
        void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) {
            fallback.resize(size);
            size_t got = 0;
            while (got < size) {
                ssize_t r = ::read(fd, &fallback[got], size - got);
                if (r <= 0) { fprintf(stderr, "Encountered error while reading %s\n", path.c_str()); leave(EXIT_FAILURE); }   // src/graph.cpp:188-192
                got += (size_t)r;
            }
            data = fallback.data();
        } else {
            data = (const char *)p;
            madvise(p, size, MADV_SEQUENTIAL);
        }
        ::close(fd);
    }
    ~Mapped() { if (data && fallback.empty() && size) munmap((void *)data, size); }
};

struct View { const char *p; uint32_t n; };
inline bool operator==(const View &a, const View &b) { return a.n == b.n && memcmp(a.p, b.p, a.n) == 0; }
struct ViewHash {
    size_t operator()(const View &v) const
    {
        uint64_t h = 0xCBF29CE484222325ull;                // FNV-1a, then a final mix
        for (uint32_t i = 0; i < v.n; ++i) { h ^= (unsigned char)v.p[i]; h *= 0x100000001B3ull; }
        h ^= h >> 32; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29;
        return (size_t)h;
    }
};

struct Names {                                             // unitig name <-> vid (src/graph.cpp:242-256)
    std::vector<std::string> name;                         // vid order = first appearance (file, then byte offset)
};

struct Rec {                                               // one parsed alignment line
    uint64_t hash;                                         // of the read key
    View key;                                              // read.substr(1, read.find('/')) (src/graph.cpp:235)
    int32_t vid;                                           // thread-local unitig id, later the global vid
};

// Which lines does the reference parse?  readSAM (src/graph.cpp:195-238) splits the BYTES of the
// file over T OpenMP threads (static schedule: the first n mod T threads get ceil(n/T) bytes, the
// rest floor(n/T)); each thread records the newlines inside its own chunk and parses only the lines
// that lie between two of its own newlines (thread 0 also gets a synthetic newline before byte 0).
// The line that starts after a chunk's last newline is parsed by nobody, and neither is a final
// line without '\n'.  strict = every line.  Calls f(begin, end) for every parsed line of chunk t.
template <class F>
void for_lines_of_chunk(const char *buf, size_t n, int T, int t, bool strict, F &&f)
{
    const size_t q = n / (size_t)T, r = n % (size_t)T;
    const size_t lo = (size_t)t * q + std::min((size_t)t, r), hi = lo + q + ((size_t)t < r ? 1 : 0);
    if (strict) {
        // every line belongs to the chunk that holds its first byte
        size_t b = lo;
        if (lo > 0) { while (b < n && buf[b - 1] != '\n' && buf[b - 1] != '\0') ++b; }
        while (b < hi && b < n) {
            size_t e = b;
            while (e < n && buf[e] != '\n' && buf[e] != '\0') ++e;
            if (e > b) f(b, e);
            b = e + 1;
        }
        return;
    }
    bool have_prev = (t == 0);                             // thread 0: position[0] = {0}
    size_t prev = 0;
    for (size_t i = lo; i < hi; ++i) {
        if (buf[i] == '\n' || buf[i] == '\0') {
            if (have_prev) {
                size_t start = prev + 1;
                if (start == 1) start = 0;                 // src/graph.cpp:218-219
                f(start, i);
            }
            prev = i; have_prev = true;
        }
    }
}

struct ThreadParse {
    std::vector<Rec> recs;
    std::unordered_map<View, int32_t, ViewHash> local;     // unitig name -> local id
    std::vector<View> local_names;
    std::vector<uint64_t> local_hash;                      // ViewHash of local_names[i]
    std::vector<uint64_t> first_pos;                       // (file << 48 | byte offset) of the first line naming it
};

void parse_sam(const Mapped &m, int file_idx, int threads, bool strict, std::vector<ThreadParse> &tp)
{
#pragma omp parallel for num_threads(threads) schedule(static, 1)
    for (int t = 0; t < threads; ++t) {
        ThreadParse &me = tp[(size_t)t];
        for_lines_of_chunk(m.data, m.size, threads, t, strict, [&](size_t b, size_t e) {
            const char *line = m.data + b;
            const size_t len = e - b;
            if (len == 0 || line[0] == '@') return;        // header (src/graph.cpp:221)
            // strtok_r(line, "\t"): empty fields are skipped; field 0 = QNAME, field 2 = RNAME
            size_t pos = 0;
            const char *tok[3] = {nullptr, nullptr, nullptr};
            size_t tlen[3] = {0, 0, 0};
            int nt = 0;
            while (nt < 3 && pos < len) {
                while (pos < len && line[pos] == '\t') ++pos;
                if (pos >= len) break;
                const char *tb = line + pos;
                const char *te = (const char *)memchr(tb, '\t', len - pos);
                const size_t l = te ? (size_t)(te - tb) : len - pos;
                tok[nt] = tb; tlen[nt] = l; ++nt;
                pos += l;
            }
            if (nt < 3) return;                            // the reference would dereference NULL here
            if (tlen[2] == 1 && tok[2][0] == '*') return;  // unmapped (src/graph.cpp:233)
            // key = read.substr(1, read.find('/')): drop the first char, keep through the first '/'
            View key{tok[0], 0};
            if (tlen[0] >= 1) {
                const char *sl = (const char *)memchr(tok[0], '/', tlen[0]);
                const size_t cnt = sl ? (size_t)(sl - tok[0]) : std::string::npos;        // find('/')
                const size_t avail = tlen[0] - 1;
                key.p = tok[0] + 1;
                key.n = (uint32_t)std::min(avail, cnt);
            }
            const View nm{tok[2], (uint32_t)tlen[2]};
            auto it = me.local.find(nm);
            int32_t lid;
            if (it == me.local.end()) {
                lid = (int32_t)me.local_names.size();
                me.local.emplace(nm, lid);
                me.local_names.push_back(nm);
                me.local_hash.push_back((uint64_t)ViewHash{}(nm));
                me.first_pos.push_back(((uint64_t)file_idx << 48) | (uint64_t)b);
            } else lid = it->second;
            me.recs.push_back(Rec{(uint64_t)ViewHash{}(key), key, lid});
        });
    }
}

// Parses both files and returns the raw (u,v) vertex pairs of all cliques, deduplicated.
std::vector<int64_t> build_edges(const std::string &path1, const std::string &path2, int threads, bool strict, Names &names,
                                 double *t_sam, double *t_merge, double *t_expand)
{
    const auto t0 = clk::now();
    Mapped m1, m2;
    m1.open(path1);                                        // src/komb2.cpp:93
    m2.open(path2);                                        // src/komb2.cpp:95
    std::vector<ThreadParse> tp1((size_t)threads), tp2((size_t)threads);
    const bool host_times = getenv("KOMB_HOST_TIMES") != nullptr;
    auto lap = [&, last = clk::now()](const char *what) mutable {
        if (host_times) fprintf(stderr, "komb2 host: %-28s %.3f s\n", what, since(last));
        last = clk::now();
    };
    lap("mmap");
    parse_sam(m1, 0, threads, strict, tp1);
    parse_sam(m2, 1, threads, strict, tp2);
    lap("parse both SAM files");
    // global vid = order of first appearance (file 1 before file 2, then byte offset): independent of T in strict mode.
    // Every thread has met most of the unitigs, so the per-thread dictionaries hold T x |V| names in total;
    // merging them serially was the largest single cost of the pipeline.  The names are partitioned by hash
    // instead: bucket b takes the names with hash % B == b from every dictionary, all buckets in parallel.
    std::vector<ThreadParse *> all;
    for (auto *tp : {&tp1, &tp2}) for (auto &th : *tp) all.push_back(&th);
    const size_t B = (size_t)threads * 4;
    std::vector<std::unordered_map<View, uint64_t, ViewHash>> bucket(B);      // name -> first position, later -> vid
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (size_t b = 0; b < B; ++b) {
        auto &m = bucket[b];
        for (ThreadParse *th : all)
            for (size_t i = 0; i < th->local_names.size(); ++i) {
                if (th->local_hash[i] % B != b) continue;
                auto ins = m.emplace(th->local_names[i], th->first_pos[i]);
                if (!ins.second && th->first_pos[i] < ins.first->second) ins.first->second = th->first_pos[i];
            }
    }
    lap("merge unitig dictionaries");
    std::vector<size_t> boff(B + 1, 0);
    for (size_t b = 0; b < B; ++b) boff[b + 1] = boff[b] + bucket[b].size();
    std::vector<std::pair<uint64_t, View>> order(boff[B]);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (size_t b = 0; b < B; ++b) {
        size_t k = boff[b];
        for (auto &kv : bucket[b]) order[k++] = std::make_pair(kv.second, kv.first);
    }
    __gnu_parallel::sort(order.begin(), order.end(), [](const auto &x, const auto &y) { return x.first < y.first; },
                         __gnu_parallel::default_parallel_tag((unsigned)threads));
    names.name.clear();
    names.name.resize(order.size());
#pragma omp parallel for num_threads(threads) schedule(static)
    for (size_t i = 0; i < order.size(); ++i) {
        const View &nm = order[i].second;
        bucket[(size_t)ViewHash{}(nm) % B].find(nm)->second = (uint64_t)i;      // no insertion: elements are written in place
        names.name[i].assign(nm.p, nm.n);
    }
    lap("number the vertices");
    // gather all records, local id -> global vid
    size_t total = 0;
    std::vector<size_t> offs;
    for (auto *tp : {&tp1, &tp2}) for (auto &th : *tp) { offs.push_back(total); total += th.recs.size(); }
    std::vector<Rec> recs(total);
    {
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (size_t k = 0; k < all.size(); ++k) {
            ThreadParse &th = *all[k];
            std::vector<int32_t> remap(th.local_names.size());
            for (size_t i = 0; i < remap.size(); ++i) remap[i] = (int32_t)bucket[th.local_hash[i] % B].find(th.local_names[i])->second;
            for (size_t i = 0; i < th.recs.size(); ++i) { Rec r = th.recs[i]; r.vid = remap[(size_t)r.vid]; recs[offs[k] + i] = r; }
            std::vector<Rec>().swap(th.recs);
        }
    }
    lap("gather records");
    *t_sam = since(t0);

    // getEdgeInfo: one group per read key over both files -- sort by (hash, key bytes, vid)
    const auto t1 = clk::now();
    auto less = [](const Rec &a, const Rec &b) {
        if (a.hash != b.hash) return a.hash < b.hash;
        if (a.key.n != b.key.n) return a.key.n < b.key.n;
        const int c = memcmp(a.key.p, b.key.p, a.key.n);
        if (c != 0) return c < 0;
        return a.vid < b.vid;
    };
    __gnu_parallel::sort(recs.begin(), recs.end(), less, __gnu_parallel::default_parallel_tag((unsigned)threads));
    *t_merge = since(t1);

    // generateGraph: every run of equal keys is a clique over its distinct vids
    const auto t2 = clk::now();
    std::vector<std::vector<uint64_t>> parts((size_t)threads);
#pragma omp parallel num_threads(threads)
    {
        const int t = omp_get_thread_num();
        size_t lo = recs.size() * (size_t)t / (size_t)threads, hi = recs.size() * (size_t)(t + 1) / (size_t)threads;
        auto same = [&](size_t i, size_t j) { return recs[i].hash == recs[j].hash && recs[i].key == recs[j].key; };
        while (lo > 0 && lo < recs.size() && same(lo - 1, lo)) ++lo;          // start at a run boundary
        while (hi > 0 && hi < recs.size() && same(hi - 1, hi)) ++hi;
        std::vector<uint64_t> &out = parts[(size_t)t];
        std::vector<int32_t> clique;
        for (size_t i = lo; i < hi;) {
            size_t j = i;
            clique.clear();
            while (j < hi && same(i, j)) { if (clique.empty() || clique.back() != recs[j].vid) clique.push_back(recs[j].vid); ++j; }
            for (size_t x = 0; x < clique.size(); ++x)
                for (size_t y = x + 1; y < clique.size(); ++y)
                    out.push_back(((uint64_t)(uint32_t)clique[x] << 32) | (uint32_t)clique[y]);   // vids ascending: x < y
            i = j;
        }
    }
    std::vector<Rec>().swap(recs);
    size_t npairs = 0;
    for (auto &v : parts) npairs += v.size();
    std::vector<uint64_t> keys;
    keys.reserve(npairs);
    for (auto &v : parts) { keys.insert(keys.end(), v.begin(), v.end()); std::vector<uint64_t>().swap(v); }
    __gnu_parallel::sort(keys.begin(), keys.end(), std::less<uint64_t>(), __gnu_parallel::default_parallel_tag((unsigned)threads));
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    std::vector<int64_t> edges(keys.size() * 2);
#pragma omp parallel for num_threads(threads)
    for (size_t i = 0; i < keys.size(); ++i) { edges[2 * i] = (int64_t)(keys[i] >> 32); edges[2 * i + 1] = (int64_t)(keys[i] & 0xFFFFFFFFu); }
    *t_expand = since(t2);
    return edges;
}

// readUnitigsFile (src/graph.cpp:565-589): name = text between '>' and the first
// space; sequence lines concatenated, last character of each line dropped.
std::unordered_map<std::string, std::string> read_unitigs(const std::string &path)
{
    std::unordered_map<std::string, std::string> u;
    FILE *fp = fopen(path.c_str(), "r");
    if (!fp) file_not_found(path);
    char *line = nullptr;
    size_t cap = 0;
    ssize_t n;
    std::string cur;
    while ((n = getline(&line, &cap, fp)) != -1) {
        std::string s(line, (size_t)n);
        if (!s.empty() && s[0] == '>') {
            const size_t sp = s.find(' ');
            cur = s.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
            u[cur] = std::string();
        } else if (!s.empty()) {
            u[cur] += s.substr(0, s.size() - 1);
        }
    }
    free(line);
    fclose(fp);
    return u;
}

// CoreA::readKOMBOutput (src/CoreA.h:24-56): field 2 = coreness, field 3 = degree; '#' lines skipped
bool read_kcore_tsv(const std::string &path, std::vector<int32_t> &core, std::vector<int32_t> &deg, std::vector<std::string> *name = nullptr)
{
    FILE *fp = fopen(path.c_str(), "r");
    if (!fp) return false;
    char *line = nullptr;
    size_t cap = 0;
    while (getline(&line, &cap, fp) != -1) {
        if (line[0] == '#') continue;
        int i = 0;
        char *save = nullptr;
        for (char *tok = strtok_r(line, "\t", &save); tok; tok = strtok_r(nullptr, "\t", &save), ++i) {
            if (i == 1 && name) name->emplace_back(tok);
            if (i == 2) core.push_back(atoi(tok));
            if (i == 3) deg.push_back(atoi(tok));
        }
    }
    free(line);
    fclose(fp);
    return true;
}

// Writes rows 0..n-1 to fp: rows are formatted by all threads into per-thread buffers (row(i, buf)
// appends the text of row i, exactly what the reference's fprintf would print) and the buffers are
// written in order -- the 10^7..10^8 fprintf calls of src/graph.cpp:468-475 and
// src/CombineCoreA.h:36-39 are the floor of the reference's own timed region.
template <class RowFn>
void write_rows(FILE *fp, int64_t n, int threads, RowFn &&row)
{
    const int64_t kBlockRows = 1 << 16;
    const int64_t nblocks = (n + kBlockRows - 1) / kBlockRows;
    for (int64_t b0 = 0; b0 < nblocks; b0 += threads) {
        const int64_t nb = std::min<int64_t>(threads, nblocks - b0);
        std::vector<std::string> out((size_t)nb);
#pragma omp parallel for num_threads(threads) schedule(static, 1)
        for (int64_t k = 0; k < nb; ++k) {
            std::string &buf = out[(size_t)k];
            const int64_t lo = (b0 + k) * kBlockRows, hi = std::min(n, lo + kBlockRows);
            buf.reserve((size_t)(hi - lo) * 24);
            for (int64_t i = lo; i < hi; ++i) row(i, buf);
        }
        for (auto &buf : out) fwrite(buf.data(), 1, buf.size(), fp);
    }
}

[[noreturn]] void die_accel(komb_ctx *ctx, const char *what, int rc)
{
    fprintf(stderr, "komb2: %s failed (%d): %s\n", what, rc, ctx ? komb_last_error(ctx) : "no context");
    leave(EXIT_FAILURE);
}

bool env_on(const char *name)
{
    const char *v = getenv(name);
    return v && *v && strcmp(v, "0") != 0;
}

// CombineCoreA::run (src/CombineCoreA.h:16-43)
void corea_stage(komb_ctx *ctx, const std::string &outdir, const std::vector<int32_t> &deg, const std::vector<int32_t> &core, int threads)
{
    const int n = (int)deg.size();
    const double dense_ratio = n ? (double)(*std::max_element(core.begin(), core.end()) / 2) : 0.0;   // integer division (:24)
    fprintf(stdout, "Dense Ratio: %f\n", dense_ratio);
    std::vector<double> score((size_t)n);
    int rc = komb_corea_scores(ctx, deg.data(), core.data(), n, score.data());
    if (rc != KOMB_OK) die_accel(ctx, "komb_corea_scores", rc);
    if ((int64_t)(n ? *std::max_element(core.begin(), core.end()) : 0) * n + (n ? *std::max_element(deg.begin(), deg.end()) : 0) > 2147483647LL)
        fprintf(stderr, "komb2: note: coreness*n+degree exceeds 2^31-1; the reference's int key (src/CoreA.h:122) would overflow here, 64-bit keys used\n");
    const double max_dmp = n ? *std::max_element(score.begin(), score.end()) : 0.0;
    fprintf(stdout, "Max CoreA score: %f\n", max_dmp);
    const std::string path = outdir + "/CoreA_anomaly.txt";
    FILE *fp = fopen(path.c_str(), "w+");
    if (!fp) file_not_found(path);
    write_rows(fp, n, threads, [&](int64_t i, std::string &buf) {
        char tmp[64];
        const int len = snprintf(tmp, sizeof(tmp), "%d\t%f\n", (int)i, score[(size_t)i]);      // src/CombineCoreA.h:38
        buf.append(tmp, (size_t)len);
    });
    fclose(fp);
}

// combineFile (src/graph.cpp:591-635; call commented out at src/komb2.cpp:126): one FASTA record per
// kcore.tsv row, header ">Unitig_<name>|<coreness>", sequence only when the unitig file has that name.
// The reference re-parses kcore.tsv for the fields; the same values are still in memory here.
void combine_file(const std::string &outdir, const Names &names, const std::vector<int32_t> &core,
                  const std::unordered_map<std::string, std::string> &unitigs, int threads)
{
    const std::string path = outdir + "/combined.fasta";
    FILE *fp = fopen(path.c_str(), "w+");
    if (!fp) file_not_found(path);
    write_rows(fp, (int64_t)core.size(), threads, [&](int64_t i, std::string &buf) {
        const std::string &nm = names.name[(size_t)i];
        buf.append(">Unitig_").append(nm).append("|").append(std::to_string(core[(size_t)i])).append("\n");
        auto it = unitigs.find(nm);
        if (it != unitigs.end()) buf.append(it->second).append("\n");
    });
    fclose(fp);
}

// getMedian (src/graph.cpp:650-665), including its window: size = end - start - 1
double v1_median(const std::vector<double> &v, int start, int end)
{
    const int size = end - start - 1;
    if (size % 2 == 0) return (v[(size_t)(start + size / 2 - 1)] + v[(size_t)(start + size / 2)]) / 2;
    return v[(size_t)(start + (size - 1) / 2)];
}

// splitAnomalousUnitigs (src/graph.cpp:667-749; call commented out at src/komb2.cpp:139). Kept as the
// reference has it, quirks included: scores are the six-decimal text of CoreA_anomaly.txt read back;
// row i of the output is decided by the i-th SMALLEST score (the sorted copy is what :725 indexes), is
// labelled "Unitig_<i>" and looks the sequence up under the name "<i>". mode "fixed" instead decides
// row i by vertex i's own score and uses the vertex's unitig name.
// The quartile windows read outside the vector for fewer than 6 scores (undefined in the reference):
// refused with a note.
bool split_anomalous(const std::string &outdir, const Names &names, const std::unordered_map<std::string, std::string> &unitigs,
                     bool fixed, int threads)
{
    const std::string in_path = outdir + "/CoreA_anomaly.txt";
    FILE *in = fopen(in_path.c_str(), "r");
    if (!in) file_not_found(in_path);
    std::vector<double> score;
    char *line = nullptr;
    size_t cap = 0;
    while (getline(&line, &cap, in) != -1) {
        const char *tab = strchr(line, '\t');
        if (tab) score.push_back(strtod(tab + 1, nullptr));
    }
    free(line);
    fclose(in);
    const int size = (int)score.size();
    if (size < 6) {
        fprintf(stderr, "komb2: note: %d CoreA scores; the reference's quartile windows (src/graph.cpp:714-724) need at least 6, split skipped\n", size);
        return false;
    }
    std::vector<double> sorted(score);
    std::sort(sorted.begin(), sorted.end());
    const double q1 = v1_median(sorted, 0, size / 2 - 1);
    const double q3 = size % 2 == 0 ? v1_median(sorted, size / 2, size - 1) : v1_median(sorted, size / 2 + 1, size - 1);
    const double cutoff = q3 + 1.5 * (q3 - q1);
    const std::vector<double> &decide = fixed ? score : sorted;

    const std::string top_path = outdir + "/top_scoring_anomalous_unitigs.txt", low_path = outdir + "/low_scoring_anomalous_unitigs.txt";
    FILE *top = fopen(top_path.c_str(), "w+"), *low = fopen(low_path.c_str(), "w+");
    if (!top) file_not_found(top_path);
    if (!low) file_not_found(low_path);
    for (int pass = 0; pass < 2; ++pass) {
        write_rows(pass == 0 ? top : low, size, threads, [&](int64_t i, std::string &buf) {
            if ((decide[(size_t)i] >= cutoff) != (pass == 0)) return;
            const std::string key = fixed && (size_t)i < names.name.size() ? names.name[(size_t)i] : std::to_string(i);
            buf.append("Unitig_").append(key).append("\n");
            auto it = unitigs.find(key);
            if (it != unitigs.end()) buf.append(it->second).append("\n");
        });
    }
    fclose(top);
    fclose(low);
    return true;
}

} // namespace

int main(int argc, const char **argv)
{
    const Args args = parse_args(argc, argv);
    const auto begin = clk::now();

    // standalone use: the reference needs -o to exist (KOMB.py creates it, KOMB.py:41-52)
    mkdir(args.outdir.c_str(), 0777);

    // --corea-only mode: resume from an existing kcore.tsv, like CoreA itself does
    if (env_on("KOMB_COREA_ONLY")) {
        std::vector<int32_t> core, deg;
        if (!read_kcore_tsv(args.outdir + "/kcore.tsv", core, deg)) file_not_found(args.outdir + "/kcore.tsv");
        komb_opts o{};
        o.device = getenv("KOMB_DEVICE") ? atoi(getenv("KOMB_DEVICE")) : 0;
        komb_ctx *ctx = komb_create(&o);
        corea_stage(ctx, args.outdir, deg, core, args.threads);
        komb_destroy(ctx);
        return 0;
    }

    // v1-outputs-only mode: combined.fasta and the split from an existing kcore.tsv + CoreA_anomaly.txt (no device)
    if (env_on("KOMB_V1_ONLY")) {
        std::vector<int32_t> core, deg;
        Names names;
        if (!read_kcore_tsv(args.outdir + "/kcore.tsv", core, deg, &names.name)) file_not_found(args.outdir + "/kcore.tsv");
        const auto unitigs = read_unitigs(args.unitigs);
        const char *mode = getenv("KOMB_V1_OUTPUTS");
        combine_file(args.outdir, names, core, unitigs, args.threads);
        split_anomalous(args.outdir, names, unitigs, mode && strcmp(mode, "fixed") == 0, args.threads);
        return 0;
    }

    // Creating the HIP context takes ~0.6 s; it runs beside the SAM parse (never in the host-only test mode).
    komb_opts opts{};
    opts.device = getenv("KOMB_DEVICE") ? atoi(getenv("KOMB_DEVICE")) : 0;
    opts.reserved[0] = KOMB_CREATE_WARM_UPLOAD;          // the upload's pinned staging buffers are made beside the SAM parse, off the critical path
    std::future<komb_ctx *> ctx_early;
    g_ctx_early = &ctx_early;
    for (const std::string *path : {&args.input, &args.input2})          // fail on a missing input before a second thread exists
        if (access(path->c_str(), R_OK) != 0) file_not_found(*path);
    if (!env_on("KOMB_STOP_AFTER_EDGES")) ctx_early = std::async(std::launch::async, [&opts]() { return komb_create(&opts); });

    const bool strict = env_on("KOMB_STRICT_SAM");
    const auto begin_komb = clk::now();
    Names names;
    double t_sam_s = 0, t_merge_s = 0, t_expand_s = 0;
    std::vector<int64_t> edges = build_edges(args.input, args.input2, args.threads, strict, names, &t_sam_s, &t_merge_s, &t_expand_s);
    fprintf(stdout, "\nTime elapsed for reading SAMs: %.3f s\n", t_sam_s);
    fprintf(stdout, "\nTime elapsed for edgeInfo: %.3f s\n", t_merge_s);
    fprintf(stdout, "\nTime elapsed for converting umapset to vec<vec>: %.3f s\n", 0.0);
    fprintf(stdout, "\nTime elapsed for constructing local edges: %.3f s\n", t_expand_s);
    auto t_generate = clk::now();
    fprintf(stdout, "\nTime elapsed for generateGraph: %.3f s\n", t_expand_s);
    auto t0 = clk::now();

    // readEdgeList (src/graph.cpp:395-453): edgelist.txt = the raw pairs
    const int64_t nv = (int64_t)names.name.size();
    t0 = clk::now();
    {
        const std::string path = args.outdir + "/edgelist.txt";
        FILE *f = fopen(path.c_str(), "w");
        if (!f) file_not_found(path);
        write_rows(f, (int64_t)(edges.size() / 2), args.threads, [&](int64_t i, std::string &buf) {
            char tmp[48];
            const int len = snprintf(tmp, sizeof(tmp), "%ld\t%ld\n", (long)edges[2 * (size_t)i], (long)edges[2 * (size_t)i + 1]);   // src/graph.cpp:425
            buf.append(tmp, (size_t)len);
        });
        fclose(f);
    }
    if (env_on("KOMB_STOP_AFTER_EDGES")) {
        const std::string path = args.outdir + "/vertex_names.txt";
        FILE *f = fopen(path.c_str(), "w");
        if (!f) file_not_found(path);
        for (int64_t v = 0; v < nv; ++v) fprintf(f, "%ld\t%s\n", (long)v, names.name[(size_t)v].c_str());
        fclose(f);
        return 0;
    }

    komb_ctx *ctx = ctx_early.get();
    if (!ctx) die_accel(nullptr, "komb_create", KOMB_ERR_NOMEM);
    fprintf(stdout, "\nTime elapsed for initializing igraph graph: %.3f s\n", since(t0));
    t0 = clk::now();
    int rc = komb_graph_from_edges(ctx, nv, (int64_t)(edges.size() / 2), edges.data());   // igraph_create + igraph_simplify
    if (rc != KOMB_OK) die_accel(ctx, "komb_graph_from_edges", rc);
    std::vector<int64_t>().swap(edges);
    fprintf(stdout, "\nTime elapsed for simplifying graph: %.3f s\n", since(t0));
    int64_t gnv = 0, gne = 0;
    komb_graph_info(ctx, &gnv, &gne);
    fprintf(stdout, "GraphInfo...\n\tNumber of vertices: %d\n", (int)gnv);
    fprintf(stdout, "\tNumber of edges: %d\n", (int)gne);

    // src/graph.cpp:446 reads the unitig FASTA here; only the (disabled) truss / v1 stages use it, so it is
    // opened now -- a missing file must fail at this point, as in the reference -- and parsed only when needed
    if (access(args.unitigs.c_str(), R_OK) != 0) file_not_found(args.unitigs);
    const bool need_unitigs = env_on("KOMB_TRUSS") || env_on("KOMB_V1_OUTPUTS");
    const auto unitigs = need_unitigs ? read_unitigs(args.unitigs) : std::unordered_map<std::string, std::string>();

    // runCore (src/graph.cpp:455-484)
    t0 = clk::now();
    std::vector<int32_t> deg((size_t)nv), core((size_t)nv);
    rc = komb_degree_coreness(ctx, deg.data(), core.data());
    if (rc != KOMB_OK) die_accel(ctx, "komb_degree_coreness", rc);
    const int max_coreness = nv ? *std::max_element(core.begin(), core.end()) : 0;
    std::vector<uint8_t> maxcore((size_t)nv, 0);
    {
        const std::string path = args.outdir + "/kcore.tsv";
        FILE *kcf = fopen(path.c_str(), "w+");
        if (!kcf) file_not_found(path);
        fprintf(kcf, "#VID\tName\tCoreness\tDegree\n");
        for (int64_t i = 0; i < nv; ++i)
            if (core[(size_t)i] == max_coreness) maxcore[(size_t)i] = 1;       // subgraph_nodes (:470-473)
        write_rows(kcf, nv, args.threads, [&](int64_t i, std::string &buf) {
            char tmp[48];
            int len = snprintf(tmp, sizeof(tmp), "%d\t", (int)i);
            buf.append(tmp, (size_t)len);
            buf.append(names.name[(size_t)i]);
            len = snprintf(tmp, sizeof(tmp), "\t%d\t%d\n", core[(size_t)i], deg[(size_t)i]);   // src/graph.cpp:474
            buf.append(tmp, (size_t)len);
        });
        fclose(kcf);
    }

    // runTruss (src/graph.cpp:486-563) -- disabled in the reference at :478, opt-in here
    if (env_on("KOMB_TRUSS") && nv > 0) {
        fprintf(stdout, "BUILDING K-TRUSS:\n");
        fprintf(stdout, "Selected unitigs in maximal core.\n");
        rc = komb_truss_run(ctx, maxcore.data());
        if (rc != KOMB_OK) die_accel(ctx, "komb_truss_run", rc);
        int64_t ne_sub = 0;
        komb_truss_count(ctx, &ne_sub);
        fprintf(stdout, "Succesfully created a %d-core subgraph, with %d edges.\n", max_coreness, (int)ne_sub);
        std::vector<int32_t> eu((size_t)ne_sub), ev((size_t)ne_sub), tr((size_t)ne_sub);
        rc = komb_truss_fetch(ctx, eu.data(), ev.data(), tr.data());
        if (rc != KOMB_OK) die_accel(ctx, "komb_truss_fetch", rc);
        fprintf(stdout, "Computed trussness of edges.\n");
        const std::string path = args.outdir + "/truss_unitigs.fasta";
        FILE *tf = fopen(path.c_str(), "w+");
        if (!tf) file_not_found(path);
        const int threshold = ne_sub ? *std::max_element(tr.begin(), tr.end()) : 0;
        std::vector<uint8_t> in((size_t)nv, 0);
        int count = 0;
        for (int64_t e = 0; e < ne_sub; ++e)
            if (tr[(size_t)e] >= threshold) { in[(size_t)eu[(size_t)e]] = 1; in[(size_t)ev[(size_t)e]] = 1; }
        for (int64_t v = 0; v < nv; ++v) {                 // the reference iterates an unordered_set<int>: order unspecified
            if (!in[(size_t)v]) continue;
            ++count;
            fprintf(tf, ">Unitig_%s\n", names.name[(size_t)v].c_str());
            auto it = unitigs.find(names.name[(size_t)v]);
            if (it != unitigs.end()) fprintf(tf, "%s\n", it->second.c_str());
        }
        fclose(tf);
        fprintf(stdout, "Found %d unitigs in %d-truss, saved at %s\n", count, threshold + 1, path.c_str());   // "+1" as src/graph.cpp:557
    }
    fprintf(stdout, "\nTime elapsed doing K-core decomposition: %.3f s\n", since(t0));
    fprintf(stdout, "Created Kcore\n");
    auto t_core = clk::now();
    fprintf(stdout, "\nTime elapsed for edgeInfo: %.3f s\n", std::chrono::duration<double>(t_core - t_generate).count());   // sic (src/komb2.cpp:124)
    const char *v1 = getenv("KOMB_V1_OUTPUTS");
    const bool v1_on = env_on("KOMB_V1_OUTPUTS");
    if (v1_on) combine_file(args.outdir, names, core, unitigs, args.threads);
    auto t_combine = clk::now();
    fprintf(stdout, "\nTime elapsed for combineFile: %.3f s\n", std::chrono::duration<double>(t_combine - t_core).count());

    corea_stage(ctx, args.outdir, deg, core, args.threads);  // anomalyDetection (src/graph.cpp:637-648)
    fprintf(stdout, "\nTime elapsed for anomalyDetection: %.3f s\n", since(t_combine));
    fprintf(stdout, "Identified anomalous unitigs\n");
    if (v1_on) split_anomalous(args.outdir, names, unitigs, strcmp(v1, "fixed") == 0, args.threads);
    fprintf(stdout, "Created anomalouss unitigs file\n");
    fprintf(stdout, "\nTime elapsed for KOMB: %.3f s\n", since(begin_komb));
    fprintf(stdout, "\nTime elapsed for analysis (sec) = %.3f \n", since(begin));
    komb_destroy(ctx);
    return 0;
}
