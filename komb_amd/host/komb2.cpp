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
//   KOMB_TRUSS=1        re-enable the runTruss stage the reference has commented
//                       out at src/graph.cpp:478 (writes truss_unitigs.fasta)
//   KOMB_STRICT_SAM=1   parse every SAM line (the reference drops the line that
//                       straddles each OpenMP byte-chunk boundary, see readSAM)
//   KOMB_DEVICE=<n>     HIP device ordinal (default 0)
//   KOMB_STOP_AFTER_EDGES=1  write edgelist.txt + vertex_names.txt and stop
//                       before touching the device (host-logic tests)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <omp.h>
#include <sys/stat.h>

#include "komb_accel.h"

namespace {

using clk = std::chrono::steady_clock;
double since(clk::time_point t0) { return std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count() / 1000000.0; }

[[noreturn]] void file_not_found(const std::string &path)
{
    // src/graph.cpp:58-61
    fprintf(stderr, "File %s could not be opened. Exiting...\n", path.c_str());
    exit(EXIT_FAILURE);
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
// read key -> set of unitig vids.  Reference: umapset (src/graph.h:24) keyed by
// read.substr(1, read.find('/')) (src/graph.cpp:235).
using ReadMap = std::unordered_map<std::string, std::vector<int32_t>>;

struct Names {                                             // unitig name <-> vid (src/graph.cpp:242-256)
    std::unordered_map<std::string, int32_t> vid;
    std::vector<std::string> name;
    int32_t get(const char *s, size_t n)
    {
        std::string k(s, n);
        auto it = vid.find(k);
        if (it != vid.end()) return it->second;
        int32_t v = (int32_t)name.size();
        vid.emplace(k, v);
        name.push_back(std::move(k));
        return v;
    }
};

std::string slurp(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f) file_not_found(path);
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    rewind(f);
    std::string buf((size_t)sz, '\0');
    if (sz > 0 && fread(&buf[0], 1, (size_t)sz, f) != (size_t)sz) {
        fprintf(stderr, "Encountered error while reading %s\n", path.c_str());   // src/graph.cpp:188-192
        exit(EXIT_FAILURE);
    }
    fclose(f);
    return buf;
}

// Which lines does the reference parse?  readSAM (src/graph.cpp:195-238) splits
// the BYTES of the file over T OpenMP threads (static schedule: the first
// n mod T threads get ceil(n/T) bytes, the rest floor(n/T)); each thread records
// the newlines inside its own chunk and parses only the lines that lie between
// two of its own newlines (thread 0 also gets a synthetic newline before byte
// 0).  The line that starts after a chunk's last newline is parsed by nobody,
// and neither is a final line without '\n'.  strict = every line.
std::vector<std::pair<size_t, size_t>> line_spans(const std::string &buf, int T, bool strict)
{
    std::vector<std::pair<size_t, size_t>> spans;          // [begin, end) without the newline
    const size_t n = buf.size();
    if (strict) {
        size_t b = 0;
        for (size_t i = 0; i <= n; ++i)
            if (i == n || buf[i] == '\n' || buf[i] == '\0') { if (i > b) spans.emplace_back(b, i); b = i + 1; }
        return spans;
    }
    const size_t q = n / (size_t)T, r = n % (size_t)T;
    size_t lo = 0;
    for (int t = 0; t < T; ++t) {
        const size_t len = q + ((size_t)t < r ? 1 : 0), hi = lo + len;
        bool have_prev = (t == 0);                         // thread 0: position[0] = {0}
        size_t prev = 0;                                   // the reference's start_pos rule: pos+1, or 0 for the synthetic entry
        for (size_t i = lo; i < hi; ++i) {
            if (buf[i] == '\n' || buf[i] == '\0') {
                if (have_prev) {
                    size_t start = prev + 1;
                    if (start == 1) start = 0;             // src/graph.cpp:218-219
                    spans.emplace_back(start, i);
                }
                prev = i; have_prev = true;
            }
        }
        lo = hi;
    }
    return spans;
}

void read_sam(const std::string &path, int threads, bool strict, ReadMap &umap, Names &names)
{
    const std::string buf = slurp(path);
    const auto spans = line_spans(buf, threads, strict);
    for (const auto &sp : spans) {
        const char *line = buf.data() + sp.first;
        const size_t len = sp.second - sp.first;
        if (len == 0 || line[0] == '@') continue;          // header (src/graph.cpp:221)
        // strtok_r(line, "\t"): empty fields are skipped; field 0 = QNAME, field 2 = RNAME
        size_t pos = 0;
        const char *tok[3] = {nullptr, nullptr, nullptr};
        size_t tlen[3] = {0, 0, 0};
        int nt = 0;
        while (nt < 3 && pos < len) {
            while (pos < len && line[pos] == '\t') ++pos;
            if (pos >= len) break;
            size_t e = pos;
            while (e < len && line[e] != '\t') ++e;
            tok[nt] = line + pos; tlen[nt] = e - pos; ++nt;
            pos = e;
        }
        if (nt < 3) continue;                              // the reference would dereference NULL here
        if (tlen[2] == 1 && tok[2][0] == '*') continue;    // unmapped (src/graph.cpp:233)
        // key = read.substr(1, read.find('/'))
        std::string read(tok[0], tlen[0]);
        const size_t slash = read.find('/');
        std::string key = read.size() >= 1 ? read.substr(1, slash) : std::string();
        const int32_t v = names.get(tok[2], tlen[2]);
        auto &set = umap[key];
        if (std::find(set.begin(), set.end(), v) == set.end()) set.push_back(v);
    }
}

// getEdgeInfo (src/graph.cpp:259-285): per read key, the union of both mates' unitig sets
void merge_mates(ReadMap &a, ReadMap &b)
{
    for (auto &kv : b) {
        auto &dst = a[kv.first];
        for (int32_t v : kv.second)
            if (std::find(dst.begin(), dst.end(), v) == dst.end()) dst.push_back(v);
    }
    ReadMap().swap(b);
}

// generateGraph (src/graph.cpp:310-352): every clique expands to all i<j pairs; a
// pair already emitted in the same orientation is skipped (the reference's
// per-thread seen-set; equal to this for -t 1).
std::vector<int64_t> expand_cliques(const ReadMap &umap)
{
    std::vector<int64_t> edges;
    std::unordered_set<uint64_t> seen;
    for (const auto &kv : umap) {
        const auto &c = kv.second;
        for (size_t i = 0; i < c.size(); ++i)
            for (size_t j = i + 1; j < c.size(); ++j) {
                const uint64_t key = ((uint64_t)(uint32_t)c[i] << 32) | (uint32_t)c[j];
                if (seen.insert(key).second) { edges.push_back(c[i]); edges.push_back(c[j]); }
            }
    }
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
bool read_kcore_tsv(const std::string &path, std::vector<int32_t> &core, std::vector<int32_t> &deg)
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
            if (i == 2) core.push_back(atoi(tok));
            if (i == 3) deg.push_back(atoi(tok));
        }
    }
    free(line);
    fclose(fp);
    return true;
}

[[noreturn]] void die_accel(komb_ctx *ctx, const char *what, int rc)
{
    fprintf(stderr, "komb2: %s failed (%d): %s\n", what, rc, ctx ? komb_last_error(ctx) : "no context");
    exit(EXIT_FAILURE);
}

bool env_on(const char *name)
{
    const char *v = getenv(name);
    return v && *v && strcmp(v, "0") != 0;
}

// CombineCoreA::run (src/CombineCoreA.h:16-43)
void corea_stage(komb_ctx *ctx, const std::string &outdir, const std::vector<int32_t> &deg, const std::vector<int32_t> &core)
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
    std::vector<char> buf(1 << 20);
    setvbuf(fp, buf.data(), _IOFBF, buf.size());
    for (int i = 0; i < n; ++i) fprintf(fp, "%d\t%f\n", i, score[(size_t)i]);
    fclose(fp);
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
        corea_stage(ctx, args.outdir, deg, core);
        komb_destroy(ctx);
        return 0;
    }

    const bool strict = env_on("KOMB_STRICT_SAM");
    const auto begin_komb = clk::now();
    ReadMap umap1, umap2;
    Names names;
    read_sam(args.input, args.threads, strict, umap1, names);      // src/komb2.cpp:93
    read_sam(args.input2, args.threads, strict, umap2, names);     // src/komb2.cpp:95
    auto t_sam = clk::now();
    fprintf(stdout, "\nTime elapsed for reading SAMs: %.3f s\n", since(begin_komb));

    merge_mates(umap1, umap2);
    auto t_edgeinfo = clk::now();
    fprintf(stdout, "\nTime elapsed for edgeInfo: %.3f s\n", std::chrono::duration<double>(t_edgeinfo - t_sam).count());

    auto t0 = clk::now();
    fprintf(stdout, "\nTime elapsed for converting umapset to vec<vec>: %.3f s\n", since(t0));
    t0 = clk::now();
    std::vector<int64_t> edges = expand_cliques(umap1);
    ReadMap().swap(umap1);
    fprintf(stdout, "\nTime elapsed for constructing local edges: %.3f s\n", since(t0));
    auto t_generate = clk::now();
    fprintf(stdout, "\nTime elapsed for generateGraph: %.3f s\n", std::chrono::duration<double>(t_generate - t_edgeinfo).count());

    // readEdgeList (src/graph.cpp:395-453): edgelist.txt = the raw pairs
    const int64_t nv = (int64_t)names.name.size();
    t0 = clk::now();
    {
        const std::string path = args.outdir + "/edgelist.txt";
        FILE *f = fopen(path.c_str(), "w");
        if (!f) file_not_found(path);
        std::vector<char> buf(1 << 20);
        setvbuf(f, buf.data(), _IOFBF, buf.size());
        for (size_t i = 0; i + 1 < edges.size(); i += 2) fprintf(f, "%ld\t%ld\n", (long)edges[i], (long)edges[i + 1]);
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

    komb_opts opts{};
    opts.device = getenv("KOMB_DEVICE") ? atoi(getenv("KOMB_DEVICE")) : 0;
    komb_ctx *ctx = komb_create(&opts);
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

    const auto unitigs = read_unitigs(args.unitigs);               // src/graph.cpp:446

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
        std::vector<char> buf(1 << 20);
        setvbuf(kcf, buf.data(), _IOFBF, buf.size());
        fprintf(kcf, "#VID\tName\tCoreness\tDegree\n");
        for (int64_t i = 0; i < nv; ++i) {
            if (core[(size_t)i] == max_coreness) maxcore[(size_t)i] = 1;       // subgraph_nodes (:470-473)
            fprintf(kcf, "%d\t%s\t%d\t%d\n", (int)i, names.name[(size_t)i].c_str(), core[(size_t)i], deg[(size_t)i]);
        }
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
    auto t_combine = clk::now();
    fprintf(stdout, "\nTime elapsed for combineFile: %.3f s\n", std::chrono::duration<double>(t_combine - t_core).count());

    corea_stage(ctx, args.outdir, deg, core);              // anomalyDetection (src/graph.cpp:637-648)
    fprintf(stdout, "\nTime elapsed for anomalyDetection: %.3f s\n", since(t_combine));
    fprintf(stdout, "Identified anomalous unitigs\n");
    fprintf(stdout, "Created anomalouss unitigs file\n");
    fprintf(stdout, "\nTime elapsed for KOMB: %.3f s\n", since(begin_komb));
    fprintf(stdout, "\nTime elapsed for analysis (sec) = %.3f \n", since(begin));
    komb_destroy(ctx);
    return 0;
}
