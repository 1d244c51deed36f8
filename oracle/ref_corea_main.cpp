// ref_corea_main.cpp -- harness around the REFERENCE's own src/CoreA.h
// (std-only header, compiled where it lies under /root/reference by
// oracle/Makefile; nothing of it is copied into this repo).  The binary goes
// to oracle/_ref/corea_ref and is used by tests/ to pin the CoreA half of the
// oracle (rows a8-a10 of SURVEY.md section 8).  TEST INFRASTRUCTURE ONLY.
//
// Modes:
//   corea_ref arrays            stdin: n, then n lines "degree coreness"
//   corea_ref tsv <kcore.tsv>   parses the file with CoreA::readKOMBOutput
// Output: one score per line, "%.17g" (exact double round trip).
#include "CoreA.h"   // found via -I/root/reference/src

#include <cstdio>
#include <cstring>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: corea_ref arrays | tsv <file>\n"); return 2; }
    CoreA core_a;
    std::vector<int> degree, coreness;
    if (!strcmp(argv[1], "arrays")) {
        long n = 0;
        if (scanf("%ld", &n) != 1) return 2;
        degree.resize(n); coreness.resize(n);
        for (long i = 0; i < n; ++i)
            if (scanf("%d %d", &degree[i], &coreness[i]) != 2) return 2;
    } else if (!strcmp(argv[1], "tsv") && argc >= 3) {
        auto cd = core_a.readKOMBOutput(argv[2]);          // src/CoreA.h:24-56
        coreness = cd.first; degree = cd.second;
    } else return 2;
    int n = (int)degree.size();
    if (n == 0) return 0;
    double *score = core_a.getAnomalyScore(degree, coreness);   // src/CoreA.h:109-140
    for (int i = 0; i < n; ++i) printf("%.17g\n", score[i]);
    free(score);
    return 0;
}
