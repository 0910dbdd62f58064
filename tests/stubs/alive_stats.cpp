// Measurement aid (test infrastructure: it links the oracle restatement): how narrow is the set of cells that can lie on a path of score
// >= S_guess in POA #1 (noisy reads against the growing graph)?  alive(i, j): H[i][j] + 5 * min(C - c_i, L - j) >= S_guess (any path through a
// cell that fails it scores less than S_guess, so an exact DP may skip it).  Reports per alignment the row extents of the alive cells for
// guesses of 1.0 / 1.15 / 1.4 x the true score deficit (profiles/round6_poa1_alive_cells.txt).
// build: g++ -O2 -mavx2 -std=c++17 -o alive_stats tests/stubs/alive_stats.cpp ; usage: alive_stats LEN DEPTH
#include <cstdio>
#include <random>
#include "../../oracle/orc_correct.hpp"
using namespace orc;
int main(int argc, char **argv) {
    const int LEN = argc > 1 ? atoi(argv[1]) : 1000, DEPTH = argc > 2 ? atoi(argv[2]) : 200;
    const double ERR = 0.10;
    std::mt19937_64 rng(7);
    auto rnd = [&]() { return (rng() >> 11) * (1.0 / 9007199254740992.0); };
    const char *ACGT = "ACGT";
    std::string tx;
    for (int i = 0; i < LEN; ++i) tx += ACGT[rng() & 3];
    std::vector<std::string> reads;
    for (int d = 0; d < DEPTH; ++d) {
        std::string s;
        const size_t cut = (size_t)(rnd() * 0.10 * LEN);
        for (size_t i = cut; i < tx.size(); ++i) {
            const char c = tx[i];
            const double r = rnd();
            if (r < 0.3 * ERR) continue;
            if (r < 0.7 * ERR) s += ACGT[rng() & 3]; else s += c;
            if (rnd() < 0.3 * ERR) s += ACGT[rng() & 3];
        }
        reads.push_back(s);
    }
    std::sort(reads.begin(), reads.end(), [](const std::string &a, const std::string &b) { return a.size() > b.size(); });
    poa_graph_t G; poa_engine_t E;
    const double slacks[3] = {1.0, 1.15, 1.4};
    for (int d = 0; d < DEPTH; ++d) {
        const std::string &s = reads[d];
        if (!G.nodes.empty() && (d % 20 == 19 || d < 4)) {
            const size_t n = G.nodes.size(), L = s.size(), W = L + 1;
            E.simd = false;
            E.align(s, G);
            std::vector<uint32_t> col(n + 1, 0);
            uint32_t C = 0;
            for (size_t i = 0; i < n; ++i) { const uint32_t v = G.rank_to_node[i]; ++C; col[i + 1] = C; for (size_t k = 0; k < G.nodes[v].aligned.size(); ++k) { ++i; col[i + 1] = C; } }
            int32_t S = 0;
            for (size_t r = 1; r <= n; ++r) for (size_t j = 1; j <= L; ++j) S = std::max(S, E.H[r * W + j]);
            const double D = 5.0 * L - S;
            printf("aln %3d: rows %5zu cols %5u L %4zu S %5d deficit %.0f (%.2f per base) |", d, n, C, L, S, D, D / L);
            for (double sl : slacks) {
                const double Sg = 5.0 * L - D * sl;
                double sumw = 0, alive = 0; int maxw = 0, rows_alive = 0; std::vector<int> ws;
                for (size_t r = 1; r <= n; ++r) {
                    int lo = -1, hi = -1;
                    for (size_t j = 1; j <= L; ++j) if (E.H[r * W + j] + 5.0 * std::min<double>(C - col[r], L - j) >= Sg) { if (lo < 0) lo = (int)j; hi = (int)j; ++alive; }
                    if (lo >= 0) { ++rows_alive; ws.push_back(hi - lo + 1); sumw += hi - lo + 1; maxw = std::max(maxw, hi - lo + 1); }
                }
                std::sort(ws.begin(), ws.end());
                printf("  guess x%.2f: alive rows %d, extent mean %.0f p99 %d max %d, cells %.3f of full |", sl, rows_alive, rows_alive ? sumw / rows_alive : 0, ws.empty() ? 0 : ws[ws.size() * 99 / 100], maxw, alive / ((double)n * L));
            }
            printf("\n");
        }
        E.simd = true;
        poa_alignment_t a = E.align(s, G);
        G.add_alignment(a, s);
    }
    return 0;
}
