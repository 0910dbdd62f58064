// Measurement aid (test infrastructure: it links the oracle restatement): how wide is the row DAG of a POA pack, and what does dealing
// the rows round-robin to T teams of wavefronts buy (kernel C, dp_rows_mt)?  Builds a pack of DEPTH noisy copies of a random
// transcript with the oracle, takes the row plan of every alignment (rows = graph nodes in topological order, predecessors = in-edges)
// and replays it on an idealised machine: a tile (row, column block) takes one time unit, a wavefront runs its rows in order.
// build: g++ -O2 -mavx2 -std=c++17 -o dag_width_sim tests/stubs/dag_width_sim.cpp ; usage: dag_width_sim LEN DEPTH
#include <immintrin.h>
#include <random>
#include <map>
#include <cstdio>
#include "../../oracle/orc_poa.hpp"
#include <random>
#include <cstdio>
#include <map>
using namespace orc;
int main(int argc, char **argv) {
    int LEN = argc > 1 ? atoi(argv[1]) : 1000, DEPTH = argc > 2 ? atoi(argv[2]) : 200; double ERR = 0.10;
    std::mt19937_64 rng(5);
    auto rnd = [&]() { return (rng() >> 11) * (1.0 / 9007199254740992.0); };
    const char *ACGT = "ACGT";
    std::string tx; for (int i = 0; i < LEN; ++i) tx += ACGT[rng() & 3];
    std::vector<std::string> reads;
    for (int d = 0; d < DEPTH; ++d) {
        std::string s;
        for (char c : tx) {
            double r = rnd();
            if (r < 0.3 * ERR) continue;                       // deletion
            if (r < 0.7 * ERR) s += ACGT[rng() & 3]; else s += c;
            if (rnd() < 0.3 * ERR) s += ACGT[rng() & 3];     // insertion
        }
        reads.push_back(s);
    }
    std::sort(reads.begin(), reads.end(), [](const std::string &a, const std::string &b) { return a.size() > b.size(); });
    poa_graph_t G; poa_engine_t E; E.simd = true;
    const int Ts[] = {1, 2, 3, 4, 8};
    const int NW = 4;
    double tot_rows = 0, tot_cp = 0, mk[5] = {0}, mkio[5] = {0};
    std::map<int, double> nin_hist, dmin_hist;
    for (int d = 0; d < DEPTH; ++d) {
        if (d > 0) {
            size_t n = G.nodes.size();
            std::vector<uint32_t> rank(n);
            for (size_t r = 0; r < n; ++r) rank[G.rank_to_node[r]] = (uint32_t)r;
            std::vector<std::vector<uint32_t>> P(n + 1);
            std::vector<uint32_t> depth(n + 1, 0);
            uint32_t cp = 0;
            for (size_t r = 1; r <= n; ++r) {
                const poa_node_t &nd = G.nodes[G.rank_to_node[r - 1]];
                uint32_t dm = 1000;
                for (uint32_t ei : nd.in_edges) { uint32_t p = rank[G.edges[ei].begin] + 1; P[r].push_back(p); depth[r] = std::max(depth[r], depth[p]); dm = std::min<uint32_t>(dm, r - p); }
                depth[r] += 1; cp = std::max(cp, depth[r]);
                nin_hist[std::min<int>(nd.in_edges.size(), 9)] += 1; dmin_hist[std::min<int>(dm, 9)] += 1;
            }
            tot_rows += n; tot_cp += cp;
            // schedule simulation: tile latency 1.0, mailbox published at 0.6 of a tile
            for (int ti = 0; ti < 5; ++ti) {
                const int T = Ts[ti];
                for (int inorder = 0; inorder < 2; ++inorder) {
                    std::vector<double> endp((n + 1) * NW, 0.0), mid((n + 1) * NW, 0.0);
                    std::vector<double> wave_free(T * NW, 0.0);
                    double last = 0;
                    for (size_t r = 1; r <= n; ++r) {
                        int t = (r - 1) % T;
                        for (int w = 0; w < NW; ++w) {
                            double st = wave_free[t * NW + w];
                            for (uint32_t p : P[r]) st = std::max(st, endp[p * NW + w]);
                            double fin = st + 1.0;
                            if (w > 0) fin = std::max(fin, mid[r * NW + w - 1] + 0.4);   // the part behind the left neighbour's prefix
                            mid[r * NW + w] = std::max(st + 0.6, w > 0 ? mid[r * NW + w - 1] + 0.05 : 0.0);
                            double pub = fin;
                            if (inorder) pub = std::max(pub, endp[(r - 1) * NW + w]);
                            endp[r * NW + w] = pub;
                            wave_free[t * NW + w] = pub;
                            last = std::max(last, pub);
                        }
                    }
                    (inorder ? mkio : mk)[ti] += last;
                }
            }
        }
        auto aln = E.align(reads[d], G);
        G.add_alignment(aln, reads[d]);
    }
    printf("len %d depth %d: rows %.0f, critical path %.0f (%.3f of rows) -> average width %.2f\n", LEN, DEPTH, tot_rows, tot_cp, tot_cp / tot_rows, tot_rows / tot_cp);
    for (int ti = 0; ti < 5; ++ti) printf("  T=%d teams x %d blocks: makespan / rows = %.3f (out-of-order publish) %.3f (in-order publish)  [tile units]\n", Ts[ti], NW, mk[ti] / tot_rows, mkio[ti] / tot_rows);
    printf("  in-degree histogram:"); for (auto &kv : nin_hist) printf(" %d:%.3f", kv.first, kv.second / tot_rows); printf("\n");
    printf("  nearest predecessor distance:"); for (auto &kv : dmin_hist) printf(" %d:%.3f", kv.first, kv.second / tot_rows); printf("\n");
}
