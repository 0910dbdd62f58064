// Measurement aid (test infrastructure: it links the oracle restatement): a model of the ALIVE WINDOWS of kernel C for POA #1 (noisy reads
// against the growing graph of their pack).  A cell can lie on a path of score >= S_guess only if
//     alive(i, j):  H[i][j] >= max(S_guess - 5 (C - c_i), S_guess - 5 (L - j))
// (c_i = column of row i, C = columns of the graph), the set of dead cells is closed under every move, and a DP that leaves dead cells out is
// exact when the score it finds is >= S_guess.  The alive cells of a row are two islands: (A) the first W_A columns while an alignment may
// still START in the row (a deep graph holds every letter in most columns: ANY sequence scores ~4 per base against it, so the cells with
// 5 j - (5 L - S_guess) <= ~4 j stay alive), and (B) ~100 columns left of the true path.  Two passes of one wavefront:
//   pass A: columns 1 .. W_A = 64 CPLB_A of the rows with S_guess - 5 (C - c_i) <= 5 W_A (below them every cell of these columns is dead),
//   pass B: 64 CPLB_B columns per row, placed BEFORE the rows run: where the previous read of the pack went through the graph (its bases'
//           nodes -> columns -> a monotone map column -> j), shifted by where pass A saw the new path leave its last column.
// Checks that make it exact (all on the computed cells): no alive cell in a row's lanes that a successor's window has left behind, none in
// the last lane, none in the last column of pass A unless pass B's window is still adjacent there and in every successor row.
// The model runs this on the TRUE matrices of the oracle (the pruned DP equals them on every alive cell as long as no check fails) and
// counts failed checks and cells; it also verifies the implication (checks pass => no alive cell outside the windows).
// build: g++ -O2 -mavx2 -std=c++17 -o /tmp/awm tests/stubs/alive_window_model.cpp ; usage: awm LEN DEPTH [SLACK=1.4] [LEFT=150] [ISL_EXTRA=16] [ISLDIV=1.0] [CPLB_B=4]
#include <cstdio>
#include <random>
#include "../../oracle/orc_correct.hpp"
using namespace orc;

struct tally_t { double alns = 0, ok = 0, wide = 0, v_lo = 0, v_hi = 0, v_a = 0, v_nox = 0, v_score = 0, bug = 0, cells = 0, full = 0, rowsA = 0, rows = 0, cplba[9] = {0}; };

int main(int argc, char **argv) {
    const int LEN = argc > 1 ? atoi(argv[1]) : 1000, DEPTH = argc > 2 ? atoi(argv[2]) : 200;
    const double SLACK = argc > 3 ? atof(argv[3]) : 1.4;
    const int LEFT = argc > 4 ? atoi(argv[4]) : 150;
    const int ISL_EXTRA = argc > 5 ? atoi(argv[5]) : 16;
    const double ISLDIV = argc > 6 ? atof(argv[6]) : 1.0;
    const int CB = argc > 7 ? atoi(argv[7]) : 4;
    const int VERB = getenv("AWM_VERBOSE") ? atoi(getenv("AWM_VERBOSE")) : 0;
    const double ERR = 0.10;
    std::mt19937_64 rng(getenv("AWM_SEED") ? atoi(getenv("AWM_SEED")) : 7);
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
    tally_t T;
    double dpb1 = 2.0, dpb2 = 2.0;            // score deficit per base of the last two alignments
    for (int d = 0; d < DEPTH; ++d) {
        const std::string &s = reads[d];
        if (!G.nodes.empty()) {
            const size_t n = G.nodes.size(), L = s.size(), W = L + 1;
            E.simd = false;
            E.align(s, G);
            std::vector<uint32_t> rank(n), col(n + 1, 0);
            for (size_t r = 0; r < n; ++r) rank[G.rank_to_node[r]] = (uint32_t)r;
            uint32_t C = 0;
            for (size_t i = 0; i < n; ++i) { const uint32_t v = G.rank_to_node[i]; ++C; col[i + 1] = C; for (size_t k = 0; k < G.nodes[v].aligned.size(); ++k) { ++i; col[i + 1] = C; } }
            int32_t S = 0;
            for (size_t r = 1; r <= n; ++r) for (size_t j = 1; j <= L; ++j) S = std::max(S, E.H[r * W + j]);
            const double D = 5.0 * L - S;
            T.alns += 1; T.full += (double)n * L;
            const double Gd = SLACK * std::max(dpb1, dpb2) * L + 30.0;
            const int32_t Sg = (int32_t)std::max(1.0, 5.0 * L - Gd);
            dpb2 = dpb1; dpb1 = D / L;
            // the previous read's way through the graph: its bases' nodes, by the edges that carry its label
            std::vector<int> jcol(C + 2, 0);
            {
                const uint32_t sid = G.num_sequences - 1;
                uint32_t v = G.seq_begin[sid];
                int j = 1;
                for (;;) {
                    jcol[col[rank[v] + 1]] = j++;
                    uint32_t nx = UINT32_MAX;
                    for (uint32_t ei : G.nodes[v].out_edges) { const auto &lb = G.edges[ei].labels; if (std::find(lb.begin(), lb.end(), sid) != lb.end()) { nx = G.edges[ei].end; break; } }
                    if (nx == UINT32_MAX) break;
                    v = nx;
                }
                for (uint32_t c = 1; c <= C; ++c) jcol[c] = std::max(jcol[c], jcol[c - 1]);
            }
            auto alive = [&](size_t r, int j) -> bool {
                if (j < 1 || j > (int)L) return false;
                const int tr = Sg - 5 * (int)(C - col[r]), tc = Sg - 5 * (int)(L - j);
                return E.H[r * W + j] >= std::max(tr, tc);
            };
            const int wa_need = (int)((5.0 * L - Sg) / ISLDIV) + ISL_EXTRA;
            int ca = 0;
            for (int c : {2, 4, 8}) if (64 * c >= wa_need) { ca = c; break; }
            if (!ca || (int)L <= 64 * ca + 64) { T.wide += 1; T.cells += (double)n * L; if (VERB) printf("aln %3d: pass A would need %d columns: full rows\n", d, wa_need); }
            else {
                const int WA = 64 * ca, GA = WA / CB;                  // pass B's first group when adjacent to pass A
                size_t iA = 0;
                for (size_t r = 1; r <= n; ++r) if (Sg - 5 * (int)(C - col[r]) <= 5 * WA) iA = r;
                // where the path leaves pass A: the row with the largest H among the alive cells of column WA
                int hx = -1; size_t rx = 0;
                for (size_t r = 1; r <= iA; ++r) if (alive(r, WA) && E.H[r * W + WA] > hx) { hx = E.H[r * W + WA]; rx = r; }
                bool f_nox = rx == 0, f_lo = false, f_hi = false, f_a = false, bug = false;
                const int delta = rx ? WA - jcol[col[rx]] : 0;
                std::vector<int> g0(n + 1, GA);
                const int gmax = std::max(GA, (int)((L + CB - 1) / CB) - 64 + 1);
                for (size_t r = 1; r <= n; ++r) { int g = (jcol[col[r]] + delta - LEFT) / CB; g0[r] = std::min(std::max(GA, g), gmax); }
                // (monotone in the column; rows of one column share it)
                std::vector<int> need(n + 1, 0), adj_succ(n + 1, 1);
                for (size_t r = 1; r <= n; ++r) for (uint32_t ei : G.nodes[G.rank_to_node[r - 1]].in_edges) { const size_t p = rank[G.edges[ei].begin] + 1; need[p] = std::max(need[p], g0[r] - g0[p]); if (g0[r] != GA) adj_succ[p] = 0; }
                for (size_t r = 1; r <= n && !f_nox; ++r) {
                    const int a = g0[r] * CB + 1, b = (g0[r] + 64) * CB;
                    for (int j = WA + 1; j <= (int)L; ++j) if (alive(r, j)) {
                        if (j < a || j > b) { if (VERB > 1) printf("   row %zu col %u: alive at %d outside %d..%d\n", r, col[r], j, a, b); bug = true; }
                        else { const int lane = (j - a) / CB; if (lane < need[r]) f_lo = true; if (lane == 63 && b < (int)L) f_hi = true; }
                    }
                    if (r <= iA && alive(r, WA) && (g0[r] != GA || !adj_succ[r])) f_a = true;
                    if (r > iA) for (int j = 1; j <= WA; ++j) if (alive(r, j)) bug = true;
                }
                if (bug && !(f_lo || f_hi || f_a || f_nox)) { T.bug += 1; printf("aln %3d: MODEL BUG: alive cells outside the windows and no check failed\n", d); }
                const bool score_ok = S >= Sg;
                const bool ok = score_ok && !f_lo && !f_hi && !f_a && !f_nox;
                T.ok += ok; T.v_lo += f_lo; T.v_hi += f_hi; T.v_a += f_a; T.v_nox += f_nox; T.v_score += !score_ok;
                T.cells += (double)iA * WA + (double)n * 64 * CB + (ok ? 0.0 : (double)n * L); T.rowsA += iA; T.rows += n; T.cplba[ca] += 1;
                if (VERB || !ok) printf("aln %3d: n %zu L %zu C %u S %d (deficit %.2f / base) guess %d: pass A %d columns x %zu rows, crossing at row %zu (column %u), delta %d: %s%s%s%s%s%s\n", d, n, L, C, S, D / L, Sg,
                                        WA, iA, rx, rx ? col[rx] : 0, delta, ok ? "ok" : "FAIL", f_lo ? " lo" : "", f_hi ? " hi" : "", f_a ? " A-edge" : "", f_nox ? " no-crossing" : "", score_ok ? "" : " score");
            }
        }
        E.simd = true;
        poa_alignment_t a = E.align(s, G);
        G.add_alignment(a, s);
    }
    printf("LEN %d DEPTH %d slack %.2f left %d: %0.f alignments, %0.f windowed ok, %0.f full rows (pass A too wide); failed: lo %0.f hi %0.f A-edge %0.f no-crossing %0.f score %0.f; model bugs %0.f\n", LEN, DEPTH, SLACK,
           LEFT, T.alns, T.ok, T.wide, T.v_lo, T.v_hi, T.v_a, T.v_nox, T.v_score, T.bug);
    printf("   cells computed (failures pay windows + full rows) %.3g of %.3g = %.3f; pass A rows %.2f of the rows, columns per lane 2: %0.f, 4: %0.f, 8: %0.f\n", T.cells, T.full, T.cells / T.full,
           T.rowsA / std::max(1.0, T.rows), T.cplba[2], T.cplba[4], T.cplba[8]);
    return 0;
}
