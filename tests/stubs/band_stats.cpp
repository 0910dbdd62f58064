// Measurement aid (test infrastructure: it links the oracle restatement): how narrow can an EXACT band be for the near-chain
// alignments of `rattle correct` -- POA #2 (corrected reads of a pack, correct.cpp:427-436) and POA #3 (pack consensi of a
// cluster, correct.cpp:520-532)?
// A local alignment path through cell (v, j) has at most M(v, j) = min(a(v), j) + min(b(v) - 1, L - j) diagonal moves, a(v) / b(v)
// upper bounds of the nodes on a path that ends / starts at v; a path with fewer than tau diagonal moves scores at most
// 5 (tau - 1).  So the band {M >= tau} is exact whenever the best score found inside it is >= 5 tau - 4 (kernel C).
// Three bounds for a / b are compared: the row index in topological order, the MSA column of the row, the exact longest path.
// build: g++ -O2 -mavx2 -std=c++17 -o band_stats tests/stubs/band_stats.cpp ; usage: band_stats LEN DEPTH PACKS
#include <cstdio>
#include <random>
#include <map>
#include "../../oracle/orc_correct.hpp"
using namespace orc;

struct stat_t { double n = 0, sum_n = 0, sum_L = 0, sum_cols = 0, sum_def = 0, sum_wA = 0, sum_wB = 0, sum_wC = 0, max_wA = 0, max_wB = 0, max_wC = 0, sum_path_lo = 0, sum_path_hi = 0, cells_full = 0, cells_C = 0, cells_prune = 0, max_prune_w = 0, sum_prune_w = 0; };

static void one_alignment(poa_graph_t &G, poa_engine_t &E, const std::string &s, stat_t &st, FILE *dump) {
    const size_t n = G.nodes.size(), L = s.size();
    if (n == 0) return;
    std::vector<uint32_t> rank(n);
    for (size_t r = 0; r < n; ++r) rank[G.rank_to_node[r]] = (uint32_t)r;
    // columns in spoa's own order
    std::vector<uint32_t> col(n + 1, 0);
    uint32_t C = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t v = G.rank_to_node[i];
        ++C;
        col[i + 1] = C;
        for (size_t k = 0; k < G.nodes[v].aligned.size(); ++k) { ++i; col[i + 1] = C; }
    }
    // exact longest paths (in nodes) ending / starting at each row
    std::vector<uint32_t> a(n + 1, 0), b(n + 2, 0);
    for (size_t r = 1; r <= n; ++r) {
        const poa_node_t &nd = G.nodes[G.rank_to_node[r - 1]];
        uint32_t m = 0;
        for (uint32_t ei : nd.in_edges) m = std::max(m, a[rank[G.edges[ei].begin] + 1]);
        a[r] = m + 1;
    }
    for (size_t r = n; r >= 1; --r) {
        const poa_node_t &nd = G.nodes[G.rank_to_node[r - 1]];
        uint32_t m = 0;
        for (uint32_t ei : nd.out_edges) m = std::max(m, b[rank[G.edges[ei].end] + 1]);
        b[r] = m + 1;
    }
    E.simd = false;
    poa_alignment_t aln = E.align(s, G);
    const size_t W = L + 1;
    int32_t S = 0;
    for (size_t r = 1; r <= n; ++r) for (size_t j = 1; j <= L; ++j) S = std::max(S, E.H[r * W + j]);
    if (S == 0) return;
    const int64_t tau = (S + 4) / 5;
    // band widths: rows with plateau >= tau
    auto width = [&](int64_t av, int64_t bv) -> int64_t {
        const int64_t P = av + bv - 1;
        if (std::min<int64_t>(P, (int64_t)L) < tau) return 0;
        const int64_t lo = std::max<int64_t>(1, tau - bv + 1), hi = std::min<int64_t>((int64_t)L, av + (int64_t)L - tau);
        return hi >= lo ? hi - lo + 1 : 0;
    };
    int64_t wA = 0, wB = 0, wC = 0, cellsC = 0;
    for (size_t r = 1; r <= n; ++r) {
        wA = std::max(wA, width((int64_t)r, (int64_t)(n - r + 1)));
        wB = std::max(wB, width((int64_t)col[r], (int64_t)(C - col[r] + 1)));
        const int64_t w = width((int64_t)a[r], (int64_t)b[r]);
        wC = std::max(wC, w); cellsC += w;
    }
    // the optimal path's offset j - a(v) range
    int64_t plo = 1 << 30, phi = -(1 << 30);
    for (auto &p : aln) if (p.first >= 0 && p.second >= 0) { const int64_t d = (int64_t)p.second + 1 - (int64_t)a[rank[p.first] + 1]; plo = std::min(plo, d); phi = std::max(phi, d); }
    // exact pruning with the true H: cells with H + 5 min(b - 1, L - j) >= S
    int64_t kept = 0, maxw = 0, sumw = 0;
    for (size_t r = 1; r <= n; ++r) {
        int64_t lo = -1, hi = -1;
        for (size_t j = 1; j <= L; ++j) {
            const int64_t ub = (int64_t)E.H[r * W + j] + 5 * std::min<int64_t>((int64_t)b[r] - 1, (int64_t)(L - j));
            if (ub >= S) { if (lo < 0) lo = (int64_t)j; hi = (int64_t)j; ++kept; }
        }
        if (lo >= 0) { maxw = std::max(maxw, hi - lo + 1); sumw += hi - lo + 1; }
    }
    st.n += 1; st.sum_n += n; st.sum_L += L; st.sum_cols += C; st.sum_def += 5.0 * L - S;
    st.sum_wA += wA; st.sum_wB += wB; st.sum_wC += wC; st.max_wA = std::max<double>(st.max_wA, wA); st.max_wB = std::max<double>(st.max_wB, wB); st.max_wC = std::max<double>(st.max_wC, wC);
    st.sum_path_lo += plo; st.sum_path_hi += phi; st.cells_full += (double)n * L; st.cells_C += cellsC; st.cells_prune += kept;
    st.max_prune_w = std::max<double>(st.max_prune_w, maxw); st.sum_prune_w += (double)sumw / n;
    if (dump) fprintf(dump, "n %zu L %zu cols %u S %d deficit %ld tau %ld  maxP %u  width rank %ld col %ld path %ld  opt-offset [%ld, %ld]  prune maxw %ld\n", n, L, C, S, (long)(5 * L - S), (long)tau,
                      *std::max_element(a.begin(), a.end()), (long)wA, (long)wB, (long)wC, (long)plo, (long)phi, (long)maxw);
}

static void report(const char *name, const stat_t &s) {
    if (s.n == 0) return;
    printf("%s: %0.f alignments, mean nodes %.0f, L %.0f, columns %.0f, score deficit 5L-S %.1f\n", name, s.n, s.sum_n / s.n, s.sum_L / s.n, s.sum_cols / s.n, s.sum_def / s.n);
    printf("   exact band width (max over rows), mean / max over alignments: rank bound %.0f / %.0f, column bound %.0f / %.0f, longest-path bound %.0f / %.0f\n",
           s.sum_wA / s.n, s.max_wA, s.sum_wB / s.n, s.max_wB, s.sum_wC / s.n, s.max_wC);
    printf("   cells: full %.3g, longest-path band %.3g (%.3f), exact pruning with true H %.3g (%.3f; max row extent %.0f, mean %.1f)\n", s.cells_full, s.cells_C, s.cells_C / s.cells_full, s.cells_prune,
           s.cells_prune / s.cells_full, s.max_prune_w, s.sum_prune_w / s.n);
    printf("   optimal path offset j - a(v): mean lo %.1f hi %.1f\n", s.sum_path_lo / s.n, s.sum_path_hi / s.n);
}

int main(int argc, char **argv) {
    const int LEN = argc > 1 ? atoi(argv[1]) : 1000, DEPTH = argc > 2 ? atoi(argv[2]) : 200, PACKS = argc > 3 ? atoi(argv[3]) : 4;
    const double ERR = 0.10;
    std::mt19937_64 rng(7);
    auto rnd = [&]() { return (rng() >> 11) * (1.0 / 9007199254740992.0); };
    const char *ACGT = "ACGT";
    std::string tx;
    for (int i = 0; i < LEN; ++i) tx += ACGT[rng() & 3];
    stat_t s2, s3;
    read_set_t consensi;
    FILE *dump = getenv("BAND_DUMP") ? stderr : nullptr;
    for (int p = 0; p < PACKS; ++p) {
        read_set_t reads;
        for (int d = 0; d < DEPTH; ++d) {
            std::string s, q;
            const size_t cut = (size_t)(rnd() * 0.10 * LEN);          // 5' truncation U[0, 10 %]
            for (size_t i = cut; i < tx.size(); ++i) {
                const char c = tx[i];
                const double r = rnd();
                if (r < 0.3 * ERR) continue;
                if (r < 0.7 * ERR) s += ACGT[rng() & 3]; else s += c;
                if (rnd() < 0.3 * ERR) s += ACGT[rng() & 3];
            }
            for (size_t i = 0; i < s.size(); ++i) { int ql = (int)std::lround(10 + 3 * (rnd() + rnd() + rnd() + rnd() - 2) * 1.7); ql = std::max(3, std::min(40, ql)); q += (char)(33 + ql); }
            reads.push_back(read_t{"@r" + std::to_string(d) + ",x", s, "+", q});
        }
        sort_read_set(reads);
        std::vector<std::string> seqs;
        for (auto &r : reads) seqs.push_back(r.seq);
        poa_simd_default() = true;
        msa_t msa = poa_msa(seqs);
        fix_msa_ends(reads, msa);
        corrected_pack_t cp = correct_read_pack(reads, msa, 0.3, 0.3, 30.0);
        read_set_t corrected = cp.reads;
        sort_read_set(corrected);
        // POA #2 with statistics
        poa_graph_t G; poa_engine_t E;
        for (auto &r : corrected) {
            if (!G.nodes.empty()) one_alignment(G, E, r.seq, s2, dump);
            E.simd = true;
            poa_alignment_t a = E.align(r.seq, G);
            G.add_alignment(a, r.seq);
        }
        msa_t m2 = G.msa();
        fix_msa_ends(corrected, m2);
        consensus_vector_t cv = generate_consensus_vector(corrected, m2);
        const std::string cons = strip_gaps(cv.consensus_nt);
        consensi.push_back(read_t{"c", cons, "+", std::string(cons.size(), 'K')});
        fprintf(stderr, "pack %d: %zu corrected reads, consensus %zu nt\n", p, corrected.size(), cons.size());
    }
    {
        poa_graph_t G; poa_engine_t E;
        for (auto &r : consensi) {
            if (!G.nodes.empty()) one_alignment(G, E, r.seq, s3, dump);
            E.simd = true;
            poa_alignment_t a = E.align(r.seq, G);
            G.add_alignment(a, r.seq);
        }
    }
    report("POA #2 (corrected reads of a pack)", s2);
    report("POA #3 (pack consensi)", s3);
    return 0;
}
