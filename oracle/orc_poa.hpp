// ORACLE (test infrastructure, never shipped in the product path).
//
// CPU restatement of the partial-order-alignment library RATTLE calls from
// correct.cpp:395-405,428-436,520-532:
//     spoa::createAlignmentEngine(AlignmentType(0)=kSW, m=5, n=-4, g=-8, e=-6)
//     engine->align(seq, graph); graph->add_alignment(aln, seq);
//     graph->generate_multiple_sequence_alignment(msa)
//
// The library is rvaser/spoa, an UN-VENDORED and UNPINNED git submodule
// (/root/reference/.gitmodules:4-6; /root/reference/spoa is empty, no gitlink
// survives).  The API shape used by correct.cpp (free function
// createAlignmentEngine with 5 arguments, createGraph, add_alignment,
// generate_multiple_sequence_alignment) bounds it to spoa >= 3.0.0 and < 4.0.0.
// What follows restates that release line's published algorithm (Lee, Grasso &
// Sharlow 2002 POA with affine gaps as implemented by spoa's scalar engine):
// Smith-Waterman sequence-to-DAG DP in topological order, first-maximum best
// cell, traceback order diagonal -> vertical -> horizontal with predecessors
// tried in in-edge insertion order, graph update with aligned-node groups, DFS
// topological sort that ranks aligned groups consecutively, MSA by rank.
// spoa's SIMD engine computes the same H/E/F values and uses the same
// traceback order, so the scalar statement stands for both.
//
// Pinning: there is no spoa source or test vector under /root/reference.  The
// only anchor is the end-to-end fixture toyset/rna/output/consensi.fq (175
// consensus sequences produced by an older RATTLE + unknown spoa commit);
// tests/test_oracle_correct.py reports how many of them this restatement
// reproduces exactly.  Beyond that fixture: PARITY UNPINNED.
#pragma once
#include <algorithm>
#include <climits>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace orc {

typedef std::vector<std::pair<int32_t, int32_t>> poa_alignment_t;   // (node_id | -1, seq_pos | -1)

struct poa_edge_t {
    uint32_t begin, end;
    std::vector<uint32_t> labels;      // sequence ids that traverse the edge
};

struct poa_node_t {
    char letter;
    std::vector<uint32_t> in_edges, out_edges;   // indices into graph.edges, insertion order
    std::vector<uint32_t> aligned;               // aligned_nodes_ids
};

struct poa_graph_t {
    std::vector<poa_node_t> nodes;
    std::vector<poa_edge_t> edges;
    std::vector<uint32_t> rank_to_node;
    std::vector<uint32_t> seq_begin;             // sequences_begin_nodes_ids
    uint32_t num_sequences = 0;
    uint64_t dp_cells = 0;                       // exact work counter (not in spoa)

    uint32_t add_node(char c) {
        nodes.push_back(poa_node_t{c, {}, {}, {}});
        return (uint32_t)nodes.size() - 1;
    }

    // Graph::add_edge: augment an existing begin->end edge, else append a new one to
    // begin.out_edges and end.in_edges.
    void add_edge(uint32_t b, uint32_t e) {
        for (uint32_t ei : nodes[b].out_edges) {
            if (edges[ei].end == e) { edges[ei].labels.push_back(num_sequences); return; }
        }
        edges.push_back(poa_edge_t{b, e, {num_sequences}});
        uint32_t id = (uint32_t)edges.size() - 1;
        nodes[b].out_edges.push_back(id);
        nodes[e].in_edges.push_back(id);
    }

    // Graph::add_sequence(sequence, begin, end): fresh chain, returns first node or -1.
    int32_t add_sequence(const std::string &s, uint32_t b, uint32_t e) {
        if (b == e) return -1;
        uint32_t first = add_node(s[b]);
        for (uint32_t i = b + 1; i < e; ++i) {
            uint32_t id = add_node(s[i]);
            add_edge(id - 1, id);
        }
        return (int32_t)first;
    }

    // Graph::topological_sort: iterative DFS over in-edges from node ids 0..n-1; a node's
    // aligned group is visited with it and receives consecutive ranks.
    void topological_sort() {
        rank_to_node.clear();
        size_t n = nodes.size();
        std::vector<uint8_t> marks(n, 0);
        std::vector<uint8_t> check_aligned(n, 1);
        std::vector<uint32_t> stack;
        for (uint32_t i = 0; i < n; ++i) {
            if (marks[i] != 0) continue;
            stack.push_back(i);
            while (!stack.empty()) {
                uint32_t v = stack.back();
                bool valid = true;
                if (marks[v] != 2) {
                    for (uint32_t ei : nodes[v].in_edges) {
                        uint32_t b = edges[ei].begin;
                        if (marks[b] != 2) { stack.push_back(b); valid = false; }
                    }
                    if (check_aligned[v]) {
                        for (uint32_t a : nodes[v].aligned) {
                            if (marks[a] != 2) { stack.push_back(a); check_aligned[a] = 0; valid = false; }
                        }
                    }
                    if (valid) {
                        marks[v] = 2;
                        if (check_aligned[v]) {
                            rank_to_node.push_back(v);
                            for (uint32_t a : nodes[v].aligned) rank_to_node.push_back(a);
                        }
                    } else {
                        marks[v] = 1;
                    }
                }
                if (valid) stack.pop_back();
            }
        }
    }

    // Graph::add_alignment(alignment, sequence) with unit weights (weights do not reach RATTLE).
    void add_alignment(const poa_alignment_t &aln, const std::string &s) {
        uint32_t L = (uint32_t)s.size();
        if (L == 0) return;
        if (aln.empty()) {
            int32_t b = add_sequence(s, 0, L);
            ++num_sequences;
            seq_begin.push_back((uint32_t)b);
            topological_sort();
            return;
        }
        std::vector<uint32_t> valid;
        for (auto &p : aln) if (p.second != -1) valid.push_back((uint32_t)p.second);
        uint32_t before = (uint32_t)nodes.size();
        int32_t begin_node = add_sequence(s, 0, valid.front());
        int32_t head = before == nodes.size() ? -1 : (int32_t)nodes.size() - 1;
        int32_t tail = add_sequence(s, valid.back() + 1, L);
        int32_t cur = -1;
        for (size_t i = 0; i < aln.size(); ++i) {
            if (aln[i].second == -1) continue;
            char letter = s[aln[i].second];
            if (aln[i].first == -1) {
                cur = (int32_t)add_node(letter);
            } else {
                uint32_t an = (uint32_t)aln[i].first;
                if (nodes[an].letter == letter) {
                    cur = (int32_t)an;
                } else {
                    int32_t hit = -1;
                    for (uint32_t a : nodes[an].aligned) {
                        if (nodes[a].letter == letter) { hit = (int32_t)a; break; }
                    }
                    if (hit == -1) {
                        cur = (int32_t)add_node(letter);
                        std::vector<uint32_t> grp = nodes[an].aligned;     // copy: nodes may reallocate
                        for (uint32_t a : grp) {
                            nodes[cur].aligned.push_back(a);
                            nodes[a].aligned.push_back((uint32_t)cur);
                        }
                        nodes[cur].aligned.push_back(an);
                        nodes[an].aligned.push_back((uint32_t)cur);
                    } else {
                        cur = hit;
                    }
                }
            }
            if (begin_node == -1) begin_node = cur;
            if (head != -1) add_edge((uint32_t)head, (uint32_t)cur);
            head = cur;
        }
        if (tail != -1) add_edge((uint32_t)head, (uint32_t)tail);
        ++num_sequences;
        seq_begin.push_back((uint32_t)begin_node);
        topological_sort();
    }

    // Graph::generate_multiple_sequence_alignment(dst) without the consensus row.
    std::vector<std::string> msa() const {
        std::vector<int32_t> col(nodes.size(), -1);
        int32_t ncol = 0;
        for (size_t i = 0; i < nodes.size(); ++i) {
            uint32_t v = rank_to_node[i];
            col[v] = ncol;
            for (size_t j = 0; j < nodes[v].aligned.size(); ++j) col[rank_to_node[++i]] = ncol;
            ++ncol;
        }
        std::vector<std::string> out;
        for (uint32_t s = 0; s < num_sequences; ++s) {
            std::string row(ncol, '-');
            uint32_t cur = seq_begin[s];
            while (true) {
                row[col[cur]] = nodes[cur].letter;
                uint32_t prev = cur;
                for (uint32_t ei : nodes[prev].out_edges) {
                    for (uint32_t lab : edges[ei].labels) {
                        if (lab == s) { cur = edges[ei].end; break; }
                    }
                    if (prev != cur) break;
                }
                if (prev == cur) break;
            }
            out.push_back(row);
        }
        return out;
    }
};

#if defined(__x86_64__)
#include <immintrin.h>
#define ORC_X86 1
#endif

struct poa_engine_t {
    int m = 5, n = -4, g = -8, e = -6;          // correct.cpp:395-396
    // simd = true: the matrices are filled by an AVX2 int16 row kernel (16 columns per instruction, horizontal gaps by an
    // in-register prefix maximum) when the CPU has AVX2 and 5 * |seq| fits 16 bits -- what spoa's own SIMD engine does.
    // Same H / F / E values, same traceback: used for the CPU baseline of bench.py, checked against the scalar fill in
    // tests/test_oracle_correct.py.
    bool simd = false;
    std::vector<int32_t> H, F, E;
    std::vector<int16_t> H16, F16, E16;

    // AlignmentEngine::align for type kSW, affine gaps (g < e).
    poa_alignment_t align(const std::string &seq, poa_graph_t &G) {
        poa_alignment_t aln;
        size_t nv = G.nodes.size();
        if (nv == 0 || seq.empty()) return aln;
        G.dp_cells += (uint64_t)seq.size() * nv;
        std::vector<uint32_t> rank(nv);
        for (size_t r = 0; r < nv; ++r) rank[G.rank_to_node[r]] = (uint32_t)r;
        long bi = -1, bj = -1;
#ifdef ORC_X86
        if (simd && 5 * seq.size() + 64 < 32000 && __builtin_cpu_supports("avx2")) {
            const size_t W = ((seq.size() + 1 + 15) / 16) * 16 + 32;
            fill_avx2(seq, G, rank, W, bi, bj);
            if (bi == -1) return aln;
            return trace(seq, G, rank, H16.data(), F16.data(), E16.data(), W, (size_t)bi, (size_t)bj);
        }
#endif
        const size_t W = seq.size() + 1;
        fill_scalar(seq, G, rank, W, bi, bj);
        if (bi == -1) return aln;
        return trace(seq, G, rank, H.data(), F.data(), E.data(), W, (size_t)bi, (size_t)bj);
    }

    void fill_scalar(const std::string &seq, poa_graph_t &G, const std::vector<uint32_t> &rank, size_t W, long &bi, long &bj) {
        const int32_t NEG = INT_MIN / 2;
        size_t nv = G.nodes.size(), R = nv + 1;
        H.assign(W * R, 0); F.assign(W * R, NEG); E.assign(W * R, NEG);
        int32_t best = 0;
        for (size_t r = 0; r < nv; ++r) {
            const poa_node_t &nd = G.nodes[G.rank_to_node[r]];
            size_t i = r + 1;
            int32_t *Hr = &H[i * W], *Fr = &F[i * W], *Er = &E[i * W];
            size_t np = nd.in_edges.size();
            for (size_t pi = 0; pi < std::max<size_t>(np, 1); ++pi) {
                size_t p = np == 0 ? 0 : rank[G.edges[nd.in_edges[pi]].begin] + 1;
                const int32_t *Hp = &H[p * W], *Fp = &F[p * W];
                for (size_t j = 1; j < W; ++j) {
                    int32_t f = std::max(Hp[j] + g, Fp[j] + e);
                    int32_t h = Hp[j - 1] + (nd.letter == seq[j - 1] ? m : n);
                    if (pi == 0) { Fr[j] = f; Hr[j] = h; }
                    else { Fr[j] = std::max(Fr[j], f); Hr[j] = std::max(Hr[j], h); }
                }
            }
            for (size_t j = 1; j < W; ++j) {
                Er[j] = std::max(Hr[j - 1] + g, Er[j - 1] + e);
                Hr[j] = std::max(Hr[j], std::max(Fr[j], Er[j]));
                Hr[j] = std::max(Hr[j], 0);
                if (best < Hr[j]) { best = Hr[j]; bi = (long)i; bj = (long)j; }   // first max in (rank, col) order
            }
        }
    }

#ifdef ORC_X86
    // whole register moved up by N 16-bit lanes, the vacated lanes filled from `fill`
    template <int N>
    __attribute__((target("avx2"))) static inline __m256i shl16(__m256i x, __m256i fill) {
        const __m256i t = _mm256_permute2x128_si256(x, fill, 0x02);       // low half: fill, high half: x.low
        if (N == 8) return t;
        return _mm256_alignr_epi8(x, t, 16 - 2 * (N & 7));
    }

    __attribute__((target("avx2"))) void fill_avx2(const std::string &seq, poa_graph_t &G, const std::vector<uint32_t> &rank, size_t W, long &bi, long &bj) {
        const int16_t NEG = -32768;
        const size_t nv = G.nodes.size(), R = nv + 1, L = seq.size();
        // no clearing pass: every row is written in full below, only the borders need values (row 0: H = 0, F = -inf; column 0
        // of every row: H = 0, E = -inf)
        if (H16.size() < W * R) { H16.resize(W * R); F16.resize(W * R); E16.resize(W * R); }
        for (size_t j = 0; j < W; ++j) { H16[j] = 0; F16[j] = NEG; E16[j] = NEG; }
        // query profile: match / mismatch score of every column for the letters that occur in the graph
        std::vector<std::vector<int16_t>> prof(256);
        const __m256i vg = _mm256_set1_epi16((short)g), ve = _mm256_set1_epi16((short)e), vneg = _mm256_set1_epi16(NEG), vzero = _mm256_setzero_si256();
        const __m256i ramp = _mm256_setr_epi16((short)(1 * e), (short)(2 * e), (short)(3 * e), (short)(4 * e), (short)(5 * e), (short)(6 * e), (short)(7 * e), (short)(8 * e),
                                               (short)(9 * e), (short)(10 * e), (short)(11 * e), (short)(12 * e), (short)(13 * e), (short)(14 * e), (short)(15 * e), (short)(16 * e));
        const __m256i ve2 = _mm256_set1_epi16((short)(2 * e)), ve4 = _mm256_set1_epi16((short)(4 * e)), ve8 = _mm256_set1_epi16((short)(8 * e));
        int32_t best = 0;
        for (size_t r = 0; r < nv; ++r) {
            const poa_node_t &nd = G.nodes[G.rank_to_node[r]];
            std::vector<int16_t> &pr = prof[(unsigned char)nd.letter];
            if (pr.empty()) {
                pr.assign(W, 0);
                for (size_t j = 1; j <= L; ++j) pr[j] = (int16_t)(nd.letter == seq[j - 1] ? m : n);
            }
            const size_t i = r + 1;
            int16_t *Hr = &H16[i * W], *Fr = &F16[i * W], *Er = &E16[i * W];
            const size_t np = nd.in_edges.size();
            for (size_t pi = 0; pi < std::max<size_t>(np, 1); ++pi) {
                const size_t p = np == 0 ? 0 : rank[G.edges[nd.in_edges[pi]].begin] + 1;
                const int16_t *Hp = &H16[p * W], *Fp = &F16[p * W];
                for (size_t j = 1; j <= L; j += 16) {
                    const __m256i hp1 = _mm256_loadu_si256((const __m256i *)(Hp + j - 1)), hpj = _mm256_loadu_si256((const __m256i *)(Hp + j));
                    const __m256i fp = _mm256_loadu_si256((const __m256i *)(Fp + j));
                    __m256i f = _mm256_max_epi16(_mm256_adds_epi16(hpj, vg), _mm256_adds_epi16(fp, ve));
                    __m256i h = _mm256_adds_epi16(hp1, _mm256_loadu_si256((const __m256i *)(pr.data() + j)));
                    if (pi != 0) {
                        f = _mm256_max_epi16(f, _mm256_loadu_si256((const __m256i *)(Fr + j)));
                        h = _mm256_max_epi16(h, _mm256_loadu_si256((const __m256i *)(Hr + j)));
                    }
                    _mm256_storeu_si256((__m256i *)(Fr + j), f);
                    _mm256_storeu_si256((__m256i *)(Hr + j), h);
                }
            }
            // Hn = max(diagonal, F, 0); E[j] = max(H[j-1] + g, E[j-1] + e) = max(Hn[j-1] + g, E[j-1] + e) because e >= g:
            // a prefix maximum with a penalty of e per lane, carried from vector to vector
            int16_t carry = NEG;                 // E of the column before the vector
            Hr[0] = 0; Fr[0] = NEG; Er[0] = NEG;
            for (size_t j = 1; j <= L; j += 16) {
                __m256i hn = _mm256_max_epi16(_mm256_max_epi16(_mm256_loadu_si256((const __m256i *)(Hr + j)), _mm256_loadu_si256((const __m256i *)(Fr + j))), vzero);
                _mm256_storeu_si256((__m256i *)(Hr + j), hn);                                   // so that column j-1 of the NEXT vector reads Hn (>= its final H minus E)
                __m256i x = _mm256_adds_epi16(_mm256_loadu_si256((const __m256i *)(Hr + j - 1)), vg);      // opening a gap after column j-1+t
                // NB: column j-1 of this vector is the previous vector's last FINAL H (>= Hn): the recurrence wants the final H
                x = _mm256_max_epi16(x, _mm256_adds_epi16(shl16<1>(x, vneg), ve));
                x = _mm256_max_epi16(x, _mm256_adds_epi16(shl16<2>(x, vneg), ve2));
                x = _mm256_max_epi16(x, _mm256_adds_epi16(shl16<4>(x, vneg), ve4));
                x = _mm256_max_epi16(x, _mm256_adds_epi16(shl16<8>(x, vneg), ve8));
                x = _mm256_max_epi16(x, _mm256_adds_epi16(_mm256_set1_epi16(carry), ramp));
                _mm256_storeu_si256((__m256i *)(Er + j), x);
                const __m256i hf = _mm256_max_epi16(hn, x);
                _mm256_storeu_si256((__m256i *)(Hr + j), hf);
                carry = Er[j + 15];
            }
            // columns beyond the sequence were computed from padding: put the borders back
            for (size_t j = L + 1; j < W; ++j) { Hr[j] = 0; Fr[j] = NEG; Er[j] = NEG; }
            int16_t rowmax = 0;
            for (size_t j = 1; j <= L; ++j) rowmax = std::max(rowmax, Hr[j]);
            if (rowmax > best) {
                best = rowmax; bi = (long)i;
                for (size_t j = 1; j <= L; ++j) if (Hr[j] == rowmax) { bj = (long)j; break; }
            }
        }
    }
#endif

    template <typename T>
    poa_alignment_t trace(const std::string &seq, poa_graph_t &G, const std::vector<uint32_t> &rank, const T *H, const T *F, const T *E, size_t W,
                          size_t i, size_t j) {
        poa_alignment_t aln;
        size_t pi_ = 0, pj_ = 0;
        while (H[i * W + j] != 0) {
            int32_t Hij = H[i * W + j];
            bool found = false, ext_left = false, ext_up = false;
            if (i != 0 && j != 0) {
                const poa_node_t &nd = G.nodes[G.rank_to_node[i - 1]];
                int32_t mc = nd.letter == seq[j - 1] ? m : n;
                size_t np = nd.in_edges.size();
                for (size_t k = 0; k < std::max<size_t>(np, 1); ++k) {
                    size_t p = np == 0 ? 0 : rank[G.edges[nd.in_edges[k]].begin] + 1;
                    if (Hij == (int32_t)H[p * W + j - 1] + mc) { pi_ = p; pj_ = j - 1; found = true; break; }
                }
            }
            if (!found && i != 0) {
                const poa_node_t &nd = G.nodes[G.rank_to_node[i - 1]];
                size_t np = nd.in_edges.size();
                for (size_t k = 0; k < std::max<size_t>(np, 1); ++k) {
                    size_t p = np == 0 ? 0 : rank[G.edges[nd.in_edges[k]].begin] + 1;
                    if ((ext_up = (Hij == (int32_t)F[p * W + j] + e)) || Hij == (int32_t)H[p * W + j] + g) {
                        pi_ = p; pj_ = j; found = true; break;
                    }
                }
            }
            if (!found && j != 0) {
                if ((ext_left = (Hij == (int32_t)E[i * W + j - 1] + e)) || Hij == (int32_t)H[i * W + j - 1] + g) {
                    pi_ = i; pj_ = j - 1; found = true;
                }
            }
            aln.emplace_back(i == pi_ ? -1 : (int32_t)G.rank_to_node[i - 1], j == pj_ ? -1 : (int32_t)(j - 1));
            i = pi_; j = pj_;
            if (ext_left) {
                while (true) {
                    aln.emplace_back(-1, (int32_t)(j - 1));
                    --j;
                    if ((int32_t)E[i * W + j] + e != (int32_t)E[i * W + j + 1]) break;
                }
            } else if (ext_up) {
                while (true) {
                    bool stop = false;
                    pi_ = 0;
                    const poa_node_t &nd = G.nodes[G.rank_to_node[i - 1]];
                    for (uint32_t ei : nd.in_edges) {
                        size_t p = rank[G.edges[ei].begin] + 1;
                        if ((stop = ((int32_t)F[i * W + j] == (int32_t)H[p * W + j] + g)) || (int32_t)F[i * W + j] == (int32_t)F[p * W + j] + e) {
                            pi_ = p; break;
                        }
                    }
                    aln.emplace_back((int32_t)G.rank_to_node[i - 1], -1);
                    i = pi_;
                    if (stop || i == 0) break;
                }
            }
        }
        std::reverse(aln.begin(), aln.end());
        return aln;
    }
};

// The call pattern of correct.cpp:398-405.
inline bool &poa_simd_default() { static bool v = false; return v; }      // bench.py's CPU baseline switches the AVX2 rows on

inline std::vector<std::string> poa_msa(const std::vector<std::string> &seqs, uint64_t *cells = nullptr) {
    poa_graph_t G;
    poa_engine_t eng;
    eng.simd = poa_simd_default();
    for (auto &s : seqs) {
        poa_alignment_t a = eng.align(s, G);
        G.add_alignment(a, s);
    }
    if (cells) *cells += G.dp_cells;
    return G.msa();
}

}  // namespace orc
