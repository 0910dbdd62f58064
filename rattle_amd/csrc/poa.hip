// Kernel C: partial-order alignment of a pack of sequences into an MSA, entirely on the device.
// Replaces the spoa calls of /root/reference/correct.cpp:395-405,428-436,520-532
//   createAlignmentEngine(kSW, m=5, n=-4, g=-8, e=-6); align(); add_alignment();
//   generate_multiple_sequence_alignment()
// (spoa is an absent, unpinned submodule; behaviour restated in oracle/orc_poa.hpp and pinned
// there against toyset/rna/output/consensi.fq).
//
// One WAVEFRONT per pack (persistent blocks pull packs from a queue, largest first).  For every
// sequence of the pack, in order:
//   1. spoa's topological sort (iterative DFS over in-edges from node ids 0..n-1, aligned
//      groups ranked consecutively) runs wave-uniformly with its mark bits and stack in LDS;
//      each node it emits is immediately given its DP row, so rows are produced in exactly
//      spoa's rank order (which decides best-cell ties).
//   2. a DP row = 64 lanes x 16 consecutive columns per 1024-column segment: predecessors' H/F
//      rows are read as aligned 32-byte blocks, the horizontal affine gap recurrence
//      E[j] = max(H[j-1]+g, E[j-1]+e) is solved exactly as a prefix-max of
//      Hn[j-1]+g-j*e (in-lane pass + wave scan), H/F/E rows are written back as int16.
//   3. best cell = first maximum in (rank, column) order; traceback in spoa's order
//      (diagonal, vertical, horizontal; predecessors in in-edge insertion order; affine
//      extension runs followed inside F / E) by lane 0.
//   4. add_alignment by lane 0: prefix/suffix chains, node reuse, aligned-group siblings,
//      edge insertion in order; the node path of the sequence is recorded.
// At the end one more DFS assigns MSA columns (a group shares a column) and every base's
// column index is written out; the host expands rows ('-' elsewhere).
//
// HBM traffic (algorithmic): 6 B per DP cell written (H,E,F int16).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace rattle {

#define POA_M 5
#define POA_N (-4)
#define POA_G (-8)
#define POA_E (-6)
#define POA_NEG (-(1 << 28))
#define POA_STACK 1024

struct poa_node {          // 32 bytes
    uint32_t info;         // letter | n_al << 8 | n_in << 16
    uint32_t in0;          // first in-edge's begin node (valid if n_in > 0)
    uint32_t more_head;    // further in-edges: index into edges[], 0xFFFFFFFF = none
    uint32_t more_tail;
    uint32_t al[4];        // aligned_nodes_ids in insertion order
};

struct poa_edge { uint32_t begin, next; };

struct poa_slot {          // per-block scratch arena (device pointers)
    poa_node *nodes;
    poa_edge *edges;
    int32_t *rank;         // node -> rank (or MSA column in the final pass)
    uint32_t *order;       // rank -> node
    int16_t *H, *F, *E;    // (rows) x Lp, column j stored at index j-1
    int32_t *aln;          // traceback output, pairs (node, pos), reversed
    uint32_t *spill;       // DFS stack spill
};

struct poa_args {
    const uint8_t *seq;
    const uint64_t *off;           // per sequence
    const uint32_t *pack_first;    // per pack
    const uint32_t *queue;         // pack ids, largest first
    uint32_t n_queue;
    uint32_t *queue_head;
    poa_slot *slots;
    uint32_t node_cap, edge_cap;
    uint64_t cell_cap;             // elements per matrix
    uint32_t aln_cap, spill_cap;
    uint32_t *out_col;             // per base: node id during the run, MSA column at the end
    uint32_t *out_width;           // per pack
    uint32_t *status;              // per pack: 0 ok, else error code
    unsigned long long *counters;  // [0] DP cells, [1] alignments, [2] final nodes, [3] rows
};

enum { POA_OK = 0, POA_ERR_NODES = 1, POA_ERR_CELLS = 2, POA_ERR_EDGES = 3, POA_ERR_ALN = 4, POA_ERR_SPILL = 5, POA_ERR_GRAPH = 6 };

struct poa_state {
    // LDS
    uint32_t *done;        // bit per node
    uint32_t *nocheck;     // bit per node (check_aligned == false)
    uint32_t *stack;       // POA_STACK entries
    // wave-uniform registers
    uint32_t n_nodes, n_edges;
    uint32_t sp, spilled;
    uint32_t err;
};

__device__ __forceinline__ bool bit_get(const uint32_t *b, uint32_t i) { return (b[i >> 5] >> (i & 31)) & 1u; }
__device__ __forceinline__ void bit_set(uint32_t *b, uint32_t i) { b[i >> 5] |= 1u << (i & 31); }

__device__ __forceinline__ uint32_t rd_letter(uint32_t info) { return info & 0xFFu; }
__device__ __forceinline__ uint32_t rd_nal(uint32_t info) { return (info >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t rd_nin(uint32_t info) { return info >> 16; }

// ---- DFS stack with spill to global --------------------------------------------------------
__device__ __forceinline__ void st_push(poa_state &S, const poa_slot &sl, const poa_args &A, uint32_t v) {
    if (S.sp == POA_STACK) {
        // spill the bottom half
        if (S.spilled + POA_STACK / 2 > A.spill_cap) { S.err = POA_ERR_SPILL; return; }
        for (uint32_t t = threadIdx.x; t < POA_STACK / 2; t += 64) sl.spill[S.spilled + t] = S.stack[t];
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < POA_STACK / 2; t += 64) S.stack[t] = S.stack[t + POA_STACK / 2];
        __syncthreads();
        S.spilled += POA_STACK / 2;
        S.sp = POA_STACK / 2;
    }
    S.stack[S.sp++] = v;
}

__device__ __forceinline__ void st_refill(poa_state &S, const poa_slot &sl) {
    if (S.sp == 0 && S.spilled > 0) {
        __syncthreads();
        S.spilled -= POA_STACK / 2;
        for (uint32_t t = threadIdx.x; t < POA_STACK / 2; t += 64) S.stack[t] = sl.spill[S.spilled + t];
        __syncthreads();
        S.sp = POA_STACK / 2;
    }
}

// ---- one DP row --------------------------------------------------------------------------------
// Row `row` (>=1) for node record `nd`; sequence s[0..L), Lp = padded width.
// Returns the row maximum of H over columns 1..L (wave-uniform).
__device__ int32_t dp_row(const poa_slot &sl, const poa_node &nd, uint32_t row, const uint8_t *__restrict__ s, uint32_t L,
                          uint32_t Lp) {
    const int lane = threadIdx.x;
    const uint32_t letter = rd_letter(nd.info);
    const uint32_t n_in = rd_nin(nd.info);
    int16_t *Hr = sl.H + (uint64_t)row * Lp;
    int16_t *Fr = sl.F + (uint64_t)row * Lp;
    int16_t *Er = sl.E + (uint64_t)row * Lp;
    int32_t row_max = 0;
    int32_t carry_e = POA_NEG;      // E'[.] prefix max entering the segment
    int32_t carry_hn = 0;           // Hn of the column left of the segment (column 0 -> H = 0)
    // predecessor rows, wave-uniform; gathered once per row (n_in is small)
    for (uint32_t seg = 0; seg < Lp; seg += 1024) {
        const uint32_t c0 = seg + lane * 16;          // storage index of this lane's first column (column j = idx+1)
        const bool act = c0 < Lp;
        int32_t hn[16], fr[16];
        int32_t sc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            uint32_t idx = c0 + t;
            uint8_t ch = (act && idx < L) ? s[idx] : 0;
            sc[t] = ch == letter ? POA_M : POA_N;
            hn[t] = POA_NEG;
            fr[t] = POA_NEG;
        }
        // iterate predecessors in in-edge order (order is irrelevant for max, kept for clarity)
        uint32_t e = nd.more_head;
        for (uint32_t k = 0; k < (n_in ? n_in : 1u); ++k) {
            int32_t hp[16], fp[16];
            int32_t hleft;                 // H[p][j-1] for the lane's first column
            if (n_in == 0) {
#pragma unroll
                for (int t = 0; t < 16; ++t) { hp[t] = 0; fp[t] = POA_NEG; }
                hleft = 0;
            } else {
                uint32_t b;
                if (k == 0) b = nd.in0;
                else { poa_edge ed = sl.edges[e]; b = ed.begin; e = ed.next; }
                uint32_t prow = (uint32_t)sl.rank[b] + 1;
                const int16_t *Hp = sl.H + (uint64_t)prow * Lp;
                const int16_t *Fp = sl.F + (uint64_t)prow * Lp;
                if (act) {
                    const uint4 *h4 = (const uint4 *)(Hp + c0);
                    const uint4 *f4 = (const uint4 *)(Fp + c0);
                    uint4 a0 = h4[0], a1 = h4[1], b0 = f4[0], b1 = f4[1];
                    uint32_t hw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    uint32_t fw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        hp[2 * t] = (int16_t)(hw[t] & 0xFFFF); hp[2 * t + 1] = (int16_t)(hw[t] >> 16);
                        fp[2 * t] = (int16_t)(fw[t] & 0xFFFF); fp[2 * t + 1] = (int16_t)(fw[t] >> 16);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 16; ++t) { hp[t] = 0; fp[t] = POA_NEG; }
                }
                // H[p][j-1] of the first column: previous lane's last element, or the element left of the segment
                int32_t up = __shfl_up(hp[15], 1, 64);
                int32_t seg_left = seg == 0 ? 0 : (int32_t)Hp[seg - 1];
                hleft = lane == 0 ? seg_left : up;
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                int32_t hl = t == 0 ? hleft : hp[t - 1];
                hn[t] = max(hn[t], hl + sc[t]);
                fr[t] = max(fr[t], max(hp[t] + POA_G, fp[t] + POA_E));
            }
        }
        // Hn = max(diag, F, 0); t_j = Hn[j-1] + g - j*e; E'[j] = prefix max; E[j] = E'[j] + j*e
#pragma unroll
        for (int t = 0; t < 16; ++t) hn[t] = max(max(hn[t], fr[t]), 0);
        int32_t left_hn = __shfl_up(hn[15], 1, 64);
        if (lane == 0) left_hn = carry_hn;
        int32_t ep[16];
        int32_t run = POA_NEG;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            int32_t j = (int32_t)(c0 + t) + 1;
            int32_t hl = t == 0 ? left_hn : hn[t - 1];
            run = max(run, hl + POA_G - j * POA_E);
            ep[t] = run;
        }
        // exclusive wave prefix max of the lane totals
        int32_t incl = act ? run : POA_NEG;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int32_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl = max(incl, o);
        }
        int32_t excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = POA_NEG;
        excl = max(excl, carry_e);
        int32_t lane_max = 0;
        uint32_t hw[8], fw[8], ew[8];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            int32_t j = (int32_t)(c0 + t) + 1;
            int32_t ev = max(ep[t], excl) + j * POA_E;
            int32_t hv = max(hn[t], ev);
            if (c0 + t < L) lane_max = max(lane_max, hv);
            uint32_t h16 = (uint32_t)(uint16_t)(int16_t)hv, f16 = (uint32_t)(uint16_t)(int16_t)fr[t],
                     e16 = (uint32_t)(uint16_t)(int16_t)ev;
            if (t & 1) { hw[t >> 1] |= h16 << 16; fw[t >> 1] |= f16 << 16; ew[t >> 1] |= e16 << 16; }
            else { hw[t >> 1] = h16; fw[t >> 1] = f16; ew[t >> 1] = e16; }
        }
        if (act) {
            uint4 *h4 = (uint4 *)(Hr + c0), *f4 = (uint4 *)(Fr + c0), *e4 = (uint4 *)(Er + c0);
            h4[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]); h4[1] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
            f4[0] = make_uint4(fw[0], fw[1], fw[2], fw[3]); f4[1] = make_uint4(fw[4], fw[5], fw[6], fw[7]);
            e4[0] = make_uint4(ew[0], ew[1], ew[2], ew[3]); e4[1] = make_uint4(ew[4], ew[5], ew[6], ew[7]);
        }
        // carries for the next segment (lane 63 always active when another segment follows)
        carry_e = __shfl(max(incl, carry_e), 63, 64);
        carry_hn = __shfl(hn[15], 63, 64);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) lane_max = max(lane_max, __shfl_xor(lane_max, d, 64));
        row_max = max(row_max, lane_max);
    }
    return row_max;
}

// ---- spoa Graph::topological_sort fused with the DP ------------------------------------------------
// mode 0: ranks + DP rows for sequence s;  mode 1: MSA columns only (rank[] receives the column).
// Returns via refs: best score / best row (first maximum in rank order); n_cols for mode 1.
__device__ void dfs_pass(poa_state &S, const poa_slot &sl, const poa_args &A, int mode, const uint8_t *s, uint32_t L, uint32_t Lp,
                         int32_t &best, uint32_t &best_row, uint32_t &n_emit, uint32_t &n_cols) {
    const uint32_t n = S.n_nodes;
    for (uint32_t t = threadIdx.x; t < (n + 31) / 32; t += 64) { S.done[t] = 0; S.nocheck[t] = 0; }
    __syncthreads();
    best = 0; best_row = 0; n_emit = 0; n_cols = 0;
    S.sp = 0; S.spilled = 0;
    for (uint32_t root = 0; root < n && !S.err; ++root) {
        if (bit_get(S.done, root)) continue;
        st_push(S, sl, A, root);
        while (!S.err) {
            st_refill(S, sl);
            if (S.sp == 0) break;
            const uint32_t v = S.stack[S.sp - 1];
            if (bit_get(S.done, v)) { --S.sp; continue; }
            const poa_node nd = sl.nodes[v];
            const uint32_t n_in = rd_nin(nd.info), n_al = rd_nal(nd.info);
            bool valid = true;
            uint32_t e = nd.more_head;
            for (uint32_t k = 0; k < n_in; ++k) {
                uint32_t b;
                if (k == 0) b = nd.in0;
                else { poa_edge ed = sl.edges[e]; b = ed.begin; e = ed.next; }
                if (!bit_get(S.done, b)) { st_push(S, sl, A, b); valid = false; }
            }
            const bool check = !bit_get(S.nocheck, v);
            if (check) {
                for (uint32_t k = 0; k < n_al; ++k) {
                    uint32_t a = nd.al[k];
                    if (!bit_get(S.done, a)) {
                        st_push(S, sl, A, a);
                        bit_set(S.nocheck, a);          // every lane writes the same word value
                        valid = false;
                    }
                }
            }
            if (!valid) continue;
            bit_set(S.done, v);
            --S.sp;                                  // top is still v
            if (!check) continue;
            // emit v, then its aligned group in list order
            for (uint32_t g = 0; g <= n_al; ++g) {
                const uint32_t u = g == 0 ? v : nd.al[g - 1];
                if (mode == 1) {
                    if (threadIdx.x == 0) sl.rank[u] = (int32_t)n_cols;
                } else {
                    const uint32_t r = n_emit;
                    if (threadIdx.x == 0) { sl.rank[u] = (int32_t)r; sl.order[r] = u; }
                    __syncthreads();                 // rank[] of this node may be read by the next row
                    const poa_node un = g == 0 ? nd : sl.nodes[u];
                    int32_t rm = dp_row(sl, un, r + 1, s, L, Lp);
                    __syncthreads();                 // row visible to later rows (cross-lane reads)
                    if (rm > best) { best = rm; best_row = r + 1; }
                }
                ++n_emit;
            }
            ++n_cols;
        }
    }
    __syncthreads();
}

// ---- graph update helpers (lane 0) --------------------------------------------------------------
__device__ uint32_t g_add_node(poa_state &S, const poa_slot &sl, const poa_args &A, uint8_t letter) {
    if (S.n_nodes >= A.node_cap) { S.err = POA_ERR_NODES; return 0; }
    poa_node nd;
    nd.info = letter; nd.in0 = 0; nd.more_head = 0xFFFFFFFFu; nd.more_tail = 0xFFFFFFFFu;
    nd.al[0] = nd.al[1] = nd.al[2] = nd.al[3] = 0;
    sl.nodes[S.n_nodes] = nd;
    return S.n_nodes++;
}

// Graph::add_edge: nothing if begin->end exists, else append to end's in-edge list.
__device__ void g_add_edge(poa_state &S, const poa_slot &sl, const poa_args &A, uint32_t b, uint32_t en) {
    poa_node nd = sl.nodes[en];
    uint32_t n_in = rd_nin(nd.info);
    if (n_in > 0) {
        if (nd.in0 == b) return;
        uint32_t e = nd.more_head;
        for (uint32_t k = 1; k < n_in; ++k) { poa_edge ed = sl.edges[e]; if (ed.begin == b) return; e = ed.next; }
    }
    if (n_in == 0) {
        nd.in0 = b;
    } else {
        if (S.n_edges >= A.edge_cap) { S.err = POA_ERR_EDGES; return; }
        uint32_t id = S.n_edges++;
        sl.edges[id] = poa_edge{b, 0xFFFFFFFFu};
        if (nd.more_head == 0xFFFFFFFFu) nd.more_head = id;
        else sl.edges[nd.more_tail].next = id;
        nd.more_tail = id;
    }
    if (n_in >= 0xFFFF) { S.err = POA_ERR_GRAPH; return; }
    nd.info += 1u << 16;
    sl.nodes[en] = nd;
}

// Graph::add_sequence(b, e): fresh chain; returns first node or -1.  Records the path.
__device__ int32_t g_add_chain(poa_state &S, const poa_slot &sl, const poa_args &A, const uint8_t *s, uint32_t b, uint32_t e,
                               uint32_t *path) {
    if (b == e) return -1;
    uint32_t first = g_add_node(S, sl, A, s[b]);
    if (S.err) return -1;
    path[b] = first;
    for (uint32_t i = b + 1; i < e; ++i) {
        uint32_t id = g_add_node(S, sl, A, s[i]);
        if (S.err) return -1;
        path[i] = id;
        // fresh node: its only in-edge
        poa_node nd = sl.nodes[id];
        nd.in0 = id - 1; nd.info += 1u << 16;
        sl.nodes[id] = nd;
    }
    return (int32_t)first;
}

__global__ __launch_bounds__(64) void poa_kernel(poa_args A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t s_pack;
    __shared__ uint32_t s_bc[8];
    const int lane = threadIdx.x;
    const poa_slot sl = A.slots[blockIdx.x];
    poa_state S;
    const uint32_t bit_words = (A.node_cap + 31) / 32;
    S.done = lds; S.nocheck = lds + bit_words; S.stack = lds + 2 * bit_words;

    while (true) {
        __syncthreads();
        if (lane == 0) s_pack = atomicAdd(A.queue_head, 1u);
        __syncthreads();
        const uint32_t qi = s_pack;
        if (qi >= A.n_queue) break;
        const uint32_t pk = A.queue[qi];
        const uint32_t q0 = A.pack_first[pk], q1 = A.pack_first[pk + 1];
        S.n_nodes = 0; S.n_edges = 0; S.err = 0; S.sp = 0; S.spilled = 0;
        unsigned long long cells = 0, rows = 0;

        for (uint32_t q = q0; q < q1 && !S.err; ++q) {
            const uint64_t so = A.off[q];
            const uint32_t L = (uint32_t)(A.off[q + 1] - so);
            const uint8_t *s = A.seq + so;
            uint32_t *path = A.out_col + so;
            if (L == 0) continue;                                   // Graph::add_alignment: empty sequence ignored
            uint32_t n_aln = 0;
            if (S.n_nodes > 0) {
                // ---- align ----
                const uint32_t Lp = (L + 15u) & ~15u;
                if ((uint64_t)(S.n_nodes + 1) * Lp > A.cell_cap) { S.err = POA_ERR_CELLS; break; }
                int32_t best; uint32_t best_row, n_emit, n_cols;
                dfs_pass(S, sl, A, 0, s, L, Lp, best, best_row, n_emit, n_cols);
                if (S.err) break;
                if (n_emit != S.n_nodes) { S.err = POA_ERR_GRAPH; break; }
                cells += (unsigned long long)S.n_nodes * L;
                rows += S.n_nodes;
                if (best > 0) {
                    // first column of the best row holding the maximum
                    const int16_t *Hb = sl.H + (uint64_t)best_row * Lp;
                    uint32_t bj = 0xFFFFFFFFu;
                    for (uint32_t c = lane; c < L; c += 64) if ((int32_t)Hb[c] == best) { bj = c + 1; break; }
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) bj = min(bj, (uint32_t)__shfl_xor((int)bj, d, 64));
                    // ---- traceback (lane 0), spoa order: diagonal, vertical, horizontal ----
                    if (lane == 0) {
                        uint32_t i = best_row, j = bj;
                        uint32_t cnt = 0;
                        uint32_t err = 0;
                        auto Hat = [&](uint32_t r, uint32_t c) -> int32_t { return (r == 0 || c == 0) ? 0 : (int32_t)sl.H[(uint64_t)r * Lp + c - 1]; };
                        auto Fat = [&](uint32_t r, uint32_t c) -> int32_t { return (r == 0 || c == 0) ? POA_NEG : (int32_t)sl.F[(uint64_t)r * Lp + c - 1]; };
                        auto Eat = [&](uint32_t r, uint32_t c) -> int32_t { return (r == 0 || c == 0) ? POA_NEG : (int32_t)sl.E[(uint64_t)r * Lp + c - 1]; };
                        auto put = [&](int32_t node, int32_t pos) {
                            if (cnt >= A.aln_cap) { err = POA_ERR_ALN; return; }
                            sl.aln[2 * cnt] = node; sl.aln[2 * cnt + 1] = pos; ++cnt;
                        };
                        while (!err && Hat(i, j) != 0) {
                            const int32_t Hij = Hat(i, j);
                            bool found = false, ext_left = false, ext_up = false;
                            uint32_t pi = 0, pj = 0;
                            const uint32_t v = i != 0 ? sl.order[i - 1] : 0;
                            poa_node nd;
                            uint32_t n_in = 0;
                            if (i != 0) { nd = sl.nodes[v]; n_in = rd_nin(nd.info); }
                            if (i != 0 && j != 0) {
                                const int32_t mc = rd_letter(nd.info) == s[j - 1] ? POA_M : POA_N;
                                uint32_t e = nd.more_head;
                                for (uint32_t k = 0; k < (n_in ? n_in : 1u); ++k) {
                                    uint32_t p = 0;
                                    if (n_in) {
                                        uint32_t b;
                                        if (k == 0) b = nd.in0; else { poa_edge ed = sl.edges[e]; b = ed.begin; e = ed.next; }
                                        p = (uint32_t)sl.rank[b] + 1;
                                    }
                                    if (Hij == Hat(p, j - 1) + mc) { pi = p; pj = j - 1; found = true; break; }
                                }
                            }
                            if (!found && i != 0) {
                                uint32_t e = nd.more_head;
                                for (uint32_t k = 0; k < (n_in ? n_in : 1u); ++k) {
                                    uint32_t p = 0;
                                    if (n_in) {
                                        uint32_t b;
                                        if (k == 0) b = nd.in0; else { poa_edge ed = sl.edges[e]; b = ed.begin; e = ed.next; }
                                        p = (uint32_t)sl.rank[b] + 1;
                                    }
                                    if ((ext_up = (Hij == Fat(p, j) + POA_E)) || Hij == Hat(p, j) + POA_G) { pi = p; pj = j; found = true; break; }
                                }
                            }
                            if (!found && j != 0) {
                                if ((ext_left = (Hij == Eat(i, j - 1) + POA_E)) || Hij == Hat(i, j - 1) + POA_G) { pi = i; pj = j - 1; found = true; }
                            }
                            if (!found) { err = POA_ERR_GRAPH; break; }
                            put(i == pi ? -1 : (int32_t)v, j == pj ? -1 : (int32_t)(j - 1));
                            i = pi; j = pj;
                            if (ext_left) {
                                while (!err) {
                                    put(-1, (int32_t)(j - 1));
                                    --j;
                                    if (Eat(i, j) + POA_E != Eat(i, j + 1)) break;
                                }
                            } else if (ext_up) {
                                while (!err) {
                                    bool stop = false;
                                    uint32_t np = 0;
                                    const uint32_t u = sl.order[i - 1];
                                    const poa_node un = sl.nodes[u];
                                    const uint32_t uin = rd_nin(un.info);
                                    uint32_t e = un.more_head;
                                    const int32_t Fij = Fat(i, j);
                                    for (uint32_t k = 0; k < uin; ++k) {
                                        uint32_t b;
                                        if (k == 0) b = un.in0; else { poa_edge ed = sl.edges[e]; b = ed.begin; e = ed.next; }
                                        uint32_t p = (uint32_t)sl.rank[b] + 1;
                                        if ((stop = (Fij == Hat(p, j) + POA_G)) || Fij == Fat(p, j) + POA_E) { np = p; break; }
                                    }
                                    put((int32_t)u, -1);
                                    i = np;
                                    if (stop || i == 0) break;
                                }
                            }
                        }
                        s_bc[0] = cnt; s_bc[1] = err;
                    }
                    __syncthreads();
                    n_aln = s_bc[0];
                    if (s_bc[1]) { S.err = s_bc[1]; break; }
                }
            }
            // ---- add_alignment (lane 0) ----
            if (lane == 0) {
                uint32_t nn = S.n_nodes, ne = S.n_edges;
                (void)nn; (void)ne;
                if (n_aln == 0) {
                    g_add_chain(S, sl, A, s, 0, L, path);
                } else {
                    // pairs are stored reversed: pair t (forward) = aln[n_aln-1-t]
                    int32_t first_valid = -1, last_valid = -1;
                    for (uint32_t t = 0; t < n_aln; ++t) {
                        int32_t pos = sl.aln[2 * (n_aln - 1 - t) + 1];
                        if (pos != -1) { if (first_valid == -1) first_valid = pos; last_valid = pos; }
                    }
                    const uint32_t before = S.n_nodes;
                    g_add_chain(S, sl, A, s, 0, (uint32_t)first_valid, path);
                    int32_t head = before == S.n_nodes ? -1 : (int32_t)S.n_nodes - 1;
                    int32_t tail = S.err ? -1 : g_add_chain(S, sl, A, s, (uint32_t)last_valid + 1, L, path);
                    for (uint32_t t = 0; t < n_aln && !S.err; ++t) {
                        const int32_t an = sl.aln[2 * (n_aln - 1 - t)], pos = sl.aln[2 * (n_aln - 1 - t) + 1];
                        if (pos == -1) continue;
                        const uint8_t letter = s[pos];
                        uint32_t cur;
                        if (an == -1) {
                            cur = g_add_node(S, sl, A, letter);
                        } else {
                            poa_node nd = sl.nodes[an];
                            if (rd_letter(nd.info) == letter) {
                                cur = (uint32_t)an;
                            } else {
                                const uint32_t n_al = rd_nal(nd.info);
                                int32_t hit = -1;
                                for (uint32_t k = 0; k < n_al; ++k)
                                    if (rd_letter(sl.nodes[nd.al[k]].info) == letter) { hit = (int32_t)nd.al[k]; break; }
                                if (hit != -1) {
                                    cur = (uint32_t)hit;
                                } else {
                                    if (n_al >= 4) { S.err = POA_ERR_GRAPH; break; }
                                    cur = g_add_node(S, sl, A, letter);
                                    if (S.err) break;
                                    poa_node cn = sl.nodes[cur];
                                    for (uint32_t k = 0; k < n_al; ++k) {
                                        const uint32_t a = nd.al[k];
                                        cn.al[k] = a;
                                        poa_node an_ = sl.nodes[a];
                                        const uint32_t c = rd_nal(an_.info);
                                        if (c >= 4) { S.err = POA_ERR_GRAPH; break; }
                                        an_.al[c] = cur; an_.info += 1u << 8;
                                        sl.nodes[a] = an_;
                                    }
                                    cn.al[n_al] = (uint32_t)an;
                                    cn.info += (n_al + 1) << 8;
                                    sl.nodes[cur] = cn;
                                    nd.al[n_al] = cur; nd.info += 1u << 8;
                                    sl.nodes[an] = nd;
                                }
                            }
                        }
                        if (S.err) break;
                        path[pos] = cur;
                        if (head != -1) g_add_edge(S, sl, A, (uint32_t)head, cur);
                        head = (int32_t)cur;
                    }
                    if (!S.err && tail != -1) g_add_edge(S, sl, A, (uint32_t)head, (uint32_t)tail);
                }
                s_bc[2] = S.n_nodes; s_bc[3] = S.n_edges; s_bc[4] = S.err;
            }
            __syncthreads();
            S.n_nodes = s_bc[2]; S.n_edges = s_bc[3]; S.err = s_bc[4];
            __syncthreads();
        }

        // ---- generate_multiple_sequence_alignment: column per node, then per base ----
        uint32_t width = 0;
        if (!S.err && S.n_nodes > 0) {
            int32_t best; uint32_t best_row, n_emit, n_cols;
            dfs_pass(S, sl, A, 1, nullptr, 0, 0, best, best_row, n_emit, n_cols);
            if (!S.err && n_emit != S.n_nodes) S.err = POA_ERR_GRAPH;
            width = n_cols;
            __syncthreads();
            if (!S.err) {
                const uint64_t b0 = A.off[q0], b1 = A.off[q1];
                for (uint64_t b = b0 + lane; b < b1; b += 64) A.out_col[b] = (uint32_t)sl.rank[A.out_col[b]];
            }
        }
        if (lane == 0) {
            A.out_width[pk] = width;
            A.status[pk] = S.err;
            atomicAdd(&A.counters[0], cells);
            atomicAdd(&A.counters[1], (unsigned long long)(q1 - q0));
            atomicAdd(&A.counters[2], (unsigned long long)S.n_nodes);
            atomicAdd(&A.counters[3], rows);
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------
int poa_msa_run(rattle_ctx *ctx, const uint8_t *seq, const uint64_t *off, uint32_t n_seqs, const uint32_t *pack_first,
                uint32_t n_packs, rattle_msa_set **out) {
    hipStream_t st = ctx->stream;
    rattle_msa_set *R = (rattle_msa_set *)calloc(1, sizeof(rattle_msa_set));
    R->n_packs = n_packs;
    R->width = (uint32_t *)calloc(std::max<uint32_t>(n_packs, 1), sizeof(uint32_t));
    R->row_offset = (uint64_t *)calloc((size_t)n_seqs + 1, sizeof(uint64_t));
    *out = R;
    if (n_packs == 0 || n_seqs == 0) { R->rows = (char *)calloc(1, 1); return 0; }
    if (pack_first[0] != 0 || pack_first[n_packs] != n_seqs) { set_error("pack_first must cover [0, n_seqs]"); return RATTLE_ERR_ARG; }
    const uint64_t total = off[n_seqs];

    // pack sizing
    std::vector<uint64_t> pbases(n_packs);
    std::vector<uint32_t> pmaxL(n_packs);
    uint64_t max_bases = 0; uint32_t maxL = 0;
    for (uint32_t p = 0; p < n_packs; ++p) {
        uint64_t b = off[pack_first[p + 1]] - off[pack_first[p]];
        uint32_t m = 0;
        for (uint32_t q = pack_first[p]; q < pack_first[p + 1]; ++q) m = std::max<uint32_t>(m, (uint32_t)(off[q + 1] - off[q]));
        pbases[p] = b; pmaxL[p] = m;
        max_bases = std::max(max_bases, b); maxL = std::max(maxL, m);
    }
    if (5ull * maxL + 16 > 32000) { set_error("sequence too long for the int16 POA kernel (> 6396 nt)"); return RATTLE_ERR_ARG; }
    for (uint64_t b = 0; b < total; ++b) {
        uint8_t c = seq[b];
        if (c == 0) { set_error("NUL byte in sequence"); return RATTLE_ERR_ARG; }
    }

    // device inputs / outputs
    dbuf<uint8_t> d_seq; dbuf<uint64_t> d_off; dbuf<uint32_t> d_pf, d_queue, d_head, d_col, d_width, d_status;
    dbuf<unsigned long long> d_cnt;
    RT_TRY(d_seq.reserve(total + 64)); RT_TRY(d_off.reserve(n_seqs + 1)); RT_TRY(d_pf.reserve(n_packs + 1));
    RT_TRY(d_queue.reserve(n_packs)); RT_TRY(d_head.reserve(1)); RT_TRY(d_col.reserve(total + 64));
    RT_TRY(d_width.reserve(n_packs)); RT_TRY(d_status.reserve(n_packs)); RT_TRY(d_cnt.reserve(8));
    RT_HIP(hipMemcpyAsync(d_seq.p, seq, total, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_off.p, off, (n_seqs + 1) * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_pf.p, pack_first, (n_packs + 1) * 4, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemsetAsync(d_cnt.p, 0, 64, st));
    RT_HIP(hipMemsetAsync(d_status.p, 0xFF, n_packs * 4, st));

    std::vector<uint32_t> todo(n_packs);
    for (uint32_t p = 0; p < n_packs; ++p) todo[p] = p;
    std::vector<uint32_t> h_status(n_packs), h_width(n_packs);

    size_t free_b = 0, total_b = 0;
    RT_HIP(hipMemGetInfo(&free_b, &total_b));
    // round 0: many slots with a modest arena; later rounds: failed packs with larger arenas
    uint32_t node_cap = 16384;
    uint64_t cell_cap = 24ull << 20;           // elements per matrix (x3 matrices x2 bytes = 144 MiB)
    int rc = 0;
    for (int round = 0; round < 6 && !todo.empty() && rc == 0; ++round) {
        // largest packs first
        std::sort(todo.begin(), todo.end(), [&](uint32_t a, uint32_t b) { return pbases[a] != pbases[b] ? pbases[a] > pbases[b] : a < b; });
        uint64_t tb = 0; uint32_t tl = 0;
        for (uint32_t p : todo) { tb = std::max(tb, pbases[p]); tl = std::max(tl, pmaxL[p]); }
        uint32_t ncap = (uint32_t)std::min<uint64_t>(node_cap, tb + 1);
        ncap = (ncap + 31u) & ~31u;
        uint32_t ecap = (uint32_t)std::min<uint64_t>(tb + 1, 0x7FFFFFFFull);
        uint32_t acap = tl + ncap + 16;
        uint32_t scap = ncap + POA_STACK;
        uint64_t ccap = std::min<uint64_t>(cell_cap, (uint64_t)(ncap + 1) * ((tl + 15u) & ~15u));
        uint64_t per_slot = (uint64_t)ncap * (32 + 4 + 4) + (uint64_t)ecap * 8 + ccap * 6 + (uint64_t)acap * 8 + (uint64_t)scap * 4 + 4096;
        uint64_t budget = (uint64_t)(free_b * 0.85);
        uint32_t max_slots = (uint32_t)std::max<uint64_t>(1, budget / per_slot);
        uint32_t n_slots = std::min<uint32_t>((uint32_t)todo.size(), std::min<uint32_t>(max_slots, 256 * 10));
        struct raw_arena { uint8_t *p = nullptr; void release() { if (p) (void)hipFree(p); p = nullptr; } } arena;
        if (hipMalloc((void **)&arena.p, (size_t)per_slot * n_slots) != hipSuccess) {
            set_error("poa: arena allocation failed"); rc = RATTLE_ERR_HIP; break;
        }
        std::vector<poa_slot> hs(n_slots);
        for (uint32_t i = 0; i < n_slots; ++i) {
            uint8_t *b = arena.p + (size_t)per_slot * i;
            auto take = [&](size_t bytes) { uint8_t *r = b; b += (bytes + 255) & ~(size_t)255; return r; };
            hs[i].nodes = (poa_node *)take((size_t)ncap * 32);
            hs[i].edges = (poa_edge *)take((size_t)ecap * 8);
            hs[i].rank = (int32_t *)take((size_t)ncap * 4);
            hs[i].order = (uint32_t *)take((size_t)ncap * 4);
            hs[i].H = (int16_t *)take(ccap * 2); hs[i].F = (int16_t *)take(ccap * 2); hs[i].E = (int16_t *)take(ccap * 2);
            hs[i].aln = (int32_t *)take((size_t)acap * 8);
            hs[i].spill = (uint32_t *)take((size_t)scap * 4);
        }
        dbuf<poa_slot> d_slots;
        rc = d_slots.reserve(n_slots);
        if (rc) break;
        hipError_t e = hipMemcpyAsync(d_slots.p, hs.data(), n_slots * sizeof(poa_slot), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_queue.p, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemsetAsync(d_head.p, 0, 4, st);
        if (e != hipSuccess) { set_error(std::string("poa setup: ") + hipGetErrorString(e)); rc = RATTLE_ERR_HIP; break; }
        poa_args A;
        A.seq = d_seq.p; A.off = d_off.p; A.pack_first = d_pf.p; A.queue = d_queue.p; A.n_queue = (uint32_t)todo.size();
        A.queue_head = d_head.p; A.slots = d_slots.p; A.node_cap = ncap; A.edge_cap = ecap; A.cell_cap = ccap;
        A.aln_cap = acap; A.spill_cap = scap; A.out_col = d_col.p; A.out_width = d_width.p; A.status = d_status.p;
        A.counters = d_cnt.p;
        size_t shm = ((size_t)((ncap + 31) / 32) * 2 + POA_STACK) * 4;
        if (shm > 60 * 1024) (void)hipFuncSetAttribute((const void *)poa_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        {
            ktimer T(ctx, K_POA, 0);
            hipLaunchKernelGGL(poa_kernel, dim3(n_slots), dim3(64), shm, st, A);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h_status.data(), d_status.p, n_packs * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        d_slots.release();
        arena.release();
        if (e != hipSuccess) { set_error(std::string("poa_kernel: ") + hipGetErrorString(e)); rc = RATTLE_ERR_HIP; break; }
        std::vector<uint32_t> again;
        for (uint32_t p : todo) {
            uint32_t s = h_status[p];
            if (s == POA_OK) continue;
            if (s == POA_ERR_NODES || s == POA_ERR_CELLS || s == POA_ERR_SPILL || s == POA_ERR_ALN) again.push_back(p);
            else { set_error("poa_kernel: pack " + std::to_string(p) + " failed with status " + std::to_string(s)); rc = RATTLE_ERR_HIP; break; }
        }
        todo.swap(again);
        node_cap = std::min<uint32_t>(node_cap * 4, 1u << 20);
        cell_cap *= 8;
    }
    if (rc == 0 && !todo.empty()) { set_error("poa: " + std::to_string(todo.size()) + " pack(s) exceed the device arena"); rc = RATTLE_ERR_HIP; }
    std::vector<uint32_t> h_col;
    unsigned long long h_cnt[8] = {0};
    if (rc == 0) {
        h_col.resize(total);
        hipError_t e = hipMemcpyAsync(h_col.data(), d_col.p, total * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h_width.data(), d_width.p, n_packs * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h_cnt, d_cnt.p, 64, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error(std::string("poa readback: ") + hipGetErrorString(e)); rc = RATTLE_ERR_HIP; }
    }
    d_seq.release(); d_off.release(); d_pf.release(); d_queue.release(); d_head.release(); d_col.release();
    d_width.release(); d_status.release(); d_cnt.release();
    if (rc) return rc;

    // expand rows on the host: row = '-' * width with each base at its column
    uint64_t bytes = 0;
    for (uint32_t p = 0; p < n_packs; ++p) {
        R->width[p] = h_width[p];
        for (uint32_t q = pack_first[p]; q < pack_first[p + 1]; ++q) { R->row_offset[q] = bytes; bytes += h_width[p]; }
    }
    R->row_offset[n_seqs] = bytes;
    R->rows = (char *)malloc(bytes + 1);
    memset(R->rows, '-', bytes);
    R->rows[bytes] = 0;
    for (uint32_t p = 0; p < n_packs; ++p)
        for (uint32_t q = pack_first[p]; q < pack_first[p + 1]; ++q) {
            char *row = R->rows + R->row_offset[q];
            for (uint64_t b = off[q]; b < off[q + 1]; ++b) row[h_col[b]] = (char)seq[b];
        }
    for (int i = 0; i < 8; ++i) R->counters[i] = h_cnt[i];
    ctx->stats[K_POA].bytes += 6ull * h_cnt[0];
    return 0;
}

}  // namespace rattle
