// Kernel C: partial-order alignment of a pack of sequences into an MSA, entirely on the device.
// Replaces the spoa calls of /root/reference/correct.cpp:395-405,428-436,520-532
//   createAlignmentEngine(kSW, m=5, n=-4, g=-8, e=-6); align(); add_alignment();
//   generate_multiple_sequence_alignment()
// (spoa is an absent, unpinned submodule; behaviour restated in oracle/orc_poa.hpp and pinned
// there against toyset/rna/output/consensi.fq).
//
// One workgroup of NW wavefronts per pack (persistent blocks pull packs from a queue, largest
// first).  For every sequence of the pack, in order:
//   1. rows: the DP runs in an incrementally maintained BLOCK ORDER (a topological order with
//      aligned groups contiguous; step 6).  DP values per graph node do not depend on the order the
//      rows are taken in; spoa's own order (DFS from node ids 0..n-1) matters only for best-score
//      ties between rows and for the MSA columns, and is derived only there.
//   2. plan: every thread gathers one row's (letter, in-degree, rows of the first four
//      predecessors, tail of the in-edge list) into two 16-byte records, so the DP loop has no
//      dependent pointer chasing.
//   3. DP (dp_rows): a row = NW x 64 lanes x CPL consecutive columns; exact affine gaps through a
//      prefix-max, one barrier per row; H/F/E rows are written as int16 for the traceback.
//   4. best cell = first maximum in (spoa rank, column) order (tie_labels); traceback in spoa's
//      order (diagonal, vertical, horizontal; predecessors in in-edge insertion order; affine
//      extension runs followed inside F / E) by thread 0.
//   5. add_alignment: prefix/suffix chains, node reuse, aligned-group siblings, edge insertion in
//      order; the node path of the sequence is recorded.
//   6. merge_order: the new nodes are merged into the row order (stable parallel merge keyed by
//      the row each must precede).
// At the end spoa's topological sort assigns MSA columns (a group shares a column) and every
// base's column index is written out (kernel D or the host expands rows, '-' elsewhere).
//
// HBM traffic (algorithmic): 6 B per DP cell written (H,E,F int16).
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace rattle {

#define POA_M 5
#define POA_N (-4)
#define POA_G (-8)
#define POA_E (-6)
#define POA_NEG (-(1 << 28))
#define POA_STACK 512
#define POA_NONE 0xFFFFFFFFu
#ifndef POA_WIDE_RING
#define POA_WIDE_RING 4                    // rows of the wide classes kept in LDS (a dword per cell: 32 KB per row at 8192 columns)
#endif
#ifndef POA_LONG_RING
#define POA_LONG_RING 3                    // rows of a segment kept in LDS beyond 8192 columns (32 KB per row)
#endif
#ifndef POA_MW_4x4
#define POA_MW_4x4 7
#endif
#ifndef POA_MW_4x6
#define POA_MW_4x6 8
#endif
#ifndef POA_RING_4x4
#define POA_RING_4x4 8
#endif
#ifndef POA_RING_4x6
#define POA_RING_4x6 4
#endif
#ifndef POA_CHAIN_SEQS
#define POA_CHAIN_SEQS 256                      // a pack of more sequences than any read pack has (split: 200) is a POA #3 group: hundreds of alignments one after the other
#endif
#define POA_MAX_LEN (1u << 20)           // H <= 5 * length must stay far below 2^28 (POA_NEG)

// node record (uint4): x = letter | n_al << 8 | n_in << 16, y = first in-edge's begin node,
// z = head / w = tail of the list of further in-edges (indices into edges[]).
// aligned record (uint4): aligned_nodes_ids in insertion order.
// edge (uint2): x = begin node, y = next edge.
// plan records, one pair per DP row: plan = {node info, node id, edge index of the 5th in-edge
// (or none), edge index of the 9th in-edge (or none)}; planb / planc = rows of predecessors 1-4 / 5-8.

struct poa_args {
    const uint8_t *seq;
    const uint64_t *off;           // per sequence
    const uint32_t *pack_first;    // per pack
    const uint32_t *queue;         // pack ids, largest first
    uint32_t n_queue;
    uint32_t *queue_head;
    uint8_t *arena;                // n_slots * slot_stride bytes
    uint64_t slot_stride;
    uint64_t o_nrec, o_nal, o_edges, o_rank, o_order, o_order2, o_srank, o_rowmax, o_lh, o_nn, o_plan, o_planb, o_planc, o_pland, o_H, o_F, o_E, o_aln, o_ainfo, o_spill;
    uint32_t node_cap, edge_cap;
    uint64_t cell_cap;             // elements per matrix
    uint32_t aln_cap, spill_cap, seq_cap;
    uint64_t o_planm;              // multi-team rows (dp_rows_mt): the row loop's own 32-byte record per row
    uint32_t ring_slots, ring_reach, ring_slack;      // ... and its ring, sized at launch: slots in LDS, rows a reader looks back, rows that may be in flight
    uint32_t band;                 // PK == 8: bytes of LDS behind the graph walks' bitmaps and stack (the band's ring + selector table; tie labels and traceback tables between two DPs)
    uint32_t debug;                // tests: bit 0 = resolve ties with the full sort, bit 1 = traceback without the LDS fast path, bit 2 = ... without its jump tables
    uint32_t *out_col;             // per base: node id during the run, MSA column at the end
    uint32_t *out_width;           // per pack
    uint32_t *status;              // per pack: 0 ok, else error code
    unsigned long long *counters;  // [0] DP cells (the reference's count: rows x columns), [1] alignments, [2] final nodes, [3] rows, [4..7] phase ticks, [8..10] dp_rows_mt's diagnosis, [11] DP cells computed, [12] / [13] alignments with a certified band / a failed certificate
    unsigned long long *timeline;  // RATTLE_POA_TIMELINE: per pack {start, end} of its workgroup's work on it (100 MHz wall clock), else null
    unsigned long long *prof;      // POA_PROFILE builds: per class [0] plan [1] DP [2] ties [3] traceback [4] add_alignment [5] merge_order [6] final sort + columns [7] whole packs
};

#ifdef POA_PROFILE
#define PT_NOW() ((unsigned long long)wall_clock64())
#else
#define PT_NOW() 0ull
#endif

enum { POA_OK = 0, POA_ERR_NODES = 1, POA_ERR_CELLS = 2, POA_ERR_EDGES = 3, POA_ERR_ALN = 4, POA_ERR_SPILL = 5, POA_ERR_GRAPH = 6, POA_ERR_SYNC = 7,
       POA_ERR_BAND = 8 /* (PK == 8) an alignment of the pack has no certified band: the pack goes to the full-row kernels */ };

struct poa_ws {                    // per-block workspace: global pointers + LDS + wave-uniform state
    uint4 *nrec, *nal, *plan, *planb, *planc, *pland;
    uint32_t *planm;                       // dp_rows_mt: 8 dwords per row
    uint2 *edges;
    int32_t *rank;                         // node -> DP row - 1 (block order), MSA column in the final pass
    uint32_t *order, *order2;              // DP row - 1 -> node (double buffer for the incremental merge)
    int32_t *srank;                        // spoa rank per node (only computed when best rows tie)
    int32_t *rowmax;                       // [row][wave] row maximum of H
    int32_t *lh;                           // [row][wave] H of the column left of the wave (written and read by lane 0 of that wave)
    uint32_t *nn;                          // new nodes of the last add_alignment: (node, anchor) pairs in path order
    int16_t *H, *F, *E;
    int32_t *aln;
    uint4 *ainfo;                          // per forward pair: kind, target node, group row range (add_alignment)
    uint32_t *spill;
    uint32_t *done, *nocheck, *stack;      // LDS
    unsigned long long *hist;
    uint32_t *ring;                        // LDS: last RING rows of packed H|F per thread (thread-private)
    int32_t *lh_ring;                      // LDS: left-column H of the last RING rows per wave
    uint8_t *sq;                           // LDS copy of the sequence, 16-byte aligned
    uint32_t n_nodes, n_edges, sp, spilled, err;
    uint32_t plain;                        // the pack's letters so far are all in {A,C,G,T} or all in {A,C,G,U}: scores by table look-up
};

// words of one node bitmap in LDS: an even number, so that what follows the two bitmaps and the stack (the ring) is 16-byte aligned
__host__ __device__ constexpr uint32_t poa_bit_words(uint32_t node_cap) { return ((node_cap + 63u) / 64u) * 2u; }
__device__ __forceinline__ bool bit_get(const uint32_t *b, uint32_t i) { return (b[i >> 5] >> (i & 31)) & 1u; }
__device__ __forceinline__ void bit_set(uint32_t *b, uint32_t i) { b[i >> 5] |= 1u << (i & 31); }
__device__ __forceinline__ uint32_t rd_letter(uint32_t info) { return info & 0xFFu; }
__device__ __forceinline__ uint32_t rd_nal(uint32_t info) { return (info >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t rd_nin(uint32_t info) { return info >> 16; }
__device__ __forceinline__ uint32_t u4_get(const uint4 &v, uint32_t k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
__device__ __forceinline__ void u4_set(uint4 &v, uint32_t k, uint32_t x) { if (k == 0) v.x = x; else if (k == 1) v.y = x; else if (k == 2) v.z = x; else v.w = x; }

// ---- wave-level primitives on DPP (gfx9 wave_shr / row_shr / row_bcast), no LDS crossbar ----------
// value of lane-1 (lane 0 receives `fill`)
__device__ __forceinline__ int32_t wave_shr1(int32_t v, int32_t fill) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
}
// inclusive prefix max over the 64 lanes
__device__ __forceinline__ int32_t wave_scan_max(int32_t v, int32_t ident) {
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x111 /*row_shr:1*/, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x112 /*row_shr:2*/, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x114 /*row_shr:4*/, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x118 /*row_shr:8*/, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x142 /*row_bcast:15*/, 0xA, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x143 /*row_bcast:31*/, 0xC, 0xF, false));
    return v;
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /*row_shr:1*/, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /*row_shr:2*/, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /*row_shr:4*/, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /*row_shr:8*/, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /*row_bcast:15*/, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /*row_bcast:31*/, 0xC, 0xF, false);
    return v;
}
// the same scan with the maximum fused into the DPP instruction (v_max_i32_dpp: a lane whose source lies outside its row
// / is masked off keeps its own value, which is the identity of the scan): 6 VALU instructions instead of 18.  The s_nop
// cover the VALU-write -> DPP-read hazard (2 wait states), which the assembler does not insert inside inline asm.
__device__ __forceinline__ int32_t wave_scan_max_fused(int32_t v) {
    asm("s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ int32_t wave_last(int32_t v) { return __builtin_amdgcn_readlane(v, 63); }
// ordering inside ONE wavefront (other waves of the block are parked at a barrier): make this
// wave's LDS / global writes visible to its own other lanes
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// vmcnt counts stores as well as loads on gfx9.  The compiler's wait insertion is conservative across the row loop: a
// register that MAY still be the target of a vector load (the plan words loaded once per 64 rows, a predecessor row re-read
// from the record on a rare path) gets an s_waitcnt vmcnt(0) before every use -- which also waits for the record stores
// of the previous row, a full memory round trip per row.  So every vector load of the row loop is completed explicitly
// where it is issued (rarely), and the hot path carries no vmcnt wait at all.
// a wave-uniform 64-bit value as a scalar, whatever register class the compiler kept it in (readfirstlane of a scalar folds away): the
// plan pointers go into inline asm with "s" constraints, and an input the register allocator had parked in vector registers made
// the backend stop with "illegal VGPR to SGPR copy" -- which unrelated edits elsewhere in the kernel kept provoking (rounds 3-6)
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ void drain_vector_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0), expcnt / lgkmcnt untouched

// Row barrier of the DP: only LDS traffic has to be ordered across the four waves, so global
// loads/stores stay in flight (a __syncthreads() would drain vmcnt every row).
#ifdef POA_SYNCTHREADS
__device__ __forceinline__ void row_barrier() { __syncthreads(); }
#else
__device__ __forceinline__ void row_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// ---- DFS stack with spill to global --------------------------------------------------------
__device__ __forceinline__ void st_push(poa_ws &S, const poa_args &A, uint32_t v) {
    if (S.sp == POA_STACK) {
        if (S.spilled + POA_STACK / 2 > A.spill_cap) { S.err = POA_ERR_SPILL; return; }
        for (uint32_t t = threadIdx.x; t < POA_STACK / 2; t += 64) S.spill[S.spilled + t] = S.stack[t];
        wave_sync();
        for (uint32_t t = threadIdx.x; t < POA_STACK / 2; t += 64) S.stack[t] = S.stack[t + POA_STACK / 2];
        wave_sync();
        S.spilled += POA_STACK / 2;
        S.sp = POA_STACK / 2;
    }
    S.stack[S.sp++] = v;
}

__device__ __forceinline__ void st_refill(poa_ws &S) {
    if (S.sp == 0 && S.spilled > 0) {
        wave_sync();
        S.spilled -= POA_STACK / 2;
        for (uint32_t t = threadIdx.x; t < POA_STACK / 2; t += 64) S.stack[t] = S.spill[S.spilled + t];
        wave_sync();
        S.sp = POA_STACK / 2;
    }
}

// ---- spoa Graph::topological_sort ----------------------------------------------------------------
// (every helper of the kernel is force-inlined: one that stays a call takes the workspace by reference, which moves it -- and every pointer
// in it -- to scratch memory: the band kernels' first build had 644 flat and 730 scratch accesses where the barrier form has 0 and 36)
// mode 1: rank[] receives the MSA column of each node (final pass);  mode 2: srank[] receives
// the spoa rank (tie-break of best rows).  The DP itself runs in the incrementally maintained
// block order (see merge_order), which is a topological order with aligned groups contiguous.
// mode 3: the marks of everything emitted before root `root0` are preset by the caller (tie_labels);
// the sort resumes at root0 and stops at the first emitted node whose DP row is in tied[0..n_tied):
// that row is returned in n_cols.
__device__ __forceinline__ void toposort(poa_ws &S, const poa_args &A, int mode, uint32_t &n_emit, uint32_t &n_cols, uint32_t root0 = 0,
                         const uint32_t *tied = nullptr, uint32_t n_tied = 0) {
    const uint32_t n = S.n_nodes;
    if (mode != 3) {
        for (uint32_t t = threadIdx.x; t < (n + 31) / 32; t += 64) { S.done[t] = 0; S.nocheck[t] = 0; }
        wave_sync();
    }
    n_emit = 0; n_cols = 0;
    S.sp = 0; S.spilled = 0;
    const bool l0 = threadIdx.x == 0;
    for (uint32_t root = root0; root < n && !S.err; ++root) {
        if (bit_get(S.done, root)) continue;
        st_push(S, A, root);
        while (!S.err) {
            st_refill(S);
            if (S.sp == 0) break;
            const uint32_t v = S.stack[S.sp - 1];
            if (bit_get(S.done, v)) { --S.sp; continue; }
            const uint4 rec = S.nrec[v];
            const uint32_t n_in = rd_nin(rec.x), n_al = rd_nal(rec.x), in0 = rec.y, more = rec.z;
            bool valid = true;
            if (n_in > 0) {
                if (!bit_get(S.done, in0)) { st_push(S, A, in0); valid = false; }
                uint32_t e = more;
                for (uint32_t k = 1; k < n_in; ++k) {
                    const uint2 ed = S.edges[e];
                    e = ed.y;
                    if (!bit_get(S.done, ed.x)) { st_push(S, A, ed.x); valid = false; }
                }
            }
            const bool check = !bit_get(S.nocheck, v);
            uint4 al = make_uint4(0, 0, 0, 0);
            if (check && n_al > 0) {
                al = S.nal[v];
                for (uint32_t k = 0; k < n_al; ++k) {
                    const uint32_t a = u4_get(al, k);
                    if (!bit_get(S.done, a)) { st_push(S, A, a); bit_set(S.nocheck, a); valid = false; }
                }
            }
            if (!valid) continue;
            bit_set(S.done, v);
            --S.sp;                                  // top is still v
            if (!check) continue;
            for (uint32_t g = 0; g <= n_al; ++g) {   // v, then its aligned group in list order
                const uint32_t u = g == 0 ? v : u4_get(al, g - 1);
                if (mode == 3) {
                    const uint32_t row = (uint32_t)S.rank[u] + 1;
                    for (uint32_t t = 0; t < n_tied; ++t) if (tied[t] == row) { n_cols = row; return; }
                } else if (l0) {
                    if (mode == 1) S.rank[u] = (int32_t)n_cols;
                    else S.srank[u] = (int32_t)n_emit;
                }
                ++n_emit;
            }
            ++n_cols;
        }
    }
    wave_sync();
}

// ---- which tied row does spoa's sort emit first, without sorting --------------------------------
// Graph::topological_sort tries the node ids 0..n-1 as DFS roots; the DFS of a root emits every
// node of its closure (predecessors, and with a node its whole aligned group) that is still
// unmarked.  So node x is emitted during root label(x) = the smallest node id whose closure
// contains x, the marks when root r starts are exactly {x : label(x) < r}, and of two nodes with
// different labels the smaller label comes first.  One backward sweep over the rows (block order:
// successors and group mates of a row all lie at or after its group) gives every label:
// label(group) = min over its members, then every predecessor of a member takes min(own, label).
// lab[] lives in the LDS ring (unused outside the DP), 16 bits per row; wave 0 runs the sweep,
// lanes 0..3 updating the first four predecessors in parallel.
// Sweeps rows hi_start, hi_start-1, ... and returns the first row it did NOT process: it stops below
// lo_stop (at a group boundary) -- a label is final once the sweep has reached its row, so the
// labels of the tied rows only need the rows above the lowest of them.
__device__ __forceinline__ uint32_t tie_labels(poa_ws &S, uint16_t *lab, uint32_t hi_start, uint32_t lo_stop) {
    const uint32_t lane = threadIdx.x;
    uint32_t in_group = 0;
    for (int32_t hi = (int32_t)hi_start; hi >= 1; hi -= 64) {
        const int32_t r = hi - (int32_t)lane;
        uint4 pl = make_uint4(0, 0, POA_NONE, 0), plb = make_uint4(0, 0, 0, 0);
        if (r >= 1) { pl = S.plan[r - 1]; plb = S.planb[r - 1]; }
        const uint32_t cnt = (uint32_t)min(64, hi);
        for (uint32_t l = 0; l < cnt; ++l) {
            const uint32_t row = (uint32_t)hi - l;
            if (row < lo_stop && in_group == 0) return row;
            const uint32_t info = __builtin_amdgcn_readlane(pl.x, l);
            const uint32_t n_in = rd_nin(info), n_al = rd_nal(info);
            if (in_group) --in_group;
            else if (n_al) {                      // last row of a group: rows row-n_al .. row share the minimum
                uint32_t m = 0xFFFFu;
                for (uint32_t g = 0; g <= n_al; ++g) m = min(m, (uint32_t)lab[row - 1 - g]);
                wave_sync();
                if (lane <= n_al) lab[row - 1 - lane] = (uint16_t)m;
                wave_sync();
                in_group = n_al;
            }
            if (n_in == 0) continue;
            const uint32_t my = lab[row - 1];
            const uint32_t p = lane == 0 ? __builtin_amdgcn_readlane(plb.x, l) : lane == 1 ? __builtin_amdgcn_readlane(plb.y, l)
                             : lane == 2 ? __builtin_amdgcn_readlane(plb.z, l) : __builtin_amdgcn_readlane(plb.w, l);
            if (lane < 4 && lane < n_in && p) { if ((uint32_t)lab[p - 1] > my) lab[p - 1] = (uint16_t)my; }
            if (n_in > 4) {
                uint32_t e = __builtin_amdgcn_readlane(pl.z, l);
                for (uint32_t k = 4; k < n_in; ++k) {
                    const uint2 ed = S.edges[e]; e = ed.y;
                    const uint32_t q = (uint32_t)S.rank[ed.x] + 1;
                    if (lane == 0 && (uint32_t)lab[q - 1] > my) lab[q - 1] = (uint16_t)my;
                }
            }
            wave_sync();
        }
    }
    return 0;
}

// ---- DP over all rows of one alignment ---------------------------------------------------------
// Rows are 1..n (rank+1); column j (1..L) is stored at index j-1, rows are Lp wide (multiple of
// CPL).  Thread tid of the block (NW wavefronts) owns the CPL consecutive columns
// j = tid*CPL+1 .. tid*CPL+CPL of every row.
//
// Horizontal affine gaps, exactly: E[j] = max(H[j-1]+g, E[j-1]+e) with E[0] = -inf, H[0] = 0 is
// E[j] = j*e + max_{k<j} u_k with u_k = Hn[k] + g - (k+1)*e, Hn = max(diagonal, F, 0) (H without
// its E term; valid because e >= g), u_0 = g - e.  So a row is: Hn from the predecessor rows,
// an exclusive prefix-max of u (in-thread, DPP wave scan, one LDS exchange of the wave totals =
// ONE workgroup barrier per row), then H = max(Hn, E).
// The column left of a wave's first column belongs to the previous wave; its H value is
// rebuilt after the barrier from published numbers (wave total without its last column, Hn of
// that column) so later rows can use it as the diagonal source.
//
// Predecessor rows: the previous row is in registers (two register sets alternate, the row loop
// is unrolled by two so nothing is copied); rows r-1 .. r-RINGN wait in a thread-private LDS
// ring, 16 bits per cell: H (< 2^14 in the classes that use the ring) and min(H - F, 2), which
// is all max(H+g, F+e) needs because g - e = -2; older rows (1-2 % of the fetches) are read back
// from the H / F matrices.  Measured distance of predecessor rows in block order: <= 2: 35 %,
// <= 4: 71 %, <= 6: 89 %, <= 8: 96 %, <= 10: 98.7 %.
//
// Best cell: every thread keeps the maximum of its columns, the first row where it occurs and
// in how many rows it occurs; cells of columns >= L are strictly smaller than some valid cell, so
// they need no masking.  Rows are compared for ties (kernel body) only when that is not unique.
template <int CPL>
__device__ __forceinline__ void load_block(const int16_t *__restrict__ p, int32_t *v) {
    if (CPL % 4 == 0) {
        const uint2 *q = (const uint2 *)p;
#pragma unroll
        for (int u = 0; u < CPL / 4; ++u) {
            const uint2 a = q[u];
            // H is never negative (local alignment): the 16-bit record is read UNSIGNED, so a register class is exact up to
            // 5 * columns < 65536, i.e. 13107 columns
            v[4 * u] = (int32_t)(a.x & 0xFFFF); v[4 * u + 1] = (int32_t)(a.x >> 16);
            v[4 * u + 2] = (int32_t)(a.y & 0xFFFF); v[4 * u + 3] = (int32_t)(a.y >> 16);
        }
    } else {
        const uint32_t *q = (const uint32_t *)p;
#pragma unroll
        for (int u = 0; u < CPL / 2; ++u) {
            const uint32_t a = q[u];
            v[2 * u] = (int32_t)(a & 0xFFFF); v[2 * u + 1] = (int32_t)(a >> 16);
        }
    }
}

// low halves of two registers in one (v_perm_b32)
__device__ __forceinline__ uint32_t pack16(int32_t lo, int32_t hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); }

template <int CPL>
__device__ __forceinline__ void store_packed(int16_t *__restrict__ p, const uint32_t *pk) {
    if (CPL % 4 == 0) {
        uint2 *q = (uint2 *)p;
#pragma unroll
        for (int u = 0; u < CPL / 4; ++u) q[u] = make_uint2(pk[2 * u], pk[2 * u + 1]);
    } else {
        uint32_t *q = (uint32_t *)p;
#pragma unroll
        for (int u = 0; u < CPL / 2; ++u) q[u] = pk[u];
    }
}

typedef short s16x2 __attribute__((ext_vector_type(2)));

// per thread and row: two bits of min(H - E, 3) per column (packed classes)
template <int CPL> struct ebits_sel { typedef uint32_t type; };
template <> struct ebits_sel<4> { typedef uint8_t type; };
template <> struct ebits_sel<6> { typedef uint16_t type; };
template <> struct ebits_sel<8> { typedef uint16_t type; };
template <int CPL> using ebits_t = typename ebits_sel<CPL>::type;

struct dp_xchg {                 // LDS
    int4 T[2];                   // double-buffered by row parity: per wave, inclusive max of u
    int2 Q[2][4];                // for wave w: {max of u of wave w-1 without its last column, Hn of that column}
    int32_t best[16];
    int32_t TW[2][16];           // blocks of more than four wavefronts (dp_rows_wide): inclusive max of u per wavefront ...
    int2 QW[2][16];              // ... and {max of u of the wavefront to the left without its last column, Hn of that column}
    uint32_t brow;               // first row that reaches the best score
    uint32_t multi;              // 1: more than one row may reach it
    uint32_t ntl;                // number of threads whose columns reach it
    uint32_t tl[16];             // the first 16 of them
};

template <int CPL, int RINGN, int NW>
__device__ void dp_rows(poa_ws &S, dp_xchg &X, uint32_t n, uint32_t L, uint32_t Lp, int32_t &best, uint32_t &best_row, bool &multi) {
    constexpr int NT = 64 * NW;
    constexpr bool WIDE = NT * CPL > 2048;       // H may need 15 bits: the ring takes a dword per cell (H | min(H-F,2) << 16)
    constexpr int RW = WIDE ? CPL : CPL / 2;     // ring dwords per thread and row
    constexpr int NWD = (CPL + 7) / 8;           // dwords of F/E nibbles per thread and row (traceback record)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t c0 = (uint32_t)tid * CPL;
    const bool act = c0 < Lp;
    uint32_t sw[(CPL + 3) / 4];                  // this thread's CPL sequence bytes
    {
        const uint16_t *sp = (const uint16_t *)(S.sq + (act ? c0 : 0));
#pragma unroll
        for (int u = 0; u < (CPL + 3) / 4; ++u) sw[u] = 0;
#pragma unroll
        for (int u = 0; u < CPL / 2; ++u) sw[u >> 1] |= (act ? (uint32_t)sp[u] : 0u) << (16 * (u & 1));
    }
    int32_t hA[CPL], fA[CPL], hB[CPL], fB[CPL];  // the two register sets of "the previous row"
#pragma unroll
    for (int t = 0; t < CPL; ++t) { hA[t] = 0; fA[t] = POA_NEG; hB[t] = 0; fB[t] = POA_NEG; }
    int32_t hlA = 0, hlB = 0;                    // H of the column left of the thread's block in that row
    uint32_t rowA = 0xFFFFFFFFu, rowB = 0xFFFFFFFFu;
    int32_t lbest = 0;
    uint32_t lrow = 0, lcnt = 0;
    uint4 my = make_uint4(0, 0, 0, 0), myb = make_uint4(0, 0, 0, 0);
    uint32_t r0 = 0;

    // one row: predecessor registers (hP, fP, hlP, rowP) -> new row in (hN, fN, hlN, rowN)
    auto step = [&](const uint32_t i, const int32_t (&hP)[CPL], const int32_t (&fP)[CPL], const int32_t hlP, const uint32_t rowP,
                    int32_t (&hN)[CPL], int32_t (&fN)[CPL], int32_t &hlN, uint32_t &rowN) __attribute__((always_inline)) {
        const uint32_t info = __builtin_amdgcn_readlane(my.x, i), more = __builtin_amdgcn_readlane(my.z, i);
        const uint32_t pw[4] = {(uint32_t)__builtin_amdgcn_readlane(myb.x, i), (uint32_t)__builtin_amdgcn_readlane(myb.y, i),
                                (uint32_t)__builtin_amdgcn_readlane(myb.z, i), (uint32_t)__builtin_amdgcn_readlane(myb.w, i)};
        const uint32_t row = r0 + i + 1;
        const uint32_t par = row & 1u;
        const uint32_t letter = rd_letter(info), n_in = rd_nin(info);
        // Over the predecessor rows p only two per-column maxima are needed: hm = max H[p][j-1] (the match
        // score of the row is the same for every p) and fm = max of max(H[p][j] + g - e, F[p][j]), so that
        // diagonal = hm + score and F = fm + e.  The first predecessor assigns, the others take the max.
        int32_t hm[CPL], fm[CPL];
        auto pred = [&](auto first_tag, const uint32_t prow) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            int32_t hd[CPL], fd[CPL];
            if (n_in == 0) {
                // virtual start row: H = 0, F = -inf
#pragma unroll
                for (int t = 0; t < CPL; ++t) { hd[t] = 0; fd[t] = POA_G - POA_E; }
            } else if (prow == rowP) {
                const int32_t hleft = wave_shr1(hP[CPL - 1], hlP);      // H[p][j-1] of the thread's first column
#pragma unroll
                for (int t = 0; t < CPL; ++t) {
                    hd[t] = t == 0 ? hleft : hP[t - 1];
                    fd[t] = max(hP[t] + (POA_G - POA_E), fP[t]);
                }
            } else if (RINGN > 0 && row - prow <= (uint32_t)RINGN) {
                const uint32_t slot = prow % (uint32_t)(RINGN > 0 ? RINGN : 1);
                const uint32_t *rp = S.ring + ((size_t)slot * NT + tid) * RW;
                int32_t hp[CPL];
                if (WIDE) {
#pragma unroll
                    for (int t = 0; t < CPL; ++t) {
                        const uint32_t a = rp[t];
                        hp[t] = (int32_t)(a & 0xFFFFu);
                        fd[t] = hp[t] - (int32_t)(a >> 16);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < CPL / 2; ++u) {
                        const uint32_t a = rp[u];
                        hp[2 * u] = (int32_t)(a & 0x3FFFu); hp[2 * u + 1] = (int32_t)((a >> 16) & 0x3FFFu);
                        fd[2 * u] = hp[2 * u] - (int32_t)((a >> 14) & 3u); fd[2 * u + 1] = hp[2 * u + 1] - (int32_t)(a >> 30);   // H - min(H-F, 2)
                    }
                }
                const int32_t hleft = wave_shr1(hp[CPL - 1], S.lh_ring[slot * 4 + wave]);
#pragma unroll
                for (int t = 0; t < CPL; ++t) hd[t] = t == 0 ? hleft : hp[t - 1];
            } else {
                int32_t hp[CPL];
                int32_t hl = 0;
                if (act) {
                    load_block<CPL>(S.H + (uint64_t)prow * Lp + c0, hp);
                    const uint32_t *np = (const uint32_t *)S.E + ((uint64_t)prow * NT + tid) * NWD;
#pragma unroll
                    for (int t = 0; t < CPL; ++t) fd[t] = hp[t] - min((int32_t)((np[t / 8] >> (4 * (t % 8))) & 3u), 2);   // H - min(H-F, 2)
                    if (lane == 0 && wave > 0) hl = S.lh[prow * 4 + wave];
                } else {
#pragma unroll
                    for (int t = 0; t < CPL; ++t) { hp[t] = 0; fd[t] = POA_G - POA_E; }
                }
                drain_vector_loads();            // rare path: leave no vector load pending for the hot path to wait on (see drain_vector_loads)
                const int32_t hleft = wave_shr1(hp[CPL - 1], hl);
#pragma unroll
                for (int t = 0; t < CPL; ++t) hd[t] = t == 0 ? hleft : hp[t - 1];
            }
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                hm[t] = FIRST ? hd[t] : max(hm[t], hd[t]);
                fm[t] = FIRST ? fd[t] : max(fm[t], fd[t]);
            }
        };
        pred(std::true_type{}, pw[0]);
        if (n_in > 1) {
            pred(std::false_type{}, pw[1]);
            if (n_in > 2) {
                pred(std::false_type{}, pw[2]);
                if (n_in > 3) {
                    pred(std::false_type{}, pw[3]);
                    uint32_t e = more;
                    for (uint32_t k = 4; k < n_in; ++k) {
                        const uint2 ed = S.edges[e]; e = ed.y;
                        const uint32_t prow = (uint32_t)S.rank[ed.x] + 1;
                        drain_vector_loads();
                        pred(std::false_type{}, prow);
                    }
                }
            }
        }
        int32_t hn[CPL], fr[CPL];
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            const int32_t sc = ((sw[t >> 2] >> (8 * (t & 3))) & 0xFFu) == letter ? POA_M : POA_N;
            fr[t] = fm[t] + POA_E;
            hn[t] = hm[t] + sc;
        }
        int32_t ex[CPL];                         // in-thread exclusive prefix max of u
        int32_t run = POA_NEG;
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            hn[t] = max(max(hn[t], fr[t]), 0);
            ex[t] = run;
            run = max(run, hn[t] + POA_G - ((int32_t)(c0 + t) + 2) * POA_E);     // u_j, j = c0+t+1
        }
        const int32_t wincl = wave_scan_max(act ? run : POA_NEG, POA_NEG);
        const int32_t texcl = wave_shr1(wincl, POA_NEG);
        int32_t base = POA_G - POA_E;            // u_0
        int32_t hl_new = 0;
        if (NW > 1) {
            if (lane == 63) {
                ((int32_t *)&X.T[par])[wave] = wincl;
                if (wave < NW - 1) X.Q[par][wave + 1] = make_int2(max(texcl, ex[CPL - 1]), hn[CPL - 1]);
            }
            row_barrier();
            const int4 T = X.T[par];             // both reads are in flight together; selects are scalar (wave is uniform)
            const int2 q = X.Q[par][wave];
            const int32_t t0 = wave > 0 ? T.x : POA_NEG, t1 = wave > 1 ? T.y : POA_NEG, t2 = wave > 2 ? T.z : POA_NEG;
            const int32_t b0 = wave > 1 ? T.x : POA_NEG, b1 = wave > 2 ? T.y : POA_NEG;
            base = max(max(base, t0), max(t1, t2));
            if (wave > 0) {
                const int32_t bp = max(max(POA_G - POA_E, b0), b1);
                const int32_t c0w = (int32_t)((uint32_t)wave * 64u * CPL);        // 1-based index of the column left of the wave
                hl_new = max(q.y, max(bp, q.x) + c0w * POA_E);
                if (lane == 0) S.lh[row * 4 + wave] = hl_new;
            }
        }
        base = max(base, texcl);
        int32_t ev[CPL];
        int32_t lm = 0;
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            const int32_t j = (int32_t)(c0 + t) + 1;
            ev[t] = max(base, ex[t]) + j * POA_E;
            hN[t] = max(hn[t], ev[t]);
            fN[t] = fr[t];
            lm = max(lm, hN[t]);
        }
        hlN = hl_new; rowN = row;
        {                                        // per-thread best: value, first row, number of rows
            const bool gt = lm > lbest, eq = lm == lbest;
            lcnt = gt ? 1u : lcnt + (eq ? 1u : 0u);
            lrow = gt ? row : lrow;
            lbest = gt ? lm : lbest;
        }
        uint32_t pkH[CPL / 2], pkF[CPL / 2];
#pragma unroll
        for (int u = 0; u < CPL / 2; ++u) {
            pkH[u] = pack16(hN[2 * u], hN[2 * u + 1]);
            pkF[u] = pack16(fr[2 * u], fr[2 * u + 1]);       // F >= g: every row has a (possibly virtual) predecessor with H >= 0
        }
        if (RINGN > 0) {
            const uint32_t slot = row % (uint32_t)(RINGN > 0 ? RINGN : 1);
            uint32_t *rp = S.ring + ((size_t)slot * NT + tid) * RW;
            if (WIDE) {
#pragma unroll
                for (int t = 0; t < CPL; ++t) rp[t] = (uint32_t)hN[t] | ((uint32_t)min(hN[t] - fr[t], 2) << 16);
            } else {
#pragma unroll
                for (int u = 0; u < CPL / 2; ++u) {
                    s16x2 hh, ff;
                    __builtin_memcpy(&hh, &pkH[u], 4); __builtin_memcpy(&ff, &pkF[u], 4);
                    const s16x2 two = {2, 2};
                    const s16x2 d = __builtin_elementwise_min(hh - ff, two);
                    uint32_t dw;
                    __builtin_memcpy(&dw, &d, 4);
                    rp[u] = pkH[u] | (dw << 14);
                }
            }
            if (lane == 0) S.lh_ring[slot * 4 + wave] = hl_new;
        }
        if (act) {
            // the traceback record: H as int16 plus a nibble per column, min(H-F,3) | min(H-E,3) << 2 (exact: see dp_rows_v3)
            store_packed<CPL>(S.H + (uint64_t)row * Lp + c0, pkH);
            uint32_t *np = (uint32_t *)S.E + ((uint64_t)row * NT + tid) * NWD;
#pragma unroll
            for (int w = 0; w < NWD; ++w) {
                uint32_t v = 0;
#pragma unroll
                for (int t = 8 * w; t < 8 * w + 8 && t < CPL; ++t)
                    v |= ((uint32_t)min(hN[t] - fr[t], 3) | ((uint32_t)min(hN[t] - ev[t], 3) << 2)) << (4 * (t - 8 * w));
                np[w] = v;
            }
        }
    };

    for (r0 = 0; r0 < n; r0 += 64) {
        const uint32_t nb = min(64u, n - r0);
        my = make_uint4(0, 0, 0, 0); myb = make_uint4(0, 0, 0, 0);
        if ((uint32_t)lane < nb) { my = S.plan[r0 + lane]; myb = S.planb[r0 + lane]; }
        drain_vector_loads();
        for (uint32_t i = 0; i < nb; i += 2) {
            step(i, hA, fA, hlA, rowA, hB, fB, hlB, rowB);
            if (i + 1 < nb) step(i + 1, hB, fB, hlB, rowB, hA, fA, hlA, rowA);
        }
    }
    // block-wide: best score, the first row that reaches it, and whether other rows may reach it too
    const int32_t wb = wave_last(wave_scan_max(lbest, 0));
    if (lane == 0) X.best[wave] = wb;
    if (tid == 0) { X.brow = 0xFFFFFFFFu; X.multi = 0; X.ntl = 0; }
    __syncthreads();
    best = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) best = max(best, X.best[w]);
    const bool mine = best > 0 && lbest == best;
    if (mine) atomicMin(&X.brow, lrow);
    __syncthreads();
    best_row = best > 0 ? X.brow : 0u;
    if (mine) {
        if (lcnt != 1 || lrow != best_row) X.multi = 1;
        const uint32_t slot = atomicAdd(&X.ntl, 1u);
        if (slot < 16) X.tl[slot] = (uint32_t)tid;
    }
    __syncthreads();
    multi = X.multi != 0;
}

// ---- the same row recurrence on packed int16 pairs (classes up to 2560 columns) -------------------
// Two columns per register (v_pk_add/max/min/sub_i16): H <= 5*2560, u = Hn + g - (j+1)e <= 28158 and
// j*e >= -15360 all fit 16 bits (H < 2^14 for the ring word); F never drops below g (every row has a predecessor with H >= 0, the
// virtual one included) so -16384 can stand for "no F".  Shifted neighbours come from v_alignbit on
// adjacent pairs (the left thread's last pair arrives by DPP), stores and ring words need no packing.
// Same results as dp_rows, ~15 % fewer VALU instructions and half the register window.
__device__ __forceinline__ s16x2 as_pk(uint32_t v) { s16x2 r; __builtin_memcpy(&r, &v, 4); return r; }
__device__ __forceinline__ uint32_t as_u(s16x2 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
__device__ __forceinline__ s16x2 pk_splat(int v) { const s16x2 r = {(short)v, (short)v}; return r; }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s16x2 pk_min(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
// (prev.hi, cur.lo): the pair one column to the left of `cur`
__device__ __forceinline__ s16x2 pk_left(uint32_t cur, uint32_t prev) { return as_pk(__builtin_amdgcn_alignbit(cur, prev, 16)); }

// ---- the packed row recurrence, third form (classes up to 2560 columns) ------------------------------
// Rounds 1-2 ran the same arithmetic with the previous row in registers and the plan in vector registers; round 3 rebuilt
// everything around the per-pair work, because the kernel is bound by VALU issue and by the length of a row's instruction
// stream (profiles/README.md, round 3):
//   * the row plan comes through the scalar cache: three s_load_dwordx4 per row (the plan arrays read as constant address
//     space after an s_dcache_inv), prefetched one row ahead -- no v_readlane per plan word, no plan registers, no 64-row chunks;
//   * every predecessor comes from the LDS ring (or, beyond RING rows, from the record): no register copy of the previous
//     row, so the row body exists once per parity and the ring slot is `row & (RING - 1)`;
//   * the record word is decoded with v_pk_lshrrev_b16 (4 instructions per pair and predecessor instead of 5);
//   * the in-thread prefix of u: v = max(RUN, u) ; EX = alignbit(v, RUN) ; RUN = max(v, swap(v)) -- 3 instructions per pair
//     instead of 5 (the half swap is an op_sel of v_pk_max_i16).
#ifndef POA_SHIFT_ONCE
#define POA_SHIFT_ONCE 0          // 1: maxima over the predecessors on the unshifted H words, ONE shift per row (1.18 x (NP + 1) fewer instructions
                                  // per row, but the shift then sits on the row's critical path: -1.3 % in an interleaved A/B, three rounds on one box)
#endif
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4 *cplan_t;      // wave-uniform loads from it are s_load_dwordx4

template <int CPL, int RING, int NW>
__device__ void dp_rows_v3(poa_ws &S, dp_xchg &X, uint32_t n, uint32_t L, uint32_t Lp, int32_t &best, uint32_t &best_row, bool &multi) {
    constexpr int NT = 64 * NW, NP = CPL / 2;
    static_assert(RING > 0, "the rows of the last RING rows live in LDS");
    static_assert(NT * CPL <= 2560 && NW <= 4, "packed rows: u = Hn + g - (j+1)e must fit 16 bits");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t c0 = (uint32_t)tid * CPL;
    const bool act = c0 < Lp;
    uint32_t sw[(CPL + 3) / 4];                  // this thread's CPL sequence bytes
    {
        const uint16_t *sp = (const uint16_t *)(S.sq + (act ? c0 : 0));
#pragma unroll
        for (int u = 0; u < (CPL + 3) / 4; ++u) sw[u] = 0;
#pragma unroll
        for (int u = 0; u < CPL / 2; ++u) sw[u >> 1] |= (act ? (uint32_t)sp[u] : 0u) << (16 * (u & 1));
    }
    // Match scores by table look-up (v_perm_b32) when the pack's alphabet allows it: (c >> 1) & 3 maps A, C, T/U, G to 0, 1, 2, 3,
    // so a row's letter gives an 8-byte table {5 or -4 as int16 halves} and a per-thread selector word per column pair picks
    // the two scores in ONE instruction (against two compares, two selects and a pack).  Columns beyond the sequence get -1.
    // T and U share an index: a pack that mixes them (or holds any other byte) takes the compare path (S.plain == 0).
    uint32_t SEL[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const uint32_t ca = (sw[(2 * u) >> 2] >> (8 * ((2 * u) & 3))) & 0xFFu, cb = (sw[(2 * u + 1) >> 2] >> (8 * ((2 * u + 1) & 3))) & 0xFFu;
        const uint32_t ia = (ca >> 1) & 3u, ib = (cb >> 1) & 3u;
        const uint32_t sa = ca ? (ia | ((ia + 4u) << 8)) : 0x0D0Du, sb = cb ? (ib | ((ib + 4u) << 8)) : 0x0D0Du;
        SEL[u] = sa | (sb << 16);
    }
    const bool plain = S.plain != 0;
    s16x2 JE[NP], UC[NP];                        // per column: j*e and g - (j+1)*e
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int j0 = (int)c0 + 2 * u + 1, j1 = j0 + 1;
        const s16x2 je = {(short)(j0 * POA_E), (short)(j1 * POA_E)};
        const s16x2 uc = {(short)(POA_G - (j0 + 1) * POA_E), (short)(POA_G - (j1 + 1) * POA_E)};
        JE[u] = je; UC[u] = act ? uc : pk_splat(-30000);      // threads beyond the row: u = Hn - 30000 < 0 < every valid u (>= 4), no select before the scan
    }
    s16x2 MXA = pk_splat(0);                    // running maximum of this thread's columns over all rows (pairs)
    const bool wave_act = (uint32_t)wave * 64u * CPL < Lp;
    uint32_t *const ring_thr = S.ring + (size_t)tid * NP;              // slot s of this thread: ring_thr + s * NT * NP
    // slot s of this wavefront: lhr[4 * s].  The LDS offset is made opaque in a VECTOR register: as a uniform value its address
    // arithmetic went through the scalar unit (shift, add, v_mov per predecessor) -- the scalar unit issues as many instructions per
    // cell as the vector unit in this kernel, one v_lshl_add does the same
    uint32_t lhr_off = (uint32_t)wave;
    asm volatile("" : "+v"(lhr_off));
    uint32_t *const lhr = (uint32_t *)S.lh_ring + lhr_off;
    uint32_t *const Hrec = (uint32_t *)S.H;                            // the record, a dword per column pair

    // the plan through the scalar cache: invalidate it first (the plan was just rewritten by vector stores, which are in L2
    // once the barrier ahead of this function has been passed); the pointers depend on the invalidate so that no load moves above it
    // Round 4: the row's COMPACT plan record (poa_kernel, step 2: letter, in-degree, the distances to the first eight
    // predecessor rows in a byte each): one s_load_dwordx4 per row instead of three, 4 + 4 plan registers instead of 12 + 12 (the
    // kernel spills SGPRs: reloads were v_readlane in the row loop), a distance compared with the ring length instead of a
    // subtraction per predecessor.  A predecessor beyond the ring (a few per cent) takes its row from planb / planc.
    uint64_t ppa = uni64((uint64_t)S.plan), ppb = uni64((uint64_t)S.planb), ppc = uni64((uint64_t)S.planc), ppd = uni64((uint64_t)S.pland);
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+s"(ppa), "+s"(ppb), "+s"(ppc), "+s"(ppd) : : "memory");
    const cplan_t cpa = (cplan_t)ppa, cpb = (cplan_t)ppb, cpc = (cplan_t)ppc, cpd = (cplan_t)ppd;

    auto step = [&](auto par_tag, const uint32_t row, const u32x4 pd) __attribute__((always_inline)) {
        constexpr uint32_t par = decltype(par_tag)::value;
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));                      // lane tests are redone per row: a hoisted mask ends up in a spilled SGPR pair (two v_readlane per use)
        const uint32_t letter = pd.x & 0xFFu;
        const uint32_t n_in = (pd.x >> 8) & 0xFFu;           // capped at 255: the walk beyond the eighth in-edge reads the real one
        uint32_t dist[8];                        // distances row - predecessor row, a byte each (255: 255 or more -- far beyond the ring, the row comes from planb / planc)
#pragma unroll
        for (int k = 0; k < 4; ++k) { dist[k] = (pd.y >> (8 * k)) & 0xFFu; dist[4 + k] = (pd.z >> (8 * k)) & 0xFFu; }
        s16x2 HM[NP], FM[NP];                    // maxima over the predecessors: H[p][j-1] and max(H[p][j] + g - e, F[p][j])
        uint32_t raw[4][NP], rawl[4];
        // k: which of the four fetch slots; kp: which in-edge (0..7: the plan holds its row; 8: `far_row` is given)
        auto fetch = [&](const int k, const int kp, const uint32_t d, const uint32_t far_row) __attribute__((always_inline)) {
            if (d <= (uint32_t)RING) {
                const uint32_t slot = (row - d) % (uint32_t)RING;
                const uint32_t *rp = ring_thr + slot * (uint32_t)(NT * NP);
                if constexpr (NP == 2) { const uint2 a2 = *(const uint2 *)rp; raw[k][0] = a2.x; raw[k][1] = a2.y; }
                else if constexpr (NP == 4) { const uint4 a4 = *(const uint4 *)rp; raw[k][0] = a4.x; raw[k][1] = a4.y; raw[k][2] = a4.z; raw[k][3] = a4.w; }
                else {
#pragma unroll
                    for (int u = 0; u < NP; ++u) raw[k][u] = rp[u];
                }
                rawl[k] = lhr[4 * slot];
            } else {
                // beyond the ring (a few per cent of the fetches): the row itself from the full plan (the byte may be saturated)
                uint32_t prow = kp >= 8 ? far_row : kp >= 4 ? cpc[row - 1][kp - 4] : cpb[row - 1][kp];
                // the scalar load lands HERE: left pending to the end of this (rare) branch, the compiler waits lgkmcnt(0) before the
                // next predecessor's ring address on the common path too (same scalar register) -- which serialised the LDS reads of
                // a row's predecessors
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(prow));
                // No divergent control flow in here (threads beyond the row and the lanes that need no left neighbour load from a valid
                // address and drop the value): with exec-mask regions inside, the compiler gave the COMMON path a flag, a second test
                // and a second branch per predecessor.
                const uint32_t cc = act ? c0 : 0u;
                const uint32_t *hq = (const uint32_t *)(S.H + (uint64_t)prow * Lp + cc);
                const bool need_left = lane == 0 && wave > 0 && act;
                const uint32_t hl16 = ((const uint16_t *)hq)[need_left ? -1 : 0];
#pragma unroll
                for (int u = 0; u < NP; ++u) { const uint32_t w = hq[u]; raw[k][u] = act ? w : 0x80008000u; }      // H = 0, H - F = 2: what a column beyond the row decodes to
                rawl[k] = need_left ? (hl16 & 0x3FFFu) << 16 : 0u;
                drain_vector_loads();            // rare path (1-2 % of the fetches): nothing stays pending past it
            }
        };
        uint32_t WL = 0;                         // H of the column left of this wavefront (high half), maximum over the predecessors
        auto combine = [&](auto first_tag, const uint32_t (&w)[NP], const uint32_t wl) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            uint32_t hp[NP];
            s16x2 FD[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                hp[u] = w[u] & 0x3FFF3FFFu;
                u16x2 wu;
                __builtin_memcpy(&wu, &w[u], 4);
                const u16x2 d = __builtin_elementwise_min(wu >> (u16x2){14, 14}, (u16x2){2, 2});      // min(H - F, 2)
                s16x2 ds;
                __builtin_memcpy(&ds, &d, 4);
                FD[u] = as_pk(hp[u]) - ds;                                                            // max(H + g - e, F)
            }
            // the maximum over the predecessors is taken on the UNSHIFTED H words (and on the words of the lane to the left of
            // the wavefront); the shift by one column commutes with it and is done once per row below, not once per predecessor
#if POA_SHIFT_ONCE
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                HM[u] = FIRST ? as_pk(hp[u]) : pk_max(HM[u], as_pk(hp[u]));
                FM[u] = FIRST ? FD[u] : pk_max(FM[u], FD[u]);
            }
            WL = FIRST ? wl : max(WL, wl);
#else
            const uint32_t left = (uint32_t)wave_shr1((int32_t)hp[NP - 1], (int32_t)wl);
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const s16x2 HD = pk_left(hp[u], u == 0 ? left : hp[u - 1]);
                HM[u] = FIRST ? HD : pk_max(HM[u], HD);
                FM[u] = FIRST ? FD[u] : pk_max(FM[u], FD[u]);
            }
#endif
        };
        if (n_in == 0) {                         // virtual start row: H = 0, F = -inf
#pragma unroll
            for (int u = 0; u < NP; ++u) { HM[u] = pk_splat(0); FM[u] = pk_splat(POA_G - POA_E); }
        } else {
            fetch(0, 0, dist[0], 0);
            if (n_in > 1) fetch(1, 1, dist[1], 0);
            if (n_in > 2) fetch(2, 2, dist[2], 0);
            if (n_in > 3) fetch(3, 3, dist[3], 0);
            combine(std::true_type{}, raw[0], rawl[0]);
            if (n_in > 1) combine(std::false_type{}, raw[1], rawl[1]);
            if (n_in > 2) combine(std::false_type{}, raw[2], rawl[2]);
            if (n_in > 3) {
                combine(std::false_type{}, raw[3], rawl[3]);
                if (n_in > 4) {
                    fetch(0, 4, dist[4], 0);
                    if (n_in > 5) fetch(1, 5, dist[5], 0);
                    if (n_in > 6) fetch(2, 6, dist[6], 0);
                    if (n_in > 7) fetch(3, 7, dist[7], 0);
                    combine(std::false_type{}, raw[0], rawl[0]);
                    if (n_in > 5) combine(std::false_type{}, raw[1], rawl[1]);
                    if (n_in > 6) combine(std::false_type{}, raw[2], rawl[2]);
                    if (n_in > 7) combine(std::false_type{}, raw[3], rawl[3]);
                    if (n_in > 8) {
                        uint32_t e = pd.w;
                        const uint32_t n_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)rd_nin(cpa[row - 1].x));         // the record's count stops at 255
                        for (uint32_t k = 8; k < n_all; ++k) {
                            const uint2 ed = S.edges[e]; e = ed.y;
                            const uint32_t prow = (uint32_t)__builtin_amdgcn_readfirstlane(S.rank[ed.x]) + 1;
                            drain_vector_loads();
                            fetch(0, 8, row - prow, prow);
                            combine(std::false_type{}, raw[0], rawl[0]);
                        }
                    }
                }
            }
        }
        if (POA_SHIFT_ONCE && n_in != 0) {       // H[p][j-1]: the maxima shifted by one column (lane 0 takes the word of the lane left of the wavefront)
            const uint32_t left = (uint32_t)wave_shr1((int32_t)as_u(HM[NP - 1]), (int32_t)WL);
#pragma unroll
            for (int u = NP - 1; u >= 0; --u) HM[u] = pk_left(as_u(HM[u]), u == 0 ? left : as_u(HM[u - 1]));
        }
        // Hn = max(diagonal, F, 0); u = Hn + g - (j+1)e; in-thread exclusive prefix max of u (pair by pair)
        s16x2 HNp[NP], EX[NP], SC[NP], FN[NP];
        s16x2 RUN = pk_splat(-32768);
        if (plain) {
            const uint32_t li = (letter >> 1) & 3u;
            const uint32_t klo = 0xFCFCFCFCu ^ (0xF9u << (8u * li)), khi = 0xFFFFFFFFu ^ (0xFFu << (8u * li));      // low / high bytes of {-4, -4, -4, -4} with 5 at li
#pragma unroll
            for (int u = 0; u < NP; ++u) SC[u] = as_pk(__builtin_amdgcn_perm(khi, klo, SEL[u]));
        } else {
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int32_t s0 = ((sw[(2 * u) >> 2] >> (8 * ((2 * u) & 3))) & 0xFFu) == letter ? POA_M : POA_N;
                const int32_t s1 = ((sw[(2 * u + 1) >> 2] >> (8 * ((2 * u + 1) & 3))) & 0xFFu) == letter ? POA_M : POA_N;
                SC[u] = as_pk(pack16(s0, s1));
            }
        }
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            FN[u] = FM[u] + pk_splat(POA_E);
            HNp[u] = pk_max(pk_max(HM[u] + SC[u], FN[u]), pk_splat(0));
            const s16x2 v = pk_max(RUN, HNp[u] + UC[u]);                                   // (max(run, u_a), max(run, u_b))
            EX[u] = as_pk(__builtin_amdgcn_alignbit(as_u(v), as_u(RUN), 16));              // (run, max(run, u_a)): both halves of RUN are equal
            RUN = pk_max(v, __builtin_shufflevector(v, v, 1, 0));                          // max(run, u_a, u_b) in both halves
        }
        const int32_t run = (int32_t)(int16_t)(as_u(RUN) & 0xFFFFu);
        const int32_t wincl = wave_scan_max_fused(run);
        const int32_t texcl = wave_shr1(wincl, POA_NEG);
        int32_t base = POA_G - POA_E;            // u_0
        int32_t hl_new = 0;
        if (NW > 1) {
            if (lane_o == 63) {
                const int32_t ex_last = (int32_t)as_u(EX[NP - 1]) >> 16, hn_last = (int32_t)as_u(HNp[NP - 1]) >> 16;
                ((int32_t *)&X.T[par])[wave] = wincl;
                if (wave < NW - 1) X.Q[par][wave + 1] = make_int2(max(texcl, ex_last), hn_last);
            }
            row_barrier();
            if (wave > 0) {
                const int4 T = X.T[par];
                const int2 q = X.Q[par][wave];
                const int32_t Tx = __builtin_amdgcn_readfirstlane(T.x), Ty = __builtin_amdgcn_readfirstlane(T.y), Tz = __builtin_amdgcn_readfirstlane(T.z);
                const int32_t t1 = wave > 1 ? Ty : POA_NEG, t2 = wave > 2 ? Tz : POA_NEG;
                const int32_t b0 = wave > 1 ? Tx : POA_NEG, b1 = wave > 2 ? Ty : POA_NEG;
                base = max(max(base, Tx), max(t1, t2));
                const int32_t qx = __builtin_amdgcn_readfirstlane(q.x), qy = __builtin_amdgcn_readfirstlane(q.y);
                const int32_t bp = max(max(POA_G - POA_E, b0), b1);
                const int32_t c0w = (int32_t)((uint32_t)wave * 64u * CPL);
                hl_new = max(qy, max(bp, qx) + c0w * POA_E);
            }
        }
        base = max(base, texcl);
        const s16x2 BASE = as_pk(pack16(base, base));
        uint32_t W[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const s16x2 EV = pk_max(BASE, EX[u]) + JE[u];
            const s16x2 HN = pk_max(HNp[u], EV);
            MXA = pk_max(MXA, HN);
            // One word per pair serves the ring, later rows and the traceback: H (14 bits) and min(H - F, 3).  The traceback's tests
            // on F (and E) can only hold where H - F (H - E) <= 2, and a clipped value H - 3 can never satisfy them (H of the cell
            // above / left is at most 8 higher), so two bits per cell are an exact record of F.  E is not recorded at all: the
            // traceback rebuilds it from the row (Eat).
            W[u] = as_u(HN) | (as_u(pk_min(HN - FN[u], pk_splat(3))) << 14);
        }
        {
            const uint32_t slot = row % (uint32_t)RING;
            uint32_t *rp = ring_thr + slot * (uint32_t)(NT * NP);
            if constexpr (NP == 2) *(uint2 *)rp = make_uint2(W[0], W[1]);
            else if constexpr (NP == 4) *(uint4 *)rp = make_uint4(W[0], W[1], W[2], W[3]);
            else {
#pragma unroll
                for (int u = 0; u < NP; ++u) rp[u] = W[u];
            }
            if (lane_o == 0) lhr[4 * slot] = (uint32_t)hl_new << 16;
            if (act) {
                uint32_t *hq = Hrec + ((uint64_t)row * Lp >> 1) + (c0 >> 1);
                if (NP % 2 == 0) {
#pragma unroll
                    for (int u = 0; u < NP / 2; ++u) ((uint2 *)hq)[u] = make_uint2(W[2 * u], W[2 * u + 1]);
                } else {
#pragma unroll
                    for (int u = 0; u < NP; ++u) hq[u] = W[u];
                }
            }
        }
    };

    if (!wave_act) {
        if (NW > 1) for (uint32_t r = 0; r < n; ++r) row_barrier();
    } else {
        // two rows per trip, two register sets for the plan record: the record of the row after next is requested into the set
        // the row just finished has left (no copies from a "next" set into a "current" one)
        u32x4 p0 = cpd[0], p1 = p0;
        for (uint32_t row = 1; row <= n; row += 2) {
            if (row < n) p1 = cpd[row];
            step(std::integral_constant<uint32_t, 1>{}, row, p0);
            if (row + 1 <= n) {
                if (row + 1 < n) p0 = cpd[row + 1];
                step(std::integral_constant<uint32_t, 0>{}, row + 1, p1);
            }
        }
    }
    // block-wide best score and the threads whose columns reach it (the first row that reaches it comes from a rescan of
    // those threads' columns in the record, kernel body)
    const int32_t lbest = act ? max((int32_t)(int16_t)(as_u(MXA) & 0xFFFFu), (int32_t)as_u(MXA) >> 16) : 0;
    const int32_t wb = wave_last(wave_scan_max(lbest, 0));
    if (lane == 0) X.best[wave] = wb;
    if (tid == 0) { X.brow = 0xFFFFFFFFu; X.multi = 1; X.ntl = 0; }
    __syncthreads();
    best = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) best = max(best, X.best[w]);
    if (best > 0 && lbest == best) {
        const uint32_t slot = atomicAdd(&X.ntl, 1u);
        if (slot < 16) X.tl[slot] = (uint32_t)tid;
    }
    __syncthreads();
    best_row = 0;
    multi = best > 0;
}

// ---- LDS words that several wavefronts poll (mailboxes, counters) -------------------------------------------------------------
// They are read and written through VOLATILE pointers (no access may be dropped, merged or moved across another) that carry the LDS
// address space in their type: a volatile access through a generic pointer is not narrowed by the compiler and comes out as
// flat_load / flat_store with system-coherence bits and an s_waitcnt vmcnt(0) behind every store -- which waits for the record
// stores of the row (a memory round trip per row: round 4's first pipeline kernel did exactly that).
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef volatile lds_u32 *sk_ptr;
__device__ __forceinline__ uint32_t sk_ld(sk_ptr p) { return *p; }
__device__ __forceinline__ void sk_st(sk_ptr p, uint32_t v) { *p = v; }
// inclusive prefix max over the lanes of a word whose two halves are EQUAL: as a signed 32-bit number such a word orders like
// its halves, so the 32-bit DPP scan works on it directly and the result is again a duplicated word
__device__ __forceinline__ uint32_t wave_scan_max_dup(uint32_t v) { return (uint32_t)wave_scan_max_fused((int32_t)v); }

// (Round 4's skewed wavefront pipeline, dp_rows_sk -- no barrier per row, wavefronts coupled by an LDS mailbox, record words or
// ready-made terms in the ring -- was the under-filled device's form for one round.  The one-team instance of dp_rows_mt below is
// the same pipeline with the lean row and matches or beats it at every load measured (profiles/round5_forms_by_load.txt); it was
// removed in round 5 with its experiment table.)

// ---- the packed row recurrence on TEAMS of wavefronts (round 5; classes up to 2560 columns, under-filled device) ------------
// Round 4's finding: a pack that has (most of) a CU to itself is bound by ONE wavefront's instruction stream -- ~240 instructions
// per row, ~100 of them scalar (plan decoding, a distance test and a branch per predecessor, ring-slot arithmetic) -- and by nothing
// else: every wavefront of the skewed pipeline runs every row.  Two things change here.
//
// (1) Rows that do not depend on each other run at the same time.  Measured on the row plans of 200-read packs at 10 % error
//     (profiles/round5_dag_width.txt): the longest dependency chain of an alignment's rows is 0.45 of its rows (only a third of
//     the rows have a predecessor in the row right before them), and dealing the rows round-robin, in row order, to four teams
//     reaches 0.453.  So the workgroup is T teams x NW column blocks: wavefront (t, w) runs the columns of block w of the rows
//     t+1, t+1+T, ...  It waits for
//       * its predecessor rows in ITS column block: done[w][t'] = the last row team t' has finished in block w; "every row
//         <= x is final" is min over t' of done[w][t'] + T > x -- ONE 16-byte LDS read and one comparison with a number the plan
//         carries (the largest predecessor row, or row - slack: that also makes the ring slot a row overwrites dead);
//       * the wavefront to its left of the SAME team for the prefix maximum and the H of its last column (a mailbox of MT_MD
//         entries per wavefront, back-pressure on the right neighbour's done counter).
//     All waits are for smaller rows, or the same row in a smaller block: no cycles (simulated on host threads,
//     tests/stubs/mt_protocol_sim.cpp).
// (2) The row is on a scalar diet.  The plan record of a row is 32 bytes (one s_load_dwordx8, fetched a row
//     ahead behind the row's first LDS wait) and holds everything READY-MADE: the two halves of the score table, the LDS byte
//     offset of the row's own ring slot and of the ring slots of its first eight predecessors, the row number to wait for.
//     Missing predecessors (and the ones beyond the ring) point once more at the entry of a predecessor that IS in the ring (a
//     maximum does not mind seeing a term twice), so the common path -- up to four in-edges, all in the ring: 88 % of the
//     rows -- has no branch at all and no per-predecessor scalar work; a second group of four takes one uniform branch; in-edges
//     beyond the ring or beyond the eighth take the slow path.  One-lane LDS writes (counters, mailbox) are 64-lane writes whose
//     other lanes hit a junk word: no exec-mask games.  The ring (ready-made terms, 4 bytes per cell) is sized at launch from
//     the LDS a workgroup gets, not at compile time: the kernel only sees offsets.
#ifndef MT_MD
#define MT_MD 4                                   // mailbox entries per wavefront (rows of its team it may run ahead of its right neighbour)
#endif
#define MT_BIG 0x7FFFFF00
#define MT_UNIT 11                                // ring offsets travel in units of 2 KB (a byte each in the plan record)
#define MT_MORE4 1u                               // plan flags: in-edges 5 .. 8 are in use
#define MT_SLOW 2u                                // ... some in-edge lies beyond the ring, or there are more than eight, or none is in the ring
#define MT_NOBASE 4u                              // no in-edge in the ring (virtual start row, or all beyond it): the maxima start from the neutral element
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) u32x8 *cplanm_t;
typedef __attribute__((address_space(3))) uint32_t *lds_p;
template <int NW, int T> struct alignas(16) mt_sync {
    int32_t done[NW][4];                          // [w][t]: last row team t has finished in column block w (teams that do not exist: MT_BIG)
    uint32_t mcnt[NW][4];                         // [w][t]: last row whose mailbox entry (w -> w + 1) team t has published
    uint32_t mail[NW][T][MT_MD][2];               // {prefix max of u over columns 1 .. last column of w (both halves), H of that column << 16}
    uint32_t junk[64 * 2 + 8];                    // where the other 63 lanes of a one-lane write go
    uint32_t abort;                               // a wavefront gave up waiting (MT_SPIN_LIMIT polls): everybody leaves, the pack fails with POA_ERR_SYNC
};
#ifndef MT_POLL_FULL
#define MT_POLL_FULL 0                            // up to this many teams a waiting row re-reads its ring entries with every poll of the counters (beyond: the counters alone, the entries once after them)
#endif
#ifndef MT_POLL_SLEEP
#define MT_POLL_SLEEP 1
#endif
#ifndef MT2_MINWAVES
#define MT2_MINWAVES 6                            // two teams, 1024-column class: 80 registers, three packs per CU (measured: 680 against 492 GCUPS at 768 packs); the wider classes need their 128
#endif
#ifndef MT_SPIN_LIMIT
#define MT_SPIN_LIMIT (1u << 20)                  // polls of one wait (>= 0.1 s): every spin is bounded -- a protocol error must cost a failed call, not a hung device
#endif

template <int CPL, int NW, int T>
__device__ void dp_rows_mt(poa_ws &S, dp_xchg &X, const poa_args &A, uint32_t n, uint32_t L, uint32_t Lp, int32_t &best, uint32_t &best_row, bool &multi) {
    constexpr int NTC = 64 * NW, NP = CPL / 2, RW = 2 * NP;
    constexpr uint32_t SLOTB = (uint32_t)NTC * RW * 4u;                 // bytes of one ring slot
    static_assert(NTC * CPL <= 2560 && CPL % 2 == 0 && T <= 4, "packed rows: u = Hn + g - (j+1)e must fit 16 bits");
    static_assert(SLOTB % (1u << MT_UNIT) == 0 && 24u * (SLOTB >> MT_UNIT) <= 255u, "a ring slot offset fits a byte of 2 KB units");
    __shared__ mt_sync<NW, T> Y;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv % NW, t = wv / NW;                                  // column block, team
    const uint32_t ct = (uint32_t)w * 64u + (uint32_t)lane;              // the thread's place in a row
    const uint32_t c0 = ct * CPL;
    const bool act = c0 < Lp;
    uint32_t sw[(CPL + 3) / 4];                  // this thread's CPL sequence bytes
    {
        const uint16_t *sp = (const uint16_t *)(S.sq + (act ? c0 : 0));
#pragma unroll
        for (int u = 0; u < (CPL + 3) / 4; ++u) sw[u] = 0;
#pragma unroll
        for (int u = 0; u < CPL / 2; ++u) sw[u >> 1] |= (act ? (uint32_t)sp[u] : 0u) << (16 * (u & 1));
    }
    uint32_t SEL[NP];                            // score table selectors (see dp_rows_v3)
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const uint32_t ca = (sw[(2 * u) >> 2] >> (8 * ((2 * u) & 3))) & 0xFFu, cb = (sw[(2 * u + 1) >> 2] >> (8 * ((2 * u + 1) & 3))) & 0xFFu;
        const uint32_t ia = (ca >> 1) & 3u, ib = (cb >> 1) & 3u;
        const uint32_t sa = ca ? (ia | ((ia + 4u) << 8)) : 0x0D0Du, sb = cb ? (ib | ((ib + 4u) << 8)) : 0x0D0Du;
        SEL[u] = sa | (sb << 16);
    }
    const bool plain = S.plain != 0;
    s16x2 JE[NP], UC[NP];                        // per column: j*e and g - (j+1)*e
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int j0 = (int)c0 + 2 * u + 1, j1 = j0 + 1;
        const s16x2 je = {(short)(j0 * POA_E), (short)(j1 * POA_E)};
        const s16x2 uc = {(short)(POA_G - (j0 + 1) * POA_E), (short)(POA_G - (j1 + 1) * POA_E)};
        JE[u] = je; UC[u] = act ? uc : pk_splat(-30000);      // threads beyond the row: u = Hn - 30000 < 0 < every valid u, no select before the scan
    }
    s16x2 MXA = pk_splat(0);                    // running maximum of this thread's columns over this team's rows (pairs)
    const uint32_t n_act = min((uint32_t)NW, (Lp + 64u * CPL - 1u) / (64u * CPL));      // column blocks this sequence reaches
    const bool wave_act = (uint32_t)w < n_act, has_left = w > 0, has_right = (uint32_t)w + 1u < n_act;
    // LDS byte addresses: this thread's entry of ring slot 0; the words a lane really writes when "lane 0" / "lane 63" publishes
    const uint32_t rb = (uint32_t)(uintptr_t)(lds_p)S.ring + ct * (uint32_t)(RW * 4);
    const uint32_t junk = (uint32_t)(uintptr_t)(lds_p)&Y.junk[2 * lane];
    const uint32_t a_done = lane == 0 ? (uint32_t)(uintptr_t)(lds_p)&Y.done[w][t] : junk;
    const uint32_t a_mcnt = lane == 63 ? (uint32_t)(uintptr_t)(lds_p)&Y.mcnt[w][t] : junk;
    const uint32_t a_mail = lane == 63 ? (uint32_t)(uintptr_t)(lds_p)&Y.mail[w][t][0][0] : junk;       // + 8 * entry
    const sk_ptr p_rdone = (sk_ptr)&Y.done[has_right ? w + 1 : w][t];
    const sk_ptr p_lcnt = (sk_ptr)&Y.mcnt[has_left ? w - 1 : 0][t], p_lmail = (sk_ptr)&Y.mail[has_left ? w - 1 : 0][t][0][0];
    uint32_t *const Hrec = (uint32_t *)S.H;                            // the record, a dword per column pair
    const uint32_t reach = A.ring_reach, slots = A.ring_slots;

    if (tid < NW * 4) { ((int32_t *)Y.done)[tid] = (tid & 3) < T ? (tid & 3) + 1 - T : MT_BIG; ((uint32_t *)Y.mcnt)[tid] = 0; }
    if (tid == 0) Y.abort = 0;
    const sk_ptr p_abort = (sk_ptr)&Y.abort;
    uint32_t dead = 0;                           // this wavefront has given up (or seen somebody who has)
    // which = 1: predecessor rows, 2: left mailbox, 3: right neighbour's back-pressure
    auto give_up = [&](const uint32_t which, const uint32_t row, const uint32_t seen, const uint32_t want) __attribute__((always_inline)) {
        if (lane == 0 && atomicExch((uint32_t *)&Y.abort, 1u) == 0u) {
            A.counters[8] = which | ((unsigned long long)t << 8) | ((unsigned long long)w << 16) | ((unsigned long long)row << 32);
            A.counters[9] = seen | ((unsigned long long)want << 32);
            A.counters[10] = n | ((unsigned long long)Lp << 32);
        }
        dead = 1u;
    };
    // the plan through the scalar cache (see dp_rows_v3)
    uint64_t ppa = uni64((uint64_t)S.plan), ppb = uni64((uint64_t)S.planb), ppc = uni64((uint64_t)S.planc), ppm = uni64((uint64_t)S.planm);
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+s"(ppa), "+s"(ppb), "+s"(ppc), "+s"(ppm) : : "memory");
    const cplan_t cpb = (cplan_t)ppb, cpc = (cplan_t)ppc;
    (void)ppa;
    const cplanm_t cpm = (cplanm_t)ppm;
    __syncthreads();                             // the counters are in place before anybody looks at them

    // ---- LDS reads of a row, issued back to back by hand and waited for ONCE ----
    // A row used to pay four or five LDS round trips one after the other (counters, ring entries, mailbox, back-pressure: ~150
    // cycles each for a wavefront that shares its SIMD with three others; measured: ~1400 cycles per row for ~110 instructions).
    // Now everything a row may need is requested in one burst at its top, IN THIS ORDER (LDS operations of a wavefront complete
    // in issue order): the done counters before the ring entries (entries at least as new as the counters that vouch for them), the
    // left mailbox's counter before its entry.  Inline asm, because the order and the single wait are the point; every destination is
    // tied to the wait below ("+v"), so nothing can be read before it has arrived.
    constexpr bool Q128 = RW % 4 == 0;           // an entry is read as 16-byte pieces (thread stride a multiple of 16) or as 8-byte pieces
    constexpr int NCH = Q128 ? RW / 4 : RW / 2;
    struct slot_regs { u32x4 q[Q128 ? NCH : 1]; u32x2 d[Q128 ? 1 : NCH]; };
    auto issue_slot = [&](const uint32_t off, slot_regs &r) __attribute__((always_inline)) {
        const uint32_t va = rb + off;
        if constexpr (Q128) {
            asm volatile("ds_read_b128 %0, %1" : "=&v"(r.q[0]) : "v"(va) : "memory");
            if constexpr (NCH > 1) asm volatile("ds_read_b128 %0, %1 offset:16" : "=&v"(r.q[1]) : "v"(va) : "memory");
        } else {
            asm volatile("ds_read_b64 %0, %1" : "=&v"(r.d[0]) : "v"(va) : "memory");
            asm volatile("ds_read_b64 %0, %1 offset:8" : "=&v"(r.d[1]) : "v"(va) : "memory");
            asm volatile("ds_read_b64 %0, %1 offset:16" : "=&v"(r.d[2]) : "v"(va) : "memory");
            if constexpr (NCH > 3) {
                asm volatile("ds_read_b64 %0, %1 offset:24" : "=&v"(r.d[3]) : "v"(va) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:32" : "=&v"(r.d[4]) : "v"(va) : "memory");
            }
        }
    };
    static_assert(NCH <= 5 && (Q128 ? NCH <= 2 : (NCH == 3 || NCH == 5)), "pieces of a ring entry");
    auto land_slot = [&](slot_regs &r) __attribute__((always_inline)) {      // (after an s_waitcnt) the values exist from here on
        if constexpr (Q128) {
            if constexpr (NCH > 1) asm volatile("" : "+v"(r.q[0]), "+v"(r.q[1]));
            else asm volatile("" : "+v"(r.q[0]));
        } else {
            if constexpr (NCH > 3) asm volatile("" : "+v"(r.d[0]), "+v"(r.d[1]), "+v"(r.d[2]), "+v"(r.d[3]), "+v"(r.d[4]));
            else asm volatile("" : "+v"(r.d[0]), "+v"(r.d[1]), "+v"(r.d[2]));
        }
    };
    auto word = [&](const slot_regs &r, const int i) __attribute__((always_inline)) -> uint32_t {
        if constexpr (Q128) return r.q[i / 4][i % 4];
        else return r.d[i / 2][i % 2];
    };
    auto ld_slot = [&](const uint32_t off, uint32_t (&r)[RW]) __attribute__((always_inline)) {      // (slow path) one entry, waited for
        slot_regs x;
        issue_slot(off, x);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        land_slot(x);
#pragma unroll
        for (int i = 0; i < RW; ++i) r[i] = word(x, i);
    };
    auto st_slot = [&](const uint32_t off, const uint32_t (&r)[RW]) __attribute__((always_inline)) {
        const lds_p p = (lds_p)(uintptr_t)(rb + off);
        if constexpr (RW % 4 == 0) {
#pragma unroll
            for (int q = 0; q < RW / 4; ++q) { u32x4 a; a.x = r[4 * q]; a.y = r[4 * q + 1]; a.z = r[4 * q + 2]; a.w = r[4 * q + 3]; ((__attribute__((address_space(3))) u32x4 *)p)[q] = a; }
        } else {
#pragma unroll
            for (int q = 0; q < RW / 2; ++q) { u32x2 a; a.x = r[2 * q]; a.y = r[2 * q + 1]; ((__attribute__((address_space(3))) u32x2 *)p)[q] = a; }
        }
    };

    uint32_t *hrow = Hrec + (((uint64_t)(uint32_t)(t + 1) * Lp) >> 1);           // this team's row of the record (wave-uniform), + c0 / 2 per thread
    const uint64_t hstep = ((uint64_t)T * Lp) >> 1;
    uint32_t mslot = 0;                                                           // mailbox entry of the current row: (row / T) % MT_MD
    // wave-uniform switches as bits of ONE scalar (tested with s_bitcmp; as booleans captured by the row lambda they came back as
    // v_cndmask / v_cmp pairs at every use)
    const uint32_t FL = (uint32_t)__builtin_amdgcn_readfirstlane((int)((has_left ? 1u : 0u) | (has_right ? 2u : 0u) | (plain ? 4u : 0u)));
    const uint32_t va_done = (uint32_t)(uintptr_t)(lds_p)&Y.done[w][0], va_rdone = (uint32_t)(uintptr_t)(lds_p)&Y.done[has_right ? w + 1 : w][t];
    const uint32_t va_lcnt = (uint32_t)(uintptr_t)(lds_p)&Y.mcnt[has_left ? w - 1 : 0][t], va_lmail = (uint32_t)(uintptr_t)(lds_p)&Y.mail[has_left ? w - 1 : 0][t][0][0];

    // one row.  pd: its plan record; nx: where the record of this team's next row is fetched to
    // The record of a row: 32 bytes, one s_load_dwordx8 (two sets of them live across the loop: with 64-byte records the kernel ran out
    // of scalar registers and, in some builds, reloaded spilled ones -- v_readlane -- 300 times per trip of the loop; a two-team pack then
    // ran no faster than a one-team pack).  x klo, y khi (score table halves), z the row that must be final, w = own slot | flags << 8 |
    // far in-edges << 16 | letter << 24, [4] / [5] the slots of in-edges 1-4 / 5-8 in a byte each, [6] in-degree, [7] edge index of
    // the ninth in-edge.  A slot is given in units of 2 KB (MT_UNIT): slot bytes are 4, 6, 8 or 10 KB.
    auto step = [&](const uint32_t row, const u32x8 pd, u32x8 &nx) __attribute__((always_inline)) {
        const uint32_t klo = pd[0], khi = pd[1], self_off = (pd[3] & 0xFFu) << MT_UNIT, ctl = pd[3] >> 8;
        const int32_t need = (int32_t)pd[2];
        const uint32_t o0 = (pd[4] & 0xFFu) << MT_UNIT, o1 = ((pd[4] >> 8) & 0xFFu) << MT_UNIT, o2 = ((pd[4] >> 16) & 0xFFu) << MT_UNIT, o3 = (pd[4] >> 24) << MT_UNIT;
        // ---- one burst: counters, the left mailbox, the four ring entries (missing predecessors repeat one that is there) ----
        u32x4 dn;
        uint32_t rdv, lcv;
        u32x2 lmv;
        slot_regs e0, e1, e2, e3;
        {
            const uint32_t va_lm = va_lmail + 8u * mslot;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b64 %3, %7"
                         : "=&v"(dn), "=&v"(rdv), "=&v"(lcv), "=&v"(lmv) : "v"(va_done), "v"(va_rdone), "v"(va_lcnt), "v"(va_lm) : "memory");
        }
        issue_slot(o0, e0); issue_slot(o1, e1); issue_slot(o2, e2); issue_slot(o3, e3);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dn), "+v"(rdv), "+v"(lcv), "+v"(lmv) : : "memory");
        land_slot(e0); land_slot(e1); land_slot(e2); land_slot(e3);
        if constexpr (T > 1) {
            // the rows this one reads are final in this column block: min over the teams of their last finished row + T > need
            int32_t m = __builtin_amdgcn_readfirstlane(min(min((int32_t)dn.x, (int32_t)dn.y), min((int32_t)dn.z, (int32_t)dn.w)));
            if (m + T <= need) {
                // not yet: poll the counters alone (the ring entries are 4 to 20 LDS cycles each: a dozen waiting wavefronts re-reading
                // them would take the LDS from the ones that work), then read the entries again
                uint32_t spins = 0;
                do {
                    if (++spins > MT_SPIN_LIMIT || sk_ld(p_abort)) { give_up(1, row, (uint32_t)m, (uint32_t)need); break; }
                    if (MT_POLL_SLEEP) __builtin_amdgcn_s_sleep(1);
                    if (MT_POLL_FULL && T <= MT_POLL_FULL) {
                        asm volatile("ds_read_b128 %0, %1" : "=&v"(dn) : "v"(va_done) : "memory");
                        issue_slot(o0, e0); issue_slot(o1, e1); issue_slot(o2, e2); issue_slot(o3, e3);
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dn) : : "memory");
                        land_slot(e0); land_slot(e1); land_slot(e2); land_slot(e3);
                    } else asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(dn) : "v"(va_done) : "memory");
                    m = __builtin_amdgcn_readfirstlane(min(min((int32_t)dn.x, (int32_t)dn.y), min((int32_t)dn.z, (int32_t)dn.w)));
                } while (m + T <= need);
                if (!(MT_POLL_FULL && T <= MT_POLL_FULL)) {
                    issue_slot(o0, e0); issue_slot(o1, e1); issue_slot(o2, e2); issue_slot(o3, e3);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    land_slot(e0); land_slot(e1); land_slot(e2); land_slot(e3);
                }
            }
        }
        s16x2 HM[NP], FM[NP];                    // maxima over the predecessors: H[p][j-1] and max(H[p][j] + g - e, F[p][j])
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            HM[u] = pk_max(pk_max(as_pk(word(e0, u)), as_pk(word(e1, u))), pk_max(as_pk(word(e2, u)), as_pk(word(e3, u))));
            FM[u] = pk_max(pk_max(as_pk(word(e0, NP + u)), as_pk(word(e1, NP + u))), pk_max(as_pk(word(e2, NP + u)), as_pk(word(e3, NP + u))));
        }
        uint32_t r0[RW];
        if (ctl & MT_MORE4) {
            issue_slot((pd[5] & 0xFFu) << MT_UNIT, e0); issue_slot(((pd[5] >> 8) & 0xFFu) << MT_UNIT, e1); issue_slot(((pd[5] >> 16) & 0xFFu) << MT_UNIT, e2); issue_slot((pd[5] >> 24) << MT_UNIT, e3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            land_slot(e0); land_slot(e1); land_slot(e2); land_slot(e3);
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                HM[u] = pk_max(pk_max(HM[u], as_pk(word(e0, u))), pk_max(pk_max(as_pk(word(e1, u)), as_pk(word(e2, u))), as_pk(word(e3, u))));
                FM[u] = pk_max(pk_max(FM[u], as_pk(word(e0, NP + u))), pk_max(pk_max(as_pk(word(e1, NP + u)), as_pk(word(e2, NP + u))), as_pk(word(e3, NP + u))));
            }
        }
        {   // the record of this team's next row: requested behind the row's LDS wait (lgkmcnt counts scalar loads too: ahead of it,
            // that wait would be for the scalar cache), it has the rest of the row to arrive
            uint64_t pp = (uint64_t)(cpm + (row + T - 1));
            uint32_t dep = as_u(HM[0]);
            asm volatile("" : "+s"(pp), "+v"(dep));
            HM[0] = as_pk(dep);
            nx = *(cplanm_t)pp;
        }
        if (ctl & MT_SLOW) {
            // in-edges beyond the ring (their record words from HBM, decoded into the ring's terms) and in-edges after the eighth
            const uint32_t n_all = pd[6], farmask = (ctl >> 8) & 0xFFu;
            if (ctl & MT_NOBASE) {               // what the four entries held is nobody's predecessor: H[p][j-1] = 0, max(H + g - e, F) = g - e
#pragma unroll
                for (int u = 0; u < NP; ++u) { HM[u] = pk_splat(0); FM[u] = pk_splat(POA_G - POA_E); }
            }
            // (lane 0 of a column block w > 0 reads the word LEFT of the block, hq[-1], from block w - 1's record: that store is ordered before
            // this load only through the mailbox waits of earlier rows, which holds when ring_reach + 1 >= teams + ring_slack -- plan_class
            // sizes the ring so, or takes the barrier form)
            auto far_fetch = [&](uint32_t prow) __attribute__((always_inline)) {
                uint32_t x[NP];
                const uint32_t cc = act ? c0 : 0u;
                const uint32_t *hq = (const uint32_t *)(S.H + (uint64_t)prow * Lp + cc);
                const bool need_left = lane == 0 && w > 0 && act;
                const uint32_t hl16 = ((const uint16_t *)hq)[need_left ? -1 : 0];
#pragma unroll
                for (int u = 0; u < NP; ++u) { const uint32_t v = hq[u]; x[u] = act ? v : 0x80008000u; }      // H = 0, H - F = 2: what a column beyond the row decodes to
                const uint32_t wl = need_left ? (hl16 & 0x3FFFu) << 16 : 0u;
                drain_vector_loads();
                uint32_t hp[NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    hp[u] = x[u] & 0x3FFF3FFFu;
                    u16x2 wu;
                    __builtin_memcpy(&wu, &x[u], 4);
                    const u16x2 d = __builtin_elementwise_min(wu >> (u16x2){14, 14}, (u16x2){2, 2});
                    s16x2 ds;
                    __builtin_memcpy(&ds, &d, 4);
                    FM[u] = pk_max(FM[u], as_pk(hp[u]) - ds);                                        // B = H - min(H - F, 2)
                }
                const uint32_t left = (uint32_t)wave_shr1((int32_t)hp[NP - 1], (int32_t)wl);
#pragma unroll
                for (int u = 0; u < NP; ++u) HM[u] = pk_max(HM[u], pk_left(hp[u], u == 0 ? left : hp[u - 1]));      // A = H shifted by a column
            };
            for (uint32_t k = 0; k < 8 && k < n_all; ++k) {
                if (!((farmask >> k) & 1u)) continue;
                uint32_t prow = k >= 4 ? cpc[row - 1][k - 4] : cpb[row - 1][k];
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(prow));
                far_fetch(prow);
            }
            if (n_all > 8) {
                uint32_t e = pd[7];
                for (uint32_t k = 8; k < n_all; ++k) {
                    const uint2 ed = S.edges[e]; e = ed.y;
                    const uint32_t prow = (uint32_t)__builtin_amdgcn_readfirstlane(S.rank[ed.x]) + 1;
                    drain_vector_loads();
                    if (row - prow <= reach) {
                        ld_slot((prow % slots) * SLOTB, r0);
#pragma unroll
                        for (int u = 0; u < NP; ++u) { HM[u] = pk_max(HM[u], as_pk(r0[u])); FM[u] = pk_max(FM[u], as_pk(r0[NP + u])); }
                    } else far_fetch(prow);
                }
            }
        }
        // ---- Hn = max(diagonal, F, 0); u = Hn + g - (j+1)e; in-thread exclusive prefix max of u (pair by pair) ----
        s16x2 HNp[NP], EX[NP], SC[NP], FN[NP];
        s16x2 RUN = pk_splat(-32768);
        if (FL & 4u) {
#pragma unroll
            for (int u = 0; u < NP; ++u) SC[u] = as_pk(__builtin_amdgcn_perm(khi, klo, SEL[u]));
        } else {
            const uint32_t letter = ctl >> 16;
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int32_t s0 = ((sw[(2 * u) >> 2] >> (8 * ((2 * u) & 3))) & 0xFFu) == letter ? POA_M : POA_N;
                const int32_t s1 = ((sw[(2 * u + 1) >> 2] >> (8 * ((2 * u + 1) & 3))) & 0xFFu) == letter ? POA_M : POA_N;
                SC[u] = as_pk(pack16(s0, s1));
            }
        }
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            FN[u] = FM[u] + pk_splat(POA_E);
            HNp[u] = pk_max(pk_max(HM[u] + SC[u], FN[u]), pk_splat(0));
            const s16x2 v = pk_max(RUN, HNp[u] + UC[u]);                                   // (max(run, u_a), max(run, u_b))
            EX[u] = as_pk(__builtin_amdgcn_alignbit(as_u(v), as_u(RUN), 16));              // (run, max(run, u_a)): both halves of RUN are equal
            RUN = pk_max(v, __builtin_shufflevector(v, v, 1, 0));                          // max(run, u_a, u_b) in both halves
        }
        const uint32_t wincl = wave_scan_max_dup(as_u(RUN));
        const uint32_t texcl = (uint32_t)wave_shr1((int32_t)wincl, (int32_t)0x80008000u);
        // ---- the prefix over the column blocks to the left (same team), ours to the right ----
        uint32_t sbase = as_u(pk_splat(POA_G - POA_E));        // u_0
        uint32_t hl = 0;
        if (FL & 1u) {
            // what the burst at the top of the row found in the left mailbox is this row's if the left wavefront had already published
            // it (a wavefront that trails its left neighbour by half a row or more: no wait at all); otherwise poll
            uint32_t cl = (uint32_t)__builtin_amdgcn_readfirstlane((int)lcv), leT = lmv.x, leH = lmv.y;
            uint32_t spins = 0;
            while ((int32_t)(cl - row) < 0) {
                if (++spins > MT_SPIN_LIMIT || sk_ld(p_abort)) { give_up(2, row, cl, row); break; }
                __builtin_amdgcn_s_sleep(1);
                cl = sk_ld(p_lcnt); leT = sk_ld(p_lmail + 2 * mslot); leH = sk_ld(p_lmail + 2 * mslot + 1);
                cl = (uint32_t)__builtin_amdgcn_readfirstlane((int)cl);
            }
            hl = (uint32_t)__builtin_amdgcn_readfirstlane((int)leH);
            sbase = (uint32_t)max((int32_t)sbase, __builtin_amdgcn_readfirstlane((int)leT));
        }
        const s16x2 BASE = pk_max(as_pk(texcl), as_pk(sbase));
        uint32_t W[NP];
        s16x2 HN[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const s16x2 EV = pk_max(BASE, EX[u]) + JE[u];
            HN[u] = pk_max(HNp[u], EV);
        }
        // What other wavefronts wait for goes out FIRST, in the order of who waits longest: the mailbox entry (the column block to the
        // right, same row), the ring entry + counter (rows that read this one); the record word, the running maximum and the record
        // store are nobody's critical path (a lone wavefront issues in order: every instruction ahead of a publication delays it).
        if (FL & 2u) {
            // the entry this row's mail goes to still holds row - MT_MD * T: its reader has finished that row (the counter read at the
            // top of the row says so almost always: it only grows)
            int32_t cr = __builtin_amdgcn_readfirstlane((int32_t)rdv);
            uint32_t spins = 0;
            while (cr + (int32_t)(MT_MD * T) < (int32_t)row) {
                if (++spins > MT_SPIN_LIMIT || sk_ld(p_abort)) { give_up(3, row, (uint32_t)cr, row); break; }
                __builtin_amdgcn_s_sleep(1);
                cr = __builtin_amdgcn_readfirstlane((int32_t)sk_ld(p_rdone));
            }
            const uint32_t tc = (uint32_t)max((int32_t)sbase, __builtin_amdgcn_readlane((int32_t)wincl, 63));
            // lane 63 writes {prefix, H of its last column (high half)} and then the counter; the other lanes write junk words
            const lds_p mp = (lds_p)(uintptr_t)(a_mail + 8u * mslot);
            { u32x2 mv; mv.x = tc; mv.y = as_u(HN[NP - 1]); *(volatile __attribute__((address_space(3))) u32x2 *)mp = mv; }
            *(volatile lds_p)(uintptr_t)a_mcnt = row;
        }
        {
            uint32_t R[RW];
            const uint32_t left = (uint32_t)wave_shr1((int32_t)as_u(HN[NP - 1]), (int32_t)hl);       // lane 0: the H the left column block published
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                R[u] = as_u(pk_left(as_u(HN[u]), u == 0 ? left : as_u(HN[u - 1])));
                R[NP + u] = as_u(pk_max(HN[u] - pk_splat(2), FN[u]));
            }
            st_slot(self_off, R);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NP; ++u)         // H (14 bits) | min(H - F, 3) << 14: the record word (traceback, rows beyond the ring)
                asm("v_lshl_or_b32 %0, %1, 14, %2" : "=v"(W[u]) : "v"(as_u(pk_min(HN[u] - FN[u], pk_splat(3)))), "v"(as_u(HN[u])));
            if (act) {                           // the record BEFORE the counter: a row that reads it back from HBM (beyond the ring) issues its load after it has seen the counter
                uint32_t *hq = hrow + (c0 >> 1);
                if (NP % 2 == 0) {
#pragma unroll
                    for (int u = 0; u < NP / 2; ++u) ((uint2 *)hq)[u] = make_uint2(W[2 * u], W[2 * u + 1]);
                } else {
#pragma unroll
                    for (int u = 0; u < NP; ++u) hq[u] = W[u];
                }
            }
            if constexpr (T > 1 || NW > 1) *(volatile lds_p)(uintptr_t)a_done = row;       // after the ring entry (LDS operations of a wavefront complete in order)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NP; ++u) MXA = pk_max(MXA, HN[u]);
        }
        hrow += hstep;
        mslot = (mslot + 1u) & (uint32_t)(MT_MD - 1);
    };

    if (wave_act) {
        // two rows per trip, two register sets for the plan record (no copies)
        u32x8 pa = cpm[t], pb = pa;
        uint32_t row = (uint32_t)t + 1u;
        while (row <= n && !__builtin_amdgcn_readfirstlane((int)dead)) {
            step(row, pa, pb);
            row += T;
            if (row > n || __builtin_amdgcn_readfirstlane((int)dead)) break;
            step(row, pb, pa);
            row += T;
        }
        *(volatile lds_p)(uintptr_t)a_done = (uint32_t)MT_BIG;          // every row of this wavefront is final
    }
    // block-wide best score and the row threads (places in a row) whose columns reach it (the first row that reaches it comes from
    // a rescan of those columns in the record, kernel body).  A place is held by T threads, one per team: flags in LDS, one entry each.
    const int32_t lbest = act ? max((int32_t)(int16_t)(as_u(MXA) & 0xFFFFu), (int32_t)as_u(MXA) >> 16) : 0;
    const int32_t wb = wave_last(wave_scan_max(lbest, 0));
    if (lane == 0) X.best[wv] = wb;
    if (tid == 0) { X.brow = 0xFFFFFFFFu; X.multi = 1; X.ntl = 0; }
    __syncthreads();                             // (also: every row loop has ended, the ring is free)
    best = 0;
#pragma unroll
    for (int q = 0; q < NW * T; ++q) best = max(best, X.best[q]);
    uint8_t *flag = (uint8_t *)S.ring;
    if (tid < NTC) flag[tid] = 0;
    __syncthreads();
    if (best > 0 && lbest == best) flag[ct] = 1;
    __syncthreads();
    if (tid < NTC && flag[tid]) {
        const uint32_t slot = atomicAdd(&X.ntl, 1u);
        if (slot < 16) X.tl[slot] = (uint32_t)tid;
    }
    __syncthreads();
    best_row = 0;
    multi = best > 0;
    if (sk_ld(p_abort)) { S.err = POA_ERR_SYNC; best = 0; multi = false; }      // (uniform: the barriers above have passed)
}

// ---- an EXACT band for near-chain graphs (PK == 8; POA #2 / #3 of `rattle correct`, correct.cpp:427-436,520-532) ------------------
// The corrected reads of a pack and the pack consensi of a cluster are near-identical: their graph is (almost) a chain and the
// alignment runs down one diagonal.  A local alignment path scores at most 5 per diagonal move and nothing for any other move, and
// a path through cell (row i, column j) of an n-row graph in topological order has at most
//     M(i, j) = min(c_i, j) + min(C - c_i, L - j)
// diagonal moves, c_i = the MSA column of row i (rows in block order; the rows of an aligned group share a column), C = columns of the
// graph: at most one diagonal move per sequence column, and at most one per graph column on either side of the cell, because every edge
// leads to a strictly later column (checked for every in-edge while the rows' records are built: a graph that breaks it gets no band).
// (Rows instead of columns would do for a chain, but every substitution a pack member adds is a row of its own in an existing column:
// the POA #3 of a 550-pack cluster reaches 1600 rows for 1300 columns.)  So with tau = L - t every path that touches a cell with M < tau scores at most 5 (tau - 1):
// if the best score S' found with those cells taken as H = 0, E = F = -inf is >= 5 tau - 4, then (i) every path of score >= S' lies
// inside {M >= tau}, so S' is the true best score and the cells that reach it are the same; (ii) the traceback is the same: a test
// `H[i][j] == H[p][j-1] + s` (or on F / E) that holds in the full matrices puts the tested cell on a path of total score S, which lies
// inside and is therefore computed exactly; one that fails in the full matrices fails a fortiori on a lower bound (DESIGN.md §4,
// tests/test_band_lemma.py for sequences and DAGs).  {M >= tau} is the parallelogram  max(1, c_i - s) <= j <= min(L, c_i + t),
// s = C - L + t, of width C - L + 2 t + 1: it fits ONE wavefront -- 64 lanes x CPLB columns, no exchange between wavefronts, no
// barrier, a record of 64 CPLB cells per row instead of L -- whenever C - L + 2 t + 1 <= 63 CPLB (CPLB = 2, 4 or 8).  The certificate is checked after the
// rows; when it fails the alignment is run again wider (t from the score found: then it must hold) or over the full rows.
// Layout: lane l of row i owns the column group G0(i) + l, G0(i) = max(0, c_i - s - 1) / CPLB (relative to the row's window; the row's
// record carries it).  The ring
// (LDS) holds ready-made predecessor terms per lane as in dp_rows_mt (A = H shifted by a column, B = max(H + g - e, F)); a reader
// finds the terms of ITS columns in row p at lane l + G0(i) - G0(p): an address offset the plan record carries ready-made.  Behind the
// 64 entries of a slot lie BAND_PAD neutral ones (what lies right of row p's window: H = 0), the first of them with the H of the
// window's last column as its diagonal source.  Everything a lane keeps per column (score selectors, j e, g - (j+1) e) moves with
// the window: an add and one LDS read every CPLB rows.
#define BAND_PAD 16
#define BD_MORE4 1u
#define BD_SLOW 2u
#define BD_NOBASE 4u
#define BD_CHAIN 8u                                // the row's one in-edge comes from the row before it, with the window where it was: that row's terms are still in registers
__host__ __device__ constexpr uint32_t band_entry_bytes(int CPLB) { return 4u * (uint32_t)CPLB; }
__host__ __device__ constexpr uint32_t band_slot_bytes(int CPLB) { return (64u + BAND_PAD) * band_entry_bytes(CPLB); }
// first column group of the window of a row in column c (bsh = log2 CPLB)
__device__ __forceinline__ uint32_t band_g0(uint32_t c, uint32_t bs, uint32_t bsh) { return c > bs + 1u ? (c - bs - 1u) >> bsh : 0u; }
// LDS the band needs behind the ring base: slots, the selector table (a dword per column pair for Lp + 66 CPLB columns), junk words
__host__ __device__ constexpr uint32_t band_lds_bytes(uint32_t slots, int CPLB, uint32_t Lp) {
    return slots * band_slot_bytes(CPLB) + (Lp / (uint32_t)CPLB + 66u) * (uint32_t)(CPLB / 2) * 4u + 64u * 4u;
}

// STRIP = true: the same rows over a VERTICAL window -- the columns strip * 64 CPLB + 1 .. (strip + 1) * 64 CPLB of every row -- with real
// values at its left edge instead of H = 0: strip after strip this is the full matrix, no certificate needed (what an alignment runs
// whose band does not fit or whose certificate fails).  A cell depends on cells in its own or a smaller column only, so the strips can
// be done one after the other; strip k leaves two numbers per row for strip k + 1 (a dword in lh[], double-buffered by strip parity, as
// in dp_rows_longr): the H of its last column (the diagonal source of the next strip's first column) and the prefix maximum of u over
// every column up to there (the horizontal gap).  Windows do not move: no lane shift, no pads.  Record: strip-major, (n + 1) rows of 64
// CPLB cells per strip.
template <int CPLB, bool STRIP = false>
__device__ __forceinline__ void dp_rows_band(poa_ws &S, dp_xchg &X, uint32_t n, uint32_t brs, uint32_t sel_off, uint32_t junk_off, int32_t &best, uint32_t strip = 0,
                                             s16x2 mxa_in = (s16x2){0, 0}, s16x2 *mxa_out = nullptr) {
    constexpr int NP = CPLB / 2;
    constexpr uint32_t EB = band_entry_bytes(CPLB);
    static_assert(CPLB == 2 || CPLB == 4 || CPLB == 8, "a ring entry is one 8- or 16-byte LDS access, or two of 16");
    const int lane = threadIdx.x & 63;
    const uint32_t ring0 = (uint32_t)(uintptr_t)(lds_p)S.ring;
    const uint32_t rb = ring0 + (uint32_t)lane * EB;                          // this lane's entry of slot 0
    const uint32_t va_sel = ring0 + sel_off + (uint32_t)lane * (NP * 4u);     // + group * NP * 4: the selectors of the lane's columns
    const uint32_t a_pad = lane == 63 ? ring0 + 64u * EB : ring0 + junk_off + 4u * (uint32_t)lane;      // where lane 63's "H of the last column" goes (+ slot)
    // the record: 64 * NP dwords per row.  Through pointers that carry the GLOBAL address space in their type: the kernel keeps its
    // workspace pointers in scratch here, what comes back from there is a generic pointer to the compiler, and a flat_store counts on
    // lgkmcnt as well -- the row's LDS wait then waited for the previous row's record store, a memory round trip per row (seen in the
    // disassembly of the first build: 665 cycles per row)
    typedef __attribute__((address_space(1))) uint32_t *gptr_t;
    typedef __attribute__((address_space(1))) const uint16_t *gptr16_t;
    const gptr_t Hrec = (gptr_t)(uintptr_t)S.H;
    uint64_t ppm = uni64((uint64_t)S.planm), ppl = uni64((uint64_t)S.lh);
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+s"(ppm), "+s"(ppl) : : "memory");
    const cplanm_t cpm = (cplanm_t)ppm;
    typedef const __attribute__((address_space(4))) uint32_t *cdw_t;
    const cdw_t cin = (cdw_t)ppl + (strip & 1u);                              // STRIP: cin[4 * row] = what the strip to the left left for this row
    typedef __attribute__((address_space(1))) uint32_t *gdw_t;
    const gdw_t cout = (gdw_t)(uintptr_t)S.lh + ((strip + 1u) & 1u);

    struct ent8 { u32x4 a, b; __device__ __forceinline__ uint32_t operator[](int i) const { return i < 4 ? a[i] : b[i - 4]; } };
    typedef typename std::conditional<NP == 1, u32x2, typename std::conditional<NP == 2, u32x4, ent8>::type>::type ent_t;
    auto issue = [&](const uint32_t off, ent_t &r) __attribute__((always_inline)) {
        const uint32_t va = rb + off;
        if constexpr (NP == 1) asm volatile("ds_read_b64 %0, %1" : "=&v"(r) : "v"(va) : "memory");
        else if constexpr (NP == 2) asm volatile("ds_read_b128 %0, %1" : "=&v"(r) : "v"(va) : "memory");
        else asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(r.a), "=&v"(r.b) : "v"(va) : "memory");
    };
    auto land4 = [&](ent_t &e0, ent_t &e1, ent_t &e2, ent_t &e3) __attribute__((always_inline)) {      // behind an s_waitcnt: the entries exist from here on
        if constexpr (NP == 4) asm volatile("" : "+v"(e0.a), "+v"(e0.b), "+v"(e1.a), "+v"(e1.b), "+v"(e2.a), "+v"(e2.b), "+v"(e3.a), "+v"(e3.b));
        else asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
    };
    uint32_t SEL[NP];
    s16x2 JE[NP], UC[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int j0 = ((int)(STRIP ? strip * 64u : 0u) + lane) * CPLB + 2 * u + 1, j1 = j0 + 1;
        JE[u] = (s16x2){(short)(j0 * POA_E), (short)(j1 * POA_E)};
        UC[u] = (s16x2){(short)(POA_G - (j0 + 1) * POA_E), (short)(POA_G - (j1 + 1) * POA_E)};
        SEL[u] = ((const lds_p)(uintptr_t)(va_sel + (STRIP ? strip * 64u * (uint32_t)(NP * 4) : 0u)))[u];
    }
    s16x2 MXA = mxa_in;
    uint32_t g0 = STRIP ? strip * 64u : 0u;                                   // first column group of the current window (wave-uniform)
    uint32_t sbase = as_u(pk_splat(POA_G - POA_E));                           // u of the column left of the window (H = 0 there)
    gptr_t hrow = Hrec + ((uint64_t)(STRIP ? strip * (n + 1u) : 0u) + 1u) * (64u * NP) + (uint32_t)lane * NP;      // row 1 (of this strip)
    s16x2 PA[NP], PB[NP];                                                     // the previous row's terms (what it wrote to the ring)
#pragma unroll
    for (int u = 0; u < NP; ++u) { PA[u] = pk_splat(0); PB[u] = pk_splat(POA_G - POA_E); }
    uint32_t cw_next = 0;                                                     // STRIP: the dword of the row after this one (a scalar load a row ahead)
    if constexpr (STRIP) { if (strip) cw_next = cin[4]; }

    auto step = [&](const uint32_t row, const u32x8 pd, u32x8 &nx) __attribute__((always_inline)) {
        const uint32_t klo = pd[0], khi = pd[1];
        const uint32_t o0 = pd[2] & 0xFFFFu, o1 = pd[2] >> 16, o2 = pd[3] & 0xFFFFu, o3 = pd[3] >> 16;
        const uint32_t self_off = pd[4] & 0xFFFFu, ctl = (pd[5] >> 16) & 0xFFu;      // (pd[4] >> 16: the window's first column group, for whoever reads this row later)
        const uint32_t dl = STRIP ? 0u : pd[5] & 0xFFu;
        // A chain row (one in-edge, from the row before: nearly every row of POA #2 / #3) takes its predecessor's terms from REGISTERS -- the
        // ring entry the previous row has just written would be an LDS write and read back to back on the critical path of every row
        // (measured: ~560 cycles per row for ~70 instructions).  (A row whose window has moved on since -- one in CPLB -- reads the ring: taking
        // the terms from the neighbour lane by DPP, with lane 63's from its own last H, did not get through this compiler.)
        const bool chain = (ctl & BD_CHAIN) != 0u;
        ent_t e0, e1, e2, e3;
        if (!chain) { issue(o0, e0); issue(o1, e1); issue(o2, e2); issue(o3, e3); }
        // STRIP: what the strip to the left says about this row: its last H (low half) and the prefix maximum of u up to there (high half)
        uint32_t hl_row = 0, sb_row = sbase;
        if constexpr (STRIP) {
            if (strip) {
                const uint32_t cw = cw_next;                                 // (it arrived before the previous row's LDS writes: see the wait there)
                hl_row = cw << 16;                                           // as the high half of a pair word: the column left of lane 0's first
                sb_row = (cw >> 16) | (cw & 0xFFFF0000u);
                uint64_t pc = (uint64_t)(cin + 4 * (uint64_t)(row + 1u));
                asm volatile("" : "+s"(pc));
                cw_next = *(cdw_t)pc;
            }
        }
        if (dl) {                                // the window moves on by dl column groups
            g0 += dl;
            const s16x2 dj = pk_splat((int)(dl * CPLB) * POA_E);
#pragma unroll
            for (int u = 0; u < NP; ++u) { JE[u] = JE[u] + dj; UC[u] = UC[u] - dj; }
            const uint32_t va = va_sel + g0 * (uint32_t)(NP * 4);
            if constexpr (NP == 1) asm volatile("ds_read_b32 %0, %1" : "=&v"(SEL[0]) : "v"(va) : "memory");
            else if constexpr (NP == 2) { u32x2 t2; asm volatile("ds_read_b64 %0, %1" : "=&v"(t2) : "v"(va) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t2) : : "memory"); SEL[0] = t2.x; SEL[1] = t2.y; }
            else { u32x4 t4; asm volatile("ds_read_b128 %0, %1" : "=&v"(t4) : "v"(va) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t4) : : "memory"); SEL[0] = t4.x; SEL[1] = t4.y; SEL[2] = t4.z; SEL[3] = t4.w; }
            const uint32_t ub = (uint32_t)((int32_t)(g0 * CPLB + 1u) * -POA_E + POA_G) & 0xFFFFu;      // 0 + g - (c + 1) e, c = g0 * CPLB
            sbase = ub | (ub << 16);
        }
        s16x2 HM[NP], FM[NP];
        if (chain) {
            // (no wait at all: this row's record arrived before the previous row's LDS writes went out, see below)
#pragma unroll
            for (int u = 0; u < NP; ++u) { HM[u] = PA[u]; FM[u] = PB[u]; }
        } else {
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the ring entries (and, when the window moved, the selectors)
            if constexpr (NP == 1) asm volatile("" : "+v"(SEL[0]));
            land4(e0, e1, e2, e3);
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                HM[u] = pk_max(pk_max(as_pk(e0[u]), as_pk(e1[u])), pk_max(as_pk(e2[u]), as_pk(e3[u])));
                FM[u] = pk_max(pk_max(as_pk(e0[NP + u]), as_pk(e1[NP + u])), pk_max(as_pk(e2[NP + u]), as_pk(e3[NP + u])));
            }
        }
        if (ctl & BD_MORE4) {
            issue(pd[6] & 0xFFFFu, e0); issue(pd[6] >> 16, e1); issue(pd[7] & 0xFFFFu, e2); issue(pd[7] >> 16, e3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            land4(e0, e1, e2, e3);
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                HM[u] = pk_max(pk_max(HM[u], as_pk(e0[u])), pk_max(pk_max(as_pk(e1[u]), as_pk(e2[u])), as_pk(e3[u])));
                FM[u] = pk_max(pk_max(FM[u], as_pk(e0[NP + u])), pk_max(pk_max(as_pk(e1[NP + u]), as_pk(e2[NP + u])), as_pk(e3[NP + u])));
            }
        }
        {   // the record of the next row, behind this row's LDS wait (lgkmcnt counts scalar loads too)
            uint64_t pp = (uint64_t)(cpm + row);
            uint32_t dep = as_u(HM[0]);
            asm volatile("" : "+s"(pp), "+v"(dep));
            HM[0] = as_pk(dep);
            nx = *(cplanm_t)pp;
        }
        if (ctl & BD_SLOW) {
            // in-edges whose terms are not in the ring (too far back, or the window has moved on by more than the pad) and in-edges
            // after the eighth: their record words from HBM, decoded into the ring's terms
            const uint32_t farmask = (pd[5] >> 8) & 0xFFu;
            uint32_t n_all = pd[5] >> 24;
            if (n_all == 255u) { n_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)rd_nin(S.plan[row - 1].x)); drain_vector_loads(); }      // (the byte saturates)
            if (ctl & BD_NOBASE) {
#pragma unroll
                for (int u = 0; u < NP; ++u) { HM[u] = pk_splat(0); FM[u] = pk_splat(POA_G - POA_E); }
            }
            auto far_fetch = [&](const uint32_t prow) __attribute__((always_inline)) {
                uint32_t gp = STRIP ? (strip ? cin[4 * (uint64_t)prow] : 0u) : cpm[prow - 1][4];      // (scalar load) row prow's record: its window's first group (STRIP: its dword from the left)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(gp));
                const uint32_t dlt = STRIP ? 0u : g0 - (gp >> 16);                // (wave-uniform) the window has moved on by dlt groups since row prow
                const uint32_t lp = (uint32_t)lane + dlt;
                const bool in = lp < 64u;
                const uint64_t prow_s = (uint64_t)prow + (STRIP ? (uint64_t)strip * (n + 1u) : 0u);
                const gptr_t hq = Hrec + (prow_s * 64u + (in ? lp : 0u)) * NP;
                const bool need_left = !STRIP && lane == 0 && dlt >= 1u && dlt <= 64u;
                uint32_t hl16 = ((gptr16_t)Hrec)[need_left ? (prow_s * 64u + dlt) * CPLB - 1u : 0u];
                if constexpr (STRIP) hl16 = gp & 0x3FFFu;                      // the H of row prow left of this strip
                uint32_t x[NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) { const uint32_t v = hq[u]; x[u] = in ? v : 0x80008000u; }      // outside row prow's window: H = 0, H - F >= 2
                const uint32_t wl = (need_left || STRIP) ? (hl16 & 0x3FFFu) << 16 : 0u;
                drain_vector_loads();
                uint32_t hp[NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    hp[u] = x[u] & 0x3FFF3FFFu;
                    u16x2 wu;
                    __builtin_memcpy(&wu, &x[u], 4);
                    const u16x2 d = __builtin_elementwise_min(wu >> (u16x2){14, 14}, (u16x2){2, 2});
                    s16x2 ds;
                    __builtin_memcpy(&ds, &d, 4);
                    FM[u] = pk_max(FM[u], as_pk(hp[u]) - ds);
                }
                const uint32_t left = (uint32_t)wave_shr1((int32_t)hp[NP - 1], (int32_t)wl);
#pragma unroll
                for (int u = 0; u < NP; ++u) HM[u] = pk_max(HM[u], pk_left(hp[u], u == 0 ? left : hp[u - 1]));
            };
            const cplan_t cpb = (cplan_t)uni64((uint64_t)S.planb), cpc = (cplan_t)uni64((uint64_t)S.planc);
            for (uint32_t k = 0; k < 8 && k < n_all; ++k) {
                if (!((farmask >> k) & 1u)) continue;
                uint32_t prow = k >= 4 ? cpc[row - 1][k - 4] : cpb[row - 1][k];
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(prow));
                far_fetch(prow);
            }
            if (n_all > 8) {
                uint32_t e = S.plan[row - 1].w;
                e = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
                drain_vector_loads();
                for (uint32_t k = 8; k < n_all; ++k) {
                    const uint2 ed = S.edges[e]; e = ed.y;
                    const uint32_t prow = (uint32_t)__builtin_amdgcn_readfirstlane(S.rank[ed.x]) + 1;
                    drain_vector_loads();
                    far_fetch(prow);
                }
            }
        }
        s16x2 HNp[NP], EX[NP], FN[NP];
        s16x2 RUN = pk_splat(-32768);
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const s16x2 SC = as_pk(__builtin_amdgcn_perm(khi, klo, SEL[u]));
            FN[u] = FM[u] + pk_splat(POA_E);
            HNp[u] = pk_max(pk_max(HM[u] + SC, FN[u]), pk_splat(0));
            const s16x2 v = pk_max(RUN, HNp[u] + UC[u]);
            EX[u] = as_pk(__builtin_amdgcn_alignbit(as_u(v), as_u(RUN), 16));
            RUN = pk_max(v, __builtin_shufflevector(v, v, 1, 0));
        }
        const uint32_t wincl = wave_scan_max_dup(as_u(RUN));
        const uint32_t texcl = (uint32_t)wave_shr1((int32_t)wincl, (int32_t)0x80008000u);
        const s16x2 BASE = pk_max(as_pk(texcl), as_pk(STRIP ? sb_row : sbase));
        s16x2 HN[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) HN[u] = pk_max(HNp[u], pk_max(BASE, EX[u]) + JE[u]);
        {
            const uint32_t left = (uint32_t)wave_shr1((int32_t)as_u(HN[NP - 1]), (int32_t)hl_row);       // lane 0: the column left of the window holds H = 0 (STRIP: what the strip to the left computed)
            uint32_t R[2 * NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                R[u] = as_u(pk_left(as_u(HN[u]), u == 0 ? left : as_u(HN[u - 1])));
                R[NP + u] = as_u(pk_max(HN[u] - pk_splat(2), FN[u]));
                PA[u] = as_pk(R[u]); PB[u] = as_pk(R[NP + u]);       // ... and for the next row, should it be a chain row
            }
            // Scalar loads return out of order: whoever uses one waits for lgkmcnt(0), LDS operations included.  So the record of the next
            // row (requested behind this row's predecessor terms) is waited for HERE, a row's worth of instructions after its request and
            // before this row's LDS writes go out -- the next row then starts without a wait, instead of sitting out these writes
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if constexpr (NP == 1) { typedef __attribute__((address_space(3))) u32x2 *e2_p; u32x2 v; v.x = R[0]; v.y = R[1]; *(e2_p)(uintptr_t)(rb + self_off) = v; }
            else {
                typedef __attribute__((address_space(3))) u32x4 *e4_p;
#pragma unroll
                for (int q = 0; q < NP / 2; ++q) { u32x4 v; v.x = R[4 * q]; v.y = R[4 * q + 1]; v.z = R[4 * q + 2]; v.w = R[4 * q + 3]; ((e4_p)(uintptr_t)(rb + self_off))[q] = v; }
            }
            if constexpr (!STRIP) {
                // the entry right of the window: its diagonal source is the H of this row's last column (lane 63; the others hit junk words)
                *(volatile lds_p)(uintptr_t)(a_pad + (lane == 63 ? self_off : 0u)) = as_u(HN[NP - 1]) >> 16;
            } else {
                // for the strip to the right: this row's last H and the prefix maximum of u over every column so far (lane 63 stores)
                const uint32_t tc = (uint32_t)max((int32_t)sb_row, __builtin_amdgcn_readlane((int32_t)wincl, 63));
                if (lane == 63) cout[4 * (uint64_t)row] = (as_u(HN[NP - 1]) >> 16) | (tc << 16);
            }
            uint32_t W[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u)
                asm("v_lshl_or_b32 %0, %1, 14, %2" : "=v"(W[u]) : "v"(as_u(pk_min(HN[u] - FN[u], pk_splat(3)))), "v"(as_u(HN[u])));
            if constexpr (NP == 1) hrow[0] = W[0];
            else if constexpr (NP == 2) { typedef __attribute__((address_space(1))) u32x2 *gptr2_t; u32x2 w2; w2.x = W[0]; w2.y = W[1]; *(gptr2_t)hrow = w2; }
            else { typedef __attribute__((address_space(1))) u32x4 *gptr4_t; u32x4 w4; w4.x = W[0]; w4.y = W[1]; w4.z = W[2]; w4.w = W[3]; *(gptr4_t)hrow = w4; }
#pragma unroll
            for (int u = 0; u < NP; ++u) MXA = pk_max(MXA, HN[u]);
        }
        hrow += 64u * NP;
    };

    {
        u32x8 pa = cpm[0], pb = pa;
        uint32_t row = 1;
        drain_vector_loads();                    // nothing pending at the loop's entry: the compiler then has no reason for a vmcnt wait inside (which would wait for the record stores)
        while (row <= n) {
            step(row, pa, pb);
            ++row;
            if (row > n) break;
            step(row, pb, pa);
            ++row;
        }
    }
    if constexpr (STRIP) { if (mxa_out) *mxa_out = MXA; drain_vector_loads(); }      // (the dwords for the next strip are in memory)
    // best score and the lanes whose cells reach it (columns beyond the sequence score -1 per diagonal move: every such cell is smaller
    // than some valid one, so they need no masking; their rows and columns come from a rescan of the record, kernel body)
    const int32_t lbest = max((int32_t)(int16_t)(as_u(MXA) & 0xFFFFu), (int32_t)as_u(MXA) >> 16);
    best = wave_last(wave_scan_max(lbest, 0));
    if (lane == 0) { X.best[0] = best; X.brow = 0xFFFFFFFFu; X.multi = 1; X.ntl = 0; }
    wave_sync();
    if (best > 0 && lbest == best) {
        const uint32_t slot = atomicAdd(&X.ntl, 1u);
        if (slot < 16) X.tl[slot] = (uint32_t)lane;
    }
}

// ---- rows of 2561 .. 8192 columns (PK == 3): 32-bit cells, up to SIXTEEN wavefronts per pack ------------------
// A pack of long reads used to be one workgroup of four wavefronts with 16-32 columns per lane in 250-410 registers: one
// wavefront per SIMD, one pack per CU, tens of seconds per pack while most of the device idled (config 5).  Here the row is
// spread over 8, 12 or 16 wavefronts of 8 columns per lane (<= 128 registers: four wavefronts per SIMD, the whole CU works on
// one or two packs), built like dp_rows_v3: plan through the scalar cache, every predecessor from the LDS ring (a dword per
// cell: H | min(H - F, 2) << 16) or, beyond RING rows, from the record.  Record as dp_rows: H as an unsigned 16-bit word
// (exact up to 13107 columns) plus a nibble per column behind S.E.  The prefix of u over the wavefronts of the block is a
// 16-lane scan of the exchanged totals.
template <int CPL, int RING, int NW>
__device__ void dp_rows_wide(poa_ws &S, dp_xchg &X, uint32_t n, uint32_t L, uint32_t Lp, int32_t &best, uint32_t &best_row, bool &multi) {
    constexpr int NT = 64 * NW;
    constexpr int NWD = (CPL + 7) / 8;           // dwords of F/E nibbles per thread and row (traceback record)
    static_assert(CPL % 4 == 0 && NW <= 16 && RING > 0 && NT * CPL <= 13056, "16-bit H record: 5 * columns < 65536");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t c0 = (uint32_t)tid * CPL;
    const bool act = c0 < Lp;
    uint32_t sw[CPL / 4];                        // this thread's CPL sequence bytes
    {
        const uint32_t *sp = (const uint32_t *)(S.sq + (act ? c0 : 0));
#pragma unroll
        for (int u = 0; u < CPL / 4; ++u) sw[u] = act ? sp[u] : 0u;
    }
    const bool wave_act = (uint32_t)wave * 64u * CPL < Lp;
    uint32_t *const ring_thr = S.ring + (size_t)tid * CPL;             // slot s of this thread: ring_thr + s * NT * CPL
    uint32_t *const lhr = (uint32_t *)S.lh_ring + wave;                // slot s of this wavefront: lhr[NW * s]
    int32_t lbest = 0;
    const int32_t je0 = ((int32_t)c0 + 1) * POA_E;                      // j * e of the thread's first column

    uint64_t ppa = uni64((uint64_t)S.plan), ppb = uni64((uint64_t)S.planb);
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+s"(ppa), "+s"(ppb) : : "memory");
    const cplan_t cpa = (cplan_t)ppa, cpb = (cplan_t)ppb;

    auto step = [&](auto par_tag, const uint32_t row, const u32x4 pa, const u32x4 pb) __attribute__((always_inline)) {
        constexpr uint32_t par = decltype(par_tag)::value;
        const uint32_t info = pa.x;
        const uint32_t letter = rd_letter(info), n_in = rd_nin(info);
        int32_t hm[CPL], fm[CPL];                // maxima over the predecessors: H[p][j-1] and max(H[p][j] + g - e, F[p][j])
        auto pred = [&](auto first_tag, const uint32_t prow) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            int32_t hp[CPL], fd[CPL];
            int32_t hl;
            if (row - prow <= (uint32_t)RING) {
                const uint32_t slot = prow % (uint32_t)RING;
                const uint4 *rp = (const uint4 *)(ring_thr + slot * (uint32_t)(NT * CPL));
#pragma unroll
                for (int u = 0; u < CPL / 4; ++u) {
                    const uint4 a = rp[u];
                    hp[4 * u] = (int32_t)(a.x & 0xFFFFu); fd[4 * u] = hp[4 * u] - (int32_t)(a.x >> 16);
                    hp[4 * u + 1] = (int32_t)(a.y & 0xFFFFu); fd[4 * u + 1] = hp[4 * u + 1] - (int32_t)(a.y >> 16);
                    hp[4 * u + 2] = (int32_t)(a.z & 0xFFFFu); fd[4 * u + 2] = hp[4 * u + 2] - (int32_t)(a.z >> 16);
                    hp[4 * u + 3] = (int32_t)(a.w & 0xFFFFu); fd[4 * u + 3] = hp[4 * u + 3] - (int32_t)(a.w >> 16);
                }
                hl = (int32_t)lhr[(uint32_t)NW * slot];
            } else {
                hl = 0;
#pragma unroll
                for (int t = 0; t < CPL; ++t) { hp[t] = 0; fd[t] = POA_G - POA_E; }
                if (act) {
                    load_block<CPL>(S.H + (uint64_t)prow * Lp + c0, hp);
                    const uint32_t *np = (const uint32_t *)S.E + ((uint64_t)prow * NT + tid) * NWD;
#pragma unroll
                    for (int t = 0; t < CPL; ++t) fd[t] = hp[t] - min((int32_t)((np[t / 8] >> (4 * (t % 8))) & 3u), 2);   // H - min(H-F, 2)
                    if (lane == 0 && wave > 0) hl = (int32_t)((const uint16_t *)S.H)[(uint64_t)prow * Lp + c0 - 1];
                }
                drain_vector_loads();            // rare path: nothing stays pending past it
            }
            const int32_t hleft = wave_shr1(hp[CPL - 1], hl);
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                const int32_t hd = t == 0 ? hleft : hp[t - 1];
                hm[t] = FIRST ? hd : max(hm[t], hd);
                fm[t] = FIRST ? fd[t] : max(fm[t], fd[t]);
            }
        };
        if (n_in == 0) {                         // virtual start row: H = 0, F = -inf
#pragma unroll
            for (int t = 0; t < CPL; ++t) { hm[t] = 0; fm[t] = POA_G - POA_E; }
        } else {
            pred(std::true_type{}, pb.x);
            if (n_in > 1) pred(std::false_type{}, pb.y);
            if (n_in > 2) pred(std::false_type{}, pb.z);
            if (n_in > 3) {
                pred(std::false_type{}, pb.w);
                uint32_t e = pa.z;
                for (uint32_t k = 4; k < n_in; ++k) {
                    const uint2 ed = S.edges[e]; e = ed.y;
                    const uint32_t prow = (uint32_t)__builtin_amdgcn_readfirstlane(S.rank[ed.x]) + 1;
                    drain_vector_loads();
                    pred(std::false_type{}, prow);
                }
            }
        }
        int32_t hn[CPL], fr[CPL], ex[CPL];
        int32_t run = POA_NEG;
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            const int32_t sc = ((sw[t >> 2] >> (8 * (t & 3))) & 0xFFu) == letter ? POA_M : POA_N;
            fr[t] = fm[t] + POA_E;
            hn[t] = max(max(hm[t] + sc, fr[t]), 0);
            ex[t] = run;
            run = max(run, hn[t] + (POA_G - POA_E) - je0 - t * POA_E);            // u_j = Hn + g - (j+1) e
        }
        const int32_t wincl = wave_scan_max_fused(act ? run : POA_NEG);
        const int32_t texcl = wave_shr1(wincl, POA_NEG);
        int32_t base = POA_G - POA_E;            // u_0
        int32_t hl_new = 0;
        {
            if (lane == 63) {
                X.TW[par][wave] = wincl;
                if (wave < NW - 1) X.QW[par][wave + 1] = make_int2(max(texcl, ex[CPL - 1]), hn[CPL - 1]);
            }
            row_barrier();
            // inclusive prefix maximum of the wavefronts' totals, in lanes 0 .. NW-1 of every wavefront
            int32_t tv = lane < NW ? X.TW[par][lane] : POA_NEG;
            tv = max(tv, __builtin_amdgcn_update_dpp(POA_NEG, tv, 0x111 /*row_shr:1*/, 0xF, 0xF, false));
            tv = max(tv, __builtin_amdgcn_update_dpp(POA_NEG, tv, 0x112 /*row_shr:2*/, 0xF, 0xF, false));
            tv = max(tv, __builtin_amdgcn_update_dpp(POA_NEG, tv, 0x114 /*row_shr:4*/, 0xF, 0xF, false));
            tv = max(tv, __builtin_amdgcn_update_dpp(POA_NEG, tv, 0x118 /*row_shr:8*/, 0xF, 0xF, false));
            if (wave > 0) {
                const int32_t pw = __builtin_amdgcn_readlane(tv, wave - 1);                     // waves 0 .. wave-1
                const int32_t pw1 = wave > 1 ? __builtin_amdgcn_readlane(tv, wave - 2) : POA_NEG;      // waves 0 .. wave-2
                const int2 q = X.QW[par][wave];
                const int32_t qx = __builtin_amdgcn_readfirstlane(q.x), qy = __builtin_amdgcn_readfirstlane(q.y);
                base = max(base, pw);
                const int32_t bp = max(POA_G - POA_E, pw1);
                const int32_t c0w = (int32_t)((uint32_t)wave * 64u * CPL);        // 1-based index of the column left of the wave
                hl_new = max(qy, max(bp, qx) + c0w * POA_E);
            }
        }
        base = max(base, texcl);
        int32_t hN[CPL];
        uint32_t nbw[NWD];
#pragma unroll
        for (int w = 0; w < NWD; ++w) nbw[w] = 0;
        uint32_t rw[CPL];
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            const int32_t ev = max(base, ex[t]) + je0 + t * POA_E;
            hN[t] = max(hn[t], ev);
            lbest = max(lbest, hN[t]);
            const uint32_t df = (uint32_t)min(hN[t] - fr[t], 3);
            nbw[t / 8] |= (df | ((uint32_t)min(hN[t] - ev, 3) << 2)) << (4 * (t % 8));
            rw[t] = (uint32_t)hN[t] | (min(df, 2u) << 16);
        }
        {
            const uint32_t slot = row % (uint32_t)RING;
            uint4 *rp = (uint4 *)(ring_thr + slot * (uint32_t)(NT * CPL));
#pragma unroll
            for (int u = 0; u < CPL / 4; ++u) rp[u] = make_uint4(rw[4 * u], rw[4 * u + 1], rw[4 * u + 2], rw[4 * u + 3]);
            if (lane == 0) lhr[(uint32_t)NW * slot] = (uint32_t)hl_new;
            if (act) {
                uint32_t pkH[CPL / 2];
#pragma unroll
                for (int u = 0; u < CPL / 2; ++u) pkH[u] = pack16(hN[2 * u], hN[2 * u + 1]);
                store_packed<CPL>(S.H + (uint64_t)row * Lp + c0, pkH);
                uint32_t *np = (uint32_t *)S.E + ((uint64_t)row * NT + tid) * NWD;
#pragma unroll
                for (int w = 0; w < NWD; ++w) np[w] = nbw[w];
            }
        }
    };

    if (!wave_act) {
        for (uint32_t r = 0; r < n; ++r) row_barrier();
    } else {
        u32x4 na = cpa[0], nb = cpb[0];          // plan of the next row, one row ahead
        for (uint32_t row = 1; row <= n; row += 2) {
            {
                const u32x4 pa = na, pb = nb;
                if (row < n) { na = cpa[row]; nb = cpb[row]; }
                step(std::integral_constant<uint32_t, 1>{}, row, pa, pb);
            }
            if (row + 1 <= n) {
                const u32x4 pa = na, pb = nb;
                if (row + 1 < n) { na = cpa[row + 1]; nb = cpb[row + 1]; }
                step(std::integral_constant<uint32_t, 0>{}, row + 1, pa, pb);
            }
        }
    }
    // block-wide best score and the threads whose columns reach it (as dp_rows_v3: the rows come from a rescan)
    if (!act) lbest = 0;
    const int32_t wb = wave_last(wave_scan_max(lbest, 0));
    if (lane == 0) X.best[wave] = wb;
    if (tid == 0) { X.brow = 0xFFFFFFFFu; X.multi = 1; X.ntl = 0; }
    __syncthreads();
    best = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) best = max(best, X.best[w]);
    if (best > 0 && lbest == best) {
        const uint32_t slot = atomicAdd(&X.ntl, 1u);
        if (slot < 16) X.tl[slot] = (uint32_t)tid;
    }
    __syncthreads();
    best_row = 0;
    multi = best > 0;
}

// ---- rows longer than any register class (PK == 2): int32 cells, segments of NT * CPL columns ---------------
// Reads beyond 8192 nt are the thin tail of a long-read run (config 5) but must not fail the call, and a pack of them is
// billions of cells.  Same recurrence, no register window and no ring: every predecessor row comes back from the int32
// H matrix and the nibble array (the column left of a thread's block too, so a full __syncthreads() orders the rows), a
// row is a loop over segments of NT * CPL columns (16 wavefronts x 8 columns: 8192), and the running prefix maximum of u
// is carried from segment to segment.  The wavefronts' totals are combined by a 16-lane scan, as in dp_rows_wide.
template <int CPL, int NW>
__device__ void dp_rows_long(poa_ws &S, dp_xchg &X, const uint8_t *s, uint32_t n, uint32_t L, uint32_t Lp, int32_t &best, uint32_t &best_row,
                             bool &multi) {
    static_assert(CPL == 8 && NW <= 16, "two int4 per thread and row segment");
    constexpr int NT = 64 * NW;
    constexpr uint32_t SEG = (uint32_t)NT * CPL;
    int32_t *H = (int32_t *)S.H;
    uint32_t *NB = (uint32_t *)S.E;                  // per eight columns: nibbles min(H-F,3) | min(H-E,3) << 2 (exact traceback record, see dp_rows_v3)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int32_t lbest = 0;
    uint32_t lrow = 0, lcnt = 0, stepc = 0;
    for (uint32_t r0 = 0; r0 < n; r0 += 64) {
        const uint32_t nb = min(64u, n - r0);
        uint4 my = make_uint4(0, 0, 0, 0), myb = make_uint4(0, 0, 0, 0);
        if ((uint32_t)lane < nb) { my = S.plan[r0 + lane]; myb = S.planb[r0 + lane]; }
        for (uint32_t i = 0; i < nb; ++i) {
            const uint32_t info = __builtin_amdgcn_readlane(my.x, i), more = __builtin_amdgcn_readlane(my.z, i);
            const uint32_t pw[4] = {(uint32_t)__builtin_amdgcn_readlane(myb.x, i), (uint32_t)__builtin_amdgcn_readlane(myb.y, i),
                                    (uint32_t)__builtin_amdgcn_readlane(myb.z, i), (uint32_t)__builtin_amdgcn_readlane(myb.w, i)};
            const uint32_t row = r0 + i + 1;
            const uint32_t letter = rd_letter(info), n_in = rd_nin(info);
            int32_t carry = POA_G - POA_E;           // max of u over the columns of all previous segments (u_0 first)
            int32_t rowmax = 0;
            for (uint32_t seg0 = 0; seg0 < Lp; seg0 += SEG, ++stepc) {
                const uint32_t par = stepc & 1u;
                const uint32_t c0 = seg0 + (uint32_t)tid * CPL;
                const bool act = c0 < Lp;
                int32_t hm[CPL], fm[CPL];
#pragma unroll
                for (int t = 0; t < CPL; ++t) { hm[t] = n_in ? POA_NEG : 0; fm[t] = n_in ? POA_NEG : POA_G - POA_E; }
                uint32_t e = more;
                for (uint32_t k = 0; k < n_in; ++k) {
                    uint32_t prow;
                    if (k < 4) prow = k == 0 ? pw[0] : k == 1 ? pw[1] : k == 2 ? pw[2] : pw[3];
                    else { const uint2 ed = S.edges[e]; e = ed.y; prow = (uint32_t)S.rank[ed.x] + 1; }
                    if (!act) continue;
                    const int32_t *hp = H + (uint64_t)prow * Lp + c0;
                    const int4 h0 = *(const int4 *)hp, h1 = *(const int4 *)(hp + 4);
                    const uint32_t nbv = NB[((uint64_t)prow * Lp + c0) >> 3];
                    const int32_t hleft = c0 ? hp[-1] : 0;
                    const int32_t hv[CPL] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                    for (int t = 0; t < CPL; ++t) {
                        hm[t] = max(hm[t], t == 0 ? hleft : hv[t - 1]);
                        fm[t] = max(fm[t], hv[t] - min((int32_t)((nbv >> (4 * t)) & 3u), 2));
                    }
                }
                int32_t hn[CPL], fr[CPL], ex[CPL];
                int32_t run = POA_NEG;
                uint32_t sq[2] = {0, 0};
                if (act) {
#pragma unroll
                    for (int t = 0; t < CPL; ++t) { const uint32_t col = c0 + t; sq[t >> 2] |= (uint32_t)(col < L ? s[col] : 0) << (8 * (t & 3)); }
                }
#pragma unroll
                for (int t = 0; t < CPL; ++t) {
                    const uint32_t col = c0 + t;
                    const int32_t sc = ((sq[t >> 2] >> (8 * (t & 3))) & 0xFFu) == letter ? POA_M : POA_N;
                    fr[t] = act ? fm[t] + POA_E : POA_G;
                    hn[t] = act ? max(max(hm[t] + sc, fr[t]), 0) : 0;
                    ex[t] = run;
                    run = max(run, hn[t] + POA_G - ((int32_t)col + 2) * POA_E);
                }
                const int32_t wincl = wave_scan_max(act ? run : POA_NEG, POA_NEG);
                const int32_t texcl = wave_shr1(wincl, POA_NEG);
                if (lane == 63) X.TW[par][wave] = wincl;
                __syncthreads();                     // also orders this block's H / nibble stores of earlier rows
                int32_t tv = lane < NW ? X.TW[par][lane] : POA_NEG;
                tv = max(tv, __builtin_amdgcn_update_dpp(POA_NEG, tv, 0x111 /*row_shr:1*/, 0xF, 0xF, false));
                tv = max(tv, __builtin_amdgcn_update_dpp(POA_NEG, tv, 0x112 /*row_shr:2*/, 0xF, 0xF, false));
                tv = max(tv, __builtin_amdgcn_update_dpp(POA_NEG, tv, 0x114 /*row_shr:4*/, 0xF, 0xF, false));
                tv = max(tv, __builtin_amdgcn_update_dpp(POA_NEG, tv, 0x118 /*row_shr:8*/, 0xF, 0xF, false));
                const int32_t before = wave > 0 ? __builtin_amdgcn_readlane(tv, wave - 1) : POA_NEG;      // wavefronts 0 .. wave-1 of this segment
                const int32_t total = __builtin_amdgcn_readlane(tv, NW - 1);
                const int32_t base = max(max(carry, before), texcl);
                int32_t hv[CPL];
                uint32_t nbo = 0;
#pragma unroll
                for (int t = 0; t < CPL; ++t) {
                    const int32_t ev = max(base, ex[t]) + ((int32_t)c0 + t + 1) * POA_E;
                    hv[t] = max(hn[t], ev);
                    nbo |= ((uint32_t)min(hv[t] - fr[t], 3) | ((uint32_t)min(hv[t] - ev, 3) << 2)) << (4 * t);
                    rowmax = act ? max(rowmax, hv[t]) : rowmax;
                }
                if (act) {
                    int32_t *hq = H + (uint64_t)row * Lp + c0;
                    *(int4 *)hq = make_int4(hv[0], hv[1], hv[2], hv[3]);
                    *(int4 *)(hq + 4) = make_int4(hv[4], hv[5], hv[6], hv[7]);
                    NB[((uint64_t)row * Lp + c0) >> 3] = nbo;
                }
                carry = max(carry, total);
            }
            const bool gt = rowmax > lbest, eq = rowmax == lbest;
            lcnt = gt ? 1u : lcnt + (eq ? 1u : 0u);
            lrow = gt ? row : lrow;
            lbest = gt ? rowmax : lbest;
        }
    }
    __syncthreads();
    const int32_t wb = wave_last(wave_scan_max(lbest, 0));
    if (lane == 0) X.best[wave] = wb;
    if (tid == 0) { X.brow = 0xFFFFFFFFu; X.multi = 0; X.ntl = 0; }
    __syncthreads();
    best = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) best = max(best, X.best[w]);
    const bool mine = best > 0 && lbest == best;
    if (mine) atomicMin(&X.brow, lrow);
    __syncthreads();
    best_row = best > 0 ? X.brow : 0u;
    if (mine) {
        if (lcnt != 1 || lrow != best_row) X.multi = 1;
        X.ntl = 17;                                  // a thread's columns are spread over the segments: ties rescan whole rows
    }
    __syncthreads();
    multi = X.multi != 0;
}

// ---- the same rows, SEGMENT-MAJOR with an LDS ring (PK == 2, RING > 0) ----------------------------------------------
// dp_rows_long walks a row segment by segment and re-reads every predecessor from HBM: ~3 us per row and segment, a pack of
// 200 reads of 15 kb sits on its CU for 80 s (config 5, profiles/README.md round 3).  A cell (row, segment) depends on the
// predecessor rows of the SAME segment and, through the prefix maximum of u and the H of the column left of the segment, on
// (row, segment - 1): so the segments can be done one after the other, each over ALL rows -- and one segment is exactly
// dp_rows_wide (plan through the scalar cache, predecessors from the ring, one barrier per row) with two numbers per row
// carried from the segment before: Cin[row] = max of u over all columns left of the segment (lh[4 row + parity]) and
// H[row][seg0 - 1] (in the record).  int32 record as dp_rows_long: H plus a nibble per column.
template <int CPL, int RING, int NW>
__device__ void dp_rows_longr(poa_ws &S, dp_xchg &X, const uint8_t *s, uint32_t n, uint32_t L, uint32_t Lp, int32_t &best, uint32_t &best_row,
                              bool &multi) {
    static_assert(CPL == 8 && NW <= 16 && RING > 0, "two int4 per thread, row and segment");
    constexpr int NT = 64 * NW;
    constexpr uint32_t SEG = (uint32_t)NT * CPL;
    int32_t *H = (int32_t *)S.H;
    uint32_t *NB = (uint32_t *)S.E;                  // per eight columns: nibbles min(H-F,3) | min(H-E,3) << 2
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *const ring_thr = S.ring + (size_t)tid * CPL;             // slot s of this thread: ring_thr + s * NT * CPL
    uint32_t *const lhr = (uint32_t *)S.lh_ring + wave;                // slot s of this wavefront: lhr[NW * s]
    int32_t lbest = 0;
    uint32_t lrow = 0, lcnt = 0;                     // first row in which this thread's columns reach lbest, and in how many (row, segment) steps they do
    uint64_t ppa = uni64((uint64_t)S.plan), ppb = uni64((uint64_t)S.planb);
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+s"(ppa), "+s"(ppb) : : "memory");
    const cplan_t cpa = (cplan_t)ppa, cpb = (cplan_t)ppb;

    uint32_t segi = 0;
    for (uint32_t seg0 = 0; seg0 < Lp; seg0 += SEG, ++segi) {
        const uint32_t c0 = seg0 + (uint32_t)tid * CPL;
        const bool act = c0 < Lp;
        const bool wave_act = seg0 + (uint32_t)wave * 64u * CPL < Lp;
        const uint32_t n_aw = min((uint32_t)NW, (Lp - seg0 + 64u * CPL - 1u) / (64u * CPL));      // wavefronts with columns in this segment
        const int32_t *Cin = S.lh + (segi & 1u);                       // Cin[4 * row]: written by the segment before
        int32_t *Cout = S.lh + ((segi + 1u) & 1u);
        uint32_t sw[CPL / 4];
#pragma unroll
        for (int u = 0; u < CPL / 4; ++u) sw[u] = 0;
        if (act) {
#pragma unroll
            for (int t = 0; t < CPL; ++t) { const uint32_t col = c0 + t; sw[t >> 2] |= (uint32_t)(col < L ? s[col] : 0) << (8 * (t & 3)); }
        }
        const int32_t je0 = ((int32_t)c0 + 1) * POA_E;
        __syncthreads();                             // the ring and the exchange area are reused from the segment before; its record stores are visible

        auto step = [&](auto par_tag, const uint32_t row, const u32x4 pa, const u32x4 pb) __attribute__((always_inline)) {
            constexpr uint32_t par = decltype(par_tag)::value;
            const uint32_t info = pa.x;
            const uint32_t letter = rd_letter(info), n_in = rd_nin(info);
            // what the segment before left for this row: the prefix maximum of u, and (wavefront 0) the H left of the segment
            int32_t cin = POA_G - POA_E, hseg = 0;
            if (seg0) {
                cin = __builtin_amdgcn_readfirstlane(Cin[4 * (size_t)row]);
                if (wave == 0) hseg = __builtin_amdgcn_readfirstlane(H[(uint64_t)row * Lp + seg0 - 1]);
            }
            int32_t hm[CPL], fm[CPL];
            auto pred = [&](auto first_tag, const uint32_t prow) __attribute__((always_inline)) {
                constexpr bool FIRST = decltype(first_tag)::value;
                int32_t hp[CPL], fd[CPL];
                int32_t hl;
                if (row - prow <= (uint32_t)RING) {
                    const uint32_t slot = prow % (uint32_t)RING;
                    const uint4 *rp = (const uint4 *)(ring_thr + slot * (uint32_t)(NT * CPL));
#pragma unroll
                    for (int u = 0; u < CPL / 4; ++u) {
                        const uint4 a = rp[u];
                        hp[4 * u] = (int32_t)(a.x & 0x0FFFFFFFu); fd[4 * u] = hp[4 * u] - (int32_t)(a.x >> 30);
                        hp[4 * u + 1] = (int32_t)(a.y & 0x0FFFFFFFu); fd[4 * u + 1] = hp[4 * u + 1] - (int32_t)(a.y >> 30);
                        hp[4 * u + 2] = (int32_t)(a.z & 0x0FFFFFFFu); fd[4 * u + 2] = hp[4 * u + 2] - (int32_t)(a.z >> 30);
                        hp[4 * u + 3] = (int32_t)(a.w & 0x0FFFFFFFu); fd[4 * u + 3] = hp[4 * u + 3] - (int32_t)(a.w >> 30);
                    }
                    hl = (int32_t)lhr[(uint32_t)NW * slot];
                } else {
                    hl = 0;
#pragma unroll
                    for (int t = 0; t < CPL; ++t) { hp[t] = 0; fd[t] = POA_G - POA_E; }
                    if (act) {
                        const int32_t *hq = H + (uint64_t)prow * Lp + c0;
                        const int4 h0 = *(const int4 *)hq, h1 = *(const int4 *)(hq + 4);
                        const uint32_t nbv = NB[((uint64_t)prow * Lp + c0) >> 3];
                        if (lane == 0 && c0) hl = hq[-1];
                        hp[0] = h0.x; hp[1] = h0.y; hp[2] = h0.z; hp[3] = h0.w; hp[4] = h1.x; hp[5] = h1.y; hp[6] = h1.z; hp[7] = h1.w;
#pragma unroll
                        for (int t = 0; t < CPL; ++t) fd[t] = hp[t] - min((int32_t)((nbv >> (4 * t)) & 3u), 2);
                    }
                    drain_vector_loads();
                }
                const int32_t hleft = wave_shr1(hp[CPL - 1], hl);
#pragma unroll
                for (int t = 0; t < CPL; ++t) {
                    const int32_t hd = t == 0 ? hleft : hp[t - 1];
                    hm[t] = FIRST ? hd : max(hm[t], hd);
                    fm[t] = FIRST ? fd[t] : max(fm[t], fd[t]);
                }
            };
            if (n_in == 0) {
#pragma unroll
                for (int t = 0; t < CPL; ++t) { hm[t] = 0; fm[t] = POA_G - POA_E; }
            } else {
                pred(std::true_type{}, pb.x);
                if (n_in > 1) pred(std::false_type{}, pb.y);
                if (n_in > 2) pred(std::false_type{}, pb.z);
                if (n_in > 3) {
                    pred(std::false_type{}, pb.w);
                    uint32_t e = pa.z;
                    for (uint32_t k = 4; k < n_in; ++k) {
                        const uint2 ed = S.edges[e]; e = ed.y;
                        const uint32_t prow = (uint32_t)__builtin_amdgcn_readfirstlane(S.rank[ed.x]) + 1;
                        drain_vector_loads();
                        pred(std::false_type{}, prow);
                    }
                }
            }
            int32_t hn[CPL], fr[CPL], ex[CPL];
            int32_t run = POA_NEG;
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                const int32_t sc = ((sw[t >> 2] >> (8 * (t & 3))) & 0xFFu) == letter ? POA_M : POA_N;
                fr[t] = fm[t] + POA_E;
                hn[t] = max(max(hm[t] + sc, fr[t]), 0);
                ex[t] = run;
                run = max(run, hn[t] + (POA_G - POA_E) - je0 - t * POA_E);
            }
            const int32_t wincl = wave_scan_max_fused(act ? run : POA_NEG);
            const int32_t texcl = wave_shr1(wincl, POA_NEG);
            int32_t base = cin;
            int32_t hl_new = hseg;
            if (lane == 63) {
                X.TW[par][wave] = wincl;
                if (wave < NW - 1) X.QW[par][wave + 1] = make_int2(max(texcl, ex[CPL - 1]), hn[CPL - 1]);
            }
            row_barrier();
            int32_t tv = lane < NW ? X.TW[par][lane] : POA_NEG;
            tv = max(tv, __builtin_amdgcn_update_dpp(POA_NEG, tv, 0x111 /*row_shr:1*/, 0xF, 0xF, false));
            tv = max(tv, __builtin_amdgcn_update_dpp(POA_NEG, tv, 0x112 /*row_shr:2*/, 0xF, 0xF, false));
            tv = max(tv, __builtin_amdgcn_update_dpp(POA_NEG, tv, 0x114 /*row_shr:4*/, 0xF, 0xF, false));
            tv = max(tv, __builtin_amdgcn_update_dpp(POA_NEG, tv, 0x118 /*row_shr:8*/, 0xF, 0xF, false));
            if (wave > 0) {
                const int32_t pw = __builtin_amdgcn_readlane(tv, wave - 1);
                const int32_t pw1 = wave > 1 ? __builtin_amdgcn_readlane(tv, wave - 2) : POA_NEG;
                const int2 q = X.QW[par][wave];
                const int32_t qx = __builtin_amdgcn_readfirstlane(q.x), qy = __builtin_amdgcn_readfirstlane(q.y);
                base = max(base, pw);
                const int32_t bp = max(cin, pw1);
                const int32_t c0w = (int32_t)(seg0 + (uint32_t)wave * 64u * CPL);        // 1-based index of the column left of the wave
                hl_new = max(qy, max(bp, qx) + c0w * POA_E);
            } else if (tid == 0) {
                Cout[4 * (size_t)row] = max(cin, __builtin_amdgcn_readlane(tv, n_aw - 1));     // for the next segment (idle wavefronts publish nothing)
            }
            base = max(base, texcl);
            uint32_t nbo = 0;
            uint32_t rw[CPL];
            int32_t hv[CPL];
            int32_t rmax = 0;
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                const int32_t ev = max(base, ex[t]) + je0 + t * POA_E;
                hv[t] = max(hn[t], ev);
                rmax = max(rmax, hv[t]);
                const uint32_t df = (uint32_t)min(hv[t] - fr[t], 3);
                nbo |= (df | ((uint32_t)min(hv[t] - ev, 3) << 2)) << (4 * t);
                rw[t] = (uint32_t)hv[t] | (min(df, 2u) << 30);
            }
            if (act) {                               // per thread: best value, the smallest row that reaches it, how often it is reached
                const bool gt = rmax > lbest, eq = rmax == lbest;
                lcnt = gt ? 1u : lcnt + (eq ? 1u : 0u);
                lrow = gt ? row : (eq ? min(lrow, row) : lrow);
                lbest = gt ? rmax : lbest;
            }
            {
                const uint32_t slot = row % (uint32_t)RING;
                uint4 *rp = (uint4 *)(ring_thr + slot * (uint32_t)(NT * CPL));
#pragma unroll
                for (int u = 0; u < CPL / 4; ++u) rp[u] = make_uint4(rw[4 * u], rw[4 * u + 1], rw[4 * u + 2], rw[4 * u + 3]);
                if (lane == 0) lhr[(uint32_t)NW * slot] = (uint32_t)hl_new;
                if (act) {
                    int32_t *hq = H + (uint64_t)row * Lp + c0;
                    *(int4 *)hq = make_int4(hv[0], hv[1], hv[2], hv[3]);
                    *(int4 *)(hq + 4) = make_int4(hv[4], hv[5], hv[6], hv[7]);
                    NB[((uint64_t)row * Lp + c0) >> 3] = nbo;
                }
            }
        };

        if (!wave_act) {
            for (uint32_t r = 0; r < n; ++r) row_barrier();
        } else {
            u32x4 na = cpa[0], nb = cpb[0];
            for (uint32_t row = 1; row <= n; row += 2) {
                {
                    const u32x4 pa = na, pb = nb;
                    if (row < n) { na = cpa[row]; nb = cpb[row]; }
                    step(std::integral_constant<uint32_t, 1>{}, row, pa, pb);
                }
                if (row + 1 <= n) {
                    const u32x4 pa = na, pb = nb;
                    if (row + 1 < n) { na = cpa[row + 1]; nb = cpb[row + 1]; }
                    step(std::integral_constant<uint32_t, 0>{}, row + 1, pa, pb);
                }
            }
        }
        drain_vector_loads();
    }
    __syncthreads();
    const int32_t wb = wave_last(wave_scan_max(lbest, 0));
    if (lane == 0) X.best[wave] = wb;
    if (tid == 0) { X.brow = 0xFFFFFFFFu; X.multi = 0; X.ntl = 0; }
    __syncthreads();
    best = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) best = max(best, X.best[w]);
    const bool mine = best > 0 && lbest == best;
    if (mine) atomicMin(&X.brow, lrow);
    __syncthreads();
    best_row = best > 0 ? X.brow : 0u;
    if (mine) {
        if (lcnt != 1 || lrow != best_row) X.multi = 1;      // another row (or the same row in another thread's later pass) reaches it too
        X.ntl = 17;                                  // a thread's columns are spread over the segments: ties rescan whole rows
    }
    __syncthreads();
    multi = X.multi != 0;
}

// ---- graph update helpers (lane 0) --------------------------------------------------------------
__device__ __forceinline__ uint32_t g_add_node(poa_ws &S, const poa_args &A, uint8_t letter) {
    if (S.n_nodes >= A.node_cap) { S.err = POA_ERR_NODES; return 0; }
    S.nrec[S.n_nodes] = make_uint4(letter, 0, POA_NONE, POA_NONE);
    return S.n_nodes++;
}

// Graph::add_edge: nothing if begin->end exists, else append to end's in-edge list.  Called by
// many threads at once for DISTINCT end nodes; edge slots come from an LDS counter.
__device__ __forceinline__ void g_add_edge(poa_ws &S, const poa_args &A, uint32_t b, uint32_t en, uint32_t *edge_counter, uint32_t *err) {
    uint4 nd = S.nrec[en];
    const uint32_t n_in = rd_nin(nd.x);
    if (n_in > 0) {
        if (nd.y == b) return;
        uint32_t e = nd.z;
        for (uint32_t k = 1; k < n_in; ++k) { const uint2 ed = S.edges[e]; if (ed.x == b) return; e = ed.y; }
    }
    if (n_in == 0) {
        nd.y = b;
    } else {
        const uint32_t id = atomicAdd(edge_counter, 1u);
        if (id >= A.edge_cap) { *err = POA_ERR_EDGES; return; }
        S.edges[id] = make_uint2(b, POA_NONE);
        if (nd.z == POA_NONE) nd.z = id;
        else S.edges[nd.w].y = id;
        nd.w = id;
    }
    if (n_in >= 0xFFFF) { *err = POA_ERR_GRAPH; return; }
    nd.x += 1u << 16;
    S.nrec[en] = nd;
}

// Graph::add_sequence(b, e): fresh chain; returns first node or -1.  Records the path.
__device__ __forceinline__ int32_t g_add_chain(poa_ws &S, const poa_args &A, const uint8_t *s, uint32_t b, uint32_t e, uint32_t *path) {
    if (b == e) return -1;
    const uint32_t first = g_add_node(S, A, s[b]);
    if (S.err) return -1;
    path[b] = first;
    for (uint32_t i = b + 1; i < e; ++i) {
        if (S.n_nodes >= A.node_cap) { S.err = POA_ERR_NODES; return -1; }
        const uint32_t id = S.n_nodes++;
        S.nrec[id] = make_uint4((uint32_t)s[i] | (1u << 16), id - 1, POA_NONE, POA_NONE);     // its only in-edge
        path[i] = id;
    }
    return (int32_t)first;
}

// PK: 0 = 32-bit registers + int16 record + nibbles (dp_rows), 1 = packed int16 pairs, record word H | min(H-F,3) << 14 (dp_rows_v3),
// 2 = int32 segments (dp_rows_long / _longr), 3 = 32-bit cells on up to 16 wavefronts (dp_rows_wide),
// 7 = the same record written by teams of wavefronts (dp_rows_mt): the RING template argument is the number of TEAMS, the ring is sized at launch,
// 8 = the exact band for near-chain graphs ALONE (dp_rows_band on one wavefront, the record word of PK 1 in band layout): a light kernel of one or four
//     wavefronts per pack for any length up to 2560; a pack with an alignment whose band cannot be certified fails with POA_ERR_BAND and is run again by
//     the full-row kernels (the host's retry loop).  CPL and RING of a PK 8 instance mean nothing.
__host__ __device__ constexpr bool pk_packed(int PK) { return PK == 1 || PK == 7 || PK == 8; }
__host__ __device__ constexpr bool pk_readymade(int PK) { return PK == 7; }
__host__ __device__ constexpr int pk_teams(int RING, int PK) { return PK == 7 ? RING : 1; }
__host__ __device__ constexpr uint32_t mt_slot_bytes(int CPL, int NW) { return 64u * NW * CPL * 4u; }      // one ring slot of dp_rows_mt: 4 bytes per cell
// dwords of LDS ring per thread and row; bytes of the ring (+ the left-column values of the barrier forms); bytes of the LDS behind
// the sequence: the graph walks' two node bitmaps, their DFS stack, the ring
__host__ __device__ constexpr uint32_t poa_ring_words(int CPL, int NW, int PK) { return pk_readymade(PK) ? CPL : !pk_packed(PK) && 64 * NW * CPL > 2048 ? CPL : CPL / 2; }
__host__ __device__ constexpr uint32_t poa_ring_bytes(int CPL, int RING, int NW, int PK) {
    return (uint32_t)RING * 64u * NW * poa_ring_words(CPL, NW, PK) * 4u + (uint32_t)RING * 4u * (NW > 4 ? NW : 4);
}
__host__ __device__ constexpr uint32_t poa_region_bytes(uint32_t node_cap, int CPL, int RING, int NW, int PK) {
    return (2u * poa_bit_words(node_cap) + POA_STACK) * 4u + poa_ring_bytes(CPL, RING, NW, PK);      // bitmaps, stack, ring: back to back
}

// minimum wavefronts per SIMD the register allocation is held to (the kernel is bound by the latency of a row's dependent
// instruction chain, hidden only by other resident wavefronts: occupancy first)
constexpr int poa_min_waves(int CPL, int NW, int PK, int RING_OR_T = 1) {
    return PK == 8 ? 4
         : PK == 7 ? (RING_OR_T == 2 && CPL == 4 ? MT2_MINWAVES : 4)
         : PK == 3 ? (NW == 12 ? 3 : 4) : PK == 2 ? 1 : NW == 4 && CPL == 4 ? (PK ? POA_MW_4x4 : 5) : NW == 4 && CPL == 6 ? (PK ? POA_MW_4x6 : 4) : 1;      // (PK 8 as PK 1)
}
template <int CPL, int RING, int NW, int PK>
__global__ __launch_bounds__(64 * NW * pk_teams(RING, PK), poa_min_waves(CPL, NW, PK, RING)) void poa_kernel(poa_args A) {
    constexpr uint32_t NT = 64 * NW * pk_teams(RING, PK);      // threads of the workgroup
    constexpr uint32_t NTC = 64 * NW;                          // threads a row is spread over
    using cell_t = typename std::conditional<PK == 2, int32_t, uint16_t>::type;     // DP matrix cell (16-bit records hold H >= 0 unsigned)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t s_pack;
    __shared__ uint32_t s_alpha;              // letters seen in the pack so far: bit 0 T, bit 1 U, bit 2 anything but A, C, G, T, U
    __shared__ uint32_t s_bc[8];
    __shared__ uint32_t s_tied[8];
    __shared__ dp_xchg X;
    const int tid = threadIdx.x;
    const bool w0 = tid < 64;                 // wave 0 runs the serial graph phases
    poa_ws S;
    {
        uint8_t *base = A.arena + (uint64_t)blockIdx.x * A.slot_stride;
        S.nrec = (uint4 *)(base + A.o_nrec); S.nal = (uint4 *)(base + A.o_nal); S.edges = (uint2 *)(base + A.o_edges);
        S.rank = (int32_t *)(base + A.o_rank); S.order = (uint32_t *)(base + A.o_order); S.order2 = (uint32_t *)(base + A.o_order2);
        S.srank = (int32_t *)(base + A.o_srank); S.rowmax = (int32_t *)(base + A.o_rowmax); S.lh = (int32_t *)(base + A.o_lh); S.nn = (uint32_t *)(base + A.o_nn); S.plan = (uint4 *)(base + A.o_plan); S.planb = (uint4 *)(base + A.o_planb); S.planc = (uint4 *)(base + A.o_planc); S.pland = (uint4 *)(base + A.o_pland); S.planm = (uint32_t *)(base + A.o_planm);
        S.H = (int16_t *)(base + A.o_H); S.F = (int16_t *)(base + A.o_F); S.E = (int16_t *)(base + A.o_E);
        S.aln = (int32_t *)(base + A.o_aln); S.ainfo = (uint4 *)(base + A.o_ainfo); S.spill = (uint32_t *)(base + A.o_spill);
        const uint32_t bit_words = poa_bit_words(A.node_cap);
        S.sq = (uint8_t *)lds;                                   // seq_cap bytes (multiple of 16)
        // (Sharing ONE region between the ring and the graph walks' bitmaps + stack was tried in round 4: 4.6 KB per workgroup, but the
        // tie labels and the traceback's chain live in that region too and lost the room they need at 1536 columns with a ring of
        // four rows -- ties fell back to the full sort, the 1536-column class went from 834 to 584 GCUPS.  Back to back again.)
        S.done = lds + A.seq_cap / 4; S.nocheck = S.done + bit_words; S.stack = S.nocheck + bit_words;
        S.hist = A.counters;
        S.ring = S.stack + POA_STACK;
        S.lh_ring = (int32_t *)(S.ring + (size_t)(PK == 7 || PK == 8 ? 0 : RING) * NT * poa_ring_words(CPL, NW, PK));
    }
    // bytes of the LDS ring (between two DPs: room for the tie labels and the traceback's chain)
    const uint32_t ring_bytes = PK == 8 ? A.band : PK == 7 ? A.ring_slots * mt_slot_bytes(CPL, NW) : poa_ring_bytes(CPL, RING, NW, PK);

    while (true) {
        __syncthreads();
        if (tid == 0) s_pack = atomicAdd(A.queue_head, 1u);
        __syncthreads();
        const uint32_t qi = s_pack;
        if (qi >= A.n_queue) break;
        const uint32_t pk = A.queue[qi];
        const uint32_t q0 = A.pack_first[pk], q1 = A.pack_first[pk + 1];
        if (A.timeline && tid == 0) A.timeline[2 * pk] = (unsigned long long)wall_clock64();
        S.n_nodes = 0; S.n_edges = 0; S.err = 0; S.sp = 0; S.spilled = 0;
        if (tid == 0) s_alpha = 0;
        unsigned long long cells = 0, rows = 0, t_topo = 0, t_dp = 0, t_tb = 0, t_add = 0, t_tie = 0, t_merge = 0;
        // the exact band (PK == 8): cells really computed, alignments whose band was certified / whose certificate failed, the t of the last
        // certified alignment of this pack (the next one asks for twice that), and the band of the alignment in hand
        unsigned long long cells_done = 0;
        uint32_t n_band_ok = 0, n_band_fail = 0, n_strip_aln = 0, bd_thist = 0, bd_sh = 0, bd_strip = 0;      // bd_strip != 0: the record is strip-major, bd_strip rows per strip
        bool band_on = false;
        (void)n_band_ok; (void)n_band_fail; (void)n_strip_aln; (void)bd_thist; (void)bd_sh; (void)bd_strip;
        (void)t_topo; (void)t_dp; (void)t_tb; (void)t_add; (void)t_tie; (void)t_merge;
        const unsigned long long t_pack0 = PT_NOW();
        (void)t_pack0;

        for (uint32_t q = q0; q < q1 && !S.err; ++q) {
            const uint64_t so = A.off[q];
            const uint32_t L = (uint32_t)(A.off[q + 1] - so);
            const uint8_t *s = A.seq + so;
            uint32_t *path = A.out_col + so;
            if (L == 0) continue;                                   // Graph::add_alignment ignores an empty sequence
            if (pk_packed(PK)) {
                // alphabet of the pack (this sequence included): decides whether the packed rows may take the score table
                uint32_t fl = 0;
                for (uint32_t t = tid; t < L; t += NT) {
                    const uint8_t c = s[t];
                    fl |= c == 'T' ? 1u : c == 'U' ? 2u : (c == 'A' || c == 'C' || c == 'G') ? 0u : 4u;
                }
                __syncthreads();
                if (fl) atomicOr(&s_alpha, fl);
                __syncthreads();
                S.plain = (s_alpha & 4u) == 0 && (s_alpha & 3u) != 3u;
            }
            uint32_t n_aln = 0;
            if (S.n_nodes > 0) {
                const uint32_t n = S.n_nodes;
                const uint32_t Lp = (L + CPL - 1) / CPL * CPL;
                if (PK == 8 ? ((uint64_t)(n + 1) * 512u > A.cell_cap || L > 2560u) : ((uint64_t)(n + 1) * Lp > A.cell_cap || (PK != 2 && Lp > NTC * CPL))) { S.err = POA_ERR_CELLS; break; }
                // ---- 1. rows are taken in the incrementally maintained block order (merge_order) ----
                unsigned long long t0 = PT_NOW();
                // ---- 2. plan + sequence to LDS (all threads) ----
#ifdef POA_PREDSTAT
                unsigned long long ps_in = 0, ps_far = 0, ps_prev = 0;      // measurement build: in-edges, those beyond the ring, those from the previous row
#endif
                for (uint32_t r = tid; r < n; r += NT) {
                    const uint32_t v = S.order[r];
                    const uint4 rec = S.nrec[v];
                    const uint32_t n_in = rd_nin(rec.x);
                    uint4 pr = make_uint4(0, 0, 0, 0), pr2 = make_uint4(0, 0, 0, 0);
                    uint32_t dy = 0, dz = 0;
                    uint32_t e = rec.z, e5 = POA_NONE;
                    for (uint32_t k = 0; k < n_in && k < 8; ++k) {
                        uint32_t b;
                        if (k == 4) e5 = e;
                        if (k == 0) b = rec.y; else { const uint2 ed = S.edges[e]; e = ed.y; b = ed.x; }
                        const uint32_t prow_k = (uint32_t)S.rank[b] + 1;
                        if (k < 4) u4_set(pr, k, prow_k); else u4_set(pr2, k - 4, prow_k);
                        {   // the compact record's distance byte (saturated)
                            const uint32_t d = min(r + 1 - prow_k, 255u);
                            if (k < 4) dy |= d << (8 * k); else dz |= d << (8 * (k - 4));
                        }
#ifdef POA_PREDSTAT
                        { const uint32_t d = r + 1 - ((uint32_t)S.rank[b] + 1); ++ps_in; if (d > (uint32_t)(RING > 0 ? RING : 1)) ++ps_far; if (d == 1) ++ps_prev; }
#endif
                    }
#ifdef POA_PREDSTAT
                    if (n_in > 8) ps_in += n_in - 8;
#endif
                    if (n_in <= 4) e5 = e;
                    S.plan[r] = make_uint4(rec.x, v, e5, e);        // .z: 5th in-edge (general code paths), .w: 9th (packed rows)
                    S.planb[r] = pr;
                    if (pk_packed(PK)) {
                        S.planc[r] = pr2;
                        // the row loop's own, compact record (one s_load_dwordx4 per row instead of three, 4 plan registers instead
                        // of 12): x = letter | min(n_in, 255) << 8, y / z = distances row - predecessor row of in-edges 1-4 / 5-8 in a
                        // byte each (saturated at 255: such a predecessor is far beyond the ring, and the row loop takes the rows of
                        // everything beyond the ring from planb / planc), w = edge index of the 9th in-edge
                        S.pland[r] = make_uint4(rd_letter(rec.x) | (min(n_in, 255u) << 8), dy, dz, e);
                    }
                    if constexpr (PK == 7) {
                        // dp_rows_mt's record, everything ready-made (see there): score table halves, LDS offsets of the row's own ring
                        // slot and of the slots of its first eight predecessors (a repeated entry for the missing ones and for the ones
                        // beyond the ring), the row that must be final before this one starts
                        constexpr uint32_t SLOTB = mt_slot_bytes(CPL, NW);
                        const uint32_t slots = A.ring_slots, reach = A.ring_reach;
                        uint32_t po[8], farmask = 0, need_row = 0, base = 0xFFFFFFFFu;
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) {
                            po[k] = 0xFFFFFFFFu;
                            if (k < n_in) {
                                const uint32_t prow_k = k < 4 ? u4_get(pr, k) : u4_get(pr2, k - 4);
                                need_row = max(need_row, prow_k);            // in the ring or not: the row must be final (and its record stored) before this one reads it
                                if (r + 1 - prow_k <= reach) { po[k] = (prow_k % slots) * SLOTB; if (base == 0xFFFFFFFFu) base = po[k]; }
                                else farmask |= 1u << k;
                            }
                        }
                        // missing in-edges (and the ones beyond the ring) read the entry of an in-edge that IS in the ring once more: the
                        // maximum does not care, and no slot has to be set aside for a neutral entry.  No in-edge in the ring at all: the
                        // row's own slot (any valid address), and MT_NOBASE tells the loop to start from the neutral element instead
                        const bool nobase = base == 0xFFFFFFFFu;
                        if (nobase) base = ((r + 1) % slots) * SLOTB;
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) if (po[k] == 0xFFFFFFFFu) po[k] = base;
                        if (n_in > 8) {                                  // (rare) the rows of the further in-edges count for the wait as well
                            uint32_t e2 = e;
                            for (uint32_t k = 8; k < n_in; ++k) { const uint2 ed = S.edges[e2]; e2 = ed.y; need_row = max(need_row, (uint32_t)S.rank[ed.x] + 1); }
                        }
                        const int32_t need = max((int32_t)need_row, (int32_t)(r + 1) - (int32_t)A.ring_slack);
                        const uint32_t letter = rd_letter(rec.x), li = (letter >> 1) & 3u;
                        const uint32_t klo = 0xFCFCFCFCu ^ (0xF9u << (8u * li)), khi = 0xFFFFFFFFu ^ (0xFFu << (8u * li));      // low / high bytes of {-4, -4, -4, -4} with 5 at li
                        const uint32_t ctl = (n_in > 4 ? MT_MORE4 : 0u) | ((farmask || n_in > 8 || nobase) ? MT_SLOW : 0u) | (nobase ? MT_NOBASE : 0u) | (farmask << 8) | (letter << 16);
                        uint4 *pm = (uint4 *)(S.planm + 8 * (size_t)r);
                        const uint32_t self_u = (((r + 1) % slots) * SLOTB) >> MT_UNIT;
                        pm[0] = make_uint4(klo, khi, (uint32_t)need, self_u | (ctl << 8));
                        pm[1] = make_uint4((po[0] >> MT_UNIT) | (po[1] >> MT_UNIT) << 8 | (po[2] >> MT_UNIT) << 16 | (po[3] >> MT_UNIT) << 24,
                                           (po[4] >> MT_UNIT) | (po[5] >> MT_UNIT) << 8 | (po[6] >> MT_UNIT) << 16 | (po[7] >> MT_UNIT) << 24, n_in, e);
                    }
                }
#ifdef POA_PREDSTAT
                atomicAdd(&A.counters[4], ps_in); atomicAdd(&A.counters[5], ps_far); atomicAdd(&A.counters[6], ps_prev);
#endif
                if (PK != 2) for (uint32_t t = tid; t < Lp; t += NT) S.sq[t] = t < L ? s[t] : 0;       // long rows read the sequence in place
                __syncthreads();
                unsigned long long t1 = PT_NOW();
                t_topo += t1 - t0;
                // ---- 3. DP (4 waves) ----
                int32_t best; uint32_t best_row;
                bool multi = false;
                band_on = false; bd_strip = 0;
                if constexpr (PK == 8) {
                    // ---- 3a. the exact band (dp_rows_band): near-chain graphs, one wavefront, certificate checked behind the rows ----
                    if (S.plain) {
                        // ---- the MSA column of every row (block order: the rows of an aligned group are contiguous and share a column) ----
                        // a row opens a column iff none of its group mates has a smaller row; columns = the running count.  col[] goes to
                        // rowmax[] (idle until the rescan behind the DP), one flag byte per row to the band's LDS (idle until the attempt)
                        uint8_t *flag = (uint8_t *)S.ring;
                        int32_t *colr = S.rowmax;                       // colr[row], row = 1 .. n
                        const uint32_t chunk = (n + NT - 1u) / NT;
                        bool fits_flags = n <= ring_bytes;
                        if (fits_flags) {
                            for (uint32_t r = tid; r < n; r += NT) {
                                const uint4 pl = S.plan[r];
                                const uint32_t n_al = rd_nal(pl.x);
                                uint32_t first = 1;
                                if (n_al) {
                                    const uint4 al = S.nal[pl.y];
                                    for (uint32_t k = 0; k < n_al; ++k) if ((uint32_t)S.rank[u4_get(al, k)] < r) first = 0;
                                }
                                flag[r] = (uint8_t)first;
                            }
                            __syncthreads();
                            uint32_t cnt = 0;
                            for (uint32_t i = tid * chunk; i < min(n, (tid + 1u) * chunk); ++i) cnt += flag[i];
                            const uint32_t incl = wave_scan_add(cnt);
                            if ((tid & 63) == 63) s_bc[(tid >> 6) & 7] = incl;          // (NT / 64 <= 4 wavefronts)
                            __syncthreads();
                            uint32_t off = incl - cnt;
                            for (uint32_t w = 0; w < (uint32_t)(tid >> 6); ++w) off += s_bc[w];
                            uint32_t tot = 0;
                            for (uint32_t w = 0; w < NT / 64u; ++w) tot += s_bc[w];
                            for (uint32_t i = tid * chunk; i < min(n, (tid + 1u) * chunk); ++i) { off += flag[i]; colr[i + 1] = (int32_t)off; }
                            __syncthreads();
                            if (tid == 0) { s_bc[0] = tot; s_bc[1] = 0; }
                            __syncthreads();
                        }
                        // (wave-uniform numbers the compiler cannot know to be uniform -- they come through LDS -- are declared so: the band's
                        // row loop addresses its plan through the scalar cache)
                        const uint32_t nu = (uint32_t)__builtin_amdgcn_readfirstlane((int)n), Lu = (uint32_t)__builtin_amdgcn_readfirstlane((int)L), rbu = (uint32_t)__builtin_amdgcn_readfirstlane((int)ring_bytes);
                        const uint32_t Cu = fits_flags ? (uint32_t)__builtin_amdgcn_readfirstlane((int)s_bc[0]) : 0u;      // columns of the graph
                        uint32_t t_want = max(8u, 2u * bd_thist + 4u);
                        for (uint32_t bsh = 1; bsh <= 3u && !band_on && fits_flags && !(A.debug & 16u); ++bsh) {
                            const uint32_t cplb = 1u << bsh;
                            const int32_t room = (int32_t)(63u * cplb) - 1 - ((int32_t)Cu - (int32_t)Lu);      // C - L + 2 t + 1 <= 63 cplb
                            if (room < 0) continue;
                            const uint32_t t = min((uint32_t)room >> 1, Lu - 1u);      // all the room there is: the window costs the same
                            // (tau = L - t <= C: some path may have tau diagonal moves.)  Two columns per lane only when the room they leave is what the
                            // last certified alignment of this pack needed twice over; four or eight columns per lane with whatever room there is
                            if ((bsh == 1u && t < t_want) || (int32_t)Cu < (int32_t)Lu - (int32_t)t) continue;
                            const uint32_t bs = Cu + t - Lu;
                            const uint32_t Lpb = (Lu + cplb - 1u) & ~(cplb - 1u);
                            uint32_t brs = 16;
                            while (brs >= 2u && band_lds_bytes(brs, (int)cplb, Lpb) > rbu) brs >>= 1;
                            if (brs < 2u) continue;
                            const uint32_t eb = 4u * cplb, slotb = (64u + BAND_PAD) * eb, np = cplb >> 1;
                            const uint32_t sel_off = brs * slotb, junk_off = sel_off + (Lpb / cplb + 66u) * np * 4u;
                            // the rows' band records: ring offsets of the in-edges with the lane shift folded in (see dp_rows_band)
                            for (uint32_t r = tid; r < n; r += NT) {
                                const uint32_t row = r + 1u;
                                const uint4 pl = S.plan[r], pb = S.planb[r], pc = S.planc[r];
                                const uint32_t n_in = rd_nin(pl.x);
                                const uint32_t cr = (uint32_t)colr[row];
                                const uint32_t g = band_g0(cr, bs, bsh), gprev = row > 1u ? band_g0((uint32_t)colr[row - 1u], bs, bsh) : 0u;
                                uint32_t po[8], farmask = 0, base = 0xFFFFFFFFu;
                                bool bad = false;
#pragma unroll
                                for (uint32_t k = 0; k < 8; ++k) {
                                    po[k] = 0xFFFFFFFFu;
                                    if (k < n_in) {
                                        const uint32_t prow = k < 4 ? u4_get(pb, k) : u4_get(pc, k - 4);
                                        const uint32_t cp = (uint32_t)colr[prow];
                                        bad |= cp >= cr;                       // an edge that does not lead to a later column: the bound above does not hold
                                        const uint32_t d = g - band_g0(cp, bs, bsh);
                                        if (row - prow <= brs && d <= BAND_PAD) { po[k] = (prow & (brs - 1u)) * slotb + d * eb; if (base == 0xFFFFFFFFu) base = po[k]; }
                                        else farmask |= 1u << k;
                                    }
                                }
                                if (n_in > 8) {
                                    uint32_t e = pl.w;
                                    for (uint32_t k = 8; k < n_in; ++k) { const uint2 ed = S.edges[e]; e = ed.y; bad |= (uint32_t)colr[(uint32_t)S.rank[ed.x] + 1u] >= cr; }
                                }
                                if (bad) s_bc[1] = 1;
                                const bool nobase = base == 0xFFFFFFFFu;
                                if (nobase) base = (row & (brs - 1u)) * slotb;
#pragma unroll
                                for (uint32_t k = 0; k < 8; ++k) if (po[k] == 0xFFFFFFFFu) po[k] = base;
                                const uint32_t letter = rd_letter(pl.x), li = (letter >> 1) & 3u;
                                const uint32_t klo = 0xFCFCFCFCu ^ (0xF9u << (8u * li)), khi = 0xFFFFFFFFu ^ (0xFFu << (8u * li));
                                uint32_t ctl = (n_in > 4 ? BD_MORE4 : 0u) | ((farmask || n_in > 8 || nobase) ? BD_SLOW : 0u) | (nobase ? BD_NOBASE : 0u);
                                if (n_in == 1u && row > 1u && pb.x == row - 1u && !farmask && g == gprev) ctl |= BD_CHAIN;
                                uint4 *pm = (uint4 *)(S.planm + 8 * (size_t)r);
                                pm[0] = make_uint4(klo, khi, po[0] | po[1] << 16, po[2] | po[3] << 16);
                                pm[1] = make_uint4(((row & (brs - 1u)) * slotb) | g << 16, min(g - gprev, 255u) | farmask << 8 | ctl << 16 | min(n_in, 255u) << 24, po[4] | po[5] << 16, po[6] | po[7] << 16);
                            }
                            __syncthreads();                             // (the flag bytes have been read: the ring's LDS may be written)
                            // what lies right of a row's window (neutral terms), and the score selectors of every column pair
                            for (uint32_t i = tid; i < brs * BAND_PAD * 2u * np; i += NT) {
                                const uint32_t w = i % (2u * np), en = (i / (2u * np)) % BAND_PAD, sl = i / (2u * np * BAND_PAD);
                                S.ring[(sl * slotb + (64u + en) * eb) / 4u + w] = w < np ? 0u : 0xFFFEFFFEu;      // A = 0, B = g - e
                            }
                            for (uint32_t i = tid; i < (Lpb / cplb + 66u) * np; i += NT) {
                                const uint32_t c = 2u * i;
                                const uint32_t ca = c < L ? S.sq[c] : 0u, cb = c + 1u < L ? S.sq[c + 1u] : 0u;
                                const uint32_t ia = (ca >> 1) & 3u, ib = (cb >> 1) & 3u;
                                const uint32_t sa = ca ? (ia | ((ia + 4u) << 8)) : 0x0D0Du, sb = cb ? (ib | ((ib + 4u) << 8)) : 0x0D0Du;
                                S.ring[sel_off / 4u + i] = sa | (sb << 16);
                            }
                            __syncthreads();
                            if (s_bc[1]) break;                          // (uniform) the columns do not order this graph's edges: no band
                            if (w0) {
                                int32_t bb = 0;
                                if (bsh == 1u) dp_rows_band<2>(S, X, nu, brs, sel_off, junk_off, bb);
                                else if (bsh == 2u) dp_rows_band<4>(S, X, nu, brs, sel_off, junk_off, bb);
                                else dp_rows_band<8>(S, X, nu, brs, sel_off, junk_off, bb);
                            }
                            __syncthreads();
                            const int32_t bb = __builtin_amdgcn_readfirstlane(X.best[0]);
                            cells_done += (unsigned long long)n * 64u * cplb;
                            if (bb >= 5 * (int32_t)(Lu - t) - 4) {      // the certificate: nothing outside the band can reach this score
                                band_on = true; bd_sh = bsh;
                                best = bb; best_row = 0; multi = true;
                                bd_thist = (5u * Lu - (uint32_t)bb) / 5u;
                                ++n_band_ok;
                            } else {
                                ++n_band_fail;
                                t_want = bb > 0 ? (5u * Lu - (uint32_t)bb) / 5u : Lu;      // the smallest t whose certificate this score passes
                                // the flag bytes are gone (the ring's LDS was used): the next, wider attempt needs only col[], which is in memory
                            }
                            __syncthreads();
                        }
                        // ---- 3b. no band: the full rows as vertical strips of 512 columns on the same row loop (dp_rows_band<8, true>), exact without a
                        // certificate.  Only for a pack three quarters of whose alignments so far HAD a band (a pack of noisy reads belongs to the full-row kernels,
                        // which spread a row over four to sixteen wavefronts); A.debug bit 3: always (tests)
                        const bool strips_ok = (A.debug & 8u) || (n_band_ok >= 8u && 4u * n_band_ok >= 3u * (q - q0));
                        const uint32_t n_strips = (Lu + 511u) >> 9;
                        if (!band_on && strips_ok && !s_bc[1] && (uint64_t)n_strips * (nu + 1u) * 512u <= A.cell_cap) {
                            const uint32_t Lpb = (Lu + 7u) & ~7u;
                            uint32_t brs = 16;
                            while (brs >= 2u && band_lds_bytes(brs, 8, Lpb) > rbu) brs >>= 1;
                            if (brs >= 2u) {
                                const uint32_t eb = 32u, slotb = (64u + BAND_PAD) * eb;
                                const uint32_t sel_off = brs * slotb, junk_off = sel_off + (Lpb / 8u + 66u) * 16u;
                                for (uint32_t r = tid; r < n; r += NT) {
                                    const uint32_t row = r + 1u;
                                    const uint4 pl = S.plan[r], pb = S.planb[r], pc = S.planc[r];
                                    const uint32_t n_in = rd_nin(pl.x);
                                    uint32_t po[8], farmask = 0, base = 0xFFFFFFFFu;
#pragma unroll
                                    for (uint32_t k = 0; k < 8; ++k) {
                                        po[k] = 0xFFFFFFFFu;
                                        if (k < n_in) {
                                            const uint32_t prow = k < 4 ? u4_get(pb, k) : u4_get(pc, k - 4);
                                            if (row - prow <= brs) { po[k] = (prow & (brs - 1u)) * slotb; if (base == 0xFFFFFFFFu) base = po[k]; }
                                            else farmask |= 1u << k;
                                        }
                                    }
                                    const bool nobase = base == 0xFFFFFFFFu;
                                    if (nobase) base = (row & (brs - 1u)) * slotb;
#pragma unroll
                                    for (uint32_t k = 0; k < 8; ++k) if (po[k] == 0xFFFFFFFFu) po[k] = base;
                                    const uint32_t letter = rd_letter(pl.x), li = (letter >> 1) & 3u;
                                    const uint32_t klo = 0xFCFCFCFCu ^ (0xF9u << (8u * li)), khi = 0xFFFFFFFFu ^ (0xFFu << (8u * li));
                                    uint32_t ctl = (n_in > 4 ? BD_MORE4 : 0u) | ((farmask || n_in > 8 || nobase) ? BD_SLOW : 0u) | (nobase ? BD_NOBASE : 0u);
                                    if (n_in == 1u && row > 1u && pb.x == row - 1u && !farmask) ctl |= BD_CHAIN;
                                    uint4 *pm = (uint4 *)(S.planm + 8 * (size_t)r);
                                    pm[0] = make_uint4(klo, khi, po[0] | po[1] << 16, po[2] | po[3] << 16);
                                    pm[1] = make_uint4((row & (brs - 1u)) * slotb, farmask << 8 | ctl << 16 | min(n_in, 255u) << 24, po[4] | po[5] << 16, po[6] | po[7] << 16);
                                }
                                for (uint32_t i = tid; i < (Lpb / 8u + 66u) * 4u; i += NT) {
                                    const uint32_t c = 2u * i;
                                    const uint32_t ca = c < L ? S.sq[c] : 0u, cb = c + 1u < L ? S.sq[c + 1u] : 0u;
                                    const uint32_t ia = (ca >> 1) & 3u, ib = (cb >> 1) & 3u;
                                    const uint32_t sa = ca ? (ia | ((ia + 4u) << 8)) : 0x0D0Du, sb = cb ? (ib | ((ib + 4u) << 8)) : 0x0D0Du;
                                    S.ring[sel_off / 4u + i] = sa | (sb << 16);
                                }
                                __syncthreads();
                                s16x2 mx = pk_splat(0);
                                for (uint32_t k = 0; k < n_strips; ++k) {
                                    if (w0) { int32_t bb = 0; dp_rows_band<8, true>(S, X, nu, brs, sel_off, junk_off, bb, k, mx, &mx); }
                                    __syncthreads();
                                }
                                band_on = true; bd_sh = 3; bd_strip = nu + 1u;
                                best = __builtin_amdgcn_readfirstlane(X.best[0]); best_row = 0; multi = true;
                                cells_done += (unsigned long long)n * 512u * n_strips;
                                ++n_strip_aln;
                            }
                        }
                    }
                }
                if constexpr (PK == 8) {
                    if (!band_on) {
                        // (diagnosis for RATTLE_TIMING: which alignment lost the band, and by how much)
                        if (tid == 0) { A.counters[14] = (unsigned long long)n | ((unsigned long long)L << 32); A.counters[15] = (unsigned long long)(q - q0) | ((unsigned long long)X.best[0] << 32); }
                        S.err = POA_ERR_BAND; break;
                    }
                }
                else if constexpr (PK == 2 && RING > 0) dp_rows_longr<CPL, RING, NW>(S, X, s, n, L, Lp, best, best_row, multi);
                else if constexpr (PK == 2) dp_rows_long<CPL, NW>(S, X, s, n, L, Lp, best, best_row, multi);
                else if constexpr (PK == 3) dp_rows_wide<CPL, RING, NW>(S, X, n, L, Lp, best, best_row, multi);
                else if constexpr (PK == 7) dp_rows_mt<CPL, NW, RING>(S, X, A, n, L, Lp, best, best_row, multi);
                else if constexpr (PK == 1) dp_rows_v3<CPL, RING, NW>(S, X, n, L, Lp, best, best_row, multi);
                else dp_rows<CPL, RING, NW>(S, X, n, L, Lp, best, best_row, multi);
                cells += (unsigned long long)n * L;
                if (PK != 8) cells_done += (unsigned long long)n * L;
                rows += n;
                unsigned long long t2 = PT_NOW();
                t_dp += t2 - t1;
                if (best > 0) {
                    // spoa takes the first maximum in ITS rank order.  dp_rows found the first row (block order)
                    // that reaches the best score; only if other rows may reach it too are the rows compared:
                    // rowmax[r] != 0 marks the rows that do (the H cells of the threads that saw the score).
                    if (tid == 0) { s_bc[5] = 0; s_bc[7] = 0xFFFFFFFFu; }
                    if (multi) {
                        for (uint32_t r = 1 + tid; r <= n; r += NT) S.rowmax[r] = 0;
                        __syncthreads();
                        const uint32_t ntl = X.ntl;
                        if (PK == 8 && band_on) {
                            // the band record: 64 << bd_sh cells per row; lane t of a row holds its window's column group t
                            const uint32_t cw = 1u << bd_sh;
                            const uint16_t *Hh = (const uint16_t *)S.H;
                            const uint32_t ns = bd_strip ? (L + 511u) >> 9 : 1u;      // (strips: a lane holds a column group in every strip)
                            if (ntl <= 16) {
                                for (uint32_t idx = tid; idx < n * ntl; idx += NT) {
                                    const uint32_t r = idx / ntl + 1, t = X.tl[idx % ntl];
                                    bool hit = false;
                                    for (uint32_t k = 0; k < ns; ++k)
                                        for (uint32_t u = 0; u < cw; ++u) hit |= (int32_t)(Hh[((((uint64_t)k * bd_strip + r) * 64u + t) << bd_sh) + u] & 0x3FFF) == best;
                                    if (hit) { S.rowmax[r] = 1; atomicMin(&X.brow, r); }
                                }
                            } else {
                                for (uint32_t r = 1 + (uint32_t)(tid >> 6); r <= n; r += NT / 64) {
                                    bool hit = false;
                                    for (uint32_t k = 0; k < ns; ++k)
                                        for (uint32_t c = tid & 63; c < 64u * cw; c += 64) hit |= (int32_t)(Hh[((((uint64_t)k * bd_strip + r) * 64u) << bd_sh) + c] & 0x3FFF) == best;
                                    if (hit) { S.rowmax[r] = 1; atomicMin(&X.brow, r); }
                                }
                            }
                        } else if (PK != 2 && ntl <= 16) {
                            for (uint32_t idx = tid; idx < n * ntl; idx += NT) {
                                const uint32_t r = idx / ntl + 1, t = X.tl[idx % ntl];
                                int32_t v[CPL];
                                load_block<CPL>(S.H + (uint64_t)r * Lp + t * CPL, v);
                                bool hit = false;
#pragma unroll
                                for (int u = 0; u < CPL; ++u) hit |= (pk_packed(PK) ? (v[u] & 0x3FFF) : v[u]) == best;
                                if (hit) { S.rowmax[r] = 1; if (pk_packed(PK) || PK == 3) atomicMin(&X.brow, r); }
                            }
                        } else {
                            for (uint32_t r = 1 + (uint32_t)(tid >> 6); r <= n; r += NT / 64) {
                                const cell_t *Hr = (const cell_t *)S.H + (uint64_t)r * Lp;
                                bool hit = false;
                                for (uint32_t c = tid & 63; c < L; c += 64) hit |= (pk_packed(PK) ? (int32_t)(Hr[c] & 0x3FFF) : (int32_t)Hr[c]) == best;
                                if (hit) { S.rowmax[r] = 1; if (pk_packed(PK) || PK == 3) atomicMin(&X.brow, r); }
                            }
                        }
                        __syncthreads();
                        uint32_t cnt = 0;
                        for (uint32_t r = 1 + tid; r <= n; r += NT) cnt += S.rowmax[r] != 0 ? 1u : 0u;
                        if (cnt) atomicAdd(&s_bc[5], cnt);
                    }
                    __syncthreads();
                    if (pk_packed(PK) || PK == 3) best_row = X.brow;  // packed and wide rows: the rescan is where the first row comes from
                    // cheap exit: if every tied row except the first has a tied direct predecessor, all of
                    // them descend from the first one, which then precedes them in ANY topological order
                    bool need_sort = s_bc[5] > 1;
                    __syncthreads();                  // every thread has read the count before the slot is reused
                    if (need_sort) {
                        if (tid == 0) s_bc[5] = 0;
                        __syncthreads();
                        for (uint32_t r = 1 + tid; r <= n; r += NT) {
                            if (S.rowmax[r] == 0 || r == best_row) continue;
                            const uint4 pl = S.plan[r - 1], plb = S.planb[r - 1];
                            const uint32_t n_in = rd_nin(pl.x);
                            bool ok = false;
                            for (uint32_t k = 0; k < n_in && k < 4; ++k) {
                                const uint32_t p = u4_get(plb, k);
                                if (p == 0) continue;
                                if (S.rowmax[p] != 0) ok = true;
                            }
                            if (!ok) s_bc[5] = 1;
                        }
                        __syncthreads();
                        need_sort = s_bc[5] != 0;
                        __syncthreads();              // ... and again before s_bc[5] becomes the best-column slot
#ifdef POA_PROFILE
                        if (tid == 0) { atomicAdd(&A.counters[4], 1ull << 32); if (need_sort) atomicAdd(&A.counters[4], 1ull); }
#endif
                    }
                    if (need_sort && RING > 0 && 2u * n <= ring_bytes && S.n_nodes <= 0xFFFFu && !(A.debug & 1u)) {
                        // labels instead of the full sort (tie_labels): 16 bits per row in the ring's LDS, which holds nothing between two DPs
                        uint16_t *lab = (uint16_t *)S.ring;
                        for (uint32_t r = tid; r < n; r += NT) lab[r] = (uint16_t)S.order[r];
                        if (tid == 0) { s_bc[7] = 0xFFFFFFFFu; s_bc[5] = 0; s_bc[6] = 0xFFFFFFFFu; }
                        __syncthreads();
                        for (uint32_t r = 1 + tid; r <= n; r += NT) {
                            if (S.rowmax[r] != 0) atomicMin(&s_bc[6], r);
                        }
                        __syncthreads();
                        if (w0) { const uint32_t next = tie_labels(S, lab, n, s_bc[6]); if (tid == 0) s_bc[6] = next; }
                        __syncthreads();
                        for (uint32_t r = 1 + tid; r <= n; r += NT) {
                            if (S.rowmax[r] != 0) atomicMin(&s_bc[7], (uint32_t)lab[r - 1]);
                        }
                        __syncthreads();
                        const uint32_t root = s_bc[7];
                        for (uint32_t r = 1 + tid; r <= n; r += NT) {
                            if (S.rowmax[r] != 0 && (uint32_t)lab[r - 1] == root) {
                                const uint32_t slot = atomicAdd(&s_bc[5], 1u);
                                if (slot < 8) s_tied[slot] = r;
                            }
                        }
                        __syncthreads();
                        const uint32_t n_tied = s_bc[5];
                        __syncthreads();
                        if (n_tied == 1) { best_row = s_tied[0]; need_sort = false; }
                        else if (n_tied <= 8) {
                            // several tied rows inside one root's DFS: finish the labels, replay only that DFS
                            for (uint32_t t = tid; t < (S.n_nodes + 31) / 32; t += NT) { S.done[t] = 0; S.nocheck[t] = 0; }
                            if (w0 && s_bc[6] >= 1) tie_labels(S, lab, s_bc[6], 0);
                            __syncthreads();
                            for (uint32_t r = tid; r < n; r += NT)
                                if ((uint32_t)lab[r] < root) { const uint32_t u = S.order[r]; atomicOr(&S.done[u >> 5], 1u << (u & 31)); }
                            __syncthreads();
                            if (w0) {
                                uint32_t n_emit, hit = 0;
                                toposort(S, A, 3, n_emit, hit, root, s_tied, n_tied);
                                if (!S.err && hit == 0) S.err = POA_ERR_GRAPH;
                                if (tid == 0) { s_bc[4] = S.err; s_bc[7] = hit; }
                            }
                            __syncthreads();
                            S.err = s_bc[4];
                            if (S.err) break;
                            best_row = s_bc[7];
                            need_sort = false;
                            __syncthreads();
                        }
                    }
                    if (need_sort) {
                        if (tid == 0) s_bc[7] = 0xFFFFFFFFu;
                        __syncthreads();
                        if (w0) {
                            uint32_t n_emit, n_cols;
                            toposort(S, A, 2, n_emit, n_cols);
                            if (!S.err && n_emit != n) S.err = POA_ERR_GRAPH;
                            if (tid == 0) s_bc[4] = S.err;
                        }
                        __syncthreads();
                        S.err = s_bc[4];
                        if (S.err) break;
                        for (uint32_t r = 1 + tid; r <= n; r += NT) {
                            if (S.rowmax[r] != 0) atomicMin(&s_bc[7], (uint32_t)S.srank[S.order[r - 1]]);
                        }
                        __syncthreads();
                        const uint32_t want = s_bc[7];
                        __syncthreads();
                        for (uint32_t r = 1 + tid; r <= n; r += NT) {
                            if (S.rowmax[r] != 0 && (uint32_t)S.srank[S.order[r - 1]] == want) s_bc[7] = r;
                        }
                        __syncthreads();
                        best_row = s_bc[7];
                    }
                    // ---- 4. best cell + traceback ----
#ifdef POA_PROFILE
                    t_tie += PT_NOW() - t2;
#endif
                    const cell_t *Hb = (const cell_t *)S.H + (uint64_t)best_row * Lp;
                    if (tid == 0) s_bc[5] = 0xFFFFFFFFu;
                    __syncthreads();
                    if (PK == 8 && band_on && bd_strip) {
                        const uint16_t *Hs = (const uint16_t *)S.H;
                        for (uint32_t c = tid; c < L; c += NT)
                            if ((int32_t)(Hs[(((uint64_t)(c >> 9) * bd_strip + best_row) * 64u << 3) + (c & 511u)] & 0x3FFF) == best) { atomicMin(&s_bc[5], c + 1); break; }
                    } else if (PK == 8 && band_on) {
                        const uint16_t *Hw = (const uint16_t *)S.H + (((uint64_t)best_row * 64u) << bd_sh);
                        const uint32_t cfirst = (S.planm[8 * (size_t)(best_row - 1) + 4] >> 16) << bd_sh;        // 0-based column of the window's first cell (the row's record carries its first group)
                        for (uint32_t c = tid; c < (64u << bd_sh); c += NT) if (cfirst + c < L && (int32_t)(Hw[c] & 0x3FFF) == best) { atomicMin(&s_bc[5], cfirst + c + 1); break; }
                    } else
                    for (uint32_t c = tid; c < L; c += NT) if ((pk_packed(PK) ? (int32_t)(Hb[c] & 0x3FFF) : (int32_t)Hb[c]) == best) { atomicMin(&s_bc[5], c + 1); break; }
                    __syncthreads();
                    const uint32_t bj = s_bc[5];
                    // Traceback by wave 0, in spoa's order of tests: diagonal (predecessors in in-edge order),
                    // vertical (F extension before opening), horizontal.  The common move -- diagonal to the FIRST
                    // predecessor -- is verified 16 steps at a time: the chain of first predecessors and the row
                    // letters sit in LDS (areas idle here), lane t fetches the H cell of step t+1, and one
                    // ballot tells how many consecutive steps hold; the first step that does not is replayed
                    // by the general code (all lanes uniformly, lane 0 writes).  aln[] receives (row | -1, pos | -1);
                    // rows become node ids in add_alignment.
                    uint16_t *tq0 = (uint16_t *)S.done;                   // [n+1] row of the first in-edge (0: none / virtual)
                    const uint32_t tb_bytes = (2u * poa_bit_words(A.node_cap) + POA_STACK) * 4u + ring_bytes;
                    const bool fast_tb = 3u * (n + 2u) <= tb_bytes && n < 0xFFFFu && !(A.debug & 2u);
                    // Round 5: jump tables over the chain of first predecessors (4 and 16 steps at once) where the LDS has room for them:
                    // lane h reaches the row of ITS step in at most 3 + 3 (+ 3) dependent LDS reads instead of h, and with all three
                    // tables the whole wavefront verifies 64 steps per round trip instead of 16 (the chain walk was sixteen dependent
                    // LDS reads per batch, by every lane).  levels: 1 = the chain alone (as before), 2 = + 4 steps, 3 = + 16 steps.
                    const uint32_t tq_levels = !fast_tb || (A.debug & 4u) ? 1u : 7u * (n + 2u) <= tb_bytes ? 3u : 5u * (n + 2u) <= tb_bytes ? 2u : 1u;
                    uint16_t *tq4 = tq0 + (n + 2u), *tq16 = tq4 + (n + 2u);
                    uint8_t *tlet = (uint8_t *)(tq0 + tq_levels * (n + 2u));          // [n+1] letter of the row's node
                    if (fast_tb) {
                        for (uint32_t r = 1 + tid; r <= n; r += NT) {
                            const uint32_t info = S.plan[r - 1].x;
                            tq0[r] = (uint16_t)(rd_nin(info) ? S.planb[r - 1].x : 0u);
                            tlet[r] = (uint8_t)rd_letter(info);
                        }
                        if (tid == 0) { tq0[0] = 0; tlet[0] = 0; }
                        __syncthreads();
                        if (tq_levels >= 2) {
                            for (uint32_t r = tid; r <= n; r += NT) tq4[r] = tq0[tq0[tq0[tq0[r]]]];
                            __syncthreads();
                        }
                        if (tq_levels >= 3) {
                            for (uint32_t r = tid; r <= n; r += NT) tq16[r] = tq4[tq4[tq4[tq4[r]]]];
                            __syncthreads();
                        }
                    }
                    if (w0) {
                        const uint32_t lane = (uint32_t)tid;
                        uint32_t i = best_row, j = bj, cnt = 0, err = 0;
                        const cell_t *H = (const cell_t *)S.H;
                        // packed classes (PK == 1): the H word carries min(H - F, 3) in its two top bits and min(H - E, 3)
                        // sits in the per-thread bit array behind S.E -- exact for every test below (dp_rows_v3)
                        // 32-bit register classes (PK == 0): H int16 plus a nibble per column behind S.E (dp_rows)
                        auto nib = [&](uint32_t r, uint32_t c) -> uint32_t {
                            if (PK == 2) return ((uint32_t)((const uint16_t *)S.E)[((uint64_t)r * Lp + c - 1) >> 2] >> (4 * ((c - 1) & 3u))) & 0xFu;   // segmented rows: per column
                            constexpr uint32_t NWD = (CPL + 7) / 8;
                            const uint32_t t = (c - 1) / CPL, k = (c - 1) % CPL;
                            return (((const uint32_t *)S.E)[((uint64_t)r * NT + t) * NWD + k / 8] >> (4 * (k % 8))) & 0xFu;
                        };
                        // the record word of row r >= 1, column c >= 1.  Band rows (PK == 8): 64 << bd_sh cells per row, the window's; a cell
                        // outside the window reads H = 0, H - F = 3 (F = E = -inf there)
                        auto recw = [&](uint32_t r, uint32_t c) -> int32_t {
                            if constexpr (PK == 8) {
                                if (band_on && bd_strip)       // strip-major full rows: 512 columns per strip, bd_strip rows per strip
                                    return (int32_t)((const uint16_t *)S.H)[(((uint64_t)((c - 1u) >> 9) * bd_strip + r) * 64u << 3) + ((c - 1u) & 511u)];
                                if (band_on) {
                                    const uint32_t g = ((c - 1u) >> bd_sh) - (S.planm[8 * (size_t)(r - 1u) + 4] >> 16);      // (unsigned: a column left of the window wraps)
                                    if (g >= 64u) return 0xC000;
                                    return (int32_t)((const uint16_t *)S.H)[((((uint64_t)r * 64u + g)) << bd_sh) + ((c - 1u) & ((1u << bd_sh) - 1u))];
                                }
                            }
                            return (int32_t)H[(uint64_t)r * Lp + c - 1];
                        };
                        auto Hat = [&](uint32_t r, uint32_t c) -> int32_t {
                            if (r == 0 || c == 0) return 0;
                            if constexpr (PK == 8) { return recw(r, c) & 0x3FFF; }
                            else {
                            const int32_t w = (int32_t)H[(uint64_t)r * Lp + c - 1];
                            return pk_packed(PK) ? (w & 0x3FFF) : w;
                            }
                        };
                        auto Fat = [&](uint32_t r, uint32_t c) -> int32_t {
                            if (r == 0 || c == 0) return POA_NEG;
                            if constexpr (PK == 8) { const int32_t w = recw(r, c); return (w & 0x3FFF) - ((w >> 14) & 3); }
                            if (pk_packed(PK)) { const int32_t w = (int32_t)H[(uint64_t)r * Lp + c - 1]; return (w & 0x3FFF) - ((w >> 14) & 3); }
                            return (int32_t)H[(uint64_t)r * Lp + c - 1] - (int32_t)(nib(r, c) & 3u);
                        };
                        auto Eat = [&](uint32_t r, uint32_t c) -> int32_t {
                            if (r == 0 || c == 0) return POA_NEG;
                            if (pk_packed(PK)) {
                                // The packed rows keep no E record: E[r][c] = max over k < c of H[r][k] + g + (c-1-k) e (H[r][0] = 0) is
                                // rebuilt from the row's H cells by the whole wavefront when the traceback asks for it -- only at
                                // horizontal moves, a few dozen times per alignment, against a store per thread and row in the DP.
                                int32_t m = POA_NEG;
                                if constexpr (PK == 8) {
                                    if (band_on) {
                                        // (band rows: the cells of the window left of c, and the column left of the window with H = 0 in the place of column 0)
                                        const uint32_t k0 = bd_strip ? 0u : (S.planm[8 * (size_t)(r - 1u) + 4] >> 16) << bd_sh;
                                        for (uint32_t k = k0 + lane; k < c; k += 64) {
                                            const int32_t h = k == k0 ? 0 : (recw(r, k) & 0x3FFF);
                                            m = max(m, h - (int32_t)k * POA_E);
                                        }
                                        return wave_last(wave_scan_max(m, POA_NEG)) + POA_G + ((int32_t)c - 1) * POA_E;
                                    }
                                }
                                const uint16_t *Hr = (const uint16_t *)S.H + (uint64_t)r * Lp;
                                for (uint32_t k = lane; k < c; k += 64) {
                                    const int32_t h = k == 0 ? 0 : (int32_t)(Hr[k - 1] & 0x3FFFu);
                                    m = max(m, h - (int32_t)k * POA_E);
                                }
                                return wave_last(wave_scan_max(m, POA_NEG)) + POA_G + ((int32_t)c - 1) * POA_E;
                            }
                            return (int32_t)H[(uint64_t)r * Lp + c - 1] - (int32_t)((nib(r, c) >> 2) & 3u);
                        };
                        auto put = [&](int32_t row, int32_t pos) {
                            if (cnt >= A.aln_cap) { err = POA_ERR_ALN; return; }
                            if (lane == 0) { S.aln[2 * cnt] = row; S.aln[2 * cnt + 1] = pos; }
                            ++cnt;
                        };
                        int32_t Hij = best;
                        while (!err && Hij != 0) {                      // H != 0 implies i != 0 and j != 0
                            uint4 fpl = make_uint4(0, 0, 0, 0), fplb = make_uint4(0, 0, 0, 0);
                            int pf_lane = -1;                              // lane holding the plan of row i, or -1
                            if (fast_tb) {
                                const uint32_t K = tq_levels == 3 ? 64u : 16u;
                                uint32_t my_i = 0, my_next = 0, cur = i;
                                if (tq_levels == 1) {
#pragma unroll
                                    for (uint32_t h = 0; h < 16; ++h) {
                                        const uint32_t nx = tq0[cur];
                                        if (lane == h) { my_i = cur; my_next = nx; }
                                        cur = nx;
                                    }
                                } else {
                                    // lane h = 16 a + 4 b + c: a jumps of sixteen, b of four, c single steps
                                    const uint32_t ja = lane >> 4, jb = (lane >> 2) & 3u, jc = lane & 3u;
                                    if (tq_levels == 3) {
#pragma unroll
                                        for (uint32_t q = 0; q < 3; ++q) if (q < ja) cur = tq16[cur];
                                    }
#pragma unroll
                                    for (uint32_t q = 0; q < 3; ++q) if (q < jb) cur = tq4[cur];
#pragma unroll
                                    for (uint32_t q = 0; q < 3; ++q) if (q < jc) cur = tq0[cur];
                                    my_i = cur; my_next = tq0[cur];
                                }
                                const bool in = lane < K && j > lane;       // column of my step: j - lane >= 1
                                const uint32_t my_j = in ? j - lane : 1u;
                                int32_t c = 0;
                                // the row plan of my step travels with the cell: the step that fails is replayed by the general
                                // code below, which then has its plan in lane m's registers instead of a second round trip
                                if (in && my_i != 0) { fpl = S.plan[my_i - 1]; fplb = S.planb[my_i - 1]; }
                                if constexpr (PK == 8) { if (in && my_next != 0 && my_j > 1) c = recw(my_next, my_j - 1) & 0x3FFF; }
                                else
                                if (in && my_next != 0 && my_j > 1) { c = (int32_t)H[(uint64_t)my_next * Lp + my_j - 2]; if (pk_packed(PK)) c &= 0x3FFF; }
                                const int32_t hcur = wave_shr1(c, Hij);     // H of my step's own cell = the cell lane-1 fetched
                                const int32_t mc = tlet[my_i] == (PK == 2 ? s[my_j - 1] : S.sq[my_j - 1]) ? POA_M : POA_N;
                                const bool ok = in && my_i != 0 && hcur != 0 && hcur == c + mc;
                                const unsigned long long okm = __ballot(ok);
                                const uint32_t m = ~okm ? (uint32_t)__builtin_ctzll(~okm) : 64u;         // consecutive verified steps
                                if (m) {
                                    if (cnt + m > A.aln_cap) { err = POA_ERR_ALN; break; }
                                    if (lane < m) { S.aln[2 * (cnt + lane)] = (int32_t)my_i; S.aln[2 * (cnt + lane) + 1] = (int32_t)(my_j - 1); }
                                    cnt += m;
                                    i = (uint32_t)__builtin_amdgcn_readlane((int)my_next, (int)(m - 1));
                                    Hij = __builtin_amdgcn_readlane(c, (int)(m - 1));
                                    j -= m;
                                    if (m == K || Hij == 0) continue;
                                }
                                pf_lane = (int)m;                          // lane m < K walked to row i at column j >= 1 (H != 0): its plan is loaded
                            }
                            // general step
                            bool found = false, ext_left = false, ext_up = false;
                            uint32_t pi = 0, pj = 0;
                            int32_t Hn = 0;
                            uint4 pl, plb;
                            if (pf_lane >= 0) {
                                pl.x = (uint32_t)__builtin_amdgcn_readlane((int)fpl.x, pf_lane); pl.y = (uint32_t)__builtin_amdgcn_readlane((int)fpl.y, pf_lane);
                                pl.z = (uint32_t)__builtin_amdgcn_readlane((int)fpl.z, pf_lane); pl.w = (uint32_t)__builtin_amdgcn_readlane((int)fpl.w, pf_lane);
                                plb.x = (uint32_t)__builtin_amdgcn_readlane((int)fplb.x, pf_lane); plb.y = (uint32_t)__builtin_amdgcn_readlane((int)fplb.y, pf_lane);
                                plb.z = (uint32_t)__builtin_amdgcn_readlane((int)fplb.z, pf_lane); plb.w = (uint32_t)__builtin_amdgcn_readlane((int)fplb.w, pf_lane);
                            } else { pl = S.plan[i - 1]; plb = S.planb[i - 1]; }
                            const uint32_t n_in = rd_nin(pl.x);
                            const uint32_t npred = n_in ? n_in : 1u;
                            // Round 5: the record words every test of this step may need -- the cells above (vertical moves, first four
                            // predecessors) and the cell to the left -- are requested TOGETHER with the diagonal candidates: a gap used to
                            // cost a memory round trip per predecessor tried, one after the other, behind the diagonal's own
                            int32_t wv[4] = {-1, -1, -1, -1}, wh = -1;          // -1: no such cell (virtual row / column 0)
                            {
                                const int32_t mc = rd_letter(pl.x) == (PK == 2 ? s[j - 1] : S.sq[j - 1]) ? POA_M : POA_N;
                                const uint32_t q0 = n_in ? plb.x : 0u, q1 = npred > 1 ? plb.y : q0, q2 = npred > 2 ? plb.z : q0, q3 = npred > 3 ? plb.w : q0;
                                if constexpr (pk_packed(PK)) {
                                    const uint32_t qq[4] = {q0, q1, q2, q3};
                                    if constexpr (PK == 8) {
#pragma unroll
                                        for (int k = 0; k < 4; ++k) if (qq[k] != 0 && (uint32_t)k < npred) wv[k] = recw(qq[k], j);
                                        if (j > 1) wh = recw(i, j - 1);
                                    } else {
#pragma unroll
                                    for (int k = 0; k < 4; ++k) if (qq[k] != 0 && (uint32_t)k < npred) wv[k] = (int32_t)H[(uint64_t)qq[k] * Lp + j - 1];
                                    if (j > 1) wh = (int32_t)H[(uint64_t)i * Lp + j - 2];
                                    }
                                }
                                const int32_t c0 = Hat(q0, j - 1), c1 = Hat(q1, j - 1), c2 = Hat(q2, j - 1), c3 = Hat(q3, j - 1);
                                if (Hij == c0 + mc) { pi = q0; Hn = c0; found = true; }
                                else if (npred > 1 && Hij == c1 + mc) { pi = q1; Hn = c1; found = true; }
                                else if (npred > 2 && Hij == c2 + mc) { pi = q2; Hn = c2; found = true; }
                                else if (npred > 3 && Hij == c3 + mc) { pi = q3; Hn = c3; found = true; }
                                else if (npred > 4) {
                                    uint32_t e = pl.z;
                                    for (uint32_t k = 4; k < npred; ++k) {
                                        const uint2 ed = S.edges[e]; e = ed.y;
                                        const uint32_t p = (uint32_t)S.rank[ed.x] + 1;
                                        const int32_t c = Hat(p, j - 1);
                                        if (Hij == c + mc) { pi = p; Hn = c; found = true; break; }
                                    }
                                }
                                if (found) pj = j - 1;
                            }
                            const bool diag = found;
                            if (!found) {
                                uint32_t e = pl.z;
                                for (uint32_t k = 0; k < npred; ++k) {
                                    uint32_t p = 0;
                                    if (n_in) { if (k < 4) p = u4_get(plb, k); else { const uint2 ed = S.edges[e]; e = ed.y; p = (uint32_t)S.rank[ed.x] + 1; } }
                                    int32_t Fp, Hp;
                                    if (pk_packed(PK) && k < 4) {            // from the words requested above
                                        const int32_t w = wv[k];
                                        Hp = w == -1 ? 0 : (w & 0x3FFF);
                                        Fp = w == -1 ? POA_NEG : (w & 0x3FFF) - ((w >> 14) & 3);
                                    } else { Fp = Fat(p, j); Hp = Hat(p, j); }
                                    if ((ext_up = (Hij == Fp + POA_E)) || Hij == Hp + POA_G) { pi = p; pj = j; found = true; break; }
                                }
                            }
                            if (!found) {
                                const int32_t Hl = pk_packed(PK) ? (wh == -1 ? 0 : (wh & 0x3FFF)) : Hat(i, j - 1);
                                if ((ext_left = (Hij == Eat(i, j - 1) + POA_E)) || Hij == Hl + POA_G) { pi = i; pj = j - 1; found = true; }
                            }
                            if (!found) { err = POA_ERR_GRAPH; break; }
                            put(i == pi ? -1 : (int32_t)i, j == pj ? -1 : (int32_t)(j - 1));
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
                                    const uint4 ul = S.plan[i - 1], ulb = S.planb[i - 1];
                                    const uint32_t uin = rd_nin(ul.x);
                                    uint32_t e = ul.z;
                                    const int32_t Fij = Fat(i, j);
                                    for (uint32_t k = 0; k < uin; ++k) {
                                        uint32_t p;
                                        if (k < 4) p = u4_get(ulb, k); else { const uint2 ed = S.edges[e]; e = ed.y; p = (uint32_t)S.rank[ed.x] + 1; }
                                        if ((stop = (Fij == Hat(p, j) + POA_G)) || Fij == Fat(p, j) + POA_E) { np = p; break; }
                                    }
                                    put((int32_t)i, -1);
                                    i = np;
                                    if (stop || i == 0) break;
                                }
                            }
                            if (diag) Hij = Hn; else Hij = Hat(i, j);
                        }
                        if (lane == 0) { s_bc[0] = cnt; s_bc[1] = err; }
                    }
                    __syncthreads();
                    n_aln = s_bc[0];
                    if (s_bc[1]) { S.err = s_bc[1]; break; }
                }
                t_tb += PT_NOW() - t2;
            }
            // ---- 5. add_alignment ----
            // (a) all threads resolve the pairs against the graph as it is (reuse / sibling / new node,
            //     row range of the aligned group); (b) thread 0 creates the nodes in spoa's id order,
            //     updates aligned groups and records, for every NEW node, the row it must be inserted
            //     before (an unaligned run goes right before the group of the next graph node on the
            //     path, a mismatch node right after the group it joins, trailing nodes at the end;
            //     anchors are non-decreasing along the path); (c) all threads add the path edges
            //     (one per sequence position, distinct target nodes).
            unsigned long long t3 = PT_NOW();
            const uint32_t n_old = S.n_nodes;
            enum { K_SKIP = 0, K_INS = 1, K_SAME = 2, K_SIB = 4, K_NEW = 3 };      // (bit 0: the pair creates a node -- what the sequential pass cannot skip)
            // the kind of every pair also goes to LDS (bitmap / stack area, idle here) so that the sequential
            // pass below touches global memory only for the pairs that create nodes
            uint8_t *kind = (uint8_t *)S.done;
            const bool use_kind = n_aln <= (poa_bit_words(A.node_cap) * 2 + POA_STACK) * 4;
            for (uint32_t f = tid; f < n_aln; f += NT) {
                const int32_t ar = S.aln[2 * (n_aln - 1 - f)], pos = S.aln[2 * (n_aln - 1 - f) + 1];
                const int32_t an = ar < 0 ? -1 : (int32_t)S.order[ar - 1];      // traceback recorded rows
                uint4 inf = make_uint4(K_SKIP, 0, 0, 0);
                if (pos != -1) {
                    if (an == -1) {
                        inf.x = K_INS;
                    } else {
                        const uint8_t letter = s[pos];
                        const uint4 nd = S.nrec[an];
                        const uint32_t n_al = rd_nal(nd.x);
                        uint4 al = make_uint4(0, 0, 0, 0);
                        if (n_al) al = S.nal[an];
                        uint32_t gs = (uint32_t)S.rank[an], ge = gs;
                        int32_t hit = -1;
                        for (uint32_t k = 0; k < n_al; ++k) {
                            const uint32_t a = u4_get(al, k);
                            const uint32_t r = (uint32_t)S.rank[a];
                            gs = min(gs, r); ge = max(ge, r);
                            if (hit == -1 && rd_letter(S.nrec[a].x) == letter) hit = (int32_t)a;
                        }
                        if (rd_letter(nd.x) == letter) { inf.x = K_SAME; inf.y = (uint32_t)an; }
                        else if (hit != -1) { inf.x = K_SIB; inf.y = (uint32_t)hit; }
                        else { inf.x = K_NEW; inf.y = (uint32_t)an; }
                        inf.z = gs; inf.w = ge + 1;
                        if (inf.x != K_NEW) path[pos] = inf.y;         // a reused node: nothing else to do for this position
                    }
                }
                S.ainfo[f] = inf;
                if (use_kind) kind[f] = (uint8_t)inf.x;
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t T = 0;
                auto nn_push = [&](uint32_t node, uint32_t anchor) { S.nn[2 * T] = node; S.nn[2 * T + 1] = anchor; ++T; };
                uint32_t e_lo = 1, e_hi = 0;                           // sequence positions p that need the edge path[p-1] -> path[p]
                if (n_aln == 0) {
                    g_add_chain(S, A, s, 0, L, path);
                    if (!S.err) for (uint32_t i = 0; i < L; ++i) nn_push(n_old + i, n_old);
                } else {
                    // pairs are stored reversed: forward pair f = aln[n_aln-1-f]
                    int32_t first_valid = -1, last_valid = -1;
                    for (uint32_t f = 0; f < n_aln && first_valid == -1; ++f) first_valid = S.aln[2 * (n_aln - 1 - f) + 1];
                    for (uint32_t f = n_aln; f-- > 0 && last_valid == -1;) last_valid = S.aln[2 * (n_aln - 1 - f) + 1];
                    const uint32_t before = S.n_nodes;
                    g_add_chain(S, A, s, 0, (uint32_t)first_valid, path);
                    uint32_t pa0 = before, pan = S.n_nodes - before;        // pending: prefix chain
                    const uint32_t suf0 = S.n_nodes;
                    if (!S.err) g_add_chain(S, A, s, (uint32_t)last_valid + 1, L, path);
                    const uint32_t sufn = S.n_nodes - suf0;
                    uint32_t pb0 = 0, pbn = 0;                              // pending: current insertion run
                    for (uint32_t f = 0; f < n_aln && !S.err; ++f) {
                        if (use_kind) {
                            // sixteen pairs at a time while nothing is pending and none of them creates a node (round 6: this loop over ~1000 reused
                            // nodes, one LDS byte per trip, was a third of a near-chain alignment's time)
                            if (pan + pbn == 0 && (f & 15u) == 0 && f + 16u <= n_aln) {
                                const uint4 kk = *(const uint4 *)(kind + f);
                                if (((kk.x | kk.y | kk.z | kk.w) & 0x01010101u) == 0) { f += 15u; continue; }
                            }
                            const uint32_t k = kind[f];
                            if (k == K_SKIP) continue;
                            if ((k == K_SAME || k == K_SIB) && pan + pbn == 0) continue;
                        }
                        const uint4 inf = S.ainfo[f];
                        if (inf.x == K_SKIP) continue;
                        const int32_t pos = S.aln[2 * (n_aln - 1 - f) + 1];
                        const uint8_t letter = s[pos];
                        uint32_t cur;
                        if (inf.x == K_INS) {
                            cur = g_add_node(S, A, letter);
                            if (S.err) break;
                            if (pbn == 0) pb0 = cur;
                            ++pbn;
                        } else {
                            if (pan + pbn > 0) {
                                for (uint32_t i = 0; i < pan; ++i) nn_push(pa0 + i, inf.z);
                                for (uint32_t i = 0; i < pbn; ++i) nn_push(pb0 + i, inf.z);
                                pan = 0; pbn = 0;
                            }
                            if (inf.x != K_NEW) {
                                cur = inf.y;
                            } else {
                                const uint32_t an = inf.y;
                                uint4 nd = S.nrec[an];
                                const uint32_t n_al = rd_nal(nd.x);
                                uint4 al = make_uint4(0, 0, 0, 0);
                                if (n_al) al = S.nal[an];
                                if (n_al >= 4) { S.err = POA_ERR_GRAPH; break; }
                                cur = g_add_node(S, A, letter);
                                if (S.err) break;
                                nn_push(cur, inf.w);
                                uint4 cal = make_uint4(0, 0, 0, 0);
                                for (uint32_t k = 0; k < n_al; ++k) {
                                    const uint32_t a = u4_get(al, k);
                                    u4_set(cal, k, a);
                                    uint4 ar = S.nrec[a];
                                    const uint32_t c = rd_nal(ar.x);
                                    if (c >= 4) { S.err = POA_ERR_GRAPH; break; }
                                    uint4 aal = c ? S.nal[a] : make_uint4(0, 0, 0, 0);
                                    u4_set(aal, c, cur);
                                    S.nal[a] = aal;
                                    ar.x += 1u << 8;
                                    S.nrec[a] = ar;
                                }
                                u4_set(cal, n_al, an);
                                S.nal[cur] = cal;
                                S.nrec[cur] = make_uint4((uint32_t)letter | ((n_al + 1) << 8), 0, POA_NONE, POA_NONE);
                                u4_set(al, n_al, cur);
                                S.nal[an] = al;
                                nd.x += 1u << 8;
                                S.nrec[an] = nd;
                            }
                        }
                        if (S.err) break;
                        path[pos] = cur;
                    }
                    if (!S.err) {
                        for (uint32_t i = 0; i < pan; ++i) nn_push(pa0 + i, n_old);
                        for (uint32_t i = 0; i < pbn; ++i) nn_push(pb0 + i, n_old);
                        for (uint32_t i = 0; i < sufn; ++i) nn_push(suf0 + i, n_old);
                        e_lo = (uint32_t)std::max(first_valid, 1);
                        e_hi = std::min((uint32_t)last_valid + 1, L - 1);
                    }
                }
                if (!S.err && T != S.n_nodes - n_old) S.err = POA_ERR_GRAPH;
                s_bc[2] = S.n_nodes; s_bc[3] = S.n_edges; s_bc[4] = S.err; s_bc[5] = e_lo; s_bc[6] = e_hi;
            }
            __syncthreads();
            S.n_nodes = s_bc[2]; S.err = s_bc[4];
            if (!S.err) {
                const uint32_t e_lo = s_bc[5], e_hi = s_bc[6];
                for (uint32_t p = e_lo + tid; p <= e_hi && e_lo <= e_hi; p += NT) g_add_edge(S, A, path[p - 1], path[p], &s_bc[3], &s_bc[4]);
            }
            __syncthreads();
            S.n_edges = s_bc[3]; S.err = s_bc[4];
            __syncthreads();
            // ---- 6. merge_order: insert the new nodes into the row order (all threads) ----
            const unsigned long long t4 = PT_NOW();
            (void)t4;
            if (!S.err) {
                const uint32_t T = S.n_nodes - n_old;
                if (T > 0) {
                    const uint32_t first = S.nn[1];
                    for (uint32_t r = first + tid; r < n_old; r += NT) {
                        uint32_t lo = 0, hi = T;                     // number of anchors <= r
                        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (S.nn[2 * mid + 1] <= r) lo = mid + 1; else hi = mid; }
                        const uint32_t v = S.order[r];
                        S.order2[r + lo] = v;
                        S.rank[v] = (int32_t)(r + lo);
                    }
                    for (uint32_t t = tid; t < T; t += NT) {
                        const uint32_t v = S.nn[2 * t], pos = S.nn[2 * t + 1] + t;
                        S.order2[pos] = v;
                        S.rank[v] = (int32_t)pos;
                    }
                    __syncthreads();
                    for (uint32_t r = first + tid; r < n_old + T; r += NT) S.order[r] = S.order2[r];
                    __syncthreads();
                }
            }
            t_add += PT_NOW() - t3;
            t_merge += PT_NOW() - t4;
        }

        // ---- generate_multiple_sequence_alignment: column per node, then per base ----
        const unsigned long long t5 = PT_NOW();
        (void)t5;
        __syncthreads();
        if (tid == 0) { s_bc[6] = 0; s_bc[4] = S.err; }
        __syncthreads();
        if (!S.err && S.n_nodes > 0) {
            if (w0) {
                uint32_t n_emit, n_cols;
                toposort(S, A, 1, n_emit, n_cols);
                if (!S.err && n_emit != S.n_nodes) S.err = POA_ERR_GRAPH;
                if (tid == 0) { s_bc[6] = n_cols; s_bc[4] = S.err; }
            }
            __syncthreads();
            S.err = s_bc[4];
            if (!S.err) {
                const uint64_t b0 = A.off[q0], b1 = A.off[q1];
                for (uint64_t b = b0 + tid; b < b1; b += NT) A.out_col[b] = (uint32_t)S.rank[A.out_col[b]];
            }
        }
        const uint32_t width = s_bc[6];
        if (tid == 0) {
            A.out_width[pk] = width;
            A.status[pk] = S.err;
            if (A.timeline) A.timeline[2 * pk + 1] = (unsigned long long)wall_clock64();
            // the reference's work is counted once, by the pass that finishes the pack (a pack that outgrows its slot or loses its band is run
            // again from the start); what the device computed is counted every time
            if (!S.err) {
                atomicAdd(&A.counters[0], cells);
                atomicAdd(&A.counters[1], (unsigned long long)(q1 - q0));
#ifndef POA_PROFILE
                atomicAdd(&A.counters[2], (unsigned long long)S.n_nodes);
#endif
                atomicAdd(&A.counters[3], rows);
            }
#ifdef POA_PROFILE
            atomicAdd(&A.counters[2], t_tie);          // tie resolution (part of counters[6])
#endif
            atomicAdd(&A.counters[11], cells_done);
            if constexpr (PK == 8) { atomicAdd(&A.counters[12], (unsigned long long)n_band_ok); atomicAdd(&A.counters[13], (unsigned long long)n_band_fail); atomicAdd(&A.counters[10], (unsigned long long)n_strip_aln); }
#ifdef POA_PROFILE
            (void)t_topo; (void)t_dp; (void)t_tb; (void)t_add;                              // counters[4]: ties << 32 | ties that needed the exact sort
            atomicAdd(&A.counters[5], t_dp);           // DP rows
            atomicAdd(&A.counters[6], t_tb);           // best cell + traceback
            atomicAdd(&A.counters[7], t_add);          // add_alignment
            const unsigned long long t_end = PT_NOW();
            atomicAdd(&A.prof[0], t_topo); atomicAdd(&A.prof[1], t_dp); atomicAdd(&A.prof[2], t_tie); atomicAdd(&A.prof[3], t_tb - t_tie);
            atomicAdd(&A.prof[4], t_add - t_merge); atomicAdd(&A.prof[5], t_merge); atomicAdd(&A.prof[6], t_end - t5); atomicAdd(&A.prof[7], t_end - t_pack0);
#endif
        }
    }
}

#ifdef POA_DEVICE_TEST      // (compile-time experiments: one instance, no host side)
template __global__ void poa_kernel<POA_DEVICE_TEST>(poa_args);
}  // namespace rattle
#else
// ---- host side ------------------------------------------------------------------------------------
template <int CPL, int RING, int NW, int PK>
static hipError_t launch_poa(const poa_args &A, uint32_t n_slots, size_t shm, hipStream_t st) {
    if (shm > 60 * 1024)
        (void)hipFuncSetAttribute((const void *)poa_kernel<CPL, RING, NW, PK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL((poa_kernel<CPL, RING, NW, PK>), dim3(n_slots), dim3(64 * NW * pk_teams(RING, PK)), shm, st, A);
    return hipGetLastError();
}

template <int CPL, int RING, int NW, int PK>
static int max_blocks_per_cu(size_t shm) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, poa_kernel<CPL, RING, NW, PK>, 64 * NW * pk_teams(RING, PK), shm) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

// Kernel variants.  A pack is one workgroup of NW wavefronts, each thread owning CPL columns, so a
// variant covers sequences up to 64*NW*CPL nt.  Four waves per pack minimise the time of one pack
// (small inputs: fewer packs than the device has room for); one or two waves per pack need no (or a
// cheaper) per-row rendezvous and leave room for more packs per CU (large inputs: throughput).
struct poa_variant {
    uint32_t cpl, ring, nw, pk;
    hipError_t (*launch)(const poa_args &, uint32_t, size_t, hipStream_t);
    int (*max_blocks)(size_t);
};
#define POA_VARIANT(CPL, RING, NW, PK) {CPL, RING, NW, PK, &launch_poa<CPL, RING, NW, PK>, &max_blocks_per_cu<CPL, RING, NW, PK>}
#define POA_CLASSES 8
#define POA_GROUPS 24                          // + the shallow packs of classes 4 .. 7 (8 .. 11), the LONG-CHAIN packs of classes 0 .. 3 (12 .. 15) and the packs of
                                               // classes 0 .. 3 the exact band is tried on (16 .. 19; long chains: 20 .. 23) as groups of their own (poa_device_run)
#ifndef POA_BAND_SPREAD
#define POA_BAND_SPREAD 400                    // the band holds columns - length + 2 t + 1 <= 504 cells per row: a pack whose lengths differ by more than this cannot stay in it
#endif
#ifndef POA_SHALLOW_READS
#define POA_SHALLOW_READS 40
#endif
static inline int poa_group_class(int g) { return g < POA_CLASSES ? g : g < 12 ? g - 4 : g < 16 ? g - 12 : g < 20 ? g - 16 : g - 20; }
static inline bool poa_group_chain(int g) { return (g >= 12 && g < 16) || g >= 20; }
static inline bool poa_group_band(int g) { return g >= 16; }
static const uint32_t k_class_cols[POA_CLASSES - 1] = {1024, 1536, 2048, 2560, 4096, 6144, 8192};
static const poa_variant k_latency[POA_CLASSES] = {POA_VARIANT(4, POA_RING_4x4, 4, 1), POA_VARIANT(6, POA_RING_4x6, 4, 1), POA_VARIANT(8, 8, 4, 1), POA_VARIANT(10, 8, 4, 1),
                                                   POA_VARIANT(8, POA_WIDE_RING, 8, 3), POA_VARIANT(8, POA_WIDE_RING, 12, 3), POA_VARIANT(8, POA_WIDE_RING, 16, 3),
                                                   POA_VARIANT(8, POA_LONG_RING, 16, 2) /* longer than 8192: int32 cells, 8192-column segments one after the other */};
static const poa_variant k_long_noring = POA_VARIANT(8, 0, 16, 2);      // ... row-major segments without a ring when the graph's bitmaps leave no LDS for it
static const poa_variant k_noring[3] = {POA_VARIANT(16, 0, 4, 0), POA_VARIANT(24, 0, 4, 0), POA_VARIANT(32, 0, 4, 0)};      // when the ring no longer fits LDS (huge graphs)
// the packed classes (up to 2560 columns), by load: see choose_variants in poa_device_run
static const poa_variant k_dense[4] = {POA_VARIANT(4, POA_RING_4x4, 4, 1), POA_VARIANT(6, POA_RING_4x6, 4, 1), POA_VARIANT(8, 8, 4, 1), POA_VARIANT(10, 8, 4, 1)};
// teams of wavefronts (dp_rows_mt; the second argument is the number of TEAMS, the ring is sized at launch): four teams for a pack
// that has a CU to itself, two when up to four or five packs share one; one team = the lean skewed pipeline (tests, measurements)
static const poa_variant k_mt4[4] = {POA_VARIANT(4, 4, 4, 7), POA_VARIANT(6, 4, 4, 7), POA_VARIANT(8, 4, 4, 7), POA_VARIANT(10, 4, 4, 7)};
static const poa_variant k_mt2[4] = {POA_VARIANT(4, 2, 4, 7), POA_VARIANT(6, 2, 4, 7), POA_VARIANT(8, 2, 4, 7), POA_VARIANT(10, 2, 4, 7)};
static const poa_variant k_mt1[4] = {POA_VARIANT(4, 1, 4, 7), POA_VARIANT(6, 1, 4, 7), POA_VARIANT(8, 1, 4, 7), POA_VARIANT(10, 1, 4, 7)};
// near-chain graphs (POA #2 / #3: the caller says so): the exact band (dp_rows_band) for any length up to 2560.  Its rows run on ONE wavefront:
// a crowded pass takes workgroups of one wavefront (sixteen packs per CU, each on its own), a sparse one workgroups of four (the plan, the
// graph update and the tables of the traceback spread over 256 threads).  Packs whose band fails are run again by the forms above.
static const poa_variant k_band1 = POA_VARIANT(4, 8, 1, 8), k_band4 = POA_VARIANT(4, 8, 4, 8);

// Every switch kernel C's host side takes from the environment (tests and measurements only; read once per call, so a test may change
// them between calls): one place, one struct.
struct poa_env {
    const char *timeline = nullptr;      // RATTLE_POA_TIMELINE=<file>: when each pack's workgroup started and finished it
    const char *mode = nullptr;          // RATTLE_POA_MODE = dense | mt4 | mt2 | mt1 | band: one form of the row loop for the packed classes
    int band = -1;                       // RATTLE_POA_BAND = 0 | 1: the exact band for near-chain graphs off / on for every call (default: on where the caller announces such graphs)
    const char *profile_json = nullptr;  // RATTLE_POA_PROFILE_JSON=<file> (POA_PROFILE builds): phase shares per class
    uint32_t node_cap = 0;               // RATTLE_POA_NODE_CAP: first-pass node capacity (tests lower it to force re-runs)
    uint64_t budget_mb = 0;              // RATTLE_POA_BUDGET_MB: arena budget (tests shrink it to exercise the skip path)
    uint32_t debug = 0;                  // RATTLE_POA_DEBUG: bit 0 full sort for ties, bit 1 traceback without the LDS chain, bit 2 ... without jump tables
    uint32_t mt_slots = 0;               // RATTLE_POA_MT_SLOTS: ring slots of the team kernels (tests: a short ring, many predecessors from HBM)
    int streams = 0;                     // RATTLE_POA_STREAMS: streams the classes of a pass are dealt onto
    bool timing = false;                 // RATTLE_TIMING: one line per class and pass
    int head_start_us = 30;              // RATTLE_POA_HEAD_START_US: see poa_head_start (0: none)
    poa_env() {
        timeline = getenv("RATTLE_POA_TIMELINE"); mode = getenv("RATTLE_POA_MODE"); profile_json = getenv("RATTLE_POA_PROFILE_JSON");
        if (const char *v = getenv("RATTLE_POA_NODE_CAP")) node_cap = (uint32_t)std::max(64, atoi(v));
        if (const char *v = getenv("RATTLE_POA_BUDGET_MB")) budget_mb = (uint64_t)atoll(v);
        if (const char *v = getenv("RATTLE_POA_DEBUG")) debug = (uint32_t)atoi(v);
        if (const char *v = getenv("RATTLE_POA_BAND")) band = atoi(v) != 0;
        if (const char *v = getenv("RATTLE_POA_MT_SLOTS")) mt_slots = (uint32_t)std::max(1, atoi(v));
        if (const char *v = getenv("RATTLE_POA_STREAMS")) streams = atoi(v);
        timing = getenv("RATTLE_TIMING") != nullptr;
        if (const char *v = getenv("RATTLE_POA_HEAD_START_US")) head_start_us = std::max(0, std::min(1000, atoi(v)));
    }
};

// A pass launches its groups on several streams at once, and the device places their workgroups in whatever order the queues reach it.
// A group of a few team workgroups that each need (most of) a CU's LDS -- the POA #3 chains of the largest clusters: ONE workgroup's work
// for the whole stage -- loses that race one time in three: the barrier-form workgroups of the other groups take some LDS on every CU, and
// the chain waits until a CU has emptied (0.2-0.3 s of a 0.65 s stage; profiles/round5_timeline_stage_noise.txt: it is one of the sources
// of the +-5 % run-to-run spread of the whole job).  So the streams that start with a device-filling group first run this kernel: a lone
// wavefront that sleeps for a few tens of microseconds -- the head start the few-workgroup groups need to be placed.
// (Stage 1's own spread -- two device-filling groups, 3.6-3.9 s and a 4.6-4.8 s outlier in one run of ten -- is NOT such a race: with every
// group given workgroups in proportion to its work, all resident from the start, it spread just as much, profiles/round5_ab_device_shares.txt.)
__global__ void poa_head_start(uint32_t ticks) {
    const uint64_t t0 = wall_clock64();           // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

// Device-resident core: sequences, offsets and the per-base column output live in HBM; the host only
// plans (lengths / pack boundaries) and reads back pack widths, statuses and counters.
// skipped != nullptr: packs that do not fit the device are flagged there (1) instead of failing the call;
// their width is 0 and their columns are undefined.
int poa_device_run(rattle_ctx *ctx, const uint8_t *d_seq_in, const uint64_t *d_off_in, const uint64_t *off, uint32_t n_seqs,
                   const uint32_t *pack_first, uint32_t n_packs, uint32_t *d_col_out, uint32_t *d_width_out, uint32_t *h_width_out,
                   unsigned long long *h_cnt, std::vector<uint8_t> *skipped) {
    hipStream_t st = ctx->stream;
    const poa_env ENV;
    for (int i = 0; i < 16; ++i) h_cnt[i] = 0;
    if (skipped) skipped->assign(n_packs, 0);
    if (n_packs == 0 || n_seqs == 0) { for (uint32_t p = 0; p < n_packs; ++p) h_width_out[p] = 0; return 0; }
    if (pack_first[0] != 0 || pack_first[n_packs] != n_seqs) { set_error("pack_first must cover [0, n_seqs]"); return RATTLE_ERR_ARG; }

    const char *mode_s = ENV.mode;
    const int force_mode = !mode_s ? 0 : mode_s[0] == 'd' ? 1 : !strcmp(mode_s, "mt4") ? 3 : !strcmp(mode_s, "mt2") ? 4 : !strcmp(mode_s, "mt1") ? 5 : !strcmp(mode_s, "band") ? 6 : 0;
    // the exact band: where the caller announces near-chain graphs (POA #2 / #3), unless a form is forced; RATTLE_POA_BAND overrides both ways
    const bool use_band = force_mode == 6 || (ENV.band >= 0 ? ENV.band == 1 && (force_mode == 0 || force_mode == 1) : force_mode == 0 && ctx->poa_shallow_graphs);
    // length class of each pack: 1024 / 1536 / 2048 / 2560 / 4096 / 6144 columns, or longer (segmented int32 rows)
    std::vector<uint64_t> pbases(n_packs);
    std::vector<uint32_t> pmaxL(n_packs);
    std::vector<uint32_t> by_class[POA_GROUPS];      // groups 0 .. 7: the column classes; 8 .. 11: the SHALLOW packs of classes 4 .. 7 (smaller slots, more of them)
    for (uint32_t p = 0; p < n_packs; ++p) {
        uint32_t m = 0, mn = 0xFFFFFFFFu;
        for (uint32_t q = pack_first[p]; q < pack_first[p + 1]; ++q) {
            const uint32_t len = (uint32_t)(off[q + 1] - off[q]);
            m = std::max<uint32_t>(m, len);
            if (len) mn = std::min(mn, len);
        }
        pbases[p] = off[pack_first[p + 1]] - off[pack_first[p]];
        pmaxL[p] = m;
        if (m > POA_MAX_LEN) {
            if (skipped) { (*skipped)[p] = 1; continue; }
            set_error("sequence longer than " + std::to_string(POA_MAX_LEN) + " nt in a POA pack");
            return RATTLE_ERR_ARG;
        }
        int cls = 0;
        while (cls < POA_CLASSES - 1 && m > k_class_cols[cls]) ++cls;
        // a long-read pack's slot is its DP record: hundreds of MB to GBs, sized by the rows its graph will reach -- which depends
        // on its depth (~1.15 + 0.02 x reads nodes per base, measured at 10 % error).  The shallow packs of a long class get a
        // group of their own, so that the deep ones do not dictate everybody's slot (config 5: 47 slots of 1.15 GB for 747 packs)
        if (cls >= 4 && pack_first[p + 1] - pack_first[p] <= POA_SHALLOW_READS) cls += 4;
        // a POA #3 group of a many-pack cluster (hundreds of pack consensi, correct.cpp:520-532) is ONE workgroup's serial work for the
        // whole pass (0.8 s at 1e6 reads in round 4, beside thousands of POA #2 packs in the dense form): such packs form groups of their
        // own, whose form is chosen by THEIR number -- teams of wavefronts, the shortest row there is -- not by the crowd's
        const bool band_pack = use_band && cls < 4 && m - std::min(m, mn) <= POA_BAND_SPREAD;      // (lengths only: what the graphs look like the kernel finds out)
        if (cls < 4 && pack_first[p + 1] - pack_first[p] > POA_CHAIN_SEQS) cls += 12;
        if (band_pack) cls += cls >= 12 ? 8 : 16;
        by_class[cls].push_back(p);
    }

    dbuf<uint32_t> d_pf, d_queue, d_status;
    dbuf<unsigned long long> d_cnt;
    RT_TRY(d_pf.reserve(n_packs + 1)); RT_TRY(d_queue.reserve(n_packs)); RT_TRY(d_status.reserve(n_packs)); RT_TRY(d_cnt.reserve(160));
    struct view { const uint8_t *p; } d_seq{d_seq_in};
    struct viewo { const uint64_t *p; } d_off{d_off_in};
    struct viewc { uint32_t *p; } d_col{d_col_out}, d_width{d_width_out};
    RT_HIP(hipMemcpyAsync(d_pf.p, pack_first, (n_packs + 1) * 4, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemsetAsync(d_cnt.p, 0, 160 * 8, st));
    RT_HIP(hipMemsetAsync(d_status.p, 0xFF, n_packs * 4, st));
    RT_HIP(hipMemsetAsync(d_width.p, 0, n_packs * 4, st));         // skipped packs keep width 0 (kernel D then leaves them alone)
    std::vector<uint32_t> h_status(n_packs);
#ifdef POA_HIST
    unsigned long long h_hist[160] = {0};
#endif
#ifdef POA_PROFILE
    unsigned long long h_prof[POA_CLASSES * 8] = {0};
#endif
#ifdef POA_BARPROF
    unsigned long long h_seg[4] = {0};
#endif

    size_t free_b = 0, total_b = 0;
    RT_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += ctx->poa_arena_bytes;            // the cached arena is ours to reuse
    hipDeviceProp_t prop;
    RT_HIP(hipGetDeviceProperties(&prop, ctx->device));
    const uint32_t n_cu = (uint32_t)std::max(1, prop.multiProcessorCount);
    int rc = 0;
    if (!ctx->poa_go) {
        RT_HIP(hipEventCreateWithFlags(&ctx->poa_go, hipEventDisableTiming));
        for (int i = 0; i < 16; ++i) {
            RT_HIP(hipStreamCreateWithFlags(&ctx->poa_st[i], hipStreamNonBlocking));
            RT_HIP(hipEventCreateWithFlags(&ctx->poa_ev[i], hipEventDisableTiming));
        }
    }
    dbuf<uint32_t> d_heads;
    RT_TRY(d_heads.reserve(32));
    // RATTLE_POA_TIMELINE=<file>: when each pack's workgroup started and finished it (measurement aid: how full the device is over a pass)
    const char *tl_path = ENV.timeline;
    dbuf<unsigned long long> d_tl;
    if (tl_path) { RT_TRY(d_tl.reserve(2 * (size_t)n_packs)); RT_HIP(hipMemsetAsync(d_tl.p, 0, 16 * (size_t)n_packs, st)); }
    struct cls_plan {
        std::vector<uint32_t> todo;
        uint32_t node_cap = 10240u;                // first-round capacity (RATTLE_POA_NODE_CAP: tests lower it to force re-runs)
        uint64_t cell_cap = 24ull << 20;           // elements per matrix
        poa_args A;
        uint64_t per_slot = 0;
        size_t shm = 0;
        uint32_t n_slots = 0;
        int bpc = 1;
        int rounds = 0;                            // passes this class has taken
        bool clamped = false;                      // cell_cap was cut to what one slot can get: packs that still fail are skipped
        bool no_teams = false;                     // a pack of this group failed with POA_ERR_SYNC: its retries take the barrier form
        bool no_band = false;                      // packs of this group failed with POA_ERR_BAND: what is left of the group takes the full rows
        const poa_variant *V = nullptr;
    } C[POA_GROUPS];
    for (int c = 0; c < POA_GROUPS; ++c) { C[c].todo = by_class[c]; if (ENV.node_cap) C[c].node_cap = ENV.node_cap; }
    // The kernel of a class is chosen per PASS from how full the device will be (round 3's verdict: one wavefront / column split
    // per class, chosen by read length only, collapsed to 0.08 of the issue roofline whenever fewer packs were resident than the
    // device has places -- 1e5 reads, the toyset, stages 2a / 3a / 3b, every rank of an 8-GPU job, the re-run of a few packs):
    // the barrier form keeps the most packs resident (seven or eight per CU: record words in the ring, 2 bytes per cell); the teams of
    // wavefronts give a pack that has (most of) a CU to itself the shortest row there is (dp_rows_mt).
    // tests / measurements: RATTLE_POA_MODE = dense | mt4 | mt2 | mt1 forces one form for the packed classes
    uint32_t live_per_cu = 1, chain_per_cu = 1;    // packs per CU the pass about to start will keep resident (all classes; the long-chain groups)
    auto choose_variants = [&]() {
        uint64_t live = 0, chains = 0;
        for (int c = 0; c < POA_GROUPS; ++c) (poa_group_chain(c) ? chains : live) += C[c].todo.size();
        live_per_cu = (uint32_t)std::max<uint64_t>(1, (live + n_cu - 1) / n_cu);
        chain_per_cu = (uint32_t)std::max<uint64_t>(1, (chains + n_cu - 1) / n_cu);
        // by packs per CU: about one -> four teams per pack (16 wavefronts: the CU is the pack's); up to four and a half -> two teams (two or
        // three packs per CU and a short queue); more -> the barrier form with record words (seven or eight per CU)
        // (in eighths of a pack per CU: 829 packs on 256 CUs are 3.2 per CU)
        const uint64_t l8 = live * 8 / n_cu;
        // measured (profiles/round5_forms_by_load.txt; 1024- / 1536-column class, GCUPS): one pack per CU -- four teams 303 / 494, two
        // teams 313 / 463; two per CU -- two teams 557 / 750; three -- two teams 680 / 624 (a short queue), one team 544 / 697, round 4's
        // pipeline 498 / 454, barrier form 441 / 529; four -- one team 631, pipeline 635 / 569, barrier 569 / 649; five and more: the barrier form
        int mode = force_mode && force_mode != 6 ? force_mode : l8 <= 10 ? 3 : l8 <= 36 ? 4 : 1;
        // near-chain graphs (POA #2 / #3, the caller says so): no row parallelism to find, and per alignment as much serial work as DP --
        // beyond two packs per CU the barrier form with everything resident wins (measured in the rank replay at 1e6 reads: 833 such
        // packs took 0.46-0.53 s on two teams, 1667 of them 0.39 s in the barrier form)
        if ((!force_mode || force_mode == 6) && ctx->poa_shallow_graphs && l8 > 18) mode = 1;
        const uint64_t c8 = chains * 8 / n_cu;
        const int chain_mode = force_mode >= 3 && force_mode != 6 ? force_mode : c8 <= 10 ? 3 : 4;
        for (int c = 0; c < POA_GROUPS; ++c) {
            C[c].V = &k_latency[poa_group_class(c)];
            const int gc = poa_group_class(c);
            if (gc < 4 && !poa_group_chain(c)) C[c].V = mode == 1 ? &k_dense[gc] : mode == 3 ? &k_mt4[gc] : mode == 4 ? &k_mt2[gc] : &k_mt1[gc];
            if (poa_group_chain(c)) C[c].V = force_mode == 1 ? &k_dense[gc] : chain_mode == 3 ? &k_mt4[gc] : chain_mode == 4 ? &k_mt2[gc] : &k_mt1[gc];
            if (C[c].no_teams && gc < 4) C[c].V = &k_dense[gc];
            if (poa_group_band(c) && !C[c].no_band) C[c].V = (poa_group_chain(c) ? c8 : l8) > 32 ? &k_band1 : &k_band4;      // (packs per CU in eighths: more than four -> one wavefront per pack)
        }
    };
    // pass 0: many slots with a modest arena; later passes: failed packs with larger arenas.
    // The column classes of one pass run concurrently on their own streams.
    // (RATTLE_POA_BUDGET_MB: tests shrink the arena to exercise the skip path)
    const uint64_t budget = ENV.budget_mb ? ENV.budget_mb << 20 : (uint64_t)(free_b * 0.85);
    // slot layout of class c for its current capacities; returns bytes per slot
    auto plan_class = [&](int c, uint64_t tb, uint32_t tl) {
        cls_plan &P = C[c];
        const uint32_t cpl = P.V->cpl;
        const int gc = poa_group_class(c);
        const bool long_rows = gc == POA_CLASSES - 1;    // int32 cells, sequence read in place (no LDS copy)
        uint32_t ncap = (uint32_t)std::min<uint64_t>(P.node_cap, tb + 1);
        ncap = (ncap + 31u) & ~31u;
        const uint32_t ecap = (uint32_t)std::min<uint64_t>(tb + 1, 0x7FFFFFFFull);
        const uint32_t acap = tl + ncap + 16;
        const uint32_t scap = ncap + POA_STACK;
        const uint32_t qcap = ((tl + cpl - 1) / cpl * cpl + 15u) & ~15u;
        const uint64_t ccap = std::min<uint64_t>(P.cell_cap, (uint64_t)(ncap + 1) * qcap);
        poa_args &A = P.A;
        uint64_t o = 0;
        auto take = [&](uint64_t bytes) { uint64_t r = o; o += (bytes + 255) & ~(uint64_t)255; return r; };
        A.o_nrec = take((uint64_t)ncap * 16); A.o_nal = take((uint64_t)ncap * 16); A.o_edges = take((uint64_t)ecap * 8);
        A.o_rank = take((uint64_t)ncap * 4); A.o_order = take((uint64_t)ncap * 4); A.o_order2 = take((uint64_t)ncap * 4);
        A.o_srank = take((uint64_t)ncap * 4); A.o_rowmax = take(((uint64_t)ncap + 1) * 16); A.o_lh = take(((uint64_t)ncap + 1) * 16); A.o_nn = take(((uint64_t)qcap + 16) * 8);
        A.o_plan = take((uint64_t)ncap * 16); A.o_planb = take((uint64_t)ncap * 16); A.o_planc = take((uint64_t)ncap * 16); A.o_pland = take((uint64_t)ncap * 16);
        A.o_planm = take(P.V->pk == 7 || P.V->pk == 8 ? ((uint64_t)ncap + 8) * 32 : 0);
        const uint64_t cell_bytes = long_rows ? 4 : 2;
        if (pk_packed(P.V->pk)) {          // H words carry F's two bits, E's two bits per column sit in a per-thread array
            A.o_H = take(ccap * 2); A.o_F = take(0); A.o_E = take(0);      // E is rebuilt on demand by the traceback
        } else if (P.V->pk == 0 || P.V->pk == 3) {   // H int16 plus a nibble per column (min(H-F,3), min(H-E,3))
            A.o_H = take(ccap * 2); A.o_F = take(0); A.o_E = take(((uint64_t)ncap + 2) * 64 * P.V->nw * ((cpl + 7) / 8) * 4);
        } else {                           // segmented int32 rows: H int32 plus a nibble per column
            A.o_H = take(ccap * cell_bytes); A.o_F = take(0); A.o_E = take(ccap / 2 + 64);
        }
        A.o_aln = take((uint64_t)acap * 8); A.o_ainfo = take((uint64_t)acap * 16); A.o_spill = take((uint64_t)scap * 4);
        P.per_slot = o;
        A.debug = ENV.debug;
        A.band = 0;
        const uint32_t lds_seq = long_rows ? 16u : qcap;
        A.ring_slots = A.ring_reach = A.ring_slack = 0;
        bool mt_ring_ok = true;
        if (P.V->pk == 7) {
            // dp_rows_mt: the ring takes the LDS a workgroup can have when `live_per_cu` packs share a CU (up to 24 slots); `slack` rows may be in flight behind the reader (two rounds of the teams when there is room, one otherwise),
            // `reach` = slots - slack rows back are served from the ring, the rest from the record in HBM
            const uint32_t teams = P.V->ring, slotb = mt_slot_bytes((int)P.V->cpl, (int)P.V->nw);
            const uint32_t fixed = lds_seq + (2u * poa_bit_words(ncap) + POA_STACK) * 4u + 64u + 3072u;      // + the kernel's static LDS
            uint32_t ppc = std::max<uint32_t>(1, std::min<uint32_t>(poa_group_chain(c) ? chain_per_cu : live_per_cu, teams == 4 ? 1u : teams == 2 ? (MT2_MINWAVES >= 6 && P.V->cpl == 4 ? 3u : 2u) : 4u));      // (registers: 16 wavefronts of 128 per CU)
            uint32_t slots = 0, slack = 0;
            for (; ppc >= 1; --ppc) {
                const uint32_t room = 160u * 1024 / ppc;
                slots = room > fixed + 2 * slotb ? std::min<uint32_t>(24, (room - fixed) / slotb) : 0;
                slack = teams > 1 ? std::max<uint32_t>(teams, std::min<uint32_t>(2 * teams, slots > 10 ? slots - 10 : 0)) : 0;
                if (slots >= slack + 6 || ppc == 1) break;
            }
            if (slots < slack + 2) { slots = slack + 2; }      // (the launch fails loudly if even that does not fit)
            if (ENV.mt_slots) { slack = teams > 1 ? teams : 0; slots = std::max<uint32_t>(slack + 2, ENV.mt_slots); }      // tests: a short ring, to exercise the record path
            // A row whose in-edge lies beyond the ring reads block w - 1's record word of the column left of its block (far_fetch, lane 0).  That
            // store is ordered before the load only through the mailbox wait of this team's PREVIOUS row (row - teams), whose left neighbour had
            // seen every row <= row - teams - slack final in block w - 1: far rows (row - prow > reach) are covered iff reach + 1 >= teams + slack.
            // Less slack first; a ring too short even then takes the barrier form (round 5's advisor: retry passes with 16 x the nodes).
            if (teams > 1 && slots - slack + 1 < teams + slack) slack = teams;
            mt_ring_ok = teams == 1 || slots - slack + 1 >= teams + slack;
            A.ring_slots = slots; A.ring_slack = slack; A.ring_reach = slots - slack;
        }
        if (P.V->pk == 8) {
            // the band's LDS behind the bitmaps and the stack: 8 ring slots of 4 columns per lane (4 when eight do not leave ten workgroups of one
            // wavefront a place on a CU), the selector table, junk words; between two DPs the tie labels and the traceback's tables
            const uint32_t fixed = lds_seq + (2u * poa_bit_words(ncap) + POA_STACK) * 4u + 64u + 1024u;
            uint32_t slots = 8;
            if (P.V->nw == 1 && 10u * (fixed + band_lds_bytes(slots, 4, qcap)) > 160u * 1024) slots = 4;
            A.band = std::max(band_lds_bytes(slots, 4, qcap), band_lds_bytes(2, 8, qcap));      // (eight columns per lane: the last resort, with a ring of two rows)
            if (const char *v = getenv("RATTLE_POA_BAND_LDS")) A.band = (uint32_t)atoi(v);      // (measurements)
        }
        auto lds_bytes = [&](const poa_variant *V) {
            if (V->pk == 8) return (size_t)lds_seq + (2u * poa_bit_words(ncap) + POA_STACK) * 4u + (size_t)A.band + 64;
            if (V->pk == 7) return (size_t)lds_seq + (2u * poa_bit_words(ncap) + POA_STACK) * 4u + (size_t)A.ring_slots * mt_slot_bytes((int)V->cpl, (int)V->nw) + 64;
            return (size_t)lds_seq + poa_region_bytes(ncap, (int)V->cpl, (int)V->ring, (int)V->nw, (int)V->pk) + 64;
        };
        // a retry pass with a huge graph: the node bitmaps leave no room for a ready-made ring -- fall back to the dense form (2 bytes per cell)
        if (gc < 4 && P.V->pk != 1 && P.V->pk != 8 && (lds_bytes(P.V) > 158u * 1024 || !mt_ring_ok)) { P.V = &k_dense[gc]; A.ring_slots = A.ring_reach = A.ring_slack = 0; A.o_planm = take(0); A.band = 0; }
        if ((gc == 4 || gc == 5 || gc == 6) && lds_bytes(P.V) > 158u * 1024) P.V = &k_noring[gc - 4];
        if (gc == POA_CLASSES - 1 && lds_bytes(P.V) > 158u * 1024) P.V = &k_long_noring;
        P.shm = lds_bytes(P.V);
        A.node_cap = ncap; A.edge_cap = ecap; A.cell_cap = ccap; A.aln_cap = acap; A.spill_cap = scap; A.seq_cap = lds_seq;
        return o;
    };
    auto give_up = [&](cls_plan &P) -> int {       // the class's remaining packs do not fit this device
        if (!skipped) { set_error("poa: " + std::to_string(P.todo.size()) + " pack(s) exceed the device arena"); return RATTLE_ERR_HIP; }
        for (uint32_t p : P.todo) (*skipped)[p] = 1;
        P.todo.clear();
        return 0;
    };
    for (int pass = 0; pass < 64 && rc == 0; ++pass) {
        bool any = false;
        uint64_t want_bytes = 0;
        choose_variants();
        for (int c = 0; c < POA_GROUPS && rc == 0; ++c) {
            cls_plan &P = C[c];
            P.n_slots = 0;
            if (P.todo.empty()) continue;
            if (P.rounds >= 6) { rc = give_up(P); continue; }
            std::sort(P.todo.begin(), P.todo.end(), [&](uint32_t a, uint32_t b) { return pbases[a] != pbases[b] ? pbases[a] > pbases[b] : a < b; });
            uint64_t tb = 0; uint32_t tl = 0;
            for (uint32_t p : P.todo) { tb = std::max(tb, pbases[p]); tl = std::max(tl, pmaxL[p]); }
            if (P.rounds == 0 && !ENV.node_cap) {
                // first-pass capacity: a pack of ~200 reads at 10 % error grows to ~5 nodes per base of its
                // longest read; packs that still outgrow it are re-run with 4x nodes
                P.node_cap = std::max<uint32_t>(P.node_cap, (uint32_t)std::min<uint64_t>(7ull * tl, 1u << 20));
                // the DP record is the arena: 7 nodes per base of the longest read (measured: ~4.5 at the end of a 200-read pack
                // at 10 % error), not a fixed floor -- a third of the memory, and of the seconds the allocation takes
                P.cell_cap = (uint64_t)(std::min<uint64_t>(std::max<uint64_t>(7ull * tl, 2048), tb + 1) + 64) * (tl + 32);
                if (P.V->pk == 8) {
                    // near-chain graphs: a few nodes beyond the longest sequence (3 x covers every pack of the bench's POA #2 / #3 four times over;
                    // a graph that outgrows it is no chain, and its pack goes to the full rows anyway); the record is the band's: 256 cells per row
                    P.node_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(3ull * tl, 1024), 1u << 20);
                    P.cell_cap = ((uint64_t)std::min<uint64_t>(P.node_cap, tb + 1) + 64) * (((uint64_t)tl + 511) / 512 * 512);      // (the band needs 512 cells per row at most; the full rows as strips of 512 columns the whole width)
                }
                if (poa_group_class(c) >= 4) {
                    // long reads: rows by depth, pack by pack (x 1.25 of the measured growth; a pack that still outgrows its slot is re-run)
                    uint64_t nmax = 0, cmax = 0;
                    for (uint32_t p : P.todo) {
                        const uint32_t reads = std::min<uint32_t>(pack_first[p + 1] - pack_first[p], 200u);
                        const uint64_t rows = std::min<uint64_t>(pbases[p] + 1, (uint64_t)(pmaxL[p] * (1.15 + 0.02 * reads) * 1.25) + 256);
                        nmax = std::max(nmax, rows);
                        cmax = std::max(cmax, (rows + 64) * (pmaxL[p] + 32));
                    }
                    P.node_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(nmax, 2048), 1u << 20);
                    P.cell_cap = cmax;
                }
            }
            uint64_t per = plan_class(c, tb, tl);
            if (per > budget) {
                // one slot does not fit the HBM that is left: cut the DP record to what does fit; a pack that
                // still fails with that is beyond this device (skip-and-report, or an error for the MSA entry)
                const uint64_t cell_b2 = poa_group_class(c) == POA_CLASSES - 1 ? 9 : 4;       // twice the bytes per cell
                const uint64_t fixed = per - (P.A.cell_cap * cell_b2 + 1) / 2;
                if (P.clamped || fixed + (1ull << 20) >= budget) { rc = give_up(P); continue; }
                P.cell_cap = (budget - fixed - (1ull << 20)) * 2 / cell_b2;
                P.clamped = true;
                per = plan_class(c, tb, tl);
                if (per > budget) { rc = give_up(P); continue; }
            }
            any = true;
            P.bpc = P.V->max_blocks(P.shm);
            uint32_t places = n_cu * (uint32_t)P.bpc;
            P.n_slots = std::min<uint32_t>((uint32_t)P.todo.size(), places);
            want_bytes += P.per_slot * P.n_slots;
        }
        if (rc || !any) break;
        if (want_bytes > budget) {
            // Memory decides how many packs of each group run at once.  A group lasts about (its work) / (its slots), so the pass
            // is shortest when every group gets slots in proportion to its work: T = sum(work x bytes per slot) / budget,
            // slots = work / T (config 5: the deep 8 - 20 kb packs are a few per cent of the packs and most of the work; a cut in
            // proportion to the number of packs left them 4 slots and the pass to one class).  Work of a pack as below: bases x
            // longest read / threads.  Groups capped by their pack count or by the device hand their share back (two rounds).
            double work[POA_GROUPS] = {0};
            uint32_t cap[POA_GROUPS] = {0};
            bool fixed[POA_GROUPS] = {false};
            for (int c = 0; c < POA_GROUPS; ++c) if (C[c].n_slots) {
                cap[c] = C[c].n_slots;
                for (uint32_t p : C[c].todo) work[c] += (double)pbases[p] * (double)pmaxL[p] / (64.0 * C[c].V->nw);
            }
            for (int round = 0; round < 3; ++round) {
                double wm = 0, left = (double)budget;
                for (int c = 0; c < POA_GROUPS; ++c) if (cap[c]) {
                    if (fixed[c]) left -= (double)C[c].per_slot * C[c].n_slots;
                    else wm += work[c] * (double)C[c].per_slot;
                }
                if (wm <= 0 || left <= 0) break;
                const double T = wm / left;
                for (int c = 0; c < POA_GROUPS; ++c) if (cap[c] && !fixed[c]) {
                    const double want = work[c] / T;
                    if (want >= cap[c]) { C[c].n_slots = cap[c]; fixed[c] = true; }
                    else C[c].n_slots = std::max<uint32_t>(1, (uint32_t)want);
                }
            }
            want_bytes = 0;
            for (int c = POA_GROUPS - 1; c >= 0; --c) if (C[c].n_slots) {
                if (want_bytes + C[c].per_slot * C[c].n_slots > budget) {
                    C[c].n_slots = (uint32_t)((budget - want_bytes) / C[c].per_slot);     // may become 0: deferred to a later pass
                }
                want_bytes += C[c].per_slot * C[c].n_slots;
            }
        }
        phase_timer T_round("    poa pass (arena+kernels)");
        if (ctx->poa_arena_bytes < want_bytes) {
            if (ctx->poa_arena) (void)hipFree(ctx->poa_arena);
            ctx->poa_arena = nullptr; ctx->poa_arena_bytes = 0;
            // allocating (and freeing) tens of GB costs seconds (the driver clears the memory, ~20 ms per GB): take a tenth
            // more so that the next stage, whose packs differ a little, does not trigger another round of it
            uint64_t take_bytes = std::min<uint64_t>(budget, want_bytes + want_bytes / 10);
            if (hipMalloc((void **)&ctx->poa_arena, take_bytes) != hipSuccess) {
                take_bytes = want_bytes;
                if (hipMalloc((void **)&ctx->poa_arena, take_bytes) != hipSuccess) { set_error("poa: arena allocation failed"); rc = RATTLE_ERR_HIP; break; }
            }
            ctx->poa_arena_bytes = take_bytes;
        }
        hipError_t e = hipMemsetAsync(d_heads.p, 0, 128, st);
        uint64_t aoff = 0;
        uint32_t qoff = 0;
        for (int c = 0; c < POA_GROUPS && e == hipSuccess; ++c) {
            cls_plan &P = C[c];
            if (!P.n_slots) continue;
            e = hipMemcpyAsync(d_queue.p + qoff, P.todo.data(), P.todo.size() * 4, hipMemcpyHostToDevice, st);
            poa_args &A = P.A;
            A.seq = d_seq.p; A.off = d_off.p; A.pack_first = d_pf.p; A.queue = d_queue.p + qoff; A.n_queue = (uint32_t)P.todo.size();
            A.queue_head = d_heads.p + c; A.arena = ctx->poa_arena + aoff; A.slot_stride = P.per_slot;
            A.out_col = d_col.p; A.out_width = d_width.p; A.status = d_status.p; A.counters = d_cnt.p; A.prof = d_cnt.p + 32 + 8 * poa_group_class(c);
            A.timeline = tl_path ? d_tl.p : nullptr;
            aoff += P.per_slot * P.n_slots;
            qoff += (uint32_t)P.todo.size();
            if (ENV.timing)
                fprintf(stderr, "[rattle]     poa class %s%u cols (%u waves x %u, %s %u%s) pass %d: %zu packs, %u slots x %.1f MB, %d blocks/CU\n",
                        poa_group_class(c) == POA_CLASSES - 1 ? "> 8192: segments of " : poa_group_band(c) ? (poa_group_chain(c) ? "(band, long chains) " : "(band) ") : c >= 12 ? "(long chains) " : c >= POA_CLASSES ? "(shallow packs) " : "", 64 * P.V->nw * P.V->cpl, P.V->nw, P.V->cpl,
                        P.V->pk == 7 ? "teams" : "ring", P.V->ring, P.V->pk == 7 ? (", ring " + std::to_string(A.ring_slots) + " reach " + std::to_string(A.ring_reach)).c_str() : "", pass, P.todo.size(), P.n_slots,
                        P.per_slot / 1e6, P.bpc);
        }
        if (e != hipSuccess) { set_error(std::string("poa setup: ") + hipGetErrorString(e)); rc = RATTLE_ERR_HIP; break; }
        {
            ktimer T(ctx, K_POA, 0);
            e = hipEventRecord(ctx->poa_go, st);
            // The classes of a pass run concurrently, but the device takes kernels from a handful of hardware queues (four by
            // default) that HIP streams share: seven classes on seven streams left the 6144-column class waiting behind the
            // 8192-column one for its whole run (config 5, profiles/round3_*).  So the classes are dealt onto at most four
            // streams, longest estimated run first onto the least loaded stream; a stream runs its classes in that order.
            // Estimate: a pack costs bases x longest read / threads of its workgroup; a class lasts as long as its longest
            // pack or its total over its slots.
            double est[POA_GROUPS] = {0};
            int order[POA_GROUPS], n_run = 0;
            for (int c = 0; c < POA_GROUPS; ++c) {
                cls_plan &P = C[c];
                if (!P.n_slots) continue;
                double tot = 0, longest = 0;
                for (uint32_t p : P.todo) { const double w = (double)pbases[p] * (double)pmaxL[p] / (64.0 * P.V->nw); tot += w; longest = std::max(longest, w); }
                est[c] = std::max(longest, tot / P.n_slots);
                order[n_run++] = c;
            }
            // groups of a handful of team workgroups go first (and get a head start, see poa_head_start)
            bool few[POA_GROUPS] = {false};
            bool any_few = false, any_crowd = false;
            for (int c = 0; c < POA_GROUPS; ++c) if (C[c].n_slots) {
                few[c] = (C[c].V->pk == 7 || (C[c].V->pk == 8 && poa_group_chain(c))) && C[c].n_slots * 8u <= n_cu;      // a handful: the chains.  (Two team groups that each take half the device are a crowd:
                                                                             // with the smaller one sent ahead, stage 1 of a rank of eight went from 0.9 to 1.08 s)
                (few[c] ? any_few : any_crowd) = true;
            }
            const bool head_start = any_few && any_crowd && ENV.head_start_us > 0;
            std::sort(order, order + n_run, [&](int a, int b) { return few[a] != few[b] ? few[a] : est[a] != est[b] ? est[a] > est[b] : a > b; });
            // (hw_queues(), abi.hip: what the HIP runtime of this process was started with -- the library raises GPU_MAX_HW_QUEUES to
            // 12 when it is loaded before the runtime starts, so that every class gets a queue of its own; an application whose
            // runtime was already running with the default is dealt four streams)
            const int hwq = hw_queues();
            const int n_streams = std::max(1, std::min(16, ENV.streams ? ENV.streams : hwq));
            double load[16] = {0};
            bool used[16] = {false};
            for (int i = 0; i < n_run && e == hipSuccess; ++i) {
                const int c = order[i];
                int sidx = 0;
                for (int t = 1; t < n_streams; ++t) if (load[t] < load[sidx]) sidx = t;
                load[sidx] += est[c];
                cls_plan &P = C[c];
                hipStream_t cs = ctx->poa_st[sidx];
                if (!used[sidx]) {
                    e = hipStreamWaitEvent(cs, ctx->poa_go, 0); used[sidx] = true;
                    if (e == hipSuccess && head_start && !few[c]) { poa_head_start<<<1, 64, 0, cs>>>((uint32_t)ENV.head_start_us * 100u); e = hipGetLastError(); }
                }
                if (e != hipSuccess) break;
                e = P.V->launch(P.A, P.n_slots, P.shm, cs);
            }
            for (int t = 0; t < n_streams && e == hipSuccess; ++t) if (used[t]) {
                e = hipEventRecord(ctx->poa_ev[t], ctx->poa_st[t]);
                if (e == hipSuccess) e = hipStreamWaitEvent(st, ctx->poa_ev[t], 0);
            }
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h_status.data(), d_status.p, n_packs * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error(std::string("poa_kernel: ") + hipGetErrorString(e)); rc = RATTLE_ERR_HIP; break; }
        for (int c = 0; c < POA_GROUPS && rc == 0; ++c) {
            cls_plan &P = C[c];
            if (!P.n_slots) continue;
            ++P.rounds;
            std::vector<uint32_t> again;
            bool sync_retry = false, band_retry = false;
            size_t cap_retry = 0;
            const bool was_band = P.V->pk == 8;
            for (uint32_t p : P.todo) {
                const uint32_t s = h_status[p];
                if (s == POA_OK) continue;
                if (s == POA_ERR_NODES || s == POA_ERR_CELLS || s == POA_ERR_SPILL || s == POA_ERR_ALN) ++cap_retry;
                if (s == POA_ERR_NODES || s == POA_ERR_CELLS || s == POA_ERR_SPILL || s == POA_ERR_ALN) again.push_back(p);
                else if (s == POA_ERR_BAND) { again.push_back(p); band_retry = true; }
                else if (s == POA_ERR_SYNC && !P.no_teams) {
                    // a wavefront of the team kernels gave up a bounded wait (a debugger stop, a trap handler, a throttled device, or a
                    // protocol error): the pack is dropped and run again in the barrier form, which cannot time out
                    again.push_back(p); sync_retry = true;
                    if (ENV.timing) {
                        unsigned long long dbg[3] = {0, 0, 0};
                        (void)hipMemcpy(dbg, d_cnt.p + 8, sizeof(dbg), hipMemcpyDeviceToHost);
                        fprintf(stderr, "[rattle]     poa: pack %u gave up wait %llu of team %llu block %llu at row %llu (saw %d, wants %d): again in the barrier form\n", p, dbg[0] & 0xFF, (dbg[0] >> 8) & 0xFF,
                                (dbg[0] >> 16) & 0xFF, dbg[0] >> 32, (int32_t)(dbg[1] & 0xFFFFFFFFu), (int32_t)(dbg[1] >> 32));
                    }
                } else {
                    std::string msg = "poa_kernel: pack " + std::to_string(p) + " failed with status " + std::to_string(s);
                    if (s == POA_ERR_SYNC) {       // a wavefront of dp_rows_mt gave up waiting: say who, for what, at which row
                        unsigned long long dbg[3] = {0, 0, 0};
                        (void)hipMemcpy(dbg, d_cnt.p + 8, sizeof(dbg), hipMemcpyDeviceToHost);
                        msg += " (wait " + std::to_string(dbg[0] & 0xFF) + " of team " + std::to_string((dbg[0] >> 8) & 0xFF) + " block " + std::to_string((dbg[0] >> 16) & 0xFF) + " at row " + std::to_string(dbg[0] >> 32) +
                               ": saw " + std::to_string((int32_t)(dbg[1] & 0xFFFFFFFFu)) + ", wants " + std::to_string((int32_t)(dbg[1] >> 32)) + "; rows " + std::to_string(dbg[2] & 0xFFFFFFFFu) + ", columns " + std::to_string(dbg[2] >> 32) + ")";
                    }
                    set_error(msg); rc = RATTLE_ERR_HIP; break;
                }
            }
            P.todo.swap(again);
            if (sync_retry) { P.no_teams = true; if (cap_retry == 0) continue; }      // same capacities, another form
            if (band_retry || (was_band && cap_retry)) {
                // what is left of this group runs over the full rows from here on, with the capacities of a first pass of those forms
                if (ENV.timing) {
                    unsigned long long dbg[2] = {0, 0};
                    (void)hipMemcpy(dbg, d_cnt.p + 14, sizeof(dbg), hipMemcpyDeviceToHost);
                    fprintf(stderr, "[rattle]     poa: %zu pack(s) of group %d have an alignment without a certified band (or outgrew the band's slot): again over the full rows (the last one: sequence %llu of its pack, %llu nt against %llu rows, best score in the band %llu)\n",
                            P.todo.size(), c, dbg[1] & 0xFFFFFFFFu, dbg[0] >> 32, dbg[0] & 0xFFFFFFFFu, dbg[1] >> 32);
                }
                P.no_band = true; P.rounds = 0; P.node_cap = ENV.node_cap ? ENV.node_cap : 10240u; P.cell_cap = 24ull << 20;
                continue;
            }
            if (!P.todo.empty()) {
                if (P.clamped) { rc = give_up(P); continue; }
                P.node_cap = std::min<uint32_t>(P.node_cap * 4, 1u << 20); P.cell_cap *= 8;
            }
        }
    }
    d_heads.release();
    if (rc == 0) {
        size_t left = 0;
        for (int c = 0; c < POA_GROUPS; ++c) left += C[c].todo.size();
        if (left) { set_error("poa: " + std::to_string(left) + " pack(s) exceed the device arena"); rc = RATTLE_ERR_HIP; }
    }
    if (rc == 0) {
        hipError_t e = hipMemcpyAsync(h_width_out, d_width.p, n_packs * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h_cnt, d_cnt.p, 128, hipMemcpyDeviceToHost, st);
#ifdef POA_PROFILE
        if (e == hipSuccess) e = hipMemcpyAsync(h_prof, d_cnt.p + 32, sizeof(h_prof), hipMemcpyDeviceToHost, st);
#endif
#ifdef POA_BARPROF
        if (e == hipSuccess) e = hipMemcpyAsync(h_seg, d_cnt.p + 16, sizeof(h_seg), hipMemcpyDeviceToHost, st);
#endif
#ifdef POA_HIST
        if (e == hipSuccess) e = hipMemcpyAsync(h_hist, d_cnt.p, 160 * 8, hipMemcpyDeviceToHost, st);
#endif
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error(std::string("poa readback: ") + hipGetErrorString(e)); rc = RATTLE_ERR_HIP; }
    }
    if (rc == 0 && ENV.timing && (h_cnt[12] || h_cnt[13] || h_cnt[10]))
        fprintf(stderr, "[rattle]     poa band: %llu alignments with a certified band, %llu failed certificates, %llu alignments over the full rows as strips; cells computed %.3g of %.3g\n", h_cnt[12], h_cnt[13], h_cnt[10],
                (double)h_cnt[11], (double)h_cnt[0]);
    if (rc == 0 && tl_path) {
        std::vector<unsigned long long> tl(2 * (size_t)n_packs);
        if (hipMemcpy(tl.data(), d_tl.p, 16 * (size_t)n_packs, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(tl_path, "a")) {
                fprintf(f, "# pass: pack class reads bases start end (ticks of 10 ns)\n");
                for (uint32_t p = 0; p < n_packs; ++p) {
                    int cls = 0;
                    while (cls < POA_CLASSES - 1 && pmaxL[p] > k_class_cols[cls]) ++cls;
                    fprintf(f, "%u %d %u %llu %llu %llu\n", p, cls, pack_first[p + 1] - pack_first[p], (unsigned long long)pbases[p], tl[2 * p], tl[2 * p + 1]);
                }
                fclose(f);
            }
        }
    }
    d_pf.release(); d_queue.release(); d_status.release(); d_cnt.release();
    if (rc) return rc;
    if (skipped) for (uint32_t p = 0; p < n_packs; ++p) if ((*skipped)[p]) h_width_out[p] = 0;
#ifdef POA_HIST
    fprintf(stderr, "[rattle] predecessor row distance histogram (first in-edge | further in-edges), d = 1..62, 63+:\n");
    for (int d = 1; d < 64; ++d) fprintf(stderr, "  d%-2d %12llu %12llu\n", d, h_hist[16 + d], h_hist[80 + d]);
#endif
#ifdef POA_PROFILE
    {
        // phase breakdown per column class (ticks of the 100 MHz wall clock, summed over the class's workgroups)
        static const char *names[8] = {"plan", "dp_rows", "ties", "traceback", "add_alignment", "merge_order", "final_sort_columns", "pack_total"};
        FILE *jf = ENV.profile_json ? fopen(ENV.profile_json, "a") : nullptr;
        for (int c = 0; c < POA_CLASSES; ++c) {
            const unsigned long long *q = h_prof + 8 * c;
            if (!q[7]) continue;
            fprintf(stderr, "[rattle] poa class %d (%u cols) phases, %% of workgroup time:", c, c < POA_CLASSES - 1 ? k_class_cols[c] : 0u);
            for (int i = 0; i < 7; ++i) fprintf(stderr, " %s %.1f", names[i], 100.0 * (double)q[i] / (double)q[7]);
            fprintf(stderr, "  (total %.3f block-seconds)\n", (double)q[7] * 1e-8);
            if (jf) {
                fprintf(jf, "{\"class\": %d, \"cols\": %u, \"n_packs\": %zu, \"block_seconds\": %.6f", c, c < POA_CLASSES - 1 ? k_class_cols[c] : 0u, by_class[c].size() + (c >= 4 ? by_class[c + 4].size() : 0), (double)q[7] * 1e-8);
                for (int i = 0; i < 7; ++i) fprintf(jf, ", \"%s_pct\": %.2f", names[i], 100.0 * (double)q[i] / (double)q[7]);
                fprintf(jf, "}\n");
            }
        }
        if (jf) fclose(jf);
    }
#endif
#ifdef POA_BARPROF
    fprintf(stderr, "[rattle] barrier cycles / dp cycles per wave: %.3f %.3f %.3f %.3f  (dp cycles per row, wave 0: %.0f)\n", (double)h_cnt[8] / h_cnt[12], (double)h_cnt[9] / h_cnt[13],
            (double)h_cnt[10] / h_cnt[14], (double)h_cnt[11] / h_cnt[15], (double)h_cnt[12] / (double)h_cnt[3]);
    fprintf(stderr, "[rattle] wave 1, cycles per row: predecessors %.0f, scores+prefix+scan %.0f, barrier %.0f, after barrier %.0f (of %.0f)\n", (double)h_seg[0] / h_cnt[3],
            (double)h_seg[1] / h_cnt[3], (double)h_cnt[9] / h_cnt[3], (double)h_seg[2] / h_cnt[3], (double)h_cnt[13] / h_cnt[3]);
#endif
    ctx->stats[K_POA].bytes += 6ull * h_cnt[0];
    return 0;
}

// Host-buffer entry (rattle_hip_poa_msa): upload, run, read the per-base columns back and expand rows.
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
    if (total && memchr(seq, 0, total)) { set_error("NUL byte in sequence"); return RATTLE_ERR_ARG; }
    dbuf<uint8_t> d_seq; dbuf<uint64_t> d_off; dbuf<uint32_t> d_col, d_width;
    RT_TRY(d_seq.reserve(total + 64)); RT_TRY(d_off.reserve(n_seqs + 1)); RT_TRY(d_col.reserve(total + 64)); RT_TRY(d_width.reserve(n_packs));
    RT_HIP(hipMemcpyAsync(d_seq.p, seq, total, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_off.p, off, (n_seqs + 1) * 8, hipMemcpyHostToDevice, st));
    std::vector<uint32_t> h_width(n_packs);
    unsigned long long h_cnt[16];
    int rc = poa_device_run(ctx, d_seq.p, d_off.p, off, n_seqs, pack_first, n_packs, d_col.p, d_width.p, h_width.data(), h_cnt, nullptr);
    if (rc == 0) {
        phase_timer T_d2h("    poa readback");
        rc = ctx->h_poa_col.reserve(total);
        if (rc == 0) {
            hipError_t e = hipMemcpyAsync(ctx->h_poa_col.p, d_col.p, total * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { set_error(std::string("poa readback: ") + hipGetErrorString(e)); rc = RATTLE_ERR_HIP; }
        }
    }
    d_seq.release(); d_off.release(); d_col.release(); d_width.release();
    if (rc) return rc;

    // expand rows on the host (one task per pack): row = '-' * width with each base at its column
    phase_timer T_expand("    poa expand rows");
    const uint32_t *h_col = ctx->h_poa_col.p;
    uint64_t bytes = 0;
    for (uint32_t p = 0; p < n_packs; ++p) {
        R->width[p] = h_width[p];
        for (uint32_t q = pack_first[p]; q < pack_first[p + 1]; ++q) { R->row_offset[q] = bytes; bytes += h_width[p]; }
    }
    R->row_offset[n_seqs] = bytes;
    R->rows = (char *)malloc(bytes + 1);
    R->rows[bytes] = 0;
    parallel_for(n_packs, 0, [&](size_t p) {
        const uint32_t q0 = pack_first[p], q1 = pack_first[p + 1];
        if (q0 == q1) return;
        memset(R->rows + R->row_offset[q0], '-', R->row_offset[q1] - R->row_offset[q0]);
        for (uint32_t q = q0; q < q1; ++q) {
            char *row = R->rows + R->row_offset[q];
            for (uint64_t b = off[q]; b < off[q + 1]; ++b) row[h_col[b]] = (char)seq[b];
        }
    });
    for (int i = 0; i < 8; ++i) R->counters[i] = h_cnt[i];
#if !defined(POA_PROFILE) && !defined(POA_PREDSTAT)
    R->counters[4] = h_cnt[11]; R->counters[5] = h_cnt[12]; R->counters[6] = h_cnt[13];      // DP cells computed; alignments with a certified band / a failed certificate
#endif
    return 0;
}

}  // namespace rattle
#endif
