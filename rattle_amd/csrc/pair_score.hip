// Kernel B: full pair comparison.  Replaces get_common_kmers (/root/reference/kmer.cpp:45-67),
// calc_similarity (/root/reference/similarity.cpp:4-97) and var (/root/reference/utils.cpp:36-55)
// as called from cluster_together (/root/reference/cluster.cpp:20-34,44-58).
//
// One wavefront per (read i, read j, strand) pair.
//  stage 1 (64 lanes): the reference merge-joins two (hash,pos)-sorted lists and then sorts
//    the matches by (pos1,pos2).  Here read i's hashes are walked in POSITION order (64
//    positions per step) and each lane binary-searches read j's sorted hash list (staged in
//    LDS); equal-hash entries of j are already pos-ascending, so a wave prefix-sum over the
//    per-lane hit counts emits the matches directly in (pos1,pos2) order -- same multiset
//    (full cross product on repeated hashes, kmer.cpp:56-61), same order, no sort.
//  stage 2 (lane 0): the reference's patience LIS with ceil-mid search, chain
//    reconstruction, co-linearity walk, and the two-pass variance in IEEE double with the
//    reference's operation order (compiled with -ffp-contract=off).
//
// Algorithmic HBM bytes per comparison: 8*(nK_i + nK_j).
#include "common.h"

namespace rattle {

#define PS_EXTRA 384            // LDS words ahead of the hash list: bit-vector + search queue

struct ps_args {
    const uint32_t *uh;          // forward hashes, position order
    const uint64_t *koff;
    const uint32_t *kh[2];
    const uint32_t *kp[2];
    const uint64_t *bv[2];       // 4096-bit vectors of the 6-mers (prefilter of the hash search)
    const uint32_t *pi, *pj;
    const uint8_t *ps;
    uint32_t n_pairs;
    int k;
    uint32_t bcap, mcap;         // LDS capacities (entries)
    int32_t *res;                // [n_pairs][4] bases, hc_bases, n_dist, n_matches
    double *var;
    uint32_t *gscratch;          // oversize path: 5*gstride words per pair, or nullptr
    uint64_t gstride;
    const uint32_t *remap;       // oversize path: launch slot -> pair index, or nullptr
};

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Ordering inside the ONE wavefront of a block.  LDS arrays: LDS operations of a wavefront complete in issue order, so only the
// compiler has to be kept from moving accesses (and vector loads in flight -- the prefetched walk, the deferred position gather --
// stay in flight; a __syncthreads() would drain vmcnt at every step).  Global scratch (oversize path): a full barrier.
// value of lane - 1 (DPP wave_shr:1, no LDS crossbar); lane 0 gets 0
__device__ __forceinline__ int ps_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138 /*wave_shr:1*/, 0xF, 0xF, false); }

template <bool BIG>
__device__ __forceinline__ void ps_sync() {
    if (BIG) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// COUNT: stage 1 only, and only the NUMBER of matches |common| is kept (res[pr] = count; no match arrays in LDS, so twice
// as many pairs are resident per CU).  The greedy driver runs it over every pair that passed the bit-vector filter:
// a pair can only be accepted if bases / min_len >= t_s (cluster.cpp:23-27) and bases <= k * |LIS| <= k * |common|, so a
// pair with k * |common| below that bar is rejected exactly without the patience search; the few others go through the
// full kernel.
// BIG: the oversize pass (match arrays in a global scratch slab, 32-bit indices); otherwise the match arrays live in LDS with
// 16-bit chain indices (cap <= 65535).
//
// What the full pass is bound by is the LATENCY of one wavefront's dependent steps (the pairs of the --iso level: 10 M of them
// at 1e6 reads, 89 us each in round 2), so the dependent global round trips were taken out of the walk: the walked hashes are
// loaded eight steps at a time (the first batch together with the searched list), a match is emitted as an INDEX into the searched list and all positions are gathered in one sweep
// after the walk (one round trip instead of one per 64 searches), and nothing in the LDS path waits on vmcnt.
template <bool COUNT, bool BIG>
__global__ __launch_bounds__(64) void pair_score_kernel(ps_args A) {
    typedef typename std::conditional<BIG, uint32_t, uint16_t>::type idx_t;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const uint32_t slot = blockIdx.x;
    const uint32_t pr = A.remap ? A.remap[slot] : slot;
    const uint32_t ri = A.pi[pr], rj = A.pj[pr];
    const int strand = A.ps[pr];
    const uint64_t oA = A.koff[ri], oB = A.koff[rj];
    const uint32_t nA = (uint32_t)(A.koff[ri + 1] - oA);
    const uint32_t nB = (uint32_t)(A.koff[rj + 1] - oB);
    // Which list is walked and which is searched.  The reference walks read i (the seed: in length-sorted order never the shorter
    // of the two) and the matches come out in (pos1, pos2) order for free.  With mixed read lengths (config 5: a 50 kb seed
    // against a 150 nt candidate, 330 M such comparisons at 5e5 reads) that is 800 search rounds for a handful of matches, so
    // when i is much the longer the SHORT list is walked (j's sorted list, any order) against i's sorted list, and the few matches
    // are sorted by (pos1, pos2) afterwards -- the same multiset (cross product on repeated hashes) in the same order.  The count
    // pass needs no order at all and always walks the shorter list.
    const bool swp = !BIG && (COUNT ? nA > nB : nA > 4u * nB + 256u);
    const uint32_t nW = swp ? nB : nA, nS = swp ? nA : nB;
    const uint32_t *__restrict__ wh = swp ? A.kh[strand] + oB : A.uh + oA;                          // walked: hashes (swap: with positions wp)
    const uint32_t *__restrict__ wp = swp ? A.kp[strand] + oB : nullptr;
    const uint32_t *__restrict__ bh_g = swp ? A.kh[0] + oA : A.kh[strand] + oB;                     // searched: sorted hashes + positions
    const uint32_t *__restrict__ bp_g = swp ? A.kp[0] + oA : A.kp[strand] + oB;

    // LDS carve: [B bit-vector 128][queue pos 128][queue hash 128][B hashes bcap][pos1 cap][pos2 cap][tv cap+2] | 16-bit: [m cap+2][p cap]
    uint32_t *s_bv = lds, *s_qp = lds + 128, *s_qh = lds + 256;
    uint32_t *s_bh = lds + PS_EXTRA;
    uint32_t cap;
    uint32_t *pos1, *pos2, *tv;
    idx_t *m, *pp;
    if (COUNT) {
        cap = 0; pos1 = pos2 = tv = nullptr; m = pp = nullptr;
    } else if (BIG) {
        cap = (uint32_t)A.gstride;
        uint32_t *g = A.gscratch + (uint64_t)slot * 5 * (A.gstride + 2);
        pos1 = g; pos2 = pos1 + cap + 2; tv = pos2 + cap + 2; m = (idx_t *)(tv + cap + 2); pp = (idx_t *)(tv + 2 * (cap + 2));
    } else {
        cap = A.mcap;
        pos1 = s_bh + A.bcap; pos2 = pos1 + cap; tv = pos2 + cap; m = (idx_t *)(tv + cap + 2); pp = m + cap + 2;
    }
    const bool b_lds = nS <= A.bcap;
    const bool pre = A.k >= 6 && nS > 0;
    const int sh = 2 * (A.k - 6);
    // everything the walk needs is requested before anything is waited for: the searched list, its 6-mer vector, the first batch
    // of the walked list
    constexpr int WB = 8;                               // walk steps per batch of loads: one round trip per 512 walked k-mers
    uint32_t hv[WB], pv[WB];
    auto fetch_walk = [&](const uint32_t base) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            const uint32_t w1 = base + 64u * (uint32_t)u + (uint32_t)lane;
            pv[u] = w1; hv[u] = 0;
            if (w1 < nW) { hv[u] = wh[w1]; if (swp) pv[u] = wp[w1]; }
        }
    };
    if (nS > 0) fetch_walk(0);
    if (b_lds) for (uint32_t t = lane; t < nS; t += 64) s_bh[t] = bh_g[t];
    // A k-mer of i can only occur in j if its leading 6-mer is in j's 4096-bit vector (k >= 6): the positions
    // that pass are queued (order kept) and searched 64 at a time, so unrelated reads -- most pairs -- cost
    // about a quarter of the searches.
    if (pre) {
        const uint32_t *bvj = swp ? (const uint32_t *)(A.bv[0] + (uint64_t)ri * 64) : (const uint32_t *)(A.bv[strand] + (uint64_t)rj * 64);
        for (uint32_t t = lane; t < 128; t += 64) s_bv[t] = bvj[t];
    }
    ps_sync<BIG>();

    // ---- stage 1: matches in (pos1,pos2) order --------------------------------------
    uint32_t total = 0;
    auto walk = [&](const uint32_t *bh) __attribute__((always_inline)) {
        auto drain = [&](const uint32_t count) __attribute__((always_inline)) {      // search queue entries [0, count), count <= 64
            uint32_t p1 = 0, cnt = 0, lo = 0;
            if ((uint32_t)lane < count) {
                p1 = s_qp[lane];
                const uint32_t h = s_qh[lane];
                uint32_t a = 0, b = nS;                    // lower_bound
                while (a < b) {
                    uint32_t mid = (a + b) >> 1;
                    if (bh[mid] < h) a = mid + 1; else b = mid;
                }
                lo = a;
                uint32_t hi = lo;
                while (hi < nS && bh[hi] == h) ++hi;
                cnt = hi - lo;
            }
            const uint32_t incl = wave_incl_scan(cnt, lane);
            const uint32_t tot = __shfl(incl, 63, 64);
            if (!COUNT && total + tot <= cap) {
                // the position of a match in the SEARCHED read is left as its index in the sorted list; one gather after the walk
                const uint32_t at = total + incl - cnt;
                if (swp) for (uint32_t t = 0; t < cnt; ++t) { pos1[at + t] = lo + t; pos2[at + t] = p1; }
                else for (uint32_t t = 0; t < cnt; ++t) { pos1[at + t] = p1; pos2[at + t] = lo + t; }
            }
            total += tot;
        };
        uint32_t qn = 0;
        auto step = [&](const uint32_t base, const uint32_t h, const uint32_t p1) __attribute__((always_inline)) {
            bool keep = false;
            if (base + (uint32_t)lane < nW) {
                const uint32_t six = h >> sh;
                keep = !pre || ((s_bv[six >> 5] >> (six & 31)) & 1u);
            }
            const unsigned long long mask = __ballot(keep);
            if (keep) {
                const uint32_t at = qn + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                s_qp[at] = p1; s_qh[at] = h;
            }
            qn += (uint32_t)__popcll(mask);
            ps_sync<false>();
            if (qn >= 64) {
                drain(64);
                const uint32_t rem = qn - 64;
                uint32_t tp = 0, th = 0;
                if ((uint32_t)lane < rem) { tp = s_qp[64 + lane]; th = s_qh[64 + lane]; }
                ps_sync<false>();
                if ((uint32_t)lane < rem) { s_qp[lane] = tp; s_qh[lane] = th; }
                ps_sync<false>();
                qn = rem;
            }
        };
        for (uint32_t base = 0; base < nW && nS > 0; base += 64u * WB) {
#pragma nounroll
            for (uint32_t u = 0; u < (uint32_t)WB && base + 64u * u < nW; ++u) {      // one copy of the step: the batch rotates through hv[0]
                step(base + 64u * u, hv[0], pv[0]);
#pragma unroll
                for (int i = 0; i + 1 < WB; ++i) { hv[i] = hv[i + 1]; pv[i] = pv[i + 1]; }
            }
            if (base + 64u * WB < nW) fetch_walk(base + 64u * WB);          // the next batch lands in the registers this one has left
        }
        if (qn) drain(qn);
    };
    if (b_lds) walk(s_bh); else walk(bh_g);
    if (COUNT) {
        if (lane == 0) A.res[pr] = (int32_t)total;
        return;
    }
    int32_t *out = A.res + (uint64_t)pr * 4;
    if (total > cap) {                                  // oversize: host reruns with global scratch
        if (lane == 0) { out[0] = INT32_MIN; out[1] = 0; out[2] = 0; out[3] = (int32_t)total; A.var[pr] = 0.0; }
        return;
    }
    ps_sync<BIG>();
    {   // indices in the searched list -> positions, all in flight together
        uint32_t *px = swp ? pos1 : pos2;
        for (uint32_t t = lane; t < total; t += 64) px[t] = bp_g[px[t]];
    }
    ps_sync<BIG>();
    if (swp && total > 1) {
        // matches of the swapped walk come in j's hash order: back into the reference's (pos1, pos2) order (kmer.cpp:65, a
        // lexicographic sort of distinct pairs) with a bitonic network over the two arrays.  The network is the all-ascending
        // form (a merge starts with the flip t ^ (k2 - 1), then halves t ^ j2): every comparator moves the smaller pair to the
        // lower index, so the entries beyond `total` behave as +infinity WITHOUT existing -- a comparator whose partner lies
        // beyond `total` is skipped.  Nothing outside [0, total) is read or written (the arrays hold `cap` entries back to back:
        // padding to the next power of two ran into pos2 / tv for 256 < total <= 400).
        uint32_t P2 = 2;
        while (P2 < total) P2 <<= 1;
        for (uint32_t k2 = 2; k2 <= P2; k2 <<= 1) {
            for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                const uint32_t flip = j2 == (k2 >> 1) ? k2 - 1u : j2;
                for (uint32_t t = lane; t < total; t += 64) {
                    const uint32_t u = t ^ flip;
                    if (u > t && u < total) {
                        const uint32_t a1 = pos1[t], a2 = pos2[t], b1 = pos1[u], b2 = pos2[u];
                        if (a1 > b1 || (a1 == b1 && a2 > b2)) { pos1[t] = b1; pos2[t] = b2; pos1[u] = a1; pos2[u] = a2; }
                    }
                }
                ps_sync<BIG>();
            }
        }
    }

    // ---- stage 2: similarity.cpp:10-31 patience LIS (strict on pos2, ceil-mid search) ---
    // The tails tv[1..l] are strictly increasing, so the reference's binary search returns
    // lo = 1 + #{idx <= l : tv[idx] < x}: the wave counts that with one read per 64 tails instead of
    // ~log2(l) dependent reads by one lane.  The elements themselves stay sequential.
    const int M = (int)total;
    const int k = A.k;
    int l = 0;
    // Matches of two reads of one gene come in co-linear runs, so nearly every element is larger than the last tail and
    // simply extends the longest chain: that case needs no search and no LDS read (the last tail and its element index
    // are carried in scalars, the elements are fetched 64 at a time and handed out by readlane).
    // Round 3: the runs are taken 64 elements at a time -- lane t tests x_t against x_(t-1) (the tail, had all elements before it
    // extended the chain), the length of the leading run of passes comes from one ballot, and the run's tails, elements and
    // back pointers are written by its lanes together.  Only an element that does not extend the chain takes the search below.
    uint32_t tail_last = 0, m_last = 0;                  // tv[l], m[l]
    bool searched = false;                               // some element did NOT extend the longest chain
    if (lane == 0) m[0] = 0;
    for (int base = 0; base < M; base += 64) {
        const uint32_t xv = base + lane < M ? pos2[base + lane] : 0u;
        const uint32_t xsh = (uint32_t)ps_shr1((int)xv);
        const int nb = min(64, M - base);
        int t0 = 0;
        while (t0 < nb) {
            const uint32_t before = lane == t0 ? tail_last : xsh;
            const bool ext = lane >= t0 && lane < nb && (xv > before || (l == 0 && lane == t0));
            const unsigned long long rest = ~(__ballot(ext) >> t0);                  // bit u: element t0 + u does NOT extend
            const int run = rest ? (int)__builtin_ctzll(rest) : 64 - t0;
            if (run > 0) {
                if (lane >= t0 && lane < t0 + run) {
                    const uint32_t i = (uint32_t)(base + lane), rk = (uint32_t)(l + 1 + lane - t0);
                    pp[i] = (idx_t)(lane == t0 ? m_last : i - 1u);
                    m[rk] = (idx_t)i; tv[rk] = xv;
                }
                l += run;
                tail_last = (uint32_t)__builtin_amdgcn_readlane((int)xv, t0 + run - 1);
                m_last = (uint32_t)(base + t0 + run - 1);
                t0 += run;
            }
            if (t0 < nb) {
                const int i = base + t0;
                const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)xv, t0);
                searched = true;
                ps_sync<BIG>();                            // the tails and elements written so far, before the wave reads them
                int lt = 0;
                for (int b2 = 0; b2 < l; b2 += 64) {
                    const int idx = b2 + (int)lane + 1;
                    const bool f = idx <= l && tv[idx] < x;
                    lt += (int)__popcll(__ballot(f));
                }
                const int lo = lt + 1;                     // <= l here
                if (lane == 0) { pp[i] = m[lo - 1]; m[lo] = (idx_t)i; tv[lo] = x; }
                if (lo == l) { tail_last = x; m_last = (uint32_t)i; }
                ++t0;
            }
        }
    }
    ps_sync<BIG>();
    int bases = 0, hc = 0, nd = 0;
    double variance = 0.0;
    if (l > 0) {
        // :37-44 chain reconstruction; chain indices overwrite m[0..l-1] (m[l] is read first).  When every element extended
        // the chain (no search ever ran) the chain is the match list itself.
        if (!searched) {
            for (int t = lane; t < l; t += 64) m[t] = (idx_t)t;
        } else if (lane == 0) {
            uint32_t cur = m[l];
            for (int i = l - 1; i >= 0; --i) { const uint32_t nx = pp[cur]; m[i] = (idx_t)cur; cur = nx; }
        }
        ps_sync<BIG>();
        // :52-85 walk; distances go to tv[] (free now).  The state (last KEPT element, previous chain element) is
        // sequential, the chain's positions are not: 64 of them are gathered per step and handed out by readlane.
        // The reference's double sum of the distances is a sum of small integers, exact in any order: kept as int64.
        // Round 3: 64 chain elements per step.  Lane t tests its element against element t-1 as the last kept one (true if t-1 was
        // kept); the leading run of passes is exact by induction from the carried state, its lanes add their own terms to
        // per-lane sums (integers: any order) and write their distances; the element after the run failed against the true
        // state, so it only moves the chain on.
        int kf = (int)pos1[m[0]], ks = (int)pos2[m[0]];          // last KEPT element
        int prev_s = ks;                                         // previous CHAIN element (.second)
        int vb = 0, vh = 0;                                      // this lane's share of bases / hc
        long long vs = 0;                                        // ... and of the distance sum
        for (int base = 1; base < l; base += 64) {
            const int idx = base + (int)lane;
            int fv = 0, sv = 0;
            if (idx < l) { const uint32_t e = m[idx]; fv = (int)pos1[e]; sv = (int)pos2[e]; }
            const int nb = min(64, l - base);
            const int fsh = ps_shr1(fv), ssh = ps_shr1(sv);
            const int s_chain = lane == 0 ? prev_s : ssh;        // .second of the previous chain element, kept or not
            int t0 = 0;
            while (t0 < nb) {
                const int d1 = fv - (lane == t0 ? kf : fsh), d2 = sv - (lane == t0 ? ks : ssh);
                const bool keep = lane >= t0 && lane < nb && ((d1 < k && d2 < k) || (d1 >= k && d2 >= k));
                const unsigned long long rest = ~(__ballot(keep) >> t0);
                const int run = rest ? (int)__builtin_ctzll(rest) : 64 - t0;
                if (run > 0) {
                    if (lane >= t0 && lane < t0 + run) {
                        const int ex = k - (sv - s_chain);
                        const int cb = ex > 0 ? k - ex : k;
                        const int dist = d2 - d1;
                        tv[nd + lane - t0] = (uint32_t)dist;
                        vb += cb; vs += dist;
                        if (dist < 10) vh += cb;
                    }
                    nd += run;
                    kf = __builtin_amdgcn_readlane(fv, t0 + run - 1); ks = __builtin_amdgcn_readlane(sv, t0 + run - 1);
                    t0 += run;
                }
                if (t0 < nb) ++t0;
            }
            prev_s = __builtin_amdgcn_readlane(sv, nb - 1);
        }
        for (int d = 32; d > 0; d >>= 1) {
            vb += __shfl_xor(vb, d, 64); vh += __shfl_xor(vh, d, 64);
            vs += (long long)__shfl_xor((unsigned long long)vs, d, 64);
        }
        bases = k + vb; hc = k + vh;
        const long long isum = vs;
        ps_sync<BIG>();
        // utils.cpp:36-55: the two sums over the deviations keep the reference's order
        if (nd > 0) {
            const double sum = (double)isum;
            const double mean = sum / (double)nd;
            double ss = 0.0, comp = 0.0;
            for (int base = 0; base < nd; base += 64) {
                const int idx = base + (int)lane;
                const double dv = idx < nd ? (double)(int)tv[idx] - mean : 0.0;
                const int dlo = (int)(uint32_t)__double_as_longlong(dv), dhi = (int)(uint32_t)((unsigned long long)__double_as_longlong(dv) >> 32);
                const int nb = min(64, nd - base);
                for (int t = 0; t < nb; ++t) {
                    const unsigned long long bits = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(dlo, t) |
                                                    ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(dhi, t) << 32);
                    const double d = __longlong_as_double((long long)bits);
                    ss += d * d;
                    comp += d;
                }
            }
            variance = (ss - comp * comp / (double)nd) / (double)(nd - 1);
        }
    }
    if (lane != 0) return;
    out[0] = bases; out[1] = hc; out[2] = nd; out[3] = M;
    A.var[pr] = variance;
}

// LDS of the full pass: 400 matches (longer match lists take the oversize pass) keep a block at 16 KB = ten pairs per CU
#define PS_MCAP 400
static inline size_t ps_full_shm(uint32_t bcap, uint32_t mcap) { return (PS_EXTRA + (size_t)bcap + 3 * (size_t)mcap + 2) * 4 + (2 * (size_t)mcap + 2) * 2 + 4; }

int launch_pair_score(rattle_ctx *ctx, uint32_t n_pairs) {
    if (n_pairs == 0) return 0;
    read_index &X = ctx->idx;
    ps_args A;
    A.uh = X.uh.p; A.koff = X.koff.p;
    A.kh[0] = X.kh[0].p; A.kp[0] = X.kp[0].p; A.kh[1] = X.kh[1].p; A.kp[1] = X.kp[1].p;
    A.bv[0] = X.bv[0].p; A.bv[1] = X.bv[1].p;
    A.pi = ctx->d_pi.p; A.pj = ctx->d_pj.p; A.ps = ctx->d_ps.p;
    A.n_pairs = n_pairs; A.k = X.k;
    A.bcap = 2048; A.mcap = PS_MCAP;
    A.res = ctx->d_res.p; A.var = ctx->d_var.p;
    A.gscratch = nullptr; A.gstride = 0; A.remap = nullptr;
    size_t shm = ps_full_shm(A.bcap, A.mcap);
    // algorithmic bytes are accounted by the caller (needs the pair list on the host)
    ktimer T(ctx, K_SCORE, 0);
    hipLaunchKernelGGL((pair_score_kernel<false, false>), dim3(n_pairs), dim3(64), shm, ctx->stream, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("pair_score launch: ") + hipGetErrorString(e)); return RATTLE_ERR_HIP; }
    return 0;
}

// pairs in ctx->d_pi/d_pj/d_ps; |common| of each pair into ctx->d_res[pair] (one int per pair)
int launch_pair_count(rattle_ctx *ctx, uint32_t n_pairs) {
    if (n_pairs == 0) return 0;
    read_index &X = ctx->idx;
    ps_args A;
    A.uh = X.uh.p; A.koff = X.koff.p;
    A.kh[0] = X.kh[0].p; A.kp[0] = X.kp[0].p; A.kh[1] = X.kh[1].p; A.kp[1] = X.kp[1].p;
    A.bv[0] = X.bv[0].p; A.bv[1] = X.bv[1].p;
    A.pi = ctx->d_pi.p; A.pj = ctx->d_pj.p; A.ps = ctx->d_ps.p;
    A.n_pairs = n_pairs; A.k = X.k;
    A.bcap = 2048; A.mcap = 0;
    A.res = ctx->d_res.p; A.var = nullptr;
    A.gscratch = nullptr; A.gstride = 0; A.remap = nullptr;
    size_t shm = (PS_EXTRA + A.bcap + 8) * 4;
    ktimer T(ctx, K_SCORE, 0);
    hipLaunchKernelGGL((pair_score_kernel<true, false>), dim3(n_pairs), dim3(64), shm, ctx->stream, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("pair_count launch: ") + hipGetErrorString(e)); return RATTLE_ERR_HIP; }
    return 0;
}

// Second pass for pairs whose match count exceeded the LDS capacity: same kernel, arrays
// in a global scratch slab sized for the largest count.  `slots` (host) lists pair indices.
int launch_pair_score_oversize(rattle_ctx *ctx, const std::vector<uint32_t> &slots, uint32_t max_matches) {
    if (slots.empty()) return 0;
    read_index &X = ctx->idx;
    dbuf<uint32_t> d_remap;
    RT_TRY(d_remap.reserve(slots.size()));
    RT_HIP(hipMemcpyAsync(d_remap.p, slots.data(), slots.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    uint64_t stride = max_matches;
    // bound the slab: process in chunks of at most ~2 GiB
    uint64_t per = 5 * (stride + 2) * 4;
    uint32_t chunk = (uint32_t)std::max<uint64_t>(1, (2ull << 30) / per);
    RT_TRY(ctx->d_scratch.reserve((size_t)std::min<uint64_t>(chunk, slots.size()) * 5 * (stride + 2)));
    int rc = 0;
    for (size_t b = 0; b < slots.size() && rc == 0; b += chunk) {
        uint32_t m = (uint32_t)std::min<size_t>(chunk, slots.size() - b);
        ps_args A;
        A.uh = X.uh.p; A.koff = X.koff.p;
        A.kh[0] = X.kh[0].p; A.kp[0] = X.kp[0].p; A.kh[1] = X.kh[1].p; A.kp[1] = X.kp[1].p;
        A.bv[0] = X.bv[0].p; A.bv[1] = X.bv[1].p;
        A.pi = ctx->d_pi.p; A.pj = ctx->d_pj.p; A.ps = ctx->d_ps.p;
        A.n_pairs = m; A.k = X.k;
        A.bcap = 2048; A.mcap = 0;
        A.res = ctx->d_res.p; A.var = ctx->d_var.p;
        A.gscratch = ctx->d_scratch.p; A.gstride = stride; A.remap = d_remap.p + b;
        ktimer T(ctx, K_SCORE, 0);
        hipLaunchKernelGGL((pair_score_kernel<false, true>), dim3(m), dim3(64), (PS_EXTRA + A.bcap + 8) * 4, ctx->stream, A);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error(std::string("pair_score(oversize) launch: ") + hipGetErrorString(e)); rc = RATTLE_ERR_HIP; }
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    d_remap.release();
    if (rc) return rc;
    if (e != hipSuccess) { set_error(std::string("pair_score(oversize): ") + hipGetErrorString(e)); return RATTLE_ERR_HIP; }
    return 0;
}

}  // namespace rattle
