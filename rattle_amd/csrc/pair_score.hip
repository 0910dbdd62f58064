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

// COUNT: stage 1 only, and only the NUMBER of matches |common| is kept (res[pr] = count; no match arrays in LDS, so three
// times as many pairs are resident per CU).  The greedy driver runs it over every pair that passed the bit-vector filter:
// a pair can only be accepted if bases / min_len >= t_s (cluster.cpp:23-27) and bases <= k * |LIS| <= k * |common|, so a
// pair with k * |common| below that bar is rejected exactly without the patience search; the few others go through the
// full kernel.
template <bool COUNT>
__global__ __launch_bounds__(64) void pair_score_kernel(ps_args A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const uint32_t slot = blockIdx.x;
    const uint32_t pr = A.remap ? A.remap[slot] : slot;
    const uint32_t ri = A.pi[pr], rj = A.pj[pr];
    const int strand = A.ps[pr];
    const uint32_t nA = (uint32_t)(A.koff[ri + 1] - A.koff[ri]);
    const uint32_t nB = (uint32_t)(A.koff[rj + 1] - A.koff[rj]);
    // Which list is walked and which is searched.  The reference walks read i (the seed: in length-sorted order never the shorter
    // of the two) and the matches come out in (pos1, pos2) order for free.  With mixed read lengths (config 5: a 50 kb seed
    // against a 150 nt candidate, 330 M such comparisons at 5e5 reads) that is 800 search rounds for a handful of matches, so
    // when i is much the longer the SHORT list is walked (j's sorted list, any order) against i's sorted list, and the few matches
    // are sorted by (pos1, pos2) afterwards -- the same multiset (cross product on repeated hashes) in the same order.  The count
    // pass needs no order at all and always walks the shorter list.
    const bool swp = !A.gscratch && (COUNT ? nA > nB : nA > 4u * nB + 256u);
    const uint32_t nW = swp ? nB : nA, nS = swp ? nA : nB;
    const uint32_t *__restrict__ wh = swp ? A.kh[strand] + A.koff[rj] : A.uh + A.koff[ri];          // walked: hashes (swap: with positions wp)
    const uint32_t *__restrict__ wp = swp ? A.kp[strand] + A.koff[rj] : nullptr;
    const uint32_t *__restrict__ bh_g = swp ? A.kh[0] + A.koff[ri] : A.kh[strand] + A.koff[rj];     // searched: sorted hashes + positions
    const uint32_t *__restrict__ bp_g = swp ? A.kp[0] + A.koff[ri] : A.kp[strand] + A.koff[rj];

    // LDS carve: [B bit-vector 128][queue pos 128][queue hash 128][B hashes bcap][pos1 mcap][pos2 mcap][m mcap+1 (+pad)][tv mcap+1 (+pad)][p mcap]
    uint32_t *s_bv = lds, *s_qp = lds + 128, *s_qh = lds + 256;
    uint32_t *s_bh = lds + PS_EXTRA;
    uint32_t cap;
    uint32_t *pos1, *pos2, *m, *tv, *pp;
    if (COUNT) {
        cap = 0; pos1 = pos2 = m = tv = pp = nullptr;
    } else if (A.gscratch) {
        cap = (uint32_t)A.gstride;
        uint32_t *g = A.gscratch + (uint64_t)slot * 5 * (A.gstride + 2);
        pos1 = g; pos2 = pos1 + cap + 2; m = pos2 + cap + 2; tv = m + cap + 2; pp = tv + cap + 2;
    } else {
        cap = A.mcap;
        pos1 = s_bh + A.bcap; pos2 = pos1 + cap; m = pos2 + cap; tv = m + cap + 2; pp = tv + cap + 2;
    }
    const bool b_lds = nS <= A.bcap;
    if (b_lds) {
        for (uint32_t t = lane; t < nS; t += 64) s_bh[t] = bh_g[t];
        __syncthreads();
    }
    const uint32_t *bh = b_lds ? (const uint32_t *)s_bh : bh_g;

    // ---- stage 1: matches in (pos1,pos2) order --------------------------------------
    // A k-mer of i can only occur in j if its leading 6-mer is in j's 4096-bit vector (k >= 6): the positions
    // that pass are queued (order kept) and searched 64 at a time, so unrelated reads -- most pairs -- cost
    // about a quarter of the searches.
    const bool pre = A.k >= 6 && nS > 0;
    const int sh = 2 * (A.k - 6);
    if (pre) {
        const uint32_t *bvj = swp ? (const uint32_t *)(A.bv[0] + (uint64_t)ri * 64) : (const uint32_t *)(A.bv[strand] + (uint64_t)rj * 64);
        for (uint32_t t = lane; t < 128; t += 64) s_bv[t] = bvj[t];
        __syncthreads();
    }
    uint32_t total = 0;
    auto drain = [&](const uint32_t count) {             // search queue entries [0, count), count <= 64
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
        uint32_t incl = wave_incl_scan(cnt, lane);
        uint32_t tot = __shfl(incl, 63, 64);
        if (!COUNT && total + tot <= cap) {
            uint32_t at = total + incl - cnt;
            if (swp) for (uint32_t t = 0; t < cnt; ++t) { pos1[at + t] = bp_g[lo + t]; pos2[at + t] = p1; }
            else for (uint32_t t = 0; t < cnt; ++t) { pos1[at + t] = p1; pos2[at + t] = bp_g[lo + t]; }
        }
        total += tot;
    };
    uint32_t qn = 0;
    for (uint32_t base = 0; base < nW && nS > 0; base += 64) {
        const uint32_t w1 = base + lane;
        uint32_t p1 = w1;                              // position of the walked k-mer in its read
        bool keep = false;
        uint32_t h = 0;
        if (w1 < nW) {
            h = wh[w1];
            if (swp) p1 = wp[w1];
            const uint32_t six = h >> sh;
            keep = !pre || ((s_bv[six >> 5] >> (six & 31)) & 1u);
        }
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const uint32_t at = qn + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            s_qp[at] = p1; s_qh[at] = h;
        }
        qn += (uint32_t)__popcll(mask);
        __syncthreads();
        if (qn >= 64) {
            drain(64);
            const uint32_t rem = qn - 64;
            uint32_t tp = 0, th = 0;
            if ((uint32_t)lane < rem) { tp = s_qp[64 + lane]; th = s_qh[64 + lane]; }
            __syncthreads();
            if ((uint32_t)lane < rem) { s_qp[lane] = tp; s_qh[lane] = th; }
            __syncthreads();
            qn = rem;
        }
    }
    if (qn) drain(qn);
    if (COUNT) {
        if (lane == 0) A.res[pr] = (int32_t)total;
        return;
    }
    int32_t *out = A.res + (uint64_t)pr * 4;
    if (total > cap) {                                  // oversize: host reruns with global scratch
        if (lane == 0) { out[0] = INT32_MIN; out[1] = 0; out[2] = 0; out[3] = (int32_t)total; A.var[pr] = 0.0; }
        return;
    }
    __syncthreads();
    if (swp && total > 1) {
        // matches of the swapped walk come in j's hash order: back into the reference's (pos1, pos2) order (kmer.cpp:65, a
        // lexicographic sort of distinct pairs) with a bitonic network over the two LDS arrays
        uint32_t P2 = 2;
        while (P2 < total) P2 <<= 1;
        for (uint32_t t = total + lane; t < P2; t += 64) { pos1[t] = 0xFFFFFFFFu; pos2[t] = 0xFFFFFFFFu; }
        __syncthreads();
        for (uint32_t k2 = 2; k2 <= P2; k2 <<= 1) {
            for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                for (uint32_t t = lane; t < P2; t += 64) {
                    const uint32_t u = t ^ j2;
                    if (u > t) {
                        const uint32_t a1 = pos1[t], a2 = pos2[t], b1 = pos1[u], b2 = pos2[u];
                        const bool gt = a1 > b1 || (a1 == b1 && a2 > b2);
                        if (gt == ((t & k2) == 0)) { pos1[t] = b1; pos2[t] = b2; pos1[u] = a1; pos2[u] = a2; }
                    }
                }
                __syncthreads();
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
    uint32_t tail_last = 0, m_last = 0;                  // tv[l], m[l]
    if (lane == 0) m[0] = 0;
    for (int base = 0; base < M; base += 64) {
        const uint32_t xv = base + lane < M ? pos2[base + lane] : 0u;
        const int nb = min(64, M - base);
        for (int t = 0; t < nb; ++t) {
            const int i = base + t;
            const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)xv, t);
            if (l == 0 || tail_last < x) {
                ++l;
                if (lane == 0) { pp[i] = m_last; m[l] = (uint32_t)i; tv[l] = x; }
                tail_last = x; m_last = (uint32_t)i;
                continue;
            }
            __syncthreads();                               // lane 0's tails and elements, before the wave reads them
            int lt = 0;
            for (int b2 = 0; b2 < l; b2 += 64) {
                const int idx = b2 + (int)lane + 1;
                const bool f = idx <= l && tv[idx] < x;
                lt += (int)__popcll(__ballot(f));
            }
            const int lo = lt + 1;                         // <= l here
            if (lane == 0) { pp[i] = m[lo - 1]; m[lo] = (uint32_t)i; tv[lo] = x; }
            if (lo == l) { tail_last = x; m_last = (uint32_t)i; }
        }
    }
    __syncthreads();
    int bases = 0, hc = 0, nd = 0;
    double variance = 0.0;
    if (l > 0) {
        // :37-44 chain reconstruction; chain indices overwrite m[0..l-1] (m[l] is read first)
        if (lane == 0) {
            uint32_t cur = m[l];
            for (int i = l - 1; i >= 0; --i) { uint32_t nx = pp[cur]; m[i] = cur; cur = nx; }
        }
        __syncthreads();
        // :52-85 walk; distances go to tv[] (free now).  The state (last KEPT element, previous chain element) is
        // sequential, the chain's positions are not: 64 of them are gathered per step and handed out by readlane.
        // The reference's double sum of the distances is a sum of small integers, exact in any order: kept as int64.
        int kf = (int)pos1[m[0]], ks = (int)pos2[m[0]];          // last KEPT element
        int prev_s = ks;                                         // previous CHAIN element (.second)
        bases = k; hc = k;
        long long isum = 0;
        for (int base = 1; base < l; base += 64) {
            const int idx = base + (int)lane;
            int fv = 0, sv = 0;
            if (idx < l) { const uint32_t e = m[idx]; fv = (int)pos1[e]; sv = (int)pos2[e]; }
            const int nb = min(64, l - base);
            for (int t = 0; t < nb; ++t) {
                const int f = __builtin_amdgcn_readlane(fv, t), s2 = __builtin_amdgcn_readlane(sv, t);
                const int d1 = f - kf, d2 = s2 - ks;
                if ((d1 < k && d2 < k) || (d1 >= k && d2 >= k)) {
                    bases += k;
                    const int ex = k - (s2 - prev_s);
                    if (ex > 0) bases -= ex;
                    const int dist = d2 - d1;
                    if (lane == 0) tv[nd] = (uint32_t)dist;
                    ++nd;
                    isum += dist;
                    if (dist < 10) { hc += k; if (ex > 0) hc -= ex; }
                    kf = f; ks = s2;
                }
                prev_s = s2;
            }
        }
        __syncthreads();
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

int launch_pair_score(rattle_ctx *ctx, uint32_t n_pairs) {
    if (n_pairs == 0) return 0;
    read_index &X = ctx->idx;
    ps_args A;
    A.uh = X.uh.p; A.koff = X.koff.p;
    A.kh[0] = X.kh[0].p; A.kp[0] = X.kp[0].p; A.kh[1] = X.kh[1].p; A.kp[1] = X.kp[1].p;
    A.bv[0] = X.bv[0].p; A.bv[1] = X.bv[1].p;
    A.pi = ctx->d_pi.p; A.pj = ctx->d_pj.p; A.ps = ctx->d_ps.p;
    A.n_pairs = n_pairs; A.k = X.k;
    A.bcap = 2048; A.mcap = 512;
    A.res = ctx->d_res.p; A.var = ctx->d_var.p;
    A.gscratch = nullptr; A.gstride = 0; A.remap = nullptr;
    size_t shm = (PS_EXTRA + A.bcap + 5 * (size_t)A.mcap + 8) * 4;
    // algorithmic bytes are accounted by the caller (needs the pair list on the host)
    ktimer T(ctx, K_SCORE, 0);
    hipLaunchKernelGGL(pair_score_kernel<false>, dim3(n_pairs), dim3(64), shm, ctx->stream, A);
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
    hipLaunchKernelGGL(pair_score_kernel<true>, dim3(n_pairs), dim3(64), shm, ctx->stream, A);
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
        hipLaunchKernelGGL(pair_score_kernel<false>, dim3(m), dim3(64), (PS_EXTRA + A.bcap + 8) * 4, ctx->stream, A);
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
