// Kernel B, count pass, seed-major: |common| of every pair that passed the bit-vector filter.
//
// |common| (kmer.cpp:45-67: the full cross product on repeated hashes) = sum over the candidate's k-mers b of the
// number of times hash(b) occurs in the seed.  The pairs are sorted by seed first; a workgroup then turns the seed's
// k-mer set into a direct-address bit set in LDS (4^k <= 2^20 bits = 128 KB for k <= 10), and each of its 16 wavefronts
// streams the hash lists of that seed's candidates past it: one coalesced load and one LDS bit test per k-mer, no
// search.  Hashes that occur more than once in the seed are kept, once per extra occurrence, in a short list that is
// consulted only on a hit, so the count is exact.  For k > 10 the hash is folded to 20 bits: the count becomes an
// UPPER bound of |common|, which is all the caller's exact rejection test needs (cluster_driver.hip: pairs that pass
// it go through the reference's full comparison).  The same holds when the repeat list overflows (low-complexity
// seeds): every hit is charged the whole list.
#include <cstring>

#include "common.h"

namespace rattle {

#define PC_THREADS 1024
#define PC_CHUNK 512            // pairs per workgroup (one or a few seeds' runs)
#define PC_REP 2048             // repeat list entries

struct pc_args {
    const uint32_t *surv;        // [n][2] (seed_slot << 1 | strand, cand_slot), sorted by seed slot
    uint32_t n;
    const uint32_t *seed_ids, *cand_ids;
    const uint32_t *uh;          // forward hashes of every read, position order
    const uint32_t *kh[2];       // per strand: hash lists
    const uint64_t *koff;
    int k;
    int32_t *res;                // [n] count per pair
};

__global__ __launch_bounds__(PC_THREADS) void pair_count_seed_kernel(pc_args A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t s_nrep, s_run_end;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int bits = 2 * A.k < 20 ? 2 * A.k : 20;
    const uint32_t nwords = bits > 5 ? 1u << (bits - 5) : 1u;
    const bool folded = 2 * A.k > 20;
    uint32_t *set = lds, *rep = lds + nwords;
    auto fold = [&](uint32_t h) { return folded ? (h ^ (h >> 20)) & 0xFFFFFu : h; };
    const uint32_t p0 = blockIdx.x * PC_CHUNK, p1 = min(A.n, p0 + PC_CHUNK);
    uint32_t p = p0;
    while (p < p1) {                                          // one run of equal seed per turn
        const uint32_t seed_slot = A.surv[2 * (uint64_t)p] >> 1;
        if (tid == 0) { s_run_end = p1; s_nrep = 0; }
        for (uint32_t w = tid * 4; w < nwords; w += PC_THREADS * 4) *(uint4 *)(set + w) = make_uint4(0, 0, 0, 0);
        if (nwords < 4 && tid < nwords) set[tid] = 0;
        __syncthreads();
        for (uint32_t q = p + 1 + tid; q < p1; q += PC_THREADS)
            if ((A.surv[2 * (uint64_t)q] >> 1) != seed_slot && (A.surv[2 * (uint64_t)(q - 1)] >> 1) == seed_slot) atomicMin(&s_run_end, q);
        const uint32_t ri = A.seed_ids[seed_slot];
        const uint32_t nA = (uint32_t)(A.koff[ri + 1] - A.koff[ri]);
        const uint32_t *__restrict__ ah = A.uh + A.koff[ri];
        for (uint32_t t = tid; t < nA; t += PC_THREADS) {
            const uint32_t f = fold(ah[t]), bit = 1u << (f & 31u);
            if (atomicOr(&set[f >> 5], bit) & bit) {          // one more occurrence of a value already in the set
                const uint32_t at = atomicAdd(&s_nrep, 1u);
                if (at < PC_REP) rep[at] = f;
            }
        }
        __syncthreads();
        const uint32_t nrep = s_nrep, run_end = s_run_end;
        const bool overflow = nrep > PC_REP;
        // wave w takes pairs p + w, p + w + 16, ...; the lanes fetch the descriptors of 64 of them at once (pair -> candidate ->
        // list offsets is three dependent loads: paid once per batch instead of once per pair) and hand them out by readlane
        constexpr uint32_t NWV = PC_THREADS / 64;
        for (uint32_t qb = p + wave; qb < run_end; qb += NWV * 64) {
            const uint32_t myq = qb + NWV * lane;
            uint32_t m_strand = 0, m_n = 0, m_lo = 0, m_hi = 0;
            if (myq < run_end) {
                const uint32_t a = A.surv[2 * (uint64_t)myq], c = A.surv[2 * (uint64_t)myq + 1];
                const uint32_t rj = A.cand_ids[c];
                const uint64_t o0 = A.koff[rj], o1 = A.koff[rj + 1];
                m_strand = a & 1u; m_n = (uint32_t)(o1 - o0); m_lo = (uint32_t)o0; m_hi = (uint32_t)(o0 >> 32);
            }
            const uint32_t nb = min(64u, (run_end - qb + NWV - 1) / NWV);
            for (uint32_t i = 0; i < nb; ++i) {
                const uint32_t strand = (uint32_t)__builtin_amdgcn_readlane((int)m_strand, (int)i);
                const uint32_t nB = (uint32_t)__builtin_amdgcn_readlane((int)m_n, (int)i);
                const uint64_t o0 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)m_lo, (int)i) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)m_hi, (int)i) << 32);
                const uint32_t *__restrict__ bh = A.kh[strand] + o0;
                uint32_t cnt = 0;
                auto test = [&](uint32_t h) {
                    const uint32_t f = fold(h);
                    if ((set[f >> 5] >> (f & 31u)) & 1u) {
                        ++cnt;
                        if (overflow) cnt += nrep;
                        else for (uint32_t r = 0; r < nrep; ++r) cnt += rep[r] == f ? 1u : 0u;
                    }
                };
                uint32_t t = lane;
                for (; t + 448 < nB; t += 512) {                  // eight loads in flight per lane
                    const uint32_t h0 = bh[t], h1 = bh[t + 64], h2 = bh[t + 128], h3 = bh[t + 192], h4 = bh[t + 256], h5 = bh[t + 320], h6 = bh[t + 384], h7 = bh[t + 448];
                    test(h0); test(h1); test(h2); test(h3); test(h4); test(h5); test(h6); test(h7);
                }
                for (; t + 192 < nB; t += 256) {
                    const uint32_t h0 = bh[t], h1 = bh[t + 64], h2 = bh[t + 128], h3 = bh[t + 192];
                    test(h0); test(h1); test(h2); test(h3);
                }
                for (; t < nB; t += 64) test(bh[t]);
                for (int d = 32; d; d >>= 1) cnt += __shfl_down(cnt, d, 64);
                if (lane == 0) A.res[qb + NWV * i] = (int32_t)min(cnt, 0x7FFFFFFFu);
            }
        }
        __syncthreads();                                      // the set is rebuilt for the next run
        p = run_end;
    }
}

// ---- survivors grouped by seed: a counting sort (round 3: hand-written; round 2 called rocPRIM's device radix sort) -------------
// Only the grouping matters -- the count kernel above walks runs of equal seed, nothing reads the order inside a run -- so one
// histogram, one scan and one scatter do: a workgroup counts the seeds of its 16 K entries in LDS, adds its counts to the global
// histogram (one atomic per workgroup and seed present, not per entry), and in the scatter reserves its share of every seed's
// range the same way and places its entries with LDS atomics.  Evaluations with more seeds than the LDS table holds (many small
// rectangles of `--iso`; those rarely take the seed-major form) use one global atomic per entry.
#define SBS_CHUNK 16384u
#define SBS_LDS 8192u

__global__ __launch_bounds__(256) void sbs_hist_kernel(const uint2 *__restrict__ surv, uint32_t n, uint32_t ns, uint32_t *__restrict__ count) {
    extern __shared__ uint32_t sbs_h[];
    const bool lds = ns <= SBS_LDS;
    const uint32_t b0 = blockIdx.x * SBS_CHUNK, b1 = min(n, b0 + SBS_CHUNK);
    if (lds) {
        for (uint32_t t = threadIdx.x; t < ns; t += 256) sbs_h[t] = 0;
        __syncthreads();
    }
    for (uint32_t i = b0 + threadIdx.x; i < b1; i += 256) {
        const uint32_t sd = surv[i].x >> 1;
        if (lds) atomicAdd(&sbs_h[sd], 1u); else atomicAdd(&count[sd], 1u);
    }
    if (lds) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < ns; t += 256) { const uint32_t c = sbs_h[t]; if (c) atomicAdd(&count[t], c); }
    }
}

// exclusive scan of count[0 .. ns) in place (one workgroup; ns is a few hundred to a few hundred thousand)
__global__ __launch_bounds__(1024) void sbs_scan_kernel(uint32_t *__restrict__ count, uint32_t ns) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < ns; b0 += 1024) {
        const uint32_t i = b0 + tid;
        const uint32_t v = i < ns ? count[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= (uint32_t)d) incl += o; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t base = carry;
        for (uint32_t w = 0; w < wave; ++w) base += wtot[w];
        if (i < ns) count[i] = base + incl - v;
        __syncthreads();
        if (tid == 1023) carry = base + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void sbs_scatter_kernel(const uint2 *__restrict__ surv, uint32_t n, uint32_t ns, uint32_t *__restrict__ cursor,
                                                          uint2 *__restrict__ out) {
    extern __shared__ uint32_t sbs_h[];
    const bool lds = ns <= SBS_LDS;
    const uint32_t b0 = blockIdx.x * SBS_CHUNK, b1 = min(n, b0 + SBS_CHUNK);
    if (lds) {
        for (uint32_t t = threadIdx.x; t < ns; t += 256) sbs_h[t] = 0;
        __syncthreads();
        for (uint32_t i = b0 + threadIdx.x; i < b1; i += 256) atomicAdd(&sbs_h[surv[i].x >> 1], 1u);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < ns; t += 256) { const uint32_t c = sbs_h[t]; if (c) sbs_h[t] = atomicAdd(&cursor[t], c); }      // this workgroup's share of seed t's range
        __syncthreads();
    }
    for (uint32_t i = b0 + threadIdx.x; i < b1; i += 256) {
        const uint2 e = surv[i];
        const uint32_t sd = e.x >> 1;
        const uint32_t at = lds ? atomicAdd(&sbs_h[sd], 1u) : atomicAdd(&cursor[sd], 1u);
        out[at] = e;
    }
}

// group the survivor list (ctx->d_surv, n entries of two words) by seed slot; n_seeds bounds the slot
int sort_survivors_by_seed(rattle_ctx *ctx, uint32_t n, uint64_t n_seeds) {
    if (n < 2) return 0;
    if (n_seeds >= 0x7FFFFFFFull) { set_error("survivor sort: too many seeds"); return RATTLE_ERR_ARG; }
    const uint32_t ns = (uint32_t)n_seeds;
    RT_TRY(ctx->d_surv2.reserve((size_t)n * 2));
    RT_TRY(ctx->d_sort_tmp.reserve(((size_t)ns + 1) * 4));
    uint32_t *count = (uint32_t *)ctx->d_sort_tmp.p;
    hipStream_t st = ctx->stream;
    RT_HIP(hipMemsetAsync(count, 0, ((size_t)ns + 1) * 4, st));
    const uint32_t blocks = (n + SBS_CHUNK - 1) / SBS_CHUNK;
    const size_t shm = ns <= SBS_LDS ? (size_t)ns * 4 : 0;
    hipLaunchKernelGGL(sbs_hist_kernel, dim3(blocks), dim3(256), shm, st, (const uint2 *)ctx->d_surv.p, n, ns, count);
    hipLaunchKernelGGL(sbs_scan_kernel, dim3(1), dim3(1024), 0, st, count, ns);
    hipLaunchKernelGGL(sbs_scatter_kernel, dim3(blocks), dim3(256), shm, st, (const uint2 *)ctx->d_surv.p, n, ns, count, (uint2 *)ctx->d_surv2.p);
    RT_HIP(hipGetLastError());
    ctx->d_surv.swap(ctx->d_surv2);
    return 0;
}

// pairs in ctx->d_surv (sorted by seed), seeds / candidates in ctx->d_seed / d_cand; count (or its upper bound, see above) of
// each pair into ctx->d_res[pair]
int launch_pair_count_seed(rattle_ctx *ctx, uint32_t n_pairs) {
    if (n_pairs == 0) return 0;
    read_index &X = ctx->idx;
    pc_args A;
    A.surv = ctx->d_surv.p; A.n = n_pairs; A.seed_ids = ctx->d_seed.p; A.cand_ids = ctx->d_cand.p;
    A.uh = X.uh.p; A.kh[0] = X.kh[0].p; A.kh[1] = X.kh[1].p; A.koff = X.koff.p; A.k = X.k;
    A.res = ctx->d_res.p;
    const int bits = 2 * X.k < 20 ? 2 * X.k : 20;
    const size_t shm = ((bits > 5 ? (size_t)1 << (bits - 5) : 4) + PC_REP) * 4;
    // the attribute is per device (rattle --devices runs one host thread per GPU): set it on every call, it costs nothing
    if (shm > 60 * 1024) RT_HIP(hipFuncSetAttribute((const void *)pair_count_seed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    ktimer T(ctx, K_SCORE, 0);
    hipLaunchKernelGGL(pair_count_seed_kernel, dim3((n_pairs + PC_CHUNK - 1) / PC_CHUNK), dim3(PC_THREADS), shm, ctx->stream, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("pair_count_seed launch: ") + hipGetErrorString(e)); return RATTLE_ERR_HIP; }
    return 0;
}

}  // namespace rattle
