// Kernel K: per-read k-mer index.  Replaces extract_kmers_from_read,
// /root/reference/kmer.cpp:6-42 (hash: kmer.hpp:25-40; reverse strand: utils.cpp:15-24).
//
// One 256-thread workgroup per (read, strand).  The read is staged in LDS as 2-bit codes,
// every thread hashes its positions, the (hash, pos) pairs are sorted in LDS (LSD radix sort, kmer_extract_lsd_kernel; the
// bitonic kernel of round 1 is kept for lists beyond 8192 k-mers and as RATTLE_KMER_SORT=bitonic for the parity test) and
// written back as two coalesced SoA streams (hash, pos).  The 4096-bit 6-mer bit-vector is
// built with LDS atomics and written as 64 u64 words.
//
// HBM traffic per read (algorithmic): L bytes read; per strand (L-k)*8 B list + 512 B
// bit-vector written, + (L-k)*4 B position-ordered hashes for the forward strand.
#include <cstring>

#include "common.h"

namespace rattle {

__device__ __forceinline__ uint32_t base_code(uint8_t c) {
    // kmer.hpp:25-31: A=0 C=1 T=2 U=2 G=3; anything else is invalid (flagged).
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'T': case 'U': return 2;
        case 'G': return 3;
    }
    return 4;
}

// Bitonic sort of P (power of two) u64 keys in LDS by a 256-thread block.
__device__ void bitonic_sort_lds(uint64_t *key, uint32_t P) {
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                uint32_t lo = 2 * t - (t & (stride - 1));      // index with bit `stride` clear
                uint32_t hi = lo + stride;
                bool up = (lo & size) == 0;
                uint64_t a = key[lo], b = key[hi];
                if ((a > b) == up) { key[lo] = b; key[hi] = a; }
            }
        }
    }
    __syncthreads();
}

// Global-memory variant for reads whose list does not fit LDS (one block per read/strand).
__device__ void bitonic_sort_global(uint64_t *key, uint64_t P) {
    for (uint64_t size = 2; size <= P; size <<= 1) {
        for (uint64_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint64_t t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                uint64_t lo = 2 * t - (t & (stride - 1));
                uint64_t hi = lo + stride;
                bool up = (lo & size) == 0;
                uint64_t a = key[lo], b = key[hi];
                if ((a > b) == up) { key[lo] = b; key[hi] = a; }
            }
            __threadfence_block();
        }
    }
    __syncthreads();
}

// items: read ids handled by this launch; grid = (n_items, strands).
// P = padded key count (power of two >= max list length in this launch), keys in dynamic LDS
// unless gscratch != nullptr (then keys live at gscratch + blockLinear*P).
__global__ __launch_bounds__(256) void kmer_extract_kernel(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ off,
                                                           const uint64_t *__restrict__ koff, const uint32_t *__restrict__ items,
                                                           int k, uint32_t P, uint32_t Lmax, uint32_t *__restrict__ uh,
                                                           uint32_t *__restrict__ kh0, uint32_t *__restrict__ kp0,
                                                           uint32_t *__restrict__ kh1, uint32_t *__restrict__ kp1,
                                                           uint64_t *__restrict__ bv0, uint64_t *__restrict__ bv1,
                                                           uint32_t *__restrict__ pc0, uint32_t *__restrict__ pc1,
                                                           uint64_t *gscratch, uint32_t *__restrict__ bad_flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t r = items[blockIdx.x];
    const int strand = blockIdx.y;
    const uint64_t o = off[r];
    const uint32_t L = (uint32_t)(off[r + 1] - o);
    const uint32_t nk = L > (uint32_t)k ? L - k : 0;
    const uint32_t nb = L > 6 ? L - 6 : 0;

    // LDS carve: [bitset 128 u32][codes Lmax bytes, padded to 16][keys P u64 (if in LDS)]
    uint32_t *bits = (uint32_t *)smem;
    uint8_t *code = smem + 512;
    uint64_t *key = gscratch ? gscratch + ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * P
                             : (uint64_t *)(smem + 512 + ((Lmax + 15u) & ~15u));

    for (uint32_t t = threadIdx.x; t < 128; t += blockDim.x) bits[t] = 0;
    bool bad = false;
    for (uint32_t p = threadIdx.x; p < L; p += blockDim.x) {
        // forward: code of seq[p]; reverse strand (reverse_complement): complement of seq[L-1-p],
        // complement code = code ^ 2 (A<->T, C<->G, U->A).
        uint32_t c = strand == 0 ? base_code(seq[o + p]) : base_code(seq[o + (L - 1 - p)]);
        if (c > 3) { bad = true; c = 0; }
        code[p] = (uint8_t)(strand == 0 ? c : (c ^ 2u));
    }
    if (bad) atomicOr(bad_flag, 1u);
    __syncthreads();

    // hashes: rolling would serialise; each thread packs k codes (k <= 16).
    const uint32_t kmask = k >= 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) {
        uint64_t kv = ~0ull;
        if (p < nk) {
            uint32_t h = 0;
            for (int i = 0; i < k; ++i) h = (h << 2) | code[p + i];
            h &= kmask;
            kv = ((uint64_t)h << 32) | p;
            if (strand == 0) uh[koff[r] + p] = h;
        }
        key[p] = kv;
    }
    for (uint32_t p = threadIdx.x; p < nb; p += blockDim.x) {       // kmer.cpp:28-36, always 6-mers
        uint32_t h = 0;
        for (int i = 0; i < 6; ++i) h = (h << 2) | code[p + i];
        atomicOr(&bits[h >> 5], 1u << (h & 31));
    }
    if (gscratch) { __threadfence_block(); bitonic_sort_global(key, P); }
    else bitonic_sort_lds(key, P);

    uint32_t *kh = strand == 0 ? kh0 : kh1;
    uint32_t *kp = strand == 0 ? kp0 : kp1;
    const uint64_t ko = koff[r];
    for (uint32_t p = threadIdx.x; p < nk; p += blockDim.x) {
        uint64_t kv = key[p];
        kh[ko + p] = (uint32_t)(kv >> 32);
        kp[ko + p] = (uint32_t)kv;
    }
    uint64_t *bv = strand == 0 ? bv0 : bv1;
    uint32_t *pc = strand == 0 ? pc0 : pc1;
    if (threadIdx.x < 64) {
        uint64_t w = (uint64_t)bits[2 * threadIdx.x] | ((uint64_t)bits[2 * threadIdx.x + 1] << 32);
        bv[(uint64_t)r * 64 + threadIdx.x] = w;
        uint32_t c = __popcll(w);
        for (int s = 32; s > 0; s >>= 1) c += __shfl_xor(c, s, 64);
        if (threadIdx.x == 0) pc[r] = c;
    }
}

// The same index with the list sorted by an LSD radix sort on the 2k hash bits, eight bits per pass, entirely in LDS (round 3:
// hand-written; rounds 1-2 used a bitonic network, then rocPRIM's block_radix_sort).  Keys (hash) and values (position, 16 bits:
// lists of up to 8192 k-mers) ping-pong between two LDS arrays.  A pass: (1) every wavefront histograms the digits of ITS
// contiguous quarter of the list with LDS atomics; (2) thread d scans digit d over (digit, wavefront) -- the start of each
// wavefront's run of each digit; (3) every wavefront walks its quarter in order, 64 keys at a time: the lanes that hold the
// same digit find each other with eight ballots, a lane's slot is its digit's cursor plus the number of such lanes below it,
// and the last of them moves the cursor.  Order is kept at every step, so the sort is stable and equal hashes stay in position
// order = the reference's std::sort on (hash, pos) pairs (kmer.cpp:38-41).  ceil(2k / 8) passes (3 for k = 10, 11) of 2
// barriers + a scan replace the 55 barriers of the bitonic network for a 1 kb read.
__device__ __forceinline__ uint32_t kx_wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}

__global__ __launch_bounds__(256) void kmer_extract_lsd_kernel(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ off,
                                                               const uint64_t *__restrict__ koff, const uint32_t *__restrict__ items,
                                                               int k, uint32_t P, uint32_t Lmax, uint32_t *__restrict__ uh,
                                                               uint32_t *__restrict__ kh0, uint32_t *__restrict__ kp0,
                                                               uint32_t *__restrict__ kh1, uint32_t *__restrict__ kp1,
                                                               uint64_t *__restrict__ bv0, uint64_t *__restrict__ bv1,
                                                               uint32_t *__restrict__ pc0, uint32_t *__restrict__ pc1,
                                                               uint32_t *__restrict__ bad_flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t r = items[blockIdx.x];
    const int strand = blockIdx.y;
    const uint64_t o = off[r];
    const uint32_t L = (uint32_t)(off[r + 1] - o);
    const uint32_t nk = L > (uint32_t)k ? L - k : 0;
    const uint32_t nb = L > 6 ? L - 6 : 0;
    // LDS carve: [bitset 128 u32][codes Lmax bytes, padded to 16][digit cursors 4 x 256][wave totals 16][keys P][keys P][positions P u16][positions P u16]
    uint32_t *bits = (uint32_t *)smem;
    uint8_t *code = smem + 512;
    uint32_t *hist = (uint32_t *)(smem + 512 + ((Lmax + 15u) & ~15u));
    uint32_t *wsum = hist + 1024;
    uint32_t *ka = wsum + 16, *kb = ka + P;
    uint16_t *va = (uint16_t *)(kb + P), *vb = va + P;
    for (uint32_t t = tid; t < 128; t += 256) bits[t] = 0;
    bool bad = false;
    for (uint32_t p = tid; p < L; p += 256) {
        uint32_t c = strand == 0 ? base_code(seq[o + p]) : base_code(seq[o + (L - 1 - p)]);
        if (c > 3) { bad = true; c = 0; }
        code[p] = (uint8_t)(strand == 0 ? c : (c ^ 2u));
    }
    if (bad) atomicOr(bad_flag, 1u);
    __syncthreads();
    const uint32_t kmask = k >= 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    const uint64_t ko = koff[r];
    for (uint32_t p = tid; p < nk; p += 256) {
        uint32_t h = 0;
        for (int j = 0; j < k; ++j) h = (h << 2) | code[p + j];
        h &= kmask;
        if (strand == 0) uh[ko + p] = h;
        ka[p] = h; va[p] = (uint16_t)p;
    }
    for (uint32_t p = tid; p < nb; p += 256) {                // kmer.cpp:28-36, always 6-mers
        uint32_t h = 0;
        for (int j = 0; j < 6; ++j) h = (h << 2) | code[p + j];
        atomicOr(&bits[h >> 5], 1u << (h & 31));
    }
    const uint32_t seg = (nk + 255u) / 256u * 64u;             // a wavefront's quarter, a multiple of 64
    const uint32_t s0 = min(nk, wave * seg), s1 = min(nk, s0 + seg);
    uint32_t *const cur = hist + wave * 256u;
    for (int shift = 0; shift < 2 * k; shift += 8) {
        for (uint32_t t = tid; t < 1024; t += 256) hist[t] = 0;
        __syncthreads();                                      // also: the keys of the previous pass (or the hashes) are in place
        for (uint32_t i = s0 + lane; i < s1; i += 64) atomicAdd(&cur[(ka[i] >> shift) & 255u], 1u);
        __syncthreads();
        {
            const uint32_t c0 = hist[tid], c1 = hist[256 + tid], c2 = hist[512 + tid], c3 = hist[768 + tid];
            const uint32_t tot = c0 + c1 + c2 + c3;
            const uint32_t incl = kx_wave_incl_scan(tot, lane);
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t base = incl - tot;
            for (uint32_t w = 0; w < wave; ++w) base += wsum[w];
            hist[tid] = base; hist[256 + tid] = base + c0; hist[512 + tid] = base + c0 + c1; hist[768 + tid] = base + c0 + c1 + c2;
        }
        __syncthreads();
        for (uint32_t b0 = s0; b0 < s1; b0 += 64) {
            const uint32_t i = b0 + lane;
            const bool valid = i < s1;
            const uint32_t kv = valid ? ka[i] : 0u, vv = valid ? (uint32_t)va[i] : 0u;
            const uint32_t d = (kv >> shift) & 255u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool one = (d >> b) & 1u;
                const unsigned long long bal = __ballot(valid && one);
                peers &= one ? bal : ~bal;
            }
            const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull)), all = (uint32_t)__popcll(peers);
            if (valid) {
                const uint32_t at = cur[d];
                kb[at + below] = kv; vb[at + below] = (uint16_t)vv;
                if (below + 1u == all) cur[d] = at + all;       // the last lane of the digit moves its cursor (after every read of it: LDS keeps a wavefront's order)
            }
        }
        __syncthreads();
        { uint32_t *tk = ka; ka = kb; kb = tk; uint16_t *tv = va; va = vb; vb = tv; }
    }
    uint32_t *kh = strand == 0 ? kh0 : kh1;
    uint32_t *kp = strand == 0 ? kp0 : kp1;
    for (uint32_t q = tid; q < nk; q += 256) { kh[ko + q] = ka[q]; kp[ko + q] = (uint32_t)va[q]; }
    uint64_t *bv = strand == 0 ? bv0 : bv1;
    uint32_t *pc = strand == 0 ? pc0 : pc1;
    if (tid < 64) {
        uint64_t w = (uint64_t)bits[2 * tid] | ((uint64_t)bits[2 * tid + 1] << 32);
        bv[(uint64_t)r * 64 + tid] = w;
        uint32_t c = __popcll(w);
        for (int sft = 32; sft > 0; sft >>= 1) c += __shfl_xor(c, sft, 64);
        if (tid == 0) pc[r] = c;
    }
}

static uint32_t pow2ceil(uint32_t x) {
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

int build_index(rattle_ctx *ctx, const uint8_t *seq, const uint64_t *off, uint32_t n, int k, int both) {
    if (k < 1 || k > 16) { set_error("kmer size must be in [1,16] (main.cpp:223)"); return RATTLE_ERR_ARG; }
    // every argument is checked before the resident index is touched
    if (n && off[0] != 0) { set_error("offsets[0] must be 0"); return RATTLE_ERR_ARG; }
    for (uint32_t i = 0; i < n; ++i) {
        if (off[i + 1] < off[i]) { set_error("offsets must be non-decreasing"); return RATTLE_ERR_ARG; }
        if (off[i + 1] - off[i] > 0x7FFFFFF0ull) { set_error("read too long"); return RATTLE_ERR_ARG; }
    }
    read_index &X = ctx->idx;
    X.n = n; X.k = k; X.both = both ? 1 : 0;
    X.h_off.assign(off, off + n + 1);
    X.h_koff.resize(n + 1);
    X.h_len.resize(n);
    uint64_t ko = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t L = off[i + 1] - off[i];
        X.h_len[i] = (uint32_t)L;
        X.h_koff[i] = ko;
        ko += L > (uint64_t)k ? L - k : 0;
    }
    X.h_koff[n] = ko;
    X.total_bases = off[n] - off[0];
    X.total_kmers = ko;
    if (n == 0) return 0;

    RT_TRY(X.seq.reserve(X.total_bases + 16));
    RT_TRY(X.off.reserve(n + 1));
    RT_TRY(X.koff.reserve(n + 1));
    RT_TRY(X.len.reserve(n));
    RT_TRY(X.uh.reserve(ko + 1));
    const int ns = both ? 2 : 1;
    for (int s = 0; s < ns; ++s) {
        RT_TRY(X.kh[s].reserve(ko + 1));
        RT_TRY(X.kp[s].reserve(ko + 1));
        RT_TRY(X.bv[s].reserve((size_t)n * 64));
        RT_TRY(X.pc[s].reserve(n));
    }
    hipStream_t st = ctx->stream;
    // host buffer, a device-resident gather, or the index's own copy (second index over the same reads: --iso)
    if (seq != X.seq.p) RT_HIP(hipMemcpyAsync(X.seq.p, seq, X.total_bases, hipMemcpyDefault, st));
    RT_HIP(hipMemcpyAsync(X.off.p, off, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(X.koff.p, X.h_koff.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(X.len.p, X.h_len.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, st));

    // Size classes by padded list length; LDS holds keys up to 8192 entries (64 KiB + codes).
    const uint32_t LDS_MAX_P = 8192;
    std::vector<std::vector<uint32_t>> cls(33);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t nk = X.h_len[i] > (uint32_t)k ? X.h_len[i] - k : 0;
        uint32_t P = pow2ceil(nk < 64 ? 64 : nk);
        int c = 0;
        while ((1u << c) < P) ++c;
        cls[c].push_back(i);
    }
    dbuf<uint32_t> d_items;
    dbuf<uint32_t> d_bad;
    dbuf<uint64_t> d_gs;
    RT_TRY(d_items.reserve(n));
    RT_TRY(d_bad.reserve(1));
    RT_HIP(hipMemsetAsync(d_bad.p, 0, sizeof(uint32_t), st));
    int rc = 0;
    {
        uint64_t bytes = X.total_bases + (uint64_t)ns * (ko * 8 + (uint64_t)n * 512) + ko * 4;
        ktimer T(ctx, K_KMER, bytes);
        size_t done = 0;
        for (int c = 0; c <= 32 && rc == 0; ++c) {
            if (cls[c].empty()) continue;
            uint32_t P = 1u << c;
            uint32_t cnt = (uint32_t)cls[c].size();
            uint32_t Lmax = 0;
            for (uint32_t r : cls[c]) Lmax = X.h_len[r] > Lmax ? X.h_len[r] : Lmax;
            hipError_t e = hipMemcpyAsync(d_items.p + done, cls[c].data(), cnt * sizeof(uint32_t), hipMemcpyHostToDevice, st);
            if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = RATTLE_ERR_HIP; break; }
            bool in_lds = P <= LDS_MAX_P;
            size_t shm = 512 + ((Lmax + 15u) & ~15u) + (in_lds ? (size_t)P * 8 : 0);
            if (shm > 160 * 1024) { set_error("read too long for the LDS code buffer"); rc = RATTLE_ERR_ARG; break; }
            // chunk launches so the global scratch for oversize reads stays bounded
            uint32_t chunk = in_lds ? cnt : (uint32_t)std::max<uint64_t>(1, (1ull << 28) / ((uint64_t)P * ns));
            for (uint32_t b = 0; b < cnt && rc == 0; b += chunk) {
                uint32_t m = std::min(chunk, cnt - b);
                uint64_t *gs = nullptr;
                if (!in_lds) {
                    rc = d_gs.reserve((size_t)m * ns * P);
                    if (rc) break;
                    gs = d_gs.p;
                }
                const bool use_bitonic = getenv("RATTLE_KMER_SORT") && !strcmp(getenv("RATTLE_KMER_SORT"), "bitonic");
                if (in_lds && !use_bitonic) {
                    // lists that fit LDS twice over (keys + 16-bit positions, two copies): the LSD radix kernel
                    const size_t shm2 = 512 + ((Lmax + 15u) & ~15u) + 4096 + 64 + (size_t)P * 12;
                    if (shm2 > 64 * 1024 &&
                        hipFuncSetAttribute((const void *)kmer_extract_lsd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2) != hipSuccess) {
                        set_error("kmer_extract: this device cannot give a workgroup " + std::to_string(shm2) + " B of LDS (read of " + std::to_string(Lmax) + " nt)");
                        rc = RATTLE_ERR_HIP;
                        break;
                    }
                    hipLaunchKernelGGL(kmer_extract_lsd_kernel, dim3(m, ns), dim3(256), shm2, st, X.seq.p, X.off.p, X.koff.p, d_items.p + done + b, k, P, Lmax,
                                       X.uh.p, X.kh[0].p, X.kp[0].p, X.kh[1].p, X.kp[1].p, X.bv[0].p, X.bv[1].p, X.pc[0].p, X.pc[1].p, d_bad.p);
                    e = hipGetLastError();
                    if (e != hipSuccess) { set_error(std::string("kmer_extract (radix) launch: ") + hipGetErrorString(e)); rc = RATTLE_ERR_HIP; }
                    continue;
                }
                if (shm > 64 * 1024 &&
                    hipFuncSetAttribute((const void *)kmer_extract_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) {
                    set_error("kmer_extract: this device cannot give a workgroup " + std::to_string(shm) + " B of LDS (read of " + std::to_string(Lmax) + " nt)");
                    rc = RATTLE_ERR_HIP;
                    break;
                }
                hipLaunchKernelGGL(kmer_extract_kernel, dim3(m, ns), dim3(256), shm, st, X.seq.p, X.off.p, X.koff.p,
                                   d_items.p + done + b, k, P, Lmax, X.uh.p, X.kh[0].p, X.kp[0].p, X.kh[1].p, X.kp[1].p,
                                   X.bv[0].p, X.bv[1].p, X.pc[0].p, X.pc[1].p, gs, d_bad.p);
                e = hipGetLastError();
                if (e != hipSuccess) { set_error(std::string("kmer_extract launch: ") + hipGetErrorString(e)); rc = RATTLE_ERR_HIP; }
                if (!in_lds) (void)hipStreamSynchronize(st);     // scratch is reused by the next chunk
            }
            done += cnt;
        }
    }
    uint32_t bad = 0;
    hipError_t e = hipMemcpyAsync(&bad, d_bad.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    d_items.release(); d_bad.release(); d_gs.release();
    if (rc) return rc;
    if (e != hipSuccess) { set_error(std::string("kmer_extract: ") + hipGetErrorString(e)); return RATTLE_ERR_HIP; }
    if (bad) { set_error("read contains a base outside A/C/G/T/U (undefined in the reference, kmer.hpp:36)"); return RATTLE_ERR_ARG; }
    return 0;
}

}  // namespace rattle
