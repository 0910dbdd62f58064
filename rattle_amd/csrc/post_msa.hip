// Kernel D: the post-MSA logic of `rattle correct` on the device, so MSA rows never leave HBM.
// Replaces /root/reference/correct.cpp:32-92 (fix_msa_ends), :94-193 (column vote) and :196-309
// (per-read correction); restated for the CPU in oracle/orc_correct.hpp.
//
// One 256-thread workgroup per pack:
//   a. expand: the pack's rows x width matrix of bases (and of quality bytes) is filled with '-'
//      and every base is scattered to the MSA column kernel C assigned to it;
//   b. fix ends: one thread per row runs the reference's scan from both ends (small leading blocks
//      followed by a long gap are blanked); the row's voting window [first, last] and the number of
//      bases trimmed at each end are kept;
//   c. vote: one thread per column walks the rows IN ROW ORDER (the reference's accumulation order
//      with one thread, so the double sums are bit-identical), then takes the means, the winner in
//      the reference's slot order with strict '>', the two occupancy tests and the quality symbol of
//      the winner's mean error;
//   d. MODE 1 (after POA #1): one thread per row rewrites the read against the column winners
//      (in place: the output index never passes the column index);
//      MODE 2 (after POA #2 / #3): the gap-stripped column winners are the pack consensus.
//
// phred_symbol (utils.cpp:6-8) is `(char)(-10*log10(p)+33)`: the host libm's log10 decides the
// truncation, so the device does not evaluate log10 at all.  The host tabulates, for every integer n
// the expression can take, the smallest double p whose symbol value is <= n (bit-level bisection with
// the host's own log10, verified exhaustively in a window around every threshold, deviations kept as
// explicit exceptions) and the device bisects that table: bit-exact by construction.
//
// HBM traffic (algorithmic): per MSA cell 1 B (+1 B quality) written by (a) and read by (c) and (d).
#include <cmath>
#include <cstring>

#include "common.h"

namespace rattle {

__device__ __forceinline__ uint8_t d_comp_base(uint8_t c) {                       // utils.hpp:8-14
    switch (c) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'T': return 'A';
        case 'G': return 'C';
        case 'U': return 'A';
    }
    return c;
}

// One wavefront per descriptor: dst[0..len) = src[0..len) or its reverse complement (qualities: reversed).
__global__ __launch_bounds__(256) void gather_kernel(const gather_desc *D, uint32_t n, const uint8_t *sseq, const uint8_t *squal,
                                                      uint8_t *dseq, uint8_t *dqual) {
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (w >= n) return;
    const gather_desc d = D[w];
    if (d.flags & 1u) {
        for (uint32_t t = lane; t < d.len; t += 64) {
            const uint64_t s = d.src + d.len - 1 - t;
            dseq[d.dst + t] = d_comp_base(sseq[s]);
            if (dqual) dqual[d.dst + t] = squal[s];
        }
    } else {
        for (uint32_t t = lane; t < d.len; t += 64) {
            dseq[d.dst + t] = sseq[d.src + t];
            if (dqual) dqual[d.dst + t] = squal[d.src + t];
        }
    }
}

int launch_gather(rattle_ctx *ctx, const gather_desc *d_desc, uint32_t n, const uint8_t *sseq, const uint8_t *squal, uint8_t *dseq,
                  uint8_t *dqual) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(gather_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, d_desc, n, sseq, squal, dseq, dqual);
    RT_HIP(hipGetLastError());
    return 0;
}

// symbol value of a mean error: smallest n with p >= lo[n - n0]; exceptions override.
__device__ int phred_value(const post_args &A, double p) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(p);
    int a = 0, b = (int)A.n_exc;
    while (a < b) { const int m = (a + b) >> 1; if (A.exc_bits[m] < bits) a = m + 1; else b = m; }
    if (a < (int)A.n_exc && A.exc_bits[a] == bits) return A.exc_val[a];
    int lo = 0, hi = A.phred_cnt - 1;          // lo[] is non-increasing in n; p >= lo[cnt-1] always holds for valid input
    while (lo < hi) { const int m = (lo + hi) >> 1; if (p >= A.phred_lo[m]) hi = m; else lo = m + 1; }
    return A.phred_n0 + lo;
}

template <bool REV>
__device__ void fix_phase(uint8_t *r, uint32_t n, uint32_t &trimmed, uint32_t &stop_pos, bool &stopped) {
#define AT(p) (REV ? r[n - 1u - (p)] : r[(p)])
    uint32_t pos = 0;
    trimmed = 0; stopped = false; stop_pos = n;
    while (pos < n) {
        while (pos < n && AT(pos) == '-') ++pos;
        uint32_t end = pos;
        int gaps = 0, sz = 0;
        while (gaps < 4 && end < n) {
            if (AT(end) == '-') ++gaps; else { ++sz; gaps = 0; }
            ++end;
        }
        if (sz < 10) {
            while (end < n && AT(end) == '-') { ++end; ++gaps; }
            if (gaps >= 20) {
                for (uint32_t t = pos; t < end; ++t) { if (REV) r[n - 1u - t] = '-'; else r[t] = '-'; }
                trimmed += (uint32_t)sz;
                pos = end;
                continue;
            }
        }
        stopped = true; stop_pos = pos;
        break;
    }
#undef AT
}

template <int MODE>
__global__ __launch_bounds__(256) void post_msa_kernel(post_args A) {
    __shared__ double s_perr[256];
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    const uint32_t q0 = A.pack_first[p], q1 = A.pack_first[p + 1], R = q1 - q0;
    const uint32_t W = A.width[p];
    uint8_t *rc = A.rowc + A.moff[p];
    uint8_t *rq = MODE == 1 ? A.rowq + A.moff[p] : nullptr;
    const uint64_t cells = (uint64_t)R * W;
    if (MODE == 1) s_perr[tid] = A.perr[tid];
    if (R == 0 || W == 0) {
        if (tid == 0 && MODE == 2) A.cons_len[p] = 0;
        if (MODE == 1) for (uint32_t i = tid; i < R; i += 256) { A.olen[q0 + i] = 0; A.tfront[q0 + i] = 0; A.tback[q0 + i] = 0; }
        return;
    }
    // ---- a. expand
    for (uint64_t t = tid; t < cells; t += 256) { rc[t] = '-'; if (MODE == 1) rq[t] = 0; }
    __syncthreads();
    for (uint32_t i = tid >> 6; i < R; i += 4) {
        const uint64_t b0 = A.off[q0 + i], b1 = A.off[q0 + i + 1];
        uint8_t *row = rc + (uint64_t)i * W;
        for (uint64_t b = b0 + (tid & 63u); b < b1; b += 64) {
            const uint32_t c = A.col[b];
            row[c] = A.seq[b];
            if (MODE == 1) rq[(uint64_t)i * W + c] = A.qual[b];
        }
    }
    __syncthreads();
    // ---- b. fix ends (correct.cpp:32-92): phase 1 from the left, phase 2 from the right
    for (uint32_t i = tid; i < R; i += 256) {
        uint8_t *row = rc + (uint64_t)i * W;
        uint32_t t1 = 0, t2 = 0, p1 = W, p2 = W;
        bool s1 = false, s2 = false;
        fix_phase<false>(row, W, t1, p1, s1);
        if (s1) fix_phase<true>(row, W, t2, p2, s2);
        int32_t first = (int32_t)W, last = -1;
        if (s1 && s2 && p1 < W && p2 < W) { first = (int32_t)p1; last = (int32_t)(W - 1u - p2); }
        A.rfirst[q0 + i] = first; A.rlast[q0 + i] = last;
        if (MODE == 1) { A.tfront[q0 + i] = t1; A.tback[q0 + i] = t2; }
    }
    __syncthreads();
    // ---- c. vote (correct.cpp:94-193), one column per thread, rows in order
    uint8_t *ccons = A.ccons + A.coff[p];
    uint8_t *cflag = A.cflag + A.coff[p];
    uint8_t *csym = A.csym + A.coff[p];
    double *cerr = A.cerr + A.coff[p];
    const uint8_t o0 = A.order[0], o1 = A.order[1], o2 = A.order[2], o3 = A.order[3], o4 = A.order[4], o5 = A.order[5];
    for (uint32_t k = tid; k < W; k += 256) {
        int occ0 = 0, occ1 = 0, occ2 = 0, occ3 = 0, occ4 = 0, occ5 = 0;
        double e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0, e4 = 0.0, e5 = 0.0;
        for (uint32_t i = 0; i < R; ++i) {
            if ((int32_t)k < A.rfirst[q0 + i] || (int32_t)k > A.rlast[q0 + i]) continue;
            const uint8_t nt = rc[(uint64_t)i * W + k];
            if (MODE == 1) {
                const double ep = nt != '-' ? s_perr[rq[(uint64_t)i * W + k]] : 0.0;
                // one slot matches; the other sums are unchanged (x + 0.0 == x for x >= +0.0)
                if (nt == o0) { ++occ0; e0 += ep; } else if (nt == o1) { ++occ1; e1 += ep; } else if (nt == o2) { ++occ2; e2 += ep; }
                else if (nt == o3) { ++occ3; e3 += ep; } else if (nt == o4) { ++occ4; e4 += ep; } else if (nt == o5) { ++occ5; e5 += ep; }
            } else {
                occ0 += nt == o0; occ1 += nt == o1; occ2 += nt == o2; occ3 += nt == o3; occ4 += nt == o4; occ5 += nt == o5;
            }
        }
        const int total = occ0 + occ1 + occ2 + occ3 + occ4 + occ5;
        int best = 0, bocc = 0;
        uint8_t nt = 0;
        double berr = 0.0;
        // reference iteration order, strict '>' (correct.cpp:174-186); means as err / double(occ)
#define SLOT(occ, err, oc)                                                      \
        { double m = err; if (occ > 0 && MODE == 1) m = err / double(occ);      \
          if (occ > best) { best = occ; nt = oc; bocc = occ; berr = m; } }
        SLOT(occ0, e0, o0) SLOT(occ1, e1, o1) SLOT(occ2, e2, o2) SLOT(occ3, e3, o3) SLOT(occ4, e4, o4) SLOT(occ5, e5, o5)
#undef SLOT
        if (nt == 0) { ccons[k] = '-'; if (MODE == 1) { cflag[k] = 0; csym[k] = 0; cerr[k] = 0.0; } continue; }
        ccons[k] = nt;
        if (MODE == 1) {
            const double ratio = double(bocc) / double(total);
            cflag[k] = (uint8_t)((ratio >= A.gap_occ ? 1 : 0) | (ratio >= A.min_occ ? 2 : 0));
            cerr[k] = berr;
            csym[k] = (uint8_t)(phred_value(A, berr) & 0xFF);
        }
    }
    __syncthreads();
    if (MODE == 1) {
        // ---- d. per-read correction (correct.cpp:196-309), in place
        for (uint32_t i = tid; i < R; i += 256) {
            uint8_t *row = rc + (uint64_t)i * W;
            uint8_t *qrow = rq + (uint64_t)i * W;
            const int32_t first = A.rfirst[q0 + i], last = A.rlast[q0 + i];
            uint32_t o = 0;
            for (int32_t k = first; k <= last; ++k) {
                const uint8_t nt = row[k], qq = qrow[k], cnt = ccons[k], fl = cflag[k];
                uint8_t es = 0, eq = 0;
                bool emit = true;
                if (cnt == '-') {
                    if (nt != '-' && !(fl & 1)) { es = nt; eq = qq; } else emit = false;
                } else if (nt == '-') {
                    if (fl & 1) { es = cnt; eq = csym[k]; } else emit = false;
                } else if (nt == cnt) {
                    es = nt; eq = qq;
                } else if ((fl & 2) && A.err_ratio * s_perr[qq] > cerr[k]) {
                    es = cnt; eq = csym[k];
                } else {
                    es = nt; eq = qq;
                }
                if (emit) { row[o] = es; qrow[o] = eq; ++o; }
            }
            A.olen[q0 + i] = o;
        }
    } else {
        // ---- d'. consensus = column winners without gaps
        if (tid < 64) {
            uint8_t *out = A.cons_out + A.coff[p];
            uint32_t o = 0;
            for (uint32_t base = 0; base < W; base += 64) {
                const uint32_t k = base + tid;
                const uint8_t c = k < W ? ccons[k] : (uint8_t)'-';
                const bool keep = c != '-';
                const unsigned long long m = __ballot(keep);
                if (keep) out[o + __popcll(m & ((1ull << tid) - 1ull))] = c;
                o += (uint32_t)__popcll(m);
            }
            if (tid == 0) A.cons_len[p] = o;
        }
    }
}

int launch_post_msa(rattle_ctx *ctx, const post_args &A, uint32_t n_packs, int mode) {
    if (n_packs == 0) return 0;
    ktimer T(ctx, K_POST, 0);
    if (mode == 1) hipLaunchKernelGGL(post_msa_kernel<1>, dim3(n_packs), dim3(256), 0, ctx->stream, A);
    else hipLaunchKernelGGL(post_msa_kernel<2>, dim3(n_packs), dim3(256), 0, ctx->stream, A);
    RT_HIP(hipGetLastError());
    return 0;
}

// ---- phred_symbol table (host) ----------------------------------------------------------------------
namespace {
inline int phred_host(double p) { return (int)(-10 * log10(p) + 33); }      // utils.cpp:6-8 before the (char) narrowing
inline double from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
}  // namespace

void build_phred_table(phred_table &T) {
    // symbol values n0 .. n0+cnt-1 cover p in [1e-36, 1e33]; quality bytes give p in [1e-10, 1e17]
    T.n0 = -300;
    const int cnt = 700;
    T.lo.assign(cnt, 0.0);
    T.exc_bits.clear(); T.exc_val.clear();
    const uint64_t bmin = 1, bmax = 0x7FEFFFFFFFFFFFFFull;
    for (int t = 0; t < cnt; ++t) {
        const int n = T.n0 + t;
        // smallest positive double with phred_host(p) <= n (phred_host is non-increasing in p up to libm wiggles)
        uint64_t a = bmin, b = bmax;
        while (a < b) { const uint64_t m = a + (b - a) / 2; if (phred_host(from_bits(m)) <= n) b = m; else a = m + 1; }
        T.lo[t] = from_bits(a);
    }
    // exhaustive check around every threshold; anything the table would get wrong becomes an exception
    auto table_value = [&](double p) {
        int lo = 0, hi = cnt - 1;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (p >= T.lo[m]) hi = m; else lo = m + 1; }
        return T.n0 + lo;
    };
    std::vector<std::pair<uint64_t, int>> exc;
    for (int t = 0; t < cnt; ++t) {
        uint64_t c; memcpy(&c, &T.lo[t], 8);
        if (c < 4096 || c > bmax - 4096) continue;
        for (uint64_t b = c - 2048; b <= c + 2048; ++b) {
            const double p = from_bits(b);
            const int want = phred_host(p);
            if (table_value(p) != want) exc.emplace_back(b, want);
        }
    }
    std::sort(exc.begin(), exc.end());
    exc.erase(std::unique(exc.begin(), exc.end()), exc.end());
    for (auto &e : exc) { T.exc_bits.push_back(e.first); T.exc_val.push_back(e.second); }
}

// Host mirror of the device lookup (phred_value), for the CPU-side test of the table construction.
int phred_lookup_host(const phred_table &T, double p) {
    uint64_t bits;
    memcpy(&bits, &p, 8);
    const auto it = std::lower_bound(T.exc_bits.begin(), T.exc_bits.end(), bits);
    if (it != T.exc_bits.end() && *it == bits) return T.exc_val[it - T.exc_bits.begin()];
    int lo = 0, hi = (int)T.lo.size() - 1;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (p >= T.lo[m]) hi = m; else lo = m + 1; }
    return T.n0 + lo;
}

}  // namespace rattle
