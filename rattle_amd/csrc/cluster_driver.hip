// Host orchestration of the greedy clustering, /root/reference/cluster.cpp:93-259, over the
// device kernels A (bv_filter) and B (pair_score).
//
// The reference walks seeds one at a time and, per seed, spawns n_threads tasks over the
// remaining reads (cluster.cpp:138-158).  cluster_together(i,j) is a pure function of
// (i, j, threshold), and read j joins seed i iff j is still un-clustered when i is
// processed.  So a BATCH of the next B un-clustered items can be evaluated in one launch
// and resolved afterwards in index order with an identical result:
//   level 1: seeds x seeds  -> which seeds are absorbed by an earlier seed (exact founders)
//   level 2: founders x all remaining items -> each item joins the FIRST founder that accepts it
// The merge passes (cluster.cpp:171-256) are the same procedure over cluster representatives.
//
// One job over several GPUs: the reference cuts the candidate loop of a seed over its threads
// (cluster.cpp:138-158: thread t takes candidates t, t+T, ...).  Here rank r of R scores the candidates at
// positions r, r+R, ... of every level-2 evaluation (level 1, 256 x 256, is replicated), the accepted
// (seed, candidate, strand) triples are all-gathered and every rank resolves them identically.
//
// Many independent clusterings at once (`--iso`: one per gene cluster, main.cpp:281-318, which the reference
// runs one after another): every clustering is a small state machine (`job`) that stops whenever it needs a
// rectangle of cluster_together verdicts; the rectangles of all jobs that are waiting go to the device in ONE
// evaluation (kernel A over the rectangle list, kernel B over the union of the surviving pairs), the verdicts
// are dealt back and the jobs advance.  A round costs a handful of launches whatever the number of genes.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace rattle {

int launch_pair_score_oversize(rattle_ctx *ctx, const std::vector<uint32_t> &slots, uint32_t max_matches);

// survivor list (seed_slot<<1|strand, cand_slot) -> explicit (read i, read j, strand) pairs
__global__ void expand_pairs_kernel(const uint32_t *__restrict__ surv, uint32_t n, const uint32_t *__restrict__ seed_ids,
                                    const uint32_t *__restrict__ cand_ids, uint32_t *__restrict__ pi, uint32_t *__restrict__ pj,
                                    uint8_t *__restrict__ ps) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint32_t a = surv[2 * (uint64_t)t], c = surv[2 * (uint64_t)t + 1];
    pi[t] = seed_ids[a >> 1];
    pj[t] = cand_ids[c];
    ps[t] = (uint8_t)(a & 1u);
}

// cluster.cpp:23-27 turned into an exact rejection on |common| (see run_local): pairs that can still reach t_s are compacted
// for the full comparison; survivors / matches per rectangle and the algorithmic bytes are summed on the way.
// stats: [0] pairs kept, [1] algorithmic bytes, then per rectangle (survivors, matches, pairs kept)
__global__ __launch_bounds__(256) void count_bound_kernel(const uint32_t *__restrict__ surv, const int32_t *__restrict__ common, uint32_t n,
                                                          const uint32_t *__restrict__ pi, const uint32_t *__restrict__ pj, const uint8_t *__restrict__ ps,
                                                          const uint32_t *__restrict__ len, uint32_t kk, double t_s,
                                                          const uint32_t *__restrict__ seed_rect, unsigned long long *__restrict__ stats,
                                                          uint32_t *__restrict__ pi2, uint32_t *__restrict__ pj2, uint8_t *__restrict__ ps2,
                                                          uint32_t *__restrict__ slot2) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < n;
    uint32_t a = 0, c = 0, ri = 0, rj = 0, rect = 0;
    unsigned long long M = 0, bytes = 0;
    bool keep = false;
    if (live) {
        a = surv[2 * (uint64_t)p]; c = surv[2 * (uint64_t)p + 1];
        M = (uint32_t)common[p];
        ri = pi[p]; rj = pj[p];
        const uint32_t li = len[ri], lj = len[rj];
        bytes = 8ull * ((li > kk ? li - kk : 0) + (lj > kk ? lj - kk : 0));
        const double mn = (double)(li < lj ? li : lj);
        keep = (double)(M * kk) / mn >= t_s;
        rect = seed_rect ? seed_rect[a >> 1] : 0u;
    }
    // one atomic per wavefront where its lanes agree on the rectangle (always, with a single rectangle)
    const uint32_t rect0 = __shfl(rect, 0, 64);
    const bool uniform = __all(!live || rect == rect0);
    unsigned long long mm = M, bb = bytes, one = live ? 1ull : 0ull;
    for (int d = 32; d; d >>= 1) { bb += __shfl_down(bb, d, 64); if (uniform) { mm += __shfl_down(mm, d, 64); one += __shfl_down(one, d, 64); } }
    const int lane = threadIdx.x & 63;
    if (lane == 0) atomicAdd(&stats[1], bb);
    if (uniform) {
        if (lane == 0 && one) { atomicAdd(&stats[2 + 3 * (size_t)rect0], one); atomicAdd(&stats[3 + 3 * (size_t)rect0], mm); }
    } else if (live) {
        atomicAdd(&stats[2 + 3 * (size_t)rect], 1ull); atomicAdd(&stats[3 + 3 * (size_t)rect], M);
    }
    const unsigned long long mask = __ballot(keep);
    if (mask) {
        if (uniform) { if (lane == 0) atomicAdd(&stats[4 + 3 * (size_t)rect0], (unsigned long long)__popcll(mask)); }
        else if (keep) atomicAdd(&stats[4 + 3 * (size_t)rect], 1ull);
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&stats[0], (unsigned long long)__popcll(mask));
        base = __shfl(base, 0, 64);
        if (keep) {
            const uint64_t at = base + (uint64_t)__popcll(mask & ((1ull << lane) - 1ull));
            pi2[at] = ri; pj2[at] = rj; ps2[at] = ps[p];
            slot2[2 * at] = a; slot2[2 * at + 1] = c;
        }
    }
}

// cluster.cpp:23-36 / :47-61 on the device, in the reference's double arithmetic (an IEEE division and two comparisons, the same
// expression the host evaluated until round 3): the accepted pairs are compacted as (seed slot << 1 | strand, candidate slot), so
// the host sees the few accepted pairs instead of four integers and a double for every full comparison (10 M of them in the
// --iso level at 1e6 reads).  Pairs whose match list did not fit LDS are listed for the oversize pass (out[0] = count, out[1] =
// oversize pairs, out[2] = their largest match count) and judged by a second launch over that list.
__global__ __launch_bounds__(256) void verdict_kernel(const int32_t *__restrict__ res, const double *__restrict__ var, uint32_t n,
                                                      const uint32_t *__restrict__ remap, const uint32_t *__restrict__ pi,
                                                      const uint32_t *__restrict__ pj, const uint32_t *__restrict__ slot2,
                                                      const uint32_t *__restrict__ len, int use_hc, double t_s, double t_v,
                                                      unsigned long long *__restrict__ out, uint32_t *__restrict__ hits,
                                                      uint32_t *__restrict__ big) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool ok = false, over = false;
    uint32_t q = 0, m_over = 0;
    if (t < n) {
        q = remap ? remap[t] : t;
        const int32_t *r = res + 4 * (size_t)q;
        if (r[0] == INT32_MIN) { over = true; m_over = (uint32_t)r[3]; }
        else {
            const uint32_t li = len[pi[q]], lj = len[pj[q]];
            const double mn = (double)(li < lj ? li : lj);
            const double score = use_hc ? (double)r[1] / mn : (double)r[0] / mn;
            ok = score >= t_s && var[q] < t_v;
        }
    }
    const unsigned long long mo = __ballot(over);
    if (mo) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&out[1], (unsigned long long)__popcll(mo));
        base = __shfl(base, 0, 64);
        if (over) { big[base + (uint64_t)__popcll(mo & ((1ull << lane) - 1ull))] = q; atomicMax(&out[2], (unsigned long long)m_over); }
    }
    const unsigned long long mk = __ballot(ok);
    if (mk) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&out[0], (unsigned long long)__popcll(mk));
        base = __shfl(base, 0, 64);
        if (ok) {
            const uint64_t at = base + (uint64_t)__popcll(mk & ((1ull << lane) - 1ull));
            hits[2 * at] = slot2[2 * (size_t)q]; hits[2 * at + 1] = slot2[2 * (size_t)q + 1];
        }
    }
}

namespace {

struct hit_t { uint32_t seed, cand; uint8_t rev; };

struct cseq { int32_t id; uint8_t rev; };

// one rectangle of cluster_together(i, j) evaluations a job waits for
struct request {
    std::vector<uint32_t> seeds, cands;       // loaded-read ids; triangular: candidates = the seeds, pairs (s, c > s) only
    bool triangular = false;
    double thr = 0.0;                         // bit-vector threshold of the pass (cluster.cpp:19,43)
    uint64_t *counters = nullptr;             // the job's work counters
    std::vector<hit_t> hits;                  // out: accepted (seed index, candidate index, strand); sorted if triangular
    uint32_t n_cands() const { return (uint32_t)(triangular ? seeds.size() : cands.size()); }
    uint64_t n_pairs() const {
        const uint64_t s = seeds.size();
        return triangular ? s * (s - 1) / 2 : s * (uint64_t)cands.size();
    }
};

// ---- device evaluation of a set of rectangles -------------------------------------------------------------
struct evaluator {
    rattle_ctx *ctx;
    const rattle_cluster_params *P;
    uint64_t launches = 0;
    // RATTLE_TIMING: where the host's wall time of a clustering goes
    double t_split[6] = {0, 0, 0, 0, 0, 0};
    std::chrono::steady_clock::time_point t_mark;
    void mark() { t_mark = std::chrono::steady_clock::now(); }
    void lap(int i) { const auto now = std::chrono::steady_clock::now(); t_split[i] += std::chrono::duration<double, std::milli>(now - t_mark).count(); t_mark = now; }
    std::vector<double> lut_thr;              // row r of the device table belongs to threshold lut_thr[r]
    std::vector<uint16_t> luts;
    std::vector<uint32_t> h_seed, h_cand, h_first, seed_req;
    std::vector<bvf_rect> rects;

    // min_common_lut[m] = smallest c with double(c)/double(m) >= thr  (cluster.cpp:19,43)
    int lut_row(double thr, uint32_t &row) {
        for (size_t r = 0; r < lut_thr.size(); ++r) if (lut_thr[r] == thr) { row = (uint32_t)r; return 0; }
        row = (uint32_t)lut_thr.size();
        lut_thr.push_back(thr);
        luts.resize((size_t)(row + 1) * 4097);
        uint16_t *lut = luts.data() + (size_t)row * 4097;
        for (int m = 0; m <= 4096; ++m) {
            double mmax = (double)m;
            int need = 0xFFFF;
            int lo = 0, hi = 4096;               // predicate is monotone in c for m > 0
            if (m > 0 && (double)(size_t)hi / mmax >= thr) {
                while (lo < hi) {
                    int mid = (lo + hi) / 2;
                    if ((double)(size_t)mid / mmax >= thr) hi = mid; else lo = mid + 1;
                }
                need = lo;
            }
            lut[m] = (uint16_t)need;
        }
        RT_TRY(ctx->d_lut.reserve(luts.size()));
        RT_HIP(hipMemcpyAsync(ctx->d_lut.p, luts.data(), luts.size() * 2, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(hipStreamSynchronize(ctx->stream));   // luts may move on the next push
        return 0;
    }

    // All requests of one greedy step.  shard: this rank scores the candidates at positions rank, rank + nranks, ...
    // of the (single, rectangular) request and the hits of all ranks are all-gathered.
    int run(std::vector<request *> &reqs, bool shard) {
        const int R = ctx->xchg.nranks, r = ctx->xchg.rank;
        if (shard && R > 1) {
            if (reqs.size() != 1 || reqs[0]->triangular) { set_error("sharded evaluation takes one rectangular request"); return RATTLE_ERR_STATE; }
            request &q = *reqs[0];
            request mine;
            mine.seeds = q.seeds; mine.thr = q.thr; mine.counters = q.counters;
            for (uint32_t c = (uint32_t)r; c < q.cands.size(); c += (uint32_t)R) mine.cands.push_back(q.cands[c]);
            uint64_t before[3] = {q.counters[0], q.counters[1], q.counters[2]};
            std::vector<request *> one{&mine};
            RT_TRY(run_chunks(one));
            // payload: three counter deltas, then (seed, global candidate position, strand) triples
            std::vector<uint8_t> pay(24 + mine.hits.size() * 12);
            for (int i = 0; i < 3; ++i) { const uint64_t d = q.counters[i] - before[i]; memcpy(pay.data() + 8 * i, &d, 8); q.counters[i] = before[i]; }
            for (size_t i = 0; i < mine.hits.size(); ++i) {
                const uint32_t t[3] = {mine.hits[i].seed, mine.hits[i].cand * (uint32_t)R + (uint32_t)r, mine.hits[i].rev};
                memcpy(pay.data() + 24 + 12 * i, t, 12);
            }
            std::vector<std::vector<uint8_t>> all;
            RT_TRY(xchg_allgatherv(ctx, pay, all));
            q.hits.clear();
            const bool replay = xchg_replaying(ctx);      // (measurement aid, common.h: the piece beside this rank's own is the WHOLE job's hit list)
            for (int p = 0; p < R; ++p) {
                const std::vector<uint8_t> &b = all[(size_t)p];
                if (replay && b.empty()) continue;
                if (b.size() < 24 || (b.size() - 24) % 12) { set_error("cluster exchange: malformed hit list"); return RATTLE_ERR_HIP; }
                if (!(replay && p == r)) for (int i = 0; i < 3; ++i) { uint64_t d; memcpy(&d, b.data() + 8 * i, 8); q.counters[i] += d; }
                for (size_t at = 24; at < b.size(); at += 12) {
                    uint32_t t[3];
                    memcpy(t, b.data() + at, 12);
                    if (replay && p != r && t[1] % (uint32_t)R == (uint32_t)r) continue;      // this rank's own hits arrived in its own piece
                    q.hits.push_back(hit_t{t[0], t[1], (uint8_t)t[2]});
                }
            }
            return 0;
        }
        if (shard && xchg_recording(ctx)) {
            // single rank, recording: the same payload a sharded job exchanges here, for the whole request
            if (reqs.size() != 1 || reqs[0]->triangular) { set_error("sharded evaluation takes one rectangular request"); return RATTLE_ERR_STATE; }
            request &q = *reqs[0];
            uint64_t before[3] = {q.counters[0], q.counters[1], q.counters[2]};
            RT_TRY(run_chunks(reqs));
            std::vector<uint8_t> pay(24 + q.hits.size() * 12);
            for (int i = 0; i < 3; ++i) { const uint64_t d = q.counters[i] - before[i]; memcpy(pay.data() + 8 * i, &d, 8); }
            for (size_t i = 0; i < q.hits.size(); ++i) {
                const uint32_t t[3] = {q.hits[i].seed, q.hits[i].cand, q.hits[i].rev};
                memcpy(pay.data() + 24 + 12 * i, t, 12);
            }
            std::vector<std::vector<uint8_t>> all;
            return xchg_allgatherv(ctx, pay, all);
        }
        return run_chunks(reqs);
    }

    static void sort_hits(std::vector<hit_t> &hits) {
        std::sort(hits.begin(), hits.end(), [](const hit_t &a, const hit_t &b) {
            return a.seed != b.seed ? a.seed < b.seed : (a.cand != b.cand ? a.cand < b.cand : a.rev < b.rev);
        });
    }

    // Every pair can survive the filter, and in the thr == 0 pass every pair does (cluster.cpp:19,43): one launch
    // keeps its pairs x strands below the 32-bit survivor counter, and below 64 M where all of them survive.
    int run_chunks(std::vector<request *> &reqs) {
        const uint64_t strands = ctx->idx.both ? 2u : 1u;
        size_t a = 0;
        while (a < reqs.size()) {
            uint64_t all_pairs = 0, sure = 0;
            size_t b = a;
            while (b < reqs.size()) {
                const uint64_t p = reqs[b]->n_pairs() * strands;
                const uint64_t s = reqs[b]->thr == 0.0 ? p : 0;
                if (b > a && (all_pairs + p > (1ull << 31) || sure + s > (64ull << 20))) break;
                all_pairs += p; sure += s;
                ++b;
            }
            RT_TRY(run_local(reqs.data() + a, b - a));
            a = b;
        }
        return 0;
    }

    int run_local(request **reqs, size_t nreq) {
        hipStream_t st = ctx->stream;
        const read_index &X = ctx->idx;
        mark();
        // ---- the rectangles side by side in one seed array and one candidate array
        uint64_t ns = 0, nc = 0, npairs = 0;
        for (size_t q = 0; q < nreq; ++q) {
            reqs[q]->hits.clear();
            if (reqs[q]->seeds.empty() || reqs[q]->n_cands() == 0) continue;
            ns += reqs[q]->seeds.size(); nc += reqs[q]->n_cands();
        }
        if (ns == 0 || nc == 0) return 0;
        if (ns >= (1u << 31) || nc >= 0xFFFFFFF0ull) { set_error("cluster evaluation: too many seeds or candidates in one step"); return RATTLE_ERR_ARG; }
        h_seed.resize(ns); h_first.resize(ns); h_cand.resize(nc);
        rects.clear();
        const bool many = nreq > 1;
        if (many) seed_req.resize(ns);
        std::vector<uint32_t> rect_req;
        uint32_t sb = 0, cb = 0, tiles = 0;
        for (size_t q = 0; q < nreq; ++q) {
            request &Q = *reqs[q];
            const uint32_t qs = (uint32_t)Q.seeds.size(), qc = Q.n_cands();
            if (qs == 0 || qc == 0) continue;
            uint32_t row = 0;
            RT_TRY(lut_row(Q.thr, row));
            memcpy(h_seed.data() + sb, Q.seeds.data(), (size_t)qs * 4);
            memcpy(h_cand.data() + cb, Q.triangular ? Q.seeds.data() : Q.cands.data(), (size_t)qc * 4);
            for (uint32_t s = 0; s < qs; ++s) h_first[sb + s] = cb + (Q.triangular ? s + 1 : 0u);
            if (many) for (uint32_t s = 0; s < qs; ++s) seed_req[sb + s] = (uint32_t)rects.size();
            rects.push_back(bvf_rect{sb, qs, cb, qc, tiles, row * 4097u, Q.thr == 0.0 ? 1u : 0u, 0u});
            rect_req.push_back((uint32_t)q);
            tiles += ((qc + 255u) / 256u) * ((qs + 31u) / 32u);
            const uint64_t p = Q.n_pairs();
            Q.counters[0] += p;
            npairs += p;
            sb += qs; cb += qc;
        }
        const uint32_t nrect = (uint32_t)rects.size();
        RT_TRY(ctx->d_seed.reserve(ns)); RT_TRY(ctx->d_first.reserve(ns)); RT_TRY(ctx->d_cand.reserve(nc));
        RT_TRY(ctx->d_rect.reserve(nrect));
        RT_TRY(ctx->d_counter.reserve(4)); RT_TRY(ctx->h_counter.reserve(4));
        RT_HIP(hipMemcpyAsync(ctx->d_seed.p, h_seed.data(), ns * 4, hipMemcpyHostToDevice, st));
        RT_HIP(hipMemcpyAsync(ctx->d_first.p, h_first.data(), ns * 4, hipMemcpyHostToDevice, st));
        RT_HIP(hipMemcpyAsync(ctx->d_cand.p, h_cand.data(), nc * 4, hipMemcpyHostToDevice, st));
        RT_HIP(hipMemcpyAsync(ctx->d_rect.p, rects.data(), (size_t)nrect * sizeof(bvf_rect), hipMemcpyHostToDevice, st));

        lap(0);
        // survivor capacity: grow and retry on overflow (count is exact even when truncated)
        size_t cap = std::max<size_t>(ctx->d_surv.cap / 2, 1u << 20);
        uint32_t nsurv = 0;
        while (true) {
            RT_TRY(ctx->d_surv.reserve(cap * 2));
            cap = ctx->d_surv.cap / 2;
            RT_HIP(hipMemsetAsync(ctx->d_counter.p, 0, 16, st));
            RT_TRY(launch_bv_filter_rects(ctx, nrect, tiles, npairs, false, true, (uint32_t)std::min<size_t>(cap, 0xFFFFFFF0u)));
            ++launches;
            RT_HIP(hipMemcpyAsync(ctx->h_counter.p, ctx->d_counter.p, 4, hipMemcpyDeviceToHost, st));
            RT_HIP(hipStreamSynchronize(st));
            nsurv = ctx->h_counter.p[0];
            if ((uint64_t)nsurv > npairs * 2) { set_error("bv_filter: survivor counter overflow"); return RATTLE_ERR_HIP; }
            if (nsurv <= cap) break;
            cap = (size_t)nsurv + nsurv / 8;
        }
        lap(1);
        if (nsurv == 0) return 0;

        RT_TRY(ctx->d_pi.reserve(nsurv));
        RT_TRY(ctx->d_pj.reserve(nsurv));
        RT_TRY(ctx->d_ps.reserve(nsurv));
        RT_TRY(ctx->d_res.reserve((size_t)nsurv * 4));
        RT_TRY(ctx->d_pi2.reserve(nsurv)); RT_TRY(ctx->d_pj2.reserve(nsurv)); RT_TRY(ctx->d_ps2.reserve(nsurv)); RT_TRY(ctx->d_slot2.reserve((size_t)nsurv * 2));
        RT_TRY(ctx->d_bound_stats.reserve(2 + 3 * (size_t)nrect + 4)); RT_TRY(ctx->h_bound_stats.reserve(2 + 3 * (size_t)nrect + 4));
        if (many) {
            RT_TRY(ctx->d_seed_rect.reserve(ns));
            RT_HIP(hipMemcpyAsync(ctx->d_seed_rect.p, seed_req.data(), ns * 4, hipMemcpyHostToDevice, st));
        }
        RT_HIP(hipMemsetAsync(ctx->d_bound_stats.p, 0, (2 + 3 * (size_t)nrect + 4) * 8, st));
        // Two count passes.  "seed": survivors sorted by seed, the seed's k-mer set as a bit set in LDS, its candidates' lists
        // streamed past it (pair_count.hip) -- pays where a seed has many surviving candidates (gene level: hundreds).
        // "search": one wavefront per pair, binary searches in the candidate's list (pair_score.hip) -- better for the short
        // runs of the --iso level.  RATTLE_PAIR_COUNT=seed|search forces one of them.
        static const char *force = getenv("RATTLE_PAIR_COUNT");
        const bool seed_major = force ? !strcmp(force, "seed") : (uint64_t)nsurv >= 64ull * ns;
        if (seed_major) RT_TRY(sort_survivors_by_seed(ctx, nsurv, ns));
        hipLaunchKernelGGL(expand_pairs_kernel, dim3((nsurv + 255) / 256), dim3(256), 0, st, ctx->d_surv.p, nsurv,
                           ctx->d_seed.p, ctx->d_cand.p, ctx->d_pi.p, ctx->d_pj.p, ctx->d_ps.p);
        // ---- pass 1: |common| of every surviving pair.  bases <= k * |LIS| <= k * |common| (similarity.cpp:52-85), so a pair
        // with double(k * |common|) / min_len < t_s cannot pass cluster.cpp:23-27 whatever its chain looks like: exact
        // rejection without the patience search.  In the low-threshold merge passes that is nearly every pair.  The test
        // (same double expression) and the compaction of the pairs that pass run on the device.
        if (seed_major) RT_TRY(launch_pair_count_seed(ctx, nsurv)); else RT_TRY(launch_pair_count(ctx, nsurv));
        const double t_s = P->t_s, t_v = P->t_v;
        const uint32_t kk = (uint32_t)X.k;
        hipLaunchKernelGGL(count_bound_kernel, dim3((nsurv + 255) / 256), dim3(256), 0, st, ctx->d_surv.p, ctx->d_res.p, nsurv,
                           ctx->d_pi.p, ctx->d_pj.p, ctx->d_ps.p, X.len.p, kk, t_s, many ? ctx->d_seed_rect.p : (const uint32_t *)nullptr,
                           ctx->d_bound_stats.p, ctx->d_pi2.p, ctx->d_pj2.p, ctx->d_ps2.p, ctx->d_slot2.p);
        launches += 3;
        RT_HIP(hipMemcpyAsync(ctx->h_bound_stats.p, ctx->d_bound_stats.p, (2 + 3 * (size_t)nrect) * 8, hipMemcpyDeviceToHost, st));
        RT_HIP(hipStreamSynchronize(st));
        lap(2);
        const uint32_t n2 = (uint32_t)ctx->h_bound_stats.p[0];
        ctx->stats[K_SCORE].bytes += ctx->h_bound_stats.p[1];
        for (uint32_t j = 0; j < nrect; ++j) {
            uint64_t *cn = reqs[rect_req[j]]->counters;
            cn[1] += ctx->h_bound_stats.p[2 + 3 * (size_t)j]; cn[2] += ctx->h_bound_stats.p[3 + 3 * (size_t)j];
            cn[5] += ctx->h_bound_stats.p[4 + 3 * (size_t)j];                  // full comparisons (cluster.cpp:20 / :44 calls that ran)
        }
        lap(3);
        if (n2 == 0) return 0;
        // ---- pass 2: the reference's full comparison for the pairs that can still be accepted
        RT_TRY(ctx->d_res.reserve((size_t)n2 * 4));
        RT_TRY(ctx->d_var.reserve(n2));
        ctx->d_pi.swap(ctx->d_pi2); ctx->d_pj.swap(ctx->d_pj2); ctx->d_ps.swap(ctx->d_ps2);       // the launchers read d_pi / d_pj / d_ps
        struct unswap { rattle_ctx *c; ~unswap() { c->d_pi.swap(c->d_pi2); c->d_pj.swap(c->d_pj2); c->d_ps.swap(c->d_ps2); } } back{ctx};
        RT_TRY(launch_pair_score(ctx, n2));
        ++launches;
        // verdicts on the device: d_surv (free since the count pass compacted it into d_slot2) takes the accepted pairs, the tail
        // of the statistics buffer the three counters, d_pi2 (= the swapped-out d_pi: n2 <= nsurv words) the oversize list
        unsigned long long *vout = ctx->d_bound_stats.p + 2 + 3 * (size_t)nrect;
        uint32_t *d_hits = ctx->d_surv.p, *d_big = ctx->d_pi2.p;
        const int use_hc = P->use_hc ? 1 : 0;
        hipLaunchKernelGGL(verdict_kernel, dim3((n2 + 255) / 256), dim3(256), 0, st, ctx->d_res.p, ctx->d_var.p, n2, (const uint32_t *)nullptr,
                           ctx->d_pi.p, ctx->d_pj.p, ctx->d_slot2.p, X.len.p, use_hc, t_s, t_v, vout, d_hits, d_big);
        ++launches;
        unsigned long long *hv = ctx->h_bound_stats.p + 2 + 3 * (size_t)nrect;
        // the first accepted pairs travel with the counters (one synchronisation per evaluation instead of two: the --iso level
        // runs ~1000 evaluations of ~1000 accepted pairs each)
        const uint32_t spec = std::min<uint32_t>(n2, 8192u);
        RT_TRY(ctx->h_surv.reserve((size_t)spec * 2 + 2));
        RT_HIP(hipMemcpyAsync(hv, vout, 24, hipMemcpyDeviceToHost, st));
        RT_HIP(hipMemcpyAsync(ctx->h_surv.p, d_hits, (size_t)spec * 8, hipMemcpyDeviceToHost, st));
        RT_HIP(hipStreamSynchronize(st));
        const bool nbig_seen = hv[1] != 0;
        if (hv[1]) {
            // pairs whose match list did not fit LDS: rerun through the global-scratch variant, then judge them
            const uint32_t nbig = (uint32_t)hv[1], big_m = (uint32_t)hv[2];
            std::vector<uint32_t> big(nbig);
            RT_HIP(hipMemcpyAsync(big.data(), d_big, (size_t)nbig * 4, hipMemcpyDeviceToHost, st));
            RT_HIP(hipStreamSynchronize(st));
            RT_TRY(launch_pair_score_oversize(ctx, big, big_m));
            ++launches;
            RT_HIP(hipMemsetAsync(vout + 1, 0, 16, st));
            hipLaunchKernelGGL(verdict_kernel, dim3((nbig + 255) / 256), dim3(256), 0, st, ctx->d_res.p, ctx->d_var.p, nbig, (const uint32_t *)d_big,
                               ctx->d_pi.p, ctx->d_pj.p, ctx->d_slot2.p, X.len.p, use_hc, t_s, t_v, vout, d_hits, d_big);
            ++launches;
            RT_HIP(hipMemcpyAsync(hv, vout, 24, hipMemcpyDeviceToHost, st));
            RT_HIP(hipStreamSynchronize(st));
            if (hv[1]) { set_error("pair_score: a pair is still oversize after the oversize pass"); return RATTLE_ERR_HIP; }
        }
        const uint32_t nhit = (uint32_t)hv[0];
        if (nhit > spec || nbig_seen) {                  // more than came along (or the oversize pass appended some): fetch them all
            RT_TRY(ctx->h_surv.reserve((size_t)nhit * 2 + 2));
            if (nhit) RT_HIP(hipMemcpyAsync(ctx->h_surv.p, d_hits, (size_t)nhit * 8, hipMemcpyDeviceToHost, st));
            RT_HIP(hipStreamSynchronize(st));
        }
        // accepted pairs back to their rectangle
        for (uint32_t q = 0; q < nhit; ++q) {
            const uint32_t a = ctx->h_surv.p[2 * (size_t)q], c = ctx->h_surv.p[2 * (size_t)q + 1];
            const uint32_t rj = many ? seed_req[a >> 1] : 0;
            const bvf_rect &J = rects[rj];
            reqs[rect_req[rj]]->hits.push_back(hit_t{(a >> 1) - J.s_base, c - J.c_base, (uint8_t)(a & 1u)});
        }
        for (uint32_t j = 0; j < nrect; ++j) if (reqs[rect_req[j]]->triangular) sort_hits(reqs[rect_req[j]]->hits);      // level 2 takes them in any order
        lap(4);
        return 0;
    }
};

// ---- one clustering (cluster.cpp:93-259) as a state machine ------------------------------------------------
struct job {
    const read_index *X = nullptr;
    const rattle_cluster_params *P = nullptr;
    const uint32_t *subset = nullptr;     // local id -> loaded read id (nullptr = identity)
    uint32_t n = 0;
    bool inner_parallel = true;           // a lone job spreads its representative choice over the host threads
    uint32_t longest = 0;                 // longest read of this job (start()): decides the seed batch
    uint64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    struct cl { cseq main; std::vector<cseq> seqs; };
    std::vector<cl> clusters;
    enum { INITIAL, MERGE, DONE } stage = INITIAL;
    double thr = 0.0;
    bool last = false;

    // the greedy pass in flight: owner[i] = index of the founder item that absorbed i (owner[i] == i for founders),
    // rev[i] = strand of the match
    enum { ROUND, WAIT_L1, FOUNDERS, WAIT_L2 } phase = ROUND;
    std::vector<uint32_t> items, owner, remaining, next, seeds_local, founders, best;
    std::vector<uint8_t> rev, taken;
    uint32_t B = 0, batch_now = 0;
    request rq;
    double t_phase[5] = {0, 0, 0, 0, 0};                   // RATTLE_TIMING: round set-up, level-1 resolve, founders + request, level-2 resolve, end of pass

    uint32_t rid(uint32_t local) const { return subset ? subset[local] : local; }
    uint32_t rlen(uint32_t local) const { return X->h_len[rid(local)]; }

    // cluster.cpp:67-91.  The reference's two stable sorts (by id descending, then by length descending) leave the
    // sequences ordered by (length desc, id desc); ids are unique in a cluster, so one sort on that key gives the
    // same order, and a cluster no merge touched is still in it.
    cseq get_main_seq(std::vector<cseq> &seqs, double repr_percentile) const {
        cseq old = seqs[0];
        auto key = [this](const cseq &a) { return ((uint64_t)rlen((uint32_t)a.id) << 32) | (uint32_t)a.id; };
        bool sorted = true;
        for (size_t i = 1; i < seqs.size() && sorted; ++i) sorted = key(seqs[i - 1]) > key(seqs[i]);
        if (!sorted) {
            std::vector<std::pair<uint64_t, uint8_t>> tmp(seqs.size());
            for (size_t i = 0; i < seqs.size(); ++i) tmp[i] = {key(seqs[i]), seqs[i].rev};
            std::sort(tmp.begin(), tmp.end(), [](const std::pair<uint64_t, uint8_t> &a, const std::pair<uint64_t, uint8_t> &b) { return a.first > b.first; });
            for (size_t i = 0; i < seqs.size(); ++i) seqs[i] = cseq{(int32_t)(uint32_t)tmp[i].first, tmp[i].second};
        }
        int nsid = seqs.size() * repr_percentile;
        cseq ns = seqs[nsid];
        while (ns.rev != old.rev && (size_t)nsid < seqs.size() - 1) { nsid++; ns = seqs[nsid]; }
        if ((size_t)nsid == seqs.size() - 1) return old;
        return ns;
    }

    void begin_pass(double t) {
        thr = t;
        const uint32_t m = (uint32_t)items.size();
        owner.resize(m);
        rev.assign(m, 0);
        remaining.resize(m);
        for (uint32_t i = 0; i < m; ++i) { owner[i] = i; remaining[i] = i; }
        phase = ROUND;
        // The outcome does not depend on the batch size (level 1 resolves the seeds exactly), the work does: level 1
        // costs B^2/2 comparisons, which is wasted where a few founders absorb everything (a gene's reads falling into
        // its few isoforms).  Start small on small problems and follow the number of founders the last round produced.
        batch_now = m > 4096 ? max_batch() : std::min<uint32_t>(max_batch(), 16);
    }

    uint32_t max_batch() const {
        // A lone clustering (the gene level): 1024 -- 77 greedy rounds instead of 119 at 1e6 reads (each ends in a handful of stream
        // synchronisations) for twice the level-1 comparisons; measured on one box: 340 ms (512), 314 ms (1024), 337 ms (2048).
        // The many clusterings of the --iso level keep 512: a gene's reads fall into few isoforms and a batch that follows the
        // founders up to 1024 doubled the full comparisons there (10.5 M -> 20.9 M, 0.91 -> 1.28 s).  RATTLE_SEED_BATCH overrides both.
        // Long reads: level 1's B^2 / 2 seed-against-seed comparisons are FULL comparisons of the longest reads of the round, and a round's
        // full pass lasts as long as its longest pair.  5e5 mixed-length reads (150 nt - 98 kb, config 5), `cluster` on one box: 22.8 s
        // (32), 20.7 (64), 19.4 (128), 20.4 (256), 23.3 (512), 32.9 (1024), 52.3 (2048) -- the step from 512 to 1024 was round 4's
        // unexplained 22.8 -> 33.8 s (profiles/round5_bisect_config5_cluster.txt).  So 256 as soon as a read is beyond the packed classes.
        static const uint32_t forced = getenv("RATTLE_SEED_BATCH") ? (uint32_t)std::max(1, atoi(getenv("RATTLE_SEED_BATCH"))) : 0u;
        return forced ? forced : longest > 4096u ? 256u : inner_parallel ? 1024u : 512u;
    }

    void start() {
        rq.counters = counters;
        items.resize(n);
        for (uint32_t i = 0; i < n; ++i) items[i] = i;
        longest = 0;
        for (uint32_t i = 0; i < n; ++i) longest = std::max(longest, rlen(i));
        stage = INITIAL;
        begin_pass(P->bv_threshold);
    }

    void choose_mains() {
        // clusters no merge touched are still in order (one linear check); only the others are sorted, on a few host
        // threads when there is enough of it (spawning a thread per core costs more than most passes' sorting)
        std::vector<uint32_t> big;
        size_t work = 0;
        for (uint32_t i = 0; i < clusters.size(); ++i) {
            if (inner_parallel && clusters[i].seqs.size() >= 4096) { big.push_back(i); work += clusters[i].seqs.size(); continue; }
            clusters[i].main = get_main_seq(clusters[i].seqs, P->repr_percentile);
        }
        parallel_for(big.size(), work >= (1u << 18) ? 16 : 1, [&](size_t t) { cl &c = clusters[big[t]]; c.main = get_main_seq(c.seqs, P->repr_percentile); });
    }

    // the merge loop's head, cluster.cpp:171: `while (thr >= min_bv_threshold || last)`
    void next_merge(double t) {
        if (!(t >= P->min_bv_threshold || last)) { stage = DONE; return; }
        stage = MERGE;
        const uint32_t nc = (uint32_t)clusters.size();
        items.resize(nc);
        for (uint32_t i = 0; i < nc; ++i) items[i] = (uint32_t)clusters[i].main.id;     // main_seq.rev ignored (:197)
        begin_pass(t);
    }

    void end_pass() {
        if (stage == INITIAL) {                                   // cluster.cpp:124-166
            std::vector<int32_t> slot(n, -1);
            std::vector<uint32_t> size(n, 0);
            for (uint32_t i = 0; i < n; ++i) ++size[owner[i]];
            for (uint32_t i = 0; i < n; ++i)
                if (owner[i] == i) {
                    slot[i] = (int32_t)clusters.size();
                    clusters.push_back(cl{{(int32_t)i, 0}, {}});
                    clusters.back().seqs.reserve(size[i]);
                    clusters.back().seqs.push_back(cseq{(int32_t)i, 0});
                }
            for (uint32_t i = 0; i < n; ++i)
                if (owner[i] != i) clusters[slot[owner[i]]].seqs.push_back(cseq{(int32_t)i, rev[i]});
            choose_mains();
            next_merge(P->bv_threshold - P->bv_falloff);
            return;
        }
        const uint32_t nc = (uint32_t)clusters.size();            // cluster.cpp:171-256
        std::vector<cl> merged;
        std::vector<int32_t> slot(nc, -1);
        for (uint32_t i = 0; i < nc; ++i)
            if (owner[i] == i) { slot[i] = (int32_t)merged.size(); merged.push_back(cl{{0, 0}, {}}); merged.back().seqs = std::move(clusters[i].seqs); }
        for (uint32_t i = 0; i < nc; ++i) {
            if (owner[i] == i) continue;
            cl &dst = merged[slot[owner[i]]];
            for (cseq s : clusters[i].seqs) {                     // :227-238
                if (rev[i]) s.rev = !s.rev;
                dst.seqs.push_back(s);
            }
        }
        clusters.swap(merged);
        choose_mains();
        if (last) { stage = DONE; return; }
        double t = thr - P->bv_falloff;                           // :251-255
        if (t < P->min_bv_threshold && !last) { last = true; t = 0.0; }
        next_merge(t);
    }

    // advance until the job needs a rectangle evaluated (true, *out) or is finished (false)
    bool step(request **out) {
        while (stage != DONE) {
            struct lapse { double &acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                           ~lapse() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } };
            const bool ending = phase == ROUND && remaining.empty();
            lapse L{t_phase[ending ? 4 : (int)phase]};
            switch (phase) {
            case ROUND: {
                if (remaining.empty()) { end_pass(); break; }
                counters[3]++;
                B = (uint32_t)std::min<size_t>(batch_now, remaining.size());
                {
                    // seeds x candidates x strands below the evaluator's per-launch bounds even if every pair survives
                    const uint64_t per_seed = (uint64_t)remaining.size() * (X->both ? 2u : 1u);
                    const uint64_t cap = thr == 0.0 ? (64ull << 20) : (1ull << 31);
                    B = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(B, cap / std::max<uint64_t>(per_seed, 1)));
                }
                seeds_local.resize(B);
                for (uint32_t s = 0; s < B; ++s) seeds_local[s] = items[remaining[s]];
                taken.assign(B, 0);
                if (B > 1) {                                      // ---- level 1: seeds x seeds
                    rq.seeds.resize(B);
                    for (uint32_t s = 0; s < B; ++s) rq.seeds[s] = rid(seeds_local[s]);
                    rq.cands.clear();
                    rq.triangular = true; rq.thr = thr;
                    phase = WAIT_L1;
                    *out = &rq;
                    return true;
                }
                phase = FOUNDERS;
                break;
            }
            case WAIT_L1: {
                // hits grouped by seed; forward verdict wins over reverse (cluster.cpp:19-40 before :43)
                const std::vector<hit_t> &hits = rq.hits;
                size_t h = 0;
                for (uint32_t s = 0; s < B; ++s) {
                    while (h < hits.size() && hits[h].seed < s) ++h;
                    if (taken[s]) continue;
                    for (size_t q = h; q < hits.size() && hits[q].seed == s; ++q) {
                        uint32_t c = hits[q].cand;
                        if (taken[c]) continue;
                        taken[c] = 1;
                        owner[remaining[c]] = remaining[s];
                        rev[remaining[c]] = hits[q].rev;
                    }
                }
                phase = FOUNDERS;
                break;
            }
            case FOUNDERS: {
                founders.clear();
                for (uint32_t s = 0; s < B; ++s) if (!taken[s]) founders.push_back(s);
                {
                    uint32_t want = 8;
                    while (want < 2 * founders.size() && want < max_batch()) want *= 2;
                    batch_now = std::min(want, max_batch());
                }
                const uint32_t nrest = (uint32_t)remaining.size() - B;
                next.clear();
                if (nrest > 0) {                                  // ---- level 2: founders x rest
                    rq.seeds.resize(founders.size());
                    for (size_t f = 0; f < founders.size(); ++f) rq.seeds[f] = rid(seeds_local[founders[f]]);
                    rq.cands.resize(nrest);
                    for (uint32_t c = 0; c < nrest; ++c) rq.cands[c] = rid(items[remaining[B + c]]);
                    rq.triangular = false; rq.thr = thr;
                    phase = WAIT_L2;
                    *out = &rq;
                    return true;
                }
                remaining.swap(next);
                phase = ROUND;
                break;
            }
            case WAIT_L2: {
                const uint32_t nrest = (uint32_t)remaining.size() - B;
                // founders in order, the first accepting founder wins, forward before reverse: the smallest (founder, strand)
                // key per candidate, whatever order the hits arrive in
                best.assign(nrest, 0xFFFFFFFFu);
                for (const hit_t &q : rq.hits) best[q.cand] = std::min(best[q.cand], (q.seed << 1) | q.rev);
                for (uint32_t c = 0; c < nrest; ++c) {
                    if (best[c] == 0xFFFFFFFFu) { next.push_back(remaining[B + c]); continue; }
                    owner[remaining[B + c]] = remaining[founders[best[c] >> 1]];
                    rev[remaining[B + c]] = (uint8_t)(best[c] & 1u);
                }
                remaining.swap(next);
                phase = ROUND;
                break;
            }
            }
        }
        return false;
    }

    rattle_cluster_set *flatten() const {
        rattle_cluster_set *R = (rattle_cluster_set *)calloc(1, sizeof(rattle_cluster_set));
        size_t nm = 0;
        for (auto &c : clusters) nm += c.seqs.size();
        R->n_clusters = (uint32_t)clusters.size();
        R->main_id = (int32_t *)malloc(sizeof(int32_t) * std::max<size_t>(1, clusters.size()));
        R->main_rev = (uint8_t *)malloc(std::max<size_t>(1, clusters.size()));
        R->offsets = (uint32_t *)malloc(sizeof(uint32_t) * (clusters.size() + 1));
        R->member_id = (int32_t *)malloc(sizeof(int32_t) * std::max<size_t>(1, nm));
        R->member_rev = (uint8_t *)malloc(std::max<size_t>(1, nm));
        uint32_t p = 0;
        for (size_t c = 0; c < clusters.size(); ++c) {
            R->main_id[c] = clusters[c].main.id;
            R->main_rev[c] = clusters[c].main.rev;
            R->offsets[c] = p;
            for (auto &s : clusters[c].seqs) { R->member_id[p] = s.id; R->member_rev[p] = s.rev; ++p; }
        }
        R->offsets[clusters.size()] = p;
        memcpy(R->counters, counters, sizeof(counters));
        return R;
    }
};

// all jobs in lockstep: every turn, the rectangles the unfinished jobs wait for are evaluated together
int run_jobs(rattle_ctx *ctx, const rattle_cluster_params *P, std::vector<job> &jobs, bool shard_level2) {
    evaluator E{ctx, P};
    std::vector<uint32_t> active(jobs.size());
    for (uint32_t i = 0; i < jobs.size(); ++i) { active[i] = i; jobs[i].inner_parallel = jobs.size() == 1; jobs[i].start(); }
    std::vector<request *> want, reqs;
    static const bool timing = getenv("RATTLE_TIMING") != nullptr;
    double t_steps = 0;
    while (!active.empty()) {
        const auto t0 = std::chrono::steady_clock::now();
        want.assign(active.size(), nullptr);
        parallel_for(active.size(), active.size() > 64 ? 16 : 1, [&](size_t i) {        // a step is microseconds of host work
            request *q = nullptr;
            if (jobs[active[i]].step(&q)) want[i] = q;
        });
        reqs.clear();
        std::vector<uint32_t> still;
        for (size_t i = 0; i < active.size(); ++i) if (want[i]) { reqs.push_back(want[i]); still.push_back(active[i]); }
        active.swap(still);
        t_steps += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (reqs.empty()) break;
        RT_TRY(E.run(reqs, shard_level2 && jobs.size() == 1 && !reqs[0]->triangular));
    }
    if (!jobs.empty()) jobs[0].counters[4] += E.launches;
    if (timing && jobs.size() == 1)
        fprintf(stderr, "[rattle]   host steps: round set-up %.1f, level-1 resolve %.1f, founders + request %.1f, level-2 resolve %.1f, end of pass %.1f ms\n",
                jobs[0].t_phase[0], jobs[0].t_phase[1], jobs[0].t_phase[2], jobs[0].t_phase[3], jobs[0].t_phase[4]);
    if (timing)
        fprintf(stderr, "[rattle]   %zu job(s): host steps %.1f ms | build+upload %.1f, filter %.1f, count pass %.1f, host bound test %.1f, full pass + verdicts %.1f ms\n",
                jobs.size(), t_steps, E.t_split[0], E.t_split[1], E.t_split[2], E.t_split[3], E.t_split[4]);
    return 0;
}

}  // namespace

int cluster_driver(rattle_ctx *ctx, const rattle_cluster_params *P, const uint32_t *subset, uint32_t n_subset,
                   rattle_cluster_set **out) {
    std::vector<job> jobs(1);
    jobs[0].X = &ctx->idx; jobs[0].P = P; jobs[0].subset = subset;
    jobs[0].n = subset ? n_subset : ctx->idx.n;
    RT_TRY(run_jobs(ctx, P, jobs, true));
    *out = jobs[0].flatten();
    return 0;
}

// subsets which[0..n_which) of (ids, sub_off), clustered independently and in lockstep; outs[which[i]] receives the result
int cluster_driver_many(rattle_ctx *ctx, const rattle_cluster_params *P, const uint32_t *ids, const uint64_t *sub_off,
                        const uint32_t *which, uint32_t n_which, rattle_cluster_set **outs) {
    static const uint32_t none = 0;
    std::vector<job> jobs(n_which);
    for (uint32_t i = 0; i < n_which; ++i) {
        const uint32_t g = which[i];
        jobs[i].X = &ctx->idx; jobs[i].P = P;
        jobs[i].n = (uint32_t)(sub_off[g + 1] - sub_off[g]);
        jobs[i].subset = jobs[i].n ? ids + sub_off[g] : &none;
    }
    RT_TRY(run_jobs(ctx, P, jobs, false));
    for (uint32_t i = 0; i < n_which; ++i) outs[which[i]] = jobs[i].flatten();
    return 0;
}

}  // namespace rattle
