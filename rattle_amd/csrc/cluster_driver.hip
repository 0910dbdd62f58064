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
#include <algorithm>
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

struct hit_t { uint32_t seed, cand; uint8_t rev; };

struct cseq { int32_t id; uint8_t rev; };

struct driver {
    rattle_ctx *ctx;
    const rattle_cluster_params *P;
    const uint32_t *subset;          // local id -> loaded read id (nullptr = identity)
    uint32_t n;
    uint64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint16_t lut[4097];
    double lut_thr = -1.0;
    std::vector<uint32_t> tmp_seed, tmp_cand, tmp_first;

    uint32_t rid(uint32_t local) const { return subset ? subset[local] : local; }
    uint32_t rlen(uint32_t local) const { return ctx->idx.h_len[rid(local)]; }

    // min_common_lut[m] = smallest c with double(c)/double(m) >= thr  (cluster.cpp:19,43)
    int set_threshold(double thr) {
        if (thr == lut_thr) return 0;
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
        lut_thr = thr;
        RT_TRY(ctx->d_lut.reserve(4097));
        RT_HIP(hipMemcpyAsync(ctx->d_lut.p, lut, sizeof(lut), hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(hipStreamSynchronize(ctx->stream));   // lut[] may be rewritten by the next call
        return 0;
    }

    // Evaluate cluster_together for every (seed s, cand c >= first[s]); seeds/cands are LOCAL ids.
    // shard: this rank scores the candidates at positions rank, rank + nranks, ... and the hits of all ranks
    // are all-gathered (every rank ends up with the same list; the callers sort it).
    std::vector<uint32_t> sh_cands, sh_first;
    int eval(const std::vector<uint32_t> &seeds, const uint32_t *cands, uint32_t n_cands, const std::vector<uint32_t> &first,
             double thr, std::vector<hit_t> &hits, bool shard = false) {
        const int R = ctx->xchg.nranks, r = ctx->xchg.rank;
        if (!shard || R <= 1) return eval_local(seeds, cands, n_cands, first, thr, hits);
        sh_cands.clear();
        for (uint32_t c = (uint32_t)r; c < n_cands; c += (uint32_t)R) sh_cands.push_back(cands[c]);
        sh_first.resize(first.size());
        for (size_t i = 0; i < first.size(); ++i) sh_first[i] = first[i] <= (uint32_t)r ? 0u : (first[i] - (uint32_t)r + (uint32_t)R - 1u) / (uint32_t)R;
        uint64_t before[3] = {counters[0], counters[1], counters[2]};
        RT_TRY(eval_local(seeds, sh_cands.data(), (uint32_t)sh_cands.size(), sh_first, thr, hits));
        // payload: three counter deltas, then (seed, global candidate position, strand) triples
        std::vector<uint8_t> mine(24 + hits.size() * 12);
        for (int i = 0; i < 3; ++i) { const uint64_t d = counters[i] - before[i]; memcpy(mine.data() + 8 * i, &d, 8); counters[i] = before[i]; }
        for (size_t i = 0; i < hits.size(); ++i) {
            const uint32_t t[3] = {hits[i].seed, hits[i].cand * (uint32_t)R + (uint32_t)r, hits[i].rev};
            memcpy(mine.data() + 24 + 12 * i, t, 12);
        }
        std::vector<std::vector<uint8_t>> all;
        RT_TRY(xchg_allgatherv(ctx, mine, all));
        hits.clear();
        for (const std::vector<uint8_t> &b : all) {
            if (b.size() < 24 || (b.size() - 24) % 12) { set_error("cluster exchange: malformed hit list"); return RATTLE_ERR_HIP; }
            for (int i = 0; i < 3; ++i) { uint64_t d; memcpy(&d, b.data() + 8 * i, 8); counters[i] += d; }
            for (size_t at = 24; at < b.size(); at += 12) {
                uint32_t t[3];
                memcpy(t, b.data() + at, 12);
                hits.push_back(hit_t{t[0], t[1], (uint8_t)t[2]});
            }
        }
        return 0;
    }

    int eval_local(const std::vector<uint32_t> &seeds, const uint32_t *cands, uint32_t n_cands, const std::vector<uint32_t> &first,
                   double thr, std::vector<hit_t> &hits) {
        hits.clear();
        uint32_t ns = (uint32_t)seeds.size();
        if (ns == 0 || n_cands == 0) return 0;
        hipStream_t st = ctx->stream;
        tmp_seed.resize(ns);
        for (uint32_t i = 0; i < ns; ++i) tmp_seed[i] = rid(seeds[i]);
        const uint32_t *cand_rids = cands;
        if (subset) {
            tmp_cand.resize(n_cands);
            for (uint32_t i = 0; i < n_cands; ++i) tmp_cand[i] = subset[cands[i]];
            cand_rids = tmp_cand.data();
        }
        RT_TRY(ctx->d_seed.reserve(ns));
        RT_TRY(ctx->d_first.reserve(ns));
        RT_TRY(ctx->d_cand.reserve(n_cands));
        RT_TRY(ctx->d_counter.reserve(4));
        RT_TRY(ctx->h_counter.reserve(4));
        RT_HIP(hipMemcpyAsync(ctx->d_seed.p, tmp_seed.data(), ns * 4, hipMemcpyHostToDevice, st));
        RT_HIP(hipMemcpyAsync(ctx->d_first.p, first.data(), ns * 4, hipMemcpyHostToDevice, st));
        RT_HIP(hipMemcpyAsync(ctx->d_cand.p, cand_rids, (size_t)n_cands * 4, hipMemcpyHostToDevice, st));
        uint64_t npairs = 0;
        for (uint32_t i = 0; i < ns; ++i) npairs += n_cands > first[i] ? n_cands - first[i] : 0;
        counters[0] += npairs;

        // survivor capacity: grow and retry on overflow (count is exact even when truncated)
        size_t cap = std::max<size_t>(ctx->d_surv.cap / 2, 1u << 20);
        uint32_t nsurv = 0;
        while (true) {
            RT_TRY(ctx->d_surv.reserve(cap * 2));
            cap = ctx->d_surv.cap / 2;
            RT_HIP(hipMemsetAsync(ctx->d_counter.p, 0, 16, st));
            RT_TRY(launch_bv_filter(ctx, ns, n_cands, thr == 0.0 ? 1 : 0, false, true, (uint32_t)std::min<size_t>(cap, 0xFFFFFFF0u)));
            counters[4]++;
            RT_HIP(hipMemcpyAsync(ctx->h_counter.p, ctx->d_counter.p, 4, hipMemcpyDeviceToHost, st));
            RT_HIP(hipStreamSynchronize(st));
            nsurv = ctx->h_counter.p[0];
            if ((uint64_t)nsurv > npairs * 2) { set_error("bv_filter: survivor counter overflow"); return RATTLE_ERR_HIP; }
            if (nsurv <= cap) break;
            cap = (size_t)nsurv + nsurv / 8;
        }
        if (nsurv == 0) return 0;
        counters[1] += nsurv;

        RT_TRY(ctx->d_pi.reserve(nsurv));
        RT_TRY(ctx->d_pj.reserve(nsurv));
        RT_TRY(ctx->d_ps.reserve(nsurv));
        RT_TRY(ctx->d_res.reserve((size_t)nsurv * 4));
        RT_TRY(ctx->h_surv.reserve((size_t)nsurv * 2));
        RT_TRY(ctx->h_res.reserve((size_t)nsurv * 4));
        hipLaunchKernelGGL(expand_pairs_kernel, dim3((nsurv + 255) / 256), dim3(256), 0, st, ctx->d_surv.p, nsurv,
                           ctx->d_seed.p, ctx->d_cand.p, ctx->d_pi.p, ctx->d_pj.p, ctx->d_ps.p);
        RT_HIP(hipMemcpyAsync(ctx->h_surv.p, ctx->d_surv.p, (size_t)nsurv * 8, hipMemcpyDeviceToHost, st));
        // ---- pass 1: |common| of every surviving pair.  bases <= k * |LIS| <= k * |common| (similarity.cpp:52-85), so a pair
        // with double(k * |common|) / min_len < t_s cannot pass cluster.cpp:23-27 whatever its chain looks like: exact
        // rejection without the patience search.  In the low-threshold merge passes that is nearly every pair.
        RT_TRY(launch_pair_count(ctx, nsurv));
        counters[4] += 2;
        RT_HIP(hipMemcpyAsync(ctx->h_res.p, ctx->d_res.p, (size_t)nsurv * 4, hipMemcpyDeviceToHost, st));
        RT_HIP(hipStreamSynchronize(st));
        const double t_s = P->t_s, t_v = P->t_v;
        const uint32_t kk = (uint32_t)ctx->idx.k;
        uint64_t alg_bytes = 0;
        // tens of millions of survivors per level-2 round: chunks in parallel, results concatenated in pair order
        const uint32_t chunk = 1u << 16;
        const uint32_t n_chunks = (nsurv + chunk - 1) / chunk;
        std::vector<std::vector<uint32_t>> part(n_chunks);
        std::vector<uint64_t> part_matches(n_chunks, 0), part_bytes(n_chunks, 0);
        parallel_for(n_chunks, n_chunks > 1 ? 0 : 1, [&](size_t ch) {
            uint64_t mt = 0, ab = 0;
            const uint32_t p1 = std::min<uint32_t>(nsurv, (uint32_t)(ch + 1) * chunk);
            for (uint32_t p = (uint32_t)ch * chunk; p < p1; ++p) {
                const uint32_t a = ctx->h_surv.p[2 * (size_t)p], c = ctx->h_surv.p[2 * (size_t)p + 1];
                const uint64_t M = (uint32_t)ctx->h_res.p[p];
                mt += M;
                const uint32_t li = rlen(seeds[a >> 1]), lj = rlen(cands[c]);
                ab += 8ull * ((li > kk ? li - kk : 0) + (lj > kk ? lj - kk : 0));
                const double mn = (double)std::min<size_t>(li, lj);
                if (double(M * kk) / mn >= t_s) part[ch].push_back(p);
            }
            part_matches[ch] = mt; part_bytes[ch] = ab;
        });
        std::vector<uint32_t> todo;
        for (uint32_t ch = 0; ch < n_chunks; ++ch) {
            todo.insert(todo.end(), part[ch].begin(), part[ch].end());
            counters[2] += part_matches[ch];
            alg_bytes += part_bytes[ch];
        }
        ctx->stats[K_SCORE].bytes += alg_bytes;
        const uint32_t n2 = (uint32_t)todo.size();
        counters[5] += n2;
        if (n2 == 0) return 0;
        // ---- pass 2: the reference's full comparison for the pairs that can still be accepted
        {
            std::vector<uint32_t> pi2(n2), pj2(n2);
            std::vector<uint8_t> ps2(n2);
            for (uint32_t q = 0; q < n2; ++q) {
                const uint32_t a = ctx->h_surv.p[2 * (size_t)todo[q]], c = ctx->h_surv.p[2 * (size_t)todo[q] + 1];
                pi2[q] = rid(seeds[a >> 1]); pj2[q] = rid(cands[c]); ps2[q] = (uint8_t)(a & 1u);
            }
            RT_TRY(ctx->d_var.reserve(n2));
            RT_TRY(ctx->h_var.reserve(n2));
            RT_HIP(hipMemcpyAsync(ctx->d_pi.p, pi2.data(), (size_t)n2 * 4, hipMemcpyHostToDevice, st));
            RT_HIP(hipMemcpyAsync(ctx->d_pj.p, pj2.data(), (size_t)n2 * 4, hipMemcpyHostToDevice, st));
            RT_HIP(hipMemcpyAsync(ctx->d_ps.p, ps2.data(), (size_t)n2, hipMemcpyHostToDevice, st));
            RT_TRY(launch_pair_score(ctx, n2));
            counters[4]++;
            RT_HIP(hipMemcpyAsync(ctx->h_res.p, ctx->d_res.p, (size_t)n2 * 16, hipMemcpyDeviceToHost, st));
            RT_HIP(hipMemcpyAsync(ctx->h_var.p, ctx->d_var.p, (size_t)n2 * 8, hipMemcpyDeviceToHost, st));
            RT_HIP(hipStreamSynchronize(st));             // also: pi2 / pj2 / ps2 may go out of scope
        }
        // pairs whose match list did not fit LDS: rerun through the global-scratch variant
        std::vector<uint32_t> big;
        uint32_t big_m = 0;
        for (uint32_t q = 0; q < n2; ++q)
            if (ctx->h_res.p[4 * (size_t)q] == INT32_MIN) {
                big.push_back(q);
                big_m = std::max(big_m, (uint32_t)ctx->h_res.p[4 * (size_t)q + 3]);
            }
        if (!big.empty()) {
            RT_TRY(launch_pair_score_oversize(ctx, big, big_m));
            counters[4]++;
            RT_HIP(hipMemcpyAsync(ctx->h_res.p, ctx->d_res.p, (size_t)n2 * 16, hipMemcpyDeviceToHost, st));
            RT_HIP(hipMemcpyAsync(ctx->h_var.p, ctx->d_var.p, (size_t)n2 * 8, hipMemcpyDeviceToHost, st));
            RT_HIP(hipStreamSynchronize(st));
        }
        // cluster.cpp:23-36 / :47-61 on the host in the reference's double arithmetic
        for (uint32_t q = 0; q < n2; ++q) {
            const uint32_t a = ctx->h_surv.p[2 * (size_t)todo[q]], c = ctx->h_surv.p[2 * (size_t)todo[q] + 1];
            const int32_t *r = ctx->h_res.p + 4 * (size_t)q;
            const uint32_t li = rlen(seeds[a >> 1]), lj = rlen(cands[c]);
            const double mn = (double)std::min<size_t>(li, lj);
            const double score = P->use_hc ? double(r[1]) / mn : double(r[0]) / mn;
            if (score >= t_s && ctx->h_var.p[q] < t_v) hits.push_back(hit_t{a >> 1, c, (uint8_t)(a & 1u)});
        }
        return 0;
    }

    // cluster.cpp:67-91
    cseq get_main_seq(std::vector<cseq> &seqs, double repr_percentile) const {
        cseq old = seqs[0];
        std::stable_sort(seqs.begin(), seqs.end(), [](const cseq &a, const cseq &b) { return a.id > b.id; });
        std::stable_sort(seqs.begin(), seqs.end(), [this](const cseq &a, const cseq &b) { return rlen(a.id) > rlen(b.id); });
        int nsid = seqs.size() * repr_percentile;
        cseq ns = seqs[nsid];
        while (ns.rev != old.rev && (size_t)nsid < seqs.size() - 1) { nsid++; ns = seqs[nsid]; }
        if ((size_t)nsid == seqs.size() - 1) return old;
        return ns;
    }

    // One greedy pass over `items` (local read id compared for each item).  owner[i] = index of
    // the founder item that absorbed i (owner[i]==i for founders); rev[i] = strand of the match.
    int greedy_pass(const std::vector<uint32_t> &items, double thr, std::vector<uint32_t> &owner, std::vector<uint8_t> &rev) {
        RT_TRY(set_threshold(thr));
        uint32_t m = (uint32_t)items.size();
        owner.resize(m);
        rev.assign(m, 0);
        for (uint32_t i = 0; i < m; ++i) owner[i] = i;
        static const uint32_t batch = getenv("RATTLE_SEED_BATCH") ? (uint32_t)atoi(getenv("RATTLE_SEED_BATCH")) : 256;
        std::vector<uint32_t> remaining(m), next, seeds_local, first, cands_local, founders, ffirst;
        for (uint32_t i = 0; i < m; ++i) remaining[i] = i;
        std::vector<uint8_t> taken;
        std::vector<hit_t> hits;
        while (!remaining.empty()) {
            counters[3]++;
            uint32_t B = (uint32_t)std::min<size_t>(batch, remaining.size());
            {
                // every pair can survive the filter (the thr == 0 pass lets all of them through, cluster.cpp:19,43): keep
                // seeds x candidates x strands below the 32-bit survivor counter, and below 64 M where all of them DO survive
                const uint64_t per_seed = (uint64_t)remaining.size() * (ctx->idx.both ? 2u : 1u);
                const uint64_t cap = thr == 0.0 ? (64ull << 20) : (1ull << 31);
                B = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(B, cap / std::max<uint64_t>(per_seed, 1)));
            }
            // ---- level 1: seeds x seeds
            seeds_local.resize(B);
            first.resize(B);
            for (uint32_t s = 0; s < B; ++s) { seeds_local[s] = items[remaining[s]]; first[s] = s + 1; }
            founders.clear();
            taken.assign(B, 0);
            if (B > 1) {
                RT_TRY(eval(seeds_local, seeds_local.data(), B, first, thr, hits));
                // group hits by seed; forward verdict wins over reverse (cluster.cpp:19-40 before :43)
                std::sort(hits.begin(), hits.end(), [](const hit_t &a, const hit_t &b) {
                    return a.seed != b.seed ? a.seed < b.seed : (a.cand != b.cand ? a.cand < b.cand : a.rev < b.rev);
                });
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
            }
            for (uint32_t s = 0; s < B; ++s) if (!taken[s]) founders.push_back(s);
            // ---- level 2: founders x rest
            uint32_t nrest = (uint32_t)remaining.size() - B;
            next.clear();
            if (nrest > 0) {
                cands_local.resize(nrest);
                for (uint32_t c = 0; c < nrest; ++c) cands_local[c] = items[remaining[B + c]];
                std::vector<uint32_t> fl(founders.size());
                ffirst.assign(founders.size(), 0);
                for (size_t f = 0; f < founders.size(); ++f) fl[f] = seeds_local[founders[f]];
                RT_TRY(eval(fl, cands_local.data(), nrest, ffirst, thr, hits, true));
                std::sort(hits.begin(), hits.end(), [](const hit_t &a, const hit_t &b) {
                    return a.seed != b.seed ? a.seed < b.seed : (a.cand != b.cand ? a.cand < b.cand : a.rev < b.rev);
                });
                taken.assign(nrest, 0);
                for (const hit_t &q : hits) {          // founders in order; first accepting founder wins
                    if (taken[q.cand]) continue;
                    taken[q.cand] = 1;
                    owner[remaining[B + q.cand]] = remaining[founders[q.seed]];
                    rev[remaining[B + q.cand]] = q.rev;
                }
                for (uint32_t c = 0; c < nrest; ++c) if (!taken[c]) next.push_back(remaining[B + c]);
            }
            remaining.swap(next);
        }
        return 0;
    }
};

int cluster_driver(rattle_ctx *ctx, const rattle_cluster_params *P, const uint32_t *subset, uint32_t n_subset,
                   rattle_cluster_set **out) {
    driver D;
    D.ctx = ctx; D.P = P; D.subset = subset;
    D.n = subset ? n_subset : ctx->idx.n;
    const uint32_t n = D.n;
    struct cl { cseq main; std::vector<cseq> seqs; };
    std::vector<cl> clusters;
    std::vector<uint32_t> owner;
    std::vector<uint8_t> rev;

    // ---- initial pass, cluster.cpp:124-166
    {
        std::vector<uint32_t> items(n);
        for (uint32_t i = 0; i < n; ++i) items[i] = i;
        RT_TRY(D.greedy_pass(items, P->bv_threshold, owner, rev));
        std::vector<int32_t> slot(n, -1);
        for (uint32_t i = 0; i < n; ++i)
            if (owner[i] == i) { slot[i] = (int32_t)clusters.size(); clusters.push_back(cl{{(int32_t)i, 0}, {cseq{(int32_t)i, 0}}}); }
        for (uint32_t i = 0; i < n; ++i)
            if (owner[i] != i) clusters[slot[owner[i]]].seqs.push_back(cseq{(int32_t)i, rev[i]});
        parallel_for(clusters.size(), 0, [&](size_t i) { clusters[i].main = D.get_main_seq(clusters[i].seqs, P->repr_percentile); });
    }
    // ---- merge passes, cluster.cpp:171-256
    double thr = P->bv_threshold - P->bv_falloff;
    bool last = false;
    while (thr >= P->min_bv_threshold || last) {
        uint32_t nc = (uint32_t)clusters.size();
        std::vector<uint32_t> items(nc);
        for (uint32_t i = 0; i < nc; ++i) items[i] = (uint32_t)clusters[i].main.id;     // main_seq.rev ignored (:197)
        RT_TRY(D.greedy_pass(items, thr, owner, rev));
        std::vector<cl> merged;
        std::vector<int32_t> slot(nc, -1);
        for (uint32_t i = 0; i < nc; ++i)
            if (owner[i] == i) { slot[i] = (int32_t)merged.size(); merged.push_back(cl{{0, 0}, {}}); merged.back().seqs = std::move(clusters[i].seqs); }
        for (uint32_t i = 0; i < nc; ++i) {
            if (owner[i] == i) continue;
            cl &dst = merged[slot[owner[i]]];
            for (cseq s : clusters[i].seqs) {            // :227-238
                if (rev[i]) s.rev = !s.rev;
                dst.seqs.push_back(s);
            }
        }
        parallel_for(merged.size(), 0, [&](size_t i) { merged[i].main = D.get_main_seq(merged[i].seqs, P->repr_percentile); });
        clusters.swap(merged);
        if (last) break;
        thr -= P->bv_falloff;                             // :251-255
        if (thr < P->min_bv_threshold && !last) { last = true; thr = 0.0; }
    }

    // ---- flatten
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
    memcpy(R->counters, D.counters, sizeof(D.counters));
    *out = R;
    return 0;
}

}  // namespace rattle
