// Host orchestration of `rattle correct`, /root/reference/correct.cpp:311-563, over kernels C and D.
//
// The reference runs a queue of packs through worker threads, two POAs per pack
// (correct.cpp:398-405, 428-436) plus one per multi-pack cluster (:520-532).  Packs are
// independent, so here each POA stage is ONE device pass over all packs, and the post-MSA logic
// (fix_msa_ends :32-92, column vote :94-193, per-read correction :196-309) is kernel D
// (post_msa.hip), one workgroup per pack.  Sequences, qualities, MSA columns and row matrices stay
// in HBM from the upload of the reads to the download of the corrected reads and consensi; between
// stages the host only sees lengths (pack widths, corrected read lengths, trim counts), from which it
// plans the next stage (pack order, length-sorted order for POA #2, gather descriptors).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "common.h"

namespace rattle {

namespace {

struct hread {
    std::string seq, qual;
    int32_t rid;
};

struct sref { int32_t rid; uint8_t rev; };

inline char comp_base(char c) {                                                  // utils.hpp:8-14
    switch (c) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'T': return 'A';
        case 'G': return 'C';
        case 'U': return 'A';
    }
    return c;
}

// pack member in the orientation the reference aligns it in (correct.cpp:343-346), trimmed by [tf, len - tb)
hread oriented_read(const uint8_t *seq, const uint8_t *qual, const uint64_t *off, sref r, uint32_t tf, uint32_t tb) {
    hread h;
    h.rid = r.rid;
    h.seq.assign((const char *)seq + off[r.rid], (const char *)seq + off[r.rid + 1]);
    h.qual.assign((const char *)qual + off[r.rid], (const char *)qual + off[r.rid + 1]);
    if (r.rev) {
        std::string rc(h.seq.size(), 'A');
        for (size_t t = 0; t < h.seq.size(); ++t) rc[t] = comp_base(h.seq[h.seq.size() - 1 - t]);
        h.seq.swap(rc);
        std::reverse(h.qual.begin(), h.qual.end());
    }
    if (tf || tb) {
        const size_t n = h.seq.size();
        const size_t a = std::min<size_t>(tf, n), b = n - std::min<size_t>(tb, n - a);
        h.seq = h.seq.substr(a, b - a);
        h.qual = h.qual.substr(a, b - a);
    }
    return h;
}

void fill_set(rattle_read_set &S, const std::vector<hread> &v, const std::vector<int32_t> &cid, const std::vector<int32_t> &nr) {
    S.n = (uint32_t)v.size();
    size_t n = std::max<size_t>(1, v.size());
    S.read_id = (int32_t *)malloc(n * 4); S.cluster_id = (int32_t *)malloc(n * 4); S.n_reads = (int32_t *)malloc(n * 4);
    S.off = (uint64_t *)malloc((v.size() + 1) * 8);
    uint64_t tot = 0;
    for (size_t i = 0; i < v.size(); ++i) { S.off[i] = tot; tot += v[i].seq.size(); }
    S.off[v.size()] = tot;
    S.seq = (char *)malloc(tot + 1); S.qual = (char *)malloc(tot + 1);
    for (size_t i = 0; i < v.size(); ++i) {
        S.read_id[i] = v[i].rid; S.cluster_id[i] = cid[i]; S.n_reads[i] = nr.empty() ? 0 : nr[i];
        const uint64_t p = S.off[i];
        memcpy(S.seq + p, v[i].seq.data(), v[i].seq.size());
        const size_t ql = std::min(v[i].qual.size(), v[i].seq.size());
        memcpy(S.qual + p, v[i].qual.data(), ql);
        if (ql < v[i].seq.size()) memset(S.qual + p + ql, '!', v[i].seq.size() - ql);
    }
    S.seq[tot] = 0; S.qual[tot] = 0;
}

// constants of kernel D: phred_symbol thresholds (host libm) and phred_err per quality byte (utils.cpp:10-13)
int ensure_post_constants(rattle_ctx *ctx) {
    if (ctx->phred_ready) return 0;
    build_phred_table(ctx->phred);
    const phred_table &T = ctx->phred;
    double perr[256];
    for (int c = 0; c < 256; ++c) { double q = (char)c - 33; perr[c] = pow(10.0, -q / 10.0); }
    RT_TRY(ctx->d_phred_lo.reserve(T.lo.size())); RT_TRY(ctx->d_perr.reserve(256));
    RT_TRY(ctx->d_exc_bits.reserve(T.exc_bits.size() + 1)); RT_TRY(ctx->d_exc_val.reserve(T.exc_val.size() + 1));
    RT_HIP(hipMemcpy(ctx->d_phred_lo.p, T.lo.data(), T.lo.size() * 8, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(ctx->d_perr.p, perr, sizeof(perr), hipMemcpyHostToDevice));
    if (!T.exc_bits.empty()) {
        RT_HIP(hipMemcpy(ctx->d_exc_bits.p, T.exc_bits.data(), T.exc_bits.size() * 8, hipMemcpyHostToDevice));
        RT_HIP(hipMemcpy(ctx->d_exc_val.p, T.exc_val.data(), T.exc_val.size() * 4, hipMemcpyHostToDevice));
    }
    ctx->phred_ready = true;
    return 0;
}

// One POA stage + kernel D over device-resident sequences.
struct stage {
    std::vector<uint64_t> off;          // [n+1] host copy of the sequence offsets
    std::vector<uint32_t> first;        // [n_packs+1]
    std::vector<uint32_t> width;        // [n_packs] MSA width (after the POA)
    std::vector<uint64_t> moff, coff;   // per pack: matrix byte offset / column-array offset
    uint64_t cells = 0, cols = 0;
    dbuf<uint8_t> seq, qual, rowc, rowq, ccons, cflag, csym, cons_out;
    dbuf<uint64_t> d_off, d_moff, d_coff;
    dbuf<uint32_t> col, d_width, d_first, tfront, tback, olen, cons_len;
    dbuf<int32_t> rfirst, rlast;
    dbuf<double> cerr;
    uint32_t n() const { return (uint32_t)off.size() - 1; }
    uint32_t n_packs() const { return (uint32_t)first.size() - 1; }
    void release() {
        seq.release(); qual.release(); rowc.release(); rowq.release(); ccons.release(); cflag.release(); csym.release();
        cons_out.release(); d_off.release(); d_moff.release(); d_coff.release(); col.release(); d_width.release();
        d_first.release(); tfront.release(); tback.release(); olen.release(); cons_len.release(); rfirst.release();
        rlast.release(); cerr.release();
    }
};

// where a run of gather descriptors reads from
struct gather_part { uint32_t begin, count; const uint8_t *src_seq, *src_qual; };

// gather the stage's sequences (descriptors built by the caller, dst offsets = st.off) and run POA + kernel D
int run_stage(rattle_ctx *ctx, stage &S, const std::vector<gather_desc> &desc, const std::vector<gather_part> &parts,
              int mode, const rattle_correct_params *P, const char *order, uint64_t *counters) {
    hipStream_t st = ctx->stream;
    const uint32_t n = S.n(), np = S.n_packs();
    const uint64_t total = S.off[n];
    S.width.assign(np, 0);
    if (np == 0) return 0;
    dbuf<gather_desc> d_desc;
    RT_TRY(d_desc.reserve(n + 1));
    RT_TRY(S.seq.reserve(total + 64)); RT_TRY(S.d_off.reserve(n + 1)); RT_TRY(S.col.reserve(total + 64));
    RT_TRY(S.d_width.reserve(np)); RT_TRY(S.d_first.reserve(np + 1));
    if (mode == 1) RT_TRY(S.qual.reserve(total + 64));
    if (n) RT_HIP(hipMemcpyAsync(d_desc.p, desc.data(), (size_t)n * sizeof(gather_desc), hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(S.d_off.p, S.off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(S.d_first.p, S.first.data(), (size_t)(np + 1) * 4, hipMemcpyHostToDevice, st));
    for (const gather_part &g : parts)
        RT_TRY(launch_gather(ctx, d_desc.p + g.begin, g.count, g.src_seq, mode == 1 ? g.src_qual : nullptr, S.seq.p, mode == 1 ? S.qual.p : nullptr));
    unsigned long long h_cnt[16];
    {
        phase_timer T("  stage: POA");
        RT_TRY(poa_device_run(ctx, S.seq.p, S.d_off.p, S.off.data(), n, S.first.data(), np, S.col.p, S.d_width.p, S.width.data(), h_cnt));
    }
    d_desc.release();
    counters[0] += h_cnt[0];
    counters[1] += h_cnt[1];
    // layout of the row matrices and per-column arrays
    S.moff.assign(np, 0); S.coff.assign(np, 0);
    uint64_t cells = 0, cols = 0;
    for (uint32_t p = 0; p < np; ++p) {
        S.moff[p] = cells; S.coff[p] = cols;
        cells += ((uint64_t)(S.first[p + 1] - S.first[p]) * S.width[p] + 15) & ~(uint64_t)15;
        cols += S.width[p];
    }
    S.cells = cells; S.cols = cols;
    phase_timer T("  stage: post-MSA kernel");
    RT_TRY(S.rowc.reserve(cells + 64)); RT_TRY(S.d_moff.reserve(np)); RT_TRY(S.d_coff.reserve(np));
    RT_TRY(S.rfirst.reserve(n + 1)); RT_TRY(S.rlast.reserve(n + 1)); RT_TRY(S.ccons.reserve(cols + 64));
    if (mode == 1) {
        RT_TRY(S.rowq.reserve(cells + 64)); RT_TRY(S.tfront.reserve(n + 1)); RT_TRY(S.tback.reserve(n + 1)); RT_TRY(S.olen.reserve(n + 1));
        RT_TRY(S.cflag.reserve(cols + 64)); RT_TRY(S.csym.reserve(cols + 64)); RT_TRY(S.cerr.reserve(cols + 8));
    } else {
        RT_TRY(S.cons_out.reserve(cols + 64)); RT_TRY(S.cons_len.reserve(np));
    }
    RT_HIP(hipMemcpyAsync(S.d_moff.p, S.moff.data(), (size_t)np * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(S.d_coff.p, S.coff.data(), (size_t)np * 8, hipMemcpyHostToDevice, st));
    post_args A;
    memset(&A, 0, sizeof(A));
    A.seq = S.seq.p; A.qual = S.qual.p; A.off = S.d_off.p; A.pack_first = S.d_first.p; A.col = S.col.p; A.width = S.d_width.p;
    A.moff = S.d_moff.p; A.coff = S.d_coff.p; A.rowc = S.rowc.p; A.rowq = S.rowq.p; A.rfirst = S.rfirst.p; A.rlast = S.rlast.p;
    A.tfront = S.tfront.p; A.tback = S.tback.p; A.olen = S.olen.p; A.ccons = S.ccons.p; A.cflag = S.cflag.p; A.csym = S.csym.p;
    A.cerr = S.cerr.p; A.cons_out = S.cons_out.p; A.cons_len = S.cons_len.p; A.perr = ctx->d_perr.p; A.phred_lo = ctx->d_phred_lo.p;
    A.exc_bits = ctx->d_exc_bits.p; A.exc_val = ctx->d_exc_val.p; A.phred_n0 = ctx->phred.n0; A.phred_cnt = (int32_t)ctx->phred.lo.size();
    A.n_exc = (uint32_t)ctx->phred.exc_bits.size();
    memcpy(A.order, order, 6);
    A.min_occ = P->min_occ; A.gap_occ = P->gap_occ; A.err_ratio = P->err_ratio;
    RT_TRY(launch_post_msa(ctx, A, np, mode));
    return 0;
}

}  // namespace

int correct_driver(rattle_ctx *ctx, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint32_t n_reads,
                   uint32_t n_clusters, const uint32_t *coff, const int32_t *mid, const uint8_t *mrev,
                   const rattle_correct_params *P, rattle_correction **out) {
    char order[8] = {0};
    memcpy(order, P->vote_order[0] ? P->vote_order : "U-GTCA", 6);
    for (int i = 0; i < 6; ++i)
        if (!order[i] || !strchr("ACGTU-", order[i])) { set_error("vote_order must be a permutation of ACGTU-"); return RATTLE_ERR_ARG; }
    const int split = P->split > 0 ? P->split : 200;
    rattle_correction *R = (rattle_correction *)calloc(1, sizeof(rattle_correction));
    *out = R;
    hipStream_t st = ctx->stream;
    phase_timer T_all("correct: total");
    RT_TRY(ensure_post_constants(ctx));

    // ---- correct.cpp:328-370 pack building: ids and strands only, the bases stay where they are
    std::vector<int32_t> pk_cid;
    std::vector<sref> S1r, small;
    std::vector<int32_t> small_cid;
    std::vector<uint32_t> cl_p0(n_clusters, 0), cl_np(n_clusters, 0);
    stage S1;
    S1.first.assign(1, 0);
    for (uint32_t c = 0; c < n_clusters; ++c) {
        const uint32_t a = coff[c], b = coff[c + 1];
        const int n = (int)(b - a);
        cl_p0[c] = (uint32_t)pk_cid.size();
        if (n <= 0) continue;
        const int n_files = (n - 1) / split + 1;
        for (int nf = 0; nf < n_files; ++nf) {
            const size_t start = S1r.size();
            for (int j = nf; j < n; j += n_files) {
                const int32_t rid = mid[a + j];
                if (rid < 0 || (uint32_t)rid >= n_reads) { set_error("cluster member id out of range"); return RATTLE_ERR_ARG; }
                S1r.push_back(sref{rid, (uint8_t)(mrev[a + j] ? 1 : 0)});
            }
            if ((int)(S1r.size() - start) > P->min_reads) {                      // :360 strict
                pk_cid.push_back((int32_t)c);
                S1.first.push_back((uint32_t)S1r.size());
                ++cl_np[c];
            } else {
                for (size_t t = start; t < S1r.size(); ++t) { small.push_back(S1r[t]); small_cid.push_back((int32_t)c); }
                S1r.resize(start);
            }
        }
    }
    const uint32_t n_packs = (uint32_t)pk_cid.size(), n1 = (uint32_t)S1r.size();
    uint64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    counters[2] = n_packs;

    // only A, C, G, T, U are defined for the vote (an unordered_map key set in the reference)
    {
        bool ok[256] = {false};
        ok['A'] = ok['C'] = ok['G'] = ok['T'] = ok['U'] = true;
        std::atomic<int> bad(0);
        const size_t chunk = 4096;
        parallel_for((n1 + chunk - 1) / chunk, P->n_threads, [&](size_t c) {
            for (size_t q = c * chunk; q < std::min<size_t>(n1, (c + 1) * chunk); ++q)
                for (uint64_t b = off[S1r[q].rid]; b < off[S1r[q].rid + 1]; ++b)
                    if (!ok[seq[b]]) { bad = 1; return; }
        });
        if (bad) { set_error("correct: read contains a base other than A, C, G, T, U"); return RATTLE_ERR_ARG; }
    }

    std::vector<hread> uncorrected;
    std::vector<int32_t> unc_cid;
    for (size_t i = 0; i < small.size(); ++i) { uncorrected.push_back(oriented_read(seq, qual, off, small[i], 0, 0)); unc_cid.push_back(small_cid[i]); }

    std::vector<uint32_t> olen(n1 + 1, 0), tfront(n1 + 1, 0), tback(n1 + 1, 0);
    std::vector<uint32_t> cons_len2, cons_len3, pk_slot, cl_slot(n_clusters, 0);
    std::vector<uint8_t> cons2, cons3;               // consensi of stage 2b+3a / 3b, concatenated at the stage's coff
    std::vector<std::string> cl_cons(n_clusters);
    stage S2a, S2, S3;
    dbuf<uint8_t> d_os, d_oq;                        // corrected reads, compacted on the device
    std::thread d2h;
    hipError_t d2h_err = hipSuccess;
    struct joiner { std::thread &t; ~joiner() { if (t.joinable()) t.join(); } } d2h_join{d2h};      // also on error returns
    if (n_packs) {
        // ---- reads -> HBM, oriented pack members gathered into stage 1 (:343-346)
        const uint64_t total_in = off[n_reads];
        dbuf<uint8_t> d_rseq, d_rqual;
        const uint8_t *dev_seq, *dev_qual;
        if (ctx->staged_seq_key == seq && ctx->staged_qual_key == qual && ctx->staged_n == n_reads && ctx->staged_total == total_in && off[0] == 0) {
            dev_seq = ctx->d_staged_seq.p; dev_qual = ctx->d_staged_qual.p;       // resident (rattle_hip_stage_reads)
        } else {
            phase_timer T("correct: upload");
            RT_TRY(d_rseq.reserve(total_in + 64)); RT_TRY(d_rqual.reserve(total_in + 64));
            RT_HIP(hipMemcpyAsync(d_rseq.p, seq, total_in, hipMemcpyHostToDevice, st));
            RT_HIP(hipMemcpyAsync(d_rqual.p, qual, total_in, hipMemcpyHostToDevice, st));
            dev_seq = d_rseq.p; dev_qual = d_rqual.p;
        }
        std::vector<gather_desc> desc(n1);
        S1.off.assign(n1 + 1, 0);
        for (uint32_t q = 0; q < n1; ++q) {
            const uint32_t len = (uint32_t)(off[S1r[q].rid + 1] - off[S1r[q].rid]);
            desc[q] = gather_desc{off[S1r[q].rid], S1.off[q], len, S1r[q].rev};
            S1.off[q + 1] = S1.off[q] + len;
        }
        // ---- POA #1 (correct.cpp:398-405) + fix ends + correction (:407-409)
        {
            phase_timer T("correct: stage 1");
            RT_TRY(run_stage(ctx, S1, desc, {gather_part{0, n1, dev_seq, dev_qual}}, 1, P, order, counters));
            RT_HIP(hipMemcpyAsync(olen.data(), S1.olen.p, (size_t)n1 * 4, hipMemcpyDeviceToHost, st));
            RT_HIP(hipMemcpyAsync(tfront.data(), S1.tfront.p, (size_t)n1 * 4, hipMemcpyDeviceToHost, st));
            RT_HIP(hipMemcpyAsync(tback.data(), S1.tback.p, (size_t)n1 * 4, hipMemcpyDeviceToHost, st));
            RT_HIP(hipStreamSynchronize(st));
        }
        d_rseq.release(); d_rqual.release();
        S1.seq.release(); S1.qual.release(); S1.col.release();

        // ---- corrected reads in pack order (:413-425): compacted on the device, one download
        {
            phase_timer T("correct: corrected reads D2H");
            std::vector<gather_desc> od;
            std::vector<uint32_t> oq;
            uint64_t tot = 0;
            for (uint32_t p = 0; p < n_packs; ++p)
                for (uint32_t q = S1.first[p]; q < S1.first[p + 1]; ++q) {
                    if (olen[q] == 0) continue;
                    od.push_back(gather_desc{S1.moff[p] + (uint64_t)(q - S1.first[p]) * S1.width[p], tot, olen[q], 0u});
                    oq.push_back(q);
                    tot += olen[q];
                }
            rattle_read_set &C = R->corrected;
            const size_t nc = od.size();
            C.n = (uint32_t)nc;
            C.read_id = (int32_t *)malloc(std::max<size_t>(1, nc) * 4); C.cluster_id = (int32_t *)malloc(std::max<size_t>(1, nc) * 4);
            C.n_reads = (int32_t *)malloc(std::max<size_t>(1, nc) * 4); C.off = (uint64_t *)malloc((nc + 1) * 8);
            C.seq = (char *)malloc(tot + 1); C.qual = (char *)malloc(tot + 1);
            C.seq[tot] = 0; C.qual[tot] = 0;
            uint32_t p = 0;
            for (size_t i = 0; i < nc; ++i) {
                while (oq[i] >= S1.first[p + 1]) ++p;
                C.read_id[i] = S1r[oq[i]].rid; C.cluster_id[i] = pk_cid[p]; C.n_reads[i] = 0; C.off[i] = od[i].dst;
            }
            C.off[nc] = tot;
            if (nc) {
                dbuf<gather_desc> d_od;
                RT_TRY(d_od.reserve(nc)); RT_TRY(d_os.reserve(tot + 64)); RT_TRY(d_oq.reserve(tot + 64));
                RT_HIP(hipMemcpyAsync(d_od.p, od.data(), nc * sizeof(gather_desc), hipMemcpyHostToDevice, st));
                RT_TRY(launch_gather(ctx, d_od.p, (uint32_t)nc, S1.rowc.p, S1.rowq.p, d_os.p, d_oq.p));
                RT_HIP(hipStreamSynchronize(st));
                d_od.release();
                // the download (2 GB at 1e6 reads, pageable destination) runs on a helper thread and the
                // copy engine while the following POA stages compute
                char *dst_s = C.seq, *dst_q = C.qual;
                const uint8_t *src_s = d_os.p, *src_q = d_oq.p;
                const int dev = ctx->device;
                d2h = std::thread([=, &d2h_err]() {
                    hipError_t e = hipSetDevice(dev);
                    if (e == hipSuccess) e = hipMemcpy(dst_s, src_s, tot, hipMemcpyDeviceToHost);
                    if (e == hipSuccess) e = hipMemcpy(dst_q, src_q, tot, hipMemcpyDeviceToHost);
                    d2h_err = e;
                });
            }
        }
        // reads whose corrected sequence came out empty: uncorrected, as fix_msa_ends left them (:289-293)
        for (uint32_t p = 0; p < n_packs; ++p)
            for (uint32_t q = S1.first[p]; q < S1.first[p + 1]; ++q)
                if (olen[q] == 0) { uncorrected.push_back(oriented_read(seq, qual, off, S1r[q], tfront[q], tback[q])); unc_cid.push_back(pk_cid[p]); }

        // ---- POA #2 over the corrected reads, stably sorted by length desc (:427-445), + consensus vote;
        // per-cluster consensus (:489-556) with POA #3 for clusters of more than one pack.
        // A POA #3 pack is sequential in its number of packs, so the clusters with many packs would
        // leave the device to one workgroup each at the end.  When there are enough of them, their
        // packs go through POA #2 first (stage 2a) and their POA #3 shares a pass with the POA #2 of
        // everything else (stage 2b+3a); the remaining small POA #3 groups follow (stage 3b).
        // (RATTLE_BIG_CLUSTER_PACKS / RATTLE_BIG_MIN_PACKS override the two thresholds: tests force the split on small inputs)
        const uint32_t BIG = std::max(2, getenv("RATTLE_BIG_CLUSTER_PACKS") ? atoi(getenv("RATTLE_BIG_CLUSTER_PACKS")) : 48);
        const uint64_t big_min = getenv("RATTLE_BIG_MIN_PACKS") ? (uint64_t)atoll(getenv("RATTLE_BIG_MIN_PACKS")) : 1024;
        std::vector<uint8_t> big(n_clusters, 0);
        {
            uint64_t big_packs = 0;
            for (uint32_t c = 0; c < n_clusters; ++c) if (cl_np[c] >= BIG) big_packs += cl_np[c];
            if (big_packs >= big_min) for (uint32_t c = 0; c < n_clusters; ++c) big[c] = cl_np[c] >= BIG;
        }
        std::vector<uint32_t> rows;
        auto add_pack2 = [&](stage &S, std::vector<gather_desc> &d, uint32_t p) {       // pack p's corrected reads, length-sorted
            rows.clear();
            for (uint32_t q = S1.first[p]; q < S1.first[p + 1]; ++q) if (olen[q]) rows.push_back(q);
            std::stable_sort(rows.begin(), rows.end(), [&](uint32_t a, uint32_t b) { return olen[a] > olen[b]; });
            for (uint32_t q : rows) {
                d.push_back(gather_desc{S1.moff[p] + (uint64_t)(q - S1.first[p]) * S1.width[p], S.off.back(), olen[q], 0u});
                S.off.push_back(S.off.back() + olen[q]);
            }
            S.first.push_back((uint32_t)S.off.size() - 1);
        };
        pk_slot.assign(n_packs, 0);
        // stage 2a: packs of the big clusters
        std::vector<uint32_t> len2a;
        {
            std::vector<gather_desc> d;
            S2a.first.assign(1, 0); S2a.off.assign(1, 0);
            for (uint32_t p = 0; p < n_packs; ++p) if (big[pk_cid[p]]) { pk_slot[p] = S2a.n_packs(); add_pack2(S2a, d, p); }
            if (S2a.n_packs()) {
                phase_timer T("correct: stage 2a");
                RT_TRY(run_stage(ctx, S2a, d, {gather_part{0, (uint32_t)d.size(), S1.rowc.p, nullptr}}, 2, P, order, counters));
                len2a.assign(S2a.n_packs(), 0);
                RT_HIP(hipMemcpyAsync(len2a.data(), S2a.cons_len.p, (size_t)S2a.n_packs() * 4, hipMemcpyDeviceToHost, st));
                RT_HIP(hipStreamSynchronize(st));
            }
        }
        // stage 2b+3a: POA #3 groups of the big clusters first, then the packs of all other clusters
        {
            phase_timer T("correct: stage 2b+3a");
            std::vector<gather_desc> d;
            S2.first.assign(1, 0); S2.off.assign(1, 0);
            for (uint32_t c = 0; c < n_clusters; ++c) {
                if (!big[c]) continue;
                cl_slot[c] = S2.n_packs();
                for (uint32_t p = cl_p0[c]; p < cl_p0[c] + cl_np[c]; ++p) {
                    d.push_back(gather_desc{S2a.coff[pk_slot[p]], S2.off.back(), len2a[pk_slot[p]], 0u});
                    S2.off.push_back(S2.off.back() + len2a[pk_slot[p]]);
                }
                S2.first.push_back((uint32_t)S2.off.size() - 1);
            }
            const uint32_t n3a = (uint32_t)d.size();
            for (uint32_t p = 0; p < n_packs; ++p) if (!big[pk_cid[p]]) { pk_slot[p] = S2.n_packs(); add_pack2(S2, d, p); }
            RT_TRY(run_stage(ctx, S2, d, {gather_part{0, n3a, S2a.cons_out.p, nullptr}, gather_part{n3a, (uint32_t)d.size() - n3a, S1.rowc.p, nullptr}},
                             2, P, order, counters));
            cons_len2.assign(S2.n_packs() + 1, 0);
            cons2.resize(S2.cols + 1);
            RT_HIP(hipMemcpyAsync(cons_len2.data(), S2.cons_len.p, (size_t)S2.n_packs() * 4, hipMemcpyDeviceToHost, st));
            if (S2.cols) RT_HIP(hipMemcpyAsync(cons2.data(), S2.cons_out.p, S2.cols, hipMemcpyDeviceToHost, st));
            RT_HIP(hipStreamSynchronize(st));
        }
        S1.release(); S2a.release();
        // stage 3b: POA #3 of the other clusters with more than one pack
        {
            std::vector<gather_desc> d;
            S3.first.assign(1, 0); S3.off.assign(1, 0);
            for (uint32_t c = 0; c < n_clusters; ++c) {
                if (cl_np[c] <= 1 || big[c]) continue;
                cl_slot[c] = S3.n_packs();
                for (uint32_t p = cl_p0[c]; p < cl_p0[c] + cl_np[c]; ++p) {
                    d.push_back(gather_desc{S2.coff[pk_slot[p]], S3.off.back(), cons_len2[pk_slot[p]], 0u});
                    S3.off.push_back(S3.off.back() + cons_len2[pk_slot[p]]);
                }
                S3.first.push_back((uint32_t)S3.off.size() - 1);
            }
            if (S3.n_packs()) {
                phase_timer T("correct: stage 3b");
                RT_TRY(run_stage(ctx, S3, d, {gather_part{0, (uint32_t)d.size(), S2.cons_out.p, nullptr}}, 2, P, order, counters));
                cons_len3.assign(S3.n_packs(), 0);
                cons3.resize(S3.cols + 1);
                RT_HIP(hipMemcpyAsync(cons_len3.data(), S3.cons_len.p, (size_t)S3.n_packs() * 4, hipMemcpyDeviceToHost, st));
                if (S3.cols) RT_HIP(hipMemcpyAsync(cons3.data(), S3.cons_out.p, S3.cols, hipMemcpyDeviceToHost, st));
                RT_HIP(hipStreamSynchronize(st));
            }
        }
        S2.release(); S3.release();
        for (uint32_t c = 0; c < n_clusters; ++c) {          // where each cluster's consensus ended up
            if (cl_np[c] == 0) continue;
            if (big[c]) cl_cons[c] = std::string((const char *)cons2.data() + S2.coff[cl_slot[c]], cons_len2[cl_slot[c]]);
            else if (cl_np[c] > 1) cl_cons[c] = std::string((const char *)cons3.data() + S3.coff[cl_slot[c]], cons_len3[cl_slot[c]]);
            else cl_cons[c] = std::string((const char *)cons2.data() + S2.coff[pk_slot[cl_p0[c]]], cons_len2[pk_slot[cl_p0[c]]]);
        }
    } else {
        fill_set(R->corrected, {}, {}, {});
    }
    if (d2h.joinable()) d2h.join();
    d_os.release(); d_oq.release();
    if (d2h_err != hipSuccess) { set_error(std::string("corrected reads download: ") + hipGetErrorString(d2h_err)); return RATTLE_ERR_HIP; }
    std::vector<hread> consensi;
    std::vector<int32_t> con_cid, con_n;
    for (uint32_t c = 0; c < n_clusters; ++c) {
        if (cl_np[c] == 0) continue;
        int total = 0;
        for (uint32_t p = cl_p0[c]; p < cl_p0[c] + cl_np[c]; ++p) total += (int)(S1.first[p + 1] - S1.first[p]);
        const std::string &s = cl_cons[c];
        consensi.push_back(hread{s, std::string(s.size(), 'K'), -1});
        con_cid.push_back((int32_t)c);
        con_n.push_back(total);
    }
    fill_set(R->uncorrected, uncorrected, unc_cid, {});
    fill_set(R->consensi, consensi, con_cid, con_n);
    memcpy(R->counters, counters, sizeof(counters));
    return 0;
}

}  // namespace rattle
