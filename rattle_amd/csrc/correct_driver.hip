// Host orchestration of `rattle correct`, /root/reference/correct.cpp:311-563, over kernels C and D.
//
// The reference runs a queue of packs through worker threads, two POAs per pack
// (correct.cpp:398-405, 428-436) plus one per multi-pack cluster (:520-532).  Packs are
// independent, so here each POA stage is ONE device pass over all packs, and the post-MSA logic
// (fix_msa_ends :32-92, column vote :94-193, per-read correction :196-309) is kernel D
// (post_msa.hip), one workgroup per pack.  Sequences, qualities, MSA columns and row matrices stay
// in HBM from the upload of the reads to the download of the corrected reads; between stages the
// host only sees lengths (pack widths, corrected read lengths, trim counts) and the pack consensi
// (a few MB), from which it plans the next stage (pack order, length-sorted order for POA #2,
// gather descriptors).
//
// One job over several GPUs: the pack list is the same on every rank (plan_packs); packs
// are LPT-assigned to ranks, both POAs of a pack stay on one GPU, pack consensi are all-gathered (the
// only data-path exchange, a few MB), POA #3 groups are LPT-assigned again.  A rank returns its own
// packs' reads and all consensi; correction_gather reassembles the single-GPU result on the root.
//
// Packs that do not fit the device are skipped and reported, never fatal (rattle_skip_list).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "common.h"

namespace rattle {

// ---- pure host planning (also behind rattle_hip_plan_packs / rattle_hip_lpt_assign) ---------------------
void lpt_assign(const std::vector<uint64_t> &cost, int nranks, std::vector<uint32_t> &owner) {
    const size_t n = cost.size();
    owner.assign(n, 0);
    if (nranks <= 1 || n == 0) return;
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    std::vector<uint64_t> load((size_t)nranks, 0);
    for (uint32_t i : order) {
        int best = 0;
        for (int r = 1; r < nranks; ++r) if (load[r] < load[best]) best = r;
        owner[i] = (uint32_t)best;
        load[best] += std::max<uint64_t>(cost[i], 1);
    }
}

int plan_packs(const uint64_t *off, uint32_t n_reads, uint32_t n_clusters, const uint32_t *coff, const int32_t *mid, const uint8_t *mrev,
               const rattle_correct_params *P, int nranks, pack_plan &out) {
    const int split = P->split > 0 ? P->split : 200;
    out = pack_plan();
    out.first.assign(1, 0);
    out.cl_p0.assign(n_clusters, 0); out.cl_np.assign(n_clusters, 0);
    for (uint32_t c = 0; c < n_clusters; ++c) {
        const uint32_t a = coff[c], b = coff[c + 1];
        const int n = (int)(b - a);
        out.cl_p0[c] = (uint32_t)out.pk_cid.size();
        if (n <= 0) continue;
        const int n_files = (n - 1) / split + 1;                                  // correct.cpp:331
        for (int nf = 0; nf < n_files; ++nf) {
            const size_t start = out.members.size();
            uint64_t sum = 0, longest = 0;
            for (int j = nf; j < n; j += n_files) {                               // :335 strided sub-packs
                const int32_t rid = mid[a + j];
                if (rid < 0 || (uint32_t)rid >= n_reads) { set_error("cluster member id out of range"); return RATTLE_ERR_ARG; }
                out.members.push_back(sref{rid, (uint8_t)(mrev[a + j] ? 1 : 0)});
                const uint64_t L = off[rid + 1] - off[rid];
                sum += L; longest = std::max(longest, L);
            }
            const bool queued = (int)(out.members.size() - start) > P->min_reads;  // :360 strict
            const bool too_big = queued && P->max_pack_cells && (6 * longest + 64) * longest > P->max_pack_cells;
            if (queued && !too_big) {
                out.pk_cid.push_back((int32_t)c);
                out.pk_local.push_back(out.cl_np[c]);
                out.pk_cost.push_back(longest * sum);
                out.first.push_back((uint32_t)out.members.size());
                ++out.cl_np[c];
            } else {
                for (size_t t = start; t < out.members.size(); ++t) {
                    out.small.push_back(out.members[t]); out.small_cid.push_back((int32_t)c);
                    out.small_why.push_back(too_big ? 1 : 0); out.small_pack.push_back((uint32_t)nf);
                }
                out.members.resize(start);
            }
        }
    }
    lpt_assign(out.pk_cost, nranks, out.pk_owner);
    return 0;
}

namespace {

struct hread {
    std::string seq, qual;
    int32_t rid;
};

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
    std::vector<uint8_t> skipped;       // [n_packs] the pack did not fit the device
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

// The POA arena is cached in the context between stages and sized by what was free when its pass began (85 % of it).  The buffers the
// stages allocate BESIDE it grow with the job: the MSA rows of a stage (two bytes per cell) and the compacted corrected reads (two bytes per
// base, + 25 % of dbuf's slack).  At 5e6 mixed reads (10 Gb) the corrected reads no longer fitted beside a 230 GB arena.  So before a
// large allocation: if it does not fit into what is free, the idle arena goes (the next POA pass allocates one that fits; ~20 ms per GB).
static void make_room(rattle_ctx *ctx, uint64_t bytes) {
    static const bool always = getenv("RATTLE_MAKE_ROOM_ALWAYS") != nullptr;      // (tests: the arena goes at every such point)
    if (!ctx->poa_arena || (!always && bytes < (4ull << 30))) return;             // (a job of the bench's size never asks the driver anything here)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    const uint64_t need = bytes + bytes / 4 + (1ull << 30);
    if (free_b >= need && !always) return;
    (void)hipDeviceSynchronize();
    (void)hipFree(ctx->poa_arena);
    ctx->poa_arena = nullptr; ctx->poa_arena_bytes = 0;
}

// gather the stage's sequences (descriptors built by the caller, dst offsets = st.off) and run POA + kernel D
int run_stage(rattle_ctx *ctx, stage &S, const std::vector<gather_desc> &desc, const std::vector<gather_part> &parts,
              int mode, const rattle_correct_params *P, const char *order, uint64_t *counters) {
    hipStream_t st = ctx->stream;
    const uint32_t n = S.n(), np = S.n_packs();
    const uint64_t total = S.off[n];
    S.width.assign(np, 0);
    S.skipped.assign(np, 0);
    if (np == 0) return 0;
    dbuf<gather_desc> d_desc;
    make_room(ctx, (uint64_t)(n + 1) * sizeof(gather_desc) + (mode == 1 ? 3 : 2) * (total + 64));
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
        // POA #2 / #3 align corrected reads / pack consensi: their graphs are almost chains -- rows that depend on each other
        // (teams of wavefronts gain nothing), and per alignment as much serial work as DP (what pays is many packs per CU)
        ctx->poa_shallow_graphs = mode == 2 && !getenv("RATTLE_NO_SHALLOW_HINT");      // (the switch: A/B of the hint itself)
        struct unhint { rattle_ctx *c; ~unhint() { c->poa_shallow_graphs = 0; } } uh{ctx};
        RT_TRY(poa_device_run(ctx, S.seq.p, S.d_off.p, S.off.data(), n, S.first.data(), np, S.col.p, S.d_width.p, S.width.data(), h_cnt, &S.skipped));
    }
    d_desc.release();
    counters[0] += h_cnt[0];
    counters[1] += h_cnt[1];
    counters[5] += h_cnt[11]; counters[6] += h_cnt[12]; counters[7] += h_cnt[13];      // DP cells computed; alignments with a certified band / a failed certificate
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
    make_room(ctx, (mode == 1 ? 2 : 1) * (cells + 64) + 4 * (cols + 64) + 20ull * (n + 1));
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

// Consensus stage (POA #2 of packs, POA #3 of clusters) whose input sequences come from the host: the pack
// consensi are a few MB and, with several ranks, arrive through the exchange.
struct cons_stage {
    stage S;
    std::vector<uint8_t> h_in;                 // concatenated input sequences (host-sourced groups)
    std::vector<uint32_t> len;                 // [n_packs] consensus length
    std::vector<uint8_t> cons;                 // consensi at S.coff
};

int fetch_consensi(rattle_ctx *ctx, cons_stage &C) {
    const uint32_t np = C.S.n_packs();
    C.len.assign(np + 1, 0);
    C.cons.assign(C.S.cols + 1, 0);
    if (!np) return 0;
    RT_HIP(hipMemcpyAsync(C.len.data(), C.S.cons_len.p, (size_t)np * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (C.S.cols) RT_HIP(hipMemcpyAsync(C.cons.data(), C.S.cons_out.p, C.S.cols, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// byte-string records of the exchange: [u32 id][u32 flag][u32 len][bytes]
void put_rec(std::vector<uint8_t> &b, uint32_t id, uint32_t flag, const char *s, uint32_t len) {
    const size_t at = b.size();
    b.resize(at + 12 + len);
    memcpy(b.data() + at, &id, 4); memcpy(b.data() + at + 4, &flag, 4); memcpy(b.data() + at + 8, &len, 4);
    if (len) memcpy(b.data() + at + 12, s, len);
}

}  // namespace

int correct_driver(rattle_ctx *ctx, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint32_t n_reads,
                   uint32_t n_clusters, const uint32_t *coff, const int32_t *mid, const uint8_t *mrev,
                   const rattle_correct_params *P, rattle_correction **out) {
    char order[8] = {0};
    memcpy(order, P->vote_order[0] ? P->vote_order : "U-GTCA", 6);
    for (int i = 0; i < 6; ++i)
        if (!order[i] || !strchr("ACGTU-", order[i])) { set_error("vote_order must be a permutation of ACGTU-"); return RATTLE_ERR_ARG; }
    rattle_correction *R = (rattle_correction *)calloc(1, sizeof(rattle_correction));
    *out = R;
    hipStream_t st = ctx->stream;
    phase_timer T_all("correct: total");
    const int rank = ctx->xchg.rank, nranks = ctx->xchg.nranks;

    // ---- correct.cpp:328-370 pack building: ids and strands only, the bases stay where they are
    pack_plan PL;
    RT_TRY(plan_packs(off, n_reads, n_clusters, coff, mid, mrev, P, nranks, PL));
    const uint32_t n_packs = (uint32_t)PL.pk_cid.size();
    uint64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    counters[2] = n_packs;

    // the order a cluster's pack consensi enter POA #3 in (default: pack order)
    std::vector<std::vector<uint32_t>> cl_perm(P->n_pack_orders ? n_clusters : 0);
    for (uint32_t i = 0; i < P->n_pack_orders; ++i) {
        if (!P->pack_order_cluster || !P->pack_order_offsets || !P->pack_order_perm) { set_error("pack order arrays missing"); return RATTLE_ERR_ARG; }
        const uint32_t c = P->pack_order_cluster[i];
        if (c >= n_clusters) { set_error("pack order: cluster out of range"); return RATTLE_ERR_ARG; }
        const uint32_t a = P->pack_order_offsets[i], b = P->pack_order_offsets[i + 1];
        std::vector<uint32_t> perm(P->pack_order_perm + a, P->pack_order_perm + b), chk(perm);
        std::sort(chk.begin(), chk.end());
        bool ok = b - a == PL.cl_np[c];
        for (uint32_t t = 0; ok && t < chk.size(); ++t) ok = chk[t] == t;
        if (!ok) { set_error("pack order of cluster " + std::to_string(c) + " is not a permutation of its " + std::to_string(PL.cl_np[c]) + " packs"); return RATTLE_ERR_ARG; }
        cl_perm[c] = perm;
    }

    // my packs, in pack order (all of them on one rank)
    std::vector<uint32_t> mine;
    for (uint32_t p = 0; p < n_packs; ++p) if ((int)PL.pk_owner[p] == rank) mine.push_back(p);
    const uint32_t nm = (uint32_t)mine.size();

    // ---- POA #2 over the corrected reads of a pack, stably sorted by length desc (:427-445), + consensus vote;
    // per-cluster consensus (:489-556) with POA #3 for clusters of more than one pack.
    // A POA #3 group is sequential in its number of packs, so the clusters with many packs would leave the
    // device to one workgroup each at the end.  When there are enough of them, their packs go through
    // POA #2 first (stage 2a) and their POA #3 (stage 3a) runs beside other work; the remaining small POA #3 groups
    // follow (stage 3b).  With several ranks each stage covers this rank's packs / groups and ends with an all-gather of
    // its consensi.
    // (RATTLE_BIG_CLUSTER_PACKS / RATTLE_BIG_MIN_PACKS override the two thresholds: tests force the split on small inputs)
    const uint32_t BIG = std::max(2, getenv("RATTLE_BIG_CLUSTER_PACKS") ? atoi(getenv("RATTLE_BIG_CLUSTER_PACKS")) : 48);
    const uint64_t big_min = getenv("RATTLE_BIG_MIN_PACKS") ? (uint64_t)atoll(getenv("RATTLE_BIG_MIN_PACKS")) : 1024;
    std::vector<uint8_t> big(n_clusters, 0);
    bool any_big = false;
    {
        uint64_t big_packs = 0;
        for (uint32_t c = 0; c < n_clusters; ++c) if (PL.cl_np[c] >= BIG) big_packs += PL.cl_np[c];
        if (big_packs >= big_min) for (uint32_t c = 0; c < n_clusters; ++c) { big[c] = PL.cl_np[c] >= BIG; any_big |= big[c] != 0; }
    }
    // (Round 4 also ran the big clusters' chain as a second flow on a helper context beside the POA #1 of everything else: byte-
    // identical, but slower -- two stage-1 passes have two tails -- and removed in round 5.  What shortens the chain instead is the
    // row loop of a lone workgroup: poa.hip, dp_rows_mt.)
    struct s1group {
        stage S;
        std::vector<sref> r;                         // the group's pack members, pack after pack
        std::vector<uint32_t> ks;                    // index in `mine` of each of its packs
        std::vector<uint32_t> olen, tfront, tback;   // per member
    } G[1];
    std::vector<uint8_t> g_of(nm, 0);
    std::vector<uint32_t> k_in(nm, 0);
    G[0].S.first.assign(1, 0);
    for (uint32_t k = 0; k < nm; ++k) {
        const uint32_t p = mine[k];
        const int g = 0;
        g_of[k] = (uint8_t)g; k_in[k] = (uint32_t)G[g].ks.size();
        G[g].ks.push_back(k);
        for (uint32_t q = PL.first[p]; q < PL.first[p + 1]; ++q) G[g].r.push_back(PL.members[q]);
        G[g].S.first.push_back((uint32_t)G[g].r.size());
    }

    // Several ranks: everything up to here depends on the arguments alone (the same on every rank).  From here on a rank
    // works on its own packs, and a failure of its own (a bad base in one of ITS reads, a HIP error, an allocation) must not
    // leave the other ranks blocked in the next all-gather: the rank remembers the error, keeps joining the exchanges with a
    // failure record, and every rank returns an error after the exchange that carried it.
    int local_rc = 0;
    std::string local_msg;
    auto local_step = [&](int r) -> bool {           // true: carry on with this rank's own work
        if (r != 0 && local_rc == 0) { local_rc = r; local_msg = rattle_hip_last_error(); }
        return local_rc == 0;
    };
#define LOCAL_TRY(call) do { if (local_rc == 0) { const int r_ = (call); if (r_ != 0) { if (nranks == 1) return r_; local_step(r_); } } } while (0)

    LOCAL_TRY(ensure_post_constants(ctx));
    // only A, C, G, T, U are defined for the vote (an unordered_map key set in the reference)
    {
        bool ok[256] = {false};
        ok['A'] = ok['C'] = ok['G'] = ok['T'] = ok['U'] = true;
        std::atomic<int> bad(0);
        const size_t chunk = 4096;
        for (int g = 0; g < 1; ++g) {
            const std::vector<sref> &R1 = G[g].r;
            const size_t n1g = R1.size();
            parallel_for((n1g + chunk - 1) / chunk, P->n_threads, [&](size_t c) {
                for (size_t q = c * chunk; q < std::min<size_t>(n1g, (c + 1) * chunk); ++q)
                    for (uint64_t b = off[R1[q].rid]; b < off[R1[q].rid + 1]; ++b)
                        if (!ok[seq[b]]) { bad = 1; return; }
            });
        }
        if (bad) {
            set_error("correct: read contains a base other than A, C, G, T, U");
            if (nranks == 1) return RATTLE_ERR_ARG;
            local_step(RATTLE_ERR_ARG);
        }
    }

    // skip list (this rank's share; unqueued entries on rank 0)
    struct skip_t { int32_t cid; uint32_t pack, stage; std::vector<int32_t> rids; };
    std::vector<skip_t> skips;

    std::vector<hread> uncorrected;
    std::vector<int32_t> unc_cid;
    std::vector<uint32_t> unc_pack;
    if (rank == 0) {
        for (size_t i = 0; i < PL.small.size(); ++i) {
            uncorrected.push_back(oriented_read(seq, qual, off, PL.small[i], 0, 0));
            unc_cid.push_back(PL.small_cid[i]); unc_pack.push_back(0xFFFFFFFFu);
            if (PL.small_why[i]) {
                if (skips.empty() || skips.back().stage != 0 || skips.back().cid != PL.small_cid[i] || skips.back().pack != PL.small_pack[i])
                    skips.push_back(skip_t{PL.small_cid[i], PL.small_pack[i], 0u, {}});
                skips.back().rids.push_back(PL.small[i].rid);
            }
        }
    }

    std::vector<uint8_t> pk_dead(n_packs, 0);       // stage at which a pack was given up (this rank's packs: exact; others: from the exchange)
    std::vector<std::string> pk_cons(n_packs);       // pack consensus (POA #2), filled for every pack by the exchanges
    std::vector<uint8_t> pk_has(n_packs, 0);
    std::vector<std::string> cl_cons(n_clusters);
    std::vector<uint8_t> cl_has(n_clusters, 0);
    dbuf<uint8_t> d_os, d_oq;                        // corrected reads, compacted on the device
    std::thread d2h;
    hipError_t d2h_err = hipSuccess;
    struct joiner { std::thread &t; ~joiner() { if (t.joinable()) t.join(); } } d2h_join{d2h};      // also on error returns
    std::vector<uint32_t> cor_pack;

    // ---- reads -> HBM once (both groups gather from them)
    dbuf<uint8_t> d_rseq, d_rqual;
    const uint8_t *dev_seq = nullptr, *dev_qual = nullptr;
    auto upload_reads = [&]() -> int {
        const uint64_t total_in = off[n_reads];
        if (ctx->staged_seq_key == seq && ctx->staged_qual_key == qual && ctx->staged_n == n_reads && ctx->staged_total == total_in && off[0] == 0) {
            dev_seq = ctx->d_staged_seq.p; dev_qual = ctx->d_staged_qual.p;       // resident (rattle_hip_stage_reads)
        } else {
            phase_timer T("correct: upload");
            RT_TRY(d_rseq.reserve(total_in + 64)); RT_TRY(d_rqual.reserve(total_in + 64));
            RT_HIP(hipMemcpyAsync(d_rseq.p, seq, total_in, hipMemcpyHostToDevice, st));
            RT_HIP(hipMemcpyAsync(d_rqual.p, qual, total_in, hipMemcpyHostToDevice, st));
            dev_seq = d_rseq.p; dev_qual = d_rqual.p;
        }
        return 0;
    };
    // ---- stage 1 of one group: oriented pack members gathered (:343-346), POA #1 (correct.cpp:398-405) + fix ends +
    // correction (:407-409); lengths back to the host
    auto stage1_group = [&](int g, uint64_t *cnt) -> int {
        s1group &X = G[g];
        const uint32_t n1 = (uint32_t)X.r.size();
        X.olen.assign(n1 + 1, 0); X.tfront.assign(n1 + 1, 0); X.tback.assign(n1 + 1, 0);
        if (X.ks.empty()) return 0;
        std::vector<gather_desc> desc(n1);
        X.S.off.assign(n1 + 1, 0);
        for (uint32_t q = 0; q < n1; ++q) {
            const uint32_t len = (uint32_t)(off[X.r[q].rid + 1] - off[X.r[q].rid]);
            desc[q] = gather_desc{off[X.r[q].rid], X.S.off[q], len, X.r[q].rev};
            X.S.off[q + 1] = X.S.off[q] + len;
        }
        phase_timer T("correct: stage 1", &ctx->stage_ms[1]);
        RT_TRY(run_stage(ctx, X.S, desc, {gather_part{0, n1, dev_seq, dev_qual}}, 1, P, order, cnt));
        RT_HIP(hipMemcpyAsync(X.olen.data(), X.S.olen.p, (size_t)n1 * 4, hipMemcpyDeviceToHost, st));
        RT_HIP(hipMemcpyAsync(X.tfront.data(), X.S.tfront.p, (size_t)n1 * 4, hipMemcpyDeviceToHost, st));
        RT_HIP(hipMemcpyAsync(X.tback.data(), X.S.tback.p, (size_t)n1 * 4, hipMemcpyDeviceToHost, st));
        RT_HIP(hipStreamSynchronize(st));
        X.S.seq.release(); X.S.qual.release(); X.S.col.release();
        for (uint32_t kk = 0; kk < X.ks.size(); ++kk)
            if (X.S.skipped[kk]) {                    // POA #1 did not fit: the pack's reads stay as they are
                pk_dead[mine[X.ks[kk]]] = 1;
                for (uint32_t q = X.S.first[kk]; q < X.S.first[kk + 1]; ++q) { X.olen[q] = 0; X.tfront[q] = 0; X.tback[q] = 0; }
            }
        return 0;
    };
    // ---- after both groups: skip entries, corrected reads in pack order (:413-425) compacted on the device and downloaded
    // behind the later stages, reads whose corrected sequence came out empty
    auto stage1_finish = [&]() -> int {
        for (uint32_t k = 0; k < nm; ++k) {
            const s1group &X = G[g_of[k]];
            const uint32_t kk = k_in[k];
            if (!X.S.skipped[kk]) continue;
            skips.push_back(skip_t{PL.pk_cid[mine[k]], PL.pk_local[mine[k]], 1u, {}});
            for (uint32_t q = X.S.first[kk]; q < X.S.first[kk + 1]; ++q) skips.back().rids.push_back(X.r[q].rid);
        }
        {
            phase_timer T("correct: corrected reads D2H");
            std::vector<gather_desc> od[1];
            std::vector<int32_t> o_rid, o_cid;
            uint64_t tot = 0;
            for (uint32_t k = 0; k < nm; ++k) {
                const int g = g_of[k];
                const s1group &X = G[g];
                const uint32_t kk = k_in[k];
                for (uint32_t q = X.S.first[kk]; q < X.S.first[kk + 1]; ++q) {
                    if (X.olen[q] == 0) continue;
                    od[g].push_back(gather_desc{X.S.moff[kk] + (uint64_t)(q - X.S.first[kk]) * X.S.width[kk], tot, X.olen[q], 0u});
                    o_rid.push_back(X.r[q].rid); o_cid.push_back(PL.pk_cid[mine[k]]);
                    cor_pack.push_back(mine[k]);
                    tot += X.olen[q];
                }
            }
            rattle_read_set &C = R->corrected;
            const size_t nc = o_rid.size();
            C.n = (uint32_t)nc;
            C.read_id = (int32_t *)malloc(std::max<size_t>(1, nc) * 4); C.cluster_id = (int32_t *)malloc(std::max<size_t>(1, nc) * 4);
            C.n_reads = (int32_t *)malloc(std::max<size_t>(1, nc) * 4); C.off = (uint64_t *)malloc((nc + 1) * 8);
            C.seq = (char *)malloc(tot + 1); C.qual = (char *)malloc(tot + 1);
            C.seq[tot] = 0; C.qual[tot] = 0;
            {
                for (size_t i = 0; i < nc; ++i) { C.off[i] = od[0][i].dst; C.read_id[i] = o_rid[i]; C.cluster_id[i] = o_cid[i]; C.n_reads[i] = 0; }
            }
            C.off[nc] = tot;
            if (nc) {
                make_room(ctx, 2 * (tot + 64) + od[0].size() * sizeof(gather_desc));
                RT_TRY(d_os.reserve(tot + 64)); RT_TRY(d_oq.reserve(tot + 64));
                for (int g = 0; g < 1; ++g) {
                    if (od[g].empty()) continue;
                    dbuf<gather_desc> d_od;
                    RT_TRY(d_od.reserve(od[g].size()));
                    RT_HIP(hipMemcpyAsync(d_od.p, od[g].data(), od[g].size() * sizeof(gather_desc), hipMemcpyHostToDevice, st));
                    RT_TRY(launch_gather(ctx, d_od.p, (uint32_t)od[g].size(), G[g].S.rowc.p, G[g].S.rowq.p, d_os.p, d_oq.p));
                    RT_HIP(hipStreamSynchronize(st));
                    d_od.release();
                }
                // the download (2 GB at 1e6 reads, pageable destination) runs on a helper thread and the
                // copy engine while the following POA stages compute
                char *dst_s = C.seq, *dst_q = C.qual;
                const uint8_t *src_s = d_os.p, *src_q = d_oq.p;
                const int dev = ctx->device;
                const rattle_read_set *ready_set = &R->corrected;
                const uint32_t *ready_pack = cor_pack.data();
                auto ready_fn = P->corrected_ready;
                void *ready_user = P->corrected_ready_user;
                d2h = std::thread([=, &d2h_err]() {
                    hipError_t e = hipSetDevice(dev);
                    if (e == hipSuccess) e = hipMemcpy(dst_s, src_s, tot, hipMemcpyDeviceToHost);
                    if (e == hipSuccess) e = hipMemcpy(dst_q, src_q, tot, hipMemcpyDeviceToHost);
                    d2h_err = e;
                    // the corrected reads are final from here on: the caller may start writing them out while POA #2 / #3 run
                    if (e == hipSuccess && ready_fn) ready_fn(ready_user, ready_set, ready_pack);
                });
            }
        }
        // reads whose corrected sequence came out empty: uncorrected, as fix_msa_ends left them (:289-293)
        for (uint32_t k = 0; k < nm; ++k) {
            const s1group &X = G[g_of[k]];
            const uint32_t kk = k_in[k];
            for (uint32_t q = X.S.first[kk]; q < X.S.first[kk + 1]; ++q)
                if (X.olen[q] == 0) {
                    uncorrected.push_back(oriented_read(seq, qual, off, X.r[q], X.tfront[q], X.tback[q]));
                    unc_cid.push_back(PL.pk_cid[mine[k]]); unc_pack.push_back(mine[k]);
                }
        }
        return 0;
    };

    std::vector<uint32_t> slot_of(n_packs, 0xFFFFFFFFu);   // my pack -> index in `mine`
    for (uint32_t k = 0; k < nm; ++k) slot_of[mine[k]] = k;

    // the live packs of a cluster in the order their consensi enter POA #3
    auto group_of = [&](uint32_t c, std::vector<uint32_t> &g) {
        g.clear();
        for (uint32_t t = 0; t < PL.cl_np[c]; ++t) {
            const uint32_t p = PL.cl_p0[c] + (cl_perm.empty() || cl_perm[c].empty() ? t : cl_perm[c][t]);
            if (pk_has[p]) g.push_back(p);
        }
    };
    // exchange the results of one stage: pack consensi (kind 0) and cluster consensi (kind 1), dead flags included
    // (my_rc / my_msg: this rank's own failure so far -- the main flow passes local_rc, the side flow its own)
    auto exchange_stage = [&](std::vector<uint8_t> &mine_bytes, const int my_rc, const std::string &my_msg) -> int {
        std::vector<std::vector<uint8_t>> all;
        if (nranks > 1) {
            if (my_rc) { mine_bytes.clear(); put_rec(mine_bytes, 0xFFFFFFFFu, 0xFFFFFFFFu, nullptr, 0); }      // failure record
            RT_TRY(xchg_allgatherv(ctx, mine_bytes, all));
            if (my_rc) { set_error(my_msg); return my_rc; }
            for (int r = 0; r < nranks; ++r) {
                uint32_t id = 0, flag = 0;
                if (all[r].size() >= 12) { memcpy(&id, all[r].data(), 4); memcpy(&flag, all[r].data() + 4, 4); }
                if (id == 0xFFFFFFFFu && flag == 0xFFFFFFFFu) { set_error("correct_reads failed on rank " + std::to_string(r)); return RATTLE_ERR_HIP; }
            }
        } else if (xchg_recording(ctx)) RT_TRY(xchg_allgatherv(ctx, mine_bytes, all));      // (measurement aid, common.h)
        else { all.resize(1); all[0].swap(mine_bytes); }
        for (const std::vector<uint8_t> &b : all) {
            size_t at = 0;
            while (at + 12 <= b.size()) {
                uint32_t id, flag, len;
                memcpy(&id, b.data() + at, 4); memcpy(&flag, b.data() + at + 4, 4); memcpy(&len, b.data() + at + 8, 4);
                const char *s = (const char *)b.data() + at + 12;
                at += 12 + len;
                const uint32_t kind = flag & 1u, dead = flag >> 1;
                if (kind == 0) { if (dead) pk_dead[id] = (uint8_t)dead; else { pk_cons[id].assign(s, len); pk_has[id] = 1; } }
                else if (!dead) { cl_cons[id].assign(s, len); cl_has[id] = 1; }
            }
        }
        return 0;
    };
    // POA #2 over a list of my packs (indices in `mine`, all of ONE stage-1 group) + POA #3 over a list of clusters (groups of pack
    // consensi from the host) in one device pass on context `cx`; results into `bytes`, skipped packs into `sk`, work counters into `cnt`
    auto cons_pass = [&](rattle_ctx *cx, const char *name, const std::vector<uint32_t> &slots2, const std::vector<uint32_t> &clusters3,
                         std::vector<uint8_t> &bytes, std::vector<skip_t> &sk, uint64_t *cnt) -> int {
        if (slots2.empty() && clusters3.empty()) return 0;
        phase_timer T(name, !strcmp(name, "correct: stage 2a") ? &ctx->stage_ms[2] : !strcmp(name, "correct: stage 2b+3a") ? &ctx->stage_ms[3] : !strcmp(name, "correct: stage 3b") ? &ctx->stage_ms[4] : nullptr);
        cons_stage C;
        stage &S = C.S;
        std::vector<gather_desc> d;
        S.first.assign(1, 0); S.off.assign(1, 0);
        std::vector<uint32_t> g, rows;
        for (uint32_t c : clusters3) {
            group_of(c, g);
            for (uint32_t p : g) {
                d.push_back(gather_desc{(uint64_t)C.h_in.size(), S.off.back(), (uint32_t)pk_cons[p].size(), 0u});
                C.h_in.insert(C.h_in.end(), pk_cons[p].begin(), pk_cons[p].end());
                S.off.push_back(S.off.back() + pk_cons[p].size());
            }
            S.first.push_back((uint32_t)S.off.size() - 1);
        }
        const uint32_t n3 = (uint32_t)d.size();
        const s1group *X = slots2.empty() ? nullptr : &G[g_of[slots2[0]]];
        for (uint32_t k : slots2) {                  // my pack k's corrected reads, length-sorted
            const uint32_t kk = k_in[k];
            rows.clear();
            for (uint32_t q = X->S.first[kk]; q < X->S.first[kk + 1]; ++q) if (X->olen[q]) rows.push_back(q);
            std::stable_sort(rows.begin(), rows.end(), [&](uint32_t a, uint32_t b) { return X->olen[a] > X->olen[b]; });
            for (uint32_t q : rows) {
                d.push_back(gather_desc{X->S.moff[kk] + (uint64_t)(q - X->S.first[kk]) * X->S.width[kk], S.off.back(), X->olen[q], 0u});
                S.off.push_back(S.off.back() + X->olen[q]);
            }
            S.first.push_back((uint32_t)S.off.size() - 1);
        }
        dbuf<uint8_t> d_in;
        make_room(cx, C.h_in.size() + 64);
        RT_TRY(d_in.reserve(C.h_in.size() + 64));
        if (!C.h_in.empty()) RT_HIP(hipMemcpyAsync(d_in.p, C.h_in.data(), C.h_in.size(), hipMemcpyHostToDevice, cx->stream));
        RT_TRY(run_stage(cx, S, d, {gather_part{0, n3, d_in.p, nullptr}, gather_part{n3, (uint32_t)d.size() - n3, X ? X->S.rowc.p : nullptr, nullptr}}, 2, P, order, cnt));
        RT_TRY(fetch_consensi(cx, C));
        uint32_t slot = 0;
        for (uint32_t c : clusters3) {
            if (S.skipped[slot]) {
                sk.push_back(skip_t{(int32_t)c, 0u, 3u, {}});
                put_rec(bytes, c, 1u | (3u << 1), nullptr, 0);
            } else put_rec(bytes, c, 1u, (const char *)C.cons.data() + S.coff[slot], C.len[slot]);
            ++slot;
        }
        for (uint32_t k : slots2) {
            const uint32_t p = mine[k], kk = k_in[k];
            if (S.skipped[slot]) {
                sk.push_back(skip_t{PL.pk_cid[p], PL.pk_local[p], 2u, {}});
                for (uint32_t q = X->S.first[kk]; q < X->S.first[kk + 1]; ++q) sk.back().rids.push_back(X->r[q].rid);
                put_rec(bytes, p, 2u << 1, nullptr, 0);
            } else put_rec(bytes, p, 0u, (const char *)C.cons.data() + S.coff[slot], C.len[slot]);
            ++slot;
        }
        return 0;
    };
    // the POA #3 groups of the big clusters that fall to this rank (LPT over ranks), once their pack consensi are in
    auto big_groups = [&](std::vector<uint32_t> &g3a) {
        std::vector<uint32_t> g3a_all, own, g;
        std::vector<uint64_t> cost;
        for (uint32_t c = 0; c < n_clusters; ++c) {
            if (!big[c]) continue;
            group_of(c, g);
            if (g.size() > 1) { g3a_all.push_back(c); cost.push_back((uint64_t)g.size() * g.size() * pk_cons[g[0]].size()); }
            else if (g.size() == 1) { cl_cons[c] = pk_cons[g[0]]; cl_has[c] = 1; }
        }
        lpt_assign(cost, nranks, own);
        for (size_t i = 0; i < g3a_all.size(); ++i) if ((int)own[i] == rank) g3a.push_back(g3a_all[i]);
    };

    uint64_t cnt_main[8] = {0};
    std::vector<skip_t> sk_2a, sk_3a, sk_2b, sk_3b;
    std::vector<uint8_t> bytes_3b;
    if (nm) LOCAL_TRY(upload_reads());
    if (nm) LOCAL_TRY(stage1_group(0, cnt_main));
    if (nm && local_rc == 0) LOCAL_TRY(stage1_finish());
    d_rseq.release(); d_rqual.release();
    std::vector<uint8_t> bytes;
    // packs given up in stage 1 are announced with the first exchange
    for (uint32_t k = 0; k < nm; ++k) if (pk_dead[mine[k]] == 1) put_rec(bytes, mine[k], 1u << 1, nullptr, 0);
    // stage 2a: my packs of the big clusters
    std::vector<uint32_t> s2a, s2b;
    for (uint32_t k = 0; k < nm; ++k) if (!pk_dead[mine[k]]) (big[PL.pk_cid[mine[k]]] ? s2a : s2b).push_back(k);
    LOCAL_TRY(cons_pass(ctx, "correct: stage 2a", s2a, {}, bytes, sk_2a, cnt_main));
    if (any_big) { RT_TRY(exchange_stage(bytes, local_rc, local_msg)); bytes.clear(); }
    // stage 2b+3a: POA #3 groups of the big clusters (LPT over ranks), then my packs of all other clusters
    std::vector<uint32_t> g3a;
    big_groups(g3a);
    LOCAL_TRY(cons_pass(ctx, "correct: stage 2b+3a", s2b, g3a, bytes, sk_3a, cnt_main));
    RT_TRY(exchange_stage(bytes, local_rc, local_msg));
    if (!nm || local_rc) {
        if (d2h.joinable()) d2h.join();
        if (R->corrected.off) { rattle_read_set &C = R->corrected; free(C.read_id); free(C.cluster_id); free(C.n_reads); free(C.off); free(C.seq); free(C.qual); C = rattle_read_set(); }
        fill_set(R->corrected, {}, {}, {});
        cor_pack.clear();
    }
    {
        if (d2h.joinable()) d2h.join();                  // the stage-1 rows are no longer needed once the download is done
        if (d2h_err != hipSuccess) {
            set_error(std::string("corrected reads download: ") + hipGetErrorString(d2h_err));
            if (nranks == 1) return RATTLE_ERR_HIP;
            local_step(RATTLE_ERR_HIP);
        }
        G[0].S.release();
        // stage 3b: POA #3 of the other clusters with more than one live pack
        std::vector<uint32_t> g3b_all, g3b, own, g;
        std::vector<uint64_t> cost;
        for (uint32_t c = 0; c < n_clusters; ++c) {
            if (big[c] || PL.cl_np[c] == 0) continue;
            group_of(c, g);
            if (g.size() > 1) { g3b_all.push_back(c); cost.push_back((uint64_t)g.size() * g.size() * pk_cons[g[0]].size()); }
            else if (g.size() == 1) { cl_cons[c] = pk_cons[g[0]]; cl_has[c] = 1; }
        }
        lpt_assign(cost, nranks, own);
        for (size_t i = 0; i < g3b_all.size(); ++i) if ((int)own[i] == rank) g3b.push_back(g3b_all[i]);
        LOCAL_TRY(cons_pass(ctx, "correct: stage 3b", {}, g3b, bytes_3b, sk_3b, cnt_main));
        // several ranks: this exchange always takes place, so that every rank leaves with the same verdict (the caller's
        // next collective is the gather of the corrected reads)
        if (!g3b_all.empty() || nranks > 1 || xchg_recording(ctx)) RT_TRY(exchange_stage(bytes_3b, local_rc, local_msg));
    }
    // skipped packs of the consensus stages in the order the sequential flow meets them: 2a, then 3a before 2b (one pass), then 3b
    for (std::vector<skip_t> *v : {&sk_2a, &sk_3a, &sk_2b, &sk_3b}) for (skip_t &x : *v) skips.push_back(std::move(x));
    counters[0] += cnt_main[0];
    counters[1] += cnt_main[1];
    counters[5] += cnt_main[5]; counters[6] += cnt_main[6]; counters[7] += cnt_main[7];
    if (d2h.joinable()) d2h.join();
    d_os.release(); d_oq.release();
#undef LOCAL_TRY

    std::vector<hread> consensi;
    std::vector<int32_t> con_cid, con_n;
    for (uint32_t c = 0; c < n_clusters; ++c) {
        if (!cl_has[c]) continue;
        int total = 0;
        for (uint32_t p = PL.cl_p0[c]; p < PL.cl_p0[c] + PL.cl_np[c]; ++p) if (pk_has[p]) total += (int)(PL.first[p + 1] - PL.first[p]);
        const std::string &s = cl_cons[c];
        consensi.push_back(hread{s, std::string(s.size(), 'K'), -1});
        con_cid.push_back((int32_t)c);
        con_n.push_back(total);
    }
    fill_set(R->uncorrected, uncorrected, unc_cid, {});
    fill_set(R->consensi, consensi, con_cid, con_n);
    R->corrected_pack = (uint32_t *)malloc(std::max<size_t>(1, cor_pack.size()) * 4);
    if (!cor_pack.empty()) memcpy(R->corrected_pack, cor_pack.data(), cor_pack.size() * 4);
    R->uncorrected_pack = (uint32_t *)malloc(std::max<size_t>(1, unc_pack.size()) * 4);
    if (!unc_pack.empty()) memcpy(R->uncorrected_pack, unc_pack.data(), unc_pack.size() * 4);
    {
        rattle_skip_list &K = R->skipped;
        const size_t n = skips.size();
        K.n = (uint32_t)n;
        K.cluster_id = (int32_t *)malloc(std::max<size_t>(1, n) * 4); K.pack = (uint32_t *)malloc(std::max<size_t>(1, n) * 4);
        K.stage = (uint32_t *)malloc(std::max<size_t>(1, n) * 4); K.read_off = (uint64_t *)malloc((n + 1) * 8);
        uint64_t tot = 0;
        for (size_t i = 0; i < n; ++i) { K.cluster_id[i] = skips[i].cid; K.pack[i] = skips[i].pack; K.stage[i] = skips[i].stage; K.read_off[i] = tot; tot += skips[i].rids.size(); }
        K.read_off[n] = tot;
        K.read_id = (int32_t *)malloc(std::max<uint64_t>(1, tot) * 4);
        for (size_t i = 0; i < n; ++i) if (!skips[i].rids.empty()) memcpy(K.read_id + K.read_off[i], skips[i].rids.data(), skips[i].rids.size() * 4);
        counters[3] = n;
        counters[4] = tot;
    }
    memcpy(R->counters, counters, sizeof(counters));
    return 0;
}

}  // namespace rattle
