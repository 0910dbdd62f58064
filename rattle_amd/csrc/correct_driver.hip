// Host orchestration of `rattle correct`, /root/reference/correct.cpp:311-563, over kernel C.
//
// The reference runs a queue of packs through worker threads, two POAs per pack
// (correct.cpp:398-405, 428-436) plus one per multi-pack cluster (:520-532).  Packs are
// independent, so here each POA stage is ONE device launch over all packs, and the cheap
// post-MSA logic (fix_msa_ends :32-92, column vote :94-193, per-read correction :196-309) runs
// between launches on host threads, one pack per task, in the reference's operation order.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace rattle {

int poa_msa_run(rattle_ctx *ctx, const uint8_t *seq, const uint64_t *off, uint32_t n_seqs, const uint32_t *pack_first,
                uint32_t n_packs, rattle_msa_set **out);

namespace {

struct hread {
    std::string seq, qual;
    int32_t rid;
};

struct pack_t {
    int32_t cid;
    std::vector<hread> reads;        // pack members (mutated by fix_msa_ends)
    std::vector<hread> corrected;    // after correction, then length-sorted for POA #2
    std::vector<hread> dropped;      // reads whose corrected sequence came out empty
    std::string consensus;
};

struct vote_t {
    char order[6];
    int slot[256];
    double perr[256];                // phred_err per quality byte, utils.cpp:10-13
    void init(const char *o) {
        memcpy(order, o, 6);
        for (int i = 0; i < 256; ++i) slot[i] = -1;
        for (int i = 0; i < 6; ++i) slot[(unsigned char)order[i]] = i;
        for (int c = 0; c < 256; ++c) { double q = (char)c - 33; perr[c] = pow(10.0, -q / 10.0); }
    }
};

inline char phred_symbol(double p) { return (char)(-10 * log10(p) + 33); }     // utils.cpp:6-8

inline char comp_base(char c) {                                                  // utils.hpp:8-14
    switch (c) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'T': return 'A';
        case 'G': return 'C';
        case 'U': return 'A';
    }
    return c;       // undefined in the reference (end() dereference); clustered reads never contain it
}

// correct.cpp:32-92.  Two phases per row: trim small leading blocks followed by a long gap,
// then reverse and do the same from the other end; `seq`/`qual` follow (plain reversal).
void fix_msa_ends(std::vector<hread> &reads, std::vector<std::string> &aln) {
    for (size_t i = 0; i < aln.size(); ++i) {
        std::string &row = aln[i];
        const size_t n = row.size();
        for (int phase = 0; phase < 2; ++phase) {
            size_t pos = 0;
            bool stopped = false;
            while (pos < n) {
                while (pos < n && row[pos] == '-') ++pos;
                size_t end = pos;
                int gaps = 0, sz = 0;
                while (gaps < 4 && end < n) {
                    if (row[end] == '-') ++gaps; else { ++sz; gaps = 0; }
                    ++end;
                }
                if (sz < 10) {
                    while (end < n && row[end] == '-') { ++end; ++gaps; }
                    if (gaps >= 20) {
                        std::fill(row.begin() + pos, row.begin() + end, '-');
                        reads[i].qual.erase(0, sz);
                        reads[i].seq.erase(0, sz);
                        pos = end;
                        continue;
                    }
                }
                std::reverse(row.begin(), row.end());
                std::reverse(reads[i].qual.begin(), reads[i].qual.end());
                std::reverse(reads[i].seq.begin(), reads[i].seq.end());
                stopped = true;
                break;
            }
            if (!stopped) break;     // ran off the row end: no reversal, no second phase (goto never taken)
        }
    }
}

struct colinfo { double err[6]; int occ[6]; int total; };

// correct.cpp:94-193 (n_threads = 1 accumulation order: rows in order).
void column_vote(const std::vector<hread> &reads, const std::vector<std::string> &aln, const vote_t &V,
                 std::vector<colinfo> &cols, std::string &cons) {
    cols.clear(); cons.clear();
    if (reads.empty() || aln.empty()) return;
    const size_t W = aln[0].size();
    cols.assign(W, colinfo{});
    for (size_t i = 0; i < reads.size(); ++i) {
        const std::string &row = aln[i];
        const std::string &q = reads[i].qual;
        long sp = -1;
        const long qn = (long)q.size();
        for (size_t k = 0; k < W; ++k) {
            const char nt = row[k];
            double ep = 0.0;
            if (nt != '-') { ++sp; ep = V.perr[(unsigned char)(sp < qn ? q[sp] : 0)]; }
            if (sp >= 0 && sp < qn) {
                const int s = V.slot[(unsigned char)nt];
                colinfo &c = cols[k];
                c.occ[s]++;
                c.err[s] += ep;
                if (sp == qn - 1) ++sp;
            }
        }
    }
    cons.resize(W);
    for (size_t k = 0; k < W; ++k) {
        colinfo &c = cols[k];
        int tot = 0;
        for (int s = 0; s < 6; ++s) tot += c.occ[s];
        c.total = tot;
        int best = 0; char nt = 0;
        for (int s = 0; s < 6; ++s) {             // reference iteration order, strict '>' (correct.cpp:174-186)
            if (c.occ[s] > 0) c.err[s] /= double(c.occ[s]);
            if (c.occ[s] > best) { best = c.occ[s]; nt = V.order[s]; }
        }
        cons[k] = nt == 0 ? '-' : nt;
    }
}

// correct.cpp:196-309
void correct_pack(pack_t &pk, const std::vector<std::string> &aln, const vote_t &V, double min_occ, double gap_occ,
                  double err_ratio) {
    std::vector<colinfo> cols;
    std::string cons;
    column_vote(pk.reads, aln, V, cols, cons);
    for (size_t i = 0; i < pk.reads.size(); ++i) {
        const std::string &row = aln[i];
        const std::string &q = pk.reads[i].qual;
        const long qn = (long)q.size();
        long sp = -1;
        hread out;
        out.rid = pk.reads[i].rid;
        out.seq.reserve(q.size() + 16);
        out.qual.reserve(q.size() + 16);
        for (size_t k = 0; k < row.size(); ++k) {
            const char nt = row[k];
            double ep = 0.0;
            if (nt != '-') { ++sp; ep = V.perr[(unsigned char)(sp < qn ? q[sp] : 0)]; }
            if (sp >= 0 && sp < qn) {
                const char cnt = cons[k];
                const int cs = V.slot[(unsigned char)cnt];
                const colinfo &c = cols[k];
                const double occ_ratio = double(c.occ[cs]) / double(c.total);
                const double cerr = c.err[cs];
                if (cnt == '-') {
                    if (nt != '-' && !(occ_ratio >= gap_occ)) { out.seq += nt; out.qual += q[sp]; }
                } else if (nt == '-') {
                    if (occ_ratio >= gap_occ) { out.seq += cnt; out.qual += phred_symbol(cerr); }
                } else if (nt == cnt) {
                    out.seq += nt; out.qual += q[sp];
                } else if (occ_ratio >= min_occ && err_ratio * ep > cerr) {
                    out.seq += cnt; out.qual += phred_symbol(cerr);
                } else {
                    out.seq += nt; out.qual += q[sp];
                }
                if (sp == qn - 1) ++sp;
            }
        }
        if (!out.seq.empty()) pk.corrected.push_back(std::move(out));
        else pk.dropped.push_back(pk.reads[i]);
    }
}

std::string strip_gaps(const std::string &s) {
    std::string o;
    o.reserve(s.size());
    for (char c : s) if (c != '-') o += c;
    return o;
}

// Run one POA stage over a list of sequence groups; returns rows per group.
int poa_stage(rattle_ctx *ctx, const std::vector<const std::vector<hread> *> &groups,
              std::vector<std::vector<std::string>> &msas, uint64_t *counters) {
    msas.assign(groups.size(), {});
    std::vector<uint64_t> off(1, 0);
    std::vector<uint32_t> first(1, 0);
    for (auto g : groups) {
        for (auto &r : *g) off.push_back(off.back() + r.seq.size());
        first.push_back((uint32_t)off.size() - 1);
    }
    std::string cat(off.back(), 'A');
    parallel_for(groups.size(), 0, [&](size_t g) {
        uint32_t q = first[g];
        for (auto &r : *groups[g]) { memcpy(&cat[off[q]], r.seq.data(), r.seq.size()); ++q; }
    });
    rattle_msa_set *ms = nullptr;
    int rc;
    {
        phase_timer T0("  poa_stage: msa_run");
        rc = poa_msa_run(ctx, (const uint8_t *)cat.data(), off.data(), (uint32_t)off.size() - 1, first.data(),
                         (uint32_t)groups.size(), &ms);
    }
    if (rc == 0) {
        phase_timer T1("  poa_stage: copy rows");
        parallel_for(groups.size(), 0, [&](size_t g) {
            uint32_t q = first[g];
            msas[g].resize(groups[g]->size());
            for (size_t i = 0; i < groups[g]->size(); ++i, ++q)
                msas[g][i].assign(ms->rows + ms->row_offset[q], ms->rows + ms->row_offset[q + 1]);
        });
        counters[0] += ms->counters[0];
        counters[1] += ms->counters[1];
    }
    rattle_hip_msa_set_free(ms);
    return rc;
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
    const size_t chunk = 1024;
    parallel_for((v.size() + chunk - 1) / chunk, 0, [&](size_t c) {
        for (size_t i = c * chunk; i < std::min(v.size(), (c + 1) * chunk); ++i) {
            S.read_id[i] = v[i].rid; S.cluster_id[i] = cid[i]; S.n_reads[i] = nr.empty() ? 0 : nr[i];
            const uint64_t p = S.off[i];
            memcpy(S.seq + p, v[i].seq.data(), v[i].seq.size());
            // qualities always have the sequence's length on this path; guard anyway
            const size_t ql = std::min(v[i].qual.size(), v[i].seq.size());
            memcpy(S.qual + p, v[i].qual.data(), ql);
            if (ql < v[i].seq.size()) memset(S.qual + p + ql, '!', v[i].seq.size() - ql);
        }
    });
    S.seq[tot] = 0; S.qual[tot] = 0;
}

}  // namespace

int correct_driver(rattle_ctx *ctx, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint32_t n_reads,
                   uint32_t n_clusters, const uint32_t *coff, const int32_t *mid, const uint8_t *mrev,
                   const rattle_correct_params *P, rattle_correction **out) {
    vote_t V;
    V.init(P->vote_order[0] ? P->vote_order : "U-GTCA");
    for (int i = 0; i < 6; ++i)
        if (!strchr("ACGTU-", V.order[i])) { set_error("vote_order must be a permutation of ACGTU-"); return RATTLE_ERR_ARG; }
    const int split = P->split > 0 ? P->split : 200;
    rattle_correction *R = (rattle_correction *)calloc(1, sizeof(rattle_correction));
    *out = R;

    phase_timer T_all("correct: total");
    std::vector<pack_t> packs;
    std::vector<hread> uncorrected;
    std::vector<int32_t> unc_cid;
    std::vector<std::vector<size_t>> cluster_packs(n_clusters);
    // ---- correct.cpp:328-370 pack building (clusters are independent: built in parallel, kept in order)
    {
        phase_timer T("correct: build packs");
        struct built { std::vector<pack_t> packs; std::vector<hread> small; int err = 0; };
        std::vector<built> B(n_clusters);
        parallel_for(n_clusters, P->n_threads, [&](size_t c) {
            const uint32_t a = coff[c], b = coff[c + 1];
            const int n = (int)(b - a);
            if (n == 0) return;
            const int n_files = (n - 1) / split + 1;
            for (int nf = 0; nf < n_files; ++nf) {
                pack_t pk;
                pk.cid = (int32_t)c;
                for (int j = nf; j < n; j += n_files) {
                    const int32_t rid = mid[a + j];
                    if (rid < 0 || (uint32_t)rid >= n_reads) { B[c].err = 1; return; }
                    hread r;
                    r.rid = rid;
                    r.seq.assign((const char *)seq + off[rid], (const char *)seq + off[rid + 1]);
                    r.qual.assign((const char *)qual + off[rid], (const char *)qual + off[rid + 1]);
                    if (mrev[a + j]) {                                   // :343-346
                        std::string rc(r.seq.size(), 'A');
                        for (size_t t = 0; t < r.seq.size(); ++t) rc[t] = comp_base(r.seq[r.seq.size() - 1 - t]);
                        r.seq.swap(rc);
                        std::reverse(r.qual.begin(), r.qual.end());
                    }
                    pk.reads.push_back(std::move(r));
                }
                if ((int)pk.reads.size() > P->min_reads) B[c].packs.push_back(std::move(pk));     // :360 strict
                else for (auto &r : pk.reads) B[c].small.push_back(std::move(r));
            }
        });
        for (uint32_t c = 0; c < n_clusters; ++c) {
            if (B[c].err) { set_error("cluster member id out of range"); return RATTLE_ERR_ARG; }
            for (auto &pk : B[c].packs) { cluster_packs[c].push_back(packs.size()); packs.push_back(std::move(pk)); }
            for (auto &r : B[c].small) { uncorrected.push_back(std::move(r)); unc_cid.push_back((int32_t)c); }
        }
    }
    uint64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    counters[2] = packs.size();

    // ---- POA #1 (correct.cpp:398-405) + fix ends + correction (:407-409)
    std::vector<std::vector<std::string>> msas;
    {
        phase_timer T("correct: POA#1 stage");
        std::vector<const std::vector<hread> *> groups;
        for (auto &p : packs) groups.push_back(&p.reads);
        RT_TRY(poa_stage(ctx, groups, msas, counters));
    }
    { phase_timer T("correct: vote+correct host");
    parallel_for(packs.size(), P->n_threads, [&](size_t i) {
        fix_msa_ends(packs[i].reads, msas[i]);
        correct_pack(packs[i], msas[i], V, P->min_occ, P->gap_occ, P->err_ratio);
        msas[i].clear();
    }); }
    std::vector<hread> corrected;
    std::vector<int32_t> cor_cid;
    for (auto &p : packs) {                                         // :413-425 (pack order)
        for (auto &r : p.corrected) { corrected.push_back(r); cor_cid.push_back(p.cid); }
        for (auto &r : p.dropped) { uncorrected.push_back(r); unc_cid.push_back(p.cid); }
    }
    // ---- POA #2 over the corrected reads, stably sorted by length desc (:427-445)
    {
        phase_timer T("correct: POA#2 stage");
        std::vector<const std::vector<hread> *> groups;
        for (auto &p : packs) {
            std::stable_sort(p.corrected.begin(), p.corrected.end(), [](const hread &a, const hread &b) { return a.seq.size() > b.seq.size(); });
            groups.push_back(&p.corrected);
        }
        RT_TRY(poa_stage(ctx, groups, msas, counters));
    }
    parallel_for(packs.size(), P->n_threads, [&](size_t i) {
        fix_msa_ends(packs[i].corrected, msas[i]);
        std::vector<colinfo> cols;
        std::string cons;
        column_vote(packs[i].corrected, msas[i], V, cols, cons);
        packs[i].consensus = strip_gaps(cons);
        msas[i].clear();
    });
    // ---- per-cluster consensus (:489-556); POA #3 for clusters with more than one pack
    std::vector<std::vector<hread>> multi;
    std::vector<uint32_t> multi_cid;
    for (uint32_t c = 0; c < n_clusters; ++c) {
        if (cluster_packs[c].size() > 1) {
            std::vector<hread> g;
            for (size_t pi : cluster_packs[c]) g.push_back(hread{packs[pi].consensus, std::string(packs[pi].consensus.size(), 'K'), -1});
            multi.push_back(std::move(g));
            multi_cid.push_back(c);
        }
    }
    std::vector<std::string> multi_cons(multi.size());
    if (!multi.empty()) {
        std::vector<const std::vector<hread> *> groups;
        for (auto &g : multi) groups.push_back(&g);
        RT_TRY(poa_stage(ctx, groups, msas, counters));
        parallel_for(multi.size(), P->n_threads, [&](size_t i) {
            fix_msa_ends(multi[i], msas[i]);
            std::vector<colinfo> cols;
            std::string cons;
            column_vote(multi[i], msas[i], V, cols, cons);
            multi_cons[i] = strip_gaps(cons);
        });
    }
    std::vector<hread> consensi;
    std::vector<int32_t> con_cid, con_n;
    size_t mi = 0;
    for (uint32_t c = 0; c < n_clusters; ++c) {
        if (cluster_packs[c].empty()) continue;
        int total = 0;
        for (size_t pi : cluster_packs[c]) total += (int)packs[pi].reads.size();
        std::string s;
        if (cluster_packs[c].size() > 1) s = multi_cons[mi++];
        else s = packs[cluster_packs[c][0]].consensus;
        consensi.push_back(hread{s, std::string(s.size(), 'K'), -1});
        con_cid.push_back((int32_t)c);
        con_n.push_back(total);
    }
    phase_timer T_fill("correct: fill results");
    fill_set(R->corrected, corrected, cor_cid, {});
    fill_set(R->uncorrected, uncorrected, unc_cid, {});
    fill_set(R->consensi, consensi, con_cid, con_n);
    memcpy(R->counters, counters, sizeof(counters));
    return 0;
}

}  // namespace rattle
