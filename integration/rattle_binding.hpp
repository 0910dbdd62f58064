// The reference-side binding: bodies for RATTLE's two seams over include/rattle_hip.h.
//
//   cluster_set_t cluster_reads(...)          /root/reference/cluster.hpp:44   (called at main.cpp:258,300,669)
//   correction_results_t correct_reads(...)   /root/reference/correct.hpp:44   (called at main.cpp:405,670)
//
// A maintainer deletes cluster.cpp's and correct.cpp's definitions of these two functions, includes this file
// in one translation unit AFTER fasta.hpp / cluster.hpp / correct.hpp (it uses only the reference's own types:
// read_t, read_set_t, cseq_t, cluster_t, cluster_set_t, correction_results_t) and links -lrattle_hip; spoa is no
// longer needed.  tests/test_binding.py compiles exactly this file against restated type declarations
// (tests/binding/ref_types.hpp) and compares its output with the Python mirror's.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <set>
#include <string>
#include <vector>

#include "rattle_hip.h"

namespace rattle_binding {

inline rattle_ctx *&ctx_slot() { static rattle_ctx *c = nullptr; return c; }     // one context per process (device RATTLE_DEVICE or 0)
inline void check(int rc) { if (rc) throw std::runtime_error(rattle_hip_last_error()); }
inline rattle_ctx *ctx() {
    if (!ctx_slot()) {
        const char *d = getenv("RATTLE_DEVICE");
        check(rattle_hip_ctx_create(d ? atoi(d) : 0, &ctx_slot()));
    }
    return ctx_slot();
}
inline void shutdown() { rattle_hip_ctx_destroy(ctx_slot()); ctx_slot() = nullptr; }

}  // namespace rattle_binding

// cluster.hpp:44.  `reads` arrive length-sorted (main.cpp:254, :283-291, :666); ids in the result index them.
cluster_set_t cluster_reads(const read_set_t &reads, int kmer_size, double t_s, double t_v, double bv_threshold, double min_bv_threshold,
                            double bv_falloff, int min_reads_cluster, bool use_hc, double repr_percentile, bool is_rna, bool /*verbose*/,
                            int /*n_threads*/) {
    using namespace rattle_binding;
    std::string cat;
    std::vector<uint64_t> off(1, 0);
    for (const read_t &r : reads) { cat += r.seq; off.push_back(cat.size()); }
    check(rattle_hip_load_reads(ctx(), (const uint8_t *)cat.data(), off.data(), (uint32_t)reads.size(), kmer_size, is_rna ? 0 : 1));
    rattle_cluster_params p;
    p.t_s = t_s; p.t_v = t_v; p.bv_threshold = bv_threshold; p.min_bv_threshold = min_bv_threshold; p.bv_falloff = bv_falloff;
    p.min_reads_cluster = min_reads_cluster; p.use_hc = use_hc ? 1 : 0; p.repr_percentile = repr_percentile; p.is_rna = is_rna ? 1 : 0;
    rattle_cluster_set *cs = nullptr;
    check(rattle_hip_cluster_reads(ctx(), &p, &cs));
    cluster_set_t out(cs->n_clusters);
    for (uint32_t c = 0; c < cs->n_clusters; ++c) {
        out[c].main_seq.seq_id = cs->main_id[c]; out[c].main_seq.rev = cs->main_rev[c] != 0;       // gene_id keeps its default -1
        for (uint32_t i = cs->offsets[c]; i < cs->offsets[c + 1]; ++i) {
            cseq_t s;
            s.seq_id = cs->member_id[i]; s.rev = cs->member_rev[i] != 0;
            out[c].seqs.push_back(s);
        }
    }
    rattle_hip_cluster_set_free(cs);
    return out;
}

// correct.hpp:44.  `reads` in file order with qualities (main.cpp:386); clusters index them by seq_id.
// Unlike the reference the read set is left untouched (correct.cpp:343-353 reverse-complements rev members and
// appends the cluster tags in place; nothing reads them afterwards).
correction_results_t correct_reads(const cluster_set_t &clusters, read_set_t &reads, double min_occ, double gap_occ, double err_ratio, int split,
                                   int min_reads, int n_threads, bool /*verbose*/, std::vector<std::string> labels) {
    using namespace rattle_binding;
    correction_results_t out;
    if (clusters.empty()) return out;
    std::string cat, qcat;
    std::vector<uint64_t> off(1, 0);
    for (const read_t &r : reads) {
        cat += r.seq;
        std::string q = r.quality;
        q.resize(r.seq.size(), '!');
        qcat += q;
        off.push_back(cat.size());
    }
    std::vector<uint32_t> coff(1, 0);
    std::vector<int32_t> mid;
    std::vector<uint8_t> mrev;
    for (const cluster_t &c : clusters) {
        for (const cseq_t &s : c.seqs) { mid.push_back(s.seq_id); mrev.push_back(s.rev ? 1 : 0); }
        coff.push_back((uint32_t)mid.size());
    }
    if (mid.empty()) { mid.push_back(0); mrev.push_back(0); }
    rattle_correct_params p;
    memset(&p, 0, sizeof(p));
    p.min_occ = min_occ; p.gap_occ = gap_occ; p.err_ratio = err_ratio; p.split = split; p.min_reads = min_reads; p.n_threads = n_threads;
    rattle_correction *R = nullptr;
    check(rattle_hip_correct_reads(ctx(), (const uint8_t *)cat.data(), (const uint8_t *)qcat.data(), off.data(), (uint32_t)reads.size(),
                                   (uint32_t)clusters.size(), coff.data(), mid.data(), mrev.data(), &p, &R));
    const bool gene_mode = clusters[0].main_seq.gene_id == -1;                                        // correct.cpp:322
    auto tag = [&](int cid) {                                                                         // correct.cpp:348-353
        const int gid = clusters[cid].main_seq.gene_id;
        if (gid == -1) return ",gene_cluster_" + std::to_string(cid);
        return ",gene_cluster_" + std::to_string(gid) + ",transcript_cluster_" + std::to_string(cid);
    };
    auto to_reads = [&](const rattle_read_set &S, bool corrected, read_set_t &dst) {
        for (uint32_t i = 0; i < S.n; ++i) {
            read_t r;
            r.header = reads[S.read_id[i]].header + tag(S.cluster_id[i]);
            r.seq.assign(S.seq + S.off[i], S.seq + S.off[i + 1]);
            r.ann = corrected ? "+" : reads[S.read_id[i]].ann;                                        // correct.cpp:286 vs :289-293, :362-366
            r.quality.assign(S.qual + S.off[i], S.qual + S.off[i + 1]);
            dst.push_back(r);
        }
    };
    to_reads(R->corrected, true, out.corrected);
    to_reads(R->uncorrected, false, out.uncorrected);
    // consensus headers, correct.cpp:453-469,495-549: labels counted over the reads of the cluster's queued packs
    std::vector<std::vector<int>> label_counts(clusters.size(), std::vector<int>(labels.size(), 0));
    if (!labels.empty()) {
        // packs the library skipped (rattle_skip_list: stage 0 indexed among all packs of the cluster, stages 1-2 among its
        // queued packs) produced no pack consensus and are not counted, as in `reads=`
        std::set<std::pair<int, uint32_t>> skip_all, skip_queued;
        for (uint32_t i = 0; i < R->skipped.n; ++i) {
            if (R->skipped.stage[i] == 0) skip_all.insert({R->skipped.cluster_id[i], R->skipped.pack[i]});
            else if (R->skipped.stage[i] <= 2) skip_queued.insert({R->skipped.cluster_id[i], R->skipped.pack[i]});
        }
        for (size_t c = 0; c < clusters.size(); ++c) {
            const int n = (int)clusters[c].seqs.size();
            if (n == 0) continue;
            const int n_files = (n - 1) / split + 1;
            uint32_t queued = 0;
            for (int nf = 0; nf < n_files; ++nf) {
                if ((n - 1 - nf) / n_files + 1 <= min_reads) continue;
                if (skip_all.count({(int)c, (uint32_t)nf})) continue;
                if (skip_queued.count({(int)c, queued++})) continue;
                for (int j = nf; j < n; j += n_files) {
                    const std::string &h = reads[clusters[c].seqs[j].seq_id].header;
                    const size_t q = h.find_first_of(",");
                    const std::string rest = q == std::string::npos ? "" : h.substr(q + 1);
                    const std::string lab = rest.substr(0, rest.find_first_of(","));
                    for (size_t l = 0; l < labels.size(); ++l) if (labels[l] == lab) label_counts[c][l]++;
                }
            }
        }
    }
    for (uint32_t i = 0; i < R->consensi.n; ++i) {
        const int cid = R->consensi.cluster_id[i];
        std::string lr;
        for (size_t l = 0; l < labels.size(); ++l) lr += labels[l] + ":" + std::to_string(label_counts[cid][l]) + ",";
        read_t r;
        if (gene_mode) r.header = "@gene_cluster_" + std::to_string(cid) + " reads=" + std::to_string(R->consensi.n_reads[i]) + " labels=" + lr;
        else r.header = "@transcript_cluster_" + std::to_string(cid) + " gene_cluster_" + std::to_string(clusters[cid].main_seq.gene_id) + " reads=" +
                        std::to_string(R->consensi.n_reads[i]) + " labels=" + lr;
        r.seq.assign(R->consensi.seq + R->consensi.off[i], R->consensi.seq + R->consensi.off[i + 1]);
        r.ann = "+";
        r.quality.assign(R->consensi.qual + R->consensi.off[i], R->consensi.qual + R->consensi.off[i + 1]);
        out.consensi.push_back(r);
    }
    rattle_hip_correction_free(R);
    return out;
}
