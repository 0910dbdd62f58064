// ORACLE (test infrastructure, never shipped in the product path).
//
// CPU restatement of RATTLE's `cluster` hot path.  Each function cites the
// reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may build, link or call anything in oracle/.
//
// Pinning: `orc::cluster_reads` reproduces toyset/rna/output/clusters.out
// (546 clusters, 8306 reads) byte-for-byte -- tests/test_oracle_cluster.py.
// The unit functions (k-mer lists, intersection, LIS, var) are additionally
// checked against the real reference TUs built into oracle/_ref/ (kmer.cpp,
// similarity.cpp, utils.cpp compile from /root/reference without stand-ins;
// cluster.cpp does not -- it needs the absent hps header -- so the driver
// logic below is pinned by the shipped fixture only).
#pragma once
#include <algorithm>
#include <bitset>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

namespace orc {

struct read_t {            // fasta.hpp:7-12
    std::string header, seq, ann, quality;
};
typedef std::vector<read_t> read_set_t;

typedef std::pair<uint32_t, int> kmer_t;        // kmer.hpp:10  (hash, pos)
typedef std::pair<int, int> kmer_match_t;       // kmer.hpp:11  (pos1, pos2)
const int KMER_BV_SIZE = 6;                     // kmer.hpp:14
const int BV_SIZE = 4096;                       // kmer.hpp:15
typedef std::bitset<BV_SIZE> kmer_bv_t;

struct read_kmers_t {                           // kmer.hpp:18-23
    std::vector<kmer_t> list_forward, list_reverse;
    kmer_bv_t bv_forward, bv_reverse;
};

struct cseq_t { int seq_id; bool rev; int gene_id = -1; };          // cluster.hpp:10-13
struct cluster_t { cseq_t main_seq; std::vector<cseq_t> seqs; };    // cluster.hpp:27-29
typedef std::vector<cluster_t> cluster_set_t;

// kmer.hpp:25-31: A=0 C=1 T=2 U=2 G=3.  Any other byte is undefined behaviour in
// the reference (end() dereference); the oracle aborts instead.
inline uint32_t base_code(char c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'T': case 'U': return 2;
        case 'G': return 3;
    }
    fprintf(stderr, "oracle: invalid base 0x%02x\n", (unsigned char)c);
    abort();
}

// utils.hpp:8-14, utils.cpp:15-24: A<->T, C<->G, U->A.
inline std::string reverse_complement(const std::string &s) {
    std::string r(s.size(), 'A');
    size_t n = s.size();
    for (size_t i = 0; i < n; ++i) {
        char c = s[n - 1 - i];
        char o;
        switch (c) {
            case 'A': o = 'T'; break;
            case 'C': o = 'G'; break;
            case 'T': o = 'A'; break;
            case 'G': o = 'C'; break;
            case 'U': o = 'A'; break;
            default: fprintf(stderr, "oracle: invalid base in revcomp\n"); abort();
        }
        r[i] = o;
    }
    return r;
}

// kmer.hpp:33-40: MSB-first 2-bit packing of s[p..p+k).
inline uint32_t hash_kmer(const std::string &s, size_t p, int k) {
    uint32_t h = 0;
    for (int i = 0; i < k; ++i) h = (h << 2) | base_code(s[p + i]);
    return h;
}

// kmer.cpp:6-42.  List has L-k entries (positions 0..L-k-1: the last k-mer is
// dropped); the bit-vector is always over 6-mers at positions 0..L-7.
// Reads shorter than k are UB in the reference (size_t underflow); here: empty.
inline read_kmers_t extract_kmers_from_read(const std::string &read, int k, bool both) {
    read_kmers_t r;
    std::string rc = reverse_complement(read);
    long L = (long)read.size();
    long nk = std::max(0L, L - k);
    long nb = std::max(0L, L - KMER_BV_SIZE);
    r.list_forward.resize(nk);
    r.list_reverse.resize(nk);          // kmer.cpp:10: allocated (zero pairs) even when unused
    for (long p = 0; p < nk; ++p) {
        r.list_forward[p] = kmer_t(hash_kmer(read, p, k), (int)p);
        if (both) r.list_reverse[p] = kmer_t(hash_kmer(rc, p, k), (int)p);
    }
    for (long p = 0; p < nb; ++p) {
        r.bv_forward.set(hash_kmer(read, p, KMER_BV_SIZE));
        if (both) r.bv_reverse.set(hash_kmer(rc, p, KMER_BV_SIZE));
    }
    std::sort(r.list_forward.begin(), r.list_forward.end());
    if (both) std::sort(r.list_reverse.begin(), r.list_reverse.end());
    return r;
}

// kmer.cpp:45-67: backward merge-join with full cross product on repeated hashes,
// result sorted by (pos1,pos2).
inline std::vector<kmer_match_t> get_common_kmers(const std::vector<kmer_t> &k1, const std::vector<kmer_t> &k2) {
    long p1 = (long)k1.size() - 1, p2 = (long)k2.size() - 1;
    std::vector<kmer_match_t> out;
    while (p1 >= 0 && p2 >= 0) {
        while (p2 >= 0 && k2[p2].first > k1[p1].first) --p2;
        long keep = p2;
        while (p2 >= 0 && k2[p2].first == k1[p1].first) {
            out.push_back(kmer_match_t(k1[p1].second, k2[p2].second));
            --p2;
        }
        p2 = keep;
        --p1;
    }
    std::sort(out.begin(), out.end());
    return out;
}

struct similarity_res_t {        // similarity.hpp:7-13 (lis itself is unused by callers)
    int llis = 0, bases = 0, hc_bases = 0;
    std::vector<int> distances;
};

// similarity.cpp:4-97.
inline similarity_res_t calc_similarity(const std::vector<kmer_match_t> &common, int k) {
    size_t n = common.size();
    std::vector<int> p(n), m(n + 1);
    int l = 0;
    for (size_t i = 0; i < n; ++i) {             // :10-31 patience with ceil-mid search
        int lo = 1, hi = l;
        while (lo <= hi) {
            int mid = (lo + hi + 1) / 2;
            if (common[m[mid]].second < common[i].second) lo = mid + 1;
            else hi = mid - 1;
        }
        p[i] = m[lo - 1];
        m[lo] = (int)i;
        if (lo > l) l = lo;
    }
    similarity_res_t res;
    if (l <= 0) return res;
    std::vector<kmer_match_t> s(l);
    int cur = m[l];
    for (int i = l - 1; i >= 0; --i) { s[i] = common[cur]; cur = p[cur]; }   // :37-44
    int bases = 0, hc = 0;
    kmer_match_t kept_prev, kept;            // last two kept elements
    int nkept = 0;
    for (int i = 0; i < l; ++i) {
        if (i == 0) {                        // :80-84
            kept = s[0]; nkept = 1; bases += k; hc += k;
            continue;
        }
        int d1 = s[i].first - kept.first, d2 = s[i].second - kept.second;
        if ((d1 < k && d2 < k) || (d1 >= k && d2 >= k)) {      // :54-59 vs last KEPT
            bases += k;
            int ex = k - (s[i].second - s[i - 1].second);      // :62 vs previous CHAIN element
            if (ex > 0) bases -= ex;
            kept_prev = kept; kept = s[i]; ++nkept;
            int dist = (kept.second - kept_prev.second) - (kept.first - kept_prev.first);   // :69-71
            res.distances.push_back(dist);
            if (dist < 10) { hc += k; if (ex > 0) hc -= ex; }  // :73-78
        }
    }
    res.llis = nkept; res.bases = bases; res.hc_bases = hc;
    return res;
}

// utils.cpp:26-34
inline double mean(const std::vector<int> &s) {
    double r = 0.0;
    for (int n : s) r += double(n);
    return r / double(s.size());
}

// utils.cpp:36-55: corrected two-pass variance; size 0 -> 0, size 1 -> 0/0 = NaN.
inline double var(const std::vector<int> &s) {
    if (s.size() == 0) return 0;
    double ss = 0.0, comp = 0.0, m = mean(s);
    for (int n : s) { double d = n - m; ss += d * d; comp += d; }
    return (ss - comp * comp / double(s.size())) / double(s.size() - 1);
}

struct work_counters_t {            // exact work counts: not in the reference
    uint64_t pair_tests = 0, full_cmp = 0, matches = 0;
};

struct kmer_index_t {
    std::vector<std::vector<kmer_t>> kmers, rev_kmers;
    std::vector<kmer_bv_t> bv, rev_bv;
    std::vector<size_t> bv_count;
};

// cluster.cpp:12-65
inline cseq_t cluster_together(const read_set_t &reads, const kmer_index_t &x, int i, int j, int k,
                               double t_s, double t_v, double bv_threshold, bool use_hc, bool is_rna,
                               work_counters_t *wc) {
    size_t bv_common = (x.bv[i] & x.bv[j]).count();
    size_t rev_bv_common = (x.bv[i] & x.rev_bv[j]).count();
    double mmax = (double)std::max(x.bv[i].count(), x.bv[j].count());      // :16 forward counts only
    if (wc) wc->pair_tests++;
    if (bv_threshold == 0 || bv_common / mmax >= bv_threshold) {           // :19
        auto common = get_common_kmers(x.kmers[i], x.kmers[j]);
        auto sim = calc_similarity(common, k);
        if (wc) { wc->full_cmp++; wc->matches += common.size(); }
        double mn = (double)std::min(reads[i].seq.size(), reads[j].seq.size());
        double score = use_hc ? double(sim.hc_bases) / mn : double(sim.bases) / mn;
        if (score >= t_s) {
            if (var(sim.distances) < t_v) return cseq_t{j, false};
        }
    }
    if (is_rna) return cseq_t{-1, false};
    if (rev_bv_common / mmax >= bv_threshold) {                            // :43
        auto common = get_common_kmers(x.kmers[i], x.rev_kmers[j]);
        auto sim = calc_similarity(common, k);
        if (wc) { wc->full_cmp++; wc->matches += common.size(); }
        double mn = (double)std::min(reads[i].seq.size(), reads[j].seq.size());
        double score = use_hc ? double(sim.hc_bases) / mn : double(sim.bases) / mn;
        if (score >= t_s) {
            if (var(sim.distances) < t_v) return cseq_t{j, true};
        }
    }
    return cseq_t{-1, false};
}

// cluster.cpp:67-91
inline cseq_t get_main_seq(std::vector<cseq_t> &seqs, const read_set_t &reads, double repr_percentile) {
    cseq_t old = seqs[0];
    std::stable_sort(seqs.begin(), seqs.end(), [](const cseq_t &a, const cseq_t &b) { return a.seq_id > b.seq_id; });
    std::stable_sort(seqs.begin(), seqs.end(), [&reads](const cseq_t &a, const cseq_t &b) {
        return reads[a.seq_id].seq.size() > reads[b.seq_id].seq.size();
    });
    int nsid = seqs.size() * repr_percentile;
    cseq_t ns = seqs[nsid];
    while (ns.rev != old.rev && (size_t)nsid < seqs.size() - 1) { nsid++; ns = seqs[nsid]; }
    if ((size_t)nsid == seqs.size() - 1) return old;
    return ns;
}

// cluster.cpp:93-259, sequential (the reference's result does not depend on n_threads:
// each j is owned by one task and get_main_seq canonicalises member order).
inline cluster_set_t cluster_reads(const read_set_t &reads, int k, double t_s, double t_v, double bv_threshold,
                                   double min_bv_threshold, double bv_falloff, int /*min_reads_cluster*/,
                                   bool use_hc, double repr_percentile, bool is_rna,
                                   work_counters_t *wc = nullptr) {
    int n = (int)reads.size();
    kmer_index_t x;
    x.kmers.resize(n); x.rev_kmers.resize(n); x.bv.resize(n); x.rev_bv.resize(n);
    for (int i = 0; i < n; ++i) {                               // :105-121
        read_kmers_t r = extract_kmers_from_read(reads[i].seq, k, !is_rna);
        x.kmers[i].swap(r.list_forward); x.rev_kmers[i].swap(r.list_reverse);
        x.bv[i] = r.bv_forward; x.rev_bv[i] = r.bv_reverse;
    }
    std::vector<bool> done(n, false);
    cluster_set_t clusters;
    for (int i = 0; i < n; ++i) {                               // :125-166
        if (done[i]) continue;
        std::vector<cseq_t> cseqs;
        cseqs.push_back(cseq_t{i, false});
        done[i] = true;
        for (int j = i + 1; j < n; ++j) {
            if (done[j]) continue;
            cseq_t s = cluster_together(reads, x, i, j, k, t_s, t_v, bv_threshold, use_hc, is_rna, wc);
            if (s.seq_id != -1) { done[s.seq_id] = true; cseqs.push_back(s); }
        }
        cluster_t c;
        c.main_seq = get_main_seq(cseqs, reads, repr_percentile);
        c.seqs = cseqs;
        clusters.push_back(c);
    }
    double thr = bv_threshold - bv_falloff;                     // :171
    bool last = false;
    while (thr >= min_bv_threshold || last) {                   // :173
        cluster_set_t tmp;
        int nc = (int)clusters.size();
        done.assign(nc, false);
        for (int i = 0; i < nc; ++i) {
            if (done[i]) continue;
            done[i] = true;
            std::vector<cseq_t> to_merge;
            to_merge.push_back(cseq_t{i, false});
            for (int j = i + 1; j < nc; ++j) {
                if (done[j]) continue;
                cseq_t s = cluster_together(reads, x, clusters[i].main_seq.seq_id, clusters[j].main_seq.seq_id,
                                            k, t_s, t_v, thr, use_hc, is_rna, wc);     // :197 main_seq.rev ignored
                if (s.seq_id != -1) { done[j] = true; to_merge.push_back(cseq_t{j, s.rev}); }
            }
            cluster_t c;
            cseq_t original = to_merge[0];
            for (auto &m : to_merge) {                           // :227-238
                for (auto s : clusters[m.seq_id].seqs) {
                    if (m.rev != original.rev) s.rev = !s.rev;
                    c.seqs.push_back(s);
                }
            }
            c.main_seq = get_main_seq(c.seqs, reads, repr_percentile);
            tmp.push_back(c);
        }
        clusters.swap(tmp);
        if (last) break;
        thr -= bv_falloff;                                      // :251-255
        if (thr < min_bv_threshold && !last) { last = true; thr = 0.0; }
    }
    return clusters;
}

// fasta.cpp:458-464
inline void sort_read_set(read_set_t &rs) {
    std::stable_sort(rs.begin(), rs.end(), [](const read_t &a, const read_t &b) { return a.seq.size() > b.seq.size(); });
}

struct cluster_params_t {
    int k = 10; double t_s = 0.2, t_v = 1000000;
    int iso_k = 11; double iso_t_s = 0.3, iso_t_v = 25;
    double bv_threshold = 0.4, bv_min_threshold = 0.2, bv_falloff = 0.05;
    int min_reads_cluster = 0; double repr_percentile = 0.15;
    bool is_rna = false, iso = false;
};

// main.cpp:254-323: `reads` already filtered, `ann` = decimal original index.
// Sorts `reads` in place (main.cpp:254) and returns clusters with ORIGINAL ids.
inline cluster_set_t cluster_command(read_set_t &reads, const cluster_params_t &P, work_counters_t *wc = nullptr) {
    sort_read_set(reads);
    cluster_set_t gene = cluster_reads(reads, P.k, P.t_s, P.t_v, P.bv_threshold, P.bv_min_threshold, P.bv_falloff,
                                       P.min_reads_cluster, false, P.repr_percentile, P.is_rna, wc);
    if (!P.iso) {                                               // :264-277
        for (auto &c : gene) {
            c.main_seq.seq_id = std::stoi(reads[c.main_seq.seq_id].ann);
            for (auto &s : c.seqs) s.seq_id = std::stoi(reads[s.seq_id].ann);
        }
        return gene;
    }
    cluster_set_t iso;                                          // :281-318
    int gi = 0;
    for (auto &c : gene) {
        std::stable_sort(c.seqs.begin(), c.seqs.end(), [](const cseq_t &a, const cseq_t &b) { return a.seq_id > b.seq_id; });
        std::stable_sort(c.seqs.begin(), c.seqs.end(), [&reads](const cseq_t &a, const cseq_t &b) {
            return reads[a.seq_id].seq.size() > reads[b.seq_id].seq.size();
        });
        read_set_t gene_reads;
        for (auto &s : c.seqs) gene_reads.push_back(reads[s.seq_id]);
        cluster_set_t sub = cluster_reads(gene_reads, P.iso_k, P.iso_t_s, P.iso_t_v, P.bv_threshold, P.bv_min_threshold,
                                          P.bv_falloff, P.min_reads_cluster, false, P.repr_percentile, P.is_rna, wc);
        for (auto &ic : sub) {
            cluster_t o;
            o.main_seq = cseq_t{std::stoi(reads[c.seqs[ic.main_seq.seq_id].seq_id].ann), ic.main_seq.rev, gi};
            for (auto &s : ic.seqs) o.seqs.push_back(cseq_t{std::stoi(reads[c.seqs[s.seq_id].seq_id].ann), s.rev, gi});
            iso.push_back(o);
        }
        ++gi;
    }
    return iso;
}

}  // namespace orc
