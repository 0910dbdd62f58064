// ORACLE (test infrastructure, never shipped in the product path).
// extern "C" surface over the restatement for ctypes (tests/, smoke(), bench cpu_baseline).
#include <chrono>
#include <cstring>

#include "orc_cluster.hpp"
#include "orc_correct.hpp"
#include "orc_io.hpp"
#include "orc_poa.hpp"

using namespace orc;

static read_set_t make_reads(const char *seqs, const uint64_t *off, uint32_t n, const char *quals) {
    read_set_t rs(n);
    for (uint32_t i = 0; i < n; ++i) {
        rs[i].seq.assign(seqs + off[i], seqs + off[i + 1]);
        rs[i].ann = std::to_string(i);
        rs[i].header = "@r" + std::to_string(i);
        if (quals) rs[i].quality.assign(quals + off[i], quals + off[i + 1]);
    }
    return rs;
}

extern "C" {

// k-mer lists + bit-vectors of one read.  Lists hold max(L-k,0) entries.
void orc_extract_kmers(const char *seq, uint32_t len, int k, int both, uint32_t *fwd_hash, int32_t *fwd_pos,
                       uint32_t *rev_hash, int32_t *rev_pos, uint64_t *bv_fwd, uint64_t *bv_rev) {
    read_kmers_t r = extract_kmers_from_read(std::string(seq, len), k, both != 0);
    for (size_t i = 0; i < r.list_forward.size(); ++i) {
        fwd_hash[i] = r.list_forward[i].first; fwd_pos[i] = r.list_forward[i].second;
        if (both) { rev_hash[i] = r.list_reverse[i].first; rev_pos[i] = r.list_reverse[i].second; }
    }
    for (int w = 0; w < 64; ++w) {
        uint64_t a = 0, b = 0;
        for (int t = 0; t < 64; ++t) {
            if (r.bv_forward[w * 64 + t]) a |= 1ull << t;
            if (r.bv_reverse[w * 64 + t]) b |= 1ull << t;
        }
        bv_fwd[w] = a; bv_rev[w] = b;
    }
}

// Full pair comparison (get_common_kmers + calc_similarity + var) of read a vs read b
// (b's forward strand if strand==0, else b's reverse-complement strand).
void orc_pair_score(const char *a, uint32_t la, const char *b, uint32_t lb, int k, int strand, int32_t *bases,
                    int32_t *hc_bases, int32_t *n_dist, double *variance, int32_t *n_matches, int32_t *dist_out,
                    int32_t dist_cap) {
    read_kmers_t ka = extract_kmers_from_read(std::string(a, la), k, false);
    read_kmers_t kb = extract_kmers_from_read(std::string(b, lb), k, true);
    auto common = get_common_kmers(ka.list_forward, strand ? kb.list_reverse : kb.list_forward);
    auto sim = calc_similarity(common, k);
    *bases = sim.bases; *hc_bases = sim.hc_bases; *n_dist = (int32_t)sim.distances.size();
    *variance = var(sim.distances);
    *n_matches = (int32_t)common.size();
    for (int i = 0; i < (int)sim.distances.size() && i < dist_cap; ++i) dist_out[i] = sim.distances[i];
}

double orc_var(const int32_t *v, uint32_t n) { return var(std::vector<int>(v, v + n)); }

// cluster_reads on reads ALREADY in processing order (length-descending).  Output:
// flattened clusters with local ids.  Returns number of clusters; member arrays must hold n.
// counters[0..2] = pair tests, full comparisons, matches.
int32_t orc_cluster_reads(const char *seqs, const uint64_t *off, uint32_t n, int k, double t_s, double t_v, double bvB,
                          double bvb, double bvf, double repr_pct, int is_rna, int32_t *cl_main_id, uint8_t *cl_main_rev,
                          uint32_t *cl_off, int32_t *mem_id, uint8_t *mem_rev, uint64_t *counters) {
    read_set_t rs = make_reads(seqs, off, n, nullptr);
    work_counters_t wc;
    cluster_set_t cs = cluster_reads(rs, k, t_s, t_v, bvB, bvb, bvf, 0, false, repr_pct, is_rna != 0, &wc);
    uint32_t p = 0;
    for (size_t c = 0; c < cs.size(); ++c) {
        cl_main_id[c] = cs[c].main_seq.seq_id; cl_main_rev[c] = cs[c].main_seq.rev; cl_off[c] = p;
        for (auto &s : cs[c].seqs) { mem_id[p] = s.seq_id; mem_rev[p] = s.rev; ++p; }
    }
    cl_off[cs.size()] = p;
    if (counters) { counters[0] = wc.pair_tests; counters[1] = wc.full_cmp; counters[2] = wc.matches; }
    return (int32_t)cs.size();
}

// POA MSA of n sequences (correct.cpp:398-405 call pattern).  Rows are written
// back-to-back into msa_out (n * width bytes); returns width, or -1 if cap too small.
int64_t orc_poa_msa(const char *seqs, const uint64_t *off, uint32_t n, char *msa_out, uint64_t cap, uint64_t *cells) {
    std::vector<std::string> v(n);
    for (uint32_t i = 0; i < n; ++i) v[i].assign(seqs + off[i], seqs + off[i + 1]);
    uint64_t c = 0;
    std::vector<std::string> msa = poa_msa(v, &c);
    if (cells) *cells = c;
    size_t W = msa.empty() ? 0 : msa[0].size();
    if (W * n > cap) return -1;
    for (uint32_t i = 0; i < n; ++i) memcpy(msa_out + (size_t)i * W, msa[i].data(), W);
    return (int64_t)W;
}

// Whole `correct` on in-memory reads (file order, with qualities) + clusters in the hps
// byte encoding.  Writes the three FASTQ texts into malloc'd buffers (caller frees with orc_free).
int32_t orc_correct(const char *seqs, const char *quals, const uint64_t *off, uint32_t n, const char *const *headers,
                    const uint8_t *clusters_hps, uint64_t clusters_len, double min_occ, double gap_occ, int split,
                    int min_reads, char **corrected, char **uncorrected, char **consensi, uint64_t *counters,
                    uint32_t n_pack_orders, const uint32_t *po_cluster, const uint32_t *po_offsets, const uint32_t *po_perm) {
    read_set_t rs = make_reads(seqs, off, n, quals);
    for (uint32_t i = 0; i < n; ++i) { rs[i].ann = "+"; if (headers) rs[i].header = headers[i]; }
    cluster_set_t cs;
    std::string b((const char *)clusters_hps, clusters_len);
    if (!hps_decode(b, 3, cs) && !hps_decode(b, 2, cs)) return -1;
    correct_counters_t cc;
    std::map<int, std::vector<int>> po;
    for (uint32_t i = 0; i < n_pack_orders; ++i) po[(int)po_cluster[i]] = std::vector<int>(po_perm + po_offsets[i], po_perm + po_offsets[i + 1]);
    correction_results_t R = correct_reads(cs, rs, min_occ, gap_occ, 30.0, split, min_reads, {}, &cc, n_pack_orders ? &po : nullptr);
    auto dump = [](const read_set_t &v) {
        std::string s;
        for (auto &r : v) s += r.header + "\n" + r.seq + "\n" + r.ann + "\n" + r.quality + "\n";
        char *p = (char *)malloc(s.size() + 1);
        memcpy(p, s.data(), s.size());
        p[s.size()] = 0;
        return p;
    };
    *corrected = dump(R.corrected); *uncorrected = dump(R.uncorrected); *consensi = dump(R.consensi);
    if (counters) { counters[0] = cc.dp_cells; counters[1] = cc.packs; counters[2] = cc.alignments; }
    return 0;
}


// fix_msa_ends (correct.cpp:32-92) on hand-built MSA rows: rows are `width` bytes each and are rewritten in place; the
// reads' sequences / qualities (concatenated at off[]) are mutated the way the reference mutates them, and their new
// lengths returned (the bytes themselves are moved to the front of each read's slot).
void orc_fix_msa_ends(char *rows, uint32_t n, uint32_t width, char *seqs, char *quals, const uint64_t *off, uint32_t *len_out) {
    read_set_t rs;
    msa_t msa;
    for (uint32_t i = 0; i < n; ++i) {
        rs.push_back(read_t{"", std::string(seqs + off[i], seqs + off[i + 1]), "+", std::string(quals + off[i], quals + off[i + 1])});
        msa.push_back(std::string(rows + (size_t)i * width, width));
    }
    fix_msa_ends(rs, msa);
    for (uint32_t i = 0; i < n; ++i) {
        memcpy(rows + (size_t)i * width, msa[i].data(), width);
        memcpy(seqs + off[i], rs[i].seq.data(), rs[i].seq.size());
        memcpy(quals + off[i], rs[i].quality.data(), rs[i].quality.size());
        len_out[i] = (uint32_t)rs[i].seq.size();
    }
}

void orc_free(void *p) { free(p); }

// AVX2 int16 row fill of the POA matrices (orc_poa.hpp); returns whether this CPU can run it.
int orc_set_poa_simd(int on) {
    poa_simd_default() = on != 0;
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx2") ? 1 : 0;
#else
    return 0;
#endif
}

// Column-vote tie-break order (6 symbols), see orc_correct.hpp.
void orc_set_cv_order(const char *o) { set_cv_order(o); }

}  // extern "C"
