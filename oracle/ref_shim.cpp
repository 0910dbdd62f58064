// ORACLE (test infrastructure).  extern "C" shim over the REAL reference translation
// units that compile from their own sources with no stand-ins (kmer.cpp, similarity.cpp,
// utils.cpp, fasta.cpp).  Built by `make ref` into oracle/_ref/libref.so from the sources
// where they lie under /root/reference; never copied into the repo.  Used to validate
// the restatement in orc_cluster.hpp (tests/test_oracle_vs_ref.py) and, on the GPU box
// (where the prebuilt .so travels), as an extra checker.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "fasta.hpp"
#include "kmer.hpp"
#include "similarity.hpp"
#include "utils.hpp"

extern "C" {

void ref_extract_kmers(const char *seq, uint32_t len, int k, int both, uint32_t *fwd_hash, int32_t *fwd_pos,
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

void ref_pair_score(const char *a, uint32_t la, const char *b, uint32_t lb, int k, int strand, int32_t *bases,
                    int32_t *hc_bases, int32_t *n_dist, double *variance, int32_t *n_matches, int32_t *dist_out,
                    int32_t dist_cap) {
    read_kmers_t ka = extract_kmers_from_read(std::string(a, la), k, false);
    read_kmers_t kb = extract_kmers_from_read(std::string(b, lb), k, true);
    auto common = get_common_kmers(ka.list_forward, strand ? kb.list_reverse : kb.list_forward);
    auto sim = calc_similarity(common, k);
    *bases = sim.bases; *n_dist = (int32_t)sim.distances.size();
    *hc_bases = common.empty() ? 0 : sim.hc_bases;     // hc_bases is uninitialised for an empty LIS (similarity.cpp:33-35)
    *variance = var(sim.distances);
    *n_matches = (int32_t)common.size();
    for (int i = 0; i < (int)sim.distances.size() && i < dist_cap; ++i) dist_out[i] = sim.distances[i];
}

double ref_var(const int32_t *v, uint32_t n) { return var(std::vector<int>(v, v + n)); }
double ref_phred_err(char c) { return phred_err(c); }
char ref_phred_symbol(double p) { return phred_symbol(p); }
void ref_reverse_complement(const char *s, uint32_t n, char *out) {
    std::string r = reverse_complement(std::string(s, n));
    memcpy(out, r.data(), n);
}

// read_fastq_file (cluster variant, fasta.cpp:272-370) + sort_read_set (fasta.cpp:462-464):
// returns the number of surviving reads; ann (original index) per read in processing order.
int32_t ref_read_fastq_cluster(const char *path, int raw, int lower, int upper, int32_t *ann_out, int32_t *len_out, int32_t cap) {
    read_set_t rs = read_fastq_file(std::string(path), std::string(""), 0, raw != 0, lower, upper);
    sort_read_set(rs);
    int32_t n = (int32_t)rs.size();
    for (int32_t i = 0; i < n && i < cap; ++i) { ann_out[i] = std::stoi(rs[i].ann); len_out[i] = (int32_t)rs[i].seq.size(); }
    return n;
}


// Any of the four readers of fasta.cpp dumped as text, one record per line: header \t seq \t ann \t quality.
// kind: 0 FASTQ plain (:207-270), 1 FASTQ cluster (:272-370), 2 FASTA plain (:33-96), 3 FASTA cluster (:98-205).
// Returns the next free record index of the cluster variants (smuggled through back().quality, main.cpp:47), else 0.
int32_t ref_dump_reads(const char *path, const char *label, int kind, int index, int raw, int lower, int upper, const char *out_path) {
    read_set_t rs;
    int32_t next = 0;
    if (kind == 0) rs = read_fastq_file(std::string(path), std::string(label));
    else if (kind == 1) rs = read_fastq_file(std::string(path), std::string(label), index, raw != 0, lower, upper);
    else if (kind == 2) rs = read_fasta_file(std::string(path), std::string(label));
    else rs = read_fasta_file(std::string(path), std::string(label), index, raw != 0, lower, upper);
    if ((kind == 1 || kind == 3) && !rs.empty()) { next = std::stoi(rs.back().quality); rs.back().quality = ""; }
    FILE *f = fopen(out_path, "wb");
    for (auto &r : rs) fprintf(f, "%s\t%s\t%s\t%s\n", r.header.c_str(), r.seq.c_str(), r.ann.c_str(), r.quality.c_str());
    fclose(f);
    return next;
}

}  // extern "C"
