// ORACLE (test infrastructure, never shipped in the product path).
// Command-line face of the restatement, mirroring main.cpp:133-324 (`cluster`) and
// :325-412 (`correct`) closely enough to diff output files against the product CLI.
#include <cstring>
#include <iostream>
#include <map>

#include "orc_cluster.hpp"
#include "orc_correct.hpp"
#include "orc_io.hpp"

using namespace orc;

static const char *opt(int argc, char **argv, const char *name, const char *def) {
    for (int i = 2; i + 1 < argc; ++i) if (!strcmp(argv[i], name)) return argv[i + 1];
    return def;
}
static bool flag(int argc, char **argv, const char *name) {
    for (int i = 2; i < argc; ++i) if (!strcmp(argv[i], name)) return true;
    return false;
}

int main(int argc, char **argv) {
    if (argc < 2) { std::cerr << "oracle_cli <cluster|correct> ...\n"; return 1; }
    std::string mode = argv[1];
    if (mode == "cluster") {
        cluster_params_t P;
        P.k = atoi(opt(argc, argv, "-k", "10"));
        P.t_s = atof(opt(argc, argv, "-s", "0.2"));
        P.t_v = atof(opt(argc, argv, "-v", "1000000"));
        P.iso_k = atoi(opt(argc, argv, "--iso-kmer-size", "11"));
        P.iso_t_s = atof(opt(argc, argv, "--iso-score-threshold", "0.3"));
        P.iso_t_v = atof(opt(argc, argv, "--iso-max-variance", "25"));
        P.bv_threshold = atof(opt(argc, argv, "-B", "0.4"));
        P.bv_min_threshold = atof(opt(argc, argv, "-b", "0.2"));
        P.bv_falloff = atof(opt(argc, argv, "-f", "0.05"));
        P.repr_percentile = atof(opt(argc, argv, "-p", "0.15"));
        P.is_rna = flag(argc, argv, "--rna");
        P.iso = flag(argc, argv, "--iso");
        int lower = atoi(opt(argc, argv, "--lower-length", "150")), upper = atoi(opt(argc, argv, "--upper-length", "100000"));
        bool raw = flag(argc, argv, "--raw");
        read_set_t reads = read_multiple_inputs_cluster(split_commas(opt(argc, argv, "-i", "")), split_commas(opt(argc, argv, "-l", "")), raw, lower, upper);
        std::cout << "Reads: " << reads.size() << std::endl;
        work_counters_t wc;
        cluster_set_t cs = cluster_command(reads, P, &wc);
        std::string out = std::string(opt(argc, argv, "-o", ".")) + "/clusters.out";
        std::ofstream f(out, std::ofstream::binary);
        std::string b = hps_encode(cs);
        f.write(b.data(), b.size());
        std::cerr << cs.size() << " clusters; pair_tests=" << wc.pair_tests << " full_cmp=" << wc.full_cmp
                  << " matches=" << wc.matches << std::endl;
        return 0;
    }
    if (mode == "correct") {
        if (strlen(opt(argc, argv, "--cv-order", "")) == 6) set_cv_order(opt(argc, argv, "--cv-order", ""));
        const std::vector<std::string> labels = split_commas(opt(argc, argv, "-l", ""));
        read_set_t reads = read_multiple_inputs(split_commas(opt(argc, argv, "-i", "")), labels);
        cluster_set_t cs = hps_read_file(opt(argc, argv, "-c", ""));
        correct_counters_t cc;
        correction_results_t R = correct_reads(cs, reads, atof(opt(argc, argv, "-m", "0.3")), atof(opt(argc, argv, "-g", "0.3")),
                                               30.0, atoi(opt(argc, argv, "-s", "200")), atoi(opt(argc, argv, "-r", "5")), labels, &cc);
        std::string o = opt(argc, argv, "-o", ".");
        write_fastq(R.corrected, o + "/corrected.fq");
        write_fastq(R.uncorrected, o + "/uncorrected.fq");
        write_fastq(R.consensi, o + "/consensi.fq");
        std::cerr << "packs=" << cc.packs << " alignments=" << cc.alignments << " dp_cells=" << cc.dp_cells << std::endl;
        return 0;
    }
    if (mode == "polish") {                                       // main.cpp:612-762
        read_set_t reads = read_fastq_plain(opt(argc, argv, "-i", ""), "");
        sort_read_set(reads);
        bool is_rna = flag(argc, argv, "--rna");
        cluster_set_t clusters = cluster_reads(reads, 6, 0.5, 25, 0.4, 0.4, 0.05, 0, false, 0.15, is_rna);
        correction_results_t R = correct_reads(clusters, reads, 0.3, 0.3, 30.0, 200, 0, {});
        int cid = 0;
        std::map<int, int> gene_map;
        int gid = -1;
        for (auto &r : R.consensi) {
            int total_reads = 0;
            for (auto &m : clusters[cid].seqs) {
                const std::string &h = reads[m.seq_id].header;     // already carries the ",gene_cluster_x" suffix: irrelevant for '=' / '_' fields used below
                auto info = split_string(h, '=');
                total_reads += std::stoi(info[1]);
                auto info_c = split_string(h, '_');
                if (h.find("transcript_cluster") != std::string::npos) {
                    int id = std::stoi(info_c[4]);
                    if (gene_map.find(id) == gene_map.end()) { if (gid == -1) gid = id; gene_map.insert({id, gid}); }
                    else gid = gene_map.find(id)->second;
                }
            }
            int rcount = std::stoi(split_string(r.header, '=')[1]);
            if (gid != -1)
                r.header = "@transcript_cluster_" + std::to_string(cid) + " gene_cluster_" + std::to_string(gid) +
                           " generated_from_transcript_clusters=" + std::to_string(rcount) + " total_reads=" + std::to_string(total_reads) + " labels=";
            else
                r.header = "@cluster_" + std::to_string(cid) + " generated_from_consensi_clusters=" + std::to_string(rcount) +
                           " total_reads=" + std::to_string(total_reads) + " labels=";
            ++cid;
            gid = -1;
        }
        write_fastq(R.consensi, std::string(opt(argc, argv, "-o", ".")) + "/transcriptome.fq");
        return 0;
    }
    if (mode == "dump-reads") {                                   // tests: the restated readers against the real fasta.cpp (oracle/_ref)
        int kind = atoi(opt(argc, argv, "--kind", "0")), index = atoi(opt(argc, argv, "--index", "0"));
        const std::string path = opt(argc, argv, "-i", ""), label = opt(argc, argv, "-l", "");
        bool raw = flag(argc, argv, "--raw");
        int lower = atoi(opt(argc, argv, "--lower-length", "150")), upper = atoi(opt(argc, argv, "--upper-length", "100000"));
        read_set_t rs = kind == 0 ? read_fastq_plain(path, label) : kind == 1 ? read_fastq_cluster(path, label, index, raw, lower, upper)
                      : kind == 2 ? read_fasta_plain(path, label) : read_fasta_cluster(path, label, index, raw, lower, upper);
        std::ofstream f(opt(argc, argv, "-o", "/dev/stdout"), std::ofstream::binary);
        for (auto &r : rs) f << r.header << "\t" << r.seq << "\t" << r.ann << "\t" << r.quality << "\n";
        std::cout << ((kind == 1 || kind == 3) ? index : 0) << std::endl;
        return 0;
    }
    std::cerr << "unknown mode\n";
    return 1;
}
