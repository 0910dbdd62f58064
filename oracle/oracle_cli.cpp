// ORACLE (test infrastructure, never shipped in the product path).
// Command-line face of the restatement, mirroring main.cpp:133-324 (`cluster`) and
// :325-412 (`correct`) closely enough to diff output files against the product CLI.
#include <cstring>
#include <iostream>

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
        int index = 0;
        read_set_t reads = read_fastq_cluster(opt(argc, argv, "-i", ""), "", index, raw, lower, upper);
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
        read_set_t reads = read_fastq_plain(opt(argc, argv, "-i", ""), "");
        cluster_set_t cs = hps_read_file(opt(argc, argv, "-c", ""));
        correct_counters_t cc;
        correction_results_t R = correct_reads(cs, reads, atof(opt(argc, argv, "-m", "0.3")), atof(opt(argc, argv, "-g", "0.3")),
                                               30.0, atoi(opt(argc, argv, "-s", "200")), atoi(opt(argc, argv, "-r", "5")), {}, &cc);
        std::string o = opt(argc, argv, "-o", ".");
        write_fastq(R.corrected, o + "/corrected.fq");
        write_fastq(R.uncorrected, o + "/uncorrected.fq");
        write_fastq(R.consensi, o + "/consensi.fq");
        std::cerr << "packs=" << cc.packs << " alignments=" << cc.alignments << " dp_cells=" << cc.dp_cells << std::endl;
        return 0;
    }
    std::cerr << "unknown mode\n";
    return 1;
}
