// Test driver around integration/rattle_binding.hpp: the call sequence of main.cpp:254-277 (`cluster`) and :386-408
// (`correct`) through the two bound functions, printing what the reference would write.
//   driver <reads.fastq> <rna 0|1> <split> <out_prefix>
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>

#include "ref_types.hpp"
#include "../../integration/rattle_binding.hpp"

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const bool is_rna = atoi(argv[2]) != 0;
    const int split = atoi(argv[3]);
    const std::string prefix = argv[4];
    read_set_t file_reads;
    {
        std::ifstream in(argv[1]);
        std::string h, s, a, q;
        while (std::getline(in, h) && std::getline(in, s) && std::getline(in, a) && std::getline(in, q)) file_reads.push_back(read_t{h, s, a, q});
    }
    try {
        // main.cpp:245-277: ann = record index, stable length sort, cluster, translate ids back
        read_set_t reads = file_reads;
        for (size_t i = 0; i < reads.size(); ++i) { reads[i].ann = std::to_string(i); reads[i].quality.clear(); }
        std::stable_sort(reads.begin(), reads.end(), [](const read_t &x, const read_t &y) { return x.seq.size() > y.seq.size(); });
        cluster_set_t clusters = cluster_reads(reads, 10, 0.2, 1000000, 0.4, 0.2, 0.05, 0, false, 0.15, is_rna, false, 1);
        for (auto &c : clusters) {
            c.main_seq.seq_id = std::stoi(reads[c.main_seq.seq_id].ann);
            for (auto &s : c.seqs) s.seq_id = std::stoi(reads[s.seq_id].ann);
        }
        {
            std::ofstream f(prefix + ".clusters.txt");
            for (auto &c : clusters) {
                f << c.main_seq.seq_id << ":" << c.main_seq.rev << " |";
                for (auto &s : c.seqs) f << " " << s.seq_id << ":" << s.rev;
                f << "\n";
            }
        }
        // main.cpp:386-408
        correction_results_t R = correct_reads(clusters, file_reads, 0.3, 0.3, 30.0, split, 5, 1, false, {});
        auto dump = [&](const read_set_t &v, const std::string &name) {
            std::ofstream f(prefix + "." + name);
            for (auto &r : v) f << r.header << "\n" << r.seq << "\n" << r.ann << "\n" << r.quality << "\n";
        };
        dump(R.corrected, "corrected.fq"); dump(R.uncorrected, "uncorrected.fq"); dump(R.consensi, "consensi.fq");
        rattle_binding::shutdown();
    } catch (const std::exception &e) {
        std::cerr << "binding driver: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
