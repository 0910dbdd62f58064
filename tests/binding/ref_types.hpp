// Test scaffolding: the DECLARATIONS the binding needs from the reference's headers, restated (no reference
// header is included and no reference function body is used): read_t / read_set_t (/root/reference/fasta.hpp:7-14),
// cseq_t / cluster_t / cluster_set_t (/root/reference/cluster.hpp:10-41, hps hooks omitted), correction_results_t
// (/root/reference/correct.hpp:32-36).
#pragma once
#include <string>
#include <vector>

struct read_t { std::string header, seq, ann, quality; };
typedef std::vector<read_t> read_set_t;
struct cseq_t { int seq_id; bool rev; int gene_id = -1; };
struct cluster_t { cseq_t main_seq; std::vector<cseq_t> seqs; };
typedef std::vector<cluster_t> cluster_set_t;
struct correction_results_t { read_set_t corrected, uncorrected, consensi; };
