/*
 * rattle_hip.h -- C ABI of librattle_hip.so, the MI355X (gfx950) implementation of
 * RATTLE's `cluster` + `correct` hot path.
 *
 * RATTLE has no FFI / plugin interface: the seams are the C++ functions
 *   cluster_reads(...)   /root/reference/cluster.hpp:44   (called at main.cpp:258,300,669)
 *   correct_reads(...)   /root/reference/correct.hpp:44   (called at main.cpp:405,670)
 * This header is the C boundary a maintainer binds underneath those two functions
 * (INTEGRATION.md shows the binding).  Plain pointers and sizes only; no C++ or torch
 * types.  All functions return 0 on success and a negative code on error, with text
 * available from rattle_hip_last_error().  Nothing throws across the boundary.
 * Input buffers are caller-owned; result objects are library-owned and released with
 * the matching *_free function.  One host thread drives one context.
 *
 * Every entry point REQUIRES a HIP device: there is no CPU fallback.
 */
#ifndef RATTLE_HIP_H
#define RATTLE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RATTLE_OK 0
#define RATTLE_ERR_HIP (-1)       /* a HIP runtime call failed (no device, OOM, launch error) */
#define RATTLE_ERR_ARG (-2)       /* invalid argument (bad base in a read, k out of range, ...) */
#define RATTLE_ERR_STATE (-3)     /* call order violated (e.g. no reads loaded) */

typedef struct rattle_ctx rattle_ctx;

const char *rattle_hip_last_error(void);
int rattle_hip_abi_version(void);

/* Context = one HIP device + its streams and device-resident read index. */
int rattle_hip_ctx_create(int device, rattle_ctx **out);
/* A context WITHOUT a device, for a coordinating process and for CPU tests of the sharding: only
 * rattle_hip_set_exchange and rattle_hip_correction_gather work on it; every compute entry point returns
 * RATTLE_ERR_STATE (there is still no CPU path for the kernels). */
int rattle_hip_ctx_create_host(rattle_ctx **out);
void rattle_hip_ctx_destroy(rattle_ctx *ctx);

/* ------------------------------------------------------------------------------------
 * a3  extract_kmers_from_read   /root/reference/kmer.cpp:6-42, kmer.hpp:25-40
 * Uploads n reads (ASCII A/C/G/T/U, concatenated; offsets has n+1 entries) and builds on
 * the device, per read: the 4096-bit 6-mer bit-vector(s) (kmer.hpp:14-16), and the
 * (hash,pos)-sorted k-mer list(s) with L-k entries.  both_strands = !is_rna.
 * Replaces any previously loaded read set.  Reads must be given in PROCESSING order
 * (cluster_reads consumes them length-descending, main.cpp:254).
 */
int rattle_hip_load_reads(rattle_ctx *ctx, const uint8_t *seq_concat, const uint64_t *offsets, uint32_t n_reads,
                          int kmer_size, int both_strands);

/* Read back one read's index (tests): hash/pos need max(L-k,0) entries, bv 64 words. */
int rattle_hip_get_read_index(rattle_ctx *ctx, uint32_t read, int strand, uint32_t *hash_out, int32_t *pos_out,
                              uint64_t *bv_out, uint32_t *bv_popcount);

/* ------------------------------------------------------------------------------------
 * a4  bit-vector filter of cluster_together   /root/reference/cluster.cpp:13-19,43
 * For every (seed s, candidate c) with c >= first_cand[s]:
 *   common_f = popcount(bv_f[seed] & bv_f[cand]); common_r = popcount(bv_f[seed] & bv_r[cand])
 *   mmax = max(popcount(bv_f[seed]), popcount(bv_f[cand]))
 *   bit0 = fwd_bypass || common_f >= min_common_lut[mmax]
 *   bit1 = both_strands && common_r >= min_common_lut[mmax]
 * min_common_lut[m] (4097 entries) is the smallest c with double(c)/double(m) >= thr
 * evaluated by the caller in the reference's own double arithmetic (0xFFFF = never).
 * out_pass is n_seeds*n_cands bytes, row-major by seed.
 */
int rattle_hip_bv_filter(rattle_ctx *ctx, const uint32_t *seed_ids, uint32_t n_seeds, const uint32_t *cand_ids,
                         uint32_t n_cands, const uint32_t *first_cand, const uint16_t *min_common_lut, int fwd_bypass,
                         uint8_t *out_pass);

/* ------------------------------------------------------------------------------------
 * a5+a6+a7  get_common_kmers + calc_similarity + var
 *           /root/reference/kmer.cpp:45-67, similarity.cpp:4-97, utils.cpp:36-55
 * For each pair p: read i_ids[p] (forward list) against read j_ids[p] (forward list if
 * strand[p]==0, reverse-complement list otherwise).  Outputs per pair: bases, hc_bases,
 * n_dist = distances.size(), variance = var(distances) (IEEE double, same operation order
 * as the reference; NaN for n_dist==1), n_matches = common.size().
 */
int rattle_hip_pair_score(rattle_ctx *ctx, const uint32_t *i_ids, const uint32_t *j_ids, const uint8_t *strand,
                          uint32_t n_pairs, int32_t *bases, int32_t *hc_bases, int32_t *n_dist, double *variance,
                          int32_t *n_matches);

/* ------------------------------------------------------------------------------------
 * a8-a11  cluster_reads   /root/reference/cluster.cpp:93-259 (signature cluster.hpp:44)
 * Runs on the reads loaded by rattle_hip_load_reads (kmer_size / strands taken from there).
 * min_reads_cluster, use_hc and verbose of the reference signature are accepted and
 * ignored exactly as the reference ignores / never sets them (cluster.cpp:242-243).
 */
typedef struct {
    double t_s, t_v;
    double bv_threshold, min_bv_threshold, bv_falloff;
    int min_reads_cluster;
    int use_hc;
    double repr_percentile;
    int is_rna;
} rattle_cluster_params;

typedef struct {
    uint32_t n_clusters;
    int32_t *main_id;       /* [n_clusters]   cluster_t::main_seq.seq_id (index into the loaded reads) */
    uint8_t *main_rev;      /* [n_clusters]   cluster_t::main_seq.rev */
    uint32_t *offsets;      /* [n_clusters+1] member range of each cluster */
    int32_t *member_id;     /* [n_members]    cseq_t::seq_id, in the order cluster_t::seqs holds them */
    uint8_t *member_rev;    /* [n_members]    cseq_t::rev */
    /* work counters.  Exact and identical on every rank of a sharded job: [0] bit-vector pair tests, [1] full
       comparisons (pairs past the filter: their common k-mers are counted).  [2] k-mer matches summed over the count pass:
       exact when the pass counts per pair, an UPPER BOUND when the seed-major count kernel folds the hash (k > 10) or a
       seed's repeat list overflows -- it depends on which form the driver picked and is for reporting only.  Local to the
       calling rank: [3] seed rounds, [4] kernel launches, [5] pairs whose match count could still reach t_s and went
       through the patience search (the others are rejected exactly on the count). */
    uint64_t counters[8];
    int32_t *gene_id;       /* [n_clusters] cseq_t::gene_id of the --iso flow (index of the gene cluster), else NULL */
} rattle_cluster_set;

/* Keep a copy of the reads (and, if not NULL, their qualities) in HBM.  Later calls of
 * rattle_hip_cluster_unsorted / rattle_hip_correct_reads that are handed the SAME host buffers
 * (pointer, read count, total bases) use the resident copy instead of uploading the bases again;
 * the host buffers must stay unchanged until rattle_hip_unstage_reads.  This is how a caller that
 * runs `cluster` and `correct` on one read set (main.cpp reads the same fastq twice) pays the PCIe
 * transfer once. */
int rattle_hip_stage_reads(rattle_ctx *ctx, const uint8_t *seq_concat, const uint8_t *qual_concat, const uint64_t *offsets,
                           uint32_t n_reads);
int rattle_hip_unstage_reads(rattle_ctx *ctx);

int rattle_hip_cluster_reads(rattle_ctx *ctx, const rattle_cluster_params *params, rattle_cluster_set **out);
/* Same, restricted to a subset of the loaded reads given in processing order (the --iso
 * second level, main.cpp:281-318).  Ids in the result are positions in `subset`. */
int rattle_hip_cluster_subset(rattle_ctx *ctx, const rattle_cluster_params *params, const uint32_t *subset,
                              uint32_t n_subset, rattle_cluster_set **out);
/* Several subsets in one call (all gene clusters of the --iso level): subset i is
 * subset_ids[subset_offsets[i] .. subset_offsets[i+1]); outs[i] receives its clusters.  The subsets
 * are independent: their greedy rounds advance in lockstep and every round is one evaluation on the
 * device (n_workers is kept for ABI compatibility and ignored).  Results are identical to n_subsets
 * calls of rattle_hip_cluster_subset. */
int rattle_hip_cluster_subsets(rattle_ctx *ctx, const rattle_cluster_params *params, const uint32_t *subset_ids,
                               const uint64_t *subset_offsets, uint32_t n_subsets, rattle_cluster_set **outs, int n_workers);
/* The `rattle cluster` flow around cluster_reads for reads in FILE order (main.cpp:254-277):
 * stable length-descending sort (sort_read_set, fasta.cpp:458-464), index, gene-level
 * cluster_reads, then ids translated back to positions in the caller's order. */
int rattle_hip_cluster_unsorted(rattle_ctx *ctx, const uint8_t *seq_concat, const uint64_t *offsets, uint32_t n_reads,
                                int kmer_size, const rattle_cluster_params *params, rattle_cluster_set **out);
/* `rattle cluster --iso` (main.cpp:254-323) for reads in FILE order: gene level with (kmer_size, params), every
 * gene cluster sub-clustered with (iso_kmer_size, iso_params); result = the transcript clusters in gene order
 * with gene_id set, ids = positions in the caller's order.  n_gene_clusters (optional) receives the first level's count. */
int rattle_hip_cluster_iso_unsorted(rattle_ctx *ctx, const uint8_t *seq_concat, const uint64_t *offsets, uint32_t n_reads,
                                    int kmer_size, int iso_kmer_size, const rattle_cluster_params *params,
                                    const rattle_cluster_params *iso_params, rattle_cluster_set **out, uint32_t *n_gene_clusters);
void rattle_hip_cluster_set_free(rattle_cluster_set *cs);

/* ------------------------------------------------------------------------------------
 * a15  spoa engine + graph as called from /root/reference/correct.cpp:395-405,428-436,520-532
 * (createAlignmentEngine(kSW,5,-4,-8,-6); align + add_alignment per sequence;
 * generate_multiple_sequence_alignment).  Packs are independent; pack p holds sequences
 * [pack_first[p], pack_first[p+1]) of the concatenated input, aligned in that order.
 * Result: for each pack its MSA width and rows (one row per sequence, '-' for gaps).
 */
typedef struct {
    uint32_t n_packs;
    uint32_t *width;        /* [n_packs]  MSA columns */
    uint64_t *row_offset;   /* [n_seqs+1] byte offset of each sequence's row in `rows` */
    char *rows;             /* rows of all packs back to back */
    uint64_t counters[8];   /* DP cells, alignments, graph nodes (sum of final sizes), rows, DP cells computed, certified bands, failed band certificates */
} rattle_msa_set;

int rattle_hip_poa_msa(rattle_ctx *ctx, const uint8_t *seq_concat, const uint64_t *offsets, uint32_t n_seqs,
                       const uint32_t *pack_first, uint32_t n_packs, rattle_msa_set **out);
void rattle_hip_msa_set_free(rattle_msa_set *ms);


/* ------------------------------------------------------------------------------------
 * a14-a20  correct_reads   /root/reference/correct.cpp:311-563 (signature correct.hpp:44)
 * Pack building (strided split, in-place reverse complement of `rev` members), POA #1 over
 * the raw reads of every pack, fix_msa_ends, column vote + per-read correction, POA #2 over
 * the corrected reads, pack consensus, POA #3 over the pack consensi of multi-pack clusters.
 * All POAs of one stage run in a single device pass (kernel C); the post-MSA logic
 * (correct.cpp:32-309) is kernel D on the device, in the reference's double arithmetic and
 * accumulation order.
 * Reads are given in FILE order with qualities (main.cpp:386); clusters index them by seq_id.
 * Outputs follow the reference's single-thread order: packs in queue order, pack consensi
 * collected in pack order (the reference's multi-thread run uses completion order).
 * Headers are not handled here: the caller derives them from read_id / cluster_id / gene_id.
 */
struct rattle_read_set_s;      /* rattle_read_set, defined below */
typedef struct {
    double min_occ, gap_occ, err_ratio;   /* 0.3, 0.3, 30.0 */
    int split, min_reads;                 /* 200, 5 */
    int n_threads;                        /* host threads for pack planning / output assembly; 0 = all cores */
    char vote_order[8];                   /* column-vote tie order, 6 symbols; "" = "U-GTCA", the
                                             iteration order of the reference's unordered_map
                                             (correct.cpp:105-110,174) under libstdc++ */
    /* Order in which a multi-pack cluster's pack consensi enter POA #3 (correct.cpp:469,520-526).  The
     * reference collects them in worker COMPLETION order when n_threads > 1; the default here is pack
     * order (the reference's single-thread order).  Optional override for n_pack_orders clusters:
     * cluster pack_order_cluster[i] takes its packs in the order
     * pack_order_perm[pack_order_offsets[i] .. pack_order_offsets[i+1]) -- a permutation of 0..np-1 over
     * the cluster's packs that passed the min_reads test, in queue order. */
    uint32_t n_pack_orders;
    const uint32_t *pack_order_cluster;
    const uint32_t *pack_order_offsets;
    const uint32_t *pack_order_perm;
    /* A pack whose POA does not fit the device (the DP record of one alignment outgrows the HBM left for
     * the arena; 100 kb reads need > 10^10 cells per alignment in the reference as well) is SKIPPED, not
     * fatal: its reads go to `uncorrected` untouched, it contributes no consensus, and it is listed in
     * rattle_correction::skipped.  max_pack_cells > 0 additionally skips, before any device work, every
     * pack whose largest alignment would need more than that many DP cells by the bound
     * (6 * longest read + 64) * longest read; 0 = no such limit. */
    uint64_t max_pack_cells;
    /* Optional (ABI 3; NULL = off): called ONCE, from a thread of the library, as soon as this call's corrected reads are
     * complete in host memory -- the consensus stages (POA #2 / #3) still run on the device, so a caller can format and
     * write corrected.fq meanwhile (the reference writes its three files after correct_reads returns, main.cpp:405-408).
     * `corrected` is the set the call will return in rattle_correction::corrected (same pointers, read-only here),
     * `corrected_pack` its per-record pack index.  Not called when the call fails before that point or has no pack. */
    void (*corrected_ready)(void *user, const struct rattle_read_set_s *corrected, const uint32_t *corrected_pack);
    void *corrected_ready_user;
} rattle_correct_params;

typedef struct rattle_read_set_s {
    uint32_t n;
    int32_t *read_id;       /* original read index, or -1 for a consensus */
    int32_t *cluster_id;    /* index of the cluster in the input cluster set */
    int32_t *n_reads;       /* consensi: reads summed over the cluster's packs; else 0 */
    uint64_t *off;          /* [n+1] */
    char *seq;
    char *qual;
} rattle_read_set;

/* Packs (stage 1, 2: POA #1 / #2 of a pack) or clusters (stage 3: POA #3) that were not processed. */
typedef struct {
    uint32_t n;
    int32_t *cluster_id;    /* [n] */
    uint32_t *pack;         /* [n] index of the pack among the cluster's queued packs (stage 3: 0) */
    uint32_t *stage;        /* [n] 1: POA #1 did not fit (reads -> uncorrected), 2: POA #2 (no pack consensus),
                                   3: POA #3 (no cluster consensus), 0: max_pack_cells rule */
    uint64_t *read_off;     /* [n+1] range of the entry's reads in read_id */
    int32_t *read_id;       /* original read indices */
} rattle_skip_list;

typedef struct {
    rattle_read_set corrected, uncorrected, consensi;
    uint64_t counters[8];   /* [0] DP cells (the reference's count: graph rows x sequence columns of every alignment), [1] alignments, [2] packs queued,
                             * [3] packs skipped, [4] reads in skipped packs, [5] DP cells the device computed (fewer where the exact band for
                             * near-identical sequences was certified), [6] alignments with a certified band, [7] failed band certificates */
    rattle_skip_list skipped;
    /* global pack index of every corrected / uncorrected record (0xFFFFFFFF: member of a pack that never
     * entered the queue); what rattle_hip_correction_gather orders the merged result by */
    uint32_t *corrected_pack, *uncorrected_pack;
} rattle_correction;

/* (A context that clustered its reads holds their k-mer index, 12-20 bytes per base.  `correct` does not use it and sizes its arena by
 * the free memory: when the index exceeds 24 GB it is released here, and the next rattle_hip_cluster_reads without a new
 * rattle_hip_load_reads returns RATTLE_ERR_STATE "no reads loaded".  rattle_hip_cluster_unsorted loads by itself and is unaffected.) */
int rattle_hip_correct_reads(rattle_ctx *ctx, const uint8_t *seq_concat, const uint8_t *qual_concat,
                             const uint64_t *offsets, uint32_t n_reads, uint32_t n_clusters,
                             const uint32_t *cluster_offsets, const int32_t *member_id, const uint8_t *member_rev,
                             const rattle_correct_params *params, rattle_correction **out);
void rattle_hip_correction_free(rattle_correction *c);
/* Optional: allocate the POA arena of correct_reads ahead of time (a caller can overlap the seconds a > 100 GB allocation takes
 * with reading its input).  bytes is a hint, clamped to what the device has free; correct_reads grows the arena if it must. */
int rattle_hip_reserve_arena(rattle_ctx *ctx, uint64_t bytes);

/* ------------------------------------------------------------------------------------
 * One job over the GPUs of a node.  The reference parallelises the same axes with host
 * threads: the strided candidate loop of a seed (cluster.cpp:138-158,189-209), the gene clusters of the
 * --iso level (main.cpp:281-318) and the pack queue of `correct` (correct.cpp:377-392).  Here a context
 * can be one RANK of nranks (one process or thread per GPU, every rank holding the same reads and making
 * the same calls):
 *   cluster_reads / cluster_unsorted : each rank scores its share of the candidates of a seed batch
 *                                      (cyclic by candidate position), hit lists are all-gathered, every
 *                                      rank resolves them identically -> identical results on all ranks;
 *   cluster_subsets                  : gene clusters LPT-assigned to ranks by size, results all-gathered;
 *   correct_reads                    : packs LPT-assigned by the cost proxy L_0 * sum L_j; both POAs of a
 *                                      pack on one GPU; pack consensi all-gathered; POA #3 groups
 *                                      LPT-assigned; every rank returns ITS packs' corrected / uncorrected
 *                                      reads and ALL consensi; rattle_hip_correction_gather reassembles
 *                                      the single-GPU result (same order, same bytes) on the root.
 * Transports: RCCL over xGMI (rattle_hip_comm_init; device buffers, the library's own stream) or a
 * caller-supplied all-gather on host buffers (rattle_hip_set_exchange; MPI / gloo / tests).
 */
/* all ranks call with their `send` (send_bytes == recv_bytes[rank]); recv receives the nranks pieces
 * back to back in rank order.  Returns 0 on success. */
typedef int (*rattle_allgatherv_fn)(void *user, const void *send, uint64_t send_bytes, void *recv, const uint64_t *recv_bytes);
int rattle_hip_set_exchange(rattle_ctx *ctx, int rank, int nranks, rattle_allgatherv_fn fn, void *user);
#define RATTLE_COMM_ID_BYTES 128
int rattle_hip_comm_unique_id(uint8_t *id_out /* RATTLE_COMM_ID_BYTES, from rank 0, to be handed to every rank */);
int rattle_hip_comm_init(rattle_ctx *ctx, int rank, int nranks, const uint8_t *id);
int rattle_hip_comm_destroy(rattle_ctx *ctx);
/* statistics of the exchange since context creation: collective calls, payload bytes received */
int rattle_hip_comm_stats(rattle_ctx *ctx, uint64_t *calls, uint64_t *bytes);
/* Collective self-test of the attached transport (every rank calls it): one ragged all-gather-v and one gather
 * to the last rank, contents verified.  0, or RATTLE_ERR_HIP with the damaged piece named.  A launcher runs it
 * once after rattle_hip_comm_init / rattle_hip_set_exchange, before it trusts the transport with a job. */
int rattle_hip_comm_probe(rattle_ctx *ctx);
/* Collective: merges the per-rank results of rattle_hip_correct_reads on `root` (*merged is NULL on the
 * other ranks).  With nranks == 1 it returns a copy. */
int rattle_hip_correction_gather(rattle_ctx *ctx, const rattle_correction *local, int root, rattle_correction **merged);

/* The work list of correct_reads without a device (pack building correct.cpp:328-370 and the static
 * assignment to ranks), for callers that schedule themselves and for the CPU tests of the sharding. */
typedef struct {
    uint32_t n_packs;
    uint32_t *pack_first;   /* [n_packs+1] member range */
    int32_t *member_id;     /* read ids, pack after pack */
    uint8_t *member_rev;
    int32_t *pack_cluster;  /* [n_packs] */
    uint32_t *pack_local;   /* [n_packs] index among the cluster's queued packs */
    uint64_t *pack_cost;    /* [n_packs] L_0 * sum L_j */
    uint32_t *pack_owner;   /* [n_packs] rank */
    uint32_t n_unqueued;    /* members of packs that never enter the queue (<= min_reads, or max_pack_cells) */
    int32_t *unqueued_id;
    int32_t *unqueued_cluster;
} rattle_pack_plan;
int rattle_hip_plan_packs(const uint64_t *offsets, uint32_t n_reads, uint32_t n_clusters, const uint32_t *cluster_offsets,
                          const int32_t *member_id, const uint8_t *member_rev, const rattle_correct_params *params, int nranks,
                          rattle_pack_plan **out);
void rattle_hip_pack_plan_free(rattle_pack_plan *p);
/* longest-processing-time-first: items by cost descending (ties: lower index) to the least loaded rank (ties: lower rank) */
int rattle_hip_lpt_assign(const uint64_t *cost, uint32_t n, int nranks, uint32_t *owner_out);

/* Test hook, needs no device: phred_symbol's value `-10*log10(p)+33` (utils.cpp:6-8, before the narrowing to
 * char) through the threshold table the post-MSA kernel bisects and through the host libm.  Returns the number
 * of explicit exceptions the table carries. */
int rattle_hip_debug_phred_symbol(double p, int *table_value, int *libm_value);

/* ------------------------------------------------------------------------------------
 * Per-kernel timing measured with HIP events on the stream the kernels run on.
 * kernel: 0 kmer_extract, 1 bv_filter, 2 pair_score, 3 poa_align, 4 post_msa.  Accumulated since
 * the last reset: total milliseconds, number of launches, algorithmic bytes moved.
 */
int rattle_hip_kernel_stats(rattle_ctx *ctx, int kernel, double *total_ms, uint64_t *launches, uint64_t *alg_bytes);
int rattle_hip_kernel_stats_reset(rattle_ctx *ctx);
/* (ABI 4) Host wall time of the stages of the calls made on this context since the last reset, in milliseconds:
 * [0] reserved, [1] `correct` stage 1 (POA #1 + correction of every pack, correct.cpp:395-426), [2] stage 2a (POA #2 of the packs of the
 * many-pack clusters, :427-470), [3] stage 2b + 3a (POA #2 of the other packs beside POA #3 of the many-pack clusters, :489-556),
 * [4] stage 3b (POA #3 of the other clusters), [5..7] reserved.  A measurement aid for the benchmark harness (which stage a slow
 * step was slow in); the reference has no counterpart. */
int rattle_hip_stage_ms(rattle_ctx *ctx, double ms_out[8], int reset);

#ifdef __cplusplus
}
#endif
#endif /* RATTLE_HIP_H */
