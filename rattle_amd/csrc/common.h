// Internal declarations shared by the HIP translation units of librattle_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/rattle_hip.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace rattle {

// RATTLE_TIMING=1: wall-clock of host phases on stderr
struct phase_timer {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    bool on;
    double *keep = nullptr;             // where the elapsed milliseconds are ADDED when the timer stops (rattle_hip_stage_ms), with or without RATTLE_TIMING
    explicit phase_timer(const char *n, double *k = nullptr) : name(n), t0(std::chrono::steady_clock::now()), keep(k) {
        static const bool enabled = getenv("RATTLE_TIMING") != nullptr;
        on = enabled;
    }
    void stop() {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (keep) { *keep += ms; keep = nullptr; }
        if (on) fprintf(stderr, "[rattle] %-28s %8.1f ms\n", name, ms);
        on = false;
    }
    ~phase_timer() { stop(); }
};

void set_error(const std::string &msg);

// Host-side data-parallel loop (post-MSA logic, packing, row expansion): dynamic chunks of one item.
template <typename F>
void parallel_for(size_t n, int n_threads, F f) {
    if (n == 0) return;
    // default width: RATTLE_HOST_THREADS, else all cores (several ranks on one host should split them)
    static const unsigned env_threads = getenv("RATTLE_HOST_THREADS") ? (unsigned)atoi(getenv("RATTLE_HOST_THREADS")) : 0u;
    unsigned hw = env_threads ? env_threads : std::thread::hardware_concurrency();
    size_t T = n_threads > 0 ? (size_t)n_threads : (hw ? hw : 1);
    T = std::min(T, n);
    if (T <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
    std::atomic<size_t> next(0);
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; ++t)
        th.emplace_back([&]() { for (size_t i = next++; i < n; i = next++) f(i); });
    for (auto &x : th) x.join();
}

#define RT_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            ::rattle::set_error(std::string(#call) + ": " + hipGetErrorString(e__) + " (" + __FILE__ + \
                                ":" + std::to_string(__LINE__) + ")");                                 \
            return RATTLE_ERR_HIP;                                                                     \
        }                                                                                              \
    } while (0)

#define RT_TRY(call)            \
    do {                        \
        int r__ = (call);       \
        if (r__ != 0) return r__; \
    } while (0)

// Growable device buffer (never shrinks; contents are not preserved across grow()).  Owns its
// allocation and frees it on destruction, so early error returns do not leak HBM; borrow() makes a
// non-owning view of another buffer (the child contexts of cluster_subsets share the parent's index).
void dev_free(void *p);

template <typename T>
struct dbuf {
    T *p = nullptr;
    size_t cap = 0;
    bool owned = true;
    dbuf() = default;
    dbuf(const dbuf &) = delete;
    dbuf &operator=(const dbuf &) = delete;
    ~dbuf() { release(); }
    int reserve(size_t n) {
        if (n <= cap) return 0;
        release();
        size_t want = n + n / 4 + 64;
        RT_HIP(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        return 0;
    }
    void borrow(const dbuf &o) {
        release();
        p = o.p; cap = o.cap; owned = false;
    }
    void swap(dbuf &o) { std::swap(p, o.p); std::swap(cap, o.cap); std::swap(owned, o.owned); }
    void release() {
        if (p && owned) dev_free(p);
        p = nullptr;
        cap = 0;
        owned = true;
    }
};

// Pinned host buffer for D2H results.
template <typename T>
struct hbuf {
    T *p = nullptr;
    size_t cap = 0;
    hbuf() = default;
    hbuf(const hbuf &) = delete;
    hbuf &operator=(const hbuf &) = delete;
    ~hbuf() { release(); }
    int reserve(size_t n) {
        if (n <= cap) return 0;
        release();
        size_t want = n + n / 4 + 64;
        RT_HIP(hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

enum { K_KMER = 0, K_FILTER = 1, K_SCORE = 2, K_POA = 3, K_POST = 4, K_COUNT = 5 };

struct kstat {
    double ms = 0;
    uint64_t launches = 0, bytes = 0;
};

// Device-resident read index (output of kernel K).
struct read_index {
    uint32_t n = 0;
    int k = 0;
    int both = 0;
    uint64_t total_bases = 0, total_kmers = 0;
    std::vector<uint64_t> h_off;      // [n+1] base offsets
    std::vector<uint64_t> h_koff;     // [n+1] k-mer list offsets (entries)
    std::vector<uint32_t> h_len;      // [n]
    dbuf<uint8_t> seq;                // ASCII bases
    dbuf<uint64_t> off, koff;         // device copies
    dbuf<uint32_t> len;
    dbuf<uint32_t> uh;                // forward hashes in POSITION order          [total_kmers]
    dbuf<uint32_t> kh[2];             // hashes sorted by (hash,pos), per strand     [total_kmers]
    dbuf<uint32_t> kp[2];             // positions in that order                     [total_kmers]
    dbuf<uint64_t> bv[2];             // bit-vectors, 64 words per read              [n*64]
    dbuf<uint32_t> pc[2];             // popcount of each bit-vector                 [n]
};

// phred_symbol(p) = (char)(-10*log10(p)+33) as a table of thresholds built with the host libm (post_msa.hip)
struct phred_table {
    int n0 = 0;
    std::vector<double> lo;             // lo[t]: smallest p whose symbol value is <= n0 + t
    std::vector<uint64_t> exc_bits;     // doubles (bit patterns, sorted) the table would get wrong
    std::vector<int32_t> exc_val;
};
void build_phred_table(phred_table &T);
int phred_lookup_host(const phred_table &T, double p);

// ---- one job over several GPUs (exchange.hip) ----------------------------------------------------------
// The path shards by independent objects: candidate reads of a seed batch, --iso gene clusters,
// `correct` packs.  A context may be one rank of nranks; the only data-path communication is an
// all-gather(v) of small byte strings (hit lists, cluster sets, pack consensi) plus the final gather of
// the corrected reads.  Two transports: RCCL on device buffers (rattle_hip_comm_init) or a caller-supplied
// host-buffer all-gather (rattle_hip_set_exchange: MPI, gloo, tests).
struct exchange {
    int rank = 0, nranks = 1;
    rattle_allgatherv_fn fn = nullptr;
    void *user = nullptr;
    void *comm = nullptr;               // ncclComm_t
    void *rccl = nullptr;               // dlopen handle of librccl.so
    uint64_t calls = 0, bytes = 0;      // statistics
    // Measurement aid (round 5: the multi-GPU curve measured on ONE GPU, tools/rank_replay.py): what the ranks of a job exchange is
    // recorded by a single-rank run (RATTLE_XCHG_RECORD=<file>: every exchange point appends the job's WHOLE payload), and a context
    // configured as rank r of R with rattle_hip_set_exchange(.., fn = NULL) under RATTLE_XCHG_REPLAY=<file> then runs its own share
    // alone: at every exchange it gets its own piece back plus the recorded whole (its peers' part of it).
    FILE *replay = nullptr;
    std::string replay_path;            // ... the file `replay` was opened from (the gather's piece files live beside it)
    // staging of the RCCL all-gather-v, kept between calls: the sharded cluster driver exchanges a few KB per greedy round
    // (119 rounds at 1e6 reads) and three hipMalloc / hipFree pairs per exchange were three device-wide synchronisations each
    dbuf<uint64_t> d_sz;
    dbuf<uint8_t> d_send, d_recv;
    hbuf<uint64_t> h_sz;
    hbuf<uint8_t> h_send, h_recv;       // pinned: the copies are asynchronous and run at link speed
    void reset() {
        rank = 0; nranks = 1; fn = nullptr; user = nullptr; comm = nullptr; calls = 0; bytes = 0;      // the loader's handle (rccl) stays
        if (replay) { fclose(replay); replay = nullptr; }
        replay_path.clear();
        d_sz.release(); d_send.release(); d_recv.release(); h_sz.release(); h_send.release(); h_recv.release();
    }
};

// `correct` work list: the packs of correct.cpp:328-370 from ids and lengths only, plus the static
// assignment of packs to ranks (LPT by the DP cost proxy L_0 * sum L_j).  Pure host code.
struct sref { int32_t rid; uint8_t rev; };
struct pack_plan {
    std::vector<sref> members;          // members of all queued packs, pack after pack
    std::vector<uint32_t> first;        // [n_packs+1] into members
    std::vector<int32_t> pk_cid;        // cluster of each pack
    std::vector<uint32_t> pk_local;     // index among its cluster's queued packs
    std::vector<uint64_t> pk_cost;
    std::vector<uint32_t> pk_owner;     // rank
    std::vector<uint32_t> cl_p0, cl_np; // per cluster: first pack, number of packs
    std::vector<sref> small;            // members of packs that are never queued, in the reference's order
    std::vector<int32_t> small_cid;
    std::vector<uint8_t> small_why;     // 0: <= min_reads (correct.cpp:360), 1: max_pack_cells rule
    std::vector<uint32_t> small_pack;   // local pack index within the cluster
};
int plan_packs(const uint64_t *off, uint32_t n_reads, uint32_t n_clusters, const uint32_t *coff, const int32_t *mid, const uint8_t *mrev,
               const rattle_correct_params *P, int nranks, pack_plan &out);
// longest-processing-time-first assignment of weighted items to nranks bins (ties: lower index / lower rank)
void lpt_assign(const std::vector<uint64_t> &cost, int nranks, std::vector<uint32_t> &owner);
// all-gather of one byte string per rank (sizes first, then the payload)
bool xchg_recording(const rattle_ctx *ctx);
bool xchg_replaying(const rattle_ctx *ctx);
// job_payload = false: a transport self-test (rattle_hip_comm_probe), kept out of the record / replay aid
int xchg_allgatherv(rattle_ctx *ctx, const std::vector<uint8_t> &mine, std::vector<std::vector<uint8_t>> &all, bool job_payload = true);

// one rectangle of kernel A's launch: seeds [s_base, s_base+ns) x candidates [c_base, c_base+nc) of the uploaded
// arrays, tiles [tile_base, ...), look-up table row at d_lut + lut_off
struct bvf_rect {
    uint32_t s_base, ns, c_base, nc, tile_base, lut_off, fwd_bypass, pad;
};

// copy descriptor of the gather kernel: flags bit 0 = reverse complement (qualities reversed)
struct gather_desc {
    uint64_t src, dst;
    uint32_t len, flags;
};

// arguments of the post-MSA kernel (post_msa.hip)
struct post_args {
    const uint8_t *seq, *qual;          // stage input, concatenated (qual: MODE 1 only)
    const uint64_t *off;                // per sequence
    const uint32_t *pack_first;         // per pack
    const uint32_t *col, *width;        // kernel C output: column per base, width per pack
    const uint64_t *moff;               // per pack: byte offset of its rows x width matrix (16-byte aligned)
    const uint64_t *coff;               // per pack: offset of its per-column arrays
    uint8_t *rowc, *rowq;               // matrices: bases / quality bytes; MODE 1 leaves the corrected read at each row start
    int32_t *rfirst, *rlast;            // per sequence: voting window after fix_msa_ends
    uint32_t *tfront, *tback;           // per sequence: bases trimmed by fix_msa_ends (MODE 1)
    uint32_t *olen;                     // per sequence: corrected length (MODE 1)
    uint8_t *ccons, *cflag, *csym;      // per column: winner, occupancy tests, quality symbol of the winner's mean error
    double *cerr;                       // per column: winner's mean error
    uint8_t *cons_out;                  // per pack (at coff): gap-stripped consensus (MODE 2)
    uint32_t *cons_len;
    const double *perr;                 // [256] phred_err per quality byte (utils.cpp:10-13), host computed
    const double *phred_lo;
    const unsigned long long *exc_bits;
    const int32_t *exc_val;
    int32_t phred_n0, phred_cnt;
    uint32_t n_exc;
    uint8_t order[8];                   // vote slot order
    double min_occ, gap_occ, err_ratio;
};

int hw_queues();      // hardware queues the HIP runtime of this process hands out (settled when the library is loaded, abi.hip)

}  // namespace rattle

struct rattle_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing = true;
    rattle::kstat stats[rattle::K_COUNT];
    rattle::read_index idx;
    // scratch for the filter / score kernels
    rattle::dbuf<uint32_t> d_seed, d_cand, d_first;
    rattle::dbuf<rattle::bvf_rect> d_rect;
    rattle::dbuf<uint32_t> d_pi2, d_pj2, d_slot2, d_seed_rect;      // pairs past the count bound, compacted on the device
    rattle::dbuf<uint8_t> d_ps2;
    rattle::dbuf<unsigned long long> d_bound_stats;
    rattle::hbuf<unsigned long long> h_bound_stats;
    rattle::dbuf<uint16_t> d_lut;
    rattle::dbuf<uint8_t> d_pass;
    rattle::dbuf<uint32_t> d_surv;          // survivor list (2 words per entry)
    rattle::dbuf<uint32_t> d_surv2;         // ... sorted by seed (double buffer of the device sort)
    rattle::dbuf<uint8_t> d_sort_tmp;
    rattle::dbuf<uint32_t> d_counter;
    rattle::dbuf<uint32_t> d_pi, d_pj;
    rattle::dbuf<uint8_t> d_ps;
    rattle::dbuf<int32_t> d_res;            // per pair: bases, hc, n_dist, n_matches
    rattle::dbuf<double> d_var;
    rattle::dbuf<uint32_t> d_scratch;       // global scratch for oversize pairs
    rattle::hbuf<uint32_t> h_surv;
    rattle::hbuf<int32_t> h_res;
    rattle::hbuf<double> h_var;
    rattle::hbuf<uint32_t> h_counter;
    // POA arena: kept across stages and calls (allocating ~100 GB costs seconds)
    uint8_t *poa_arena = nullptr;
    size_t poa_arena_bytes = 0;
    hipStream_t poa_st[16] = {};     // one per column class: classes run concurrently
    double stage_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // host wall time of the last calls' stages, summed until read (rattle_hip_stage_ms): [0] cluster, [1] correct stage 1, [2] 2a, [3] 2b+3a, [4] 3b
    int poa_shallow_graphs = 0;      // hint of the caller for the next poa_device_run: near-identical sequences (POA #2 / #3), graphs that are almost chains
    hipEvent_t poa_ev[16] = {};
    hipEvent_t poa_go = nullptr;
    rattle::hbuf<uint32_t> h_poa_col;       // pinned staging for the per-base MSA columns
    // reads staged in HBM by rattle_hip_stage_reads (keys: the host buffers they were copied from)
    const uint8_t *staged_seq_key = nullptr, *staged_qual_key = nullptr;
    uint32_t staged_n = 0;
    uint64_t staged_total = 0;
    rattle::dbuf<uint8_t> d_staged_seq, d_staged_qual;
    // post-MSA kernel constants (built on first use)
    rattle::phred_table phred;
    rattle::dbuf<double> d_phred_lo, d_perr;
    rattle::dbuf<unsigned long long> d_exc_bits;
    rattle::dbuf<int32_t> d_exc_val;
    bool phred_ready = false;
    rattle::exchange xchg;
    rattle_ctx() = default;
    rattle_ctx(const rattle_ctx &) = delete;
    rattle_ctx &operator=(const rattle_ctx &) = delete;
    // streams, events and the arena; the dbuf / hbuf members release themselves
    ~rattle_ctx() {
        if (device < 0) return;                 // host-only context: nothing on a device
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        if (poa_arena) (void)hipFree(poa_arena);
        for (int i = 0; i < 16; ++i) { if (poa_st[i]) (void)hipStreamDestroy(poa_st[i]); if (poa_ev[i]) (void)hipEventDestroy(poa_ev[i]); }
        if (poa_go) (void)hipEventDestroy(poa_go);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace rattle {

// Event-timed launch bracket: records into ctx->stats[which].
struct ktimer {
    rattle_ctx *c;
    int which;
    uint64_t bytes;
    ktimer(rattle_ctx *c_, int w, uint64_t b) : c(c_), which(w), bytes(b) {
        if (c->timing) (void)hipEventRecord(c->ev0, c->stream);
    }
    ~ktimer() {
        c->stats[which].launches++;
        c->stats[which].bytes += bytes;
        if (c->timing) {
            (void)hipEventRecord(c->ev1, c->stream);
            (void)hipEventSynchronize(c->ev1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
            c->stats[which].ms += ms;
        }
    }
};

// kmer_extract.hip
int build_index(rattle_ctx *ctx, const uint8_t *seq, const uint64_t *off, uint32_t n, int k, int both);
// bv_filter.hip : device-side seeds/cands already uploaded in ctx->d_seed/d_cand/d_first/d_lut.
// Writes dense pass bytes (if dense) and/or appends survivors (seed_slot<<1|strand, cand_slot).
int launch_bv_filter(rattle_ctx *ctx, uint32_t n_seeds, uint32_t n_cands, int fwd_bypass, bool dense, bool list,
                     uint32_t list_cap);
// the same over a list of rectangles already uploaded to ctx->d_rect (look-up table rows in ctx->d_lut)
int launch_bv_filter_rects(rattle_ctx *ctx, uint32_t n_rects, uint32_t n_tiles, uint64_t pairs, bool dense, bool list, uint32_t list_cap);
// pair_score.hip : pairs in ctx->d_pi/d_pj/d_ps; results in ctx->d_res (4 ints per pair) + ctx->d_var.
int launch_pair_score(rattle_ctx *ctx, uint32_t n_pairs);
// pair_count.hip : the count pass, seed-major (survivors sorted by seed, the seed's k-mer set as an LDS bit set)
int sort_survivors_by_seed(rattle_ctx *ctx, uint32_t n, uint64_t n_seeds);
int launch_pair_count_seed(rattle_ctx *ctx, uint32_t n_pairs);
// the same pairs, |common| only (d_res[pair]); see pair_score.hip
int launch_pair_count(rattle_ctx *ctx, uint32_t n_pairs);
// poa.hip : device-resident POA over packs (sequences, offsets, column output in HBM)
int poa_device_run(rattle_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, const uint64_t *h_off, uint32_t n_seqs,
                   const uint32_t *h_pack_first, uint32_t n_packs, uint32_t *d_col, uint32_t *d_width, uint32_t *h_width,
                   unsigned long long *h_cnt, std::vector<uint8_t> *skipped);
// post_msa.hip
int launch_gather(rattle_ctx *ctx, const gather_desc *d_desc, uint32_t n, const uint8_t *sseq, const uint8_t *squal, uint8_t *dseq,
                  uint8_t *dqual);
int launch_post_msa(rattle_ctx *ctx, const post_args &A, uint32_t n_packs, int mode);
// cluster_driver.cpp
int cluster_driver_many(rattle_ctx *ctx, const rattle_cluster_params *P, const uint32_t *ids, const uint64_t *sub_off,
                        const uint32_t *which, uint32_t n_which, rattle_cluster_set **outs);
int cluster_driver(rattle_ctx *ctx, const rattle_cluster_params *P, const uint32_t *subset, uint32_t n_subset,
                   rattle_cluster_set **out);

}  // namespace rattle
