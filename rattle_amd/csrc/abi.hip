// extern "C" surface of librattle_hip.so (include/rattle_hip.h).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <dirent.h>
#include <unistd.h>

#include <memory>

#include "common.h"

namespace rattle {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

int launch_pair_score_oversize(rattle_ctx *ctx, const std::vector<uint32_t> &slots, uint32_t max_matches);
int poa_msa_run(rattle_ctx *ctx, const uint8_t *seq, const uint64_t *off, uint32_t n_seqs, const uint32_t *pack_first,
                uint32_t n_packs, rattle_msa_set **out);

int correct_driver(rattle_ctx *ctx, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint32_t n_reads,
                   uint32_t n_clusters, const uint32_t *coff, const int32_t *mid, const uint8_t *mrev,
                   const rattle_correct_params *P, rattle_correction **out);

// Hardware queues the HIP runtime of this process hands out (kernel C runs one stream per POA column class and wants a queue
// for each: classes sharing a queue run one after the other, poa.hip).  The runtime reads GPU_MAX_HW_QUEUES once, when it
// starts -- the first HIP call of the process -- so the value is settled when this library is LOADED, not when kernel C runs:
// if the variable is set by then it is taken at its word; if it is not and the runtime has not started yet (no /dev/kfd
// descriptor in the process), the library sets it to 12 itself; otherwise the runtime is already running with its default of 4.
// RATTLE_NO_ENV_SETUP=1 keeps the library's hands off the environment (a multi-threaded host that cannot have setenv() called under
// it at dlopen time, or one that wants the runtime's default for its other HIP users): kernel C then deals its classes onto four streams.
static int g_hw_queues = 4;
__attribute__((constructor)) static void settle_hw_queues() {
    if (const char *v = getenv("GPU_MAX_HW_QUEUES")) { g_hw_queues = std::max(1, atoi(v)); return; }
    if (getenv("RATTLE_NO_ENV_SETUP")) return;
    bool runtime_up = false;
    if (DIR *d = opendir("/proc/self/fd")) {                 // every descriptor of the process, whatever its number
        char link[300], target[256];
        while (struct dirent *e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            snprintf(link, sizeof link, "/proc/self/fd/%s", e->d_name);
            const ssize_t n = readlink(link, target, sizeof target - 1);
            if (n > 0) { target[n] = 0; if (strcmp(target, "/dev/kfd") == 0) { runtime_up = true; break; } }
        }
        closedir(d);
    }
    if (!runtime_up) { setenv("GPU_MAX_HW_QUEUES", "12", 0); g_hw_queues = 12; }
}
int hw_queues() { return g_hw_queues; }

void dev_free(void *p) { if (p) (void)hipFree(p); }

}  // namespace rattle

using namespace rattle;

// compute entry points need the context's device; a host-only context (rattle_hip_ctx_create_host) has none
static int use_device(rattle_ctx *c) {
    if (c->device < 0) { set_error("this context has no device (rattle_hip_ctx_create_host): only the exchange entry points work"); return RATTLE_ERR_STATE; }
    RT_HIP(hipSetDevice(c->device));
    return 0;
}

extern "C" {

const char *rattle_hip_last_error(void) { return g_err.c_str(); }
int rattle_hip_abi_version(void) { return 4; }

int rattle_hip_ctx_create(int device, rattle_ctx **out) {
    if (!out) { set_error("out is null"); return RATTLE_ERR_ARG; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        set_error(std::string("no HIP device available (librattle_hip has no CPU fallback): ") + hipGetErrorString(e));
        return RATTLE_ERR_HIP;
    }
    if (device < 0 || device >= n) { set_error("device index out of range"); return RATTLE_ERR_ARG; }
    RT_HIP(hipSetDevice(device));
    rattle_ctx *c = new rattle_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess) {
        set_error("stream/event creation failed");
        delete c;
        return RATTLE_ERR_HIP;
    }
    if (getenv("RATTLE_NO_KERNEL_TIMING")) c->timing = false;
    *out = c;
    return 0;
}

int rattle_hip_ctx_create_host(rattle_ctx **out) {
    if (!out) { set_error("out is null"); return RATTLE_ERR_ARG; }
    rattle_ctx *c = new rattle_ctx();
    c->device = -1;
    c->timing = false;
    *out = c;
    return 0;
}

void rattle_hip_ctx_destroy(rattle_ctx *c) {
    if (!c) return;
    (void)rattle_hip_comm_destroy(c);
    delete c;            // ~rattle_ctx: streams, events, arena; the buffers release themselves
}

int rattle_hip_load_reads(rattle_ctx *c, const uint8_t *seq, const uint64_t *off, uint32_t n, int k, int both) {
    if (!c || !off || (n && !seq)) { set_error("null argument"); return RATTLE_ERR_ARG; }
    RT_TRY(use_device(c));
    return build_index(c, seq, off, n, k, both);
}

int rattle_hip_get_read_index(rattle_ctx *c, uint32_t r, int strand, uint32_t *hash_out, int32_t *pos_out, uint64_t *bv_out,
                              uint32_t *pc_out) {
    if (!c) { set_error("null ctx"); return RATTLE_ERR_ARG; }
    read_index &X = c->idx;
    if (r >= X.n) { set_error("read out of range"); return RATTLE_ERR_ARG; }
    if (strand < 0 || strand > 1 || (strand == 1 && !X.both)) { set_error("strand not indexed"); return RATTLE_ERR_ARG; }
    RT_TRY(use_device(c));
    uint64_t ko = X.h_koff[r], nk = X.h_koff[r + 1] - ko;
    if (nk && hash_out) RT_HIP(hipMemcpy(hash_out, X.kh[strand].p + ko, nk * 4, hipMemcpyDeviceToHost));
    if (nk && pos_out) RT_HIP(hipMemcpy(pos_out, X.kp[strand].p + ko, nk * 4, hipMemcpyDeviceToHost));
    if (bv_out) RT_HIP(hipMemcpy(bv_out, X.bv[strand].p + (uint64_t)r * 64, 512, hipMemcpyDeviceToHost));
    if (pc_out) RT_HIP(hipMemcpy(pc_out, X.pc[strand].p + r, 4, hipMemcpyDeviceToHost));
    return 0;
}

int rattle_hip_bv_filter(rattle_ctx *c, const uint32_t *seed_ids, uint32_t n_seeds, const uint32_t *cand_ids, uint32_t n_cands,
                         const uint32_t *first_cand, const uint16_t *lut, int fwd_bypass, uint8_t *out_pass) {
    if (!c || !seed_ids || !cand_ids || !first_cand || !lut || !out_pass) { set_error("null argument"); return RATTLE_ERR_ARG; }
    if (c->idx.n == 0) { set_error("no reads loaded"); return RATTLE_ERR_STATE; }
    for (uint32_t i = 0; i < n_seeds; ++i) if (seed_ids[i] >= c->idx.n) { set_error("seed id out of range"); return RATTLE_ERR_ARG; }
    for (uint32_t i = 0; i < n_cands; ++i) if (cand_ids[i] >= c->idx.n) { set_error("cand id out of range"); return RATTLE_ERR_ARG; }
    if (n_seeds == 0 || n_cands == 0) return 0;
    RT_TRY(use_device(c));
    hipStream_t st = c->stream;
    RT_TRY(c->d_seed.reserve(n_seeds)); RT_TRY(c->d_first.reserve(n_seeds)); RT_TRY(c->d_cand.reserve(n_cands));
    RT_TRY(c->d_lut.reserve(4097)); RT_TRY(c->d_pass.reserve((size_t)n_seeds * n_cands)); RT_TRY(c->d_counter.reserve(4));
    RT_HIP(hipMemcpyAsync(c->d_seed.p, seed_ids, n_seeds * 4, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(c->d_first.p, first_cand, n_seeds * 4, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(c->d_cand.p, cand_ids, (size_t)n_cands * 4, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(c->d_lut.p, lut, 4097 * 2, hipMemcpyHostToDevice, st));
    RT_TRY(launch_bv_filter(c, n_seeds, n_cands, fwd_bypass, true, false, 0));
    RT_HIP(hipMemcpyAsync(out_pass, c->d_pass.p, (size_t)n_seeds * n_cands, hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    return 0;
}

int rattle_hip_pair_score(rattle_ctx *c, const uint32_t *i_ids, const uint32_t *j_ids, const uint8_t *strand, uint32_t n,
                          int32_t *bases, int32_t *hc_bases, int32_t *n_dist, double *variance, int32_t *n_matches) {
    if (!c || !i_ids || !j_ids || !strand) { set_error("null argument"); return RATTLE_ERR_ARG; }
    read_index &X = c->idx;
    if (X.n == 0) { set_error("no reads loaded"); return RATTLE_ERR_STATE; }
    for (uint32_t p = 0; p < n; ++p) {
        if (i_ids[p] >= X.n || j_ids[p] >= X.n) { set_error("read id out of range"); return RATTLE_ERR_ARG; }
        if (strand[p] > 1 || (strand[p] == 1 && !X.both)) { set_error("strand not indexed"); return RATTLE_ERR_ARG; }
    }
    if (n == 0) return 0;
    RT_TRY(use_device(c));
    hipStream_t st = c->stream;
    RT_TRY(c->d_pi.reserve(n)); RT_TRY(c->d_pj.reserve(n)); RT_TRY(c->d_ps.reserve(n));
    RT_TRY(c->d_res.reserve((size_t)n * 4)); RT_TRY(c->d_var.reserve(n));
    RT_HIP(hipMemcpyAsync(c->d_pi.p, i_ids, (size_t)n * 4, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(c->d_pj.p, j_ids, (size_t)n * 4, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(c->d_ps.p, strand, n, hipMemcpyHostToDevice, st));
    RT_TRY(launch_pair_score(c, n));
    std::vector<int32_t> res((size_t)n * 4);
    std::vector<double> var(n);
    RT_HIP(hipMemcpyAsync(res.data(), c->d_res.p, (size_t)n * 16, hipMemcpyDeviceToHost, st));
    RT_HIP(hipMemcpyAsync(var.data(), c->d_var.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    std::vector<uint32_t> big;
    uint32_t big_m = 0;
    for (uint32_t p = 0; p < n; ++p)
        if (res[4 * (size_t)p] == INT32_MIN) { big.push_back(p); big_m = std::max(big_m, (uint32_t)res[4 * (size_t)p + 3]); }
    if (!big.empty()) {
        RT_TRY(launch_pair_score_oversize(c, big, big_m));
        RT_HIP(hipMemcpy(res.data(), c->d_res.p, (size_t)n * 16, hipMemcpyDeviceToHost));
        RT_HIP(hipMemcpy(var.data(), c->d_var.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    }
    uint64_t bytes = 0;
    for (uint32_t p = 0; p < n; ++p) {
        if (bases) bases[p] = res[4 * (size_t)p];
        if (hc_bases) hc_bases[p] = res[4 * (size_t)p + 1];
        if (n_dist) n_dist[p] = res[4 * (size_t)p + 2];
        if (n_matches) n_matches[p] = res[4 * (size_t)p + 3];
        if (variance) variance[p] = var[p];
        bytes += 8ull * ((X.h_koff[i_ids[p] + 1] - X.h_koff[i_ids[p]]) + (X.h_koff[j_ids[p] + 1] - X.h_koff[j_ids[p]]));
    }
    c->stats[K_SCORE].bytes += bytes;
    return 0;
}

int rattle_hip_cluster_reads(rattle_ctx *c, const rattle_cluster_params *P, rattle_cluster_set **out) {
    if (!c || !P || !out) { set_error("null argument"); return RATTLE_ERR_ARG; }
    *out = nullptr;
    RT_TRY(use_device(c));
    if (c->idx.k == 0) { set_error("no reads loaded"); return RATTLE_ERR_STATE; }      // (never loaded, or the index made room for a big `correct`)
    if (!P->is_rna && !c->idx.both) { set_error("cDNA mode needs the reads loaded with both_strands=1"); return RATTLE_ERR_STATE; }
    // cluster.cpp:42: `if (is_rna) return {-1,false}` comes before the reverse test, so a both-strand index would over-cluster
    if (P->is_rna && c->idx.both) { set_error("--rna mode needs the reads loaded with both_strands=0"); return RATTLE_ERR_STATE; }
    return cluster_driver(c, P, nullptr, 0, out);
}

int rattle_hip_cluster_subset(rattle_ctx *c, const rattle_cluster_params *P, const uint32_t *subset, uint32_t n_subset,
                              rattle_cluster_set **out) {
    if (!c || !P || !out || (n_subset && !subset)) { set_error("null argument"); return RATTLE_ERR_ARG; }
    *out = nullptr;
    for (uint32_t i = 0; i < n_subset; ++i) if (subset[i] >= c->idx.n) { set_error("subset id out of range"); return RATTLE_ERR_ARG; }
    RT_TRY(use_device(c));
    if (!P->is_rna && !c->idx.both) { set_error("cDNA mode needs the reads loaded with both_strands=1"); return RATTLE_ERR_STATE; }
    // cluster.cpp:42: `if (is_rna) return {-1,false}` comes before the reverse test, so a both-strand index would over-cluster
    if (P->is_rna && c->idx.both) { set_error("--rna mode needs the reads loaded with both_strands=0"); return RATTLE_ERR_STATE; }
    static const uint32_t none = 0;
    return cluster_driver(c, P, n_subset ? subset : &none, n_subset, out);
}

// Many independent subsets at once (the --iso second level: one subset per gene cluster).  Small
// problems are launch- and synchronisation-bound, so the subsets' greedy rounds advance in lockstep and every
// round is ONE evaluation on the device (cluster_driver.hip: cluster_driver_many).
int rattle_hip_cluster_subsets(rattle_ctx *c, const rattle_cluster_params *P, const uint32_t *ids, const uint64_t *sub_off,
                               uint32_t n_subsets, rattle_cluster_set **outs, int n_workers) {
    if (!c || !P || !outs || !sub_off || (n_subsets && sub_off[n_subsets] && !ids)) { set_error("null argument"); return RATTLE_ERR_ARG; }
    for (uint32_t i = 0; i < n_subsets; ++i) outs[i] = nullptr;
    for (uint64_t i = 0; i < sub_off[n_subsets]; ++i) if (ids[i] >= c->idx.n) { set_error("subset id out of range"); return RATTLE_ERR_ARG; }
    RT_TRY(use_device(c));
    if (!P->is_rna && !c->idx.both) { set_error("cDNA mode needs the reads loaded with both_strands=1"); return RATTLE_ERR_STATE; }
    // cluster.cpp:42: `if (is_rna) return {-1,false}` comes before the reverse test, so a both-strand index would over-cluster
    if (P->is_rna && c->idx.both) { set_error("--rna mode needs the reads loaded with both_strands=0"); return RATTLE_ERR_STATE; }
    RT_HIP(hipStreamSynchronize(c->stream));
    // several ranks: the subsets (gene clusters, main.cpp:281-318) are independent -> LPT over ranks by the
    // pair count proxy n^2, every rank clusters its own and the results are all-gathered at the end
    const int R = c->xchg.nranks, rk = c->xchg.rank;
    std::vector<uint32_t> owner(n_subsets, 0);
    if (R > 1) {
        std::vector<uint64_t> cost(n_subsets);
        for (uint32_t i = 0; i < n_subsets; ++i) { const uint64_t n = sub_off[i + 1] - sub_off[i]; cost[i] = n * n; }
        lpt_assign(cost, R, owner);
    }
    std::vector<uint32_t> order;                                  // my subsets, largest first
    for (uint32_t i = 0; i < n_subsets; ++i) if ((int)owner[i] == rk) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        const uint64_t la = sub_off[a + 1] - sub_off[a], lb = sub_off[b + 1] - sub_off[b];
        return la != lb ? la > lb : a < b;
    });
    (void)n_workers;                                              // round 1 ran one host thread per worker; the subsets now advance in lockstep
    int rc = cluster_driver_many(c, P, ids, sub_off, order.data(), (uint32_t)order.size(), outs);
    std::string err_msg = rc != 0 ? rattle_hip_last_error() : "";
    if (rc != 0) for (uint32_t i : order) { rattle_hip_cluster_set_free(outs[i]); outs[i] = nullptr; }
    if (R > 1) {
        // all ranks take part in the exchange even after a local failure (an empty payload marks it)
        std::vector<uint8_t> mine;
        auto put = [&mine](const void *p, size_t n) { const size_t at = mine.size(); mine.resize(at + n); if (n) memcpy(mine.data() + at, p, n); };
        if (rc == 0)
            for (uint32_t i : order) {
                const rattle_cluster_set *cs = outs[i];
                const uint32_t hdr[3] = {i, cs->n_clusters, cs->offsets[cs->n_clusters]};
                put(hdr, 12); put(cs->counters, 64);
                put(cs->main_id, (size_t)hdr[1] * 4); put(cs->offsets, ((size_t)hdr[1] + 1) * 4); put(cs->member_id, (size_t)hdr[2] * 4);
                put(cs->main_rev, hdr[1]); put(cs->member_rev, hdr[2]);
                const uint8_t z[4] = {0, 0, 0, 0};
                put(z, (4 - (hdr[1] + hdr[2]) % 4) % 4);
            }
        else { const uint32_t bad = 0xFFFFFFFFu; put(&bad, 4); }
        std::vector<std::vector<uint8_t>> all;
        int xrc = xchg_allgatherv(c, mine, all);
        if (rc == 0) rc = xrc;
        for (int r = 0; r < R && rc == 0; ++r) {
            if (r == rk) continue;
            const std::vector<uint8_t> &b = all[r];
            if (b.size() == 4) { set_error("cluster_subsets failed on rank " + std::to_string(r)); rc = RATTLE_ERR_HIP; break; }
            size_t at = 0;
            while (at + 12 <= b.size()) {
                uint32_t hdr[3];
                memcpy(hdr, b.data() + at, 12); at += 12;
                if (hdr[0] >= n_subsets || outs[hdr[0]]) { set_error("cluster_subsets exchange: malformed record"); rc = RATTLE_ERR_HIP; break; }
                rattle_cluster_set *cs = (rattle_cluster_set *)calloc(1, sizeof(rattle_cluster_set));
                cs->n_clusters = hdr[1];
                memcpy(cs->counters, b.data() + at, 64); at += 64;
                cs->main_id = (int32_t *)malloc(std::max<size_t>(1, hdr[1]) * 4); cs->offsets = (uint32_t *)malloc(((size_t)hdr[1] + 1) * 4);
                cs->member_id = (int32_t *)malloc(std::max<size_t>(1, hdr[2]) * 4);
                cs->main_rev = (uint8_t *)malloc(std::max<size_t>(1, hdr[1])); cs->member_rev = (uint8_t *)malloc(std::max<size_t>(1, hdr[2]));
                memcpy(cs->main_id, b.data() + at, (size_t)hdr[1] * 4); at += (size_t)hdr[1] * 4;
                memcpy(cs->offsets, b.data() + at, ((size_t)hdr[1] + 1) * 4); at += ((size_t)hdr[1] + 1) * 4;
                memcpy(cs->member_id, b.data() + at, (size_t)hdr[2] * 4); at += (size_t)hdr[2] * 4;
                memcpy(cs->main_rev, b.data() + at, hdr[1]); at += hdr[1];
                memcpy(cs->member_rev, b.data() + at, hdr[2]); at += hdr[2];
                at += (4 - (hdr[1] + hdr[2]) % 4) % 4;
                outs[hdr[0]] = cs;
            }
        }
    }
    if (rc != 0) {
        for (uint32_t i = 0; i < n_subsets; ++i) { rattle_hip_cluster_set_free(outs[i]); outs[i] = nullptr; }
        return rc;
    }
    return 0;
}

int rattle_hip_stage_reads(rattle_ctx *c, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint32_t n) {
    if (!c || !off || (n && !seq)) { set_error("null argument"); return RATTLE_ERR_ARG; }
    RT_TRY(use_device(c));
    const uint64_t total = off[n] - off[0];
    c->staged_seq_key = nullptr; c->staged_qual_key = nullptr;
    RT_TRY(c->d_staged_seq.reserve(total + 64));
    RT_HIP(hipMemcpy(c->d_staged_seq.p, seq + off[0], total, hipMemcpyHostToDevice));
    if (qual) {
        RT_TRY(c->d_staged_qual.reserve(total + 64));
        RT_HIP(hipMemcpy(c->d_staged_qual.p, qual + off[0], total, hipMemcpyHostToDevice));
        c->staged_qual_key = qual;
    }
    c->staged_seq_key = seq; c->staged_n = n; c->staged_total = total;
    return 0;
}

int rattle_hip_unstage_reads(rattle_ctx *c) {
    if (!c) { set_error("null ctx"); return RATTLE_ERR_ARG; }
    c->staged_seq_key = nullptr; c->staged_qual_key = nullptr; c->staged_n = 0; c->staged_total = 0;
    c->d_staged_seq.release(); c->d_staged_qual.release();
    return 0;
}

// main.cpp:254-262: stable sort by length, longest first (counting sort; lengths are small integers), the reads
// gathered into processing order (on the device when they are staged there) and indexed with k-mer size k
static int sort_and_index(rattle_ctx *c, const uint8_t *seq, const uint64_t *off, uint32_t n, int k, int both, std::vector<uint32_t> &order) {
    order.resize(n);
    {
        phase_timer T("cluster: sort + gather");
        uint64_t max_len = 0;
        for (uint32_t i = 0; i < n; ++i) max_len = std::max<uint64_t>(max_len, off[i + 1] - off[i]);
        if (max_len <= (1u << 24)) {
            std::vector<uint32_t> start(max_len + 2, 0);
            for (uint32_t i = 0; i < n; ++i) ++start[max_len - (off[i + 1] - off[i]) + 1];      // bucket 0 = longest
            for (uint64_t b = 0; b <= max_len; ++b) start[b + 1] += start[b];
            for (uint32_t i = 0; i < n; ++i) order[start[max_len - (off[i + 1] - off[i])]++] = i;
        } else {
            for (uint32_t i = 0; i < n; ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [off](uint32_t a, uint32_t b) { return off[a + 1] - off[a] > off[b + 1] - off[b]; });
        }
    }
    std::vector<uint64_t> soff(n + 1);
    {
        uint64_t p = 0;
        for (uint32_t i = 0; i < n; ++i) { soff[i] = p; p += off[order[i] + 1] - off[order[i]]; }
        soff[n] = p;
    }
    if (n && c->staged_seq_key == seq && c->staged_n == n && c->staged_total == off[n] - off[0] && off[0] == 0) {
        // the reads are resident (rattle_hip_stage_reads): gather them into processing order on the device
        phase_timer T("cluster: device gather + build_index");
        std::vector<gather_desc> desc(n);
        for (uint32_t i = 0; i < n; ++i) desc[i] = gather_desc{off[order[i]], soff[i], (uint32_t)(soff[i + 1] - soff[i]), 0u};
        dbuf<gather_desc> d_desc;
        dbuf<uint8_t> d_cat;
        RT_TRY(d_desc.reserve(n)); RT_TRY(d_cat.reserve(soff[n] + 64));
        RT_HIP(hipMemcpyAsync(d_desc.p, desc.data(), (size_t)n * sizeof(gather_desc), hipMemcpyHostToDevice, c->stream));
        RT_TRY(launch_gather(c, d_desc.p, n, c->d_staged_seq.p, nullptr, d_cat.p, nullptr));
        int rc = build_index(c, d_cat.p, soff.data(), n, k, both);
        RT_HIP(hipStreamSynchronize(c->stream));
        return rc;
    }
    std::unique_ptr<uint8_t[]> cat(new uint8_t[off[n] - off[0] + 1]);
    {
        phase_timer T("cluster: host gather");
        const size_t chunk = 2048;
        parallel_for((n + chunk - 1) / chunk, 0, [&](size_t ch) {
            for (size_t i = ch * chunk; i < std::min<size_t>(n, (ch + 1) * chunk); ++i)
                memcpy(cat.get() + soff[i], seq + off[order[i]], soff[i + 1] - soff[i]);
        });
    }
    phase_timer T("cluster: build_index");
    return build_index(c, cat.get(), soff.data(), n, k, both);
}

int rattle_hip_cluster_unsorted(rattle_ctx *c, const uint8_t *seq, const uint64_t *off, uint32_t n, int k,
                                const rattle_cluster_params *P, rattle_cluster_set **out) {
    if (!c || !off || !P || !out || (n && !seq)) { set_error("null argument"); return RATTLE_ERR_ARG; }
    *out = nullptr;
    RT_TRY(use_device(c));
    phase_timer T_all("cluster_unsorted: total");
    std::vector<uint32_t> order;
    RT_TRY(sort_and_index(c, seq, off, n, k, P->is_rna ? 0 : 1, order));
    { phase_timer T("cluster: greedy driver"); RT_TRY(cluster_driver(c, P, nullptr, 0, out)); }
    rattle_cluster_set *cs = *out;
    const uint32_t nm = cs->offsets[cs->n_clusters];
    for (uint32_t i = 0; i < cs->n_clusters; ++i) cs->main_id[i] = (int32_t)order[cs->main_id[i]];
    for (uint32_t i = 0; i < nm; ++i) cs->member_id[i] = (int32_t)order[cs->member_id[i]];
    return 0;
}

// `rattle cluster --iso`, main.cpp:254-323, for reads in FILE order: gene level with (k, gene params), then every
// gene cluster's members -- re-sorted by length desc, ties larger id first (:285-291; the order cluster_reads
// already leaves them in) -- clustered again with (iso_k, iso params); transcript clusters appended in gene
// order, each carrying its gene's index.
int rattle_hip_cluster_iso_unsorted(rattle_ctx *c, const uint8_t *seq, const uint64_t *off, uint32_t n, int k, int iso_k,
                                    const rattle_cluster_params *P, const rattle_cluster_params *iso_P, rattle_cluster_set **out,
                                    uint32_t *n_gene_clusters) {
    if (!c || !off || !P || !iso_P || !out || (n && !seq)) { set_error("null argument"); return RATTLE_ERR_ARG; }
    *out = nullptr;
    RT_TRY(use_device(c));
    phase_timer T_all("cluster_iso_unsorted: total");
    std::vector<uint32_t> order;
    RT_TRY(sort_and_index(c, seq, off, n, k, P->is_rna ? 0 : 1, order));
    rattle_cluster_set *gene = nullptr;
    { phase_timer T("cluster: gene level"); RT_TRY(cluster_driver(c, P, nullptr, 0, &gene)); }
    struct freer { rattle_cluster_set *p; ~freer() { rattle_hip_cluster_set_free(p); } } gene_free{gene};
    if (n_gene_clusters) *n_gene_clusters = gene->n_clusters;
    // second index with the iso k-mer size over the same (already sorted, device-resident) reads
    {
        phase_timer T("cluster: iso index");
        std::vector<uint64_t> soff = c->idx.h_off;
        RT_TRY(build_index(c, c->idx.seq.p, soff.data(), n, iso_k, iso_P->is_rna ? 0 : 1));
    }
    const uint32_t G = gene->n_clusters;
    std::vector<uint32_t> ids(gene->offsets[G] ? gene->offsets[G] : 1);
    std::vector<uint64_t> sub_off(G + 1, 0);
    for (uint32_t g = 0; g <= G; ++g) sub_off[g] = gene->offsets[g];
    for (uint32_t i = 0; i < gene->offsets[G]; ++i) ids[i] = (uint32_t)gene->member_id[i];
    std::vector<rattle_cluster_set *> subs(G ? G : 1, nullptr);
    { phase_timer T("cluster: iso level"); RT_TRY(rattle_hip_cluster_subsets(c, iso_P, ids.data(), sub_off.data(), G, subs.data(), 0)); }
    size_t nc = 0, nm = 0;
    for (uint32_t g = 0; g < G; ++g) { nc += subs[g]->n_clusters; nm += subs[g]->offsets[subs[g]->n_clusters]; }
    rattle_cluster_set *R = (rattle_cluster_set *)calloc(1, sizeof(rattle_cluster_set));
    R->n_clusters = (uint32_t)nc;
    R->main_id = (int32_t *)malloc(std::max<size_t>(1, nc) * 4); R->main_rev = (uint8_t *)malloc(std::max<size_t>(1, nc));
    R->gene_id = (int32_t *)malloc(std::max<size_t>(1, nc) * 4);
    R->offsets = (uint32_t *)malloc((nc + 1) * 4);
    R->member_id = (int32_t *)malloc(std::max<size_t>(1, nm) * 4); R->member_rev = (uint8_t *)malloc(std::max<size_t>(1, nm));
    uint32_t ci = 0, mi = 0;
    for (int i = 0; i < 8; ++i) R->counters[i] = gene->counters[i];
    for (uint32_t g = 0; g < G; ++g) {
        const rattle_cluster_set *S = subs[g];
        const uint32_t *gids = ids.data() + sub_off[g];
        for (uint32_t k2 = 0; k2 < S->n_clusters; ++k2) {
            R->main_id[ci] = (int32_t)order[gids[S->main_id[k2]]]; R->main_rev[ci] = S->main_rev[k2]; R->gene_id[ci] = (int32_t)g;
            R->offsets[ci] = mi;
            for (uint32_t t = S->offsets[k2]; t < S->offsets[k2 + 1]; ++t) { R->member_id[mi] = (int32_t)order[gids[S->member_id[t]]]; R->member_rev[mi] = S->member_rev[t]; ++mi; }
            ++ci;
        }
        for (int i = 0; i < 8; ++i) R->counters[i] += S->counters[i];
        rattle_hip_cluster_set_free(subs[g]);
    }
    R->offsets[nc] = mi;
    *out = R;
    return 0;
}

void rattle_hip_cluster_set_free(rattle_cluster_set *cs) {
    if (!cs) return;
    free(cs->main_id); free(cs->main_rev); free(cs->offsets); free(cs->member_id); free(cs->member_rev); free(cs->gene_id);
    free(cs);
}

int rattle_hip_poa_msa(rattle_ctx *c, const uint8_t *seq, const uint64_t *off, uint32_t n_seqs, const uint32_t *pack_first,
                       uint32_t n_packs, rattle_msa_set **out) {
    if (!c || !off || !pack_first || !out) { set_error("null argument"); return RATTLE_ERR_ARG; }
    *out = nullptr;
    RT_TRY(use_device(c));
    return poa_msa_run(c, seq, off, n_seqs, pack_first, n_packs, out);
}

void rattle_hip_msa_set_free(rattle_msa_set *ms) {
    if (!ms) return;
    free(ms->width); free(ms->row_offset); free(ms->rows);
    free(ms);
}

int rattle_hip_correct_reads(rattle_ctx *c, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint32_t n_reads,
                             uint32_t n_clusters, const uint32_t *coff, const int32_t *mid, const uint8_t *mrev,
                             const rattle_correct_params *P, rattle_correction **out) {
    if (!c || !off || !P || !out || (n_clusters && (!coff || !mid || !mrev)) || (n_reads && (!seq || !qual))) {
        set_error("null argument");
        return RATTLE_ERR_ARG;
    }
    *out = nullptr;
    RT_TRY(use_device(c));
    {
        // A context that clustered its reads still holds their k-mer index (12-20 bytes per base) and kernel B's work lists; `correct`
        // uses none of it and sizes its arena by what is free.  Beyond 24 GB the index goes first (3e6 mixed reads: 72 GB of it beside the
        // arena ran stage 2 out of memory); the reads must then be loaded again before the next cluster call, which is what every
        // caller of the big jobs does anyway (RATTLE_ERR_STATE "no reads loaded" otherwise).
        read_index &X = c->idx;
        const uint64_t held = 4ull * (X.uh.cap + X.kh[0].cap + X.kh[1].cap + X.kp[0].cap + X.kp[1].cap) + 8ull * (X.bv[0].cap + X.bv[1].cap);
        const char *keep_mb = getenv("RATTLE_INDEX_KEEP_MB");            // (tests lower the bound to see the index go)
        if (held > (keep_mb ? (uint64_t)strtoull(keep_mb, nullptr, 10) << 20 : 24ull << 30)) {
            X.uh.release(); X.kh[0].release(); X.kh[1].release(); X.kp[0].release(); X.kp[1].release(); X.bv[0].release(); X.bv[1].release();
            X.pc[0].release(); X.pc[1].release(); X.seq.release(); X.off.release(); X.koff.release(); X.len.release();
            X.n = 0; X.k = 0; X.total_bases = X.total_kmers = 0;
            c->d_surv.release(); c->d_surv2.release(); c->d_sort_tmp.release(); c->d_pi.release(); c->d_pj.release(); c->d_ps.release();
            c->d_pi2.release(); c->d_pj2.release(); c->d_slot2.release(); c->d_ps2.release(); c->d_res.release(); c->d_var.release();
            c->d_scratch.release(); c->d_pass.release();
        }
    }
    int rc = correct_driver(c, seq, qual, off, n_reads, n_clusters, coff, mid, mrev, P, out);
    if (rc != 0 && *out) { rattle_hip_correction_free(*out); *out = nullptr; }
    return rc;
}

int rattle_hip_reserve_arena(rattle_ctx *c, uint64_t bytes) {
    if (!c) { set_error("null ctx"); return RATTLE_ERR_ARG; }
    RT_TRY(use_device(c));
    size_t free_b = 0, total_b = 0;
    RT_HIP(hipMemGetInfo(&free_b, &total_b));
    const uint64_t take = std::min<uint64_t>(bytes, (uint64_t)((free_b + c->poa_arena_bytes) * 0.8));
    if (take <= c->poa_arena_bytes) return 0;
    if (c->poa_arena) (void)hipFree(c->poa_arena);
    c->poa_arena = nullptr; c->poa_arena_bytes = 0;
    RT_HIP(hipMalloc((void **)&c->poa_arena, take));
    c->poa_arena_bytes = take;
    return 0;
}

static void free_set(rattle_read_set &s) {
    free(s.read_id); free(s.cluster_id); free(s.n_reads); free(s.off); free(s.seq); free(s.qual);
}

void rattle_hip_correction_free(rattle_correction *r) {
    if (!r) return;
    free_set(r->corrected); free_set(r->uncorrected); free_set(r->consensi);
    free(r->skipped.cluster_id); free(r->skipped.pack); free(r->skipped.stage); free(r->skipped.read_off); free(r->skipped.read_id);
    free(r->corrected_pack); free(r->uncorrected_pack);
    free(r);
}

// Test hook (no device needed): the value of `-10*log10(p)+33` before the narrowing to char, once through the
// threshold table kernel D uses and once through the host libm (utils.cpp:6-8).  They must agree for every p.
int rattle_hip_debug_phred_symbol(double p, int *table_value, int *libm_value) {
    static std::mutex mu;
    static phred_table T;
    static bool ready = false;
    {
        std::lock_guard<std::mutex> g(mu);
        if (!ready) { build_phred_table(T); ready = true; }
    }
    if (table_value) *table_value = phred_lookup_host(T, p);
    if (libm_value) *libm_value = (int)(-10 * log10(p) + 33);
    return (int)T.exc_bits.size();
}

int rattle_hip_kernel_stats(rattle_ctx *c, int kernel, double *ms, uint64_t *launches, uint64_t *bytes) {
    if (!c || kernel < 0 || kernel >= K_COUNT) { set_error("bad kernel id"); return RATTLE_ERR_ARG; }
    if (ms) *ms = c->stats[kernel].ms;
    if (launches) *launches = c->stats[kernel].launches;
    if (bytes) *bytes = c->stats[kernel].bytes;
    return 0;
}

int rattle_hip_kernel_stats_reset(rattle_ctx *c) {
    if (!c) { set_error("null ctx"); return RATTLE_ERR_ARG; }
    for (int i = 0; i < K_COUNT; ++i) c->stats[i] = kstat();
    return 0;
}

int rattle_hip_stage_ms(rattle_ctx *c, double *ms_out, int reset) {
    if (!c || !ms_out) { set_error("null argument"); return RATTLE_ERR_ARG; }
    for (int i = 0; i < 8; ++i) { ms_out[i] = c->stage_ms[i]; if (reset) c->stage_ms[i] = 0; }
    return 0;
}

}  // extern "C"
