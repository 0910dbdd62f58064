// One job over the GPUs of a node: the all-gather(v) behind the sharded `cluster` / `correct` paths
//  and the reassembly of a sharded correction.
//
// The reference parallelises the candidate loop of a seed (cluster.cpp:138-158,189-209), the --iso gene
// clusters (main.cpp:281-318) and the pack queue (correct.cpp:377-392) with host threads in one address
// space.  Here each of those axes is cut over ranks (one per GPU); what the ranks must tell each other
// is small (hit lists, cluster sets, pack consensi), so the exchange is an all-gather of one byte
// string per rank, and once per `correct` a gather of the corrected reads to the root.
//   * RCCL transport: device staging buffers, ncclAllGather of the sizes, then one ncclBroadcast per
//     rank inside a group (= all-gather-v) on the context's stream; xGMI is point-to-point, so the
//     broadcasts of different roots use different links.  librccl.so is dlopen'ed on first use: a
//     single-GPU process never loads it.
//   * host transport: a caller-supplied all-gather-v on host buffers (MPI, gloo, tests).
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace rattle {

namespace {

// the part of the RCCL API used here (rccl.h; types restated so the header is not needed at build time)
struct nccl_uid { char internal[128]; };
typedef void *nccl_comm;
enum { NCCL_UINT8 = 1, NCCL_UINT64 = 5 };
struct rccl_api {
    void *h = nullptr;
    int (*GetUniqueId)(nccl_uid *) = nullptr;
    int (*CommInitRank)(nccl_comm *, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

rccl_api &rccl() { static rccl_api A; return A; }

int load_rccl() {
    // one loader for the process: rattle --devices runs one host thread per GPU, and the handle must not be visible before
    // every symbol is resolved
    static std::once_flag once;
    static int rc = 0;
    static std::string why;
    std::call_once(once, []() {
        rccl_api T;
        const char *chosen = getenv("RATTLE_RCCL_LIB");                 // a particular RCCL build (or the tests' double)
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (chosen && *chosen) T.h = dlopen(chosen, RTLD_NOW | RTLD_LOCAL);
        else for (const char *n : names) { T.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (T.h) break; }
        if (!T.h) { why = std::string(chosen && *chosen ? chosen : "librccl.so") + " not loadable: " + dlerror(); rc = RATTLE_ERR_HIP; return; }
#define SYM(field, name) *(void **)(&T.field) = dlsym(T.h, name); if (!T.field) { why = "librccl.so lacks " name; rc = RATTLE_ERR_HIP; return; }
        SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
        SYM(AllGather, "ncclAllGather") SYM(Broadcast, "ncclBroadcast") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
        SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        rccl() = T;
    });
    if (rc) set_error(why);
    return rc;
}

#define RT_NCCL(call)                                                                                    \
    do {                                                                                                 \
        int r__ = (call);                                                                                \
        if (r__ != 0) {                                                                                  \
            set_error(std::string(#call) + ": " + rccl().GetErrorString(r__) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
            return RATTLE_ERR_HIP;                                                                       \
        }                                                                                                \
    } while (0)

// all-gather-v of device buffers through RCCL: piece r (bytes[r] long) lands at recv + displ[r]
int rccl_allgatherv(rattle_ctx *ctx, const uint8_t *d_send, uint8_t *d_recv, const std::vector<uint64_t> &bytes) {
    exchange &X = ctx->xchg;
    rccl_api &A = rccl();
    RT_NCCL(A.GroupStart());
    uint64_t at = 0;
    int bad = 0;                                  // an error inside the group must not leave it open
    for (int r = 0; r < X.nranks && !bad; ++r) {
        if (bytes[r]) bad = A.Broadcast(r == X.rank ? d_send : d_recv + at, d_recv + at, bytes[r], NCCL_UINT8, r, (nccl_comm)X.comm, ctx->stream);
        at += bytes[r];
    }
    const int end = A.GroupEnd();
    if (bad) { set_error(std::string("ncclBroadcast: ") + A.GetErrorString(bad)); return RATTLE_ERR_HIP; }
    RT_NCCL(end);
    return 0;
}

}  // namespace

// RATTLE_XCHG_RECORD (single rank): the file every exchange point appends the job's whole payload to
static FILE *xchg_record_file() {
    static FILE *f = [] {
        const char *path = getenv("RATTLE_XCHG_RECORD");
        FILE *g = path ? fopen(path, "wb") : nullptr;
        if (g) atexit([] { if (FILE *h = xchg_record_file()) fclose(h); });
        return g;
    }();
    return f;
}
bool xchg_recording(const rattle_ctx *ctx) { return ctx->xchg.nranks == 1 && xchg_record_file() != nullptr; }
bool xchg_replaying(const rattle_ctx *ctx) { return ctx->xchg.replay != nullptr; }

int xchg_allgatherv(rattle_ctx *ctx, const std::vector<uint8_t> &mine, std::vector<std::vector<uint8_t>> &all, bool job_payload) {
    exchange &X = ctx->xchg;
    all.assign((size_t)X.nranks, {});
    if (X.nranks == 1) {
        all[0] = mine;
        if (FILE *f = job_payload ? xchg_record_file() : nullptr) {
            const uint64_t n = mine.size();
            if (fwrite(&n, 8, 1, f) != 1 || (n && fwrite(mine.data(), 1, n, f) != n)) { set_error("exchange record: write failed"); return RATTLE_ERR_HIP; }
            fflush(f);
        }
        return 0;
    }
    ++X.calls;
    if (X.replay && !job_payload) { all[(size_t)X.rank] = mine; return 0; }      // (a self-test has no entry in the record)
    if (X.replay) {
        // this rank's own piece, and the recorded whole of the single-rank job in the next rank's place (the caller drops what is its own in it)
        uint64_t n = 0;
        if (fread(&n, 8, 1, X.replay) != 1) { set_error("exchange replay: the record has no entry for this exchange"); return RATTLE_ERR_STATE; }
        std::vector<uint8_t> &whole = all[(size_t)((X.rank + 1) % X.nranks)];
        whole.resize(n);
        if (n && fread(whole.data(), 1, n, X.replay) != n) { set_error("exchange replay: short record"); return RATTLE_ERR_STATE; }
        all[(size_t)X.rank] = mine;
        X.bytes += n;
        return 0;
    }
    std::vector<uint64_t> bytes((size_t)X.nranks, 0);
    const uint64_t my_bytes = mine.size();
    if (X.comm) {
        hipStream_t st = ctx->stream;
        RT_TRY(X.d_sz.reserve((size_t)X.nranks + 1)); RT_TRY(X.h_sz.reserve((size_t)X.nranks + 1));
        X.h_sz.p[X.nranks] = my_bytes;
        RT_HIP(hipMemcpyAsync(X.d_sz.p + X.nranks, X.h_sz.p + X.nranks, 8, hipMemcpyHostToDevice, st));
        RT_NCCL(rccl().AllGather(X.d_sz.p + X.nranks, X.d_sz.p, 1, NCCL_UINT64, (nccl_comm)X.comm, st));
        RT_HIP(hipMemcpyAsync(X.h_sz.p, X.d_sz.p, (size_t)X.nranks * 8, hipMemcpyDeviceToHost, st));
        RT_HIP(hipStreamSynchronize(st));
        uint64_t total = 0;
        for (int r = 0; r < X.nranks; ++r) { bytes[r] = X.h_sz.p[r]; total += bytes[r]; }
        RT_TRY(X.d_send.reserve(my_bytes + 16)); RT_TRY(X.d_recv.reserve(total + 16));
        RT_TRY(X.h_send.reserve(my_bytes + 16)); RT_TRY(X.h_recv.reserve(total + 16));
        if (my_bytes) {
            memcpy(X.h_send.p, mine.data(), my_bytes);
            RT_HIP(hipMemcpyAsync(X.d_send.p, X.h_send.p, my_bytes, hipMemcpyHostToDevice, st));
        }
        RT_TRY(rccl_allgatherv(ctx, X.d_send.p, X.d_recv.p, bytes));
        if (total) RT_HIP(hipMemcpyAsync(X.h_recv.p, X.d_recv.p, total, hipMemcpyDeviceToHost, st));
        RT_HIP(hipStreamSynchronize(st));
        uint64_t at = 0;
        for (int r = 0; r < X.nranks; ++r) { all[r].assign(X.h_recv.p + at, X.h_recv.p + at + bytes[r]); at += bytes[r]; }
        X.bytes += total;
        return 0;
    }
    if (!X.fn) { set_error("several ranks but no exchange transport (rattle_hip_comm_init / rattle_hip_set_exchange)"); return RATTLE_ERR_STATE; }
    std::vector<uint64_t> eight((size_t)X.nranks, 8);
    if (X.fn(X.user, &my_bytes, 8, bytes.data(), eight.data()) != 0) { set_error("exchange callback failed (sizes)"); return RATTLE_ERR_HIP; }
    uint64_t total = 0;
    for (uint64_t b : bytes) total += b;
    std::vector<uint8_t> flat(total + 1);
    static const uint8_t none = 0;
    if (X.fn(X.user, my_bytes ? mine.data() : &none, my_bytes, flat.data(), bytes.data()) != 0) { set_error("exchange callback failed (payload)"); return RATTLE_ERR_HIP; }
    uint64_t at = 0;
    for (int r = 0; r < X.nranks; ++r) { all[r].assign(flat.begin() + at, flat.begin() + at + bytes[r]); at += bytes[r]; }
    X.bytes += total;
    return 0;
}

// gather of one byte string per rank on `root` (nothing elsewhere).  RCCL: send / recv pairs in a group, the root's pieces land
// in ONE host buffer that is not zero-filled first (GBs of corrected reads); host transport: the caller's all-gather-v.
struct gathered {
    std::vector<std::vector<uint8_t>> owned;      // host transport: the all-gather's pieces
    std::unique_ptr<uint8_t[]> flat;              // RCCL: the root's receive buffer
    std::vector<const uint8_t *> p;               // piece r = p[r][0 .. n[r])
    std::vector<size_t> n;
};

static int xchg_gatherv(rattle_ctx *ctx, const std::vector<uint8_t> &mine, int root, gathered &G) {
    exchange &X = ctx->xchg;
    G.p.assign((size_t)X.nranks, nullptr); G.n.assign((size_t)X.nranks, 0);
    if (X.replay) {
        // the ranks' pieces travel through files beside the record: every rank leaves its piece, the root (run last) picks the others up
        const std::string base = X.replay_path + ".gather.";
        if (X.rank != root) {
            FILE *f = fopen((base + std::to_string(X.rank)).c_str(), "wb");
            if (!f || (mine.size() && fwrite(mine.data(), 1, mine.size(), f) != mine.size())) { if (f) fclose(f); set_error("gather replay: cannot write this rank's piece"); return RATTLE_ERR_HIP; }
            fclose(f);
            return 0;
        }
        G.owned.assign((size_t)X.nranks, {});
        for (int r = 0; r < X.nranks; ++r) {
            if (r == root) { G.owned[r] = mine; continue; }
            FILE *f = fopen((base + std::to_string(r)).c_str(), "rb");
            if (!f) { set_error("gather replay: the piece of rank " + std::to_string(r) + " is missing (run the root last)"); return RATTLE_ERR_STATE; }
            fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
            G.owned[r].resize((size_t)n);
            const bool ok = n == 0 || fread(G.owned[r].data(), 1, (size_t)n, f) == (size_t)n;
            fclose(f);
            if (!ok) { set_error("gather replay: short piece"); return RATTLE_ERR_STATE; }
        }
        for (int r = 0; r < X.nranks; ++r) { G.p[r] = G.owned[r].data(); G.n[r] = G.owned[r].size(); }
        return 0;
    }
    if (X.nranks == 1 || !X.comm) {
        RT_TRY(xchg_allgatherv(ctx, mine, G.owned));
        if (X.rank == root) for (int r = 0; r < X.nranks; ++r) { G.p[r] = G.owned[r].data(); G.n[r] = G.owned[r].size(); }
        return 0;
    }
    ++X.calls;
    hipStream_t st = ctx->stream;
    std::vector<uint64_t> bytes((size_t)X.nranks, 0);
    const uint64_t my_bytes = mine.size();
    RT_TRY(X.d_sz.reserve((size_t)X.nranks + 1)); RT_TRY(X.h_sz.reserve((size_t)X.nranks + 1));
    X.h_sz.p[X.nranks] = my_bytes;
    RT_HIP(hipMemcpyAsync(X.d_sz.p + X.nranks, X.h_sz.p + X.nranks, 8, hipMemcpyHostToDevice, st));
    RT_NCCL(rccl().AllGather(X.d_sz.p + X.nranks, X.d_sz.p, 1, NCCL_UINT64, (nccl_comm)X.comm, st));
    RT_HIP(hipMemcpyAsync(X.h_sz.p, X.d_sz.p, (size_t)X.nranks * 8, hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    for (int r = 0; r < X.nranks; ++r) bytes[r] = X.h_sz.p[r];
    rccl_api &A = rccl();
    if (X.rank != root) {
        dbuf<uint8_t> d_send;
        RT_TRY(d_send.reserve(my_bytes + 16));
        if (my_bytes) {
            RT_HIP(hipMemcpyAsync(d_send.p, mine.data(), my_bytes, hipMemcpyHostToDevice, st));
            RT_NCCL(A.Send(d_send.p, my_bytes, NCCL_UINT8, root, (nccl_comm)X.comm, st));
        }
        RT_HIP(hipStreamSynchronize(st));
        return 0;
    }
    uint64_t total = 0;
    for (int r = 0; r < X.nranks; ++r) if (r != root) total += bytes[r];
    dbuf<uint8_t> d_recv;
    RT_TRY(d_recv.reserve(total + 16));
    RT_NCCL(A.GroupStart());
    uint64_t at = 0;
    int bad = 0;
    for (int r = 0; r < X.nranks && !bad; ++r) {
        if (r == root || !bytes[r]) continue;
        bad = A.Recv(d_recv.p + at, bytes[r], NCCL_UINT8, r, (nccl_comm)X.comm, st);
        at += bytes[r];
    }
    const int end = A.GroupEnd();
    if (bad) { set_error(std::string("ncclRecv: ") + A.GetErrorString(bad)); return RATTLE_ERR_HIP; }
    RT_NCCL(end);
    G.flat.reset(new uint8_t[total + 16]);
    if (total) RT_HIP(hipMemcpyAsync(G.flat.get(), d_recv.p, total, hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    at = 0;
    for (int r = 0; r < X.nranks; ++r) {
        if (r == root) { G.p[r] = mine.data(); G.n[r] = mine.size(); continue; }
        G.p[r] = G.flat.get() + at; G.n[r] = bytes[r];
        at += bytes[r];
    }
    X.bytes += total;
    return 0;
}

// ---- (de)serialisation of a correction for the gather -------------------------------------------------
namespace {

template <typename T>
void put(std::vector<uint8_t> &b, const T *p, size_t n) {
    // (insert, not resize + memcpy: a resize zero-fills what the copy then overwrites -- 0.16 s of a rank's 250 MB piece at eight ranks)
    if (n) b.insert(b.end(), (const uint8_t *)p, (const uint8_t *)p + n * sizeof(T));
}
template <typename T>
const T *take(const uint8_t *b, size_t &at, size_t n) {
    const T *p = (const T *)(b + at);
    at += n * sizeof(T);
    return p;
}

// reads of a set as (key, piece index) so the root can order them; sequences are copied by the merge
struct rec_ref { uint64_t key; uint32_t piece, idx; };

struct piece_view {                      // one rank's serialised set
    uint32_t n = 0;
    const int32_t *read_id = nullptr, *cluster_id = nullptr, *n_reads = nullptr;
    const uint32_t *pack = nullptr;
    const uint64_t *off = nullptr;
    const char *seq = nullptr, *qual = nullptr;
};

void put_set(std::vector<uint8_t> &b, const rattle_read_set &S, const uint32_t *pack) {
    const uint64_t n = S.n;
    b.reserve(b.size() + 16 + (n + 1) * 8 + n * 16 + 2 * S.off[n] + 8);      // one growth, not a doubling per array
    put(b, &n, 1);
    put(b, S.off, n + 1);                // 8-byte fields first: every array stays naturally aligned
    put(b, S.read_id, n); put(b, S.cluster_id, n); put(b, S.n_reads, n);
    std::vector<uint32_t> none;
    if (!pack) { none.assign(n, 0); pack = none.data(); }
    put(b, pack, n);                     // 4 x 4n bytes: still 8-byte aligned
    const uint64_t tot = S.off[n];
    put(b, S.seq, tot); put(b, S.qual, tot);
    const uint64_t pad = (8 - (2 * tot) % 8) % 8;
    const char z[8] = {0};
    put(b, z, pad);
}

piece_view take_set(const uint8_t *b, size_t &at) {
    piece_view V;
    const uint64_t n = *take<uint64_t>(b, at, 1);
    V.n = (uint32_t)n;
    V.off = take<uint64_t>(b, at, n + 1);
    V.read_id = take<int32_t>(b, at, n); V.cluster_id = take<int32_t>(b, at, n); V.n_reads = take<int32_t>(b, at, n);
    V.pack = take<uint32_t>(b, at, n);
    const uint64_t tot = V.off[n];
    V.seq = take<char>(b, at, tot); V.qual = take<char>(b, at, tot);
    at += (8 - (2 * tot) % 8) % 8;
    return V;
}

void alloc_set(rattle_read_set &S, uint32_t n, uint64_t tot) {
    S.n = n;
    const size_t m = std::max<size_t>(1, n);
    S.read_id = (int32_t *)malloc(m * 4); S.cluster_id = (int32_t *)malloc(m * 4); S.n_reads = (int32_t *)malloc(m * 4);
    S.off = (uint64_t *)malloc(((size_t)n + 1) * 8);
    S.seq = (char *)malloc(tot + 1); S.qual = (char *)malloc(tot + 1);
    S.seq[tot] = 0; S.qual[tot] = 0;
}

// merge the pieces' records in key order (stable: equal keys keep piece order, then record order)
void merge_sets(const std::vector<piece_view> &V, rattle_read_set &S, uint32_t **pack_out) {
    std::vector<rec_ref> refs;
    uint64_t tot = 0;
    for (uint32_t p = 0; p < V.size(); ++p) {
        for (uint32_t i = 0; i < V[p].n; ++i) {
            // members of packs that never entered the queue come first (the reference pushes them while it builds the packs)
            const uint64_t key = V[p].pack[i] == 0xFFFFFFFFu ? 0 : 1 + (uint64_t)V[p].pack[i];
            refs.push_back(rec_ref{key, p, i});
        }
        tot += V[p].n ? V[p].off[V[p].n] : 0;
    }
    // A rank's records arrive in pack order already (round 5: the root's merge of 1e6 reads took 0.5 s, a third of it this sort):
    // a counting sort over the keys (pack indices) is stable and linear
    {
        uint64_t kmax = 0;
        for (const rec_ref &r : refs) kmax = std::max(kmax, r.key);
        if (kmax < (1ull << 26) && refs.size() > 4096) {
            std::vector<uint32_t> start(kmax + 2, 0);
            for (const rec_ref &r : refs) ++start[r.key + 1];
            for (size_t k = 1; k < start.size(); ++k) start[k] += start[k - 1];
            std::vector<rec_ref> sorted(refs.size());
            for (const rec_ref &r : refs) sorted[start[r.key]++] = r;          // pieces in rank order, records in their order: stable
            refs.swap(sorted);
        } else std::stable_sort(refs.begin(), refs.end(), [](const rec_ref &a, const rec_ref &b) { return a.key < b.key; });
    }
    alloc_set(S, (uint32_t)refs.size(), tot);
    uint32_t *pk = (uint32_t *)malloc(std::max<size_t>(1, refs.size()) * 4);
    uint64_t at = 0;
    for (size_t i = 0; i < refs.size(); ++i) {
        const piece_view &P = V[refs[i].piece];
        const uint32_t j = refs[i].idx;
        S.off[i] = at;
        at += P.off[j + 1] - P.off[j];
    }
    const size_t chunk = 4096;
    parallel_for((refs.size() + chunk - 1) / chunk, refs.size() > 8 * chunk ? 32 : 1, [&](size_t c) {      // 32 host threads: 2 x 2 GB of copies into fresh pages
        for (size_t i = c * chunk; i < std::min(refs.size(), (c + 1) * chunk); ++i) {
            const piece_view &P = V[refs[i].piece];
            const uint32_t j = refs[i].idx;
            const uint64_t len = P.off[j + 1] - P.off[j];
            S.read_id[i] = P.read_id[j]; S.cluster_id[i] = P.cluster_id[j]; S.n_reads[i] = P.n_reads[j];
            pk[i] = P.pack[j];
            memcpy(S.seq + S.off[i], P.seq + P.off[j], len); memcpy(S.qual + S.off[i], P.qual + P.off[j], len);
        }
    });
    S.off[refs.size()] = at;
    if (pack_out) *pack_out = pk; else free(pk);
}

}  // namespace

int correction_gather(rattle_ctx *ctx, const rattle_correction *L, int root, rattle_correction **merged) {
    exchange &X = ctx->xchg;
    *merged = nullptr;
    if (root < 0 || root >= X.nranks) { set_error("root out of range"); return RATTLE_ERR_ARG; }
    std::vector<uint8_t> mine;
    phase_timer T_ser("gather: serialise this rank's share");
    // (the root's own share never travels: it is merged from where it lies -- serialising it was 0.15 s of the root's time at eight ranks,
    // 0.6 s at two -- and an empty piece goes into the transport in its place)
    if (X.rank != root) {
    put_set(mine, L->corrected, L->corrected_pack);
    put_set(mine, L->uncorrected, L->uncorrected_pack);
    {
        const rattle_skip_list &K = L->skipped;
        const uint64_t n = K.n, tot = K.read_off[K.n];
        put(mine, &n, 1); put(mine, K.read_off, n + 1);
        put(mine, K.cluster_id, n); put(mine, K.pack, n); put(mine, K.stage, n); put(mine, K.read_id, tot);
        if ((3 * n + tot) & 1) { const uint32_t z = 0; put(mine, &z, 1); }
        put(mine, L->counters, 8);
    }
    }
    T_ser.stop();
    gathered G;
    {
        phase_timer T_tr("gather: transport");
        RT_TRY(xchg_gatherv(ctx, mine, root, G));
    }
    if (X.rank != root) return 0;
    phase_timer T_merge("gather: merge on the root");
    rattle_correction *R = (rattle_correction *)calloc(1, sizeof(rattle_correction));
    std::vector<piece_view> cor, unc;
    struct skip_ref { int32_t cid; uint32_t pack, stage; const int32_t *rid; uint64_t n; };
    std::vector<skip_ref> sk;
    uint64_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> no_pack_c, no_pack_u;
    auto view_of = [](const rattle_read_set &S, const uint32_t *pack, std::vector<uint32_t> &none) {
        piece_view V;
        V.n = S.n; V.read_id = S.read_id; V.cluster_id = S.cluster_id; V.n_reads = S.n_reads; V.off = S.off; V.seq = S.seq; V.qual = S.qual;
        if (!pack) { none.assign(S.n, 0); pack = none.data(); }
        V.pack = pack;
        return V;
    };
    for (int r = 0; r < X.nranks; ++r) {
        if (r == root) {                         // this rank's own share, in place
            cor.push_back(view_of(L->corrected, L->corrected_pack, no_pack_c));
            unc.push_back(view_of(L->uncorrected, L->uncorrected_pack, no_pack_u));
            const rattle_skip_list &K = L->skipped;
            for (uint64_t i = 0; i < K.n; ++i) sk.push_back(skip_ref{K.cluster_id[i], K.pack[i], K.stage[i], K.read_id + K.read_off[i], K.read_off[i + 1] - K.read_off[i]});
            cnt[0] += L->counters[0]; cnt[1] += L->counters[1]; cnt[2] = L->counters[2]; cnt[3] += L->counters[3]; cnt[4] += L->counters[4];
            cnt[5] += L->counters[5]; cnt[6] += L->counters[6]; cnt[7] += L->counters[7];
            continue;
        }
        size_t at = 0;
        const uint8_t *b = G.p[r];
        cor.push_back(take_set(b, at));
        unc.push_back(take_set(b, at));
        const uint64_t n = *take<uint64_t>(b, at, 1);
        const uint64_t *ro = take<uint64_t>(b, at, n + 1);
        const int32_t *cid = take<int32_t>(b, at, n);
        const uint32_t *pk = take<uint32_t>(b, at, n), *stg = take<uint32_t>(b, at, n);
        const uint64_t tot = ro[n];
        const int32_t *rid = take<int32_t>(b, at, tot);
        if ((3 * n + tot) & 1) take<uint32_t>(b, at, 1);
        const uint64_t *c8 = take<uint64_t>(b, at, 8);
        for (uint64_t i = 0; i < n; ++i) sk.push_back(skip_ref{cid[i], pk[i], stg[i], rid + ro[i], ro[i + 1] - ro[i]});
        cnt[0] += c8[0]; cnt[1] += c8[1];                 // DP cells and alignments add up; the pack count is global already
        cnt[2] = c8[2];
        cnt[3] += c8[3]; cnt[4] += c8[4];
        cnt[5] += c8[5]; cnt[6] += c8[6]; cnt[7] += c8[7];      // cells computed, certified bands, failed certificates add up as well
    }
    merge_sets(cor, R->corrected, &R->corrected_pack);
    merge_sets(unc, R->uncorrected, &R->uncorrected_pack);
    // consensi are complete on every rank
    {
        const rattle_read_set &S = L->consensi;
        alloc_set(R->consensi, S.n, S.off[S.n]);
        memcpy(R->consensi.read_id, S.read_id, (size_t)S.n * 4); memcpy(R->consensi.cluster_id, S.cluster_id, (size_t)S.n * 4);
        memcpy(R->consensi.n_reads, S.n_reads, (size_t)S.n * 4); memcpy(R->consensi.off, S.off, ((size_t)S.n + 1) * 8);
        memcpy(R->consensi.seq, S.seq, S.off[S.n]); memcpy(R->consensi.qual, S.qual, S.off[S.n]);
    }
    // skip list ordered by (cluster, pack, stage)
    std::stable_sort(sk.begin(), sk.end(), [](const skip_ref &a, const skip_ref &b) {
        return a.cid != b.cid ? a.cid < b.cid : (a.pack != b.pack ? a.pack < b.pack : a.stage < b.stage);
    });
    {
        rattle_skip_list &K = R->skipped;
        const size_t n = sk.size();
        K.n = (uint32_t)n;
        K.cluster_id = (int32_t *)malloc(std::max<size_t>(1, n) * 4); K.pack = (uint32_t *)malloc(std::max<size_t>(1, n) * 4);
        K.stage = (uint32_t *)malloc(std::max<size_t>(1, n) * 4); K.read_off = (uint64_t *)malloc((n + 1) * 8);
        uint64_t tot = 0;
        for (size_t i = 0; i < n; ++i) { K.cluster_id[i] = sk[i].cid; K.pack[i] = sk[i].pack; K.stage[i] = sk[i].stage; K.read_off[i] = tot; tot += sk[i].n; }
        K.read_off[n] = tot;
        K.read_id = (int32_t *)malloc(std::max<uint64_t>(1, tot) * 4);
        for (size_t i = 0; i < n; ++i) if (sk[i].n) memcpy(K.read_id + K.read_off[i], sk[i].rid, sk[i].n * 4);
    }
    memcpy(R->counters, cnt, sizeof(cnt));
    *merged = R;
    return 0;
}

}  // namespace rattle

using namespace rattle;

extern "C" {

int rattle_hip_set_exchange(rattle_ctx *c, int rank, int nranks, rattle_allgatherv_fn fn, void *user) {
    const char *replay = getenv("RATTLE_XCHG_REPLAY");      // (measurement aid: this rank alone, its peers' payloads from a record; common.h)
    if (!c || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !fn && !replay)) { set_error("bad exchange arguments"); return RATTLE_ERR_ARG; }
    if (c->xchg.comm) { set_error("an RCCL communicator is attached: rattle_hip_comm_destroy first"); return RATTLE_ERR_STATE; }
    if (c->xchg.replay) { fclose(c->xchg.replay); c->xchg.replay = nullptr; }
    c->xchg.replay_path.clear();
    if (nranks > 1 && !fn) {
        c->xchg.replay = fopen(replay, "rb");
        if (!c->xchg.replay) { set_error(std::string("exchange replay: cannot open ") + replay); return RATTLE_ERR_ARG; }
        c->xchg.replay_path = replay;
    }
    c->xchg.rank = rank; c->xchg.nranks = nranks; c->xchg.fn = fn; c->xchg.user = user;
    return 0;
}

int rattle_hip_comm_unique_id(uint8_t *id_out) {
    if (!id_out) { set_error("null argument"); return RATTLE_ERR_ARG; }
    RT_TRY(load_rccl());
    nccl_uid id;
    RT_NCCL(rccl().GetUniqueId(&id));
    memcpy(id_out, id.internal, RATTLE_COMM_ID_BYTES);
    return 0;
}

int rattle_hip_comm_init(rattle_ctx *c, int rank, int nranks, const uint8_t *id) {
    if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) { set_error("bad communicator arguments"); return RATTLE_ERR_ARG; }
    if (c->xchg.comm) { set_error("communicator already attached"); return RATTLE_ERR_STATE; }
    if (c->device < 0) { set_error("an RCCL communicator needs a device context"); return RATTLE_ERR_STATE; }
    RT_TRY(load_rccl());
    RT_HIP(hipSetDevice(c->device));
    nccl_uid u;
    memcpy(u.internal, id, RATTLE_COMM_ID_BYTES);
    nccl_comm comm = nullptr;
    RT_NCCL(rccl().CommInitRank(&comm, nranks, u, rank));
    c->xchg.comm = comm; c->xchg.rank = rank; c->xchg.nranks = nranks; c->xchg.fn = nullptr; c->xchg.user = nullptr;
    return 0;
}

int rattle_hip_comm_destroy(rattle_ctx *c) {
    if (!c) { set_error("null ctx"); return RATTLE_ERR_ARG; }
    if (c->xchg.comm) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)rccl().CommDestroy((nccl_comm)c->xchg.comm);
    }
    c->xchg.reset();
    return 0;
}

int rattle_hip_comm_stats(rattle_ctx *c, uint64_t *calls, uint64_t *bytes) {
    if (!c) { set_error("null ctx"); return RATTLE_ERR_ARG; }
    if (calls) *calls = c->xchg.calls;
    if (bytes) *bytes = c->xchg.bytes;
    return 0;
}

int rattle_hip_comm_probe(rattle_ctx *c) {
    if (!c) { set_error("null ctx"); return RATTLE_ERR_ARG; }
    exchange &X = c->xchg;
    if (c->device >= 0) RT_HIP(hipSetDevice(c->device));
    // ragged pieces (one of them empty) through both exchange shapes the sharded paths use
    std::vector<uint8_t> mine((size_t)((X.rank * 1000 + 7) % 2501) * (X.rank % 3 != 1), (uint8_t)(X.rank + 1));
    auto check = [&](const std::vector<const uint8_t *> &p, const std::vector<size_t> &n) -> int {
        for (int r = 0; r < X.nranks; ++r) {
            const size_t want = (size_t)((r * 1000 + 7) % 2501) * (r % 3 != 1);
            bool ok = n[r] == want;
            for (size_t i = 0; ok && i < want; ++i) ok = p[r][i] == (uint8_t)(r + 1);
            if (!ok) { set_error("exchange probe: the piece of rank " + std::to_string(r) + " arrived damaged on rank " + std::to_string(X.rank)); return RATTLE_ERR_HIP; }
        }
        return 0;
    };
    if (X.replay) return 0;                    // the replay aid has no peers to probe
    std::vector<std::vector<uint8_t>> all;
    RT_TRY(xchg_allgatherv(c, mine, all, false));
    {
        std::vector<const uint8_t *> p; std::vector<size_t> n;
        for (const std::vector<uint8_t> &v : all) { p.push_back(v.data()); n.push_back(v.size()); }
        RT_TRY(check(p, n));
    }
    if (X.nranks > 1) {
        gathered G;
        RT_TRY(xchg_gatherv(c, mine, X.nranks - 1, G));
        if (X.rank == X.nranks - 1) RT_TRY(check(G.p, G.n));
    }
    return 0;
}

int rattle_hip_correction_gather(rattle_ctx *c, const rattle_correction *local, int root, rattle_correction **merged) {
    if (!c || !local || !merged) { set_error("null argument"); return RATTLE_ERR_ARG; }
    if (c->device >= 0) RT_HIP(hipSetDevice(c->device));
    return correction_gather(c, local, root, merged);
}

int rattle_hip_plan_packs(const uint64_t *off, uint32_t n_reads, uint32_t n_clusters, const uint32_t *coff, const int32_t *mid,
                          const uint8_t *mrev, const rattle_correct_params *P, int nranks, rattle_pack_plan **out) {
    if (!off || !P || !out || nranks < 1 || (n_clusters && (!coff || !mid || !mrev))) { set_error("null argument"); return RATTLE_ERR_ARG; }
    *out = nullptr;
    pack_plan PL;
    RT_TRY(plan_packs(off, n_reads, n_clusters, coff, mid, mrev, P, nranks, PL));
    rattle_pack_plan *R = (rattle_pack_plan *)calloc(1, sizeof(rattle_pack_plan));
    const size_t np = PL.pk_cid.size(), nm = PL.members.size(), ns = PL.small.size();
    R->n_packs = (uint32_t)np;
    R->pack_first = (uint32_t *)malloc((np + 1) * 4);
    memcpy(R->pack_first, PL.first.data(), (np + 1) * 4);
    R->member_id = (int32_t *)malloc(std::max<size_t>(1, nm) * 4); R->member_rev = (uint8_t *)malloc(std::max<size_t>(1, nm));
    for (size_t i = 0; i < nm; ++i) { R->member_id[i] = PL.members[i].rid; R->member_rev[i] = PL.members[i].rev; }
    R->pack_cluster = (int32_t *)malloc(std::max<size_t>(1, np) * 4); R->pack_local = (uint32_t *)malloc(std::max<size_t>(1, np) * 4);
    R->pack_cost = (uint64_t *)malloc(std::max<size_t>(1, np) * 8); R->pack_owner = (uint32_t *)malloc(std::max<size_t>(1, np) * 4);
    if (np) {
        memcpy(R->pack_cluster, PL.pk_cid.data(), np * 4); memcpy(R->pack_local, PL.pk_local.data(), np * 4);
        memcpy(R->pack_cost, PL.pk_cost.data(), np * 8); memcpy(R->pack_owner, PL.pk_owner.data(), np * 4);
    }
    R->n_unqueued = (uint32_t)ns;
    R->unqueued_id = (int32_t *)malloc(std::max<size_t>(1, ns) * 4); R->unqueued_cluster = (int32_t *)malloc(std::max<size_t>(1, ns) * 4);
    for (size_t i = 0; i < ns; ++i) { R->unqueued_id[i] = PL.small[i].rid; R->unqueued_cluster[i] = PL.small_cid[i]; }
    *out = R;
    return 0;
}

void rattle_hip_pack_plan_free(rattle_pack_plan *p) {
    if (!p) return;
    free(p->pack_first); free(p->member_id); free(p->member_rev); free(p->pack_cluster); free(p->pack_local);
    free(p->pack_cost); free(p->pack_owner); free(p->unqueued_id); free(p->unqueued_cluster);
    free(p);
}

int rattle_hip_lpt_assign(const uint64_t *cost, uint32_t n, int nranks, uint32_t *owner_out) {
    if ((n && (!cost || !owner_out)) || nranks < 1) { set_error("bad argument"); return RATTLE_ERR_ARG; }
    std::vector<uint64_t> c(cost, cost + n);
    std::vector<uint32_t> o;
    lpt_assign(c, nranks, o);
    if (n) memcpy(owner_out, o.data(), (size_t)n * 4);
    return 0;
}

}  // extern "C"
