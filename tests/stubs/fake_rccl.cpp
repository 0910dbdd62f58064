// TEST DOUBLE for librccl.so: the ten entry points rattle_amd/csrc/exchange.hip binds, carried over files in a
// shared directory (FAKE_RCCL_DIR) so that several ranks can sit on ONE GPU -- real RCCL refuses two ranks on
// a device ("Duplicate GPU detected"), and the test boxes have one.  It checks what a real run would trip
// over: the order and pairing of the calls, group semantics (nothing moves before ncclGroupEnd), counts,
// roots and the device pointers handed in.  Synchronous: the stream is drained, data goes through the host.
// Never loaded by the product unless RATTLE_RCCL_LIB points at it (tests/test_gpu_dist.py does).
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct comm { int nranks, rank; std::string dir; uint64_t seq = 0; std::vector<uint64_t> sent, got; };
struct op { int kind; const void *src; void *dst; size_t bytes; int peer; comm *c; hipStream_t st; };   // 0 allgather 1 bcast 2 send 3 recv
int depth = 0;
std::vector<op> queue;
const size_t elt[] = {1, 1, 4, 4, 8, 8, 2, 4, 8};

bool publish(const std::string &path, const void *dev, size_t bytes) {
    std::vector<char> h(bytes);
    if (bytes && hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
    const std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    if (bytes && fwrite(h.data(), 1, bytes, f) != bytes) { fclose(f); return false; }
    fclose(f);
    return rename(tmp.c_str(), path.c_str()) == 0;
}

bool collect(const std::string &path, void *dev, size_t bytes) {
    struct stat sb;
    for (int spin = 0; stat(path.c_str(), &sb) != 0; ++spin) {
        if (spin > 600000) { fprintf(stderr, "fake_rccl: no peer wrote %s\n", path.c_str()); return false; }     // 10 min
        usleep(1000);
    }
    if ((size_t)sb.st_size != bytes) { fprintf(stderr, "fake_rccl: %s holds %zu bytes, receiver expects %zu\n", path.c_str(), (size_t)sb.st_size, bytes); return false; }
    std::vector<char> h(bytes);
    FILE *f = fopen(path.c_str(), "rb");
    if (!f || (bytes && fread(h.data(), 1, bytes, f) != bytes)) return false;
    fclose(f);
    return !bytes || hipMemcpy(dev, h.data(), bytes, hipMemcpyHostToDevice) == hipSuccess;
}

int run(std::vector<op> &ops) {
    for (op &o : ops) if (hipStreamSynchronize(o.st) != hipSuccess) return 1;
    // collectives take their sequence number in call order (the same on every rank, or the job hangs -- as it would on RCCL)
    std::vector<std::string> name(ops.size());
    for (size_t i = 0; i < ops.size(); ++i) {
        op &o = ops[i];
        comm *c = o.c;
        if (o.kind <= 1) name[i] = c->dir + "/c" + std::to_string(c->seq++);
        else if (o.kind == 2) name[i] = c->dir + "/p" + std::to_string(c->rank) + "_" + std::to_string(o.peer) + "_" + std::to_string(c->sent[o.peer]++);
        else name[i] = c->dir + "/p" + std::to_string(o.peer) + "_" + std::to_string(c->rank) + "_" + std::to_string(c->got[o.peer]++);
    }
    for (size_t i = 0; i < ops.size(); ++i) {                   // everything this rank gives
        op &o = ops[i];
        if (o.kind == 0 && !publish(name[i] + "_r" + std::to_string(o.c->rank), o.src, o.bytes)) return 1;
        if (o.kind == 1 && o.peer == o.c->rank && !publish(name[i] + "_root" + std::to_string(o.peer), o.src, o.bytes)) return 1;
        if (o.kind == 2 && !publish(name[i], o.src, o.bytes)) return 1;
    }
    for (size_t i = 0; i < ops.size(); ++i) {                   // everything it takes
        op &o = ops[i];
        if (o.kind == 0) for (int r = 0; r < o.c->nranks; ++r) if (!collect(name[i] + "_r" + std::to_string(r), (char *)o.dst + (size_t)r * o.bytes, o.bytes)) return 1;
        if (o.kind == 1 && !collect(name[i] + "_root" + std::to_string(o.peer), o.dst, o.bytes)) return 1;   // the root too: send -> recv buffer
        if (o.kind == 3 && !collect(name[i], o.dst, o.bytes)) return 1;
    }
    return 0;
}

int submit(const op &o) {
    if (!o.c || (o.bytes && ((o.kind != 3 && !o.src && !(o.kind == 1 && o.peer != o.c->rank)) || (o.kind != 2 && !o.dst)))) { fprintf(stderr, "fake_rccl: null buffer in op %d\n", o.kind); return 4; }
    if (o.kind >= 1 && (o.peer < 0 || o.peer >= o.c->nranks)) { fprintf(stderr, "fake_rccl: peer %d out of range\n", o.peer); return 4; }
    if (o.kind >= 2 && o.peer == o.c->rank) { fprintf(stderr, "fake_rccl: send/recv to self\n"); return 4; }
    if (depth) { queue.push_back(o); return 0; }
    std::vector<op> one{o};
    return run(one);
}

}  // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "job%ld_%ld", (long)getpid(), (long)random());
    return 0;
}

int ncclCommInitRank(void **out, int nranks, ncclUniqueId id, int rank) {
    const char *base = getenv("FAKE_RCCL_DIR");
    if (!base || rank < 0 || rank >= nranks) return 4;
    comm *c = new comm{nranks, rank, std::string(base) + "/" + id.internal};
    c->sent.assign(nranks, 0); c->got.assign(nranks, 0);
    mkdir(c->dir.c_str(), 0700);
    *out = c;
    return 0;
}

int ncclCommDestroy(void *c) { delete (comm *)c; return 0; }
int ncclGroupStart() { ++depth; return 0; }
int ncclGroupEnd() {
    if (depth <= 0) return 4;
    if (--depth) return 0;
    std::vector<op> ops;
    ops.swap(queue);
    return run(ops);
}
int ncclAllGather(const void *s, void *r, size_t count, int type, void *c, hipStream_t st) { return submit({0, s, r, count * elt[type], 0, (comm *)c, st}); }
int ncclBroadcast(const void *s, void *r, size_t count, int type, int root, void *c, hipStream_t st) { return submit({1, s, r, count * elt[type], root, (comm *)c, st}); }
int ncclSend(const void *s, size_t count, int type, int peer, void *c, hipStream_t st) { return submit({2, s, nullptr, count * elt[type], peer, (comm *)c, st}); }
int ncclRecv(void *r, size_t count, int type, int peer, void *c, hipStream_t st) { return submit({3, nullptr, r, count * elt[type], peer, (comm *)c, st}); }
const char *ncclGetErrorString(int e) { return e == 0 ? "ok" : e == 4 ? "fake_rccl: invalid argument" : "fake_rccl: transfer failed"; }

}
