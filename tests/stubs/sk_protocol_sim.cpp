// The mailbox protocol of kernel C's skewed wavefront pipeline (rattle_amd/csrc/poa.hip: dp_rows_sk) restated on host threads:
// one thread per wavefront, std::atomic (sequentially consistent) standing for LDS operations that complete in a wavefront's
// issue order.  What a wavefront READS from its left neighbour's mailbox is checked against what that neighbour must have
// written for exactly that row; a protocol that lets a reader run ahead of its writer, lets a writer overwrite a slot still
// needed, or deadlocks (the run is bounded by the caller's timeout) fails here without a GPU.
// usage: sk_protocol_sim NW N_ACT ROWS D RING FMT SEED      (FMT 0: record words in the ring, 1: ready-made terms)
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

static int NW, N_ACT, ROWS, D, RING, FMT;
struct mail { std::atomic<uint32_t> cnt{0}; std::vector<std::atomic<uint32_t>> T, H; mail() {} };
static std::vector<mail> M;
static std::atomic<int> bad{0};

static uint32_t valT(int w, uint32_t r) { return 0x10000000u + (uint32_t)w * 0x01000000u + r * 2654435761u % 0x00FFFFFFu; }
static uint32_t valH(int w, uint32_t r) { return 0x70000000u + (uint32_t)w * 0x01000000u + r * 40503u % 0x00FFFFFFu; }
#define CHECK(c, what) do { if (!(c)) { if (!bad.exchange(1)) fprintf(stderr, "wave %d row %u: %s\n", w, row, what); return; } } while (0)

static void wave(int w, uint32_t seed) {
    std::mt19937 rng(seed * 977u + (uint32_t)w);
    const bool act = w < N_ACT, has_left = w > 0, has_right = w + 1 < N_ACT;
    if (!act) return;
    const uint32_t KEEP = FMT ? 0u : (uint32_t)RING;
    mail *L = has_left ? &M[w - 1] : nullptr, *R = has_right ? &M[w + 1] : nullptr, *me = &M[w];
    auto jitter = [&]() { const uint32_t x = rng() & 63u; if (x == 0) std::this_thread::yield(); else if (x < 3) for (volatile int i = 0; i < 200; ++i) {} };
    for (uint32_t row = 1; row <= (uint32_t)ROWS && !bad; ++row) {
        const uint32_t mslot = row % (uint32_t)D;
        uint32_t cl = 0, cr = 0, leT = 0, leH = 0;
        if (has_left) { cl = L->cnt.load(); leT = L->T[mslot].load(); leH = L->H[mslot].load(); }
        if (has_right) cr = R->cnt.load();
        auto wait_left = [&](uint32_t need) {
            while ((int32_t)(cl - need) < 0) { std::this_thread::yield(); cl = L->cnt.load(); leT = L->T[mslot].load(); leH = L->H[mslot].load(); if (bad) return; }
        };
        jitter();
        if (FMT == 0 && has_left) wait_left(row);
        if (FMT == 0 && has_left) {
            // the Hl of the ring predecessors, the previous row among them
            for (uint32_t d = 1; d <= (uint32_t)RING && d < row; ++d) {
                if ((rng() & 3u) && d != 1) continue;
                const uint32_t p = row - d;
                CHECK(L->H[p % (uint32_t)D].load() == valH(w - 1, p), "stale or overwritten Hl of a ring predecessor");
            }
        }
        jitter();
        if (has_left) {
            if (FMT == 1) wait_left(row + 1);
            if (bad) return;
            CHECK(leT == valT(w - 1, row), "prefix of the left wavefront is not this row's");
            if (FMT == 1) CHECK(leH == valH(w - 1, row), "Hl of the left wavefront is not this row's");
        }
        if (has_right) {
            while ((int32_t)(cr + ((uint32_t)D - KEEP) - row) < 0) { std::this_thread::yield(); cr = R->cnt.load(); if (bad) return; }
            me->T[mslot].store(valT(w, row));
        }
        if (N_ACT > 1) me->cnt.store(row);
        jitter();
        if (has_right) me->H[mslot].store(valH(w, row));
    }
    if (N_ACT > 1) me->cnt.store((uint32_t)ROWS + 1u);
}

int main(int argc, char **argv) {
    if (argc < 8) return 2;
    NW = atoi(argv[1]); N_ACT = atoi(argv[2]); ROWS = atoi(argv[3]); D = atoi(argv[4]); RING = atoi(argv[5]); FMT = atoi(argv[6]);
    const uint32_t seed = (uint32_t)atoi(argv[7]);
    M = std::vector<mail>(NW);
    for (auto &m : M) { m.T = std::vector<std::atomic<uint32_t>>(D); m.H = std::vector<std::atomic<uint32_t>>(D); }
    std::vector<std::thread> th;
    for (int w = 0; w < NW; ++w) th.emplace_back(wave, w, seed);
    for (auto &t : th) t.join();
    if (bad) return 1;
    puts("SK_PROTOCOL_OK");
    return 0;
}
