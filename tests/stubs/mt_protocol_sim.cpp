// The synchronisation of kernel C's teams of wavefronts (rattle_amd/csrc/poa.hip: dp_rows_mt) restated on host threads: one thread
// per wavefront (team t, column block w), std::atomic (sequentially consistent) standing for LDS operations that complete in a
// wavefront's issue order.  Rows are dealt round-robin to the teams; a row reads the ring entries of its predecessor rows in its
// own column block (after "every row <= need is final here": min over the teams' done counters + T > need), takes the prefix of
// the same row from the column block to its left (mailbox of MD entries per wavefront, back-pressure on the right neighbour's
// done counter), writes its own ring entry, then its done counter.  Every ring entry a row reads must hold exactly the
// predecessor row it wants (not an older row: read too early; not a newer one: overwritten while needed), every mailbox entry
// exactly this row's, and nobody may wait forever (the caller's timeout bounds the run).
// usage: mt_protocol_sim NW N_ACT T ROWS SLOTS SLACK MD SEED [REACH]      (REACH: rows a reader looks back; the kernel's own: SLOTS - SLACK)
#include <atomic>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

static int NW, N_ACT, T, ROWS, SLOTS, SLACK, MD;
static const int32_t BIG = 0x7FFFFF00;
struct row_plan { std::vector<uint32_t> preds; int32_t need; };
static std::vector<row_plan> P;                               // 1-based rows
static std::vector<std::atomic<int32_t>> done_;               // [w][4]
static std::vector<std::atomic<uint32_t>> mcnt, mailT, mailH; // [w][t], [w][t][MD]
static std::vector<std::atomic<uint32_t>> ring;               // [w][slot]: the row whose entry is there
static std::atomic<int> bad{0};

static uint32_t valT(int w, uint32_t r) { return 0x10000000u + (uint32_t)w * 0x01000000u + r * 2654435761u % 0x00FFFFFFu; }
static uint32_t valH(int w, uint32_t r) { return 0x70000000u + (uint32_t)w * 0x01000000u + r * 40503u % 0x00FFFFFFu; }
#define CHECK(c, what) do { if (!(c)) { if (!bad.exchange(1)) fprintf(stderr, "team %d block %d row %u: %s\n", t, w, row, what); return; } } while (0)

static void wave(int t, int w, uint32_t seed) {
    std::mt19937 rng(seed * 977u + (uint32_t)(t * 16 + w));
    if (w >= N_ACT) return;
    const bool has_left = w > 0, has_right = w + 1 < N_ACT;
    auto jitter = [&]() { const uint32_t x = rng() & 63u; if (x == 0) std::this_thread::yield(); else if (x < 4) for (volatile int i = 0; i < 300; ++i) {} };
    uint32_t mslot = 0;
    for (uint32_t row = (uint32_t)t + 1; row <= (uint32_t)ROWS && !bad; row += (uint32_t)T) {
        const row_plan &pl = P[row];
        if (T > 1) {
            while (true) {
                int32_t m = BIG;
                for (int q = 0; q < 4; ++q) m = std::min(m, done_[w * 4 + q].load());
                if (m + T > pl.need) break;
                std::this_thread::yield();
                if (bad) return;
            }
        }
        jitter();
        for (uint32_t p : pl.preds) CHECK(ring[(size_t)w * (SLOTS + 1) + p % (uint32_t)SLOTS].load() == p, "a predecessor's ring entry is not that row's (read too early, or overwritten)");
        jitter();
        if (has_left) {
            while ((int32_t)(mcnt[(w - 1) * 4 + t].load() - row) < 0) { std::this_thread::yield(); if (bad) return; }
            CHECK(mailT[((size_t)(w - 1) * 4 + t) * MD + mslot].load() == valT(w - 1, row), "prefix of the left column block is not this row's");
            CHECK(mailH[((size_t)(w - 1) * 4 + t) * MD + mslot].load() == valH(w - 1, row), "H of the left column block is not this row's");
        }
        jitter();
        if (has_right) {
            while (done_[(w + 1) * 4 + t].load() + MD * T < (int32_t)row) { std::this_thread::yield(); if (bad) return; }
            mailT[((size_t)w * 4 + t) * MD + mslot].store(valT(w, row));
            mailH[((size_t)w * 4 + t) * MD + mslot].store(valH(w, row));
            mcnt[w * 4 + t].store(row);
        }
        jitter();
        ring[(size_t)w * (SLOTS + 1) + row % (uint32_t)SLOTS].store(row);
        done_[w * 4 + t].store((int32_t)row);
        mslot = (mslot + 1u) % (uint32_t)MD;
    }
    done_[w * 4 + t].store(BIG);
}

int main(int argc, char **argv) {
    if (argc < 9) return 2;
    NW = atoi(argv[1]); N_ACT = atoi(argv[2]); T = atoi(argv[3]); ROWS = atoi(argv[4]); SLOTS = atoi(argv[5]); SLACK = atoi(argv[6]); MD = atoi(argv[7]);
    const uint32_t seed = (uint32_t)atoi(argv[8]);
    const int reach = argc > 9 ? atoi(argv[9]) : SLOTS - SLACK;
    if (T < 1 || T > 4 || reach < 1 || (T > 1 && SLACK < 1)) return 2;
    // the plan the kernel's step 2 would build: in-ring predecessors (distance <= reach) and the row to wait for
    std::mt19937 rng(seed);
    P.resize(ROWS + 1);
    for (int r = 1; r <= ROWS; ++r) {
        const int n_in = (rng() % 10u) < 5 ? 1 : 1 + (int)(rng() % 8u);
        uint32_t need_row = 0;
        for (int k = 0; k < n_in; ++k) {
            int d = 1;
            while ((rng() & 3u) != 0 && d < 40) ++d;                    // geometric, mean 4: a third of the rows have the row before them as a predecessor
            if (d >= r) continue;
            if (d <= reach) { P[r].preds.push_back((uint32_t)(r - d)); need_row = std::max(need_row, (uint32_t)(r - d)); }
        }
        P[r].need = std::max((int32_t)need_row, (int32_t)r - SLACK);
    }
    done_ = std::vector<std::atomic<int32_t>>((size_t)NW * 4);
    for (int w = 0; w < NW; ++w) for (int q = 0; q < 4; ++q) done_[w * 4 + q].store(q < T ? q + 1 - T : BIG);
    mcnt = std::vector<std::atomic<uint32_t>>((size_t)NW * 4);
    mailT = std::vector<std::atomic<uint32_t>>((size_t)NW * 4 * MD); mailH = std::vector<std::atomic<uint32_t>>((size_t)NW * 4 * MD);
    ring = std::vector<std::atomic<uint32_t>>((size_t)NW * (SLOTS + 1));
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) for (int w = 0; w < NW; ++w) th.emplace_back(wave, t, w, seed);
    for (auto &x : th) x.join();
    if (bad) return 1;
    puts("MT_PROTOCOL_OK");
    return 0;
}
