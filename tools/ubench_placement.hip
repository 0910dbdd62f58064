// Where do the wavefronts of a workgroup go?  Launches WG workgroups of NW wavefronts (as many as the device holds at once, each spinning
// for a while so that they are co-resident) and tabulates, per CU, which SIMD the wavefront 0 of every resident workgroup sits on.
// build: hipcc --offload-arch=gfx950 -O2 -o ubench_placement tools/ubench_placement.hip ; usage: ubench_placement [NW] [LDS_BYTES]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void probe(unsigned *out, int nw, unsigned long long ticks) {
    extern __shared__ unsigned lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * nw + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * nw + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    lds[threadIdx.x] = hw;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int main(int argc, char **argv) {
    const int nw = argc > 1 ? atoi(argv[1]) : 4, ldsb = argc > 2 ? atoi(argv[2]) : 20000, nblk = argc > 3 ? atoi(argv[3]) : 1792;
    unsigned *d; hipMalloc(&d, nblk * nw * 8);
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(64 * nw), ldsb, 0, d, nw, 2000000ull);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nblk * nw * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;      // cu key -> histogram of wave 0's SIMD
    long same_order = 0;
    for (int b = 0; b < nblk; ++b) {
        const unsigned hw0 = h[(b * nw) * 2], xcc = h[(b * nw) * 2 + 1];
        const unsigned key = (xcc << 8) | ((hw0 >> 8) & 0xFF);
        auto &v = per_cu[key]; v.resize(5);
        v[(hw0 >> 4) & 3]++; v[4]++;
        bool ord = true;
        for (int w = 0; w < nw; ++w) ord &= ((h[(b * nw + w) * 2] >> 4) & 3) == (unsigned)((((hw0 >> 4) & 3) + w) & 3);
        same_order += ord;
    }
    int shown = 0;
    long tot[4] = {0, 0, 0, 0};
    for (auto &kv : per_cu) { for (int k = 0; k < 4; ++k) tot[k] += kv.second[k]; if (shown++ < 6) printf("cu %04x: %d workgroups, wave 0 on SIMD 0/1/2/3: %d %d %d %d\n", kv.first, kv.second[4], kv.second[0], kv.second[1], kv.second[2], kv.second[3]); }
    printf("%zu CUs seen; wave 0 of all workgroups on SIMD 0/1/2/3: %ld %ld %ld %ld; workgroups whose wavefronts sit on consecutive SIMDs: %ld of %d\n", per_cu.size(), tot[0], tot[1], tot[2], tot[3], same_order, nblk);
    // which workgroups share a CU: print the block indices of the first CU
    for (auto &kv : per_cu) { printf("blocks on cu %04x:", kv.first); for (int b = 0; b < nblk; ++b) { const unsigned key = (h[(b * nw) * 2 + 1] << 8) | ((h[(b * nw) * 2] >> 8) & 0xFF); if (key == kv.first) printf(" %d", b); } printf("\n"); break; }
    return 0;
}
