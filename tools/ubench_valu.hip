// Issue rate of the VALU instructions kernel C's row loop is made of, on this GPU: independent streams per wave,
// 4 waves per SIMD, so the number is the SIMD's issue throughput (wave64 instructions per cycle), not a latency.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip ; run: ./ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define OPS8(ins, tail) ins " %0, %0" tail "\n" ins " %1, %1" tail "\n" ins " %2, %2" tail "\n" ins " %3, %3" tail "\n" ins " %4, %4" tail "\n" ins " %5, %5" tail "\n" ins " %6, %6" tail "\n" ins " %7, %7" tail
#define REGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    unsigned b = blockIdx.x + 1, c = 0x00030003u;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile(OPS8("v_pk_max_i16", ", %8") : REGS : "v"(b));) }
        if (OP == 1) { REP8(asm volatile(OPS8("v_pk_add_u16", ", %8") : REGS : "v"(b));) }
        if (OP == 2) { REP8(asm volatile(OPS8("v_max_i32", ", %8") : REGS : "v"(b));) }
        if (OP == 3) { REP8(asm volatile(OPS8("v_alignbit_b32", ", %8, 16") : REGS : "v"(b));) }
        if (OP == 4) { REP8(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : REGS : "v"(b));) }
        if (OP == 5) { REP8(asm volatile(OPS8("v_and_b32", ", %8") : REGS : "v"(b));) }
        if (OP == 6) { REP8(asm volatile(OPS8("v_perm_b32", ", %8, %9") : REGS : "v"(b), "v"(c));) }
        if (OP == 7) { REP8(asm volatile(OPS8("v_max3_i32", ", %8, %9") : REGS : "v"(b), "v"(c));) }
        if (OP == 8) { REP8(asm volatile(OPS8("v_pk_min_i16", ", 3 op_sel_hi:[1,0]") : REGS : "v"(b));) }
        if (OP == 9) { REP8(asm volatile(OPS8("v_lshl_or_b32", ", 4, %8") : REGS : "v"(b));) }
        if (OP == 10) { REP8(asm volatile(OPS8("v_cndmask_b32", ", %8, vcc") : REGS : "v"(b) : "vcc");) }
        if (OP == 11) { REP8(asm volatile(OPS8("v_pk_sub_i16", ", %8") : REGS : "v"(b));) }
        if (OP == 12) { REP8(asm volatile(OPS8("v_bcnt_u32_b32", ", %8") : REGS : "v"(b));) }
        if (OP == 13) { REP8(asm volatile(OPS8("v_and_b32", ", %8") : REGS : "s"(i));) }
        if (OP == 14) { REP8(asm volatile(OPS8("v_add3_u32", ", %8, %9") : REGS : "v"(b), "v"(c));) }
        if (OP == 15) { REP8(asm volatile(OPS8("v_xor_b32", ", %8") : REGS : "v"(b));) }
        if (OP == 16) { REP8(asm volatile(OPS8("v_add_u32", ", %8") : REGS : "v"(b));) }
        if (OP == 17) { REP8(asm volatile("v_and_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %0, %1\n v_and_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %2, %3\n v_and_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %4, %5\n v_and_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %6, %7" : REGS : "s"(i));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP>
double run(const char *name, unsigned *d, int blocks) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 64;          // wave-instructions
    const double per_simd_cycle = winstr / (ms * 1e-3) / (256.0 * 4) / 2.4e9;
    printf("%-18s %8.3f ms  %7.3f T wave-instr/s  = %.3f per SIMD-cycle at 2.4 GHz (%.2f cycles each)\n", name, ms, winstr / (ms * 1e-3) / 1e12, per_simd_cycle, 1.0 / per_simd_cycle);
    return ms;
}

int main(int argc, char **argv) {
    unsigned *d;
    const int per_cu = argc > 1 ? atoi(argv[1]) : 4;      // blocks of 4 waves per CU = waves per SIMD
    const int blocks = 256 * per_cu;
    hipMalloc(&d, blocks * 256 * 4);
    printf("%d waves per SIMD\n", per_cu);
    run<0>("v_pk_max_i16", d, blocks); run<1>("v_pk_add_u16", d, blocks); run<11>("v_pk_sub_i16", d, blocks); run<8>("v_pk_min_i16 imm", d, blocks);
    run<2>("v_max_i32", d, blocks); run<7>("v_max3_i32", d, blocks); run<3>("v_alignbit_b32", d, blocks); run<6>("v_perm_b32", d, blocks);
    run<4>("v_mov_b32_dpp", d, blocks); run<5>("v_and_b32", d, blocks); run<9>("v_lshl_or_b32", d, blocks); run<10>("v_cndmask_b32", d, blocks);
    run<12>("v_bcnt_u32_b32", d, blocks); run<13>("v_and_b32 sgpr", d, blocks); run<14>("v_add3_u32", d, blocks); run<15>("v_xor_b32", d, blocks);
    run<16>("v_add_u32", d, blocks); run<17>("and(sgpr)+bcnt mix", d, blocks);
    return 0;
}
