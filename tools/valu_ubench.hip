// Micro-benchmark: issue rate of the VALU ops the DP kernel is built from, on gfx950.
// Usage: hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o /tmp/valu_ubench && /tmp/valu_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define REP 64
#define ITER 8000
// clk[2 * wave] = shader cycles (s_memtime), clk[2 * wave + 1] = ticks of the constant reference clock (s_memrealtime) the wave's loop took
template <int OP> __global__ void k(unsigned* out, unsigned seed, unsigned long long* clk)
{
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + i + 1);
    unsigned b = seed ^ 0x00030003u, c = 0x00040004u;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 3) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 5) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
                if (OP == 6) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (OP == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 9) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 10) asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b));
                if (OP == 11) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 12) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(unsigned long long*)&a[i & 6]) : "v"(*(unsigned long long*)&a[(i & 6) ^ 2]));
                if (OP == 13) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 100) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
                if (OP == 101) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a[i]));
                if (OP == 102) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 103) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 104) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 105) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 106) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 107) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 108) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 109) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 110) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 111) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 112) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 113) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 114) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[i]) : "v"(b));
                if (OP == 115) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a[i]));
                if (OP == 116) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 117) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 118) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 119) asm volatile("v_pk_lshlrev_b16 %0, 2, %0" : "+v"(a[i]));
                if (OP == 120) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 121) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 122) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (OP == 123) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(0x5555555555555555ull));
                if (OP == 124) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (OP == 125) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 126) asm volatile("v_sad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 127) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            }
        }
    }
    unsigned s = 0; for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) { const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; clk[2 * w] = t1 - t0; clk[2 * w + 1] = r1 - r0; }
}
// One workgroup of 1024 threads per CU, forced by a dynamic LDS request only one workgroup fits beside: every SIMD then holds exactly four waves for
// the whole kernel (a plain grid of small workgroups is NOT spread evenly over the CUs: the wall time then measures the most crowded CU, and the
// first version of this bench overstated every cost by ~1.5x that way).
template <int OP> void run(const char* name, unsigned* d)
{
    const int wps = 4, blocks = 256, threads = 1024; const size_t lds = 96 * 1024;
    static bool once = false;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    static unsigned long long* clk = nullptr; if (!clk) hipMalloc(&clk, 256 * 16 * 2 * sizeof(unsigned long long));
    int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    k<OP><<<blocks, threads, lds>>>(d, 12345u, clk); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, threads, lds>>>(d, 12345u, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int n_waves = blocks * threads / 64;
    std::vector<unsigned long long> h(2 * n_waves); hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, ref = 0, cmax = 0; for (int w = 0; w < n_waves; ++w) { cyc += (double)h[2 * w]; ref += (double)h[2 * w + 1]; if ((double)h[2 * w] > cmax) cmax = (double)h[2 * w]; }
    const double ghz = cyc / ref * wall_khz * 1e-6;              // shader cycles per second while the loops ran, from the two in-kernel clocks
    const double insts_per_simd = (double)wps * ITER * REP;      // wave-instructions issued on one SIMD
    if (!once) { once = true; printf("# 256 workgroups x 1024 threads, one per CU (LDS-forced): 4 waves per SIMD; ITER %d x REP %d\n", ITER, REP); }
    printf("%-28s %.3f ms wall -> %.2f cycles per wave-instr per SIMD at the measured %.3f GHz (%.2f at nominal 2.4); by s_memtime: mean wave span %.2f, slowest %.2f cycles per instr\n",
           name, ms, ms * 1e6 / insts_per_simd * ghz, ghz, ms * 1e6 / insts_per_simd * 2.4, cyc / n_waves / insts_per_simd, cmax / insts_per_simd);
}
int main()
{
    unsigned* d; hipMalloc(&d, 256 * 1024 * 4);
    run<0>("v_pk_add_u16", d); run<1>("v_pk_min_i16", d); run<13>("v_pk_min_u16", d); run<10>("v_pk_add_i16 clamp", d); run<5>("v_pk_mad_u16", d);
    run<4>("v_perm_b32", d); run<6>("v_mov_b32_dpp row_shr:1", d); run<2>("v_add_u32", d); run<3>("v_min_i32", d); run<9>("v_min3_i32", d);
    run<8>("v_and_b32", d); run<11>("v_lshl_or_b32", d); run<14>("v_cndmask_b32", d); run<7>("v_fma_f32", d); run<12>("v_pk_fma_f32", d);
    run<100>("v_mov_b32", d); run<101>("v_lshlrev_b32", d); run<102>("v_or_b32", d); run<103>("v_xor_b32", d); run<104>("v_sub_u32", d); run<105>("v_min_u32", d); run<106>("v_max_i32", d); run<107>("v_min_f32", d); run<108>("v_add_f32", d); run<109>("v_bfi_b32", d); run<110>("v_and_or_b32", d); run<111>("v_or3_b32", d); run<112>("v_add3_u32", d); run<113>("v_lshl_add_u32", d); run<114>("v_alignbit_b32", d); run<115>("v_bfe_u32", d); run<116>("v_min_u16", d); run<117>("v_add_u16", d); run<118>("v_pk_sub_u16", d); run<119>("v_pk_lshlrev_b16", d); run<120>("v_pk_mul_lo_u16", d); run<121>("v_mad_u32_u24", d); run<122>("v_cmp+v_cndmask", d); run<123>("v_cndmask_b32(sgpr)", d); run<124>("v_mov_b32_dpp wave_shr", d); run<125>("v_med3_i32", d); run<126>("v_sad_u16", d); run<127>("v_pk_max_i16", d);
    return 0;
}
