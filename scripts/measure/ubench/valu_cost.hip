// Issue cost of the instruction classes hpk_stencil_s is made of, at its occupancy (1024-thread workgroups, one per
// CU, 4 waves per SIMD): cycles per instruction per SIMD = kernel cycles / (instructions per wave x 4 waves).
// build: hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_IT 4096
template <int KIND>
__global__ void __launch_bounds__(1024) bench(double* out, unsigned long long* clk, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
        if (KIND == 0) {            // 4 independent v_add_f64
            asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
        } else if (KIND == 1) {     // 4 independent v_add_u32
            asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0));
        } else if (KIND == 2) {     // 4 v_mov_b32 dpp row_shr:1
            asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
        } else if (KIND == 3) {     // 4 v_add_u32 dpp row_bcast:15
            asm volatile("v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                         "v_add_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                         "v_add_u32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                         "v_add_u32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
        } else if (KIND == 4) {     // 4 v_mov dpp wave_shr:1
            asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
        } else if (KIND == 5) {     // 4 v_mul_f64
            asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
        } else if (KIND == 6) {     // 4 v_cndmask_b32 + the compare that feeds them
            asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_cndmask_b32 %2, %2, %0, vcc"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
        } else if (KIND == 7) {     // 4 v_add_f32
            asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f0));
        } else if (KIND == 8) {     // 2 v_readlane + 2 SALU consumers
            int s0, s1;
            asm volatile("v_readlane_b32 %0, %2, 3\n v_readlane_b32 %1, %3, 5\n s_add_u32 %0, %0, %1\n s_nop 0"
                         : "=s"(s0), "=s"(s1) : "v"(u0), "v"(u1));
            u2 += (unsigned)s0;
        } else if (KIND == 9) {     // 4 SALU
            int s = i;
            asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 2\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 4" : "+s"(s));
            u2 += (unsigned)s;
        } else if (KIND == 10) {    // 4 v_cvt_f64_f32
            asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
        } else if (KIND == 12) {    // 4 v_rcp_f64
            asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (KIND == 13) {    // 4 v_div_scale_f64
            asm volatile("v_div_scale_f64 %0, vcc, %0, %4, %0\n v_div_scale_f64 %1, vcc, %1, %4, %1\n v_div_scale_f64 %2, vcc, %2, %4, %2\n v_div_scale_f64 %3, vcc, %3, %4, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) : "vcc");
        } else if (KIND == 14) {    // 4 v_div_fmas_f64
            asm volatile("v_div_fmas_f64 %0, %0, %4, %0\n v_div_fmas_f64 %1, %1, %4, %1\n v_div_fmas_f64 %2, %2, %4, %2\n v_div_fmas_f64 %3, %3, %4, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) : "vcc");
        } else if (KIND == 15) {    // 4 v_div_fixup_f64
            asm volatile("v_div_fixup_f64 %0, %0, %4, %0\n v_div_fixup_f64 %1, %1, %4, %1\n v_div_fixup_f64 %2, %2, %4, %2\n v_div_fixup_f64 %3, %3, %4, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
        } else if (KIND == 16) {    // 4 v_fma_f64
            asm volatile("v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
        } else if (KIND == 17) {    // 4 v_cmp_ge_f64 (to an SGPR pair)
            unsigned long long m0, m1;
            asm volatile("v_cmp_ge_f64 %0, %2, %3\n v_cmp_ge_f64 %1, %3, %4\n v_cmp_ge_f64 %0, %4, %5\n v_cmp_ge_f64 %1, %5, %2"
                         : "=s"(m0), "=s"(m1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            u2 += (unsigned)m0 + (unsigned)m1;
        } else if (KIND == 18) {    // 4 v_mul_lo_u32
            asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0));
        } else if (KIND == 19) {    // 4 v_lshl_add_u64
            asm volatile("v_lshl_add_u64 %0, %0, 3, %4\n v_lshl_add_u64 %1, %1, 3, %4\n v_lshl_add_u64 %2, %2, 3, %4\n v_lshl_add_u64 %3, %3, 3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
        } else if (KIND == 20) {    // 4 v_mul_u32_u24
            asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0));
        } else if (KIND == 21) {    // 4 v_cmp_ge_u64
            unsigned long long m0, m1;
            asm volatile("v_cmp_ge_u64 %0, %2, %3\n v_cmp_ge_u64 %1, %3, %4\n v_cmp_ge_u64 %0, %4, %5\n v_cmp_ge_u64 %1, %5, %2"
                         : "=s"(m0), "=s"(m1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            u2 += (unsigned)m0 + (unsigned)m1;
        } else if (KIND == 11) {    // 4 v_pk_add_f32 (two f32 per lane each)
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + u0 + u1 + u2 + u3 + f0 + f1 + f2 + f3;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, double* out, unsigned long long* clk) {
    hipLaunchKernelGGL(bench<KIND>, dim3(256), dim3(1024), 0, 0, out, clk, 1.5);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<KIND>, dim3(256), dim3(1024), 0, 0, out, clk, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0; for (auto v : h) { mean += v; if (v > mx) mx = v; } mean /= h.size();
    // 4 instructions per iteration per wave, 4 waves per SIMD
    printf("%-28s kernel %.3f ms  ticks/wave mean %.0f max %.0f  -> %.2f ticks per instr per SIMD (max wave / (4 instr x %d it x 4 waves))\n",
           name, ms, mean, mx, mx / (4.0 * N_IT * 4.0), N_IT);
}

int main() {
    double* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&clk, 256 * 16 * 8);
    run<0>("v_add_f64", out, clk);
    run<5>("v_mul_f64", out, clk);
    run<10>("v_cvt_f64_f32", out, clk);
    run<7>("v_add_f32", out, clk);
    run<11>("v_pk_add_f32", out, clk);
    run<1>("v_add_u32", out, clk);
    run<6>("v_cmp + 3 v_cndmask", out, clk);
    run<2>("v_mov_dpp row_shr:1", out, clk);
    run<3>("v_add_u32_dpp row_bcast:15", out, clk);
    run<4>("v_mov_dpp wave_shr:1", out, clk);
    run<16>("v_fma_f64", out, clk);
    run<12>("v_rcp_f64", out, clk);
    run<13>("v_div_scale_f64", out, clk);
    run<14>("v_div_fmas_f64", out, clk);
    run<15>("v_div_fixup_f64", out, clk);
    run<17>("v_cmp_ge_f64 -> sgpr", out, clk);
    run<21>("v_cmp_ge_u64 -> sgpr", out, clk);
    run<18>("v_mul_lo_u32", out, clk);
    run<20>("v_mul_u32_u24", out, clk);
    run<19>("v_lshl_add_u64", out, clk);
    run<8>("2 v_readlane + s_add + s_nop", out, clk);
    run<9>("4 s_add_u32", out, clk);
    return 0;
}
