// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_exp_f32 vs DPP add vs permlane swap on gfx950,
// 8 waves per SIMD, every CU busy.  Prints wave-instructions per SIMD-cycle (assuming 2.4 GHz is NOT assumed:
// reports ns per instruction per SIMD and the ratio between variants).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const float c = 1.0001f, d = 0.0001f;
    const float2_ c2 = {c, c}, d2 = {d, d};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            a0 = __builtin_fmaf(a0, c, d); a1 = __builtin_fmaf(a1, c, d); a2 = __builtin_fmaf(a2, c, d); a3 = __builtin_fmaf(a3, c, d);
            a4 = __builtin_fmaf(a4, c, d); a5 = __builtin_fmaf(a5, c, d); a6 = __builtin_fmaf(a6, c, d); a7 = __builtin_fmaf(a7, c, d);
        } else if (MODE == 1) {
            p0 = __builtin_elementwise_fma(p0, c2, d2); p1 = __builtin_elementwise_fma(p1, c2, d2);
            p2 = __builtin_elementwise_fma(p2, c2, d2); p3 = __builtin_elementwise_fma(p3, c2, d2);
            p4 = __builtin_elementwise_fma(p4, c2, d2); p5 = __builtin_elementwise_fma(p5, c2, d2);
            p6 = __builtin_elementwise_fma(p6, c2, d2); p7 = __builtin_elementwise_fma(p7, c2, d2);
        } else if (MODE == 2) {
            a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
            a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
        } else if (MODE == 3) {
            #define DPP(x) x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true))
            DPP(a0); DPP(a1); DPP(a2); DPP(a3); DPP(a4); DPP(a5); DPP(a6); DPP(a7);
        } else if (MODE == 4) {
            #define SW(x, y) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false); x = __uint_as_float(r[0]); y = __uint_as_float(r[1]); }
            SW(a0, a1); SW(a2, a3); SW(a4, a5); SW(a6, a7); SW(a0, a2); SW(a1, a3); SW(a4, a6); SW(a5, a7);
        } else if (MODE == 5) {
            a0 = a0 > c ? a1 : a0; a1 = a1 > c ? a2 : a1; a2 = a2 > c ? a3 : a2; a3 = a3 > c ? a4 : a3;
            a4 = a4 > c ? a5 : a4; a5 = a5 > c ? a6 : a5; a6 = a6 > c ? a7 : a6; a7 = a7 > c ? a0 : a7;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                          p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}

template <int MODE>
float run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 20000;
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_add_f32_dpp", "v_permlane32_swap", "v_cmp+v_cndmask"};
    float ms[6] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters), run<4>(out, iters), run<5>(out, iters)};
    for (int m = 0; m < 6; ++m) {
        // per SIMD: 8 waves x iters x 8 instr (x2 for mode 5)
        const double instr = 8.0 * iters * 8 * (m == 5 ? 2 : 1);
        printf("%-20s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", names[m], ms[m],
               ms[m] * 1e6 / instr, ms[m] * 1e6 / instr * 2.4);
    }
    return 0;
}
