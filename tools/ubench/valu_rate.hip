// Micro-benchmark: issue cost of the instructions the blend kernels are built from, on gfx950, 8 waves per SIMD, every CU
// busy.  Reports ns per wave-instruction per SIMD (no clock assumed) and "cycles" at the nominal 2.4 GHz; under this
// load the chip does not hold 2.4 GHz, so read the plain v_fma_f32 row as "2 cycles" (MI355X_MICROARCH.md: wave64
// v_fma_f32 issues in 2 cycles on the SIMD-32) and every other row relative to it.
//
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
//
// Rows "seq: ..." time whole reduction sequences of the blend backward (10 per-lane values -> 10 wave sums) the way the
// kernel runs them, so that a restructuring can be priced before it is built.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));

#define R8(M) M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)

__device__ __forceinline__ float swap32_add(float x, float y) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float x, float y) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int XOR>
__device__ __forceinline__ float swz_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x001F | (XOR << 10)));
}
__device__ __forceinline__ float row_sum_all(float v) {
    v = swz_add<1>(v); v = swz_add<2>(v); v = swz_add<4>(v); v = swz_add<8>(v);
    return v;
}
// the reduction of blend_backward_kernel<false> as of round 2 (7 swaps, 20 adds, 13 swizzles, 2 selects)
__device__ __forceinline__ float reduce10_r2(float a0, float a1, float a2, float a3, float a4, float a5, float a6,
                                             float a7, float a8, float a9, int lane) {
    const float b0 = swap32_add(a0, a1), b1 = swap32_add(a2, a3), b2 = swap32_add(a4, a5);
    const float b3 = swap32_add(a6, a7), b4 = swap32_add(a8, a9);
    float c0 = swap16_add(b0, b1), c1 = swap16_add(b2, b3), c2 = swz_add<16>(b4);
    c0 = row_sum_all(c0); c1 = row_sum_all(c1); c2 = row_sum_all(c2);
    const int pos = lane & 15;
    return pos == 15 ? c0 : (pos == 0 ? c1 : c2);
}
// x/y-separable variant: lane = 8 x + y.  Seven values {h, h dx, h dx^2, wr, wg, wb, wz} (+ h again) are summed over x with
// the packing swaps (x = lane bits 5..3), the three y-weighted copies are formed on the row sums, then ONE 3-stage
// butterfly over y on two registers: 6 swaps, 14 adds, 8 swizzles, 2 muls, 3 selects.
__device__ __forceinline__ float reduce_sep(float h, float hx, float hxx, float wr, float wg, float wb, float wz,
                                            float dy, unsigned long long m_dy, unsigned long long m_dy2,
                                            unsigned long long m_sel) {
    const float b0 = swap32_add(h, hx), b1 = swap32_add(hxx, wr), b2 = swap32_add(wg, wb), b3 = swap32_add(wz, h);
    float c0 = swap16_add(b0, b1);     // rows: h hxx hx wr
    float c1 = swap16_add(b2, b3);     // rows: wg wz wb h
    c0 = swz_add<8>(c0);
    c1 = swz_add<8>(c1);
    const float c0y = c0 * dy, c1y = c1 * (dy * dy);
    c0 = __builtin_amdgcn_inverse_ballot_w64(m_dy) ? c0y : c0;
    c1 = __builtin_amdgcn_inverse_ballot_w64(m_dy2) ? c1y : c1;
    c0 = swz_add<1>(c0); c0 = swz_add<2>(c0); c0 = swz_add<4>(c0);
    c1 = swz_add<1>(c1); c1 = swz_add<2>(c1); c1 = swz_add<4>(c1);
    return __builtin_amdgcn_inverse_ballot_w64(m_sel) ? c1 : c0;
}

enum {
    M_FMA, M_PKFMA, M_EXP, M_RCP, M_DPP_SHR, M_DPP_QUAD, M_DPP_MOV, M_SWAP32, M_SWAP16, M_CMP_CND, M_CNDMASK, M_READLANE,
    M_SWZ_ADD, M_BPERM_ADD, M_SWZ_ONLY, M_CND_SALU, M_CND_VCC1, M_CMPX, M_WRITELANE, M_DPP_BANK, M_MOV, M_CND_E64_INV, M_CND_VCC_MIX, M_FMA_LO32, M_FMA_ROW0, M_SALU_ANDN2, M_SALU_FF1, M_SALU_CSEL, M_SALU_MIX12, M_SALU_MIX11, M_FMA16, M_SMEM_X8, M_SMEM_MIX, M_DSREAD_B128, M_DSREAD_MIX, M_SEQ_R2, M_SEQ_SEP, M_COUNT
};
static const char* names[M_COUNT] = {
    "v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_add_f32_dpp row_shr:1", "v_add_f32_dpp quad_perm",
    "v_mov_b32_dpp row_shr:1", "v_permlane32_swap", "v_permlane16_swap", "v_cmp+v_cndmask (pair)",
    "v_cndmask (SGPR mask)", "v_readlane+v_add (pair)", "ds_swizzle+v_add (pair)", "ds_bpermute+v_add (pair)",
    "ds_swizzle chain (LDS pipe)", "v_cndmask, mask rewritten by SALU", "v_cndmask x8 after ONE v_cmp", "v_cmpx + 7 plain under EXEC",
    "v_writelane (SGPR lane sel)", "v_add_f32_dpp row_ror:8 bank_mask", "v_mov_b32", "v_cndmask_e64 x8, SGPR mask invariant", "v_cndmask_e32 vcc + v_fma alternating (pair)",
    "v_fma_f32, EXEC = lanes 0..31 only", "v_fma_f32, EXEC = lanes 0..15 only",
    "s_andn2_b64 x8 (alone)", "s_ff1_i32_b64 x8 (alone)", "s_cselect_b64+s_cmp x4 (alone)", "8 SALU + 16 v_fma interleaved (per trip)",
    "8 SALU + 8 v_fma interleaved (per trip)", "16 v_fma (per trip)", "s_load_dwordx8 x2 + wait (per trip)", "s_load_dwordx8 x2 + 16 v_fma (per trip)",
    "ds_read_b128 x3 uniform addr + wait (per trip)", "ds_read_b128 x3 + 16 v_fma (per trip)", "seq: reduce10 round 2", "seq: reduce x/y-separable"};
// wave-instructions of the row's kind per loop trip per wave (what the ns figure is divided by)
static const double per_trip[M_COUNT] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 16, 8, 16, 16, 16, 8, 8, 9, 8, 8, 8, 8, 8, 16, 8, 8, 8, 8, 8, 1, 1, 1, 1, 1, 1, 1, 1, 1};

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed, unsigned long long mask, const float* tab) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 64 * 12];
    if (MODE == M_DSREAD_B128 || MODE == M_DSREAD_MIX) {
        for (int i = threadIdx.x; i < 4 * 64 * 12; i += 256) lds[i] = seed + i;
        __syncthreads();
    }
    unsigned long long s0 = mask, s1 = mask + 1, s2 = mask + 2, s3 = mask + 3, s4 = mask + 4, s5 = mask + 5, s6 = mask + 6, s7 = mask + 7;
    int t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0;
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const float c = 1.0001f, d = 0.0001f;
    const float2_ c2 = {c, c}, d2 = {d, d};
    const int lane = threadIdx.x & 63;
    const int bp = (lane ^ 32) * 4;
    const float src_never_written = seed * 3.f + lane;
    for (int i = 0; i < iters; ++i) {
        if (MODE == M_FMA) {
#define F(x) x = __builtin_fmaf(x, c, d);
            R8(F)
#undef F
        } else if (MODE == M_PKFMA) {
            p0 = __builtin_elementwise_fma(p0, c2, d2); p1 = __builtin_elementwise_fma(p1, c2, d2);
            p2 = __builtin_elementwise_fma(p2, c2, d2); p3 = __builtin_elementwise_fma(p3, c2, d2);
            p4 = __builtin_elementwise_fma(p4, c2, d2); p5 = __builtin_elementwise_fma(p5, c2, d2);
            p6 = __builtin_elementwise_fma(p6, c2, d2); p7 = __builtin_elementwise_fma(p7, c2, d2);
        } else if (MODE == M_EXP) {
#define F(x) x = __builtin_amdgcn_exp2f(x);
            R8(F)
#undef F
        } else if (MODE == M_RCP) {
#define F(x) x = __builtin_amdgcn_rcpf(x);
            R8(F)
#undef F
        } else if (MODE == M_DPP_SHR) {
#define F(x) x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
            R8(F)
#undef F
        } else if (MODE == M_DPP_QUAD) {
#define F(x) x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));   /* quad_perm:[1,0,3,2] */
            R8(F)
#undef F
        } else if (MODE == M_DPP_MOV) {
#define F(x) x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
            R8(F)
#undef F
        } else if (MODE == M_SWAP32) {
#define SW(x, y) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false); x = __uint_as_float(r[0]); y = __uint_as_float(r[1]); }
            SW(a0, a1) SW(a2, a3) SW(a4, a5) SW(a6, a7) SW(a0, a2) SW(a1, a3) SW(a4, a6) SW(a5, a7)
#undef SW
        } else if (MODE == M_SWAP16) {
#define SW(x, y) { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false); x = __uint_as_float(r[0]); y = __uint_as_float(r[1]); }
            SW(a0, a1) SW(a2, a3) SW(a4, a5) SW(a6, a7) SW(a0, a2) SW(a1, a3) SW(a4, a6) SW(a5, a7)
#undef SW
        } else if (MODE == M_CMP_CND) {
            a0 = a0 > c ? a1 : a0; a1 = a1 > c ? a2 : a1; a2 = a2 > c ? a3 : a2; a3 = a3 > c ? a4 : a3;
            a4 = a4 > c ? a5 : a4; a5 = a5 > c ? a6 : a5; a6 = a6 > c ? a7 : a6; a7 = a7 > c ? a0 : a7;
        } else if (MODE == M_CNDMASK) {
            const bool s = __builtin_amdgcn_inverse_ballot_w64(mask);
            a0 = s ? a1 : a0; a1 = s ? a2 : a1; a2 = s ? a3 : a2; a3 = s ? a4 : a3;
            a4 = s ? a5 : a4; a5 = s ? a6 : a5; a6 = s ? a7 : a6; a7 = s ? a0 : a7;
        } else if (MODE == M_READLANE) {
#define F(x) x = x + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 5));
            R8(F)
#undef F
        } else if (MODE == M_SWZ_ADD) {
#define F(x) x = swz_add<1>(x);
            R8(F)
#undef F
        } else if (MODE == M_BPERM_ADD) {
#define F(x) x = x + __int_as_float(__builtin_amdgcn_ds_bpermute(bp, __float_as_int(x)));
            R8(F)
#undef F
        } else if (MODE == M_SWZ_ONLY) {
#define F(x) x = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x001F | (1 << 10)));
            R8(F)
#undef F
        } else if (MODE == M_CND_SALU) {
            asm volatile("s_xor_b64 %0, %0, 0x55" : "+s"(mask));          // the mask is (re)written by the scalar unit every trip
            const bool s = __builtin_amdgcn_inverse_ballot_w64(mask);
            a0 = s ? a1 : a0; a1 = s ? a2 : a1; a2 = s ? a3 : a2; a3 = s ? a4 : a3;
            a4 = s ? a5 : a4; a5 = s ? a6 : a5; a6 = s ? a7 : a6; a7 = s ? a0 : a7;
        } else if (MODE == M_CND_VCC1) {
            const bool s = a0 > c;                                           // ONE v_cmp (VALU writes the mask), eight selects on it
            a0 = s ? a1 : a0; a1 = s ? a2 : a1; a2 = s ? a3 : a2; a3 = s ? a4 : a3;
            a4 = s ? a5 : a4; a5 = s ? a6 : a5; a6 = s ? a7 : a6; a7 = s ? a0 : a7;
        } else if (MODE == M_CMPX) {
            if (a0 > c) {                                                    // v_cmpx / s_and_saveexec, then plain work under the narrowed EXEC
                a1 = __builtin_fmaf(a1, c, d); a2 = __builtin_fmaf(a2, c, d); a3 = __builtin_fmaf(a3, c, d); a4 = __builtin_fmaf(a4, c, d);
                a5 = __builtin_fmaf(a5, c, d); a6 = __builtin_fmaf(a6, c, d); a7 = __builtin_fmaf(a7, c, d);
            }
            a0 += d;
        } else if (MODE == M_WRITELANE) {
            const int sl = (i * 16) & 63;
#define F(x) asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(x) : "s"(i), "s"(sl) : "m0");
            R8(F)
#undef F
        } else if (MODE == M_DPP_BANK) {
            // packing DPP add: only the lanes of banks 2, 3 (lane bit 3 set) are written, the others keep their value -- two of
            // these with complementary bank masks halve two values into one register.  (Sources never written in this loop.)
#define F(r) asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(r) : "v"(src_never_written));
            R8(F)
#undef F
        } else if (MODE == M_MOV) {
#define F(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(a0));
            F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7) F(p0.x)
#undef F
        } else if (MODE == M_CND_E64_INV) {
#define F(r) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r) : "v"(src_never_written), "s"(mask));
            R8(F)
#undef F
        } else if (MODE == M_CND_VCC_MIX) {
#define F(r) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_fma_f32 %0, %0, %0, %1" : "+v"(r) : "v"(src_never_written));
            asm volatile("s_mov_b64 vcc, %0" :: "s"(mask) : "vcc");
            R8(F)
#undef F
        } else if (MODE == M_FMA_LO32 || MODE == M_FMA_ROW0) {
            // does the SIMD skip the passes of a wave64 instruction whose lanes are all switched off?  (one asm block: the
            // compiler never sees the narrowed EXEC)
            const unsigned long long em = MODE == M_FMA_LO32 ? 0xFFFFFFFFull : 0xFFFFull;
            asm volatile("s_mov_b64 exec, %8\n\t"
                         "v_fma_f32 %0, %0, %9, %10\n\tv_fma_f32 %1, %1, %9, %10\n\tv_fma_f32 %2, %2, %9, %10\n\tv_fma_f32 %3, %3, %9, %10\n\t"
                         "v_fma_f32 %4, %4, %9, %10\n\tv_fma_f32 %5, %5, %9, %10\n\tv_fma_f32 %6, %6, %9, %10\n\tv_fma_f32 %7, %7, %9, %10\n\t"
                         "s_mov_b64 exec, -1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(em), "v"(c), "v"(d));
        } else if (MODE == M_SALU_ANDN2) {
            // the scalar pipe alone: 8 independent SALU instructions per trip from each of the 8 waves of every SIMD
            asm volatile("s_andn2_b64 %0, %0, %8\n\ts_andn2_b64 %1, %1, %8\n\ts_andn2_b64 %2, %2, %8\n\ts_andn2_b64 %3, %3, %8\n\t"
                         "s_andn2_b64 %4, %4, %8\n\ts_andn2_b64 %5, %5, %8\n\ts_andn2_b64 %6, %6, %8\n\ts_andn2_b64 %7, %7, %8"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(mask) : "scc");
        } else if (MODE == M_SALU_FF1) {
            asm volatile("s_ff1_i32_b64 %0, %8\n\ts_ff1_i32_b64 %1, %9\n\ts_ff1_i32_b64 %2, %10\n\ts_ff1_i32_b64 %3, %11\n\t"
                         "s_ff1_i32_b64 %4, %12\n\ts_ff1_i32_b64 %5, %13\n\ts_ff1_i32_b64 %6, %14\n\ts_ff1_i32_b64 %7, %15"
                         : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7)
                         : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7));
            s0 += (unsigned)t0; s1 += (unsigned)t1; s2 += (unsigned)t2; s3 += (unsigned)t3; s4 += (unsigned)t4; s5 += (unsigned)t5; s6 += (unsigned)t6; s7 += (unsigned)t7;
        } else if (MODE == M_SALU_CSEL) {
            asm volatile("s_cmp_lg_u64 %0, 0\n\ts_cselect_b64 %1, %1, %4\n\ts_cmp_lg_u64 %1, 0\n\ts_cselect_b64 %2, %2, %4\n\t"
                         "s_cmp_lg_u64 %2, 0\n\ts_cselect_b64 %3, %3, %4\n\ts_cmp_lg_u64 %3, 0\n\ts_cselect_b64 %0, %0, %4"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(mask) : "scc");
        } else if (MODE == M_SALU_MIX12 || MODE == M_SALU_MIX11 || MODE == M_FMA16) {
            // one scalar instruction after every two (one) vector ones -- the blend kernels' ratio is 0.5 scalar : 1 vector
#define SV(sr, x, y) asm volatile("s_andn2_b64 %0, %0, %3\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5" : "+s"(sr), "+v"(x), "+v"(y) : "s"(mask), "v"(c), "v"(d) : "scc");
#define SV1(sr, x) asm volatile("s_andn2_b64 %0, %0, %2\n\tv_fma_f32 %1, %1, %3, %4" : "+s"(sr), "+v"(x) : "s"(mask), "v"(c), "v"(d) : "scc");
#define VV(x, y) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(c), "v"(d));
            if (MODE == M_SALU_MIX12) { SV(s0, a0, a1) SV(s1, a2, a3) SV(s2, a4, a5) SV(s3, a6, a7) SV(s4, a0, a1) SV(s5, a2, a3) SV(s6, a4, a5) SV(s7, a6, a7) }
            else if (MODE == M_SALU_MIX11) { SV1(s0, a0) SV1(s1, a1) SV1(s2, a2) SV1(s3, a3) SV1(s4, a4) SV1(s5, a5) SV1(s6, a6) SV1(s7, a7) }
            else { VV(a0, a1) VV(a2, a3) VV(a4, a5) VV(a6, a7) VV(a0, a1) VV(a2, a3) VV(a4, a5) VV(a6, a7) }
#undef SV
#undef SV1
#undef VV
        } else if (MODE == M_SMEM_X8 || MODE == M_SMEM_MIX) {
            // the backward's record fetch: two scalar loads of a wave-uniform address that changes every trip (L2-resident table)
            typedef float f8 __attribute__((ext_vector_type(8)));
            f8 r0, r1;
            const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            const unsigned off = ((((unsigned)i * 2654435761u) ^ (blockIdx.x * 97u + wv * 13u)) & 0xFFFFu) * 64u;
            const unsigned long long tb = (unsigned long long)tab;
            const unsigned long long ad = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(tb >> 32)) << 32 |
                                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tb)) + off;
            asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x20" : "=&s"(r0), "=&s"(r1) : "s"(ad));
            if (MODE == M_SMEM_MIX) {
#define VV(x, y) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(c), "v"(d));
                VV(a0, a1) VV(a2, a3) VV(a4, a5) VV(a6, a7) VV(a0, a1) VV(a2, a3) VV(a4, a5) VV(a6, a7)
#undef VV
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(a0) : "s"(r0[0]), "s"(r1[7]));
        } else if (MODE == M_DSREAD_B128 || MODE == M_DSREAD_MIX) {
            // the forward's record fetch: three broadcast LDS reads of a wave-uniform address
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 r0, r1, r2;
            const unsigned ad = (unsigned)(uintptr_t)lds + ((threadIdx.x >> 6) * 3072u) + (((unsigned)i * 7u) & 63u) * 48u;
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32" : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(ad));
            if (MODE == M_DSREAD_MIX) {
#define VV(x, y) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(c), "v"(d));
                VV(a1, a2) VV(a3, a4) VV(a5, a6) VV(a7, a1) VV(a2, a3) VV(a4, a5) VV(a6, a7) VV(a1, a2)
#undef VV
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3" : "+v"(a0) : "v"(r0[0]), "v"(r1[3]), "v"(r2[1]));
        } else if (MODE == M_SEQ_R2) {
            const float m = reduce10_r2(a0, a1, a2, a3, a4, a5, a6, a7, a0 + c, a1 + c, lane);
            a0 += m; a1 -= m; a2 += d; a3 += d; a4 += d; a5 += d; a6 += d; a7 += d;     // 8 plain ops of "other work"
        } else if (MODE == M_SEQ_SEP) {
            const float m = reduce_sep(a0, a1, a2, a3, a4, a5, a6, a7, mask, mask >> 7, mask >> 13);
            a0 += m; a1 -= m; a2 += d; a3 += d; a4 += d; a5 += d; a6 += d; a7 += d;
        }
    }
    if ((s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7) == 0x1234567ull && (t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7) == 77) a0 += 1.f;   // keep the scalar chains alive
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                          p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}

static float* g_tab = nullptr;
template <int MODE>
float run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, 0x00FF00FF00FF00FFull, g_tab);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0x00FF00FF00FF00FFull, g_tab);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int M>
void all(float* out, int iters, float* ms) {
    ms[M] = run<M>(out, iters);
    if constexpr (M + 1 < M_COUNT) all<M + 1>(out, iters, ms);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&g_tab, 65536 * 64 + 64); hipMemset(g_tab, 0, 65536 * 64 + 64);
    const int iters = 20000;
    float ms[M_COUNT];
    all<0>(out, iters, ms);
    const double fma_ns = ms[M_FMA] * 1e6 / (8.0 * iters * per_trip[M_FMA]);
    printf("%-32s %9s  %s\n", "instruction", "ms", "ns per wave-instruction per SIMD (cycles @2.4 GHz; x plain v_fma_f32)");
    for (int m = 0; m < M_COUNT; ++m) {
        const double ns = ms[m] * 1e6 / (8.0 * iters * per_trip[m]);   // 8 waves per SIMD
        printf("%-32s %9.3f  %7.2f ns  (%6.2f cycles; %5.2f x fma)%s\n", names[m], ms[m], ns, ns * 2.4, ns / fma_ns,
               m >= M_SEQ_R2 ? "  [whole sequence + 8 plain ops]" : (per_trip[m] == 1 ? "  [whole trip]" : ""));
    }
    printf("chip-wide plain-FMA issue ceiling measured here: %.1f G wave-inst/s (spec: 1024 SIMDs x 2.4 GHz / 2 = 1228.8)\n",
           1024.0 / fma_ns);
    // one machine-readable line: ns of SIMD issue per wave-instruction, by the classes tools/isa_mix.py counts
    auto ns_of = [&](int m) { return ms[m] * 1e6 / (8.0 * iters * per_trip[m]); };
    printf("JSON {\"valu_plain\": %.4f, \"valu_trans\": %.4f, \"valu_dpp\": %.4f, \"valu_cndmask\": %.4f, "
           "\"valu_permlane_swap\": %.4f, \"valu_lane\": %.4f, \"lds_swizzle_pipe\": %.4f, \"fma_ginst_s\": %.1f, "
           "\"salu_alone\": %.4f, \"salu_behind_2_valu\": %.4f, \"salu_behind_1_valu\": %.4f, \"smem_x2_wait_trip_alone\": %.4f, "
           "\"lds_bcast_b128_x3_behind_16_valu\": %.4f, "
           "\"unit\": \"ns of SIMD issue per wave-instruction, 8 waves per SIMD, every CU busy; salu_behind_N_valu: what a scalar "
           "instruction adds to a stream of N v_fma per scalar instruction (the CU's one scalar ALU serves four SIMDs)\"}\n",
           fma_ns, 0.5 * (ns_of(M_EXP) + ns_of(M_RCP)), 0.5 * (ns_of(M_DPP_SHR) + ns_of(M_DPP_BANK)), ns_of(M_CND_SALU),
           0.5 * (ns_of(M_SWAP32) + ns_of(M_SWAP16)), 2.0 * ns_of(M_READLANE) - fma_ns, ns_of(M_SWZ_ONLY), 1024.0 / fma_ns,
           ns_of(M_SALU_ANDN2), (ns_of(M_SALU_MIX12) - ns_of(M_FMA16)) / 8.0, (ns_of(M_SALU_MIX11) - 0.5 * ns_of(M_FMA16)) / 8.0,
           ns_of(M_SMEM_X8), (ns_of(M_DSREAD_MIX) - ns_of(M_FMA16)) / 3.0);
    return 0;
}
