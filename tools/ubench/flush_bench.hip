// tools/ubench/flush_bench.hip -- pricing the batch flush of the blend backward on the matrix pipe.
//
// The blend backward (monogs_amd/csrc/blend.hip, blend_backward_t_kernel) leaves two per-pixel factors per survivor in an LDS
// slab (h = G dL/dalpha, w = alpha T) and every BT_SLOTS survivors "turns round" and forms the ten per-Gaussian sums
//     S[g, k] = sum_p F[g, p] * B[p, k]          F = {h, w} rows, B = {1, u, v, u^2, uv, v^2 | dL/d{r, g, b, depth}}
// with plain VALU + a DPP butterfly (bt_flush: ~75 VALU per four survivors).  That contraction is GEMM-shaped, and the
// matrix pipe is idle in this kernel.  This program times two complete flushes on identical inputs, 8 waves per SIMD, every
// CU busy, with `work` dependent FMAs per survivor standing in for the per-pixel side of the walk:
//     variant 0   mgs::bt_flush<false> itself (this file includes blend.hip), batches of 4
//     variant 1   bt_flush_mfma below: 16 x v_mfma_f32_16x16x4_f32 per batch of 6 survivors (basis functions on the M side,
//                 factor rows on the N side), raw moments shifted to the Gaussian's centre in-lane, the six gradient lines
//                 transposed through LDS into ONE 60-lane atomic instruction
// and checks that both leave the same sums in the gradient lines.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I monogs_amd/csrc tools/ubench/flush_bench.hip -o tools/ubench/flush_bench
//   tools/ubench/flush_bench [survivors per wave, multiple of 12; default 1200]
#include "../../monogs_amd/csrc/blend.hip"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace mgs {
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
}  // namespace mgs
using namespace mgs;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_S = 6;                    // survivors per MFMA batch
constexpr int MF_ROW = 68;                 // slab row stride in floats (64 pixels + 4: the 16 lanes of a DPP row hit 64 distinct banks)
constexpr int MF_ROWS = 2 * MF_S;          // h rows 0..5, w rows 6..11
constexpr int MF_TQ = 384 + 4 * 256;       // bytes of one q-slice of the basis table: shared monomials, then the four waves' dL/dpixel

struct MfMeta { float x, y; uint32_t g, pad; };

// One survivor's inputs, a pure function of (wave id, survivor number, pixel): what the per-pixel side of the walk would leave.
__device__ __forceinline__ void make_factors(uint32_t wid, uint32_t n, int lane, int work, float& h, float& w) {
    float a = __uint_as_float(0x3f000000u | ((wid * 2654435761u + n * 40503u + (uint32_t)lane * 9973u) & 0x7fffffu));   // [0.5, 1)
    float b = a * 0.37f + 0.11f;
    for (int i = 0; i < work; i += 4) {    // `work` dependent plain VALU instructions (the per-pixel side: ~30), four per trip
        a = __builtin_fmaf(a, 0.999f, b);
        b = __builtin_fmaf(b, 0.501f, -0.25f * a) + 0.2f;
        asm volatile("" : "+v"(a), "+v"(b));
    }
    h = a - 0.75f;
    w = b;
}
__device__ __forceinline__ void make_meta(uint32_t wid, uint32_t n, float qx0, float qy0, float& cx, float& cy, uint32_t& gid, uint32_t lines) {
    const uint32_t r = wid * 7919u + n * 104729u;
    cx = qx0 + (float)(r % 41u) - 16.25f;
    cy = qy0 + (float)((r / 41u) % 37u) - 14.5f;
    gid = (wid * 64u + (n % 64u)) % lines;
}
__device__ __forceinline__ float pix_const(uint32_t wid, int ch, int p) {      // dL/dpixel of channel ch at pixel p of the quadrant
    return 0.001f * (float)((int)((wid * 31u + (uint32_t)ch * 17u + (uint32_t)p * 7u) % 61u) - 30);
}

// ---------------------------------------------------------------------------------------------------------------------
// variant 0: the library's flush
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_valu(float* __restrict__ grad_acc, uint32_t lines, int n_surv, int work) {
    __shared__ __attribute__((aligned(16))) float s_fac[4][2][BT_SLOTS][WAVE];
    __shared__ BtMetaRec s_meta[4][BT_SLOTS];
    __shared__ __attribute__((aligned(16))) float s_pix[4][4][16][4];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t wid = blockIdx.x * 4u + (uint32_t)wave;
    const float qx0 = (float)((wid % 240u) * 8u), qy0 = (float)((wid / 240u % 135u) * 8u);
    const int q = lane & 15;
    if (lane < 16)
        for (int ch = 0; ch < 4; ++ch)
            for (int i = 0; i < 4; ++i) s_pix[wave][ch][q][i] = pix_const(wid, ch, (q >> 1) * 8 + 4 * (q & 1) + i);
    const int slot = bt_slot10(q);
    const uint32_t slot_bytes = slot < 0 ? 0u : (uint32_t)slot * 4u;
    const unsigned long long m_out = __builtin_amdgcn_ballot_w64(slot >= 0);
    const unsigned long long m_q0 = __builtin_amdgcn_ballot_w64((q & 3) == 0), m_q1 = __builtin_amdgcn_ballot_w64((q & 3) == 1);
    float* const fhl = &s_fac[wave][0][0][lane];
    BtMetaRec* const fmeta = &s_meta[wave][0];
    const BtLane bl{&s_fac[wave][0][0][0], &s_pix[wave][0][0][0], fmeta, lane, qx0, qy0, slot_bytes, m_out, m_q0, m_q1, grad_acc};
    int k = 0;
    for (int n = 0; n < n_surv; ++n) {
        float h, w, cx, cy;
        uint32_t gid;
        make_factors(wid, (uint32_t)n, lane, work, h, w);
        make_meta(wid, (uint32_t)n, qx0, qy0, cx, cy, gid, lines);
        fhl[k * WAVE] = h;
        fhl[(BT_SLOTS + k) * WAVE] = w;
        if (lane == 0) fmeta[k] = BtMetaRec{cx, cy, gid};
        if (++k == BT_SLOTS) {
            bt_flush<false>(bl, BT_SLOTS);
            k = 0;
        }
    }
    if (k) bt_flush<false>(bl, k);
}

// ---------------------------------------------------------------------------------------------------------------------
// variant 1: the same sums on the matrix pipe
//
// D[i][j] = sum_p A[i][p] B[p][j],  sixteen v_mfma_f32_16x16x4_f32 (K = 64 pixels), lane l = 16 kq + c:
//   A (16 x 4 per instruction): lane holds A[i = c][k = kq]: BASIS rows        i:  0..3  = {1, u, v, u^2}
//                                                                                   4..7  = {1, u, v, u v}
//                                                                                   8..11 = {1, u, v, v^2}
//                                                                                   12..15 = dL/d{r, g, b, depth}
//   B (4 x 16): lane holds B[k = kq][j = c]: FACTOR columns   j: 0..5 = h of survivor j,  8..13 = w of survivor j - 8
//   D: four registers, lane holds D[i = 4 kq + r][j = c]
// so lane (kq = 0, c = s) ends with {M0, M1, M2, M3} of survivor s in its four registers, lanes (1, s) / (2, s) with
// {M0, M1, M2, M4} / {M0, M1, M2, M5} (the three low monomials are repeated so that the shift to the Gaussian's centre
//     sum h dx      = a M0 - M1                 (dx = cx - px = a - u,  a = cx - qx0;  dy likewise with b)
//     sum h dx^2    = a^2 M0 - 2 a M1 + M3
//     sum h dx dy   = a b M0 - b M1 - a M2 + M4
//     sum h dy^2    = b^2 M0 - 2 b M2 + M5
// needs no other lane), and lane (3, 8 + s) with the four colour / depth sums of survivor s.  Pixel of (instruction t, kq):
// p = 16 kq + t, so a lane's sixteen operands are four ds_read_b128 of consecutive floats on either side.
// The ten sums of a survivor then sit in four lanes; they cross to ten adjacent lanes of ONE register through a 48-byte line
// image per survivor in LDS (aliased onto the slab, whose contents are dead by then), and one atomic instruction with 60 active
// lanes covers the six 64-byte gradient lines: one memory-side request per line, as bt_flush.
// ---------------------------------------------------------------------------------------------------------------------
struct MfLane {
    const char* tab;        // this lane's basis operands: tab + q * MF_TQ, 16 bytes each
    const char* slab;       // this lane's factor operands: slab + q * 16
    float* wave_slab;       // the wave's slab (the line images alias its first 288 bytes)
    const MfMeta* meta;
    float qx0, qy0;
    int lane;
    float* grad_acc;
};

__device__ __forceinline__ void bt_flush_mfma(const MfLane L, int n) {
    int lane = L.lane;
    asm volatile("" : "+v"(lane));
    const int c = lane & 15, kq = lane >> 4;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(L.tab + q * MF_TQ);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(L.slab + q * 16);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, d, 0, 0, 0);
    }
    // ---- shift the raw moments to the Gaussian's centre (lanes kq <= 2, c < 6); lanes (3, 8 + s) pass through
    const int s = c & 7;
    const MfMeta me = L.meta[s < MF_S ? s : 0];
    const float a = me.x - L.qx0, b = me.y - L.qy0;
    const float t1 = __builtin_fmaf(a, d.x, -d.y), t2 = __builtin_fmaf(b, d.x, -d.z);
    const bool g0 = kq == 0, g2 = kq == 2;
    const float p = g0 ? a : b, qq = g2 ? b : a, T = g2 ? t2 : t1, Mx = g0 ? d.y : d.z;
    const float o3 = __builtin_fmaf(p, T, __builtin_fmaf(-qq, Mx, d.w));
    // ---- line images: [survivor][12 floats] = {DR, DG, DB, DDEPTH | SX, SY, SXX, SXY | SYY, SH, -, -}
    float* const img = L.wave_slab + s * 12;
    if (kq == 3) {
        if (c >= 8 && c < 8 + MF_S) *reinterpret_cast<f32x4*>(img) = d;
    } else if (c < MF_S) {
        if (kq == 0) {
            img[4] = t1; img[5] = t2; img[6] = o3; img[9] = d.x;
        } else {
            img[kq == 1 ? 7 : 8] = o3;
        }
    }
    // ---- one register, lane = 10 s' + e: element e of survivor s'
    const int sp = (lane * 205) >> 11, e = lane - 10 * sp;          // lane / 10 for lane < 64
    const bool live = sp < n;                                       // n <= 6: lanes 60..63 are never live
    const int slot = e < 4 ? G_DR + e : (e < 8 ? G_SX + (e - 4) : (e == 8 ? G_SYY : G_SH));
    const float v = L.wave_slab[(live ? sp : 0) * 12 + e];
    const uint32_t gid = L.meta[live ? sp : 0].g;
    const uint32_t off = gid * (uint32_t)(GRAD_FLOATS * sizeof(float)) + (uint32_t)slot * 4u;
    if (live) asm volatile("global_atomic_add_f32 %0, %1, %2" ::"v"(off), "v"(v), "s"(L.grad_acc) : "memory");
}

__global__ void __launch_bounds__(256) k_mfma(float* __restrict__ grad_acc, uint32_t lines, int n_surv, int work) {
    __shared__ __attribute__((aligned(16))) float s_slab[4][MF_ROWS][MF_ROW];
    __shared__ __attribute__((aligned(16))) char s_tab[4 * MF_TQ];
    __shared__ __attribute__((aligned(16))) MfMeta s_meta[4][MF_S];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t wid = blockIdx.x * 4u + (uint32_t)wave;
    const float qx0 = (float)((wid % 240u) * 8u), qy0 = (float)((wid / 240u % 135u) * 8u);
    // ---- basis table: [q][ shared: kq 4 x monomial 6 x 4 floats | wave 4 x (kq 4 x channel 4 x 4 floats) ], pixel p = 16 kq + 4 q + e
    for (int idx = threadIdx.x; idx < 4 * 4 * 6 * 4; idx += 256) {
        const int e = idx & 3, m = (idx >> 2) % 6, kq = (idx / 24) & 3, q = idx / 96;
        const int p = 16 * kq + 4 * q + e, u = p & 7, v = p >> 3;
        const float val = m == 0 ? 1.f : m == 1 ? (float)u : m == 2 ? (float)v : m == 3 ? (float)(u * u) : m == 4 ? (float)(u * v) : (float)(v * v);
        reinterpret_cast<float*>(s_tab + q * MF_TQ)[(kq * 6 + m) * 4 + e] = val;
    }
    for (int idx = lane; idx < 4 * 4 * 4 * 4; idx += 64) {
        const int e = idx & 3, ch = (idx >> 2) & 3, kq = (idx >> 4) & 3, q = idx >> 6;
        reinterpret_cast<float*>(s_tab + q * MF_TQ + 384 + wave * 256)[(kq * 4 + ch) * 4 + e] = pix_const(wid, ch, 16 * kq + 4 * q + e);
    }
    __syncthreads();
    const int c = lane & 15, kq = lane >> 4;
    const int mono = c < 4 ? c : (c < 7 ? c - 4 : (c == 7 ? 4 : (c < 11 ? c - 8 : 5)));
    const char* tab = s_tab + (c < 12 ? (kq * 6 + mono) * 16 : 384 + wave * 256 + (kq * 4 + (c - 12)) * 16);
    const int row = c < MF_S ? c : ((c >= 8 && c < 8 + MF_S) ? MF_S + (c - 8) : 0);
    const char* slab = reinterpret_cast<const char*>(&s_slab[wave][row][16 * kq]);
    const MfLane ml{tab, slab, &s_slab[wave][0][0], &s_meta[wave][0], qx0, qy0, lane, grad_acc};
    float* const col = &s_slab[wave][0][lane];          // this lane's (pixel's) column of the slab
    int k = 0;
    for (int n = 0; n < n_surv; ++n) {
        float h, w, cx, cy;
        uint32_t gid;
        make_factors(wid, (uint32_t)n, lane, work, h, w);
        make_meta(wid, (uint32_t)n, qx0, qy0, cx, cy, gid, lines);
        col[k * MF_ROW] = h;
        col[(MF_S + k) * MF_ROW] = w;
        if (lane == 0) s_meta[wave][k] = MfMeta{cx, cy, gid, 0u};
        if (++k == MF_S) {
            bt_flush_mfma(ml, MF_S);
            k = 0;
        }
    }
    if (k) bt_flush_mfma(ml, k);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n_surv = argc > 1 ? atoi(argv[1]) : 1200;
    if (n_surv <= 0 || n_surv % 12) { fprintf(stderr, "survivors per wave must be a positive multiple of 12\n"); return 1; }
    const int blocks = 256 * 8;                          // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    const uint32_t lines = (uint32_t)blocks * 4u * 64u;  // gradient lines: 64 per wave
    float *ga, *gb;
    CK(hipMalloc(&ga, (size_t)lines * 64));
    CK(hipMalloc(&gb, (size_t)lines * 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // ---- correctness: the two variants leave the same sums
    CK(hipMemset(ga, 0, (size_t)lines * 64));
    CK(hipMemset(gb, 0, (size_t)lines * 64));
    hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, ga, lines, 120, 4);
    hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, gb, lines, 120, 4);
    CK(hipDeviceSynchronize());
    std::vector<float> ha((size_t)lines * 16), hb((size_t)lines * 16);
    CK(hipMemcpy(ha.data(), ga, ha.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), gb, hb.size() * 4, hipMemcpyDeviceToHost));
    double num[10] = {0}, den[10] = {0};
    for (size_t i = 0; i < ha.size(); ++i) {
        const int sl = (int)(i % 16);
        if (sl < 10) { const double d = (double)ha[i] - hb[i]; num[sl] += d * d; den[sl] += (double)ha[i] * ha[i]; }
        else if (ha[i] != 0.f || hb[i] != 0.f) { fprintf(stderr, "slot %d written\n", sl); return 2; }
    }
    bool ok = true;
    printf("relative L2 difference per gradient slot (mfma vs valu flush):");
    for (int sl = 0; sl < 10; ++sl) {
        const double r = std::sqrt(num[sl] / (den[sl] > 0 ? den[sl] : 1.0));
        printf(" %.1e", r);
        ok = ok && den[sl] > 0 && r < 2e-5;
    }
    printf("  -> %s\n", ok ? "AGREE" : "DIFFER");
    // ---- timing
    printf("%-34s %10s %10s   ns of SIMD time per survivor (8 waves per SIMD, %d survivors per wave)\n", "per-survivor work", "valu", "mfma", n_surv);
    for (int work : {0, 16, 32, 48}) {
        float ms[2];
        for (int var = 0; var < 2; ++var) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (var == 0) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, ga, lines, n_surv, work);
                else hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, gb, lines, n_surv, work);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[var], e0, e1));
            }
        }
        const double per = 1e6 / (8.0 * n_surv);       // ms -> ns per survivor per SIMD
        printf("%2d dependent plain VALU + 2 ds_write  %8.2f ns %8.2f ns   (%.3f ms vs %.3f ms)%s\n", work, ms[0] * per, ms[1] * per, ms[0], ms[1],
               work == 32 ? "   <- the backward's pixel side is ~30 VALU" : "");
    }
    printf("%s\n", ok ? "OK" : "MISMATCH");
    return ok ? 0 : 3;
}
