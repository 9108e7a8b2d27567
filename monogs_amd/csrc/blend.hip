// Per-tile alpha blending, forward and backward.
//
// Boundary replaced: upstream `renderCUDA` forward / backward of the un-vendored rasteriser
// (SURVEY.md section 2.1 K6, K7); per-fragment rule corroborated by
// /root/reference/viewer/gl_render/shaders/gau_frag.glsl:20-25; output semantics pinned by the
// consumers /root/reference/utils/slam_utils.py:71,91,143 and utils/slam_tracker.py:414.
//
// gfx950 mapping.  A 16x16 tile is one 256-thread workgroup, but its four wavefronts never
// synchronise: wave w owns the 8x8 pixel quadrant (w&1, w>>1), one pixel per lane, and walks the
// tile's instance list on its own, 64 instances per step:
//   1. lane l loads instance l of the step: the Gaussian index (coalesced) and its 64-byte record;
//   2. lane l tests the record's alpha>=1/255 bounding box against the wave's quadrant;
//      __ballot compacts the survivors into a 64-bit mask held in scalar registers;
//   3. the wave pops survivors off the mask; v_readlane broadcasts the survivor's record into
//      SGPRs, and all 64 lanes (pixels) evaluate it with scalar operands.
// No LDS, no barriers; a quadrant whose pixels are all saturated retires early.
// Skipping an instance for a whole quadrant never changes a pixel: every skipped pair has
// alpha < 1/255 and the per-pixel rule would have skipped it too.
#include "common.h"

namespace mgs {

__device__ __forceinline__ float bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t bcast(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

struct BlendArgs {
    const float* __restrict__ rec;
    const uint32_t* __restrict__ point_list;
    const uint2* __restrict__ ranges;
    const float* __restrict__ bg;
    int W, H, gx;
};

// record of the instance held by a lane + quadrant box test
struct LaneRec {
    float4 r0, r1, r2;
    uint32_t gid;
};

__device__ __forceinline__ bool load_and_test(const BlendArgs& a, uint32_t i, uint32_t end, float qx0, float qy0,
                                              LaneRec& L) {
    const bool valid = i < end;
    L.gid = 0;
    bool hit = false;
    if (valid) {
        L.gid = a.point_list[i];
        const float4* r = reinterpret_cast<const float4*>(a.rec + (size_t)L.gid * REC_FLOATS);
        L.r0 = r[0];
        L.r1 = r[1];
        L.r2 = r[2];
        const float ex = L.r1.w, ey = L.r2.w;
        hit = (L.r0.x + ex >= qx0) && (L.r0.x - ex <= qx0 + (float)(SUB - 1)) && (L.r0.y + ey >= qy0) &&
              (L.r0.y - ey <= qy0 + (float)(SUB - 1));
    }
    return hit;
}

__global__ void __launch_bounds__(256) blend_forward_kernel(BlendArgs a, float* __restrict__ out_color,
                                                            float* __restrict__ out_depth,
                                                            float* __restrict__ out_opacity,
                                                            float* __restrict__ final_T,
                                                            uint32_t* __restrict__ n_contrib,
                                                            int32_t* __restrict__ n_touched) {
    const int tile = blockIdx.x;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qx0i = tx * TILE + (wave & 1) * SUB, qy0i = ty * TILE + (wave >> 1) * SUB;
    const int pxi = qx0i + (lane & 7), pyi = qy0i + (lane >> 3);
    const bool inside = pxi < a.W && pyi < a.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const float qx0 = (float)qx0i, qy0 = (float)qy0i;
    const uint2 range = a.ranges[tile];

    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    for (uint32_t base = range.x; base < range.y; base += WAVE) {
        if (__ballot(!done) == 0ull) break;
        LaneRec L;
        const bool hit = load_and_test(a, base + lane, range.y, qx0, qy0, L);
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float gxp = bcast(L.r0.x, j), gyp = bcast(L.r0.y, j), gz = bcast(L.r0.z, j), op = bcast(L.r0.w, j);
            const float ca = bcast(L.r1.x, j), cb = bcast(L.r1.y, j), cc = bcast(L.r1.z, j);
            const float cr = bcast(L.r2.x, j), cg = bcast(L.r2.y, j), cbl = bcast(L.r2.z, j);
            const float dx = gxp - pxf, dy = gyp - pyf;
            const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
            const float alpha = fminf(0.99f, op * __expf(power));
            const bool act = !done && !(power > 0.f) && !(alpha < 1.0f / 255.0f);
            const float test_T = T * (1.f - alpha);
            const bool stop = act && (test_T < 0.0001f);
            done = done || stop;
            const bool contrib = act && !stop;
            if (contrib) {
                const float w = alpha * T;
                C0 += cr * w;
                C1 += cg * w;
                C2 += cbl * w;
                D += gz * w;
                T = test_T;
                last = (base - range.x) + (uint32_t)j + 1u;
            }
            const unsigned long long touched = __ballot(contrib && test_T > 0.5f);
            if (touched) {
                const uint32_t gid = bcast(L.gid, j);
                if (lane == 0) atomicAdd(n_touched + gid, (int)__popcll(touched));
            }
            if (__ballot(!done) == 0ull) break;
        }
    }
    if (inside) {
        const size_t pix = (size_t)pyi * a.W + pxi, HW = (size_t)a.H * a.W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C0 + T * a.bg[0];
        out_color[HW + pix] = C1 + T * a.bg[1];
        out_color[2 * HW + pix] = C2 + T * a.bg[2];
        out_depth[pix] = D;
        out_opacity[pix] = 1.f - T;
    }
}

int launch_blend_forward(const mgs_camera& cam, const GeometryState& g, const BinningState& b,
                         const ImageState& img, float* out_color, float* out_depth, float* out_opacity,
                         int32_t* n_touched, hipStream_t s) {
    BlendArgs a;
    a.rec = g.rec; a.point_list = b.vals_sorted; a.ranges = img.ranges; a.bg = cam.bg;
    a.W = cam.image_width; a.H = cam.image_height; a.gx = tiles_x(a.W);
    const int ntiles = a.gx * tiles_y(a.H);
    if (ntiles == 0) return 0;
    hipLaunchKernelGGL(blend_forward_kernel, dim3(ntiles), dim3(256), 0, s, a, out_color, out_depth, out_opacity,
                       img.final_T, img.n_contrib, n_touched);
    MGS_HIP(hipGetLastError());
    return 0;
}

// =================================================================================================
// backward: the same walk, back to front, from the pixel's last contributor
// =================================================================================================

// sum over the 64 lanes, result valid in lane 63 (4 row-local DPP steps + 2 row broadcasts)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of each row holds the row sum
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2,3 -> lane 63 holds the wave sum
    return v;
}

// ---- packed reduction of 10 per-lane values over the 64 lanes --------------------------------
// stage 1: v_permlane32_swap pairs (x, y): lanes 0-31 then hold x[l]+x[l+32], lanes 32-63 y[l-32]+y[l]
// stage 2: v_permlane16_swap pairs those: each 16-lane row then holds 16 partials of ONE value
// stage 3: row-local DPP sum -> lane 15 of each row
__device__ __forceinline__ float swap32_add(float x, float y) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float x, float y) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float row_sum_to_lane15(float v) {
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    return v;
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// Where reduce10 leaves value k (k = gradient slot G_DX..G_DDEPTH):
//   lane 15: a0   lane 31: a2   lane 47: a1   lane 63: a3      (register c0)
//   lane  0: a4   lane 16: a6   lane 32: a5   lane 48: a7      (c1 rotated right by 1 inside rows)
//   lane  1: a8   lane 33: a9                                  (c2 rotated right by 2)
__device__ __forceinline__ int reduce10_slot(int lane) {
    const int row = lane >> 4, pos = lane & 15;
    if (pos == 15) return row == 0 ? 0 : row == 1 ? 2 : row == 2 ? 1 : 3;
    if (pos == 0) return row == 0 ? 4 : row == 1 ? 6 : row == 2 ? 5 : 7;
    if (pos == 1) return row == 0 ? 8 : row == 2 ? 9 : -1;
    return -1;
}
__device__ __forceinline__ float reduce10(float a0, float a1, float a2, float a3, float a4, float a5, float a6,
                                          float a7, float a8, float a9, int lane) {
    const float b0 = swap32_add(a0, a1), b1 = swap32_add(a2, a3), b2 = swap32_add(a4, a5);
    const float b3 = swap32_add(a6, a7), b4 = swap32_add(a8, a9);
    float c0 = swap16_add(b0, b1);     // rows: a0 a2 a1 a3
    float c1 = swap16_add(b2, b3);     // rows: a4 a6 a5 a7
    float c2 = swap16_add(b4, b4);     // rows: a8 a8 a9 a9
    c0 = row_sum_to_lane15(c0);
    c1 = row_sum_to_lane15(c1);
    c2 = row_sum_to_lane15(c2);
    const float c1r = dpp_mov<0x121>(c1);   // row_ror:1 -> lane 0 of each row
    const float c2r = dpp_mov<0x122>(c2);   // row_ror:2 -> lane 1 of each row
    const int pos = lane & 15;
    return pos == 15 ? c0 : (pos == 0 ? c1r : c2r);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}

__global__ void __launch_bounds__(256) blend_backward_kernel(BlendArgs a, const float* __restrict__ final_T,
                                                             const uint32_t* __restrict__ n_contrib,
                                                             const float* __restrict__ dL_dcolor,
                                                             const float* __restrict__ dL_ddepth,
                                                             float* __restrict__ grad_acc) {
    const int tile = blockIdx.x;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qx0i = tx * TILE + (wave & 1) * SUB, qy0i = ty * TILE + (wave >> 1) * SUB;
    const int pxi = qx0i + (lane & 7), pyi = qy0i + (lane >> 3);
    const bool inside = pxi < a.W && pyi < a.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const float qx0 = (float)qx0i, qy0 = (float)qy0i;
    const uint2 range = a.ranges[tile];
    if (range.y <= range.x) return;

    const size_t pix = (size_t)pyi * a.W + pxi, HW = (size_t)a.H * a.W;
    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    const float g0 = inside ? dL_dcolor[pix] : 0.f;
    const float g1 = inside ? dL_dcolor[HW + pix] : 0.f;
    const float g2 = inside ? dL_dcolor[2 * HW + pix] : 0.f;
    const float gd = inside ? dL_ddepth[pix] : 0.f;
    const float bg_dot = a.bg[0] * g0 + a.bg[1] * g1 + a.bg[2] * g2;

    const uint32_t maxc = wave_max_u32(last);   // wave-uniform
    if (maxc == 0) return;
    const int slot = reduce10_slot(lane);

    float T = T_final;
    float A = 0.f;            // sum over channels of (colour behind) * dL/dpixel, blended back to front
    float last_alpha = 0.f, last_q = 0.f;
    const uint32_t end = range.x + maxc;

    for (int b = (int)((maxc - 1) / WAVE); b >= 0; --b) {
        const uint32_t base = range.x + (uint32_t)b * WAVE;
        LaneRec L;
        const bool hit = load_and_test(a, base + lane, end, qx0, qy0, L);
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int j = 63 - __builtin_clzll(mask);
            mask &= ~(1ull << j);
            const uint32_t k = (uint32_t)b * WAVE + (uint32_t)j + 1u;     // 1-based position in the tile list
            const float gxp = bcast(L.r0.x, j), gyp = bcast(L.r0.y, j), gz = bcast(L.r0.z, j), op = bcast(L.r0.w, j);
            const float ca = bcast(L.r1.x, j), cb = bcast(L.r1.y, j), cc = bcast(L.r1.z, j);
            const float cr = bcast(L.r2.x, j), cg = bcast(L.r2.y, j), cbl = bcast(L.r2.z, j);
            const float dx = gxp - pxf, dy = gyp - pyf;
            const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(0.99f, op * G);
            const bool act = (k <= last) && !(power > 0.f) && !(alpha < 1.0f / 255.0f);
            if (__ballot(act) == 0ull) continue;

            float v_dx = 0.f, v_dy = 0.f, v_ca = 0.f, v_cb = 0.f, v_cc = 0.f, v_op = 0.f;
            float v_r = 0.f, v_g = 0.f, v_b = 0.f, v_z = 0.f;
            if (act) {
                const float inv = 1.f / (1.f - alpha);
                T = T * inv;
                const float w = alpha * T;
                const float q = (cr * g0 + cg * g1) + (cbl * g2 + gz * gd);
                A = last_alpha * last_q + (1.f - last_alpha) * A;
                last_q = q;
                last_alpha = alpha;
                float dL_dalpha = (q - A) * T;
                dL_dalpha += (-T_final * inv) * bg_dot;
                const float dL_dG = op * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * ca - gdy * cb;
                const float dG_ddely = -gdy * cc - gdx * cb;
                v_dx = dL_dG * dG_ddelx;
                v_dy = dL_dG * dG_ddely;
                v_ca = -0.5f * gdx * dx * dL_dG;
                v_cb = -gdx * dy * dL_dG;
                v_cc = -0.5f * gdy * dy * dL_dG;
                v_op = G * dL_dalpha;
                v_r = w * g0;
                v_g = w * g1;
                v_b = w * g2;
                v_z = w * gd;
            }
            // ---- 10 wave sums in 28 VALU ops: two swap stages pack the values, then one 16-lane
            //      row reduction per packed register (see reduce10 above); one atomic instruction
            //      with 10 active lanes adds the whole record (a single 64-byte segment).
            const float m = reduce10(v_dx, v_dy, v_ca, v_cb, v_cc, v_op, v_r, v_g, v_b, v_z, lane);
            const uint32_t gid = bcast(L.gid, j);
            if (slot >= 0) atomicAdd(grad_acc + (size_t)gid * GRAD_FLOATS + slot, m);
        }
    }
}

int launch_blend_backward(const mgs_camera& cam, const GeometryState& g, const BinningState& b,
                          const ImageState& img, const float* dL_dcolor, const float* dL_ddepth, float* grad_acc,
                          hipStream_t s) {
    BlendArgs a;
    a.rec = g.rec; a.point_list = b.vals_sorted; a.ranges = img.ranges; a.bg = cam.bg;
    a.W = cam.image_width; a.H = cam.image_height; a.gx = tiles_x(a.W);
    const int ntiles = a.gx * tiles_y(a.H);
    if (ntiles == 0) return 0;
    hipLaunchKernelGGL(blend_backward_kernel, dim3(ntiles), dim3(256), 0, s, a, img.final_T, img.n_contrib, dL_dcolor,
                       dL_ddepth, grad_acc);
    MGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace mgs
