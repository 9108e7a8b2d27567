// Fused photometric / geometric losses of the two SLAM hot loops, forward value + analytic gradients.
//
// Caller-side widening of the rasteriser path: these replace the ~60 small
// elementwise / boolean-index / reduction kernels (and the host syncs of `mask.any()` and boolean
// indexing) that PyTorch launches per iteration for
//   get_loss_mapping   /root/reference/utils/slam_utils.py:101-146
//   get_loss_tracking  /root/reference/utils/slam_utils.py:58-98
// and they emit dL/dcolor[3,H,W], dL/ddepth[1,H,W] directly in the layout the blend backward reads.
//
//   mapping :  L = lambda * mean_{mask, 3 ch} |rgb - gt| + (1 - lambda) * mean_{gt_depth > 0} |depth - gt_depth|
//   tracking:  L = 0.5 * mean(opacity) * mean_{all 3HW} (m |rgb - gt|) + mean_{gt_depth > 0, opacity > 0.99} |depth - gt_depth|
//              with m = mask * grad_mask * (opacity > 0.99); the depth term is 0 when its mask is empty
//   rgb = exp(a) * render + b   (exposure; identity when `init`)
//   invert_depth (slam_utils.py:83-88, :138-141): the depth term compares 1 / (depth + eps) with 1 / (gt_depth + eps),
//   eps = 1e-6 in the tracking loss and 0 in the mapping loss, as the reference writes them
//
// Forward: one reduction kernel (<= 256 workgroups write partial sums) + a one-wave finalize kernel that adds
// them in a fixed order (no atomics, no memset, bitwise reproducible); backward: one elementwise kernel.  HBM-bound: ~44 B/pixel read forward, ~60 B/pixel backward.
#include "common.h"

namespace mgs {

constexpr int LS_THREADS = 256;
constexpr int LS_MAX_BLOCKS = 256;      // forward reduction: two stages, no atomics (64 workgroups were latency-bound: 26 us at VGA)
// scratch: 16 floats of results followed by LS_MAX_BLOCKS x 8 floats of per-workgroup partial sums
constexpr int LP_SUMS = 7;          // per-workgroup partial sums: s_rgb, c_rgb, s_d, c_d, s_op, u_a, u_b
enum : int { LP_L1_RGB = 6, LP_L1_D = 7, LP_SCALE_RGB = 8, LP_SCALE_D = 9, LP_DAB = 10, LP_LOSS = 12, LP_UA = 13, LP_UB = 14, LP_N = 16, LP_PART = 8 };
// scratch[LP_DAB .. LP_DAB+1]: d(exposure_a), d(exposure_b) of the fused value + gradients call.  The exposure gradients are
// SUMS over the pixels of terms that differ from the forward's only by a scalar factor (the loss scale): the forward kernel
// accumulates the unscaled sums U_a = sum exp(a) (s . x), U_b = sum (s0 + s1 + s2) in its per-workgroup partials, and ONE thread
// of the backward stores scale * U -- no atomics (300 workgroups adding into the same two floats cost 6.6 of the backward's
// 13.4 us at VGA), no clear, bitwise reproducible.

struct LossArgs {
    const float *render, *depth, *opacity, *gt_rgb, *gt_depth, *exp_a, *exp_b;
    const uint8_t *mask, *grad_mask;
    int W, H, tracking, init, invert_depth;
    float lambda_rgb;
};

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
// exp(a) * render + b - gt as PyTorch evaluates `torch.exp(a) * image + b` and the subtraction that follows
// (/root/reference/utils/slam_utils.py:74,122): a multiply, an add, a subtract, each rounded on its own.  The SIGN of this
// residual is the whole gradient of the L1 term, so a fused multiply-add here flips the pixels whose residual rounds to
// zero one way and not the other (one element of 921 600 on the C2 frame: 1e-4 of relative L2 in dL/dmeans3D).
__device__ __forceinline__ float residual(float ea, float x, float eb, float gt) { return __fsub_rn(__fadd_rn(__fmul_rn(ea, x), eb), gt); }

// per-pixel terms of the forward sums (shared by the scalar and the 4-pixel paths)
struct FwdAcc { float s_rgb, c_rgb, s_d, c_d, s_op, u_a, u_b; };
__device__ __forceinline__ void fwd_pixel(const LossArgs& a, float ea, float eb, float gd, bool mask, bool gmask, float op,
                                          float r0, float r1, float r2, float g0, float g1, float g2, float d, FwdAcc& acc) {
    bool m_rgb = mask, m_d = gd > 0.f;
    if (a.tracking) {
        const bool opaque = op > 0.99f;
        acc.s_op += op;
        m_rgb = m_rgb && gmask && opaque;
        m_d = m_d && opaque;
    }
    if (m_rgb) {
        const float d0 = residual(ea, r0, eb, g0), d1 = residual(ea, r1, eb, g1), d2 = residual(ea, r2, eb, g2);
        acc.s_rgb += fabsf(d0) + fabsf(d1) + fabsf(d2);
        acc.c_rgb += 3.f;
        const float s0 = sgn(d0), s1 = sgn(d1), s2 = sgn(d2);
        acc.u_a += ea * (s0 * r0 + s1 * r1 + s2 * r2);      // d/da of exp(a) x + b, unscaled
        acc.u_b += s0 + s1 + s2;
    }
    if (m_d) {
        const float eps = a.tracking ? 1e-6f : 0.f;
        acc.s_d += a.invert_depth ? fabsf(1.f / (d + eps) - 1.f / (gd + eps)) : fabsf(d - gd);
        acc.c_d += 1.f;
    }
}

// VEC4: every image is read four pixels at a time (16-byte loads; needs H*W % 4 == 0 and 16-byte-aligned images): a
// thread then has all its loads in flight at once instead of ~11 dependent-latency scalar loads per pixel.
template <bool VEC4>
__global__ void __launch_bounds__(LS_THREADS) loss_forward_kernel(LossArgs a, float* __restrict__ part) {
    const size_t HW = (size_t)a.W * a.H;
    const float ea = a.init ? 1.f : expf(a.exp_a[0]), eb = a.init ? 0.f : a.exp_b[0];      // (torch.exp: the accurate one)
    FwdAcc acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * LS_THREADS;
    if (VEC4) {
        const size_t NQ = HW / 4;
        const float4 *R0 = (const float4*)a.render, *R1 = (const float4*)(a.render + HW), *R2 = (const float4*)(a.render + 2 * HW);
        const float4 *G0 = (const float4*)a.gt_rgb, *G1 = (const float4*)(a.gt_rgb + HW), *G2 = (const float4*)(a.gt_rgb + 2 * HW);
        for (size_t q = (size_t)blockIdx.x * LS_THREADS + threadIdx.x; q < NQ; q += stride) {
            const float4 gd = ((const float4*)a.gt_depth)[q], d = ((const float4*)a.depth)[q];
            const float4 r0 = R0[q], r1 = R1[q], r2 = R2[q], g0 = G0[q], g1 = G1[q], g2 = G2[q];
            const uchar4 mk = a.mask ? ((const uchar4*)a.mask)[q] : make_uchar4(1, 1, 1, 1);
            const uchar4 gm = a.tracking ? ((const uchar4*)a.grad_mask)[q] : make_uchar4(1, 1, 1, 1);
            const float4 op = a.tracking ? ((const float4*)a.opacity)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            fwd_pixel(a, ea, eb, gd.x, mk.x != 0, gm.x != 0, op.x, r0.x, r1.x, r2.x, g0.x, g1.x, g2.x, d.x, acc);
            fwd_pixel(a, ea, eb, gd.y, mk.y != 0, gm.y != 0, op.y, r0.y, r1.y, r2.y, g0.y, g1.y, g2.y, d.y, acc);
            fwd_pixel(a, ea, eb, gd.z, mk.z != 0, gm.z != 0, op.z, r0.z, r1.z, r2.z, g0.z, g1.z, g2.z, d.z, acc);
            fwd_pixel(a, ea, eb, gd.w, mk.w != 0, gm.w != 0, op.w, r0.w, r1.w, r2.w, g0.w, g1.w, g2.w, d.w, acc);
        }
    } else {
        for (size_t p = (size_t)blockIdx.x * LS_THREADS + threadIdx.x; p < HW; p += stride)
            fwd_pixel(a, ea, eb, a.gt_depth[p], a.mask ? a.mask[p] != 0 : true, a.tracking ? a.grad_mask[p] != 0 : true,
                      a.tracking ? a.opacity[p] : 0.f, a.render[p], a.render[HW + p], a.render[2 * HW + p], a.gt_rgb[p],
                      a.gt_rgb[HW + p], a.gt_rgb[2 * HW + p], a.depth[p], acc);
    }
    // the seven sums of the workgroup with ONE pair of barriers (wave sums by shuffles, the four waves' sums through LDS, added
    // in the order ((w0 + w1) + (w2 + w3)) by seven threads)
    float v7[LP_SUMS] = {acc.s_rgb, acc.c_rgb, acc.s_d, acc.c_d, acc.s_op, acc.u_a, acc.u_b};
#pragma unroll
    for (int k = 0; k < LP_SUMS; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v7[k] += __shfl_xor(v7[k], o, 64);
    __shared__ float s_w[4][LP_SUMS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < LP_SUMS; ++k) s_w[wv][k] = v7[k];
    }
    __syncthreads();
    if (threadIdx.x < LP_SUMS) {
        const int k = threadIdx.x;
        part[LP_N + (size_t)blockIdx.x * LP_PART + k] = (s_w[0][k] + s_w[1][k]) + (s_w[2][k] + s_w[3][k]);
    }
}

struct LossScalars { float l1_rgb, l1_d, scale_rgb, scale_d, loss; };

// the scalar part of the loss from the five sums (see the formulas at the top of the file)
__device__ __forceinline__ LossScalars loss_scalars(const LossArgs& a, const float v[5]) {
    const size_t HW = (size_t)a.W * a.H;
    const float S_rgb = v[0], C_rgb = v[1], S_d = v[2], C_d = v[3], S_op = v[4];
    LossScalars r;
    if (a.tracking) {
        const float mean_op = S_op / (float)HW;
        r.l1_rgb = mean_op * (S_rgb / (3.f * (float)HW));
        r.l1_d = C_d > 0.f ? S_d / C_d : 0.f;
        r.scale_rgb = 0.5f * mean_op / (3.f * (float)HW);
        r.scale_d = C_d > 0.f ? 1.f / C_d : 0.f;
        r.loss = 0.5f * r.l1_rgb + r.l1_d;
    } else {
        r.l1_rgb = S_rgb / C_rgb;                       // NaN when the mask is empty, like torch's mean of nothing
        r.l1_d = S_d / C_d;
        r.scale_rgb = a.lambda_rgb / C_rgb;
        r.scale_d = (1.f - a.lambda_rgb) / C_d;
        r.loss = a.lambda_rgb * r.l1_rgb + (1.f - a.lambda_rgb) * r.l1_d;
    }
    return r;
}

// one wave: sum the per-workgroup partials in a fixed order (lane l adds blocks l, l+64, ...: bitwise reproducible)
__device__ __forceinline__ void sum_partials(const float* __restrict__ part, int nblocks, int lane, float v[LP_SUMS]) {
#pragma unroll
    for (int k = 0; k < LP_SUMS; ++k) v[k] = 0.f;
    for (int b = lane; b < nblocks; b += WAVE) {
        const float* o = part + LP_N + (size_t)b * LP_PART;
#pragma unroll
        for (int k = 0; k < LP_SUMS; ++k) v[k] += o[k];
    }
#pragma unroll
    for (int k = 0; k < LP_SUMS; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
}

__global__ void loss_finalize_kernel(LossArgs a, int nblocks, float* __restrict__ part, float* __restrict__ loss_out) {
    const int lane = threadIdx.x;
    float v[LP_SUMS];
    sum_partials(part, nblocks, lane, v);
    if (lane == 0) {
        const LossScalars r = loss_scalars(a, v);
        part[LP_L1_RGB] = r.l1_rgb;
        part[LP_L1_D] = r.l1_d;
        part[LP_SCALE_RGB] = r.scale_rgb;
        part[LP_SCALE_D] = r.scale_d;
        part[LP_UA] = v[5];
        part[LP_UB] = v[6];
        loss_out[0] = r.loss;
    }
}

// fwd_blocks > 0: "fused" mode -- no finalize kernel ran: every workgroup adds the forward's partial sums up itself (its
// first wave, fixed order, <= 8 KB out of L2) and workgroup 0 also stores the loss value for whoever wants to log it.
template <bool VEC4>
__global__ void __launch_bounds__(LS_THREADS) loss_backward_kernel(LossArgs a, float* __restrict__ part, int fwd_blocks,
                                                                   const float* __restrict__ grad_out,
                                                                   float* __restrict__ d_render,
                                                                   float* __restrict__ d_depth,
                                                                   float* __restrict__ d_ab) {
    __shared__ float s_scale[2];
    const size_t HW = (size_t)a.W * a.H;
    const float go = grad_out ? grad_out[0] : 1.f;
    const float ea = a.init ? 1.f : expf(a.exp_a[0]), eb = a.init ? 0.f : a.exp_b[0];      // (torch.exp: the accurate one)
    if (fwd_blocks > 0) {
        if (threadIdx.x < WAVE) {
            float v[LP_SUMS];
            sum_partials(part, fwd_blocks, (int)threadIdx.x, v);
            if (threadIdx.x == 0) {
                const LossScalars r = loss_scalars(a, v);
                s_scale[0] = r.scale_rgb; s_scale[1] = r.scale_d;
                if (blockIdx.x == 0) {
                    part[LP_L1_RGB] = r.l1_rgb; part[LP_L1_D] = r.l1_d; part[LP_SCALE_RGB] = r.scale_rgb;
                    part[LP_SCALE_D] = r.scale_d; part[LP_LOSS] = r.loss;
                    if (d_ab && !a.init) { d_ab[0] = go * r.scale_rgb * v[5]; d_ab[1] = go * r.scale_rgb * v[6]; }
                }
            }
        }
        __syncthreads();
    } else if (threadIdx.x == 0) {
        s_scale[0] = part[LP_SCALE_RGB]; s_scale[1] = part[LP_SCALE_D];
        if (blockIdx.x == 0 && d_ab && !a.init) {
            d_ab[0] = go * s_scale[0] * part[LP_UA];
            d_ab[1] = go * s_scale[0] * part[LP_UB];
        }
    }
    if (fwd_blocks <= 0) __syncthreads();
    const float k_rgb = go * s_scale[0], k_d = go * s_scale[1];
    const size_t stride = (size_t)gridDim.x * LS_THREADS;
    // one pixel: the three colour gradients and the depth gradient (the exposure gradients came out of the forward's sums)
    auto pixel = [&](float gd, bool mask, bool gmask, float op, float x0, float x1, float x2, float t0, float t1, float t2,
                     float d, float& o0, float& o1, float& o2, float& od) {
        bool m_rgb = mask, m_d = gd > 0.f;
        if (a.tracking) {
            const bool opaque = op > 0.99f;
            m_rgb = m_rgb && gmask && opaque;
            m_d = m_d && opaque;
        }
        o0 = o1 = o2 = 0.f;
        if (m_rgb) {
            const float s0 = sgn(residual(ea, x0, eb, t0)), s1 = sgn(residual(ea, x1, eb, t1)), s2 = sgn(residual(ea, x2, eb, t2));
            o0 = k_rgb * ea * s0; o1 = k_rgb * ea * s1; o2 = k_rgb * ea * s2;
        }
        if (a.invert_depth) {                       // d/dd |1/(d + eps) - 1/(gd + eps)| = -sgn(.) / (d + eps)^2
            const float eps = a.tracking ? 1e-6f : 0.f, inv = 1.f / (d + eps);
            od = m_d ? -k_d * sgn(inv - 1.f / (gd + eps)) * (inv * inv) : 0.f;
        } else {
            od = m_d ? k_d * sgn(d - gd) : 0.f;
        }
    };
    if (VEC4) {
        const size_t NQ = HW / 4;
        const float4 *R0 = (const float4*)a.render, *R1 = (const float4*)(a.render + HW), *R2 = (const float4*)(a.render + 2 * HW);
        const float4 *G0 = (const float4*)a.gt_rgb, *G1 = (const float4*)(a.gt_rgb + HW), *G2 = (const float4*)(a.gt_rgb + 2 * HW);
        float4 *O0 = (float4*)d_render, *O1 = (float4*)(d_render + HW), *O2 = (float4*)(d_render + 2 * HW);
        for (size_t q = (size_t)blockIdx.x * LS_THREADS + threadIdx.x; q < NQ; q += stride) {
            const float4 gd = ((const float4*)a.gt_depth)[q], d = ((const float4*)a.depth)[q];
            const float4 r0 = R0[q], r1 = R1[q], r2 = R2[q], t0 = G0[q], t1 = G1[q], t2 = G2[q];
            const uchar4 mk = a.mask ? ((const uchar4*)a.mask)[q] : make_uchar4(1, 1, 1, 1);
            const uchar4 gm = a.tracking ? ((const uchar4*)a.grad_mask)[q] : make_uchar4(1, 1, 1, 1);
            const float4 op = a.tracking ? ((const float4*)a.opacity)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o0, o1, o2, od;
            pixel(gd.x, mk.x != 0, gm.x != 0, op.x, r0.x, r1.x, r2.x, t0.x, t1.x, t2.x, d.x, o0.x, o1.x, o2.x, od.x);
            pixel(gd.y, mk.y != 0, gm.y != 0, op.y, r0.y, r1.y, r2.y, t0.y, t1.y, t2.y, d.y, o0.y, o1.y, o2.y, od.y);
            pixel(gd.z, mk.z != 0, gm.z != 0, op.z, r0.z, r1.z, r2.z, t0.z, t1.z, t2.z, d.z, o0.z, o1.z, o2.z, od.z);
            pixel(gd.w, mk.w != 0, gm.w != 0, op.w, r0.w, r1.w, r2.w, t0.w, t1.w, t2.w, d.w, o0.w, o1.w, o2.w, od.w);
            O0[q] = o0; O1[q] = o1; O2[q] = o2;
            ((float4*)d_depth)[q] = od;
        }
    } else {
        for (size_t p = (size_t)blockIdx.x * LS_THREADS + threadIdx.x; p < HW; p += stride) {
            float o0, o1, o2, od;
            pixel(a.gt_depth[p], a.mask ? a.mask[p] != 0 : true, a.tracking ? a.grad_mask[p] != 0 : true,
                  a.tracking ? a.opacity[p] : 0.f, a.render[p], a.render[HW + p], a.render[2 * HW + p], a.gt_rgb[p],
                  a.gt_rgb[HW + p], a.gt_rgb[2 * HW + p], a.depth[p], o0, o1, o2, od);
            d_render[p] = o0; d_render[HW + p] = o1; d_render[2 * HW + p] = o2;
            d_depth[p] = od;
        }
    }
}

// four pixels per thread on the vector path
static bool loss_vec4(const LossArgs& a, const float* d_render, const float* d_depth) {
    const size_t HW = (size_t)a.W * a.H;
    auto al = [](const void* p, size_t n) { return p == nullptr || ((size_t)p % n) == 0; };
    return HW % 4 == 0 && al(a.render, 16) && al(a.depth, 16) && al(a.opacity, 16) && al(a.gt_rgb, 16) && al(a.gt_depth, 16) &&
           al(a.mask, 4) && al(a.grad_mask, 4) && al(d_render, 16) && al(d_depth, 16);
}
static int loss_grid(int W, int H) {          // ~4 pixels per thread: one quad on the vector path
    const size_t HW = (size_t)W * H;
    size_t nb = (HW + LS_THREADS * 4 - 1) / (LS_THREADS * 4);
    return (int)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
}
static int loss_fwd_grid(int W, int H) {
    const int g = loss_grid(W, H);
    return g > LS_MAX_BLOCKS ? LS_MAX_BLOCKS : g;
}

int launch_loss_forward(const LossArgs& a, float* partials, float* loss_out, hipStream_t s) {
    const int nb = loss_fwd_grid(a.W, a.H);
    if (loss_vec4(a, nullptr, nullptr)) hipLaunchKernelGGL(loss_forward_kernel<true>, dim3(nb), dim3(LS_THREADS), 0, s, a, partials);
    else hipLaunchKernelGGL(loss_forward_kernel<false>, dim3(nb), dim3(LS_THREADS), 0, s, a, partials);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(WAVE), 0, s, a, nb, partials, loss_out);
    MGS_HIP(hipGetLastError());
    return 0;
}

int launch_loss_backward(const LossArgs& a, float* partials, const float* grad_out, float* d_render,
                         float* d_depth, float* d_ab, hipStream_t s) {
    // (d_ab is STORED by one thread of the kernel: no clear)
    if (loss_vec4(a, d_render, d_depth))
        hipLaunchKernelGGL(loss_backward_kernel<true>, dim3(loss_grid(a.W, a.H)), dim3(LS_THREADS), 0, s, a, partials, 0, grad_out,
                           d_render, d_depth, d_ab);
    else
        hipLaunchKernelGGL(loss_backward_kernel<false>, dim3(loss_grid(a.W, a.H)), dim3(LS_THREADS), 0, s, a, partials, 0, grad_out,
                           d_render, d_depth, d_ab);
    MGS_HIP(hipGetLastError());
    return 0;
}

// value + gradients in TWO launches (no finalize kernel, no autograd node for the scalar, hence no ones-fill either)
int launch_loss_grads(const LossArgs& a, float* partials, float* d_render, float* d_depth, hipStream_t s) {
    const int nb = loss_fwd_grid(a.W, a.H);
    float* d_ab = a.init ? nullptr : partials + LP_DAB;
    if (loss_vec4(a, d_render, d_depth)) {
        hipLaunchKernelGGL(loss_forward_kernel<true>, dim3(nb), dim3(LS_THREADS), 0, s, a, partials);
        hipLaunchKernelGGL(loss_backward_kernel<true>, dim3(loss_grid(a.W, a.H)), dim3(LS_THREADS), 0, s, a, partials, nb, nullptr,
                           d_render, d_depth, d_ab);
    } else {
        hipLaunchKernelGGL(loss_forward_kernel<false>, dim3(nb), dim3(LS_THREADS), 0, s, a, partials);
        hipLaunchKernelGGL(loss_backward_kernel<false>, dim3(loss_grid(a.W, a.H)), dim3(LS_THREADS), 0, s, a, partials, nb, nullptr,
                           d_render, d_depth, d_ab);
    }
    MGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace mgs

using namespace mgs;

extern "C" {

// `mode`: MGS_LOSS_TRACKING | MGS_LOSS_INVERT_DEPTH (0 = get_loss_mapping)
static int fill_args(LossArgs& a, int32_t W, int32_t H, int32_t mode, int32_t init, float lambda_rgb,
                     const float* render, const float* depth, const float* opacity, const float* gt_rgb,
                     const float* gt_depth, const uint8_t* mask, const uint8_t* grad_mask, const float* exp_a,
                     const float* exp_b) {
    if (W <= 0 || H <= 0) { set_error("image size must be positive"); return 1; }
    if (mode & ~(MGS_LOSS_TRACKING | MGS_LOSS_INVERT_DEPTH)) { set_error("unknown loss mode bits"); return 1; }
    const int tracking = (mode & MGS_LOSS_TRACKING) ? 1 : 0;
    if (!render || !depth || !gt_rgb || !gt_depth) { set_error("render, depth, gt_rgb, gt_depth must be non-NULL"); return 1; }
    if (tracking && (!opacity || !grad_mask)) { set_error("tracking loss needs opacity and grad_mask"); return 1; }
    if (!init && (!exp_a || !exp_b)) { set_error("exposure_a / exposure_b must be non-NULL unless init"); return 1; }
    a.render = render; a.depth = depth; a.opacity = opacity; a.gt_rgb = gt_rgb; a.gt_depth = gt_depth;
    a.exp_a = exp_a; a.exp_b = exp_b; a.mask = mask; a.grad_mask = grad_mask;
    a.W = W; a.H = H; a.tracking = tracking; a.init = init; a.lambda_rgb = lambda_rgb;
    a.invert_depth = (mode & MGS_LOSS_INVERT_DEPTH) ? 1 : 0;
    return 0;
}

size_t mgs_loss_scratch_bytes(void) { return (LP_N + LS_MAX_BLOCKS * LP_PART) * sizeof(float); }

int mgs_loss_forward(int32_t W, int32_t H, int32_t mode, int32_t init, float lambda_rgb, const float* render,
                     const float* depth, const float* opacity, const float* gt_rgb, const float* gt_depth,
                     const uint8_t* mask, const uint8_t* grad_mask, const float* exposure_a, const float* exposure_b,
                     float* scratch, float* loss_out, void* stream) {
    LossArgs a;
    if (fill_args(a, W, H, mode, init, lambda_rgb, render, depth, opacity, gt_rgb, gt_depth, mask, grad_mask,
                  exposure_a, exposure_b)) return 1;
    if (!scratch || !loss_out) { set_error("scratch and loss_out must be non-NULL"); return 1; }
    return launch_loss_forward(a, scratch, loss_out, (hipStream_t)stream);
}

int mgs_loss_backward(int32_t W, int32_t H, int32_t mode, int32_t init, float lambda_rgb, const float* render,
                      const float* depth, const float* opacity, const float* gt_rgb, const float* gt_depth,
                      const uint8_t* mask, const uint8_t* grad_mask, const float* exposure_a, const float* exposure_b,
                      const float* scratch, const float* grad_out, float* d_render, float* d_depth, float* d_exposure,
                      void* stream) {
    LossArgs a;
    if (fill_args(a, W, H, mode, init, lambda_rgb, render, depth, opacity, gt_rgb, gt_depth, mask, grad_mask,
                  exposure_a, exposure_b)) return 1;
    if (!scratch || !d_render || !d_depth) { set_error("scratch, d_render, d_depth must be non-NULL"); return 1; }
    return launch_loss_backward(a, const_cast<float*>(scratch), grad_out, d_render, d_depth, d_exposure, (hipStream_t)stream);
}

int mgs_loss_grads(int32_t W, int32_t H, int32_t mode, int32_t init, float lambda_rgb, const float* render,
                   const float* depth, const float* opacity, const float* gt_rgb, const float* gt_depth,
                   const uint8_t* mask, const uint8_t* grad_mask, const float* exposure_a, const float* exposure_b,
                   float* scratch, float* d_render, float* d_depth, void* stream) {
    LossArgs a;
    if (fill_args(a, W, H, mode, init, lambda_rgb, render, depth, opacity, gt_rgb, gt_depth, mask, grad_mask,
                  exposure_a, exposure_b)) return 1;
    if (!scratch || !d_render || !d_depth) { set_error("scratch, d_render, d_depth must be non-NULL"); return 1; }
    return launch_loss_grads(a, scratch, d_render, d_depth, (hipStream_t)stream);
}

}  // extern "C"
