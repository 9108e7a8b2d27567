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
//
// Forward: one reduction kernel (<= 256 workgroups write partial sums) + a one-wave finalize kernel that adds
// them in a fixed order (no atomics, no memset, bitwise reproducible); backward: one elementwise kernel.  HBM-bound: ~44 B/pixel read forward, ~60 B/pixel backward.
#include "common.h"

namespace mgs {

constexpr int LS_THREADS = 256;
constexpr int LS_MAX_BLOCKS = 256;      // forward reduction: two stages, no atomics (64 workgroups were latency-bound: 26 us at VGA)
// scratch: 16 floats of results followed by LS_MAX_BLOCKS x 8 floats of per-workgroup partial sums
enum : int { LP_L1_RGB = 6, LP_L1_D = 7, LP_SCALE_RGB = 8, LP_SCALE_D = 9, LP_DAB = 10, LP_N = 16, LP_PART = 8 };
// scratch[LP_DAB .. LP_DAB+1]: left ZERO by the forward; a backward may accumulate d(exposure_a), d(exposure_b) there
// (d_exposure == scratch + LP_DAB), which saves the launch that clears a separate buffer

struct LossArgs {
    const float *render, *depth, *opacity, *gt_rgb, *gt_depth, *exp_a, *exp_b;
    const uint8_t *mask, *grad_mask;
    int W, H, tracking, init;
    float lambda_rgb;
};

__device__ __forceinline__ float block_sum(float v, float* smem) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wv] = v;
    __syncthreads();
    return (smem[0] + smem[1]) + (smem[2] + smem[3]);
}

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

__global__ void __launch_bounds__(LS_THREADS) loss_forward_kernel(LossArgs a, float* __restrict__ part) {
    __shared__ float smem[4];
    const size_t HW = (size_t)a.W * a.H;
    const float ea = a.init ? 1.f : __expf(a.exp_a[0]), eb = a.init ? 0.f : a.exp_b[0];
    float s_rgb = 0.f, c_rgb = 0.f, s_d = 0.f, c_d = 0.f, s_op = 0.f;
    for (size_t p = (size_t)blockIdx.x * LS_THREADS + threadIdx.x; p < HW; p += (size_t)gridDim.x * LS_THREADS) {
        const float gd = a.gt_depth[p];
        bool m_rgb = a.mask ? a.mask[p] != 0 : true;
        bool m_d = gd > 0.f;
        if (a.tracking) {
            const float op = a.opacity[p];
            const bool opaque = op > 0.99f;
            s_op += op;
            m_rgb = m_rgb && (a.grad_mask[p] != 0) && opaque;
            m_d = m_d && opaque;
        }
        if (m_rgb) {
            const float r0 = ea * a.render[p] + eb - a.gt_rgb[p];
            const float r1 = ea * a.render[HW + p] + eb - a.gt_rgb[HW + p];
            const float r2 = ea * a.render[2 * HW + p] + eb - a.gt_rgb[2 * HW + p];
            s_rgb += fabsf(r0) + fabsf(r1) + fabsf(r2);
            c_rgb += 3.f;
        }
        if (m_d) {
            s_d += fabsf(a.depth[p] - gd);
            c_d += 1.f;
        }
    }
    s_rgb = block_sum(s_rgb, smem);
    c_rgb = block_sum(c_rgb, smem);
    s_d = block_sum(s_d, smem);
    c_d = block_sum(c_d, smem);
    s_op = block_sum(s_op, smem);
    if (threadIdx.x == 0) {
        float* o = part + LP_N + (size_t)blockIdx.x * LP_PART;
        o[0] = s_rgb; o[1] = c_rgb; o[2] = s_d; o[3] = c_d; o[4] = s_op;
    }
}

// one wave: sum the per-workgroup partials in a fixed order (bitwise reproducible) and finish the scalar
__global__ void loss_finalize_kernel(LossArgs a, int nblocks, float* __restrict__ part, float* __restrict__ loss_out) {
    const int lane = threadIdx.x;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = lane; b < nblocks; b += WAVE) {          // fixed order: lane l adds blocks l, l+64, ...
        const float* o = part + LP_N + (size_t)b * LP_PART;
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += o[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
    if (lane == 0) {
        const size_t HW = (size_t)a.W * a.H;
        const float S_rgb = v[0], C_rgb = v[1], S_d = v[2], C_d = v[3], S_op = v[4];
        float l1_rgb, l1_d, scale_rgb, scale_d, loss;
        if (a.tracking) {
            const float mean_op = S_op / (float)HW;
            l1_rgb = mean_op * (S_rgb / (3.f * (float)HW));
            l1_d = C_d > 0.f ? S_d / C_d : 0.f;
            scale_rgb = 0.5f * mean_op / (3.f * (float)HW);
            scale_d = C_d > 0.f ? 1.f / C_d : 0.f;
            loss = 0.5f * l1_rgb + l1_d;
        } else {
            l1_rgb = S_rgb / C_rgb;                       // NaN when the mask is empty, like torch's mean of nothing
            l1_d = S_d / C_d;
            scale_rgb = a.lambda_rgb / C_rgb;
            scale_d = (1.f - a.lambda_rgb) / C_d;
            loss = a.lambda_rgb * l1_rgb + (1.f - a.lambda_rgb) * l1_d;
        }
        part[LP_L1_RGB] = l1_rgb;
        part[LP_L1_D] = l1_d;
        part[LP_SCALE_RGB] = scale_rgb;
        part[LP_SCALE_D] = scale_d;
        part[LP_DAB] = 0.f;
        part[LP_DAB + 1] = 0.f;
        loss_out[0] = loss;
    }
}

__global__ void __launch_bounds__(LS_THREADS) loss_backward_kernel(LossArgs a, const float* __restrict__ part,
                                                                   const float* __restrict__ grad_out,
                                                                   float* __restrict__ d_render,
                                                                   float* __restrict__ d_depth,
                                                                   float* __restrict__ d_ab) {
    __shared__ float smem[4];
    const size_t HW = (size_t)a.W * a.H;
    const float go = grad_out ? grad_out[0] : 1.f;
    const float ea = a.init ? 1.f : __expf(a.exp_a[0]), eb = a.init ? 0.f : a.exp_b[0];
    const float k_rgb = go * part[LP_SCALE_RGB], k_d = go * part[LP_SCALE_D];
    float g_a = 0.f, g_b = 0.f;
    for (size_t p = (size_t)blockIdx.x * LS_THREADS + threadIdx.x; p < HW; p += (size_t)gridDim.x * LS_THREADS) {
        const float gd = a.gt_depth[p];
        bool m_rgb = a.mask ? a.mask[p] != 0 : true;
        bool m_d = gd > 0.f;
        if (a.tracking) {
            const bool opaque = a.opacity[p] > 0.99f;
            m_rgb = m_rgb && (a.grad_mask[p] != 0) && opaque;
            m_d = m_d && opaque;
        }
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        if (m_rgb) {
            const float x0 = a.render[p], x1 = a.render[HW + p], x2 = a.render[2 * HW + p];
            const float s0 = sgn(ea * x0 + eb - a.gt_rgb[p]);
            const float s1 = sgn(ea * x1 + eb - a.gt_rgb[HW + p]);
            const float s2 = sgn(ea * x2 + eb - a.gt_rgb[2 * HW + p]);
            o0 = k_rgb * ea * s0; o1 = k_rgb * ea * s1; o2 = k_rgb * ea * s2;
            g_a += k_rgb * ea * (s0 * x0 + s1 * x1 + s2 * x2);      // d/da of exp(a) x + b
            g_b += k_rgb * (s0 + s1 + s2);
        }
        d_render[p] = o0; d_render[HW + p] = o1; d_render[2 * HW + p] = o2;
        d_depth[p] = m_d ? k_d * sgn(a.depth[p] - gd) : 0.f;
    }
    if (d_ab && !a.init) {
        g_a = block_sum(g_a, smem);
        g_b = block_sum(g_b, smem);
        if (threadIdx.x == 0) {
            if (g_a != 0.f) atomicAdd(d_ab, g_a);
            if (g_b != 0.f) atomicAdd(d_ab + 1, g_b);
        }
    }
}

static int loss_grid(int W, int H) {
    const size_t HW = (size_t)W * H;
    size_t nb = (HW + LS_THREADS * 4 - 1) / (LS_THREADS * 4);
    return (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}
static int loss_fwd_grid(int W, int H) {
    const int g = loss_grid(W, H);
    return g > LS_MAX_BLOCKS ? LS_MAX_BLOCKS : g;
}

int launch_loss_forward(const LossArgs& a, float* partials, float* loss_out, hipStream_t s) {
    const int nb = loss_fwd_grid(a.W, a.H);
    hipLaunchKernelGGL(loss_forward_kernel, dim3(nb), dim3(LS_THREADS), 0, s, a, partials);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(WAVE), 0, s, a, nb, partials, loss_out);
    MGS_HIP(hipGetLastError());
    return 0;
}

int launch_loss_backward(const LossArgs& a, const float* partials, const float* grad_out, float* d_render,
                         float* d_depth, float* d_ab, hipStream_t s) {
    if (d_ab && d_ab != partials + LP_DAB) MGS_HIP(zero_fill(d_ab, 2 * sizeof(float), s));
    hipLaunchKernelGGL(loss_backward_kernel, dim3(loss_grid(a.W, a.H)), dim3(LS_THREADS), 0, s, a, partials, grad_out,
                       d_render, d_depth, d_ab);
    MGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace mgs

using namespace mgs;

extern "C" {

static int fill_args(LossArgs& a, int32_t W, int32_t H, int32_t tracking, int32_t init, float lambda_rgb,
                     const float* render, const float* depth, const float* opacity, const float* gt_rgb,
                     const float* gt_depth, const uint8_t* mask, const uint8_t* grad_mask, const float* exp_a,
                     const float* exp_b) {
    if (W <= 0 || H <= 0) { set_error("image size must be positive"); return 1; }
    if (!render || !depth || !gt_rgb || !gt_depth) { set_error("render, depth, gt_rgb, gt_depth must be non-NULL"); return 1; }
    if (tracking && (!opacity || !grad_mask)) { set_error("tracking loss needs opacity and grad_mask"); return 1; }
    if (!init && (!exp_a || !exp_b)) { set_error("exposure_a / exposure_b must be non-NULL unless init"); return 1; }
    a.render = render; a.depth = depth; a.opacity = opacity; a.gt_rgb = gt_rgb; a.gt_depth = gt_depth;
    a.exp_a = exp_a; a.exp_b = exp_b; a.mask = mask; a.grad_mask = grad_mask;
    a.W = W; a.H = H; a.tracking = tracking; a.init = init; a.lambda_rgb = lambda_rgb;
    return 0;
}

size_t mgs_loss_scratch_bytes(void) { return (LP_N + LS_MAX_BLOCKS * LP_PART) * sizeof(float); }

int mgs_loss_forward(int32_t W, int32_t H, int32_t tracking, int32_t init, float lambda_rgb, const float* render,
                     const float* depth, const float* opacity, const float* gt_rgb, const float* gt_depth,
                     const uint8_t* mask, const uint8_t* grad_mask, const float* exposure_a, const float* exposure_b,
                     float* scratch, float* loss_out, void* stream) {
    LossArgs a;
    if (fill_args(a, W, H, tracking, init, lambda_rgb, render, depth, opacity, gt_rgb, gt_depth, mask, grad_mask,
                  exposure_a, exposure_b)) return 1;
    if (!scratch || !loss_out) { set_error("scratch and loss_out must be non-NULL"); return 1; }
    return launch_loss_forward(a, scratch, loss_out, (hipStream_t)stream);
}

int mgs_loss_backward(int32_t W, int32_t H, int32_t tracking, int32_t init, float lambda_rgb, const float* render,
                      const float* depth, const float* opacity, const float* gt_rgb, const float* gt_depth,
                      const uint8_t* mask, const uint8_t* grad_mask, const float* exposure_a, const float* exposure_b,
                      const float* scratch, const float* grad_out, float* d_render, float* d_depth, float* d_exposure,
                      void* stream) {
    LossArgs a;
    if (fill_args(a, W, H, tracking, init, lambda_rgb, render, depth, opacity, gt_rgb, gt_depth, mask, grad_mask,
                  exposure_a, exposure_b)) return 1;
    if (!scratch || !d_render || !d_depth) { set_error("scratch, d_render, d_depth must be non-NULL"); return 1; }
    return launch_loss_backward(a, scratch, grad_out, d_render, d_depth, d_exposure, (hipStream_t)stream);
}

}  // extern "C"
