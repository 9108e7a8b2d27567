// Fused optimiser step for the Gaussian parameter groups and the densification statistics that follow
// every mapping iteration.
//
//   * mgs_adam_step: torch.optim.Adam (defaults: no weight decay, no amsgrad) over up to 8 tensors with one
//     learning rate each -- the five groups xyz / f_dc / opacity / scaling / rotation the reference builds at
//     /root/reference/gaussian_splatting/scene/gaussian_model.py:398-442 -- in ONE launch; the step count lives
//     on the device, so the launch is hipGraph-capturable.
//   * mgs_densify_stats: for one rendered keyframe,
//         xyz_gradient_accum[v] += || viewspace_grad[v, :2] ||,  denom[v] += 1      (gaussian_model.py:888-892)
//         max_radii_2d[v] = max(max_radii_2d[v], radii[v])                          (utils/slam_mapper.py:453-457)
//     over the visible Gaussians v (radii > 0), one launch instead of ~8 boolean-index kernels.
// Both are HBM-bound elementwise kernels (Adam: 16 B read + 12 B written per parameter).
#include "common.h"

namespace mgs {

constexpr int ADAM_MAX_TENSORS = 8;
struct AdamArgs {
    float* param[ADAM_MAX_TENSORS];
    const float* grad[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    uint64_t end[ADAM_MAX_TENSORS];      // exclusive prefix end of each tensor in the flattened index space
    float lr[ADAM_MAX_TENSORS];
    const float* lr_dev;                 // [n] learning rates in device memory (a schedule stepped on the device), or NULL: lr[]
    int n;
    double beta1, beta2;                 // doubles, as torch keeps them: 1 - beta and the bias corrections are formed in double
    float eps;
    int step;                            // used when step_dev is NULL
    const int* step_dev;                 // [n] device counters (one per tensor), already incremented for this step
};

// one counter per tensor, as torch keeps state["step"] per parameter: a tensor without a gradient is skipped and
// its count does not advance (Adam's bias correction then stays in step with torch after map surgery)
__global__ void adam_bump_kernel(AdamArgs a, int* step_dev) {
    const int t = threadIdx.x;
    if (t < a.n && a.grad[t]) step_dev[t] += 1;
}

__global__ void __launch_bounds__(256) adam_kernel(AdamArgs a) {
    __shared__ float s_bc1[ADAM_MAX_TENSORS], s_bc2[ADAM_MAX_TENSORS], s_lr[ADAM_MAX_TENSORS];
    if ((int)threadIdx.x < a.n) {
        const int step = a.step_dev ? a.step_dev[threadIdx.x] : a.step;
        s_bc1[threadIdx.x] = (float)(1.0 - pow(a.beta1, (double)step));
        s_bc2[threadIdx.x] = (float)sqrt(1.0 - pow(a.beta2, (double)step));
        s_lr[threadIdx.x] = a.lr_dev ? a.lr_dev[threadIdx.x] : a.lr[threadIdx.x];
    }
    __syncthreads();
    const uint64_t total = a.end[a.n - 1];
    const float b1 = (float)a.beta1, b2 = (float)a.beta2, w1 = (float)(1.0 - a.beta1), w2 = (float)(1.0 - a.beta2);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        int t = 0;
        while (i >= a.end[t]) ++t;
        if (!a.grad[t]) continue;            // no gradient this step: torch.optim.Adam skips the tensor (state untouched)
        const uint64_t j = i - (t ? a.end[t - 1] : 0ull);
        const float g = a.grad[t][j];
        const float m = b1 * a.m[t][j] + w1 * g;
        const float v = b2 * a.v[t][j] + w2 * g * g;
        a.m[t][j] = m;
        a.v[t][j] = v;
        const float denom = sqrtf(v) / s_bc2[t] + a.eps;
        a.param[t][j] -= (s_lr[t] / s_bc1[t]) * (m / denom);
    }
}

__global__ void __launch_bounds__(256) densify_stats_kernel(int P, const float* __restrict__ vs_grad /* [P,3] */,
                                                            const int32_t* __restrict__ radii,
                                                            float* __restrict__ grad_accum, float* __restrict__ denom,
                                                            float* __restrict__ max_radii) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    if (grad_accum) {
        const float gx = vs_grad[3 * i], gy = vs_grad[3 * i + 1];
        grad_accum[i] += sqrtf(gx * gx + gy * gy);
    }
    if (denom) denom[i] += 1.f;
    if (max_radii) max_radii[i] = fmaxf(max_radii[i], (float)r);
}

// ---- the statistics of ONE mapping iteration over the keyframes a rank rendered -------------
// What /root/reference/utils/slam_mapper.py does after loss.backward() for every keyframe of the window, in one launch:
//   occ_aware_visibility[kf] = n_touched_kf > 0                                              (:400-404)   -> packed bits
//   max_radii_2d[v] = max(max_radii_2d[v], radii_kf[v]);  xyz_gradient_accum[v] += ||dL/dmean2D_kf[v, :2]||;  denom[v] += 1
//                                                              over v = radii_kf > 0        (:453-460, gaussian_model.py:888-892)
// The norm is taken per keyframe BEFORE the sum (norm of sums != sum of norms), the keyframes are added in window order --
// bit for bit what the reference's loop over keyframes leaves in the three arrays when `accumulate` (the single-rank case: the
// arrays ARE the map's running statistics).  Sharded windows pass accumulate = 0: the arrays then receive this rank's share
// of the iteration only, travel through the collectives, and mgs_window_apply folds the reduced values into the map's.
constexpr int WS_MAX_KF = 32;               // the fork's window holds up to 30 keyframes (/root/reference/slam.py:75)
struct WindowStatsArgs {
    const float* grad2d[WS_MAX_KF];         // [P,3] each: dL/dmeans2D of the keyframe's render (may be NULL: no gradient)
    const int32_t* radii[WS_MAX_KF];
    const int32_t* n_touched[WS_MAX_KF];
    int K, P, accumulate;
    float *norm, *vis, *maxr;               // [P] each
    unsigned long long* bits;               // [K][words]: bit i of word w = (n_touched[64 w + i] > 0); may be NULL
    size_t words;                           // = ceil(P / 64)
};
__global__ void __launch_bounds__(256) window_stats_kernel(WindowStatsArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool in = i < a.P;
    float norm = 0.f, vis = 0.f, maxr = 0.f;
    if (in && a.accumulate) { norm = a.norm[i]; vis = a.vis[i]; maxr = a.maxr[i]; }
    for (int k = 0; k < a.K; ++k) {
        const int r = in ? a.radii[k][i] : 0;
        if (r > 0) {
            if (a.grad2d[k]) {
                const float gx = a.grad2d[k][3 * (size_t)i], gy = a.grad2d[k][3 * (size_t)i + 1];
                norm += sqrtf(gx * gx + gy * gy);
            }
            vis += 1.f;
            maxr = fmaxf(maxr, (float)r);
        }
        if (a.bits) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(in && a.n_touched[k][i] > 0);
            if ((threadIdx.x & 63) == 0 && (size_t)(i >> 6) < a.words) a.bits[(size_t)k * a.words + (size_t)(i >> 6)] = m;
        }
    }
    if (in) { a.norm[i] = norm; a.vis[i] = vis; a.maxr[i] = maxr; }
}
__global__ void __launch_bounds__(256) window_apply_kernel(int P, const float* __restrict__ d_norm,
                                                           const float* __restrict__ d_vis, const float* __restrict__ d_maxr,
                                                           float* __restrict__ accum, float* __restrict__ denom,
                                                           float* __restrict__ maxr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    accum[i] += d_norm[i];
    denom[i] += d_vis[i];
    maxr[i] = fmaxf(maxr[i], d_maxr[i]);
}
// GaussianModel.update_learning_rate (gaussian_model.py:451-465 -> general_utils.helper, :79-94) stepped on the device:
// iteration += 1; lr = schedule(iteration).  One thread, double precision as the host computes it.
__global__ void lr_schedule_kernel(int* iteration, float* lr_out, double lr_init, double lr_final, int delay_steps,
                                   double delay_mult, int max_steps) {
    const int it = iteration[0] + 1;
    iteration[0] = it;
    double lr = 0.0;
    if (it >= 0 && !(lr_init == 0.0 && lr_final == 0.0)) {
        double delay = 1.0;
        if (delay_steps > 0) {
            const double x = fmin(fmax((double)it / (double)delay_steps, 0.0), 1.0);
            delay = delay_mult + (1.0 - delay_mult) * sin(0.5 * 3.14159265358979323846 * x);
        }
        const double t = fmin(fmax((double)it / (double)max_steps, 0.0), 1.0);
        lr = delay * exp(log(lr_init) * (1.0 - t) + log(lr_final) * t);
    }
    lr_out[0] = (float)lr;
}

}  // namespace mgs

using namespace mgs;

extern "C" {

int mgs_window_stats(int32_t P, int32_t n_keyframes, const float* const* grad_means2D, const int32_t* const* radii,
                     const int32_t* const* n_touched, float* grad_norm, float* visible, float* max_radii,
                     int32_t accumulate, uint64_t* visibility_bits, void* stream) {
    if (P < 0 || n_keyframes < 0 || n_keyframes > WS_MAX_KF) { set_error("mgs_window_stats: 0..32 keyframes"); return 1; }
    if (P == 0 || n_keyframes == 0) return 0;
    if (!grad_means2D || !radii || !grad_norm || !visible || !max_radii || (visibility_bits && !n_touched)) {
        set_error("mgs_window_stats: NULL argument");
        return 1;
    }
    WindowStatsArgs a;
    for (int k = 0; k < WS_MAX_KF; ++k) {
        const bool on = k < n_keyframes;
        if (on && (!radii[k] || (visibility_bits && !n_touched[k]))) { set_error("mgs_window_stats: NULL keyframe tensor"); return 1; }
        a.grad2d[k] = on ? grad_means2D[k] : nullptr;
        a.radii[k] = on ? radii[k] : nullptr;
        a.n_touched[k] = (on && n_touched) ? n_touched[k] : nullptr;
    }
    a.K = n_keyframes; a.P = P; a.accumulate = accumulate ? 1 : 0;
    a.norm = grad_norm; a.vis = visible; a.maxr = max_radii;
    a.bits = (unsigned long long*)visibility_bits;
    a.words = ((size_t)P + 63) / 64;
    hipLaunchKernelGGL(window_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    MGS_HIP(hipGetLastError());
    return 0;
}

int mgs_window_apply(int32_t P, const float* grad_norm, const float* visible, const float* max_radii,
                     float* xyz_gradient_accum, float* denom, float* max_radii_2d, void* stream) {
    if (P < 0 || (P > 0 && (!grad_norm || !visible || !max_radii || !xyz_gradient_accum || !denom || !max_radii_2d))) {
        set_error("mgs_window_apply: bad arguments");
        return 1;
    }
    if (P == 0) return 0;
    hipLaunchKernelGGL(window_apply_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, grad_norm, visible,
                       max_radii, xyz_gradient_accum, denom, max_radii_2d);
    MGS_HIP(hipGetLastError());
    return 0;
}

int mgs_lr_schedule_step(int32_t* iteration, float* lr_out, double lr_init, double lr_final, int32_t lr_delay_steps,
                         double lr_delay_mult, int32_t max_steps, void* stream) {
    if (!iteration || !lr_out || max_steps <= 0) { set_error("mgs_lr_schedule_step: bad arguments"); return 1; }
    hipLaunchKernelGGL(lr_schedule_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, iteration, lr_out, lr_init, lr_final,
                       lr_delay_steps, lr_delay_mult, max_steps);
    MGS_HIP(hipGetLastError());
    return 0;
}

int mgs_adam_step(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const uint64_t* numel, const float* lr, double beta1, double beta2,
                  double eps, int32_t step, int32_t* step_counter, const float* lr_dev, void* stream) {
    if (n_tensors < 1 || n_tensors > ADAM_MAX_TENSORS) { set_error("mgs_adam_step: 1..8 tensors"); return 1; }
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr) { set_error("mgs_adam_step: NULL table"); return 1; }
    if (!step_counter && step < 1) { set_error("step is 1-based"); return 1; }
    AdamArgs a;
    uint64_t run = 0;
    for (int t = 0; t < ADAM_MAX_TENSORS; ++t) {
        const bool on = t < n_tensors;
        if (on && (!params[t] || !exp_avg[t] || !exp_avg_sq[t])) { set_error("mgs_adam_step: NULL tensor"); return 1; }
        a.param[t] = on ? params[t] : nullptr;
        a.grad[t] = on ? grads[t] : nullptr;
        a.m[t] = on ? exp_avg[t] : nullptr;
        a.v[t] = on ? exp_avg_sq[t] : nullptr;
        if (on) run += numel[t];
        a.end[t] = run;
        a.lr[t] = on ? lr[t] : 0.f;
    }
    a.n = n_tensors; a.beta1 = beta1; a.beta2 = beta2; a.eps = (float)eps; a.step = step; a.step_dev = step_counter;
    a.lr_dev = lr_dev;
    if (run == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (step_counter) hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(ADAM_MAX_TENSORS), 0, s, a, step_counter);
    const uint64_t blocks = (run + 255) / 256;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, a);
    MGS_HIP(hipGetLastError());
    return 0;
}

int mgs_densify_stats(int32_t P, const float* viewspace_grad, const int32_t* radii, float* xyz_gradient_accum,
                      float* denom, float* max_radii_2d, void* stream) {
    if (P < 0 || (P > 0 && (!radii || (xyz_gradient_accum && !viewspace_grad)))) { set_error("bad arguments"); return 1; }
    if (P == 0) return 0;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, viewspace_grad,
                       radii, xyz_gradient_accum, denom, max_radii_2d);
    MGS_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Fused activations of the map: what GaussianModel.get_rotation / get_scaling /
// get_opacity compute with F.normalize / exp / sigmoid (/root/reference/gaussian_splatting/scene/
// gaussian_model.py:84-106) plus the isotropic scale expansion of render()
// (/root/reference/gaussian_splatting/gaussian_renderer/__init__.py:101-104), forward and backward, one launch each.
// ------------------------------------------------------------------------------------------------
namespace mgs {

__global__ void __launch_bounds__(256) activate_forward_kernel(int P, int scale_dim, const float* __restrict__ rot_raw,
                                                               const float* __restrict__ scale_raw,
                                                               const float* __restrict__ opac_raw,
                                                               float* __restrict__ rot, float* __restrict__ scales3,
                                                               float* __restrict__ opac) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float4 q = reinterpret_cast<const float4*>(rot_raw)[i];
    const float n = fmaxf(sqrtf((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w)), 1e-12f);   // F.normalize eps
    const float inv = 1.f / n;
    reinterpret_cast<float4*>(rot)[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    if (scale_dim == 1) {
        const float s = expf(scale_raw[i]);
        scales3[3 * i] = s; scales3[3 * i + 1] = s; scales3[3 * i + 2] = s;
    } else {
        scales3[3 * i] = expf(scale_raw[3 * i]);
        scales3[3 * i + 1] = expf(scale_raw[3 * i + 1]);
        scales3[3 * i + 2] = expf(scale_raw[3 * i + 2]);
    }
    opac[i] = 1.f / (1.f + expf(-opac_raw[i]));
}

__global__ void __launch_bounds__(256) activate_backward_kernel(int P, int scale_dim, const float* __restrict__ rot_raw,
                                                                const float* __restrict__ scales3,
                                                                const float* __restrict__ opac,
                                                                const float* __restrict__ g_rot,
                                                                const float* __restrict__ g_scales3,
                                                                const float* __restrict__ g_opac,
                                                                float* __restrict__ d_rot_raw,
                                                                float* __restrict__ d_scale_raw,
                                                                float* __restrict__ d_opac_raw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    if (d_rot_raw) {
        const float4 q = reinterpret_cast<const float4*>(rot_raw)[i];
        const float4 g = g_rot ? reinterpret_cast<const float4*>(g_rot)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float n = sqrtf((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w));
        float4 o;
        if (n > 1e-12f) {                 // d(q/|q|) = (g - u (u.g)) / |q|
            const float inv = 1.f / n;
            const float ux = q.x * inv, uy = q.y * inv, uz = q.z * inv, uw = q.w * inv;
            const float ug = (ux * g.x + uy * g.y) + (uz * g.z + uw * g.w);
            o = make_float4((g.x - ux * ug) * inv, (g.y - uy * ug) * inv, (g.z - uz * ug) * inv, (g.w - uw * ug) * inv);
        } else {                          // clamped denominator: y = q / eps
            o = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
        }
        reinterpret_cast<float4*>(d_rot_raw)[i] = o;
    }
    if (d_scale_raw) {
        const float gx = g_scales3 ? g_scales3[3 * i] : 0.f, gy = g_scales3 ? g_scales3[3 * i + 1] : 0.f;
        const float gz = g_scales3 ? g_scales3[3 * i + 2] : 0.f;
        if (scale_dim == 1) {
            d_scale_raw[i] = scales3[3 * i] * ((gx + gy) + gz);
        } else {
            d_scale_raw[3 * i] = scales3[3 * i] * gx;
            d_scale_raw[3 * i + 1] = scales3[3 * i + 1] * gy;
            d_scale_raw[3 * i + 2] = scales3[3 * i + 2] * gz;
        }
    }
    if (d_opac_raw) {
        const float s = opac[i];
        d_opac_raw[i] = (g_opac ? g_opac[i] : 0.f) * s * (1.f - s);
    }
}

}  // namespace mgs

extern "C" {

int mgs_activate_forward(int32_t P, int32_t scale_dim, const float* rot_raw, const float* scale_raw,
                         const float* opacity_raw, float* rotations, float* scales3, float* opacities, void* stream) {
    if (P < 0 || (scale_dim != 1 && scale_dim != 3)) { mgs::set_error("mgs_activate_forward: bad P / scale_dim"); return 1; }
    if (P == 0) return 0;
    if (!rot_raw || !scale_raw || !opacity_raw || !rotations || !scales3 || !opacities) { mgs::set_error("NULL tensor"); return 1; }
    hipLaunchKernelGGL(mgs::activate_forward_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, scale_dim,
                       rot_raw, scale_raw, opacity_raw, rotations, scales3, opacities);
    MGS_HIP(hipGetLastError());
    return 0;
}

int mgs_activate_backward(int32_t P, int32_t scale_dim, const float* rot_raw, const float* scales3,
                          const float* opacities, const float* grad_rotations, const float* grad_scales3,
                          const float* grad_opacities, float* d_rot_raw, float* d_scale_raw, float* d_opacity_raw,
                          void* stream) {
    if (P < 0 || (scale_dim != 1 && scale_dim != 3)) { mgs::set_error("mgs_activate_backward: bad P / scale_dim"); return 1; }
    if (P == 0) return 0;
    if (!rot_raw || !scales3 || !opacities) { mgs::set_error("NULL tensor"); return 1; }
    hipLaunchKernelGGL(mgs::activate_backward_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P,
                       scale_dim, rot_raw, scales3, opacities, grad_rotations, grad_scales3, grad_opacities, d_rot_raw,
                       d_scale_raw, d_opacity_raw);
    MGS_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// dst[i] = sum_k src_k[i]: the gradients that the N keyframe renders of a mapping window return for the SAME map tensor,
// added up in ONE launch (autograd's engine adds them pairwise: N - 1 launches per shared tensor, ~20 per iteration for a
// window of four -- /root/reference/utils/slam_mapper.py:273-394 sums the losses and calls backward once).
// The order of the adds is fixed (k = 0, 1, ...): bitwise reproducible.
// ------------------------------------------------------------------------------------------------
namespace mgs {
constexpr int SUM_MAX_SRC = 16;
struct SumArgs {
    const float* src[SUM_MAX_SRC];
    float* dst;
    size_t count;
    int n, vec;
};
__global__ void __launch_bounds__(256) sum_buffers_kernel(SumArgs a) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if (a.vec) {
        const size_t nv = a.count / 4;
        for (size_t i = i0; i < nv; i += stride) {
            float4 acc = reinterpret_cast<const float4*>(a.src[0])[i];
            for (int k = 1; k < a.n; ++k) {
                const float4 v = reinterpret_cast<const float4*>(a.src[k])[i];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            reinterpret_cast<float4*>(a.dst)[i] = acc;
        }
        for (size_t i = nv * 4 + i0; i < a.count; i += stride) {
            float acc = a.src[0][i];
            for (int k = 1; k < a.n; ++k) acc += a.src[k][i];
            a.dst[i] = acc;
        }
    } else {
        for (size_t i = i0; i < a.count; i += stride) {
            float acc = a.src[0][i];
            for (int k = 1; k < a.n; ++k) acc += a.src[k][i];
            a.dst[i] = acc;
        }
    }
}
}  // namespace mgs

extern "C" int mgs_sum_buffers(int32_t n_src, const float* const* src /* host array of device pointers */, float* dst,
                               uint64_t count, void* stream) {
    using namespace mgs;
    if (n_src < 1 || n_src > SUM_MAX_SRC || !src || !dst) { set_error("mgs_sum_buffers: 1..16 sources, non-NULL pointers"); return 1; }
    if (count == 0) return 0;
    SumArgs a;
    bool vec = ((size_t)dst % 16) == 0;
    for (int k = 0; k < n_src; ++k) {
        if (!src[k]) { set_error("mgs_sum_buffers: NULL source"); return 1; }
        a.src[k] = src[k];
        vec = vec && ((size_t)src[k] % 16) == 0;
    }
    for (int k = n_src; k < SUM_MAX_SRC; ++k) a.src[k] = nullptr;
    a.dst = dst; a.count = (size_t)count; a.n = n_src; a.vec = vec ? 1 : 0;
    const size_t items = vec ? (count + 3) / 4 : count;
    size_t blocks = (items + 255) / 256;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(sum_buffers_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    MGS_HIP(hipGetLastError());
    return 0;
}
