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
    int n;
    float beta1, beta2, eps;
    int step;                            // used when step_dev is NULL
    const int* step_dev;                 // device counter, already incremented for this step
};

__global__ void adam_bump_kernel(int* step_dev) { step_dev[0] += 1; }

__global__ void __launch_bounds__(256) adam_kernel(AdamArgs a) {
    const int step = a.step_dev ? a.step_dev[0] : a.step;
    const float bc1 = 1.f - powf(a.beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(a.beta2, (float)step));
    const uint64_t total = a.end[a.n - 1];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        int t = 0;
        while (i >= a.end[t]) ++t;
        const uint64_t j = i - (t ? a.end[t - 1] : 0ull);
        const float g = a.grad[t] ? a.grad[t][j] : 0.f;
        const float m = a.beta1 * a.m[t][j] + (1.f - a.beta1) * g;
        const float v = a.beta2 * a.v[t][j] + (1.f - a.beta2) * g * g;
        a.m[t][j] = m;
        a.v[t][j] = v;
        const float denom = sqrtf(v) / bc2_sqrt + a.eps;
        a.param[t][j] -= (a.lr[t] / bc1) * (m / denom);
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

}  // namespace mgs

using namespace mgs;

extern "C" {

int mgs_adam_step(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const uint64_t* numel, const float* lr, float beta1, float beta2,
                  float eps, int32_t step, int32_t* step_counter, void* stream) {
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
    a.n = n_tensors; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.step = step; a.step_dev = step_counter;
    if (run == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (step_counter) hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(1), 0, s, step_counter);
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
