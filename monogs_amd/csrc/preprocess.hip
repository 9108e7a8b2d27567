// Per-Gaussian kernels: forward projection (cull, EWA covariance, radius, tile rectangle, colour)
// and the matching backward (dL/d{mean3D, scale, rotation, cov3D, SH, tau}).
//
// Boundary replaced: upstream `preprocessCUDA` forward/backward + `computeCov2DCUDA` backward of the
// un-vendored rasteriser; maths corroborated by
// /root/reference/viewer/gl_render/shaders/gau_vert.glsl:60-107,149-154,173-210 and
// /root/reference/gaussian_splatting/utils/general_utils.py:113-136; pose convention from
// /root/reference/utils/pose_utils.py:61-93.
//
// The geometry that decides integers (radius, tile rectangle) and sort keys (depth bits) is computed
// with one IEEE operation per source operator in a fixed association order, and this file is built
// with FP contraction off, so a float32 CPU restatement reproduces those integers bit for bit.
#include "common.h"

#pragma clang fp contract(off)

namespace mgs {

__device__ __forceinline__ float dot3p(float m0, float m1, float m2, float m3, float x, float y, float z) {
    return ((m0 * x + m1 * y) + m2 * z) + m3;
}

// SH basis constants (/root/reference/gaussian_splatting/utils/sh_utils.py:24-52)
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

struct Cam {
    float V[16], PM[16], PR[16];
    float campos[3];
    float tanfovx, tanfovy, focal_x, focal_y, mod;
    int W, H, gx, gy, deg, M;
};

__device__ __forceinline__ void load_cam(Cam& c, const float* V_, const float* PM_, const float* PR_,
                                         const float* campos_) {
    const const_float_p V = MGS_CONST(V_), PM = MGS_CONST(PM_), PR = MGS_CONST(PR_), campos = MGS_CONST(campos_);
#pragma unroll
    for (int i = 0; i < 16; ++i) { c.V[i] = V[i]; c.PM[i] = PM[i]; }
    if (PR_) {                   // (wave-uniform; the forward does not need the raw projection)
#pragma unroll
        for (int i = 0; i < 16; ++i) c.PR[i] = PR[i];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) c.PR[i] = 0.f;
    }
    c.campos[0] = campos[0]; c.campos[1] = campos[1]; c.campos[2] = campos[2];
}

// An empty asm that "uses" the camera block: pins its scalar loads in front of the arithmetic (the compiler otherwise
// sinks each group to its first use, i.e. behind the per-lane loads' wait: a second, dependent round trip).
__device__ __forceinline__ void pin_cam(const Cam& c, bool with_pr) {
    asm volatile("" ::"s"(c.V[0]), "s"(c.V[1]), "s"(c.V[2]), "s"(c.V[3]), "s"(c.V[4]), "s"(c.V[5]), "s"(c.V[6]), "s"(c.V[7]),
                 "s"(c.V[8]), "s"(c.V[9]), "s"(c.V[10]), "s"(c.V[11]), "s"(c.V[12]), "s"(c.V[13]), "s"(c.V[14]), "s"(c.V[15]));
    asm volatile("" ::"s"(c.PM[0]), "s"(c.PM[1]), "s"(c.PM[2]), "s"(c.PM[3]), "s"(c.PM[4]), "s"(c.PM[5]), "s"(c.PM[6]),
                 "s"(c.PM[7]), "s"(c.PM[8]), "s"(c.PM[9]), "s"(c.PM[10]), "s"(c.PM[11]), "s"(c.PM[12]), "s"(c.PM[13]),
                 "s"(c.PM[14]), "s"(c.PM[15]), "s"(c.campos[0]), "s"(c.campos[1]), "s"(c.campos[2]));
    if (with_pr)
        asm volatile("" ::"s"(c.PR[0]), "s"(c.PR[1]), "s"(c.PR[2]), "s"(c.PR[3]), "s"(c.PR[4]), "s"(c.PR[5]), "s"(c.PR[6]),
                     "s"(c.PR[7]), "s"(c.PR[8]), "s"(c.PR[9]), "s"(c.PR[10]), "s"(c.PR[11]), "s"(c.PR[12]), "s"(c.PR[13]),
                     "s"(c.PR[14]), "s"(c.PR[15]));
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float s[3], const float q[4], float mod, float cov[6],
                                                     float Mm[9]) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    float Rm[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)};
    const float sx = s[0] * mod, sy = s[1] * mod, sz = s[2] * mod;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Mm[3 * i + 0] = Rm[3 * i + 0] * sx;
        Mm[3 * i + 1] = Rm[3 * i + 1] * sy;
        Mm[3 * i + 2] = Rm[3 * i + 2] * sz;
    }
    auto e = [&](int i, int j) {
        return (Mm[3 * i] * Mm[3 * j] + Mm[3 * i + 1] * Mm[3 * j + 1]) + Mm[3 * i + 2] * Mm[3 * j + 2];
    };
    cov[0] = e(0, 0); cov[1] = e(0, 1); cov[2] = e(0, 2);
    cov[3] = e(1, 1); cov[4] = e(1, 2); cov[5] = e(2, 2);
}

// EWA projection intermediates shared by forward and backward
struct Proj {
    float tx, ty, tz;        // clamped view-space mean used by the Jacobian
    bool clamp_x, clamp_y;
    float J00, J02, J11, J12;
    float T0[3], T1[3];
    float cxx, cxy, cyy;     // 2-D covariance incl. the +0.3 low-pass
};

__device__ __forceinline__ void project_cov(const Cam& c, const float pv[3], const float cov[6], Proj& p) {
    const float limx = 1.3f * c.tanfovx, limy = 1.3f * c.tanfovy;
    const float tz = pv[2];
    const float txtz = pv[0] / tz, tytz = pv[1] / tz;
    p.clamp_x = (txtz < -limx) || (txtz > limx);
    p.clamp_y = (tytz < -limy) || (tytz > limy);
    p.tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    p.ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    p.tz = tz;
    const float tz2 = tz * tz;
    p.J00 = c.focal_x / tz;
    p.J02 = -(c.focal_x * p.tx) / tz2;
    p.J11 = c.focal_y / tz;
    p.J12 = -(c.focal_y * p.ty) / tz2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {       // Rv(i,j) = V[4*j+i]
        p.T0[j] = p.J00 * c.V[4 * j + 0] + p.J02 * c.V[4 * j + 2];
        p.T1[j] = p.J11 * c.V[4 * j + 1] + p.J12 * c.V[4 * j + 2];
    }
    const float S[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
    float U0[3], U1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        U0[j] = (p.T0[0] * S[0][j] + p.T0[1] * S[1][j]) + p.T0[2] * S[2][j];
        U1[j] = (p.T1[0] * S[0][j] + p.T1[1] * S[1][j]) + p.T1[2] * S[2][j];
    }
    p.cxx = ((U0[0] * p.T0[0] + U0[1] * p.T0[1]) + U0[2] * p.T0[2]) + 0.3f;
    p.cxy = (U0[0] * p.T1[0] + U0[1] * p.T1[1]) + U0[2] * p.T1[2];
    p.cyy = ((U1[0] * p.T1[0] + U1[1] * p.T1[1]) + U1[2] * p.T1[2]) + 0.3f;
}

__device__ __forceinline__ void tile_rect(float px, float py, float radius, int gx, int gy, int& x0, int& y0,
                                          int& x1, int& y1) {
    x0 = min(gx, max(0, (int)((px - radius) / (float)TILE)));
    y0 = min(gy, max(0, (int)((py - radius) / (float)TILE)));
    x1 = min(gx, max(0, (int)(((px + radius) + (float)(TILE - 1)) / (float)TILE)));
    y1 = min(gy, max(0, (int)(((py + radius) + (float)(TILE - 1)) / (float)TILE)));
}

// Direction-dependent colour.  sh: [M][3] for this Gaussian.
__device__ __forceinline__ void eval_sh(int deg, const float* sh, const float d[3], float out[3]) {
    const float x = d[0], y = d[1], z = d[2];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float res = SH_C0 * sh[ch];
        if (deg > 0) {
            res = res - SH_C1 * y * sh[3 + ch] + SH_C1 * z * sh[6 + ch] - SH_C1 * x * sh[9 + ch];
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * sh[12 + ch] + SH_C2[1] * yz * sh[15 + ch] +
                      SH_C2[2] * (2.f * zz - xx - yy) * sh[18 + ch] + SH_C2[3] * xz * sh[21 + ch] +
                      SH_C2[4] * (xx - yy) * sh[24 + ch];
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3.f * xx - yy) * sh[27 + ch] + SH_C3[1] * xy * z * sh[30 + ch] +
                          SH_C3[2] * y * (4.f * zz - xx - yy) * sh[33 + ch] +
                          SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * sh[36 + ch] +
                          SH_C3[4] * x * (4.f * zz - xx - yy) * sh[39 + ch] + SH_C3[5] * z * (xx - yy) * sh[42 + ch] +
                          SH_C3[6] * x * (xx - 3.f * yy) * sh[45 + ch];
                }
            }
        }
        out[ch] = res + 0.5f;
    }
}

struct PreArgs {
    const float *means3D, *shs, *colors, *opacities, *scales, *rotations, *cov3D;
    const float *V, *PM, *campos;
    float* rec;
    uint32_t* depth_key;
    int rect_packed;          // rect is uint32[P]: x0 | y0 << 8 | w << 16 | h << 24
    uint8_t* clamped;
    uint2* rect;
    int32_t* radii;
    float tanfovx, tanfovy, focal_x, focal_y, mod;
    int P, W, H, gx, gy, deg, M;
    int iso;                  // scales holds ONE float per Gaussian
    int rot_aligned16;        // rotations may be read with 16-byte loads
    uint32_t* zero_ptr;       // scratch of the depth sort that follows: cleared here instead of by its own launch
    size_t zero_words;
    float* grad_acc;          // optional: accumulator of the backward that will follow ([P][16] + pose slots + 16): the
                              //           lines of the visible Gaussians and the pose part are cleared here
};

// 16-byte per-lane loads where the caller's tensor allows it (torch allocations are 256-byte aligned; a sliced view may not be)
__device__ __forceinline__ void load4(const float* __restrict__ p, bool aligned16, float o[4]) {
    if (aligned16) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3];
    }
}

struct PreOut {
    float4 r0, r1, r2, r3;     // the 64-byte blend record
    int radius;
    uint2 rect;
    uint32_t key;
};

// One Gaussian, inputs already in registers.  false = culled (near plane, degenerate covariance, off screen).
template <bool COV, bool PRECOMP>
__device__ __forceinline__ bool preprocess_one(const PreArgs& a, const Cam& c, int idx, float x, float y, float z,
                                               const float cov_in[6], const float s[3], const float q[4],
                                               const float col_in[3], float opac, PreOut& o) {
    float pv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pv[k] = dot3p(c.V[k], c.V[4 + k], c.V[8 + k], c.V[12 + k], x, y, z);
    if (!(pv[2] > 0.2f)) return false;                    // near cull (also rejects NaN)
    float ph[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ph[k] = dot3p(c.PM[k], c.PM[4 + k], c.PM[8 + k], c.PM[12 + k], x, y, z);
    const float p_w = 1.0f / (ph[3] + 0.0000001f);
    const float projx = ph[0] * p_w, projy = ph[1] * p_w;

    float cov[6];
    if (COV) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cov[k] = cov_in[k];
    } else {
        float Mm[9];
        cov3d_from_scale_rot(s, q, a.mod, cov, Mm);
    }
    Proj p;
    project_cov(c, pv, cov, p);
    const float det = p.cxx * p.cyy - p.cxy * p.cxy;
    if (det == 0.0f) return false;
    const float det_inv = 1.f / det;
    const float ca = p.cyy * det_inv, cb = -p.cxy * det_inv, cc = p.cxx * det_inv;
    const float mid = 0.5f * (p.cxx + p.cyy);
    const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lam = fmaxf(mid + disc, mid - disc);
    const float radius = ceilf(3.f * sqrtf(lam));
    const float px = ((projx + 1.0f) * (float)a.W - 1.0f) * 0.5f;
    const float py = ((projy + 1.0f) * (float)a.H - 1.0f) * 0.5f;
    if (!(isfinite(px) && isfinite(py) && isfinite(radius))) return false;
    int x0, y0, x1, y1;
    tile_rect(px, py, radius, a.gx, a.gy, x0, y0, x1, y1);
    const int ntile = (x1 - x0) * (y1 - y0);
    if (ntile == 0) return false;

    float rgb[3];
    if (PRECOMP) {
        rgb[0] = col_in[0]; rgb[1] = col_in[1]; rgb[2] = col_in[2];
    } else {
        float d[3] = {x - c.campos[0], y - c.campos[1], z - c.campos[2]};
        const float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        d[0] = d[0] / len; d[1] = d[1] / len; d[2] = d[2] / len;
        eval_sh(a.deg, a.shs + (size_t)idx * a.M * 3, d, rgb);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            a.clamped[4 * idx + ch] = rgb[ch] < 0.f;
            rgb[ch] = fmaxf(rgb[ch], 0.f);
        }
    }

    // Half extents of the axis-aligned box around the alpha >= 1/255 ellipse, padded: the blend
    // kernels skip an instance for a whole 8x8 quadrant when the box misses the quadrant, which
    // cannot change any pixel because every skipped pair has alpha < 1/255.
    float ex = -1.f, ey = -1.f;
    float na = 0.f, nb = 0.f, nc = 0.f;       // conic / (2 tau): the alpha >= 1/255 ellipse is  d^T N d <= 1
    const float tau = logf(255.0f * opac);
    if (tau > 0.f) {
        ex = sqrtf(2.f * tau * p.cxx) * 1.001f + 0.05f;
        ey = sqrtf(2.f * tau * p.cyy) * 1.001f + 0.05f;
        const float it = 1.f / (2.f * tau);
        na = ca * it; nb = cb * it; nc = cc * it;
    } else if (!(tau <= 0.f)) {   // NaN opacity: never skip, let the blend propagate it
        ex = ey = 3.0e38f;
    }

    o.r0 = make_float4(px, py, ex, ey);
    // conic pre-scaled for the blend kernels: alpha = opacity * exp2(ka dx^2 + kc dy^2 + kb dx dy)
    constexpr float LOG2E = 1.4426950408889634f;
    o.r1 = make_float4(-0.5f * LOG2E * ca, -LOG2E * cb, -0.5f * LOG2E * cc, opac);
    o.r2 = make_float4(rgb[0], rgb[1], rgb[2], pv[2]);
    o.r3 = make_float4(na, nb, nc, radius);
    o.radius = (int)radius;
    o.rect = make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)(x1 - x0) | ((uint32_t)(y1 - y0) << 16));
    o.key = __float_as_uint(pv[2]);            // > 0.2, so the bit pattern orders like the value
    return true;
}

// COV: the 3-D covariance is given (else scales + rotations).  PRECOMP: colours are given (else spherical harmonics).
// Compile-time, so that the input loads below form one branch-free group.
template <bool COV, bool PRECOMP>
__global__ void __launch_bounds__(256) preprocess_forward_kernel(PreArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;        // (launched with 256 threads; blockDim would be a packet fetch)
    grid_zero(a.zero_ptr, a.zero_words, (size_t)((a.P + 255) / 256) * 256, 256);
    if (a.grad_acc && blockIdx.x == 0) {              // pose-gradient slots + the six output floats: (TAU_SLOTS + 1) lines
        float4* t4 = reinterpret_cast<float4*>(a.grad_acc + (size_t)a.P * GRAD_FLOATS);
        for (int i = threadIdx.x; i < (TAU_SLOTS + 1) * 4; i += 256) t4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (idx >= a.P) return;
    // ---- every per-Gaussian input is requested up front: ONE memory round trip before the arithmetic starts (a load
    //      placed behind each early exit used to cost a dependent round trip of its own; the 32 bytes this reads for a
    //      Gaussian that turns out to be culled are cheaper than the latency they hide)
    Cam c;
    load_cam(c, a.V, a.PM, nullptr, a.campos);
    c.tanfovx = a.tanfovx; c.tanfovy = a.tanfovy; c.focal_x = a.focal_x; c.focal_y = a.focal_y;
    const float x = a.means3D[3 * idx], y = a.means3D[3 * idx + 1], z = a.means3D[3 * idx + 2];
    const float opac = a.opacities[idx];
    float cov_in[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f};
    float col_in[3] = {0.f, 0.f, 0.f};
    if (COV) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cov_in[k] = a.cov3D[6 * idx + k];
    } else {
        // isotropic maps hold ONE scale per Gaussian: the same address three times, no branch
        const float* sp = a.scales + (a.iso ? (size_t)idx : 3 * (size_t)idx);
        const int o1 = a.iso ? 0 : 1, o2 = a.iso ? 0 : 2;
        s[0] = sp[0]; s[1] = sp[o1]; s[2] = sp[o2];
        load4(a.rotations + 4 * (size_t)idx, a.rot_aligned16 != 0, q);
    }
    if (PRECOMP) { col_in[0] = a.colors[3 * idx]; col_in[1] = a.colors[3 * idx + 1]; col_in[2] = a.colors[3 * idx + 2]; }
    // (an empty asm that "uses" every loaded value: the compiler otherwise sinks each load to its first use, behind the
    //  early exits, and the kernel pays one dependent memory round trip per group again)
    asm volatile("" ::"v"(x), "v"(y), "v"(z), "v"(opac), "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(q[0]), "v"(q[1]), "v"(q[2]),
                 "v"(q[3]), "v"(col_in[0]), "v"(col_in[1]), "v"(col_in[2]));
    if (COV) asm volatile("" ::"v"(cov_in[0]), "v"(cov_in[1]), "v"(cov_in[2]), "v"(cov_in[3]), "v"(cov_in[4]), "v"(cov_in[5]));
    pin_cam(c, false);

    PreOut o;
    const bool vis = preprocess_one<COV, PRECOMP>(a, c, idx, x, y, z, cov_in, s, q, col_in, opac, o);
    // ---- every output written once (culled Gaussians: radius 0, empty rectangle, key behind every visible one)
    a.radii[idx] = vis ? o.radius : 0;
    if (a.rect_packed) {       // (large maps on grids of <= 255 x 255 tiles: the depth sort carries this word with the pair)
        const uint32_t r4 = (o.rect.x & 0xFFu) | ((o.rect.x >> 16) << 8) | ((o.rect.y & 0xFFu) << 16) | ((o.rect.y >> 16) << 24);
        reinterpret_cast<uint32_t*>(a.rect)[idx] = vis ? r4 : 0u;
    } else {
        a.rect[idx] = vis ? o.rect : make_uint2(0u, 0u);
    }
    a.depth_key[idx] = vis ? o.key : 0xFFFFFFFFu;
    {
        // The 64-byte record.  A lane writing its own record is four store instructions of 16 bytes every 64 bytes: each
        // touches all 64 lines of the wave's 4 KB a quarter each.  A full wave hands its records over through LDS instead
        // (rows of 80 bytes: conflict-free 16-byte writes) and lane l stores 16 bytes of the records (l >> 2) + 16 k,
        // k = 0..3: each store instruction covers one contiguous KB minus the culled Gaussians' holes -- the shape the
        // gradient-line clear below already has.
        __shared__ float4 s_rec[4][WAVE * 5];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wave_first = idx - lane;
        if (wave_first + WAVE <= a.P) {                       // wave-uniform: every lane is here
            const unsigned long long vm = __builtin_amdgcn_ballot_w64(vis);
            float4* mine = s_rec[wv] + lane * 5;
            mine[0] = o.r0; mine[1] = o.r1; mine[2] = o.r2; mine[3] = o.r3;
            float4* base = reinterpret_cast<float4*>(a.rec + (size_t)wave_first * REC_FLOATS);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int src = (lane >> 2) + 16 * k;
                const float4 v = s_rec[wv][src * 5 + (lane & 3)];
                if ((vm >> src) & 1ull) base[k * WAVE + lane] = v;
            }
        } else if (vis) {
            float4* rec = reinterpret_cast<float4*>(a.rec + (size_t)idx * REC_FLOATS);
            rec[0] = o.r0; rec[1] = o.r1; rec[2] = o.r2; rec[3] = o.r3;
        }
    }
    if (a.grad_acc) {
        // The blend backward adds into the 64-byte gradient line of a visible Gaussian and the per-Gaussian backward reads
        // it: those lines (only) are cleared here.  A full wave clears its 64 lines together -- lane l writes 16 bytes of
        // the lines (l >> 2) + 16 k, k = 0..3, so each store instruction covers one contiguous KB minus the invisible
        // Gaussians' holes (a lane writing its own line issued four 16-byte stores 64 bytes apart: +40 us at 2 M).
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const int lane = threadIdx.x & 63, wave_first = idx - lane;
        if (wave_first + WAVE <= a.P) {                       // wave-uniform: every lane is here
            const unsigned long long vm = __builtin_amdgcn_ballot_w64(vis);
            float4* base = reinterpret_cast<float4*>(a.grad_acc + (size_t)wave_first * GRAD_FLOATS);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((vm >> ((lane >> 2) + 16 * k)) & 1ull) base[k * WAVE + lane] = z4;
        } else if (vis) {
            float4* ga = reinterpret_cast<float4*>(a.grad_acc + (size_t)idx * GRAD_FLOATS);
            ga[0] = z4; ga[1] = z4; ga[2] = z4; ga[3] = z4;
        }
    }
}

int launch_preprocess_forward(const mgs_camera& cam, int P, const float* means3D, const float* shs,
                              const float* colors_precomp, const float* opacities, const float* scales,
                              const float* rotations, const float* cov3D_precomp, const GeometryState& g,
                              int32_t* radii, float* prepare_grad_acc, hipStream_t s) {
    PreArgs a;
    a.means3D = means3D; a.shs = shs; a.colors = colors_precomp; a.opacities = opacities;
    a.scales = scales; a.rotations = rotations; a.cov3D = cov3D_precomp;
    a.V = cam.viewmatrix; a.PM = cam.projmatrix; a.campos = cam.campos;
    a.rec = g.rec; a.depth_key = g.depth_key;
    a.grad_acc = prepare_grad_acc;
    a.rot_aligned16 = rotations && ((size_t)rotations % 16 == 0);
    a.clamped = g.clamped; a.rect = g.rect; a.radii = radii;
    a.tanfovx = cam.tanfovx; a.tanfovy = cam.tanfovy;
    a.focal_x = (float)cam.image_width / (2.0f * cam.tanfovx);
    a.focal_y = (float)cam.image_height / (2.0f * cam.tanfovy);
    a.mod = cam.scale_modifier;
    a.iso = cam.scale_dim == 1;
    a.P = P; a.W = cam.image_width; a.H = cam.image_height;
    a.gx = tiles_x(a.W); a.gy = tiles_y(a.H); a.deg = cam.sh_degree; a.M = cam.sh_coeffs;
    if (P == 0) return 0;
    a.rect_packed = depth_sort_payload(P, cam.image_width, cam.image_height) ? 1 : 0;
    radix_depth_zero_region(g.sort_temp, (uint64_t)P, &a.zero_ptr, &a.zero_words);
    // + the two small tables carved right in front of it (scan status words, tile-sort digit counts)
    a.zero_words += (size_t)((char*)a.zero_ptr - (char*)g.scan_status) / 4;
    a.zero_ptr = (uint32_t*)g.scan_status;
    const dim3 grid((P + 255) / 256), block(256);
    if (cov3D_precomp && colors_precomp) hipLaunchKernelGGL((preprocess_forward_kernel<true, true>), grid, block, 0, s, a);
    else if (cov3D_precomp) hipLaunchKernelGGL((preprocess_forward_kernel<true, false>), grid, block, 0, s, a);
    else if (colors_precomp) hipLaunchKernelGGL((preprocess_forward_kernel<false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((preprocess_forward_kernel<false, false>), grid, block, 0, s, a);
    MGS_HIP(hipGetLastError());
    return 0;
}

// =================================================================================================
// backward
// =================================================================================================
struct BwdArgs {
    GeomBackwardArgs g;
    const float *V, *PM, *PR, *campos;
    const float* rec;
    const uint8_t* clamped;
    float tanfovx, tanfovy, focal_x, focal_y, mod;
    int P, W, H, deg, M;
    int iso;                  // scales / dL_dscales hold ONE float per Gaussian
    int rot_aligned16, drot_aligned16;     // rotations / dL_drotations may be accessed 16 bytes at a time
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// SH = false: colours are precomputed (MonoGS's path) -- the spherical-harmonics backward is compiled out,
// which takes the kernel from 168 to far fewer VGPRs (occupancy) on the path that matters.
template <bool SH, bool COV>
__global__ void __launch_bounds__(256) geom_backward_kernel(BwdArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;        // (launched with 256 threads; blockDim would be a packet fetch)
    float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool in = idx < a.P;
    const size_t ci = in ? (size_t)idx : 0;                // clamped index: the lanes past P load element 0 and ignore it
    // ---- every input is requested up front -- camera block (scalar loads), radius, mean, covariance or scale +
    //      rotation, opacity and the Gaussian's 64-byte line of blend sums (three 16-byte loads) -- so that the kernel
    //      waits for memory ONCE.  Loads placed behind `radii > 0` and behind each other's first use used to cost five
    //      dependent round trips (78 % of the wave-cycles were s_waitcnt, profiles/r02); the line of a culled Gaussian
    //      that this reads for nothing is cheaper than a round trip.
    Cam c;
    load_cam(c, a.V, a.PM, a.PR, a.campos);
    c.tanfovx = a.tanfovx; c.tanfovy = a.tanfovy; c.focal_x = a.focal_x; c.focal_y = a.focal_y;
    const int rad = a.g.radii[ci];
    const float x = a.g.means3D[3 * ci], y = a.g.means3D[3 * ci + 1], z = a.g.means3D[3 * ci + 2];
    const float opq = a.g.opacities[ci];
    // the 64-byte line of blend sums: a full wave fetches its 64 lines with four coalesced 1-KB loads (lane l takes 16 bytes
    // of the lines (l >> 2) + 16 k) and deals them out through LDS (rows of 80 bytes), instead of three loads per lane
    // that each touch all 64 lines a quarter each
    __shared__ float4 s_ga[4][WAVE * 5];
    const int lane_ = threadIdx.x & 63, wv_ = threadIdx.x >> 6, wave_first = idx - lane_;
    const bool full_wave = wave_first + WAVE <= a.P;          // wave-uniform
    // (full wave: t_k = 16 bytes of line (lane >> 2) + 16 k; else t_0..t_2 = this lane's own line)
    const float4* gsrc = full_wave ? reinterpret_cast<const float4*>(a.g.grad_acc + (size_t)wave_first * GRAD_FLOATS) + lane_
                                   : reinterpret_cast<const float4*>(a.g.grad_acc + ci * GRAD_FLOATS);
    const int gstep = full_wave ? WAVE : 1;
    const float4 t0 = gsrc[0], t1 = gsrc[gstep], t2 = gsrc[2 * gstep], t3 = gsrc[3 * gstep];
    float cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, Mm[9];
    float s[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f};
    if (COV) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cov[k] = a.g.cov3D_precomp[6 * ci + k];
        asm volatile("" ::"v"(cov[0]), "v"(cov[1]), "v"(cov[2]), "v"(cov[3]), "v"(cov[4]), "v"(cov[5]));
    } else {
        const float* sp = a.g.scales + (a.iso ? ci : 3 * ci);       // isotropic: the same address three times, no branch
        const int o1 = a.iso ? 0 : 1, o2 = a.iso ? 0 : 2;
        s[0] = sp[0]; s[1] = sp[o1]; s[2] = sp[o2];
        load4(a.g.rotations + 4 * ci, a.rot_aligned16 != 0, q);
        asm volatile("" ::"v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]));
    }
    asm volatile("" ::"v"(rad), "v"(x), "v"(y), "v"(z), "v"(opq));
    float4 ga0 = t0, ga1 = t1, ga2 = t2;
    if (full_wave) {
        float4* row = s_ga[wv_] + (lane_ >> 2) * 5 + (lane_ & 3);
        row[0] = t0; row[16 * 5] = t1; row[32 * 5] = t2; row[48 * 5] = t3;
        ga0 = s_ga[wv_][lane_ * 5 + 0]; ga1 = s_ga[wv_][lane_ * 5 + 1]; ga2 = s_ga[wv_][lane_ * 5 + 2];
    }
    pin_cam(c, true);
    const float ga[GRAD_FLOATS] = {ga0.x, ga0.y, ga0.z, ga0.w, ga1.x, ga1.y, ga1.z, ga1.w, ga2.x, ga2.y, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool live = in && rad > 0;
    // The [P,3] outputs (means2D, means3D, colours, anisotropic scales) are collected here and stored at the end through a
    // per-wave LDS transpose: a lane writing its own 12 bytes makes every store instruction touch 12 cache lines a
    // third each; transposed, each store is 256 contiguous bytes.  Zero for culled Gaussians.
    float o_m2[2] = {0.f, 0.f}, o_m3[3] = {0.f, 0.f, 0.f}, o_col[3] = {0.f, 0.f, 0.f}, o_sc[3] = {0.f, 0.f, 0.f};
    float o_rot[4] = {0.f, 0.f, 0.f, 0.f}, o_cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, o_opac = 0.f, o_siso = 0.f;
    if (in && !live && a.g.dL_dsh) for (int k = 0; k < a.M * 3; ++k) a.g.dL_dsh[(size_t)idx * a.M * 3 + k] = 0.f;
    if (live) {
        float pv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) pv[k] = dot3p(c.V[k], c.V[4 + k], c.V[8 + k], c.V[12 + k], x, y, z);
        if (!COV) cov3d_from_scale_rot(s, q, a.mod, cov, Mm);
        Proj p;
        project_cov(c, pv, cov, p);

        // ---- conic -> 2-D covariance.  Q = C^-1;  dL/dC = -Q G Q with G the symmetric matrix
        //      [[gA, gB/2],[gB/2, gC]] (gB is the derivative w.r.t. the scalar b of power = -a dx^2/2 - c dy^2/2 - b dx dy)
        // the conic, exactly as the forward computed it (same operations on the same inputs)
        const float det_q = p.cxx * p.cyy - p.cxy * p.cxy;
        const float det_inv_q = 1.f / det_q;
        const float qa = p.cyy * det_inv_q, qb = -p.cxy * det_inv_q, qc = p.cxx * det_inv_q;
        // raw moments -> gradients w.r.t. the conic entries and the pixel-space mean (see common.h)
        const float gA = -0.5f * opq * ga[G_SXX], gBh = 0.5f * (-opq * ga[G_SXY]), gC = -0.5f * opq * ga[G_SYY];
        // N = G Q
        const float n00 = gA * qa + gBh * qb, n01 = gA * qb + gBh * qc;
        const float n10 = gBh * qa + gC * qb, n11 = gBh * qb + gC * qc;
        // Mc = -Q N  (symmetric)
        const float m00 = -(qa * n00 + qb * n10);
        const float m01 = -(qa * n01 + qb * n11);
        const float m11 = -(qb * n01 + qc * n11);

        // ---- C = T S T^T (+0.3 I):  dL/dS = T^T Mc T,  dL/dT = 2 Mc T S
        float A0[3], A1[3];                                   // A = Mc T  (2x3)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            A0[j] = m00 * p.T0[j] + m01 * p.T1[j];
            A1[j] = m01 * p.T0[j] + m11 * p.T1[j];
        }
        float Gs[3][3];                                       // full symmetric gradient w.r.t. Sigma
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Gs[i][j] = p.T0[i] * A0[j] + p.T1[i] * A1[j];
        const float S[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
        float dT0[3], dT1[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            dT0[j] = 2.f * ((A0[0] * S[0][j] + A0[1] * S[1][j]) + A0[2] * S[2][j]);
            dT1[j] = 2.f * ((A1[0] * S[0][j] + A1[1] * S[1][j]) + A1[2] * S[2][j]);
        }
        // ---- T = J Rv:  dL/dJ = dT Rv^T,  dL/dRv = J^T dT
        const float dJ00 = (dT0[0] * c.V[0] + dT0[1] * c.V[4]) + dT0[2] * c.V[8];     // row 0 of Rv
        const float dJ02 = (dT0[0] * c.V[2] + dT0[1] * c.V[6]) + dT0[2] * c.V[10];    // row 2 of Rv
        const float dJ11 = (dT1[0] * c.V[1] + dT1[1] * c.V[5]) + dT1[2] * c.V[9];     // row 1
        const float dJ12 = (dT1[0] * c.V[2] + dT1[1] * c.V[6]) + dT1[2] * c.V[10];
        float dRv[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            dRv[0][j] = p.J00 * dT0[j];
            dRv[1][j] = p.J11 * dT1[j];
            dRv[2][j] = p.J02 * dT0[j] + p.J12 * dT1[j];
        }
        const float tz = p.tz, itz = 1.f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
        float gpc[3];   // dL/d p_c, accumulated over the covariance, mean and depth paths
        gpc[0] = p.clamp_x ? 0.f : -c.focal_x * itz2 * dJ02;
        gpc[1] = p.clamp_y ? 0.f : -c.focal_y * itz2 * dJ12;
        gpc[2] = -c.focal_x * itz2 * dJ00 - c.focal_y * itz2 * dJ11 + 2.f * c.focal_x * p.tx * itz3 * dJ02 +
                 2.f * c.focal_y * p.ty * itz3 * dJ12;

        // ---- pixel mean:  pix = ((ndc + 1) S - 1)/2,  ndc = ph.xy / (ph.w + 1e-7),  ph = P_raw [p_c; 1]
        const float gpx = -opq * (qa * ga[G_SX] + qb * ga[G_SY]);
        const float gpy = -opq * (qc * ga[G_SY] + qb * ga[G_SX]);
        const float gnx = gpx * 0.5f * (float)a.W, gny = gpy * 0.5f * (float)a.H;
        o_m2[0] = gnx; o_m2[1] = gny;
        float ph[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ph[k] = dot3p(c.PM[k], c.PM[4 + k], c.PM[8 + k], c.PM[12 + k], x, y, z);
        const float p_w = 1.0f / (ph[3] + 0.0000001f);
        const float dph[4] = {gnx * p_w, gny * p_w, 0.f, -(gnx * ph[0] + gny * ph[1]) * p_w * p_w};
#pragma unroll
        for (int j = 0; j < 3; ++j)     // dL/dp_c[j] += sum_k dph[k] * Praw(k,j),  Praw(k,j) = PR[4*j+k]
            gpc[j] += ((dph[0] * c.PR[4 * j] + dph[1] * c.PR[4 * j + 1]) + dph[2] * c.PR[4 * j + 2]) +
                      dph[3] * c.PR[4 * j + 3];
        // ---- depth = p_c.z
        gpc[2] += ga[G_DDEPTH];

        // ---- colour
        float gmean_w[3] = {0.f, 0.f, 0.f};   // direct world-space contributions (SH view direction)
        if (!SH) {
            if (a.g.dL_dcolors) { o_col[0] = ga[G_DR]; o_col[1] = ga[G_DG]; o_col[2] = ga[G_DB]; }
        } else {
            float gc[3] = {ga[G_DR], ga[G_DG], ga[G_DB]};
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) if (a.clamped[4 * idx + ch]) gc[ch] = 0.f;
            float dv[3] = {x - c.campos[0], y - c.campos[1], z - c.campos[2]};
            const float len2 = (dv[0] * dv[0] + dv[1] * dv[1]) + dv[2] * dv[2];
            const float len = sqrtf(len2), ilen = 1.f / len;
            const float dx = dv[0] * ilen, dy = dv[1] * ilen, dz = dv[2] * ilen;
            const float* sh = a.g.shs + (size_t)idx * a.M * 3;
            float* dsh = a.g.dL_dsh ? a.g.dL_dsh + (size_t)idx * a.M * 3 : nullptr;
            float gdir[3] = {0.f, 0.f, 0.f};
            const int deg = a.deg;
            if (dsh) for (int k = (deg + 1) * (deg + 1) * 3; k < a.M * 3; ++k) dsh[k] = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float g = gc[ch];
                if (dsh) dsh[ch] = SH_C0 * g;
                if (deg > 0) {
                    if (dsh) { dsh[3 + ch] = -SH_C1 * dy * g; dsh[6 + ch] = SH_C1 * dz * g; dsh[9 + ch] = -SH_C1 * dx * g; }
                    float gx_ = -SH_C1 * sh[9 + ch], gy_ = -SH_C1 * sh[3 + ch], gz_ = SH_C1 * sh[6 + ch];
                    if (deg > 1) {
                        const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
                        if (dsh) {
                            dsh[12 + ch] = SH_C2[0] * xy * g; dsh[15 + ch] = SH_C2[1] * yz * g;
                            dsh[18 + ch] = SH_C2[2] * (2.f * zz - xx - yy) * g;
                            dsh[21 + ch] = SH_C2[3] * xz * g; dsh[24 + ch] = SH_C2[4] * (xx - yy) * g;
                        }
                        gx_ += SH_C2[0] * dy * sh[12 + ch] + SH_C2[2] * 2.f * -dx * sh[18 + ch] + SH_C2[3] * dz * sh[21 + ch] + SH_C2[4] * 2.f * dx * sh[24 + ch];
                        gy_ += SH_C2[0] * dx * sh[12 + ch] + SH_C2[1] * dz * sh[15 + ch] + SH_C2[2] * 2.f * -dy * sh[18 + ch] + SH_C2[4] * 2.f * -dy * sh[24 + ch];
                        gz_ += SH_C2[1] * dy * sh[15 + ch] + SH_C2[2] * 2.f * 2.f * dz * sh[18 + ch] + SH_C2[3] * dx * sh[21 + ch];
                        if (deg > 2) {
                            if (dsh) {
                                dsh[27 + ch] = SH_C3[0] * dy * (3.f * xx - yy) * g; dsh[30 + ch] = SH_C3[1] * xy * dz * g;
                                dsh[33 + ch] = SH_C3[2] * dy * (4.f * zz - xx - yy) * g;
                                dsh[36 + ch] = SH_C3[3] * dz * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                                dsh[39 + ch] = SH_C3[4] * dx * (4.f * zz - xx - yy) * g;
                                dsh[42 + ch] = SH_C3[5] * dz * (xx - yy) * g; dsh[45 + ch] = SH_C3[6] * dx * (xx - 3.f * yy) * g;
                            }
                            gx_ += SH_C3[0] * sh[27 + ch] * 3.f * 2.f * xy + SH_C3[1] * sh[30 + ch] * yz +
                                   SH_C3[2] * sh[33 + ch] * -2.f * xy + SH_C3[3] * sh[36 + ch] * -3.f * 2.f * xz +
                                   SH_C3[4] * sh[39 + ch] * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * sh[42 + ch] * 2.f * xz +
                                   SH_C3[6] * sh[45 + ch] * 3.f * (xx - yy);
                            gy_ += SH_C3[0] * sh[27 + ch] * 3.f * (xx - yy) + SH_C3[1] * sh[30 + ch] * xz +
                                   SH_C3[2] * sh[33 + ch] * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * sh[36 + ch] * -3.f * 2.f * yz +
                                   SH_C3[4] * sh[39 + ch] * -2.f * xy + SH_C3[5] * sh[42 + ch] * -2.f * yz +
                                   SH_C3[6] * sh[45 + ch] * -3.f * 2.f * xy;
                            gz_ += SH_C3[1] * sh[30 + ch] * xy + SH_C3[2] * sh[33 + ch] * 4.f * 2.f * yz +
                                   SH_C3[3] * sh[36 + ch] * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * sh[39 + ch] * 4.f * 2.f * xz +
                                   SH_C3[5] * sh[42 + ch] * (xx - yy);
                        }
                    }
                    gdir[0] += gx_ * g; gdir[1] += gy_ * g; gdir[2] += gz_ * g;
                }
            }
            // through the normalisation d = v/|v|:  dL/dv = (g - d (d.g)) / |v|
            const float dg = (dx * gdir[0] + dy * gdir[1]) + dz * gdir[2];
            gmean_w[0] = (gdir[0] - dx * dg) * ilen;
            gmean_w[1] = (gdir[1] - dy * dg) * ilen;
            gmean_w[2] = (gdir[2] - dz * dg) * ilen;
            // campos = -R^T (t + rho) to first order  =>  dL/drho += R gmean_w   (dL/dcampos = -gmean_w)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                tau[i] += (c.V[i] * gmean_w[0] + c.V[4 + i] * gmean_w[1]) + c.V[8 + i] * gmean_w[2];
        }
        o_opac = ga[G_SH];

        // ---- world-space mean:  p_c = Rv p + t
        if (a.g.dL_dmeans3D) {
#pragma unroll
            for (int j = 0; j < 3; ++j)   // (Rv^T gpc)[j] = sum_i Rv(i,j) gpc[i],  Rv(i,j) = V[4*j+i]
                o_m3[j] = ((c.V[4 * j] * gpc[0] + c.V[4 * j + 1] * gpc[1]) + c.V[4 * j + 2] * gpc[2]) + gmean_w[j];
        }
        // ---- pose:  d p_c / d rho = I,  d p_c / d theta = -[p_c]x,  d Rv(:,j) / d theta = -[Rv(:,j)]x
        tau[0] += gpc[0]; tau[1] += gpc[1]; tau[2] += gpc[2];
        tau[3] += pv[1] * gpc[2] - pv[2] * gpc[1];
        tau[4] += pv[2] * gpc[0] - pv[0] * gpc[2];
        tau[5] += pv[0] * gpc[1] - pv[1] * gpc[0];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float r0 = c.V[4 * j], r1_ = c.V[4 * j + 1], r2 = c.V[4 * j + 2];   // column j of Rv
            tau[3] += r1_ * dRv[2][j] - r2 * dRv[1][j];
            tau[4] += r2 * dRv[0][j] - r0 * dRv[2][j];
            tau[5] += r0 * dRv[1][j] - r1_ * dRv[0][j];
        }

        // ---- Sigma = M M^T, M = R diag(mod*s)
        if (COV) {
            o_cov[0] = Gs[0][0]; o_cov[1] = 2.f * Gs[0][1]; o_cov[2] = 2.f * Gs[0][2];
            o_cov[3] = Gs[1][1]; o_cov[4] = 2.f * Gs[1][2]; o_cov[5] = Gs[2][2];
        } else {
            float dM[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    dM[3 * i + j] = 2.f * ((Gs[i][0] * Mm[j] + Gs[i][1] * Mm[3 + j]) + Gs[i][2] * Mm[6 + j]);
            const float r = q[0], qx = q[1], qy = q[2], qz = q[3];
            const float Rm[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - r * qz), 2.f * (qx * qz + r * qy),
                                 2.f * (qx * qy + r * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - r * qx),
                                 2.f * (qx * qz - r * qy), 2.f * (qy * qz + r * qx), 1.f - 2.f * (qx * qx + qy * qy)};
            const float sm[3] = {s[0] * a.mod, s[1] * a.mod, s[2] * a.mod};
            if (a.g.dL_dscales) {
                float ds[3];
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    ds[j] = a.mod * ((dM[j] * Rm[j] + dM[3 + j] * Rm[3 + j]) + dM[6 + j] * Rm[6 + j]);
                if (a.iso) o_siso = (ds[0] + ds[1]) + ds[2];                    // backward of the isotropic expansion
                else { o_sc[0] = ds[0]; o_sc[1] = ds[1]; o_sc[2] = ds[2]; }
            }
            if (a.g.dL_drotations) {
                float dR[9];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) dR[3 * i + j] = dM[3 * i + j] * sm[j];
                // derivative of the (unnormalised) quaternion -> matrix map above
                const float dr = 2.f * (qz * (dR[3] - dR[1]) + qy * (dR[2] - dR[6]) + qx * (dR[7] - dR[5]));
                const float dqx = 2.f * (qy * (dR[1] + dR[3]) + qz * (dR[2] + dR[6]) + r * (dR[7] - dR[5])) -
                                  4.f * qx * (dR[4] + dR[8]);
                const float dqy = 2.f * (qx * (dR[1] + dR[3]) + r * (dR[2] - dR[6]) + qz * (dR[5] + dR[7])) -
                                  4.f * qy * (dR[0] + dR[8]);
                const float dqz = 2.f * (r * (dR[3] - dR[1]) + qx * (dR[2] + dR[6]) + qy * (dR[5] + dR[7])) -
                                  4.f * qz * (dR[0] + dR[4]);
                o_rot[0] = dr; o_rot[1] = dqx; o_rot[2] = dqy; o_rot[3] = dqz;
            }
        }
    }
    if (in) {   // ---- per-Gaussian outputs whose rows are 4, 16 or 24 bytes: one store each (zeros for culled Gaussians)
        if (a.g.dL_dopacity) a.g.dL_dopacity[idx] = o_opac;
        if (a.g.dL_dscales && a.iso) a.g.dL_dscales[idx] = o_siso;
        if (a.g.dL_drotations) {
            float* o = a.g.dL_drotations + 4 * (size_t)idx;
            if (a.drot_aligned16) *reinterpret_cast<float4*>(o) = make_float4(o_rot[0], o_rot[1], o_rot[2], o_rot[3]);
            else { o[0] = o_rot[0]; o[1] = o_rot[1]; o[2] = o_rot[2]; o[3] = o_rot[3]; }
        }
        if (COV && a.g.dL_dcov3D) {
            float* o = a.g.dL_dcov3D + 6 * (size_t)idx;
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = o_cov[k];
        }
    }
    if (in) {   // ---- the three-float outputs: 12 contiguous bytes per lane, 768 per wave = ONE global_store_dwordx3 each
        struct F3 { float x, y, z; };
        auto store3 = [&](float* __restrict__ out, float v0, float v1, float v2) {
            if (out) *reinterpret_cast<F3*>(out + 3 * (size_t)idx) = F3{v0, v1, v2};
        };
        store3(a.g.dL_dmeans2D, o_m2[0], o_m2[1], 0.f);
        store3(a.g.dL_dmeans3D, o_m3[0], o_m3[1], o_m3[2]);
        store3(a.g.dL_dcolors, o_col[0], o_col[1], o_col[2]);
        if (!a.iso) store3(a.g.dL_dscales, o_sc[0], o_sc[1], o_sc[2]);
    }
    if (a.g.dL_dtau) {
        __shared__ float part[4][6];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float v = wave_sum(tau[k]);
            if (lane == 0) part[wv][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            const float v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
            // Thousands of workgroups adding into the SAME 24 bytes serialise at the memory-side atomic unit (at C5 the
            // 7 813 x 6 same-address adds were 45 % of this kernel: 0.116 vs 0.065 ms without them).  Large launches
            // spread them over TAU_SLOTS lines (one 64-byte line per slot) and tau_finalize_kernel adds the slots up.
            if (v != 0.f) {
                if (a.g.tau_part) atomicAdd(a.g.tau_part + (size_t)(blockIdx.x % TAU_SLOTS) * 16 + threadIdx.x, v);
                else atomicAdd(a.g.dL_dtau + threadIdx.x, v);
            }
        }
    }
}

// dL_dtau[k] = sum over the slots (one block; fixed order, so the only non-determinism left is inside a slot)
__global__ void __launch_bounds__(256) tau_finalize_kernel(const float* __restrict__ tau_part, float* __restrict__ dL_dtau) {
    __shared__ float part[4][6];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    float v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = t < TAU_SLOTS ? tau_part[(size_t)t * 16 + k] : 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float r = wave_sum(v[k]);
        if (lane == 0) part[wv][k] = r;
    }
    __syncthreads();
    if (t < 6) dL_dtau[t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
}

int launch_geom_backward(const mgs_camera& cam, int P, const GeometryState& g, const GeomBackwardArgs& ga,
                         hipStream_t s) {
    BwdArgs a;
    a.g = ga;
    a.V = cam.viewmatrix; a.PM = cam.projmatrix; a.PR = cam.projmatrix_raw; a.campos = cam.campos;
    a.rec = g.rec; a.clamped = g.clamped;
    a.tanfovx = cam.tanfovx; a.tanfovy = cam.tanfovy;
    a.focal_x = (float)cam.image_width / (2.0f * cam.tanfovx);
    a.focal_y = (float)cam.image_height / (2.0f * cam.tanfovy);
    a.mod = cam.scale_modifier;
    a.iso = cam.scale_dim == 1;
    a.P = P; a.W = cam.image_width; a.H = cam.image_height; a.deg = cam.sh_degree; a.M = cam.sh_coeffs;
    if (P == 0) return 0;
    a.rot_aligned16 = ga.rotations && ((size_t)ga.rotations % 16 == 0);
    a.drot_aligned16 = ga.dL_drotations && ((size_t)ga.dL_drotations % 16 == 0);
    const dim3 grid((P + 255) / 256), block(256);
    const bool cov = ga.cov3D_precomp != nullptr;
    if (ga.colors_precomp && cov) hipLaunchKernelGGL((geom_backward_kernel<false, true>), grid, block, 0, s, a);
    else if (ga.colors_precomp) hipLaunchKernelGGL((geom_backward_kernel<false, false>), grid, block, 0, s, a);
    else if (cov) hipLaunchKernelGGL((geom_backward_kernel<true, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((geom_backward_kernel<true, false>), grid, block, 0, s, a);
    if (ga.tau_part && ga.dL_dtau) hipLaunchKernelGGL(tau_finalize_kernel, dim3(1), dim3(256), 0, s, ga.tau_part, ga.dL_dtau);
    MGS_HIP(hipGetLastError());
    return 0;
}

// =================================================================================================
__global__ void mark_visible_kernel(int P, const float* means3D, const float* V, uint8_t* visible) {
    const int idx = blockIdx.x * 256 + threadIdx.x;        // (launched with 256 threads; blockDim would be a packet fetch)
    if (idx >= P) return;
    const float x = means3D[3 * idx], y = means3D[3 * idx + 1], z = means3D[3 * idx + 2];
    const float vz = dot3p(V[2], V[6], V[10], V[14], x, y, z);
    visible[idx] = vz > 0.2f;
}

int launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* visible, hipStream_t s) {
    if (P == 0) return 0;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, visible);
    MGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace mgs
