// distCUDA2: mean squared distance to the 3 nearest other points, exact.
//
// Boundary replaced: simple_knn._C.distCUDA2 (un-vendored; call site
// /root/reference/gaussian_splatting/scene/gaussian_model.py:294-302).  Self is excluded by index, so
// coincident duplicates count with distance 0; with fewer than 4 points the result carries FLT_MAX terms
// (SURVEY.md Appendix A).
//
// Two gfx950 paths, both exact and bit-identical to each other (same pair formula, the three smallest
// distances summed in ascending order):
//   * P < 6144 (MonoGS calls it with 5k-25k back-projected points per keyframe, SURVEY.md section 8a
//     row a12): an LDS-tiled all-pairs sweep, one query per lane, candidate tiles of 1024 points staged in LDS
//     and read back as wave-wide broadcasts.  No index to build; < 1 ms at these sizes.
//   * larger clouds (whole-map queries, the reference's own TODO at gaussian_model.py:293): points are
//     sorted along a 30-bit Morton curve with the radix sort of radix_sort.hip, cut into boxes of 64
//     consecutive points with their bounding boxes, and ONE WAVE answers the 64 queries of a box: lane l
//     tests candidate box l of each group of 64 against the wave's own bounding box (wave-uniform prune
//     against the largest third-best distance of the wave), a per-lane point-to-box test decides whether any
//     lane still needs the box, and the survivors are staged through LDS and swept by all 64 lanes as
//     broadcasts.  Work drops from P^2 pairs to about P x (a few thousand) pairs.
#include "common.h"

#include <stdlib.h>

namespace mgs {

constexpr int KNN_THREADS = 256;
constexpr int KNN_TILE = 1024;
constexpr int KNN_BOX = 64;                 // points per Morton box = one wave of queries
constexpr int KNN_GRID_MIN = 6144;          // below this the all-pairs sweep wins (no sort, no index).  Measured, round 2:
                                            // 4 k points 0.147 vs 0.207 ms, 8 k 0.291 vs 0.246, 16 k 0.58 vs 0.33, 32 k 1.17 vs 0.36 ms
constexpr float KNN_FLT_MAX = 3.402823466e+38f;

// the ONE pair formula both paths use (explicit fma chain so both compile to the same three instructions)
__device__ __forceinline__ float pair_d2(float cx, float cy, float cz, float qx, float qy, float qz) {
    const float dx = cx - qx, dy = cy - qy, dz = cz - qz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}
// insert d into the sorted triple (b0 <= b1 <= b2)
__device__ __forceinline__ void insert3(float d, float& b0, float& b1, float& b2) {
    const float m2 = fminf(b2, d);
    const float n1 = fminf(b1, m2), n2 = fmaxf(b1, m2);
    const float n0 = fminf(b0, n1), n1b = fmaxf(b0, n1);
    b0 = n0; b1 = n1b; b2 = n2;
}

__global__ void __launch_bounds__(KNN_THREADS) knn_kernel(int P, const float* __restrict__ pts,
                                                          float* __restrict__ out) {
    __shared__ float4 tile[KNN_TILE];
    const int idx = blockIdx.x * KNN_THREADS + threadIdx.x;
    const bool live = idx < P;
    const float qx = live ? pts[3 * idx] : 0.f, qy = live ? pts[3 * idx + 1] : 0.f, qz = live ? pts[3 * idx + 2] : 0.f;
    float b0 = KNN_FLT_MAX, b1 = b0, b2 = b0;   // FLT_MAX, as upstream initialises its best list
    for (int start = 0; start < P; start += KNN_TILE) {
        const int n = min(KNN_TILE, P - start);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += KNN_THREADS) {
            const int s = start + i;
            tile[i] = make_float4(pts[3 * s], pts[3 * s + 1], pts[3 * s + 2], 0.f);
        }
        __syncthreads();
        for (int i = 0; i < n; ++i) {
            const float4 c = tile[i];
            float d = pair_d2(c.x, c.y, c.z, qx, qy, qz);
            d = (start + i == idx) ? KNN_FLT_MAX : d;
            insert3(d, b0, b1, b2);
        }
    }
    if (live) out[idx] = (b0 + b1 + b2) / 3.f;
}

// ------------------------------------------------------------------------------------------------
// Morton-box path
// ------------------------------------------------------------------------------------------------
struct KnnScratch {
    int32_t* bbox;            // [8] ordered-int min xyz (0..2), max xyz (4..6)
    uint32_t *code_a, *code_b, *idx_a, *idx_b;
    float4* sorted;           // [P] x, y, z, bits(original index), Morton order
    float4* box_lo;           // [nb]
    float4* box_hi;           // [nb]
    void* sort_temp;
    size_t bytes;
};
static KnnScratch knn_carve(void* base, int P) {
    KnnScratch k;
    const size_t n = (size_t)P, nb = (n + KNN_BOX - 1) / KNN_BOX;
    char* p = (char*)align_up((size_t)base, 256);
    char* p0 = p;
    auto take = [&](size_t bytes) { char* r = p; p += align_up(bytes, 256); return r; };
    k.bbox = (int32_t*)take(8 * sizeof(int32_t));
    k.code_a = (uint32_t*)take(n * 4); k.code_b = (uint32_t*)take(n * 4);
    k.idx_a = (uint32_t*)take(n * 4); k.idx_b = (uint32_t*)take(n * 4);
    k.sorted = (float4*)take(n * 16);
    k.box_lo = (float4*)take(nb * 16); k.box_hi = (float4*)take(nb * 16);
    k.sort_temp = take(radix_temp_bytes(n, 32));
    k.bytes = (size_t)(p - p0) + 256;
    return k;
}

// float <-> int whose signed order is the float order (for atomicMin / atomicMax)
__device__ __forceinline__ int32_t f2ord(float f) { const int32_t i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float ord2f(int32_t i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void knn_bbox_init_kernel(int32_t* bbox) {
    if (threadIdx.x < 8) bbox[threadIdx.x] = threadIdx.x < 4 ? 0x7FFFFFFF : (int32_t)0x80000000;
}

__global__ void __launch_bounds__(KNN_THREADS) knn_bbox_kernel(int P, const float* __restrict__ pts, int32_t* __restrict__ bbox) {
    float lo[3] = {KNN_FLT_MAX, KNN_FLT_MAX, KNN_FLT_MAX}, hi[3] = {-KNN_FLT_MAX, -KNN_FLT_MAX, -KNN_FLT_MAX};
    for (int i = blockIdx.x * KNN_THREADS + threadIdx.x; i < P; i += gridDim.x * KNN_THREADS) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * i + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(bbox + a, f2ord(lo[a]));
            atomicMax(bbox + 4 + a, f2ord(hi[a]));
        }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {       // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void __launch_bounds__(KNN_THREADS) knn_morton_kernel(int P, const float* __restrict__ pts,
                                                                 const int32_t* __restrict__ bbox,
                                                                 uint32_t* __restrict__ code, uint32_t* __restrict__ idx) {
    const int i = blockIdx.x * KNN_THREADS + threadIdx.x;
    if (i >= P) return;
    uint32_t c = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = ord2f(bbox[a]), hi = ord2f(bbox[4 + a]);
        const float ext = hi - lo;
        const float t = ext > 0.f ? (pts[3 * i + a] - lo) / ext * 1023.f : 0.f;
        const uint32_t q = (uint32_t)fminf(fmaxf(t, 0.f), 1023.f);     // NaN -> 0
        c |= spread10(q) << (2 - a);
    }
    code[i] = c;
    idx[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(KNN_THREADS) knn_gather_kernel(int P, const float* __restrict__ pts,
                                                                 const uint32_t* __restrict__ idx_sorted,
                                                                 float4* __restrict__ sorted, float4* __restrict__ box_lo,
                                                                 float4* __restrict__ box_hi) {
    const int i = blockIdx.x * KNN_THREADS + threadIdx.x;          // position along the curve; a wave = one box
    const bool live = i < P;
    float v[3] = {0.f, 0.f, 0.f};
    if (live) {
        const uint32_t s = idx_sorted[i];
        v[0] = pts[3 * (size_t)s]; v[1] = pts[3 * (size_t)s + 1]; v[2] = pts[3 * (size_t)s + 2];
        sorted[i] = make_float4(v[0], v[1], v[2], __uint_as_float(s));
    }
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = live ? v[a] : KNN_FLT_MAX;
        hi[a] = live ? v[a] : -KNN_FLT_MAX;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
        }
    }
    const int first = i & ~63;
    if ((threadIdx.x & 63) == 0 && first < P) {
        box_lo[first / KNN_BOX] = make_float4(lo[0], lo[1], lo[2], 0.f);
        box_hi[first / KNN_BOX] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
}

__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float gap(float lo, float hi, float qlo, float qhi) {   // distance between [lo,hi] and [qlo,qhi]
    return fmaxf(0.f, fmaxf(lo - qhi, qlo - hi));
}

__global__ void __launch_bounds__(KNN_THREADS) knn_query_kernel(int P, int nb, const float4* __restrict__ sorted,
                                                                const float4* __restrict__ box_lo,
                                                                const float4* __restrict__ box_hi, float* __restrict__ out) {
    __shared__ float4 cand[KNN_THREADS / 64][KNN_BOX];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qb = blockIdx.x * (KNN_THREADS / 64) + wv;            // this wave's box
    if (qb >= nb) return;                                           // whole wave; no block-level barrier below
    const int qi = qb * KNN_BOX + lane;
    const bool live = qi < P;
    const float4 q = sorted[live ? qi : P - 1];
    const float4 qlo = box_lo[qb], qhi = box_hi[qb];
    float b0 = KNN_FLT_MAX, b1 = b0, b2 = b0;

    auto sweep = [&](int b) {                   // all 64 lanes against the (<= 64) points of box b
        const int ci = b * KNN_BOX + lane;
        __builtin_amdgcn_wave_barrier();
        cand[wv][lane] = ci < P ? sorted[ci] : make_float4(KNN_FLT_MAX, KNN_FLT_MAX, KNN_FLT_MAX, 0.f);
        __builtin_amdgcn_wave_barrier();
        const int n = min(KNN_BOX, P - b * KNN_BOX);
        const int self = b == qb ? lane : -1;
        for (int i = 0; i < n; ++i) {
            const float4 c = cand[wv][i];
            float d = pair_d2(c.x, c.y, c.z, q.x, q.y, q.z);
            d = (i == self) ? KNN_FLT_MAX : d;
            insert3(d, b0, b1, b2);
        }
    };
    // seed from the neighbourhood along the curve
    sweep(qb);
    if (qb > 0) sweep(qb - 1);
    if (qb + 1 < nb) sweep(qb + 1);
    float wmax = wave_max_f32(live ? b2 : 0.f);     // no lane of the wave can still use a box farther than this

    // A box can only matter if (a lower bound of) its distance is below a third-best; the 0.99999 absorbs the
    // rounding difference between the bound and the pair formula, and the strict < drops boxes that could at best tie.
    for (int base = 0; base < nb; base += 64) {
        const int bi = base + lane;
        bool want = bi < nb && (bi < qb - 1 || bi > qb + 1);
        if (want) {
            const float4 lo = box_lo[bi], hi = box_hi[bi];
            const float gx = gap(lo.x, hi.x, qlo.x, qhi.x), gy = gap(lo.y, hi.y, qlo.y, qhi.y), gz = gap(lo.z, hi.z, qlo.z, qhi.z);
            want = fmaf(gz, gz, fmaf(gy, gy, gx * gx)) * 0.99999f < wmax;
        }
        unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int b = base + j;                                  // wave-uniform
            const float4 lo = box_lo[b], hi = box_hi[b];
            const float gx = gap(lo.x, hi.x, q.x, q.x), gy = gap(lo.y, hi.y, q.y, q.y), gz = gap(lo.z, hi.z, q.z, q.z);
            const bool need = live && (fmaf(gz, gz, fmaf(gy, gy, gx * gx)) * 0.99999f < b2);
            if (__builtin_amdgcn_ballot_w64(need) == 0ull) continue;
            sweep(b);
            wmax = wave_max_f32(live ? b2 : 0.f);
        }
    }
    if (live) out[__float_as_uint(q.w)] = (b0 + b1 + b2) / 3.f;
}

int g_opt_knn_grid_min = -1;         // mgs_debug_set_option("knn_grid_min", n): tests force either path at any size
static int knn_grid_min() { return g_opt_knn_grid_min >= 0 ? g_opt_knn_grid_min : KNN_GRID_MIN; }

size_t knn_scratch_bytes(int P) {
    if (P < 4) return 256;
    return knn_carve(nullptr, P).bytes;              // sized for either path, whatever the environment says later
}

int launch_knn(int P, const float* points, float* out, void* scratch, hipStream_t s) {
    if (P == 0) return 0;
    if (P < 4 || P < knn_grid_min()) {
        hipLaunchKernelGGL(knn_kernel, dim3((P + KNN_THREADS - 1) / KNN_THREADS), dim3(KNN_THREADS), 0, s, P, points, out);
        MGS_HIP(hipGetLastError());
        return 0;
    }
    if (!scratch) { set_error("mgs_dist2_knn: scratch is NULL"); return 1; }
    const KnnScratch k = knn_carve(scratch, P);
    const int nb = (P + KNN_BOX - 1) / KNN_BOX;
    const int blocks = (P + KNN_THREADS - 1) / KNN_THREADS;
    hipLaunchKernelGGL(knn_bbox_init_kernel, dim3(1), dim3(64), 0, s, k.bbox);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(min(blocks, 1024)), dim3(KNN_THREADS), 0, s, P, points, k.bbox);
    hipLaunchKernelGGL(knn_morton_kernel, dim3(blocks), dim3(KNN_THREADS), 0, s, P, points, k.bbox, k.code_a, k.idx_a);
    if (int rc = radix_sort_pairs(k.code_a, k.idx_a, k.code_b, k.idx_b, (uint64_t)P, 32, k.sort_temp, s, nullptr)) return rc;
    const uint32_t* idx_sorted = radix_result_in_b(32) ? k.idx_b : k.idx_a;
    hipLaunchKernelGGL(knn_gather_kernel, dim3(blocks), dim3(KNN_THREADS), 0, s, P, points, idx_sorted, k.sorted, k.box_lo, k.box_hi);
    hipLaunchKernelGGL(knn_query_kernel, dim3((nb + 3) / 4), dim3(KNN_THREADS), 0, s, P, nb, k.sorted, k.box_lo, k.box_hi, out);
    MGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace mgs

// ------------------------------------------------------------------------------------------------
// Keyframe back-projection: the gather + exposure + unprojection + camera->world
// part of GaussianModel.create_viewpoint_pcd (/root/reference/gaussian_splatting/scene/gaussian_model.py:121-319)
// for the N selected pixels, one launch.  `sel` indexes pixels in the reference's flattening order
// (x outer, y inner: i = x * H + y, gaussian_model.py:180-187,221-233); the pixel centre is (x + 0.5, y + 0.5).
// ------------------------------------------------------------------------------------------------
namespace mgs {

__global__ void __launch_bounds__(256) backproject_kernel(int N, int W, int H, const int64_t* __restrict__ sel,
                                                          const float* __restrict__ rgb, const float* __restrict__ depth,
                                                          const int32_t* __restrict__ seg, const float* __restrict__ exp_a,
                                                          const float* __restrict__ exp_b, float fx, float fy, float cx,
                                                          float cy, const float* __restrict__ R, const float* __restrict__ T,
                                                          float* __restrict__ pts, float* __restrict__ feat,
                                                          int32_t* __restrict__ ids) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int64_t s = sel[i];
    const int x = (int)(s / H), y = (int)(s % H);
    const size_t pix = (size_t)y * W + x, HW = (size_t)H * W;
    const float z = depth[pix];
    const float xc = ((float)x + 0.5f - cx) / fx * z, yc = ((float)y + 0.5f - cy) / fy * z;
    const float dx = xc - T[0], dy = yc - T[1], dz = z - T[2];          // p_w = R^T (p_c - t)
    pts[3 * i] = R[0] * dx + R[3] * dy + R[6] * dz;
    pts[3 * i + 1] = R[1] * dx + R[4] * dy + R[7] * dz;
    pts[3 * i + 2] = R[2] * dx + R[5] * dy + R[8] * dz;
    float c0 = rgb[pix], c1 = rgb[HW + pix], c2 = rgb[2 * HW + pix];
    if (exp_a) {                                                        // exposure learned during tracking, clamped to [0, 1]
        const float ea = expf(exp_a[0]), eb = exp_b ? exp_b[0] : 0.f;
        c0 = fminf(fmaxf(ea * c0 + eb, 0.f), 1.f);
        c1 = fminf(fmaxf(ea * c1 + eb, 0.f), 1.f);
        c2 = fminf(fmaxf(ea * c2 + eb, 0.f), 1.f);
    }
    feat[3 * i] = c0; feat[3 * i + 1] = c1; feat[3 * i + 2] = c2;
    if (ids) ids[i] = seg ? seg[pix] : 0;
}

}  // namespace mgs

extern "C" int mgs_backproject(int32_t N, int32_t W, int32_t H, const int64_t* selected, const float* rgb,
                               const float* depth, const int32_t* segmentation, const float* exposure_a,
                               const float* exposure_b, float fx, float fy, float cx, float cy, const float* R,
                               const float* T, float* points, float* features, int32_t* ids, void* stream) {
    if (N < 0 || W <= 0 || H <= 0) { mgs::set_error("mgs_backproject: bad sizes"); return 1; }
    if (N == 0) return 0;
    if (!selected || !rgb || !depth || !R || !T || !points || !features) { mgs::set_error("mgs_backproject: NULL argument"); return 1; }
    hipLaunchKernelGGL(mgs::backproject_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, W, H, selected,
                       rgb, depth, segmentation, exposure_a, exposure_b, fx, fy, cx, cy, R, T, points, features, ids);
    MGS_HIP(hipGetLastError());
    return 0;
}
