// Internal definitions shared by the HIP translation units of libmonogs_raster.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/monogs_raster.h"

namespace mgs {

constexpr int TILE = MGS_TILE;          // 16x16 pixel tiles define the binning tables
constexpr int WAVE = 64;                // gfx950 wavefront
constexpr int SUB = 8;                  // one wave blends an 8x8 quadrant of a tile
constexpr int REC_FLOATS = 16;          // per-Gaussian blend record: one 64-byte line
constexpr int GRAD_FLOATS = 16;         // per-Gaussian gradient accumulator: one 64-byte line

// blend record slots (float index inside the 64-byte record written by preprocess):
//   float4 #0 {px, py, ex, ey}   what a lane needs for the quadrant cull test
//   float4 #1 {ka, kb, kc, opacity}      } fetched with scalar loads for a surviving instance;
//                                          (ka, kb, kc) = (-log2e/2 a, -log2e b, -log2e/2 c) of the conic, so that
//                                          alpha = opacity * exp2(ka dx^2 + kb dx dy + kc dy^2): 5 VALU ops + v_exp_f32
//   float4 #2 {r, g, b, depth}           }
//   float4 #3 {na, nb, nc, radius}   conic / (2 ln(255 o)) for the exact ellipse-vs-quadrant cull test; radius for duplicate
enum : int {
    R_X = 0, R_Y = 1, R_EX = 2, R_EY = 3,
    R_CA = 4, R_CB = 5, R_CC = 6, R_OPAC = 7,
    R_R = 8, R_G = 9, R_B = 10, R_DEPTH = 11,
    R_NA = 12, R_NB = 13, R_NC = 14, R_RADIUS = 15
};
// gradient accumulator slots (float atomics by the blend backward).  With h = G * dL/dalpha summed
// over the pixels of every tile the Gaussian touches, the blend backward stores the RAW moments
//   S_X = sum h dx, S_Y = sum h dy, S_XX = sum h dx^2, S_XY = sum h dx dy, S_YY = sum h dy^2, S_H = sum h
// (d = mean2D - pixel); the per-Gaussian constants (conic, opacity, -1/2) are applied once per
// Gaussian in geom_backward instead of once per (pixel, instance) pair -- the map is linear.
enum : int {
    G_SX = 0, G_SY = 1, G_SXX = 2, G_SXY = 3, G_SYY = 4, G_SH = 5,
    G_DR = 6, G_DG = 7, G_DB = 8, G_DDEPTH = 9
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Wave-uniform read-only inputs (camera matrices, background colour): read through the CONSTANT address space, so the
// backend always selects scalar loads (s_load_dwordx8/x16, all issued at once) for them.  Through a generic pointer
// it falls back to per-lane vector loads with a vmcnt(0) wait after each group as soon as the kernel also stores to
// global memory (its no-clobber analysis gives up): the per-Gaussian kernels spent seven dependent memory round trips
// on their 51 camera floats that way.  Valid because these buffers are written by an EARLIER kernel, never by the one
// reading them (kernel boundaries invalidate the scalar cache).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const float __attribute__((address_space(4))) * const_float_p;
#define MGS_CONST(p) ((mgs::const_float_p)(p))
#else
typedef const float* const_float_p;
#define MGS_CONST(p) (p)
#endif

// ---- 64-lane inclusive scan (u32 add) on the DPP data path: 4 row shifts + 2 row broadcasts, ~60 cycles, instead of
// six __shfl_up round trips through ds_bpermute (the LDS crossbar, >100 cycles each).
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
#define MGS_DPP_ADD(ctrl, row_mask) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, row_mask, 0xf, false)
    MGS_DPP_ADD(0x111, 0xf);      // row_shr:1
    MGS_DPP_ADD(0x112, 0xf);      // row_shr:2
    MGS_DPP_ADD(0x114, 0xf);      // row_shr:4
    MGS_DPP_ADD(0x118, 0xf);      // row_shr:8   -> inclusive within each row of 16
    MGS_DPP_ADD(0x142, 0xa);      // row_bcast:15 into rows 1, 3
    MGS_DPP_ADD(0x143, 0xc);      // row_bcast:31 into rows 2, 3
#undef MGS_DPP_ADD
    return v;
}

// ---- zero fill as a plain kernel ------------------------------------------------------------
// Every clear on the hot path is a kernel, never hipMemsetAsync / hipMemcpyAsync: inside a captured hipGraph
// those become memset / memcpy NODES, and on this ROCm a graph holding them faulted ("write access to a
// read-only page") once unrelated eager copies ran between two replays (tools/graph_probe.py).  Kernel nodes
// replay exactly as launched.  `ptr` must be 4-byte aligned and `bytes` a multiple of 4.
__global__ static void zero_fill_kernel(uint32_t* __restrict__ p, size_t n_words, int vec) {
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if (vec) {
        uint4* q = reinterpret_cast<uint4*>(p);
        const size_t nv = n_words / 4;
        for (size_t i = i0; i < nv; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
        for (size_t i = nv * 4 + i0; i < n_words; i += stride) p[i] = 0u;
    } else {
        for (size_t i = i0; i < n_words; i += stride) p[i] = 0u;
    }
}
// the same clear, done by the threads of a kernel that runs anyway (p 16-byte aligned): saves a launch
// (`threads` = total threads of the launch, passed by the host: blockDim / gridDim would be fetched from the dispatch
//  packet with a vector load + wait, a memory round trip at the very start of the kernel)
__device__ __forceinline__ void grid_zero(uint32_t* __restrict__ p, size_t n_words, size_t threads, int block) {
    const size_t i0 = (size_t)blockIdx.x * block + threadIdx.x, stride = threads;
    uint4* q = reinterpret_cast<uint4*>(p);
    const size_t nv = n_words / 4;
    for (size_t i = i0; i < nv; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = nv * 4 + i0; i < n_words; i += stride) p[i] = 0u;
}
// two regions in one launch (the second one small): gradient accumulator + the 6 pose-gradient floats
__global__ static void zero_fill2_kernel(uint32_t* __restrict__ p, size_t n_words, uint32_t* __restrict__ p2, int n2) {
    grid_zero(p, n_words, (size_t)gridDim.x * 256, 256);
    if (blockIdx.x == 0 && (int)threadIdx.x < n2) p2[threadIdx.x] = 0u;
}
static inline hipError_t zero_fill2(void* ptr, size_t bytes, void* ptr2, size_t bytes2, hipStream_t s) {
    const size_t n_words = bytes / 4;                       // ptr 16-byte aligned; bytes2 <= 1 KiB
    size_t blocks = (n_words / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(zero_fill2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t*)ptr, n_words,
                       (uint32_t*)ptr2, ptr2 ? (int)(bytes2 / 4) : 0);
    return hipGetLastError();
}
static inline hipError_t zero_fill(void* ptr, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    const size_t n_words = bytes / 4;
    const int vec = ((size_t)ptr % 16 == 0) ? 1 : 0;
    const size_t items = vec ? (n_words + 3) / 4 : n_words;
    size_t blocks = (items + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t*)ptr, n_words, vec);
    return hipGetLastError();
}

// ---- scratch carving (every sub-buffer 256-byte aligned) ---------------------------------
struct GeometryState {
    float* rec;               // [P][16]
    uint32_t* depth_key;      // [P] float32 bits of the view-space depth (0xFFFFFFFF when culled); sort input
    uint32_t* depth_alt;      // [P] ping-pong partner of depth_key
    uint32_t* iota;           // [P] ping-pong buffers of the depth sort's values (the first pass takes value = index and
    uint32_t* iota_alt;       // [P] reads neither)
    uint32_t* perm;           // [P] Gaussian index in (depth, index) order -- culled ones last: written by the sort's final pass
    uint32_t* point_offsets;  // [P] inclusive scan, in depth order, of the tiles each Gaussian touches (w * h of its rectangle)
    uint32_t* scan_blocks;    // [scan_nblocks(P) + 64]
    uint8_t* clamped;         // [P][4] SH colour clamp flags
    uint2* rect;              // [P] tile rectangle {x0 | y0 << 16, w | h << 16} (w = h = 0: culled), by Gaussian index; or
                              //     (depth_sort_payload()) uint32[2 P]: the first half x0 | y0 << 8 | w << 16 | h << 24,
                              //     which the depth sort carries along as a payload, the second half its ping-pong partner
    uint2* rect_sorted;       // [P] the same in depth order: written by the last pass of the depth sort, so that the
                              //     scan and duplicate read it coalesced instead of gathering through perm[]
    uint64_t* scan_status;    // [SCAN_SMALL_MAX_BLOCKS + 1] block totals of the single-launch scan + its ticket (zeroed by preprocess)
    uint32_t* tile_hist;      // [4][256] digit counts of the TILE sort's keys, counted by duplicate_kernel while it emits
                              //          them (zeroed by preprocess; lives here because the binning scratch only exists
                              //          once the instance count is known)
    void* sort_temp;          // depth sort scratch (directly behind tile_hist: preprocess clears all three in one sweep)
    size_t sort_temp_bytes;
    char* end;                // one past the last carved byte
    static size_t bytes(int P);
    static GeometryState carve(void* base, int P);
};
constexpr int SCAN_SMALL_MAX_BLOCKS = 64;     // up to this many scan blocks (P <= 131 072) the scan is ONE launch

struct ImageState {
    float* final_T;      // [H*W]
    uint32_t* n_contrib; // [H*W]
    uint2* ranges;       // [tiles]
    static size_t bytes(int W, int H);
    static ImageState carve(void* base, int W, int H);
};

struct BinningState {
    uint32_t* keys_a;          // tile id per instance, emitted in (depth, index) order; sort input
    uint32_t* vals_a;
    uint32_t* keys_b;          // ping-pong partners
    uint32_t* vals_b;
    uint32_t* keys_sorted;     // = keys_a or keys_b: tile ids grouped by tile (stable)
    uint32_t* vals_sorted;     // = vals_a or vals_b: "point_list", Gaussian index per instance in blend order
    void* sort_temp;
    size_t sort_temp_bytes;
    uint32_t* count;           // [2] capacity mode: live instance count, overflow flag
    static size_t bytes(uint64_t R, int W, int H);
    static BinningState carve(void* base, uint64_t R, int W, int H);
};

constexpr int SCAN_ITEMS = 2048;   // items per scan block
inline int scan_nblocks(int P) { return (P + SCAN_ITEMS - 1) / SCAN_ITEMS; }

inline int tiles_x(int W) { return (W + TILE - 1) / TILE; }
inline int tiles_y(int H) { return (H + TILE - 1) / TILE; }
inline int tile_bits(int W, int H) {     // bits of the tile id that can be set
    uint32_t n = (uint32_t)(tiles_x(W) * tiles_y(H));
    int b = 0;
    while ((1u << b) < n && b < 31) ++b;
    return b == 0 ? 1 : b;
}

// ---- error plumbing --------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define MGS_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            mgs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

// ---- stage launchers (one per translation unit) ------------------------------------------
struct StageTimer;  // api.hip

int launch_preprocess_forward(const mgs_camera& cam, int P, const float* means3D, const float* shs,
                              const float* colors_precomp, const float* opacities, const float* scales,
                              const float* rotations, const float* cov3D_precomp,
                              const GeometryState& g, int32_t* radii, float* prepare_grad_acc, hipStream_t s);
// backward scratch: [grad_acc P x 16][pose-gradient slots TAU_SLOTS x 16][dL/dtau of a prepared backward: 16]
constexpr int TAU_SLOTS = 256;              // one 64-byte line each
inline float* backward_grad_acc(void* scratch) { return (float*)align_up((size_t)scratch, 256); }
inline float* backward_tau_part(void* scratch, int P) { return backward_grad_acc(scratch) + (size_t)P * GRAD_FLOATS; }
inline float* backward_tau_out(void* scratch, int P) { return backward_tau_part(scratch, P) + (size_t)TAU_SLOTS * 16; }
int launch_scan(const GeometryState& g, int P, hipStream_t s, bool exclusive = false);
// small maps: depth sort + rectangle gather + scan in ONE single-workgroup launch (binning.hip); replaces launch_depth_sort + launch_scan
bool depth_chain_is_small(int P);
int launch_depth_chain_small(const GeometryState& g, int P, hipStream_t s);
// `r_cap`: capacity of the binning buffers; `count` (device): [0] live instance count min(R, r_cap), [1] overflow flag
int launch_duplicate(const mgs_camera& cam, int P, const GeometryState& g, const BinningState& b, uint64_t r_cap,
                     int32_t* n_touched, const ImageState& img, uint64_t sort_n, int sort_bits, uint32_t* count,
                     uint32_t* overflow, hipStream_t s);
size_t sort_temp_bytes(uint64_t n, int bits);
size_t radix_temp_bytes(uint64_t n, int bits);
size_t radix_depth_temp_bytes(uint64_t n);
bool radix_result_in_b(int bits);
constexpr int RADIX_ERROR_WORDS = 4;      // = RS_MAX_PASSES (radix_sort.hip)
const uint32_t* radix_error_flag(void* temp, uint64_t n, int bits);
const uint32_t* radix_depth_error_flag(void* temp, uint64_t n);
__device__ __forceinline__ uint32_t radix_failed(const uint32_t* __restrict__ e) { return (e[0] | e[1]) | (e[2] | e[3]); }
void radix_zero_region(void* temp, uint64_t n, int bits, uint32_t** ptr, size_t* words);   // what must be 0 before a sort
void radix_depth_zero_region(void* temp, uint64_t n, uint32_t** ptr, size_t* words);
// `ext_hist` ([4][256], one-sweep path only): digit counts of the keys already counted by the kernel that produced them
// -- the sort then launches no histogram kernel.  radix_wants_hist(n) tells the producer whether they will be used.
int radix_sort_pairs(uint32_t* ka, uint32_t* va, uint32_t* kb, uint32_t* vb, uint64_t n, int bits, void* temp,
                     hipStream_t s, const uint32_t* n_dev = nullptr, bool temp_zeroed = false,
                     const uint2* aux_in = nullptr, uint2* aux_out = nullptr,   // last pass also writes aux_out[i] = aux_in[value i]
                     const uint32_t* ext_hist = nullptr,
                     bool aux_empty_for_ones = false,        // a key of all ones gets aux (0, 0) without the fetch
                     bool exclusive = false,                 // MGS_FLAG_EXCLUSIVE_DEVICE: small sorts may use block ids as tile ids
                     uint2* ranges = nullptr);               // keys are small integers: ranges[key] = {first, last + 1} sorted position
                                                             // (the final pass: atomicMin / atomicMax on words preset to {~0, 0})
bool radix_wants_hist(uint64_t n);
// The forward's first sort (radix_sort.hip): depth keys -> perm + rect_sorted; three 9-bit passes (+ a fourth that only
// runs for depths beyond 13 107 units).  radix_depth_payload(n): the sort carries the packed rectangles itself.
bool radix_depth_payload(uint64_t n);
int radix_sort_depth(uint32_t* keys, uint32_t* key_b, uint32_t* val_a, uint32_t* val_b, void* rect, bool payload,
                     uint32_t* perm, uint2* rect_sorted, uint64_t n, void* temp, hipStream_t s, bool temp_zeroed,
                     bool exclusive = false);
// the packed form needs both tile-grid dimensions to fit a byte
inline bool depth_sort_payload(int P, int W, int H) { return radix_depth_payload((uint64_t)P) && tiles_x(W) <= 255 && tiles_y(H) <= 255; }
int launch_depth_sort(const GeometryState& g, int P, bool payload, hipStream_t s, bool exclusive = false);
int launch_sort(const GeometryState& g, const BinningState& b, uint64_t R, int bits, hipStream_t s,
                const uint32_t* n_dev = nullptr, bool exclusive = false, uint2* ranges = nullptr);
int set_radix_spin_limit(uint32_t limit);
extern int g_opt_radix_ballot_rank, g_opt_radix_scanned, g_opt_radix_xcd_band, g_opt_radix_tile_items, g_opt_knn_grid_min, g_opt_scan_small, g_opt_dup_slot_major, g_opt_blend_bwd_transposed, g_opt_blend_lds_pad_fwd, g_opt_blend_lds_pad_bwd, g_opt_depth_small;      // test knobs (mgs_debug_set_option)
// `sort_err`: the tile sort's error words (NULL: nothing was sorted); a raised word empties every tile and sets
// MGS_STATUS_TILE_SORT_TIMEOUT in *status (what ranges_kernel did until round 4)
int launch_blend_forward(const mgs_camera& cam, const GeometryState& g, const BinningState& b,
                         const ImageState& img, float* out_color, float* out_depth, float* out_opacity,
                         int32_t* n_touched, const uint32_t* sort_err, uint32_t* status, hipStream_t s);
int launch_blend_backward(const mgs_camera& cam, const GeometryState& g, const BinningState& b,
                          const ImageState& img, const float* dL_dcolor, const float* dL_ddepth,
                          float* grad_acc, bool pose_only, hipStream_t s);
int launch_blend_backward_stats(const mgs_camera& cam, const GeometryState& g, const BinningState& b,
                                const ImageState& img, unsigned long long* stats, hipStream_t s);
int launch_valu_ceiling(float* out, int iters, hipStream_t s);
struct GeomBackwardArgs {
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    const int32_t* radii;
    const float* grad_acc;   // [P][16] from the blend backward
    float *dL_dmeans2D, *dL_dcolors, *dL_dopacity, *dL_dmeans3D, *dL_dcov3D, *dL_dsh, *dL_dscales,
        *dL_drotations, *dL_dtau;
    float* tau_part;         // [TAU_SLOTS][16] zeroed partial pose gradients (large launches), or NULL: add into dL_dtau directly
};
constexpr int TAU_DIRECT_MAX_BLOCKS = 256;  // up to this many workgroups the direct same-address adds are cheaper than a launch
int launch_geom_backward(const mgs_camera& cam, int P, const GeometryState& g, const GeomBackwardArgs& a,
                         hipStream_t s);
int launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* visible, hipStream_t s);
size_t knn_scratch_bytes(int P);
int launch_knn(int P, const float* points, float* out, void* scratch, hipStream_t s);

}  // namespace mgs
