// Binning: prefix sum of tiles touched, duplicate-with-keys, (tile | depth) sort, per-tile ranges.
//
// Boundary replaced: upstream cub::DeviceScan::InclusiveSum, duplicateWithKeys,
// cub::DeviceRadixSort::SortPairs and identifyTileRanges.
// Integer work; results are bit-exact by construction:
//   key   = tile_id << 32 | float32 bits of the view-space depth,  tile_id = y * grid_x + x
//   order = ascending key, ties in emission order (Gaussian index ascending, stable sort)
#include "common.h"

#include <cstring>
#include <rocprim/rocprim.hpp>

namespace mgs {

// ------------------------------------------------------------------------------------------------
// inclusive scan of uint32, three launches: per-block reduce+local scan, scan of block sums, add.
// 256 threads x 8 items.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER_THREAD = SCAN_ITEMS / SCAN_THREADS;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(v, o, 64);
        if (lane >= o) v += n;
    }
    return v;
}

// block-wide exclusive prefix of one value per thread; returns the prefix, writes the total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t& total, uint32_t* smem /* >=5 */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) smem[wv] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wv; ++w) base += smem[w];
    total = smem[0] + smem[1] + smem[2] + smem[3];
    __syncthreads();
    return base + incl - v;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_local_kernel(const uint32_t* in, uint32_t* out,
                                                                  uint32_t* block_sums, int n) {
    __shared__ uint32_t smem[8];
    const int base = blockIdx.x * SCAN_ITEMS + threadIdx.x * SCAN_PER_THREAD;
    uint32_t v[SCAN_PER_THREAD];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0u;
        sum += v[i];
    }
    uint32_t total;
    uint32_t run = block_excl_scan(sum, total, smem);
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) {
        run += v[i];
        if (base + i < n) out[base + i] = run;
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single block: exclusive scan of the block sums in place (nblocks <= a few thousand)
__global__ void __launch_bounds__(SCAN_THREADS) scan_blocks_kernel(uint32_t* block_sums, int nblocks) {
    __shared__ uint32_t smem[8];
    uint32_t carry = 0;
    for (int start = 0; start < nblocks; start += SCAN_THREADS) {
        const int i = start + threadIdx.x;
        const uint32_t v = i < nblocks ? block_sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan(v, total, smem);
        if (i < nblocks) block_sums[i] = carry + ex;
        carry += total;
    }
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_add_kernel(uint32_t* out, const uint32_t* block_sums, int n) {
    const uint32_t add = block_sums[blockIdx.x];
    const int base = blockIdx.x * SCAN_ITEMS + threadIdx.x * SCAN_PER_THREAD;
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i)
        if (base + i < n) out[base + i] += add;
}

int launch_scan(const GeometryState& g, int P, hipStream_t s) {
    if (P == 0) return 0;
    const int nb = scan_nblocks(P);
    hipLaunchKernelGGL(scan_local_kernel, dim3(nb), dim3(SCAN_THREADS), 0, s, g.tiles_touched, g.point_offsets,
                       g.scan_blocks, P);
    if (nb > 1) {
        hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, g.scan_blocks, nb);
        hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(SCAN_THREADS), 0, s, g.point_offsets, g.scan_blocks, P);
    }
    MGS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// duplicate with keys: one thread per Gaussian walks its tile rectangle (y outer, x inner)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) duplicate_kernel(int P, const float* __restrict__ rec,
                                                        const uint32_t* __restrict__ offsets,
                                                        const uint32_t* __restrict__ tiles_touched, uint64_t* keys,
                                                        uint32_t* vals, int gx, int gy) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    if (tiles_touched[idx] == 0) return;          // culled: its record was never written
    uint32_t off = idx == 0 ? 0u : offsets[idx - 1];
    const float4 r0 = reinterpret_cast<const float4*>(rec + (size_t)idx * REC_FLOATS)[0];
    const float radius = rec[(size_t)idx * REC_FLOATS + R_RADIUS];
    const float px = r0.x, py = r0.y;
    // same expressions as preprocess (integer truncation of a float quotient)
    const int x0 = min(gx, max(0, (int)((px - radius) / (float)TILE)));
    const int y0 = min(gy, max(0, (int)((py - radius) / (float)TILE)));
    const int x1 = min(gx, max(0, (int)(((px + radius) + (float)(TILE - 1)) / (float)TILE)));
    const int y1 = min(gy, max(0, (int)(((py + radius) + (float)(TILE - 1)) / (float)TILE)));
    const uint64_t depth_bits = (uint64_t)__float_as_uint(rec[(size_t)idx * REC_FLOATS + R_DEPTH]);
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | depth_bits;
            vals[off] = (uint32_t)idx;
            ++off;
        }
}

int launch_duplicate(const mgs_camera& cam, int P, const GeometryState& g, const BinningState& b, hipStream_t s) {
    if (P == 0) return 0;
    hipLaunchKernelGGL(duplicate_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g.rec, g.point_offsets, g.tiles_touched,
                       b.keys_unsorted, b.vals_unsorted, tiles_x(cam.image_width), tiles_y(cam.image_height));
    MGS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// sort (round 1: rocPRIM's LSD radix sort restricted to the live key bits)
// ------------------------------------------------------------------------------------------------
size_t sort_temp_bytes(uint64_t R, int bits) {
    size_t bytes = 0;
    if (R == 0) return 256;
    hipError_t e = rocprim::radix_sort_pairs<rocprim::default_config, uint64_t*, uint64_t*, uint32_t*, uint32_t*>(
        nullptr, bytes, nullptr, nullptr, nullptr, nullptr, (size_t)R, 0u, (unsigned)bits, (hipStream_t)0, false);
    if (e != hipSuccess) return 0;
    return bytes + 256;
}

int launch_sort(const BinningState& b, uint64_t R, int bits, hipStream_t s) {
    if (R == 0) return 0;
    size_t bytes = b.sort_temp_bytes;
    MGS_HIP((rocprim::radix_sort_pairs<rocprim::default_config, uint64_t*, uint64_t*, uint32_t*, uint32_t*>(
        b.sort_temp, bytes, b.keys_unsorted, b.keys_sorted, b.vals_unsorted, b.vals_sorted, (size_t)R, 0u,
        (unsigned)bits, s, false)));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// tile ranges
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ranges_kernel(uint64_t R, const uint64_t* __restrict__ keys, uint2* ranges) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const uint32_t t = (uint32_t)(keys[i] >> 32);
    if (i == 0) {
        ranges[t].x = 0;
    } else {
        const uint32_t tp = (uint32_t)(keys[i - 1] >> 32);
        if (t != tp) {
            ranges[tp].y = (uint32_t)i;
            ranges[t].x = (uint32_t)i;
        }
    }
    if (i == R - 1) ranges[t].y = (uint32_t)R;
}

int launch_ranges(const BinningState& b, uint64_t R, const ImageState& img, int ntiles, hipStream_t s) {
    MGS_HIP(hipMemsetAsync(img.ranges, 0, (size_t)ntiles * sizeof(uint2), s));
    if (R == 0) return 0;
    hipLaunchKernelGGL(ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, R, b.keys_sorted, img.ranges);
    MGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace mgs
