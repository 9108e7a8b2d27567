// Binning: depth order of the Gaussians, prefix sum of tiles touched, duplicate-with-tile-ids,
// stable grouping by tile, per-tile ranges.
//
// Boundary replaced: upstream cub::DeviceScan::InclusiveSum, duplicateWithKeys,
// cub::DeviceRadixSort::SortPairs and identifyTileRanges.
// Integer work; the per-tile lists are bit-exact with the upstream definition
//   order inside a tile = ascending (float32 bits of the view-space depth, Gaussian index)
// but they are produced with far less sort traffic than one 64-bit (tile | depth) sort of all R
// instances:  (1) stable sort of the P Gaussians by their 32 depth bits (ties keep index order),
// (2) instances emitted in that order, (3) stable sort of the R instances on the tile id only
// (13 bits at 1080p: 2 radix passes over 8-byte pairs instead of 6 passes over 12-byte pairs).
// Stability of both sorts makes the result identical to the single 64-bit sort.  The sort itself is
// the hand-written LSD radix sort of radix_sort.hip.  The last pass of the depth sort also lays the tile
// rectangles out in depth order, so the scan and duplicate below read everything coalesced.
#include "common.h"


namespace mgs {

// ------------------------------------------------------------------------------------------------
// inclusive scan of uint32 in two launches: per-block local scan + block sums, exclusive scan of the block sums
// (+ the grand total).  Global offset of element i = local[i] + block_sums[i / SCAN_ITEMS].  256 threads x 8 items.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
__device__ __forceinline__ int scan_nblocks_dev(int P) { return (P + SCAN_ITEMS - 1) / SCAN_ITEMS; }
constexpr int SCAN_PER_THREAD = SCAN_ITEMS / SCAN_THREADS;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int) { return wave_incl_scan_dpp(v); }

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

// Items of a scan block are dealt out wave by wave, row by row: wave w owns items [w * 512, (w + 1) * 512) of the block, item
// (i, lane) = i * 64 + lane of them -- every load and every store of a wave is 64 consecutive elements.  (The first version
// gave a thread 8 CONSECUTIVE items: each of its 8 loads / stores touched all 64 lines of the wave's span an eighth each,
// the shape that cost the per-Gaussian forward 30 us at C5.)  Row i of a wave is scanned with the DPP wave scan and
// offset by the rows before it; the waves are joined through LDS.  Returns the block-wide total; incl[i] = inclusive sum up
// to item (i, lane) within the block.
__device__ __forceinline__ uint32_t scan_block_rows(const uint2* __restrict__ rects, int n, int cbase, int lane, int wv,
                                                    uint32_t (&incl)[SCAN_PER_THREAD], uint32_t* smem /* >= 4 */) {
    uint32_t v[SCAN_PER_THREAD];
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) {
        const int idx = cbase + i * WAVE + lane;
        const uint32_t wh = idx < n ? rects[idx].y : 0u;                   // tiles touched = w * h of the rectangle
        v[i] = (wh & 0xFFFFu) * (wh >> 16);
    }
    uint32_t run = 0;                                                      // wave-uniform: rows before this one
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) {
        const uint32_t sc = wave_incl_scan_dpp(v[i]);
        incl[i] = run + sc;
        run += (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
    }
    if (lane == 0) smem[wv] = run;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wv; ++w) wbase += smem[w];
    const uint32_t total = (smem[0] + smem[1]) + (smem[2] + smem[3]);
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) incl[i] += wbase;
    return total;
}

// out[i] = inclusive sum of the tiles touched by the Gaussians j <= i in depth order (rects = rect_sorted), within the block
__global__ void __launch_bounds__(SCAN_THREADS) scan_local_kernel(const uint2* __restrict__ rects, uint32_t* out,
                                                                  uint32_t* block_sums, int n) {
    __shared__ uint32_t smem[8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cbase = blockIdx.x * SCAN_ITEMS + wv * (SCAN_PER_THREAD * WAVE);
    uint32_t incl[SCAN_PER_THREAD];
    const uint32_t total = scan_block_rows(rects, n, cbase, lane, wv, incl, smem);
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) {
        const int idx = cbase + i * WAVE + lane;
        if (idx < n) out[idx] = incl[i];
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
    if (threadIdx.x == 0) block_sums[nblocks] = carry;      // grand total = number of instances R (read back by the exact path)
}

// ONE launch for P <= SCAN_SMALL_MAX_BLOCKS * SCAN_ITEMS (every SLAM-sized map): each workgroup publishes its total as a
// self-describing 8-byte word {1 << 63 | total} (one relaxed agent-scope store, polled with relaxed agent-scope loads:
// the radix sort's protocol), lane j of its first wave waits for block j < its own, the wave adds them up.  Which block
// of the input a workgroup takes is decided by an atomic ticket (status[SCAN_SMALL_MAX_BLOCKS], zeroed with the status
// words), as in the radix sort: a workgroup then only ever waits for workgroups that have already started, whatever
// else shares the device.  Under MGS_FLAG_EXCLUSIVE_DEVICE the <= 64 workgroups are co-resident (one per CU) and the
// block id serves.  The spin is bounded anyway and raises the depth sort's error word (the consumers of the scan are
// the consumers of that sort).  out[] then holds GLOBAL inclusive sums; the grand total goes to block_sums[nblocks].
constexpr uint32_t SCAN_SPIN_LIMIT = 1u << 22;
__global__ void __launch_bounds__(SCAN_THREADS) scan_small_kernel(const uint2* __restrict__ rects, uint32_t* out,
                                                                  uint32_t* block_sums, uint64_t* status, int n, int nb,
                                                                  uint32_t* err, int ticketed) {
    __shared__ uint32_t smem[8];
    __shared__ uint32_t s_prefix, s_bid;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int bid = (int)blockIdx.x;
    if (ticketed) {
        if (threadIdx.x == 0) s_bid = atomicAdd((uint32_t*)(status + SCAN_SMALL_MAX_BLOCKS), 1u);
        __syncthreads();
        bid = (int)s_bid;
    }
    const int cbase = bid * SCAN_ITEMS + wv * (SCAN_PER_THREAD * WAVE);
    uint32_t incl[SCAN_PER_THREAD];
    const uint32_t total = scan_block_rows(rects, n, cbase, lane, wv, incl, smem);
    if (threadIdx.x == 0)
        __hip_atomic_store(status + bid, (1ull << 63) | (uint64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < WAVE) {
        uint32_t before = 0;
        if ((int)threadIdx.x < bid) {
            uint64_t w = 0;
            uint32_t spins = 0;
            while (true) {
                w = __hip_atomic_load(status + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (w >> 63) break;
                if (++spins > SCAN_SPIN_LIMIT) { atomicExch(err, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            before = (uint32_t)w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
        if (threadIdx.x == 0) s_prefix = before;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; ++i) {
        const int idx = cbase + i * WAVE + lane;
        if (idx < n) out[idx] = incl[i] + s_prefix;
    }
    if (threadIdx.x == 0 && bid == nb - 1) block_sums[nb] = s_prefix + total;
}
int g_opt_scan_small = -1;          // mgs_debug_set_option("scan_small", -1 | 0 | 1): 0 = always the two-launch scan
bool scan_is_small(int P) { return g_opt_scan_small != 0 && scan_nblocks(P) <= SCAN_SMALL_MAX_BLOCKS; }

// (There is no third "add the block offset" pass: the only consumer of the offsets, duplicate_kernel, adds
//  block_sums[i / SCAN_ITEMS] itself -- one launch and a 16-byte-per-Gaussian read-modify-write less.)

// (Measured and rejected: a single-workgroup scan for small P -- 1024 threads x a serial run of dependent
//  gathers each -- took ~300 us at P = 38 k against ~15 us for the three launches below: latency, not launches.)
int launch_scan(const GeometryState& g, int P, hipStream_t s, bool exclusive) {
    if (P == 0) return 0;
    const int nb = scan_nblocks(P);
    if (scan_is_small(P)) {
        hipLaunchKernelGGL(scan_small_kernel, dim3(nb), dim3(SCAN_THREADS), 0, s, g.rect_sorted, g.point_offsets, g.scan_blocks,
                           g.scan_status, P, nb, const_cast<uint32_t*>(radix_depth_error_flag(g.sort_temp, (uint64_t)P)),
                           exclusive ? 0 : 1);
        MGS_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(scan_local_kernel, dim3(nb), dim3(SCAN_THREADS), 0, s, g.rect_sorted, g.point_offsets,
                       g.scan_blocks, P);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, g.scan_blocks, nb);
    MGS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Small maps (P <= DS_SMALL_MAX: what MonoGS's pruning leaves, 8 k - 25 k Gaussians): the WHOLE depth chain -- digit
// histograms, four stable radix passes, the gather of the tile rectangles into depth order and the prefix sum of tiles
// touched -- in ONE launch of ONE 1024-thread workgroup, everything resident in the compute unit's 160 KB of LDS.
//
// Inside a replayed tracking iteration that chain was six launches (rs_hist 4.8 us, four one-sweep passes 6.7 - 9.8 us each,
// scan_small 4.9 us: 40.7 us of a 210 us iteration, profiles/r04_tracking_replay_kernel_sequence.txt), every one of them at
// the floor of its own dependent round-trip chain through L2 (publish digit counts, gather the earlier tiles', scatter), for
// <= 25 k keys that fit one CU.  Here nothing waits for another workgroup and nothing leaves the CU between the passes:
//   * LDS: keys u32[24576] (96 KB) + indices u16[24576] (48 KB) + per-wave digit counters, two 16-bit counts to a word
//     (16 waves x 128 words = 8 KB; a wave holds <= 1536 keys);
//   * wave w owns the contiguous chunk [w * chunk, (w + 1) * chunk) of the array, row i of it is 64 consecutive elements:
//     a pass reads its rows into registers (key, index), ranks every element among its wave's elements of the same digit with
//     ONE returning LDS atomic (the radix sort's ranking: lanes of one DS instruction that hit one address are applied in
//     ascending lane order, a wave's DS instructions in program order -- stable), the 256 digit columns of the 16 x 256 table
//     are scanned by 256 threads, and the elements go back into the SAME arrays at their destinations (every element was
//     read before the barrier in front of the first write);
//   * after the last pass: perm[] (coalesced), the rectangles gathered through the sorted indices (a culled key -- all ones
//     -- gets the empty rectangle without the fetch), w x h scanned row by row with the DPP wave scan, the waves joined
//     through LDS, the grand total where scan_blocks_kernel leaves it.
// Same keys, same stable LSD order: perm, rect_sorted and point_offsets are bit-identical to the multi-launch path
// (tests/test_gpu_sort.py::test_small_depth_chain_matches_the_multi_launch_path; option "depth_small" = 0 disables it).
// ------------------------------------------------------------------------------------------------
constexpr int DS_THREADS = 1024, DS_WAVES = DS_THREADS / WAVE, DS_ROWS = 24;
constexpr int DS_SMALL_MAX = DS_THREADS * DS_ROWS;          // 24 576
__global__ void __launch_bounds__(DS_THREADS) depth_chain_small_kernel(const uint32_t* __restrict__ depth_key,
                                                                       const uint2* __restrict__ rect,
                                                                       uint32_t* __restrict__ perm, uint2* __restrict__ rect_sorted,
                                                                       uint32_t* __restrict__ offsets,
                                                                       uint32_t* __restrict__ grand_total, int n
#ifdef DS_TRACE
                                                                       , unsigned long long* trace
#define DS_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); trace[i] = t_; } } while (0)
#else
#define DS_STAMP(i) do { } while (0)
#endif
                                                                       ) {
    __shared__ uint32_t s_key[DS_SMALL_MAX];
    __shared__ uint16_t s_idx[DS_SMALL_MAX];
    __shared__ uint32_t s_tab[DS_WAVES][128];                 // [wave][digit >> 1]: count of digit d in bits 16 (d & 1) .. +15
    __shared__ uint32_t s_part[DS_WAVES];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rows = (n + DS_THREADS - 1) / DS_THREADS;       // rows of 64 per wave (<= DS_ROWS)
    const int chunk = rows * WAVE;
    const int e0 = wv * chunk + lane;
    // ---- load: key + index, in index order (the order stability refers to)
#pragma unroll
    for (int i = 0; i < DS_ROWS; ++i) {
        const int e = e0 + i * WAVE;
        if (i < rows && e < n) {
            s_key[e] = depth_key[e];
            s_idx[e] = (uint16_t)e;
        }
    }
    uint16_t* const tab16 = reinterpret_cast<uint16_t*>(&s_tab[0][0]);     // [wave][digit] as 16-bit entries
    DS_STAMP(0);
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        for (int t = threadIdx.x; t < DS_WAVES * 128; t += DS_THREADS) (&s_tab[0][0])[t] = 0u;
        __syncthreads();                                       // (also: the previous pass's / the load's LDS writes are visible)
        uint32_t k[DS_ROWS], vr[DS_ROWS];                      // key; index << 16 | rank within (wave, digit)
#pragma unroll
        for (int i = 0; i < DS_ROWS; ++i) {
            const int e = e0 + i * WAVE;
            k[i] = 0u; vr[i] = 0u;
            if (i < rows && e < n) {
                k[i] = s_key[e];
                const uint32_t d = (k[i] >> shift) & 255u, sh = 16u * (d & 1u);
                const uint32_t old = atomicAdd(&s_tab[wv][d >> 1], 1u << sh);
                vr[i] = ((uint32_t)s_idx[e] << 16) | ((old >> sh) & 0xFFFFu);
            }
        }
        __syncthreads();
        DS_STAMP(1 + 4 * pass);
        // column scan: thread d < 256 turns the 16 per-wave counts of digit d into exclusive prefixes over the waves
        uint32_t tot = 0;
        if (threadIdx.x < 256) {
#pragma unroll
            for (int w = 0; w < DS_WAVES; ++w) {
                const uint32_t c = tab16[w * 256 + threadIdx.x];
                tab16[w * 256 + threadIdx.x] = (uint16_t)tot;
                tot += c;
            }
        }
        // exclusive scan of the 256 digit totals (threads of the first four waves), added to every wave's entry
        if (threadIdx.x < 256) {
            const uint32_t incl = wave_incl_scan_dpp(tot);
            if (lane == 63) s_part[wv] = incl;
            tot = incl - tot;                                  // exclusive within the wave
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            uint32_t base = tot;
            for (int w = 0; w < wv; ++w) base += s_part[w];
#pragma unroll
            for (int w = 0; w < DS_WAVES; ++w) tab16[w * 256 + threadIdx.x] = (uint16_t)(tab16[w * 256 + threadIdx.x] + base);
        }
        __syncthreads();
        DS_STAMP(2 + 4 * pass);
#pragma unroll
        for (int i = 0; i < DS_ROWS; ++i) {
            const int e = e0 + i * WAVE;
            if (i < rows && e < n) {
                const uint32_t d = (k[i] >> shift) & 255u;
                const uint32_t dst = (uint32_t)tab16[wv * 256 + d] + (vr[i] & 0xFFFFu);
                s_key[dst] = k[i];
                s_idx[dst] = (uint16_t)(vr[i] >> 16);
            }
        }
        __syncthreads();                                       // (every wave is done with the table before the next pass clears it)
        DS_STAMP(3 + 4 * pass);
    }
    // ---- outputs, in depth order: perm, rect_sorted, inclusive sums of tiles touched
    uint2 rc[DS_ROWS];
#pragma unroll
    for (int i = 0; i < DS_ROWS; ++i) {
        const int e = e0 + i * WAVE;
        rc[i] = make_uint2(0u, 0u);
        if (i < rows && e < n) {
            const uint32_t g = s_idx[e];
            perm[e] = g;
            if (s_key[e] != 0xFFFFFFFFu) rc[i] = rect[g];
        }
    }
    DS_STAMP(17);
    uint32_t incl[DS_ROWS], run = 0;
#pragma unroll
    for (int i = 0; i < DS_ROWS; ++i) {
        const uint32_t tt = (rc[i].y & 0xFFFFu) * (rc[i].y >> 16);
        const uint32_t sc = wave_incl_scan_dpp(tt);
        incl[i] = run + sc;
        run += (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
    }
    if (lane == 0) s_part[wv] = run;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (int w = 0; w < DS_WAVES; ++w) {
        const uint32_t p = s_part[w];
        wbase += w < wv ? p : 0u;
        total += p;
    }
#pragma unroll
    for (int i = 0; i < DS_ROWS; ++i) {
        const int e = e0 + i * WAVE;
        if (i < rows && e < n) {
            rect_sorted[e] = rc[i];
            offsets[e] = incl[i] + wbase;
        }
    }
    if (threadIdx.x == 0) *grand_total = total;
    DS_STAMP(18);
}
// MEASURED, NOT THE DEFAULT (tools/ubench/depth_small_bench, 20 000 keys): 63 us against 40.7 us for the six launches --
// count + rank 4.5 - 5.7 us per pass of evenly spread digits and 14.3 us for the exponent byte (64 lanes on three counters),
// scatter 2.6 us, the rectangle gather 9.1 us (ONE compute unit's texture path takes one cache line per cycle), the stores 5 us:
// a single CU's LDS and memory pipes are what 16 - 40 CUs share in the multi-launch path.  Replayed room run: tracking 4 525
// against 5 467 it/s.  Kept behind mgs_debug_set_option("depth_small", 1) with its test; DESIGN.md section 4.
int g_opt_depth_small = 0;          // mgs_debug_set_option("depth_small", 0 | 1): 1 = the single-workgroup chain for maps <= 24 576
bool depth_chain_is_small(int P) { return g_opt_depth_small > 0 && P > 0 && P <= DS_SMALL_MAX; }
int launch_depth_chain_small(const GeometryState& g, int P, hipStream_t s) {
    hipLaunchKernelGGL(depth_chain_small_kernel, dim3(1), dim3(DS_THREADS), 0, s, g.depth_key, g.rect, g.perm, g.rect_sorted,
                       g.point_offsets, g.scan_blocks + scan_nblocks(P), P
#ifdef DS_TRACE
                       , (unsigned long long*)nullptr
#endif
                       );
    MGS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// duplicate: thread i takes the i-th Gaussian in depth order and walks its tile rectangle
// (y outer, x inner), emitting (tile id, Gaussian index)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) duplicate_kernel(int P, const uint2* __restrict__ rects,
                                                        const uint32_t* __restrict__ perm,
                                                        const uint32_t* __restrict__ offsets /* block-local inclusive sums */,
                                                        const uint32_t* __restrict__ block_sums /* [nb] exclusive, [nb] = total */,
                                                        uint32_t* keys,
                                                        uint32_t* vals, int gx, int gy, uint32_t r_cap,
                                                        int32_t* __restrict__ n_touched, uint2* __restrict__ ranges,
                                                        int ntiles, uint32_t* __restrict__ zero_ptr, size_t zero_words,
                                                        uint32_t* __restrict__ count, uint32_t* __restrict__ overflow,
                                                        const uint32_t* __restrict__ depth_err, int offsets_global,
                                                        uint32_t* __restrict__ hist /* [4][256] or NULL */, int hist_passes,
                                                        int n_threads /* of this launch */, int slot_major) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // (launched with 256 threads; blockDim would be a packet fetch)
    const int lane = threadIdx.x & 63;
    // scratch of the tile sort that follows (was its own launch)
    grid_zero(zero_ptr, zero_words, (size_t)n_threads, 256);
    // a depth sort whose look-back timed out left perm / rect_sorted / offsets partly unwritten: emit nothing (the live
    // count is published as 0, so the tile sort, the ranges and the blend kernels have nothing to do either)
    const bool depth_bad = depth_err && radix_failed(depth_err) != 0u;
    if (i == 0 && count) {                               // capacity mode: live instance count + overflow flag (was a launch)
        const uint32_t R = (P > 0 && !depth_bad) ? block_sums[scan_nblocks_dev(P)] : 0u;
        count[0] = min(R, r_cap);
        count[1] = R > r_cap ? 1u : 0u;
    }
    // status word of this forward (see MGS_STATUS_* in monogs_raster.h): capacity overflow | depth-sort look-back timeout;
    // blend_forward_kernel adds the tile sort's flag
    if (i == 0 && overflow)
        overflow[0] = (count ? count[1] : 0u) | (depth_bad ? (uint32_t)MGS_STATUS_DEPTH_SORT_TIMEOUT : 0u);
    // empty-tile default, in the form the tile sort's final pass takes its minima / maxima over (blend_forward_kernel turns
    // what is still {~0, 0} afterwards into {0, 0})
    if (i < ntiles) ranges[i] = make_uint2(0xFFFFFFFFu, 0u);
    if (i < P) n_touched[i] = 0;                         // (was a memset)
    uint32_t idx = 0, nt = 0, off = 0;
    int x0 = 0, y0 = 0, x1 = 0;
    if (i < P) {                                         // everything in depth order and coalesced: no gather
        idx = perm[i];
        const uint2 r = rects[i];                        // the rectangle preprocess computed (0 x 0: culled)
        const int w = (int)(r.y & 0xFFFFu), h = (int)(r.y >> 16);
        nt = (uint32_t)(w * h);
        x0 = (int)(r.x & 0xFFFFu); y0 = (int)(r.x >> 16); x1 = x0 + w;
    }
    if (depth_bad) nt = 0;                               // (wave-uniform, grid-uniform)
    if (nt) off = i == 0 ? 0u : offsets[i - 1] + (offsets_global ? 0u : block_sums[(i - 1) / SCAN_ITEMS]);
    // Load-balanced emission.  The 64 Gaussians of the wave own the consecutive output slots [S, E); the wave walks
    // that range 64 slots at a time (aligned, so every store is one coalesced 256-byte line) and each lane finds the
    // owner of its slot: owners mark their first slot in LDS, an inclusive max-scan spreads the mark to the right.
    // A thread walking its own rectangle instead wrote 64 interleaved streams per wave (73 us at C5 for 26 MB).
    __shared__ uint32_t s_own[4][WAVE];
    __shared__ uint32_t s_idx[4][WAVE], s_off[4][WAVE];
    __shared__ int s_x0[4][WAVE], s_y0[4][WAVE], s_w[4][WAVE];
    // Digit counts of the emitted tile ids (every pass of the tile sort), collected in LDS while the keys are written and
    // added to the global table once per workgroup: the one-sweep sort then needs no histogram launch of its own.
    __shared__ uint32_t s_hist[2][256];                                     // tile ids have <= 16 bits: two digits
    if (hist) {
        s_hist[0][threadIdx.x] = 0u; s_hist[1][threadIdx.x] = 0u;
        __syncthreads();
    }
    const int wv = threadIdx.x >> 6;
    if (slot_major) {
        // ---- emission balanced by OUTPUT SLOTS.  Gaussian-major (below), a wave owns the slots of its 64 Gaussians: in depth
        // order the near ones come first and cover hundreds of tiles each, so the first waves of the launch walk tens of
        // thousands of slots while most walk a few dozen (102 k Gaussians at 1200x680: 81 us for 1.3 M instances, against
        // 19 us for 5.3 M at C5 where every splat is small).  Here every wave of the launch owns an equal share [s0, s1) of
        // the slots, finds the Gaussian that owns s0 by a 64-way search of the scanned offsets (3 dependent probes at
        // 100 k Gaussians) and walks the Gaussians from there, 64 at a time, with the same owner-marking as below.
        const uint32_t R_all = (P > 0 && !depth_bad) ? block_sums[scan_nblocks_dev(P)] : 0u;
        const uint32_t R_emit = min(R_all, r_cap);
        const uint32_t n_waves = (uint32_t)n_threads / WAVE;
        const uint32_t CH = (((R_emit + n_waves - 1u) / n_waves) + 63u) & ~63u;
        const uint32_t wave_id = (uint32_t)blockIdx.x * 4u + (uint32_t)wv;
        const uint64_t s0w = (uint64_t)wave_id * CH;                             // (64-bit: wave_id * CH may pass 2^32 for R near 2^32)
        const uint32_t s0 = s0w < (uint64_t)R_emit ? (uint32_t)s0w : R_emit, s1 = (uint64_t)s0 + CH < (uint64_t)R_emit ? s0 + CH : R_emit;
        auto incl = [&](uint32_t k) { return offsets[k] + (offsets_global ? 0u : block_sums[k / SCAN_ITEMS]); };
        uint32_t g = 0;
        if (s0 < s1) {      // smallest g with incl(g) > s0: it exists because incl(P - 1) = R_all > s0
            uint32_t lo = 0, hi = (uint32_t)P;
            while (hi - lo > 1u) {
                const uint32_t span = hi - lo, step = (span + 63u) / 64u;
                const uint32_t probe = lo + min(((uint32_t)lane + 1u) * step, span) - 1u;
                const unsigned long long above = __builtin_amdgcn_ballot_w64(incl(probe) > s0);      // monotone in the lane
                const int k = __builtin_ctzll(above);                                                // (the last probe is hi - 1: set)
                const uint32_t pk = lo + min(((uint32_t)k + 1u) * step, span) - 1u;
                const uint32_t pk1 = k == 0 ? lo : lo + min((uint32_t)k * step, span);               // one past probe k - 1
                lo = pk1; hi = pk + 1u;
            }
            g = lo;
        }
        uint32_t sl = s0;
        while (sl < s1 && g < (uint32_t)P) {                                   // wave-uniform
            const uint32_t gi = g + (uint32_t)lane;
            const bool valid = gi < (uint32_t)P;
            uint32_t inc = 0, ntg = 0, idg = 0;
            int gx0 = 0, gy0 = 0, gw = 1;
            if (valid) {
                inc = incl(gi);
                const uint2 r = rects[gi];
                gw = (int)(r.y & 0xFFFFu);
                ntg = (uint32_t)(gw * (int)(r.y >> 16));
                gx0 = (int)(r.x & 0xFFFFu); gy0 = (int)(r.x >> 16);
                idg = perm[gi];
            }
            const uint32_t exc = inc - ntg;                                    // first slot of the Gaussian
            uint32_t Eg = valid ? inc : 0u;                                    // end of the group's slots = the largest inclusive sum
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) Eg = max(Eg, (uint32_t)__shfl_xor((int)Eg, o2, 64));
            __builtin_amdgcn_wave_barrier();
            s_idx[wv][lane] = idg; s_off[wv][lane] = exc; s_x0[wv][lane] = gx0; s_y0[wv][lane] = gy0; s_w[wv][lane] = max(gw, 1);
            const uint32_t e = min(s1, Eg);
            const uint32_t cb0 = sl & ~63u;
            const unsigned long long before = __builtin_amdgcn_ballot_w64(ntg != 0u && exc < cb0);
            uint32_t carry = before ? 64u - (uint32_t)__builtin_clzll(before) : 0u;      // owner (lane + 1) of the slot before the chunk
            for (uint32_t cb = cb0; cb < e; cb += WAVE) {
                __builtin_amdgcn_wave_barrier();
                s_own[wv][lane] = 0u;
                __builtin_amdgcn_wave_barrier();
                if (ntg != 0u && exc >= cb && exc < cb + WAVE) s_own[wv][exc - cb] = (uint32_t)lane + 1u;
                __builtin_amdgcn_wave_barrier();
                uint32_t m = s_own[wv][lane];
#define MGS_DPP_MAX(ctrl, row_mask) m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, ctrl, row_mask, 0xf, false))
                MGS_DPP_MAX(0x111, 0xf); MGS_DPP_MAX(0x112, 0xf); MGS_DPP_MAX(0x114, 0xf); MGS_DPP_MAX(0x118, 0xf);
                MGS_DPP_MAX(0x142, 0xa); MGS_DPP_MAX(0x143, 0xc);
#undef MGS_DPP_MAX
                const uint32_t owner = max(m, carry);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)owner, 63);
                const uint32_t o = cb + lane;
                if (owner != 0u && o >= sl && o < e) {                        // (e <= R_emit <= r_cap: never past the buffers)
                    const uint32_t L = owner - 1u;
                    const uint32_t t = o - s_off[wv][L];
                    const uint32_t w = (uint32_t)s_w[wv][L];
                    const uint32_t yy = t / w, xx = t - yy * w;
                    const uint32_t key = (uint32_t)((s_y0[wv][L] + (int)yy) * gx + s_x0[wv][L] + (int)xx);
                    keys[o] = key;
                    vals[o] = s_idx[wv][L];
                    if (hist) {
                        atomicAdd(&s_hist[0][key & 0xFFu], 1u);
                        if (hist_passes > 1) {      // (few values: the lanes that share the first active lane's digit add once)
                            const uint32_t d1 = (key >> 8) & 0xFFu;
                            const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)d1);
                            const unsigned long long same = __builtin_amdgcn_ballot_w64(d1 == lead);
                            if (d1 != lead) atomicAdd(&s_hist[1][d1], 1u);
                            else if (lane == (int)__builtin_ctzll(same)) atomicAdd(&s_hist[1][lead], (uint32_t)__popcll(same));
                        }
                    }
                }
            }
            sl = max(sl, e);
            if (Eg <= sl) g += WAVE;                                           // the group is used up
        }
        if (hist) {
            __syncthreads();
            const uint32_t c0 = s_hist[0][threadIdx.x], c1 = s_hist[1][threadIdx.x];
            if (c0) atomicAdd(hist + threadIdx.x, c0);
            if (c1) atomicAdd(hist + 256 + threadIdx.x, c1);
        }
        return;
    }
    s_idx[wv][lane] = idx; s_off[wv][lane] = off; s_x0[wv][lane] = x0; s_y0[wv][lane] = y0; s_w[wv][lane] = max(x1 - x0, 1);
    const unsigned long long live = __builtin_amdgcn_ballot_w64(nt != 0);
    if (live == 0ull && !hist) return;                                      // wave-uniform
    const int first = live ? __builtin_ctzll(live) : 0, lastl = live ? 63 - __builtin_clzll(live) : 0;
    const uint32_t S = live ? (uint32_t)__builtin_amdgcn_readlane((int)off, first) : 0u;
    const uint32_t E = live ? (uint32_t)__builtin_amdgcn_readlane((int)(off + nt), lastl) : 0u;
    uint32_t carry = 0;                                                     // owner (lane + 1) of the slot before the chunk
    for (uint32_t cb = S & ~63u; cb < E; cb += WAVE) {
        __builtin_amdgcn_wave_barrier();
        s_own[wv][lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (nt != 0 && off >= cb && off < cb + WAVE) s_own[wv][off - cb] = (uint32_t)lane + 1u;
        __builtin_amdgcn_wave_barrier();
        uint32_t m = s_own[wv][lane];
#define MGS_DPP_MAX(ctrl, row_mask) m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, ctrl, row_mask, 0xf, false))
        MGS_DPP_MAX(0x111, 0xf); MGS_DPP_MAX(0x112, 0xf); MGS_DPP_MAX(0x114, 0xf); MGS_DPP_MAX(0x118, 0xf);
        MGS_DPP_MAX(0x142, 0xa); MGS_DPP_MAX(0x143, 0xc);
#undef MGS_DPP_MAX
        const uint32_t owner = max(m, carry);
        carry = (uint32_t)__builtin_amdgcn_readlane((int)owner, 63);
        const uint32_t o = cb + lane;
        if (owner != 0u && o < E && o < r_cap) {                            // capacity mode: never write past the buffers
            const uint32_t L = owner - 1u;
            const uint32_t t = o - s_off[wv][L];
            const uint32_t w = (uint32_t)s_w[wv][L];
            const uint32_t yy = t / w, xx = t - yy * w;
            const uint32_t key = (uint32_t)((s_y0[wv][L] + (int)yy) * gx + s_x0[wv][L] + (int)xx);
            keys[o] = key;
            vals[o] = s_idx[wv][L];
            if (hist) {
                atomicAdd(&s_hist[0][key & 0xFFu], 1u);
                if (hist_passes > 1) {
                    // the high digit takes a handful of values: the lanes that share the first active lane's digit add once
                    const uint32_t d1 = (key >> 8) & 0xFFu;
                    const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)d1);
                    const unsigned long long same = __builtin_amdgcn_ballot_w64(d1 == lead);
                    if (d1 != lead) atomicAdd(&s_hist[1][d1], 1u);
                    else if (lane == (int)__builtin_ctzll(same)) atomicAdd(&s_hist[1][lead], (uint32_t)__popcll(same));
                }
            }
        }
    }
    if (hist) {
        __syncthreads();
        const uint32_t c0 = s_hist[0][threadIdx.x], c1 = s_hist[1][threadIdx.x];
        if (c0) atomicAdd(hist + threadIdx.x, c0);
        if (c1) atomicAdd(hist + 256 + threadIdx.x, c1);
    }
}

constexpr uint64_t DUP_SLOT_MAJOR_MIN = 768ull * 1024;      // instances (capacity in capacity mode)
int g_opt_dup_slot_major = -1;      // mgs_debug_set_option("dup_slot_major", -1 | 0 | 1): -1 = by size

int launch_duplicate(const mgs_camera& cam, int P, const GeometryState& g, const BinningState& b, uint64_t r_cap,
                     int32_t* n_touched, const ImageState& img, uint64_t sort_n, int sort_bits, uint32_t* count,
                     uint32_t* overflow, hipStream_t s) {
    const uint32_t* depth_err = P > 0 ? radix_depth_error_flag(g.sort_temp, (uint64_t)P) : nullptr;
    const int ntiles = tiles_x(cam.image_width) * tiles_y(cam.image_height);
    int n = P > ntiles ? P : ntiles;
    if (n == 0) return 0;
    // Many instances AND many instances per Gaussian (big splats: the near ones cover hundreds of tiles): emission balanced by
    // output slots (see the kernel), with enough waves for ~512 slots each.  Otherwise the Gaussian-major walk: no search, fewer
    // loads per slot (C5, 2.6 instances per Gaussian: 22 us against 31 us slot-major; 102 k Gaussians at 1200x680, 13 per
    // Gaussian: 81 us against 8.6 us; 39 k at VGA, 415 k instances: 7.5 against 8.8 us).
    const bool slot_major = P > 0 && g_opt_dup_slot_major != 0 &&
                            (g_opt_dup_slot_major > 0 || (sort_n >= DUP_SLOT_MAJOR_MIN && sort_n >= 6ull * (uint64_t)P));
    if (slot_major) {
        const uint64_t want = sort_n / 8;
        if (want > (uint64_t)n) n = (int)(want > 0x3FFFFFFFull ? 0x3FFFFFFFull : want);
    }
    const int n_threads = (n + 255) / 256 * 256;
    uint32_t* zero_ptr = nullptr;
    size_t zero_words = 0;
    if (sort_n > 0) radix_zero_region(b.sort_temp, sort_n, sort_bits, &zero_ptr, &zero_words);
    const bool count_digits = P > 0 && sort_bits <= 16 && radix_wants_hist(sort_n);
    hipLaunchKernelGGL(duplicate_kernel, dim3(n_threads / 256), dim3(256), 0, s, P, g.rect_sorted, g.perm, g.point_offsets,
                       g.scan_blocks, b.keys_a, b.vals_a, tiles_x(cam.image_width), tiles_y(cam.image_height),
                       (uint32_t)(r_cap > 0xFFFFFFFFull ? 0xFFFFFFFFull : r_cap), n_touched, img.ranges, ntiles, zero_ptr,
                       zero_words, count, overflow, depth_err, (P > 0 && (scan_is_small(P) || depth_chain_is_small(P))) ? 1 : 0,
                       count_digits ? g.tile_hist : nullptr, (sort_bits + 7) / 8, n_threads, slot_major ? 1 : 0);
    MGS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// the two stable sorts (radix_sort.hip)
// ------------------------------------------------------------------------------------------------
size_t sort_temp_bytes(uint64_t n, int bits) { return radix_temp_bytes(n, bits); }

int launch_depth_sort(const GeometryState& g, int P, bool payload, hipStream_t s, bool exclusive) {
    // the scratch was cleared by preprocess_forward_kernel; the final pass also lays the tile rectangles out in depth
    // order for the scan and duplicate (`payload`: they travelled with the pairs, else it gathers them)
    return radix_sort_depth(g.depth_key, g.depth_alt, g.iota, g.iota_alt, g.rect, payload, g.perm, g.rect_sorted, (uint64_t)P,
                            g.sort_temp, s, true, exclusive);
}

int launch_sort(const GeometryState& g, const BinningState& b, uint64_t R, int bits, hipStream_t s, const uint32_t* n_dev,
                bool exclusive, uint2* ranges) {
    if (R == 0) return 0;
    // the scratch was cleared by duplicate_kernel, which also counted the digits of the keys it emitted (small sorts; the
    // same predicate as launch_duplicate's `count_digits` -- R > 0 implies P > 0, forward_render_impl)
    const bool counted = bits <= 16 && radix_wants_hist(R) && g.tile_hist != nullptr;
    return radix_sort_pairs(b.keys_a, b.vals_a, b.keys_b, b.vals_b, R, bits, b.sort_temp, s, n_dev, true, nullptr, nullptr,
                            counted ? g.tile_hist : nullptr, false, exclusive, ranges);
}

// ------------------------------------------------------------------------------------------------
// tile ranges: written by the tile sort's final pass (radix_sort.hip, RsPassArgs::ranges) since round 4 -- the kernel that
// compared neighbouring sorted keys (one launch, 4 R bytes of reads) is gone; blend_forward_kernel checks the sort's error
// words and normalises the empty tiles.
// ------------------------------------------------------------------------------------------------

}  // namespace mgs
