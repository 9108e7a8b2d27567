// Stable LSD radix sort of (uint32 key, uint32 value) pairs, hand-written for gfx950 (wave64).
//
// Used twice per forward (binning.hip): the P Gaussians by their 32 depth bits, then the R
// (tile id, Gaussian index) instances by tile id.  Replaces upstream's cub::DeviceRadixSort call
// (SURVEY.md section 2.1 K4).
//
// One 8-bit digit per pass.  A workgroup (256 threads = 4 waves) takes a tile of 1024-4096 pairs, ranks them stably,
// lays the tile out digit by digit in LDS and writes each digit's run to its final place (coalesced runs instead of a
// 256-way scatter).  Two ways of knowing where a tile's runs go, bit-identical results, chosen by size (rs_scanned()):
//
// "One sweep" (<= 640 k pairs: every sort of a SLAM-sized map; fewest launches):
//   * ONE histogram kernel reads the keys once and counts every digit of every pass (each pass kernel
//     scans its 256 counts into exclusive global bases itself);
//   * per pass ONE kernel: the tile publishes its 256 digit counts and obtains the sum of the counts of all EARLIER
//     tiles by decoupled look-back.
// "Counted tiles" (> 640 k pairs; rs_scanned): per pass rs_tile_hist_kernel writes the digit counts of every tile,
//   counts[tile][digit], and adds them into the rows of the tile's ancestors in a tree of fan-out 8 over the tiles; the
//   scatter kernel sums the siblings before each of its ancestors (<= 7 rows per level, <= 16 at the top, all requested
//   together before the keys: one round trip hidden behind the key loads) -- no waiting between workgroups, and no scan
//   launch.  With hundreds of co-resident tiles the look-back's status traffic and round trips dominated the pass.
// The last pass can also gather an auxiliary array through the sorted values (aux_out[pos] = aux_in[value]).
//
// Ranking (round 3): rank of a pair among the pairs of its wave with the same digit = the value a returning LDS atomic
// add on the wave's digit counter hands back -- ONE DS instruction per item.  The first version matched digits with eight
// ballots per item and per-lane 64-bit mask arithmetic: ~120 VALU wave-instructions per item-instruction, which made the
// scatter kernel VALU-issue-bound (tools/ubench/sort_bench.hip: 5.9-7.9 us of a tile's 14-19 us, now 0.8-1.4 us; the C5
// tile sort 110.6 -> 85.9 us, the depth sort 141.5 -> 130.2 us before anything else changed).  Stability needs the lanes
// of one DS instruction that hit one address to be applied in ascending lane order, and a wave's DS instructions in
// program order.  The LDS unit does both (sort_bench and tests/test_gpu_sort.py compare against a stable CPU sort on
// keys with 1-5 distinct digits per wave-instruction, the adversarial case); the ballot ranking is kept behind
// mgs_debug_set_option("radix_ballot_rank", 1) as the reference the tests compare with.
//
// One-sweep details:
// Inter-workgroup protocol (MI355X_MICROARCH.md "Workgroup dispatch ... visibility", form R2): every
// status word is a self-describing 8-byte granule {flag:2, count:62} written by ONE agent-scope
// relaxed atomic store and polled with agent-scope relaxed atomic loads; no other data crosses
// workgroups, so no fence is needed.  Tile ids are handed out by an atomic ticket, so a tile only
// ever waits for tiles that already started (placement-independent forward progress); every spin is
// bounded and raises an error flag instead of hanging.
#include "common.h"

namespace mgs {

// tools/ubench/sort_bench.hip compiles this file with RS_TRACE: thread 0 of every tile stamps the 100 MHz wall clock at the
// phase boundaries of rs_pass_kernel (nothing of it exists in the library build)
#ifdef RS_TRACE
__device__ uint64_t* g_rs_trace = nullptr;      // [tiles][8]
#define RS_STAMP(k) do { if (g_rs_trace && threadIdx.x == 0) g_rs_trace[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#define RS_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
__device__ uint64_t* g_rs_htrace = nullptr;     // [tiles][4]: the same for rs_tile_hist_kernel
#define RS_HSTAMP(k) do { if (g_rs_htrace && threadIdx.x == 0) g_rs_htrace[(size_t)blockIdx.x * 4 + (k)] = wall_clock64(); } while (0)
#else
#define RS_HSTAMP(k)
#define RS_STAMP(k)
#define RS_DRAIN()
#endif

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / WAVE;
constexpr int RS_ITEMS = 16;                       // pairs per thread for large sorts: 4096-pair tiles
constexpr int RS_ITEMS_WIDE = 12;                  // ... whose 4096-pair tiles would not all be resident at once (see rs_tile_items)
constexpr int RS_ITEMS_MID = 8;
constexpr int RS_ITEMS_SMALL = 4;                  // small sorts are latency-bound: 1024-pair tiles rank 4x faster
// Measured on MI355X with the ballot ranking (depth sort of P keys): 40 k: 72 us (16) -> 50 us (4); 400 k: 122 us (32) / 83 us (8);
// 400 k: 108 us (4); 2 M: 158 us (16) / 208 us (8) / 152 us (32).  A few hundred tiles is the sweet spot between the
// per-tile latency and the length of the look-back chain.
// Above this many pairs the counted-tiles path takes over.  Round 2: 1 M keys (depth sort): one sweep 0.134 ms, counted
// 0.101 ms; 500 k: 0.087 vs 0.092 ms; 300 k: 0.075 vs 0.088 ms -- the crossover sits between 512 k and 1 M.
constexpr uint64_t RS_ONE_SWEEP_MAX = 640ull * 1024;
static inline int rs_items(uint64_t n) {
    return n <= 192ull * 1024 ? RS_ITEMS_SMALL : (n <= RS_ONE_SWEEP_MAX ? RS_ITEMS_MID : RS_ITEMS);
}
constexpr int RS_RADIX = 256;
constexpr int RS_MAX_PASSES = 4;
constexpr uint64_t RS_FLAG_LOCAL = 1ull << 62;     // count of this tile only
constexpr uint64_t RS_FLAG_GLOBAL = 2ull << 62;    // inclusive count over tiles 0..this
constexpr uint64_t RS_COUNT_MASK = (1ull << 62) - 1;
constexpr uint32_t RS_SPIN_LIMIT = 1u << 22;
// the bound actually used by the look-back spin: a device word so that a test can shrink it (mgs_debug_set_radix_spin_limit)
__device__ uint32_t g_rs_spin_limit = RS_SPIN_LIMIT;
// predecessor status words fetched per look-back step.  (Measured, round 2: wider windows for the small sorts -- 64 words
// for the 40-tile depth sort of a 40 k map, 32 for the 200-tile tile sort -- on the theory that the look-back is a chain
// of round trips: 12.5-14 us per pass instead of 9.2-10.7, 15.6-16.3 instead of 14.1-14.3.  Most predecessors have
// already published their inclusive count when a tile looks back; the extra loads and the longer consume loop only cost.
//  Narrower windows, 8 and 4 words: 52.9 / 53.4 us against 56.2 us for the depth sort at 100 k -- inside the noise.)
constexpr int RS_WINDOW = 16;

// large sorts take the counted-tiles path; a test knob forces either one (no environment lookups on the launch path)
int g_opt_radix_scanned = -1;       // mgs_debug_set_option("radix_scanned", -1 | 0 | 1): -1 = by size
int g_opt_radix_ballot_rank = 0;    // mgs_debug_set_option("radix_ballot_rank", 1): rank with ballots instead of LDS atomics
static inline bool rs_scanned(uint64_t n) {
    if (g_opt_radix_scanned >= 0) return g_opt_radix_scanned == 1 && rs_items(n) == RS_ITEMS;
    return rs_items(n) == RS_ITEMS;          // > 640 k pairs: hundreds of tiles
}
// Pairs per thread.  On the counted-tiles path a tile lives ~10-20 us and the kernel ends when the last one does: if the
// tiles do not all fit on the chip at once, the stragglers start when the first ones retire and the pass takes two tile
// lives (tile sort at C5, 5.27 M pairs = 1287 tiles of 4096 against 1024 resident: tile starts at 0 and at 14 us).  4096-pair
// tiles: 84 VGPRs + 24 KB of LDS = 4-5 workgroups per CU; 3072-pair tiles: 68 VGPRs + 20 KB = 7 per CU (1792), and 1716 tiles
// at C5 (82.7 us against 85.9; 2048-pair tiles lose it again to the doubled count tables, 104.6 us).
static inline int rs_tile_items(uint64_t n, bool scanned) {
    const int it = rs_items(n);
    if (it != RS_ITEMS || !scanned) return it;
    return n > 1024ull * RS_THREADS * RS_ITEMS ? RS_ITEMS_WIDE : RS_ITEMS;
}
static inline int rs_passes(int bits) { return (bits + 7) / 8; }
static inline uint32_t rs_tiles(uint64_t n, bool scanned) {
    const uint64_t tile = (uint64_t)RS_THREADS * rs_tile_items(n, scanned);
    return (uint32_t)((n + tile - 1) / tile);
}
static inline uint32_t rs_tiles(uint64_t n) { return rs_tiles(n, rs_scanned(n)); }
// counted-tiles path: the digit counts of the tiles form the leaves of a tree of fan-out 4 (level k node i = the sum over
// tiles [i * 4^k, (i + 1) * 4^k)); at most six levels, the top one has <= 8 nodes up to 8192 tiles.  A scatter workgroup
// adds the siblings before each of its ancestors (<= 3 rows per level: one per wave) and the top level.
constexpr int RS_FAN_LOG = 2;             // (fan-out 8 needs two row slots per wave and level: 40 VGPRs of rows in flight
                                          //  instead of 28, which cost the 3072-pair kernel two resident workgroups per CU)
constexpr int RS_TOP_BATCH = 8;           // top-level rows summed per round trip (two per wave)
constexpr int RS_MAX_LEVELS = 6;          // levels 0-4 + the top: one round trip up to 8 * 1024 = 8192 tiles (25-33 M pairs),
                                          // beyond that the top level is walked 8 rows at a time
struct RsTree {
    int levels;                            // level 0 = the tiles themselves
    uint32_t rows[RS_MAX_LEVELS];          // nodes of level k
    uint32_t off[RS_MAX_LEVELS];           // first row of level k >= 1 in the sums table (level 0 lives in `counts`)
    uint32_t sum_rows;                     // rows of levels >= 1
};
static inline RsTree rs_tree(uint32_t tiles) {
    RsTree tr;
    tr.levels = 1; tr.rows[0] = tiles; tr.off[0] = 0; tr.sum_rows = 0;
    while (tr.rows[tr.levels - 1] > (uint32_t)RS_TOP_BATCH && tr.levels < RS_MAX_LEVELS) {
        tr.rows[tr.levels] = (tr.rows[tr.levels - 1] + (1u << RS_FAN_LOG) - 1u) >> RS_FAN_LOG;
        tr.off[tr.levels] = tr.sum_rows;
        tr.sum_rows += tr.rows[tr.levels];
        ++tr.levels;
    }
    return tr;
}

// temp layout: [hist: RS_MAX_PASSES*256 u32][tickets: RS_MAX_PASSES u32][error: RS_MAX_PASSES u32][pad]
//   one sweep:     [status: passes * tiles * 256 u64]
//   counted tiles: [sums: passes * (rows of tree levels >= 1) * 256 u32][counts: tiles * 256 u32 (rewritten by every
//                  pass, never cleared)]
struct RsTemp {
    uint32_t* hist;
    uint32_t* tickets;
    uint32_t* error;
    uint64_t* status;
    uint32_t* gsum;
    uint32_t* counts;
    size_t zero_bytes;     // everything from `hist` that must be zero before the sort
    size_t bytes;
};
static RsTemp rs_carve(void* temp, uint64_t n, int bits, bool scanned) {
    char* p = (char*)align_up((size_t)temp, 256);
    RsTemp t;
    t.hist = (uint32_t*)p;
    t.tickets = t.hist + RS_MAX_PASSES * RS_RADIX;
    t.error = t.tickets + RS_MAX_PASSES;      // one word per pass
    char* q = (char*)align_up((size_t)(t.error + RS_MAX_PASSES), 256);
    t.status = (uint64_t*)q;
    t.gsum = (uint32_t*)q;
    const uint32_t tiles = rs_tiles(n, scanned);
    if (scanned) {
        const size_t gsum_bytes = (size_t)rs_passes(bits) * rs_tree(tiles).sum_rows * RS_RADIX * sizeof(uint32_t);
        t.counts = (uint32_t*)(q + gsum_bytes);
        t.zero_bytes = (size_t)(q - p) + gsum_bytes;
        t.bytes = t.zero_bytes + (size_t)tiles * RS_RADIX * sizeof(uint32_t);
    } else {
        t.counts = nullptr;
        t.zero_bytes = (size_t)(q - p) + (size_t)rs_passes(bits) * tiles * RS_RADIX * sizeof(uint64_t);
        t.bytes = t.zero_bytes;
    }
    return t;
}
size_t radix_temp_bytes(uint64_t n, int bits) {
    if (n == 0) return 256;
    // either path may be forced on a buffer sized earlier (test knob): size for the larger one
    const size_t a = rs_carve(nullptr, n, bits, false).bytes;
    const size_t b = rs_items(n) == RS_ITEMS ? rs_carve(nullptr, n, bits, true).bytes : 0;
    return (a > b ? a : b) + 512;
}
static RsTemp rs_carve(void* temp, uint64_t n, int bits) { return rs_carve(temp, n, bits, rs_scanned(n)); }

// ------------------------------------------------------------------------------------------------

// Digit counts of every pass in one read of the keys.  Each wave keeps a private copy of the
// histograms in LDS (4x fewer same-address LDS atomics; the tile-id digits are low-entropy).
__global__ void __launch_bounds__(RS_THREADS) rs_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                             const uint32_t* __restrict__ n_dev, int npasses,
                                                             uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[RS_WAVES][RS_MAX_PASSES][RS_RADIX];
    if (n_dev) n = min(n, n_dev[0]);          // capacity mode: the live count is on the device
    for (int i = threadIdx.x; i < RS_WAVES * RS_MAX_PASSES * RS_RADIX; i += RS_THREADS) (&lh[0][0][0])[i] = 0;
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    for (uint32_t i = blockIdx.x * RS_THREADS + threadIdx.x; i < n; i += gridDim.x * RS_THREADS) {
        const uint32_t k = keys[i];
        for (int p = 0; p < npasses; ++p) atomicAdd(&lh[wv][p][(k >> (8 * p)) & 0xFF], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npasses * RS_RADIX; i += RS_THREADS) {
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) c += (&lh[w][0][0])[i];
        if (c) atomicAdd(hist + i, c);
    }
}

// ---- counted tiles, step 1: digit counts of every tile for one pass, counts[tile][digit] (one coalesced 1-KB row per
// tile) and the levels >= 1 of the tree over them.  A workgroup of 1024 threads takes the four tiles of ONE level-1 node
// (256 threads each), so that level is a plain store; the higher levels are memory-side adds of 256 contiguous bytes per
// wave-instruction into rows zeroed with the sort's scratch.  (One 256-thread workgroup per tile adding into every
// level: 27 k wave-atomics per pass of the C5 tile sort = 4.7 us of a 14 us kernel at the memory side's ~1.3 TB/s.)
// The LDS counters are private to a wave AND to lane & 3: the 64 lanes of a wave-instruction that hit one counter are
// applied one after the other, and the top byte of a float depth takes ~5 values (5.5 us of counting against 1.7 us
// for a uniform byte before the split).
// The scatter kernel turns the tree into its offsets itself: there is no scan launch (round 3: the per-digit row scan
// over counts[digit][tile] was a launch of 5-6 us per pass, 6 passes per forward at C5).
struct RsTreeArgs {
    int levels;
    uint32_t off[RS_MAX_LEVELS];
    uint32_t top_rows;
};
constexpr int RS_HIST_COPIES = 4;
// TPW = tiles per workgroup: 4 (1024 threads, level 1 stored) from 1024 tiles up, where the adds into level 1 would cost
// more than they do below (and where a quarter as many workgroups still fill the chip); 1 (256 threads, every level added)
// for fewer tiles -- at 489 tiles, 123 workgroups of four left half the CUs idle and the depth sort 10 us slower.
template <int ITEMS, int TPW>
__global__ void __launch_bounds__(RS_THREADS * TPW) rs_tile_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                        const uint32_t* __restrict__ n_dev, int shift,
                                                                        uint32_t tiles, RsTreeArgs tr,
                                                                        uint32_t* __restrict__ counts,
                                                                        uint32_t* __restrict__ sums) {
    static_assert(TPW == 1 || TPW == (1 << RS_FAN_LOG), "a workgroup counts one tile or one level-1 node");
    constexpr uint32_t TILE_PAIRS = RS_THREADS * ITEMS;
    __shared__ uint32_t wh[TPW][RS_WAVES * RS_HIST_COPIES][RS_RADIX];      // 16 KB per tile
    const int t = threadIdx.x & (RS_THREADS - 1), q = threadIdx.x >> 8, wv = t >> 6, lane = t & 63;
    RS_HSTAMP(0);
    const uint32_t n_live = n_dev ? min(n, n_dev[0]) : n;
    const uint32_t tile = blockIdx.x * TPW + q, tile_start = tile * TILE_PAIRS;
    const uint32_t tile_n = tile_start < n_live ? min(TILE_PAIRS, n_live - tile_start) : 0u;     // 0: dead (capacity mode) or past the end
    uint32_t k[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t e = (uint32_t)i * RS_THREADS + t;
        k[i] = e < tile_n ? keys[tile_start + e] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RS_WAVES * RS_HIST_COPIES; ++i) wh[q][i][t] = 0;
    __syncthreads();
    RS_DRAIN();
    RS_HSTAMP(1);
    uint32_t* mine = wh[q][wv * RS_HIST_COPIES + (lane & (RS_HIST_COPIES - 1))];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if ((uint32_t)i * RS_THREADS + t < tile_n) atomicAdd(&mine[(k[i] >> shift) & 0xFFu], 1u);
    __syncthreads();
    RS_HSTAMP(2);
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < RS_WAVES * RS_HIST_COPIES; ++i) c += wh[q][i][t];
    if (tile < tiles) counts[(size_t)tile * RS_RADIX + t] = c;       // dead tiles: zeros
    if (TPW == 1) {
        if (c)
            for (int l = 1; l < tr.levels; ++l)
                atomicAdd(sums + ((size_t)tr.off[l] + (tile >> (RS_FAN_LOG * l))) * RS_RADIX + t, c);
    } else if (tr.levels > 1) {
        __syncthreads();
        wh[q][0][t] = c;
        __syncthreads();
        if (q == 0) {
            uint32_t s1 = 0;
#pragma unroll
            for (int j = 0; j < TPW; ++j) s1 += wh[j][0][t];
            sums[((size_t)tr.off[1] + blockIdx.x) * RS_RADIX + t] = s1;         // level 1: this workgroup's own row
            if (s1)
                for (int l = 2; l < tr.levels; ++l)
                    atomicAdd(sums + ((size_t)tr.off[l] + (blockIdx.x >> (RS_FAN_LOG * (l - 1)))) * RS_RADIX + t, s1);
        }
    }
    RS_DRAIN();
    RS_HSTAMP(3);
}
template <int ITEMS>
static void rs_launch_tile_hist(const uint32_t* keys, uint32_t n, const uint32_t* n_dev, int shift, uint32_t tiles,
                                const RsTreeArgs& tr, uint32_t* counts, uint32_t* sums, hipStream_t s) {
    if (tiles >= 1024u)
        hipLaunchKernelGGL((rs_tile_hist_kernel<ITEMS, 4>), dim3((tiles + 3) / 4), dim3(RS_THREADS * 4), 0, s, keys, n, n_dev, shift,
                           tiles, tr, counts, sums);
    else
        hipLaunchKernelGGL((rs_tile_hist_kernel<ITEMS, 1>), dim3(tiles), dim3(RS_THREADS), 0, s, keys, n, n_dev, shift, tiles, tr,
                           counts, sums);
}

struct RsPassArgs {
    const uint32_t* kin;
    uint32_t* kout;
    const uint32_t* vin;
    uint32_t* vout;
    uint32_t n;               // number of pairs (capacity when n_dev is set)
    const uint32_t* n_dev;    // optional: live count on the device (<= n after clamping)
    int shift;
    const uint32_t* hist;     // one sweep: [256] global count of each digit for this pass
    uint64_t* status;         // one sweep: [tiles][256]
    uint32_t* ticket;
    uint32_t* error;          // [RS_MAX_PASSES] one word per pass: pass p raises error[p] when a look-back spin times out
    int pass;
    const uint2* aux_in;      // optional (last pass): aux_out[final position] = aux_in[value]
    uint2* aux_out;
    int aux_skip_ones;        // a pair whose key is 0xFFFFFFFF gets aux (0, 0) without a fetch (depth sort: culled Gaussians)
    const uint32_t* counts;   // counted tiles: [tiles][256] digit counts of every tile for this pass
    const uint32_t* sums;     // counted tiles: the levels >= 1 of the tree over them (RsTreeArgs)
    RsTreeArgs tree;
};

// value x[i] sits at slot[i] of the digit-ordered tile: returns in x[i] the value of slot i * 256 + t (the staging buffer
// is free again after the barrier the caller places before its next use)
template <int ITEMS>
__device__ __forceinline__ void rs_exchange(uint32_t* __restrict__ sbuf, const uint32_t (&slot)[ITEMS], uint32_t (&x)[ITEMS], int t) {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (slot[i] != 0xFFFFFFFFu) sbuf[slot[i]] = x[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) x[i] = sbuf[min((uint32_t)i * RS_THREADS + (uint32_t)t, (uint32_t)(RS_THREADS * ITEMS - 1))];
}

// SCANNED = false: one sweep -- the tile publishes its digit counts and finds the sum over earlier tiles by decoupled
//                  look-back (fewest launches: right for the small sorts of SLAM-sized maps).
// SCANNED = true:  counted tiles -- the digit counts of every tile and of every group of tiles were written beforehand
//                  by rs_tile_hist_kernel; the tile adds up what lies before it.
// BALLOT:          rank with eight ballots per item instead of one returning LDS atomic (reference for the tests).
template <int ITEMS, bool SCANNED, bool BALLOT, int WINDOW = RS_WINDOW>
__global__ void __launch_bounds__(RS_THREADS, (SCANNED && ITEMS == RS_ITEMS_WIDE) ? 7 : 1) rs_pass_kernel(RsPassArgs a) {
    constexpr int TILE_PAIRS = RS_THREADS * ITEMS;
    __shared__ uint32_t wave_hist[RS_WAVES][RS_RADIX];
    __shared__ uint32_t digit_base[RS_RADIX];
    __shared__ int64_t gbase[RS_RADIX];                  // global position of LDS slot 0 for each digit (may be negative)
    __shared__ uint32_t sbuf[TILE_PAIRS];                // keys and values take turns in ONE staging buffer: half the LDS of
                                                         // two, so more tiles are resident per CU (two barriers more)
    __shared__ uint32_t wsum[RS_WAVES];
    __shared__ uint32_t s_tile;

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    RS_STAMP(0);
    if (!SCANNED) {
        // A timed-out look-back in an EARLIER pass left part of this pass's input unwritten: ranking it against the
        // histogram of the original keys could place pairs past the end of the buffers.  Such a pass does nothing (the
        // words of earlier passes cannot change while this one runs, so every tile takes the same decision), and the
        // consumers of the sort treat a raised flag as "no output" (duplicate_kernel, ranges_kernel).
        uint32_t failed = 0;
        for (int q = 0; q < a.pass; ++q) failed |= a.error[q];
        if (failed) return;
    }
    if (t == 0) s_tile = (!SCANNED && a.ticket) ? atomicAdd(a.ticket, 1u) : blockIdx.x;   // a returning atomic is a ~2 us round trip
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) wave_hist[w][t] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t tile_start = tile * (uint32_t)TILE_PAIRS;
    const uint32_t n_live = a.n_dev ? min(a.n, a.n_dev[0]) : a.n;
    if (tile_start >= n_live) return;        // capacity mode: tiles past the live count have nothing to do
                                             // (tickets are dense, so no live tile ever looks back at them)
    const uint32_t tile_n = min((uint32_t)TILE_PAIRS, n_live - tile_start);

    // ---- digit t: count over the whole input (hcount) and, counted tiles, over everything before this tile (before).
    //      Requested first: needed only after the ranking, their round trips hide behind the key loads.
    //      Counted tiles: a wave-instruction of 16 bytes per lane reads a whole 256-digit row, so the rows are dealt out
    //      to the four waves -- per level the <= 3 siblings before this tile's ancestor (one per wave), at the top all
    //      <= 8 nodes (two per wave).  Every slot is loaded unconditionally (a row index clamped into
    //      the table, weight 0) so that all of them are in flight together: ONE round trip.
    uint32_t hcount = 0, before = 0;
    uint4 lv[RS_MAX_LEVELS - 1], tv[RS_TOP_BATCH / RS_WAVES];
    uint32_t lw[RS_MAX_LEVELS - 1], tw[RS_TOP_BATCH / RS_WAVES], tb[RS_TOP_BATCH / RS_WAVES];
    const int levels = a.tree.levels;
    const uint32_t top_node = tile >> (RS_FAN_LOG * (levels - 1));
    const uint4* __restrict__ c4 = reinterpret_cast<const uint4*>(a.counts);
    const uint4* __restrict__ top4 =
        levels == 1 ? c4 : reinterpret_cast<const uint4*>(a.sums) + (size_t)a.tree.off[levels - 1] * (RS_RADIX / 4);
    if (SCANNED) {
#pragma unroll
        for (int l = 0; l < RS_MAX_LEVELS - 1; ++l) {
            const bool have = l < levels - 1;                                   // wave-uniform
            const uint32_t node = tile >> (RS_FAN_LOG * l), first = node & ~3u, nb = node & 3u;
            const uint4* base = (l == 0 || !have) ? c4 : reinterpret_cast<const uint4*>(a.sums) + (size_t)a.tree.off[l] * (RS_RADIX / 4);
            lw[l] = (have && (uint32_t)wv < nb) ? 1u : 0u;                     // wave w takes sibling w
            lv[l] = base[(size_t)(lw[l] ? first + (uint32_t)wv : 0u) * (RS_RADIX / 4) + lane];
        }
#pragma unroll
        for (int q = 0; q < RS_TOP_BATCH / RS_WAVES; ++q) {
            const uint32_t r = (uint32_t)wv + 4u * q;
            tw[q] = r < a.tree.top_rows ? 1u : 0u;
            tb[q] = r < top_node ? 1u : 0u;
            tv[q] = top4[(size_t)(tw[q] ? r : 0u) * (RS_RADIX / 4) + lane];
        }
    } else {
        hcount = a.hist[t];
    }

    // ---- load (wave-striped: item i of lane l of wave w is element w*(ITEMS*64) + i*64 + l of the tile) and rank
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t e = (uint32_t)wv * (ITEMS * WAVE) + (uint32_t)i * WAVE + lane;
        const bool valid = e < tile_n;
        key[i] = valid ? a.kin[tile_start + e] : 0xFFFFFFFFu;
        val[i] = valid ? a.vin[tile_start + e] : 0u;
    }
    if (SCANNED) {
        // (in program order behind the key loads: waiting for the rows leaves the keys in flight)
        uint4 ab = make_uint4(0u, 0u, 0u, 0u), at = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int l = 0; l < RS_MAX_LEVELS - 1; ++l) {
            ab.x += lw[l] ? lv[l].x : 0u; ab.y += lw[l] ? lv[l].y : 0u; ab.z += lw[l] ? lv[l].z : 0u; ab.w += lw[l] ? lv[l].w : 0u;
        }
#pragma unroll
        for (int q = 0; q < RS_TOP_BATCH / RS_WAVES; ++q) {
            at.x += tw[q] ? tv[q].x : 0u; at.y += tw[q] ? tv[q].y : 0u; at.z += tw[q] ? tv[q].z : 0u; at.w += tw[q] ? tv[q].w : 0u;
            ab.x += tb[q] ? tv[q].x : 0u; ab.y += tb[q] ? tv[q].y : 0u; ab.z += tb[q] ? tv[q].z : 0u; ab.w += tb[q] ? tv[q].w : 0u;
        }
        for (uint32_t r0 = RS_TOP_BATCH; r0 < a.tree.top_rows; r0 += RS_TOP_BATCH)      // > 8192 tiles only
#pragma unroll
            for (int q = 0; q < RS_TOP_BATCH / RS_WAVES; ++q) {
                const uint32_t r = r0 + (uint32_t)wv + 4u * q;
                if (r < a.tree.top_rows) {
                    const uint4 v = top4[(size_t)r * (RS_RADIX / 4) + lane];
                    at.x += v.x; at.y += v.y; at.z += v.z; at.w += v.w;
                    if (r < top_node) { ab.x += v.x; ab.y += v.y; ab.z += v.z; ab.w += v.w; }
                }
            }
        // the four waves' partial rows meet in the (still unused) staging buffer: [wave][before | total][256]
        uint4* part = reinterpret_cast<uint4*>(sbuf);
        part[(wv * 2 + 0) * (RS_RADIX / 4) + lane] = ab;
        part[(wv * 2 + 1) * (RS_RADIX / 4) + lane] = at;
    }
    RS_DRAIN();
    RS_STAMP(1);
    if (!BALLOT) {
        // rank among the pairs of this wave with the same digit, in (item, lane) order = the counter's value before this
        // lane's add (see the header on the order the LDS unit applies them in)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t e = (uint32_t)wv * (ITEMS * WAVE) + (uint32_t)i * WAVE + lane;
            const uint32_t d = (key[i] >> a.shift) & 0xFFu;
            rank[i] = 0;
            if (e < tile_n) rank[i] = __hip_atomic_fetch_add(&wave_hist[wv][d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t e = (uint32_t)wv * (ITEMS * WAVE) + (uint32_t)i * WAVE + lane;
            const bool valid = e < tile_n;
            const uint32_t d = (key[i] >> a.shift) & 0xFFu;
            unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bm = __builtin_amdgcn_ballot_w64(valid && bit);
                peers &= bit ? bm : ~bm;
            }
            rank[i] = 0;
            if (valid) {
                const uint32_t prev = wave_hist[wv][d];                     // every peer reads the same counter ...
                rank[i] = prev + (uint32_t)__popcll(peers & lt_mask);
                if ((peers & lt_mask) == 0ull) wave_hist[wv][d] = prev + (uint32_t)__popcll(peers);   // ... the first peer bumps it
            }
        }
    }
    __syncthreads();
    if (SCANNED) {
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            before += sbuf[(w * 2 + 0) * RS_RADIX + t];
            hcount += sbuf[(w * 2 + 1) * RS_RADIX + t];
        }
    }
    RS_STAMP(2);

    // ---- thread d owns digit d: per-wave exclusive prefixes, tile total, position of the digit inside the tile
    uint32_t total;
    {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            const uint32_t c = wave_hist[w][t];
            wave_hist[w][t] = run;
            run += c;
        }
        total = run;
    }
    const uint32_t incl = wave_incl_scan_dpp(total);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t wadd = 0;
    for (int w = 0; w < wv; ++w) wadd += wsum[w];
    const uint32_t dbase = wadd + incl - total;
    digit_base[t] = dbase;
    // exclusive global base of digit t = scan of this pass's 256 digit totals: cheaper here than a separate launch
    const uint32_t hincl = wave_incl_scan_dpp(hcount);
    __syncthreads();                     // wsum is reused
    if (lane == 63) wsum[wv] = hincl;
    __syncthreads();
    uint32_t hadd = 0;
    for (int w = 0; w < wv; ++w) hadd += wsum[w];
    const uint32_t gdigit_base = hadd + hincl - hcount;
    if (SCANNED) {
        gbase[t] = (int64_t)gdigit_base + (int64_t)before - (int64_t)dbase;
        __syncthreads();
    } else {
        // ---- publish, look back, publish (one digit per thread)
        uint64_t* my = a.status + (size_t)tile * RS_RADIX + t;
        uint64_t prefix = 0;
        if (tile == 0) {
            __hip_atomic_store(my, RS_FLAG_GLOBAL | (uint64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(my, RS_FLAG_LOCAL | (uint64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // Batched look-back: RS_WINDOW predecessor words are requested at once (independent loads, one
            // memory latency), then consumed nearest-first.  With every resident tile starting together the
            // serial walk costs ~sqrt(2*tiles) dependent round trips; the window divides that by RS_WINDOW.
            int64_t j = (int64_t)tile - 1;
            bool found = false;
            uint32_t spins = 0;
            while (!found) {
                uint64_t w[WINDOW];
#pragma unroll
                for (int q = 0; q < WINDOW; ++q) {
                    const int64_t jj = j - q;
                    w[q] = jj >= 0 ? __hip_atomic_load(a.status + (size_t)jj * RS_RADIX + t, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT)
                                   : RS_FLAG_GLOBAL;              // virtual tile -1: inclusive count 0
                }
                int used = 0;
#pragma unroll
                for (int q = 0; q < WINDOW; ++q) {
                    if (!found && used == q) {
                        const uint64_t f = w[q] >> 62;
                        if (f != 0ull) {
                            prefix += w[q] & RS_COUNT_MASK;
                            ++used;
                            found = (f != 1ull);
                        }
                    }
                }
                j -= used;
                if (!found && used == 0) {                       // nearest predecessor not published yet
                    if (++spins > g_rs_spin_limit) {
                        atomicExch(a.error + a.pass, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __hip_atomic_store(my, RS_FLAG_GLOBAL | (prefix + (uint64_t)total), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        gbase[t] = (int64_t)gdigit_base + (int64_t)prefix - (int64_t)dbase;
        __syncthreads();
    }

    RS_STAMP(3);
    // ---- lay the tile out digit by digit in LDS (stable), then stream each run to its final place
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t e = (uint32_t)wv * (ITEMS * WAVE) + (uint32_t)i * WAVE + lane;
        const uint32_t d = (key[i] >> a.shift) & 0xFFu;
        rank[i] = e < tile_n ? digit_base[d] + wave_hist[wv][d] + rank[i] : 0xFFFFFFFFu;      // slot inside the tile
    }
    rs_exchange<ITEMS>(sbuf, rank, key, t);
    __syncthreads();
    rs_exchange<ITEMS>(sbuf, rank, val, t);
    RS_STAMP(4);
    int64_t g[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t p = (uint32_t)i * RS_THREADS + t;
        g[i] = p < tile_n ? gbase[(key[i] >> a.shift) & 0xFFu] + (int64_t)p : -1;
        if (g[i] >= 0) {
            a.kout[g[i]] = key[i];
            a.vout[g[i]] = val[i];
        }
    }
    RS_STAMP(5);
    RS_DRAIN();
    RS_STAMP(6);
    if (a.aux_out) {
        // the gather through the sorted values: every load of the thread in flight before the first store; a culled
        // Gaussian (key of all ones) has the empty rectangle by construction and is not fetched.  (At 2 M keys the
        // gather costs ~25 us -- 1.5 M fetches of a 128-byte line for 8 bytes each, at the Infinity Cache's ~8.6 TB/s
        // for random rows -- against ~11 us for a plain pass.)
        const uint2* __restrict__ ain = a.aux_in;
        uint2* __restrict__ aout = a.aux_out;
        uint2 r[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            r[i] = (g[i] >= 0 && !(a.aux_skip_ones && key[i] == 0xFFFFFFFFu)) ? ain[val[i]] : make_uint2(0u, 0u);
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if (g[i] >= 0) aout[g[i]] = r[i];
    }
    RS_DRAIN();
    RS_STAMP(7);
}

// Sorts n pairs on key bits [0, bits).  With `n_dev` the live pair count is read on the device
// (min(n, *n_dev)) and n is only the capacity that sizes grids and scratch: no host-side size needed.
// `ka`/`va` hold the input; the passes ping-pong between (ka, va) and (kb, vb).  The sorted pairs end in (kb, vb) when
// the number of passes is odd and in (ka, va) when it is even (radix_result_in_b tells which).
bool radix_result_in_b(int bits) { return (rs_passes(bits) & 1) != 0; }

// RADIX_ERROR_WORDS device words (one per pass) that a timed-out look-back spin sets to 1; any non-zero word = the
// sort's output is invalid (checked by the kernels that consume it and by the caller at its next sync point)
const uint32_t* radix_error_flag(void* temp, uint64_t n, int bits) { return rs_carve(temp, n ? n : 1, bits).error; }

// The region of `temp` that must be zero when the sort starts (histograms, tickets, error flag, status words / group
// sums); a kernel that runs right before the sort can clear it with grid_zero() and pass temp_zeroed = true.
void radix_zero_region(void* temp, uint64_t n, int bits, uint32_t** ptr, size_t* words) {
    const RsTemp t = rs_carve(temp, n ? n : 1, bits);
    *ptr = t.hist;
    *words = t.zero_bytes / 4;
}

// does a sort of n pairs read a global digit histogram (one-sweep path) -- i.e. is it worth counting one while the keys are produced?
bool radix_wants_hist(uint64_t n) { return n > 0 && !rs_scanned(n); }

template <int ITEMS, bool SCANNED>
static void rs_launch_pass(const RsPassArgs& a, uint32_t tiles, hipStream_t s) {
    if (g_opt_radix_ballot_rank) hipLaunchKernelGGL((rs_pass_kernel<ITEMS, SCANNED, true>), dim3(tiles), dim3(RS_THREADS), 0, s, a);
    else hipLaunchKernelGGL((rs_pass_kernel<ITEMS, SCANNED, false>), dim3(tiles), dim3(RS_THREADS), 0, s, a);
}

int radix_sort_pairs(uint32_t* ka, uint32_t* va, uint32_t* kb, uint32_t* vb, uint64_t n, int bits, void* temp,
                     hipStream_t s, const uint32_t* n_dev, bool temp_zeroed, const uint2* aux_in, uint2* aux_out,
                     const uint32_t* ext_hist, bool aux_empty_for_ones) {
    if (n == 0) return 0;
    if (n >= (1ull << 32)) { set_error("radix sort: more than 2^32-1 pairs"); return 1; }
    const int npasses = rs_passes(bits);
    if (npasses > RS_MAX_PASSES) { set_error("radix sort: more than 32 key bits"); return 1; }
    const RsTemp t = rs_carve(temp, n, bits);
    if (!temp_zeroed) MGS_HIP(zero_fill(t.hist, t.zero_bytes, s));
    const uint32_t tiles = rs_tiles(n);
    const uint32_t hist_blocks = min(tiles, 1024u);
    const bool scanned = rs_scanned(n);
    const int items = rs_tile_items(n, scanned);
    const RsTree tree = rs_tree(tiles);
    RsTreeArgs ta;
    ta.levels = tree.levels;
    for (int l = 0; l < RS_MAX_LEVELS; ++l) ta.off[l] = l < tree.levels ? tree.off[l] : 0u;
    ta.top_rows = tree.rows[tree.levels - 1];
    if (!scanned && !ext_hist)
        hipLaunchKernelGGL(rs_hist_kernel, dim3(hist_blocks), dim3(RS_THREADS), 0, s, ka, (uint32_t)n, n_dev, npasses, t.hist);
    const uint32_t* ghist = (!scanned && ext_hist) ? ext_hist : t.hist;
    uint32_t *kin = ka, *vin = va, *kout = kb, *vout = vb;
    for (int p = 0; p < npasses; ++p) {
        RsPassArgs a;
        a.kin = kin; a.kout = kout; a.vin = vin; a.vout = vout;
        a.n = (uint32_t)n; a.n_dev = n_dev; a.shift = 8 * p;
        a.hist = ghist + p * RS_RADIX;
        a.status = t.status + (size_t)p * tiles * RS_RADIX;
        // <= one workgroup per CU: the whole grid is co-resident whatever the dispatch order, so block ids are safe
        a.ticket = tiles <= 256u ? nullptr : t.tickets + p;
        a.error = t.error;
        a.pass = p;
        a.counts = t.counts; a.sums = t.gsum + (size_t)p * tree.sum_rows * RS_RADIX;
        a.tree = ta;
        const bool last = p == npasses - 1;
        a.aux_in = last ? aux_in : nullptr; a.aux_out = last ? aux_out : nullptr;
        a.aux_skip_ones = aux_empty_for_ones ? 1 : 0;
        if (scanned) {
            uint32_t* gs = t.gsum + (size_t)p * tree.sum_rows * RS_RADIX;
            if (items == RS_ITEMS_WIDE) {
                rs_launch_tile_hist<RS_ITEMS_WIDE>(kin, (uint32_t)n, n_dev, a.shift, tiles, ta, t.counts, gs, s);
                rs_launch_pass<RS_ITEMS_WIDE, true>(a, tiles, s);
            } else {
                rs_launch_tile_hist<RS_ITEMS>(kin, (uint32_t)n, n_dev, a.shift, tiles, ta, t.counts, gs, s);
                rs_launch_pass<RS_ITEMS, true>(a, tiles, s);
            }
        } else if (items == RS_ITEMS_SMALL)
            rs_launch_pass<RS_ITEMS_SMALL, false>(a, tiles, s);
        else if (items == RS_ITEMS_MID)
            rs_launch_pass<RS_ITEMS_MID, false>(a, tiles, s);
        else
            rs_launch_pass<RS_ITEMS, false>(a, tiles, s);
        uint32_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    MGS_HIP(hipGetLastError());
    return 0;
}

int set_radix_spin_limit(uint32_t limit) {
    const uint32_t v = limit == 0xFFFFFFFFu ? RS_SPIN_LIMIT : limit;
    MGS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rs_spin_limit), &v, sizeof(v)));
    return 0;
}

}  // namespace mgs
