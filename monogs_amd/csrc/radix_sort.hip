// Stable LSD radix sort of (uint32 key, uint32 value) pairs, hand-written for gfx950 (wave64).
//
// Used twice per forward (binning.hip): the P Gaussians by their 32 depth bits, then the R
// (tile id, Gaussian index) instances by tile id.  Replaces upstream's cub::DeviceRadixSort call
// (SURVEY.md section 2.1 K4).
//
// One 8- or 9-bit digit per pass.  A workgroup (256 threads = 4 waves) takes a tile of 1024-4096 pairs, ranks them stably,
// lays the tile out digit by digit in LDS and writes each digit's run to its final place (coalesced runs instead of a
// 256-way scatter).  Two ways of knowing where a tile's runs go, bit-identical results, chosen by size (rs_scanned()):
//
// "One sweep" (<= 512 k pairs: the sorts of a SLAM-sized map at VGA; fewest launches):
//   * ONE histogram kernel reads the keys once and counts every digit of every pass (each pass kernel
//     scans its 256 counts into exclusive global bases itself);
//   * per pass ONE kernel: the tile publishes its 256 digit counts and obtains the sum of the counts of all EARLIER
//     tiles by a two-level gather (its group of 16, then the group sums).
// "Counted tiles" (> 512 k pairs; rs_scanned): per pass rs_tile_hist_kernel writes the digit counts of every tile,
//   counts[tile][digit], and adds them into the rows of the tile's ancestors in a tree of fan-out 4 over the tiles; the
//   scatter kernel sums the siblings before each of its ancestors (<= 3 rows per level, <= 8 at the top, all requested
//   together before the keys: one round trip hidden behind the key loads) -- no waiting between workgroups, and no scan
//   launch.  With hundreds of co-resident tiles the status traffic and round trips of waiting dominated the pass.
// The final pass can also gather an auxiliary array through the sorted values (aux_out[pos] = aux_in[value]).
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
// Depth keys (radix_sort_depth): the float bits of a view-space depth > 0.2 (the rasteriser's near cull), all ones for a
// culled Gaussian.  key - bits(0.2f) is < 2^27 - 1 for every depth below 13 107 units, so the sort runs THREE passes of 9
// bits on it instead of four of 8 on the raw bits (culled keys keep their all-ones pattern and stay last).  The pass that
// reads the keys first raises a device flag if some visible depth lies beyond; only then does a fourth pass (the top five
// bits) do anything -- it is always launched (its workgroups return at once when the flag is clear), so the result is
// exact for every input with no host decision in between; the third pass writes the final arrays itself when the flag is
// clear.  The rectangle of every Gaussian travels with the pair as a 4-byte payload on the counted-tiles path (packed
// x0 | y0 << 8 | w << 16 | h << 24 by preprocess when the tile grid is at most 255 x 255), because gathering it through
// the sorted indices at the end costs a 128-byte line per Gaussian: 25 us of the 37 us last pass at 2 M Gaussians.  The
// final pass unpacks it and writes no keys (nobody reads sorted depth keys); the first pass takes value = index instead
// of reading an iota array.
//
// One-sweep details:
// Inter-workgroup protocol (MI355X_MICROARCH.md "Workgroup dispatch ... visibility", form R2): every
// status word is a self-describing 8-byte granule {flag:2, count:62} written by ONE agent-scope
// relaxed atomic store and polled with agent-scope relaxed atomic loads; no other data crosses
// workgroups, so no fence is needed.  The tile ids are handed out by an atomic ticket, so a tile only ever waits for
// tiles that already started (placement-independent forward progress; block ids only for <= 256 tiles on a device the
// caller declares exclusive, see rs_run); every spin is bounded and raises an error flag instead of hanging.
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
// Above this many pairs the counted-tiles path takes over.  Round 3, after both paths changed (LDS-atomic ranking, the
// tree of counts instead of a scan launch, the all-gather instead of the look-back walk), whole sort in us, one sweep /
// counted tiles: tile sort 100 k: 29.1 / 32.6, 200 k: 33.5 / 33.1, 300 k: 35.7 / 33.8, 415 k: 38.3 / 34.5, 640 k: 50.0 / 37.5;
// depth sort 200 k: 59.6 / 62.1, 300 k: 63.2 / 63.8, 415 k: 67.0 / 65.8, 640 k: 92.4 / 71.2 (round 2 had it between 512 k and 1 M).
// Inside a hipGraph replay every kernel node costs ~4.8 us whatever it does, and the counted path has twice the launches:
// the captured TUM-like run (39 k Gaussians, 415 k instances) tracks at 4171 it/s with the one-sweep tile sort against
// 4088 with counted tiles, the Replica-like one (103 k, ~1 M instances) at 1653 against 1724.
constexpr uint64_t RS_ONE_SWEEP_MAX = 512ull * 1024;
static inline int rs_items(uint64_t n) {
    return n <= 192ull * 1024 ? RS_ITEMS_SMALL : (n <= RS_ONE_SWEEP_MAX ? RS_ITEMS_MID : RS_ITEMS);
}
constexpr int RS_MAX_RADIX = 512;
constexpr int RS_MAX_PASSES = 4;
constexpr uint64_t RS_FLAG_LOCAL = 1ull << 62;     // count of this tile only
constexpr uint64_t RS_FLAG_GLOBAL = 2ull << 62;    // inclusive count over tiles 0..this
constexpr uint64_t RS_COUNT_MASK = (1ull << 62) - 1;
constexpr uint32_t RS_SPIN_LIMIT = 1u << 22;
// the bound actually used by the look-back spin: a device word so that a test can shrink it (mgs_debug_set_radix_spin_limit)
__device__ uint32_t g_rs_spin_limit = RS_SPIN_LIMIT;
// (The first one-sweep version looked back 16 predecessors at a time for a tile with an inclusive count.  Round 2 tried
//  wider windows -- 64 words for the 40-tile depth sort of a 40 k map, 32 for the 200-tile tile sort: 12.5-14 us per pass
//  instead of 9.2-10.7, 15.6-16.3 instead of 14.1-14.3 -- and narrower ones, inside the noise.  Round 3 replaced the walk
//  by the two-level all-gather in rs_pass_kernel: depth sort of 40 k keys 55.5 -> 45.8 us, tile sort of 415 k instances
//  51.1 -> 38.1 us.)
constexpr uint32_t RS_GROUP = 16;                  // tiles per group of the one-sweep all-gather

// depth keys: see the header.  key - RS_DEPTH_SUB < RS_DEPTH_NARROW for every visible depth below 13 107.2
constexpr uint32_t RS_DEPTH_SUB = 0x3E4CCCCDu;             // float bits of 0.2f (preprocess culls depth <= 0.2)
constexpr uint32_t RS_DEPTH_NARROW = (1u << 27) - 1u;
// digit source of a key: culled keys (all ones) keep their pattern, so they sort behind every visible key in every mode
__device__ __forceinline__ uint32_t rs_xf(uint32_t k, uint32_t sub) { return k == 0xFFFFFFFFu ? k : k - sub; }

// large sorts take the counted-tiles path; a test knob forces either one (no environment lookups on the launch path)
int g_opt_radix_scanned = -1;       // mgs_debug_set_option("radix_scanned", -1 | 0 | 1): -1 = by size
int g_opt_radix_ballot_rank = 0;    // mgs_debug_set_option("radix_ballot_rank", 1): rank with ballots instead of LDS atomics
int g_opt_radix_xcd_band = 1;       // mgs_debug_set_option("radix_xcd_band", 0): counted tiles in block-id order instead of one band per XCD
constexpr uint64_t RS_SCANNED_MIN = 64ull * 1024;       // the knob can force counted tiles down to here
static inline bool rs_scanned(uint64_t n) {
    if (g_opt_radix_scanned >= 0) return g_opt_radix_scanned == 1 && n > RS_SCANNED_MIN;
    return n > RS_ONE_SWEEP_MAX;
}
// Pairs per thread.  On the counted-tiles path a tile lives ~10-20 us and the kernel ends when the last one does: if the
// tiles do not all fit on the chip at once, the stragglers start when the first ones retire and the pass takes two tile
// lives (tile sort at C5, 5.27 M pairs = 1287 tiles of 4096 against 1024 resident: tile starts at 0 and at 14 us).  4096-pair
// tiles: 84-92 VGPRs + 24 KB of LDS = 4-5 workgroups per CU; 3072-pair tiles: 72 VGPRs + 20 KB = 7 per CU (1792), and 1716
// tiles at C5 (82.7 us against 85.9; 2048-pair tiles lose it again to the doubled count tables, 104.6 us).
int g_opt_radix_tile_items = 0;     // mgs_debug_set_option("radix_tile_items", 8 | 12 | 16): pairs per thread on the counted-tiles path (0: by size)
static inline int rs_tile_items(uint64_t n, bool scanned) {
    if (!scanned) return rs_items(n);
    if (g_opt_radix_tile_items == RS_ITEMS_MID || g_opt_radix_tile_items == RS_ITEMS_WIDE || g_opt_radix_tile_items == RS_ITEMS)
        return g_opt_radix_tile_items;
    return n > 1024ull * RS_THREADS * RS_ITEMS ? RS_ITEMS_WIDE : RS_ITEMS;
}
static inline uint32_t rs_tiles(uint64_t n, bool scanned) {
    const uint64_t tile = (uint64_t)RS_THREADS * rs_tile_items(n, scanned);
    return (uint32_t)((n + tile - 1) / tile);
}
static inline uint32_t rs_tiles(uint64_t n) { return rs_tiles(n, rs_scanned(n)); }

// ---- what a sort does, pass by pass (host side)
struct RsPlan {
    int npasses;
    int shift[RS_MAX_PASSES];
    int db[RS_MAX_PASSES];         // 8 or 9: the pass ranks 256 or 512 digits
    int radix;                     // stride of the per-pass tables: the largest radix of the plan
    uint32_t sub;
    bool depth;                    // pass 2 is final unless the wide flag is up, pass 3 runs only then
};
static inline RsPlan rs_plan_plain(int bits) {
    RsPlan pl;
    pl.npasses = (bits + 7) / 8;
    for (int p = 0; p < RS_MAX_PASSES; ++p) { pl.shift[p] = 8 * p; pl.db[p] = 8; }
    pl.radix = 256; pl.sub = 0u; pl.depth = false;
    return pl;
}
static inline RsPlan rs_plan_depth() {
    RsPlan pl;
    pl.npasses = 4;
    for (int p = 0; p < 3; ++p) { pl.shift[p] = 9 * p; pl.db[p] = 9; }
    pl.shift[3] = 27; pl.db[3] = 8;            // five significant bits
    pl.radix = 512; pl.sub = RS_DEPTH_SUB; pl.depth = true;
    return pl;
}

// The narrow passes pay on the counted-tiles path (one launch pair and one trip over the data less: 114.5 -> 101.7 us at 2 M
// keys).  One sweep: a 512-digit pass publishes and looks back at twice the status words and the histogram kernel counts
// into 2048 LDS counters per wave -- three of them + the empty fourth took what four 256-digit passes take (51.8 against
// 52.6 us at 40 k keys), so small maps keep the plain plan (with the depth sort's outputs: no iota read, no keys written).
static inline RsPlan rs_plan_for_depth(uint64_t n) { return rs_scanned(n) ? rs_plan_depth() : rs_plan_plain(32); }

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

// temp layout: [hist: RS_MAX_PASSES * radix u32][tickets: RS_MAX_PASSES u32][error: RS_MAX_PASSES u32][wide flag][pad]
//   one sweep:     [status: passes * tiles * radix u64]
//   counted tiles: [sums: passes * (rows of tree levels >= 1) * radix u32][counts: tiles * radix u32 (rewritten by
//                  every pass, never cleared)]
// one sweep: status rows of a pass = one per tile + one per group of RS_GROUP tiles
static inline size_t rs_status_rows(uint32_t tiles) { return (size_t)tiles + (tiles + RS_GROUP - 1) / RS_GROUP; }
struct RsTemp {
    uint32_t* hist;
    uint32_t* tickets;
    uint32_t* error;
    uint32_t* wide;
    uint64_t* status;
    uint32_t* sums;
    uint32_t* counts;
    size_t zero_bytes;     // everything from `hist` that must be zero before the sort
    size_t bytes;
};
static RsTemp rs_carve(void* temp, uint64_t n, const RsPlan& pl, bool scanned) {
    char* p = (char*)align_up((size_t)temp, 256);
    RsTemp t;
    t.hist = (uint32_t*)p;
    t.tickets = t.hist + RS_MAX_PASSES * pl.radix;
    t.error = t.tickets + RS_MAX_PASSES;      // one word per pass
    t.wide = t.error + RS_MAX_PASSES;
    char* q = (char*)align_up((size_t)(t.wide + 1), 256);
    t.status = (uint64_t*)q;
    t.sums = (uint32_t*)q;
    const uint32_t tiles = rs_tiles(n, scanned);
    if (scanned) {
        const size_t sums_bytes = (size_t)pl.npasses * rs_tree(tiles).sum_rows * pl.radix * sizeof(uint32_t);
        t.counts = (uint32_t*)(q + sums_bytes);
        t.zero_bytes = (size_t)(q - p) + sums_bytes;
        t.bytes = t.zero_bytes + (size_t)tiles * pl.radix * sizeof(uint32_t);
    } else {
        t.counts = nullptr;
        t.zero_bytes = (size_t)(q - p) + (size_t)pl.npasses * rs_status_rows(tiles) * pl.radix * sizeof(uint64_t);
        t.bytes = t.zero_bytes;
    }
    return t;
}
static size_t rs_temp_bytes(uint64_t n, const RsPlan& pl) {
    if (n == 0) return 256;
    // either path may be forced on a buffer sized earlier (test knob): size for the larger one
    const size_t a = rs_carve(nullptr, n, pl, false).bytes;
    const size_t b = n > RS_SCANNED_MIN ? rs_carve(nullptr, n, pl, true).bytes : 0;
    return (a > b ? a : b) + 512;
}
size_t radix_temp_bytes(uint64_t n, int bits) { return rs_temp_bytes(n, rs_plan_plain(bits)); }
size_t radix_depth_temp_bytes(uint64_t n) {
    const size_t a = rs_temp_bytes(n, rs_plan_depth()), b = rs_temp_bytes(n, rs_plan_plain(32));
    return a > b ? a : b;
}

// ------------------------------------------------------------------------------------------------

// One sweep: digit counts of every pass in one read of the keys.  Each wave keeps a private copy of the
// histograms in LDS (4x fewer same-address LDS atomics; the tile-id digits are low-entropy).  Depth keys: also raises
// the wide flag when a visible key lies beyond the three narrow passes.
struct RsHistArgs {
    int npasses, radix;
    int shift[RS_MAX_PASSES], mask[RS_MAX_PASSES];
    uint32_t sub;
    uint32_t* wide;           // depth keys only
};
__global__ void __launch_bounds__(RS_THREADS) rs_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                             const uint32_t* __restrict__ n_dev, RsHistArgs h,
                                                             uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[RS_WAVES][RS_MAX_PASSES * RS_MAX_RADIX];        // 32 KB
    if (n_dev) n = min(n, n_dev[0]);          // capacity mode: the live count is on the device
    const int words = h.npasses * h.radix;
    for (int w = 0; w < RS_WAVES; ++w)
        for (int i = threadIdx.x; i < words; i += RS_THREADS) lh[w][i] = 0;
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    bool far = false;
    for (uint32_t i = blockIdx.x * RS_THREADS + threadIdx.x; i < n; i += gridDim.x * RS_THREADS) {
        const uint32_t k = keys[i], x = rs_xf(k, h.sub);
        far |= k != 0xFFFFFFFFu && x >= RS_DEPTH_NARROW;
        for (int p = 0; p < h.npasses; ++p) atomicAdd(&lh[wv][p * h.radix + ((x >> h.shift[p]) & h.mask[p])], 1u);
    }
    if (h.wide && far) atomicOr(h.wide, 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += RS_THREADS) {
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) c += lh[w][i];
        if (c) atomicAdd(hist + i, c);
    }
}

// ---- counted tiles, step 1: digit counts of every tile for one pass, counts[tile][digit] (one coalesced row per tile)
// and the levels >= 1 of the tree over them.  From 1024 tiles up a workgroup of 1024 threads takes the four tiles of ONE
// level-1 node (256 threads each), so that level is a plain store; the higher levels are memory-side adds of 256
// contiguous bytes per wave-instruction into rows zeroed with the sort's scratch.  (One 256-thread workgroup per tile
// adding into every level: 27 k wave-atomics per pass of the C5 tile sort = 4.7 us of a 14 us kernel at the memory side's
// ~1.3 TB/s.  Below 1024 tiles that is the cheaper form: at 489 tiles, 123 workgroups of four left half the CUs idle and
// the depth sort 10 us slower.)
// The LDS counters are private to a wave AND to a few lanes of it: the 64 lanes of a wave-instruction that hit one
// counter are applied one after the other, and the top byte of a float depth takes ~5 values (5.5 us of counting
// against 1.7 us for a uniform byte before the split).
// The scatter kernel turns the tree into its offsets itself: there is no scan launch (round 3: the per-digit row scan
// over counts[digit][tile] was a launch of 5-6 us per pass, 6 passes per forward at C5).
struct RsTreeArgs {
    int levels;
    uint32_t off[RS_MAX_LEVELS];
    uint32_t top_rows;
};
struct RsTileHistArgs {
    const uint32_t* keys;
    uint32_t n;
    const uint32_t* n_dev;
    int shift;
    uint32_t sub;
    uint32_t tiles;
    RsTreeArgs tr;
    uint32_t* counts;
    uint32_t* sums;
    uint32_t* wide;           // depth keys: the wide flag
    int detect;               // ... this launch reads the keys first: raise it for a visible key beyond the narrow passes
    int only_if_wide;         // ... this launch belongs to the fourth pass
};
// TPW = tiles per workgroup (1 or 4), DB = digit bits (8 or 9)
template <int ITEMS, int TPW, int DB>
__global__ void __launch_bounds__(RS_THREADS * TPW) rs_tile_hist_kernel(RsTileHistArgs a) {
    static_assert(TPW == 1 || TPW == (1 << RS_FAN_LOG), "a workgroup counts one tile or one level-1 node");
    constexpr int RADIX = 1 << DB, DPT = RADIX / RS_THREADS, COPIES = 16 / DPT;     // 16 KB of counters per tile
    constexpr uint32_t TILE_PAIRS = RS_THREADS * ITEMS;
    __shared__ uint32_t wh[TPW][COPIES][RADIX];
    const int t = threadIdx.x & (RS_THREADS - 1), q = threadIdx.x >> 8, wv = t >> 6, lane = t & 63;
    if (a.only_if_wide && a.wide[0] == 0u) return;
    RS_HSTAMP(0);
    const uint32_t n_live = a.n_dev ? min(a.n, a.n_dev[0]) : a.n;
    // capacity mode launches tiles for the capacity (1.5 x the live count by default).  A workgroup whose tiles are all
    // dead has nothing to write: nobody reads the count row of a dead tile (a tile adds up rows of EARLIER siblings only),
    // and the tree rows it would add zeros to were cleared with the sort's scratch.  (With a single level -- at most 8 tiles,
    // never the case on this path -- the count rows ARE the top level, which is read in full.)
    if (a.tr.levels > 1 && (uint64_t)blockIdx.x * TPW * TILE_PAIRS >= n_live) return;
    const uint32_t tile = blockIdx.x * TPW + q, tile_start = tile * TILE_PAIRS;
    const uint32_t tile_n = tile_start < n_live ? min(TILE_PAIRS, n_live - tile_start) : 0u;     // 0: dead (capacity mode) or past the end
    uint32_t k[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t e = (uint32_t)i * RS_THREADS + t;
        k[i] = e < tile_n ? a.keys[tile_start + e] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < COPIES; ++i)
#pragma unroll
        for (int j = 0; j < DPT; ++j) wh[q][i][t * DPT + j] = 0;
    __syncthreads();
    RS_DRAIN();
    RS_HSTAMP(1);
    uint32_t* mine = wh[q][wv * (COPIES / RS_WAVES) + (lane & (COPIES / RS_WAVES - 1))];
    bool far = false;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t x = rs_xf(k[i], a.sub);
        far |= k[i] != 0xFFFFFFFFu && x >= RS_DEPTH_NARROW;
        if ((uint32_t)i * RS_THREADS + t < tile_n) atomicAdd(&mine[(x >> a.shift) & (RADIX - 1)], 1u);
    }
    if (a.detect && far) atomicOr(a.wide, 1u);
    __syncthreads();
    RS_HSTAMP(2);
    uint32_t c[DPT];
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        c[j] = 0;
#pragma unroll
        for (int i = 0; i < COPIES; ++i) c[j] += wh[q][i][t * DPT + j];
        if (tile < a.tiles) a.counts[(size_t)tile * RADIX + t * DPT + j] = c[j];       // dead tiles: zeros
    }
    if (TPW == 1) {
#pragma unroll
        for (int j = 0; j < DPT; ++j)
            if (c[j])
                for (int l = 1; l < a.tr.levels; ++l)
                    atomicAdd(a.sums + ((size_t)a.tr.off[l] + (tile >> (RS_FAN_LOG * l))) * RADIX + t * DPT + j, c[j]);
    } else if (a.tr.levels > 1) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < DPT; ++j) wh[q][0][t * DPT + j] = c[j];
        __syncthreads();
        if (q == 0) {
#pragma unroll
            for (int j = 0; j < DPT; ++j) {
                uint32_t s1 = 0;
#pragma unroll
                for (int m = 0; m < TPW; ++m) s1 += wh[m][0][t * DPT + j];
                a.sums[((size_t)a.tr.off[1] + blockIdx.x) * RADIX + t * DPT + j] = s1;         // level 1: this workgroup's own row
                if (s1)
                    for (int l = 2; l < a.tr.levels; ++l)
                        atomicAdd(a.sums + ((size_t)a.tr.off[l] + (blockIdx.x >> (RS_FAN_LOG * (l - 1)))) * RADIX + t * DPT + j, s1);
            }
        }
    }
    RS_DRAIN();
    RS_HSTAMP(3);
}
template <int ITEMS, int DB>
static void rs_launch_tile_hist(const RsTileHistArgs& a, hipStream_t s) {
    if (a.tiles >= 1024u)
        hipLaunchKernelGGL((rs_tile_hist_kernel<ITEMS, 4, DB>), dim3((a.tiles + 3) / 4), dim3(RS_THREADS * 4), 0, s, a);
    else
        hipLaunchKernelGGL((rs_tile_hist_kernel<ITEMS, 1, DB>), dim3(a.tiles), dim3(RS_THREADS), 0, s, a);
}

struct RsPassArgs {
    const uint32_t* kin;
    const uint32_t* vin;      // NULL: value = index of the pair (first pass of the depth sort: the VIDX kernels)
    const uint32_t* pin;      // payload (PAYLOAD kernels)
    uint32_t* kout;           // outputs of a pass that is not the final one
    uint32_t* vout;
    uint32_t* pout;
    uint32_t* vfinal;         // outputs of the final pass: values, keys (optional), and the auxiliary array (optional):
    uint32_t* kfinal;
    const uint2* aux_in;      //   aux_out[position] = aux_in[value] (a gather), or with a payload = its unpacked rectangle
    uint2* aux_out;
    int aux_skip_ones;        // a pair whose key is 0xFFFFFFFF gets aux (0, 0) without a fetch (depth sort: culled Gaussians)
    uint32_t n;               // number of pairs (capacity when n_dev is set)
    const uint32_t* n_dev;    // optional: live count on the device (<= n after clamping)
    int shift;
    uint32_t sub;
    const uint32_t* hist;     // one sweep: [radix] global count of each digit for this pass
    uint64_t* status;         // one sweep: [tiles + groups][radix]
    uint32_t tiles;
    uint32_t* ticket;
    uint32_t* error;          // [RS_MAX_PASSES] one word per pass: pass p raises error[p] when a look-back spin times out
    int pass;
    int xcd_band;             // counted tiles: tile = band mapping of the block id (see rs_pass_kernel)
    int last;                 // cond 0: this pass is the final one
    int cond;                 // 0: plain; 1: final unless the wide flag is up (depth, third pass); 2: runs only if it is (fourth)
    const uint32_t* wide;
    uint2* ranges;            // final pass of a sort whose keys are SMALL integers (tile ids): [key] receives {first, last + 1}
                              // position of the key in the sorted order (min / max atomics on words preset to {~0, 0}): the
                              // per-tile ranges of the rasteriser, which used to be a launch of their own (ranges_kernel)
    const uint32_t* counts;   // counted tiles: [tiles][radix] digit counts of every tile for this pass
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

// exclusive global base of a thread's digits = block-wide scan of this pass's digit totals (cheaper inside the scatter
// kernel than a launch of its own)
template <int DPT>
__device__ __forceinline__ void rs_digit_bases(const uint32_t (&hcount)[DPT], uint32_t (&gdigit_base)[DPT], uint32_t* wsum, int lane, int wv) {
    uint32_t hsum = 0;
#pragma unroll
    for (int j = 0; j < DPT; ++j) hsum += hcount[j];
    const uint32_t hincl = wave_incl_scan_dpp(hsum);
    __syncthreads();                     // wsum is reused
    if (lane == 63) wsum[wv] = hincl;
    __syncthreads();
    uint32_t hadd = 0;
    for (int w = 0; w < wv; ++w) hadd += wsum[w];
    uint32_t run = hadd + hincl - hsum;
#pragma unroll
    for (int j = 0; j < DPT; ++j) { gdigit_base[j] = run; run += hcount[j]; }
}

// one-sweep gather: adds the counts of the status rows [0, cnt) (stride `stride` words, this thread's digit) into `pre`.
// RS_GROUP rows per batch of loads; spins (bounded) until every word of the batch is published.  Returns false if it
// gave up.
__device__ __forceinline__ bool rs_gather(const uint64_t* row, size_t stride, uint32_t cnt, uint32_t& pre, uint32_t& spins,
                                          uint32_t* err) {
    constexpr uint32_t G = RS_GROUP;
    for (uint32_t r0 = 0; r0 < cnt; r0 += G) {
        uint64_t w[G];
        bool ready;
        do {
#pragma unroll
            for (uint32_t q = 0; q < G; ++q) {
                const uint32_t r = r0 + q;
                w[q] = r < cnt ? __hip_atomic_load(row + (size_t)r * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                              : RS_FLAG_GLOBAL;          // nothing there: published, count 0
            }
            ready = true;
#pragma unroll
            for (uint32_t q = 0; q < G; ++q) ready &= (w[q] >> 62) != 0ull;
            if (!ready) {
                if (++spins > g_rs_spin_limit) { atomicExch(err, 1u); return false; }
                __builtin_amdgcn_s_sleep(1);
            }
        } while (!ready);
#pragma unroll
        for (uint32_t q = 0; q < G; ++q) pre += (uint32_t)(w[q] & RS_COUNT_MASK);
    }
    return true;
}

// DB:              digit bits, 8 or 9: 256 or 512 digits, one or two (adjacent) per thread.
// SCANNED = false: one sweep -- the tile publishes its digit counts and finds the sum over earlier tiles by decoupled
//                  look-back (fewest launches: right for the small sorts of SLAM-sized maps).
// SCANNED = true:  counted tiles -- the digit counts of every tile and the tree over them were written beforehand by
//                  rs_tile_hist_kernel; the tile adds up what lies before it.
// PAYLOAD:         a second value travels with the pair.
// BALLOT:          rank with one ballot per digit bit and item instead of one returning LDS atomic (reference for the tests).
// VIDX:            value = index of the pair; vin is not read (first pass of the depth sort).
template <int ITEMS, int DB, bool SCANNED, bool PAYLOAD, bool BALLOT, bool VIDX>
__global__ void __launch_bounds__(RS_THREADS, (SCANNED && ITEMS == RS_ITEMS_WIDE && DB == 8 && !PAYLOAD) ? 7 : 1) rs_pass_kernel(RsPassArgs a) {
    constexpr int TILE_PAIRS = RS_THREADS * ITEMS;
    constexpr int RADIX = 1 << DB, DPT = RADIX / RS_THREADS;
    constexpr int SBUF = (SCANNED && 2 * RS_WAVES * RADIX > TILE_PAIRS) ? 2 * RS_WAVES * RADIX : TILE_PAIRS;
    __shared__ uint32_t wave_hist[RS_WAVES][RADIX];
    __shared__ uint32_t digit_base[RADIX];
    __shared__ uint32_t gbase[RADIX];                    // global position of LDS slot 0 for each digit, modulo 2^32 (it may be
                                                         // "negative"; slot + gbase is a position < 2^32 again)
    __shared__ uint32_t sbuf[SBUF];                      // keys, values (and payload) take turns in ONE staging buffer: half the
                                                         // LDS of two, so more tiles are resident per CU (two barriers more)
    __shared__ uint32_t wsum[RS_WAVES];
    __shared__ uint32_t s_tile;

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const bool wide = a.wide ? a.wide[0] != 0u : false;
    if (a.cond == 2 && !wide) return;
    RS_STAMP(0);
    const bool final_pass = a.cond == 0 ? a.last != 0 : (a.cond == 1 ? !wide : true);
    if (!SCANNED) {
        // A timed-out look-back in an EARLIER pass left part of this pass's input unwritten: ranking it against the
        // histogram of the original keys could place pairs past the end of the buffers.  Such a pass does nothing (the
        // words of earlier passes cannot change while this one runs, so every tile takes the same decision), and the
        // consumers of the sort treat a raised flag as "no output" (duplicate_kernel, blend_forward_kernel).
        uint32_t failed = 0;
        for (int q = 0; q < a.pass; ++q) failed |= a.error[q];
        if (failed) return;
    }
    const uint32_t n_live = a.n_dev ? min(a.n, a.n_dev[0]) : a.n;
    if (t == 0) {
        uint32_t id = blockIdx.x;
        if (a.xcd_band) {
            // Workgroups are dealt round-robin to the 8 XCDs, each with an L2 of its own.  The runs that neighbouring tiles
            // write for one digit are adjacent in memory and short (tile sort at C5: 3072 pairs / 256 digits = 48 bytes of
            // keys per run, 32 bytes in a 512-digit depth pass), so with tile = block id every 128-byte line of the output
            // is written in pieces by several L2s.  One contiguous band of the LIVE tiles per XCD lets the pieces meet in
            // ONE L2 before the line leaves it: tile sort 83 -> 58 us, depth sort 102 -> 94 us at C5 (round 4).  (Capacity
            // mode launches tiles for the capacity: banding those would leave the XCDs of the last bands without work.)
            const uint32_t live = (n_live + (uint32_t)TILE_PAIRS - 1) / (uint32_t)TILE_PAIRS;
            const uint32_t q = live >> 3, r = live & 7u, x = id & 7u, j = id >> 3;
            id = j < q + (x < r ? 1u : 0u) ? x * q + min(x, r) + j : 0xFFFFFFFFu;      // (its XCD's band is handed out: nothing to do)
        }
        s_tile = (!SCANNED && a.ticket) ? atomicAdd(a.ticket, 1u) : id;   // a returning atomic is a ~2 us round trip
    }
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w)
#pragma unroll
        for (int j = 0; j < DPT; ++j) wave_hist[w][t * DPT + j] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile == 0xFFFFFFFFu) return;
    const uint32_t tile_start = tile * (uint32_t)TILE_PAIRS;
    if (tile_start >= n_live) return;        // capacity mode: tiles past the live count have nothing to do
                                             // (tickets are dense, so no live tile ever looks back at them)
    const uint32_t tile_n = min((uint32_t)TILE_PAIRS, n_live - tile_start);

    // ---- digits t * DPT + j: count over the whole input (hcount) and, counted tiles, over everything before this tile
    //      (before).  Requested first: needed only after the ranking, their round trips hide behind the key loads.
    //      Counted tiles: a row of RADIX counts is RADIX / 256 wave-instructions of 16 bytes per lane, so the rows are
    //      dealt out to the four waves -- per level the <= 3 siblings before this tile's ancestor (one per wave), at the
    //      top all <= 8 nodes (two per wave).  Every slot is loaded unconditionally (a row index clamped into the
    //      table, weight 0) so that all of them are in flight together: ONE round trip.
    uint32_t hcount[DPT], before[DPT];
#pragma unroll
    for (int j = 0; j < DPT; ++j) hcount[j] = before[j] = 0;
    constexpr int R4 = RADIX / 4;                         // uint4 per row
    uint4 lv[RS_MAX_LEVELS - 1][DPT], tv[RS_TOP_BATCH / RS_WAVES][DPT];
    uint32_t lw[RS_MAX_LEVELS - 1], tw[RS_TOP_BATCH / RS_WAVES], tb[RS_TOP_BATCH / RS_WAVES];
    const int levels = a.tree.levels;
    const uint32_t top_node = tile >> (RS_FAN_LOG * (levels - 1));
    const uint4* __restrict__ c4 = reinterpret_cast<const uint4*>(a.counts);
    const uint4* __restrict__ top4 = levels == 1 ? c4 : reinterpret_cast<const uint4*>(a.sums) + (size_t)a.tree.off[levels - 1] * R4;
    if (SCANNED) {
#pragma unroll
        for (int l = 0; l < RS_MAX_LEVELS - 1; ++l) {
            const bool have = l < levels - 1;                                   // wave-uniform
            const uint32_t node = tile >> (RS_FAN_LOG * l), first = node & ~3u, nb = node & 3u;
            const uint4* base = (l == 0 || !have) ? c4 : reinterpret_cast<const uint4*>(a.sums) + (size_t)a.tree.off[l] * R4;
            lw[l] = (have && (uint32_t)wv < nb) ? 1u : 0u;                     // wave w takes sibling w
#pragma unroll
            for (int j = 0; j < DPT; ++j) lv[l][j] = base[(size_t)(lw[l] ? first + (uint32_t)wv : 0u) * R4 + j * WAVE + lane];
        }
#pragma unroll
        for (int q = 0; q < RS_TOP_BATCH / RS_WAVES; ++q) {
            const uint32_t r = (uint32_t)wv + 4u * q;
            tw[q] = r < a.tree.top_rows ? 1u : 0u;
            tb[q] = r < top_node ? 1u : 0u;
#pragma unroll
            for (int j = 0; j < DPT; ++j) tv[q][j] = top4[(size_t)(tw[q] ? r : 0u) * R4 + j * WAVE + lane];
        }
    } else {
#pragma unroll
        for (int j = 0; j < DPT; ++j) hcount[j] = a.hist[t * DPT + j];
    }

    // ---- load (wave-striped: item i of lane l of wave w is element w*(ITEMS*64) + i*64 + l of the tile) and rank
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS], pay[PAYLOAD ? ITEMS : 1];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t e = (uint32_t)wv * (ITEMS * WAVE) + (uint32_t)i * WAVE + lane;
        const bool valid = e < tile_n;
        key[i] = valid ? a.kin[tile_start + e] : 0xFFFFFFFFu;
        if constexpr (VIDX) val[i] = tile_start + e;
        else val[i] = valid ? a.vin[tile_start + e] : 0u;
        if constexpr (PAYLOAD) pay[i] = valid ? a.pin[tile_start + e] : 0u;
    }
    if (SCANNED) {
        // (in program order behind the key loads: waiting for the rows leaves the keys in flight)
        uint4* part = reinterpret_cast<uint4*>(sbuf);             // [wave][before | total][RADIX], still unused otherwise
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            uint4 ab = make_uint4(0u, 0u, 0u, 0u), at = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int l = 0; l < RS_MAX_LEVELS - 1; ++l) {
                ab.x += lw[l] ? lv[l][j].x : 0u; ab.y += lw[l] ? lv[l][j].y : 0u; ab.z += lw[l] ? lv[l][j].z : 0u; ab.w += lw[l] ? lv[l][j].w : 0u;
            }
#pragma unroll
            for (int q = 0; q < RS_TOP_BATCH / RS_WAVES; ++q) {
                at.x += tw[q] ? tv[q][j].x : 0u; at.y += tw[q] ? tv[q][j].y : 0u; at.z += tw[q] ? tv[q][j].z : 0u; at.w += tw[q] ? tv[q][j].w : 0u;
                ab.x += tb[q] ? tv[q][j].x : 0u; ab.y += tb[q] ? tv[q][j].y : 0u; ab.z += tb[q] ? tv[q][j].z : 0u; ab.w += tb[q] ? tv[q][j].w : 0u;
            }
            for (uint32_t r0 = RS_TOP_BATCH; r0 < a.tree.top_rows; r0 += RS_TOP_BATCH)      // > 8192 tiles only
#pragma unroll
                for (int q = 0; q < RS_TOP_BATCH / RS_WAVES; ++q) {
                    const uint32_t r = r0 + (uint32_t)wv + 4u * q;
                    if (r < a.tree.top_rows) {
                        const uint4 v = top4[(size_t)r * R4 + j * WAVE + lane];
                        at.x += v.x; at.y += v.y; at.z += v.z; at.w += v.w;
                        if (r < top_node) { ab.x += v.x; ab.y += v.y; ab.z += v.z; ab.w += v.w; }
                    }
                }
            part[(wv * 2 + 0) * R4 + j * WAVE + lane] = ab;
            part[(wv * 2 + 1) * R4 + j * WAVE + lane] = at;
        }
    }
    RS_DRAIN();
    RS_STAMP(1);
    if (!BALLOT) {
        // rank among the pairs of this wave with the same digit, in (item, lane) order = the counter's value before this
        // lane's add (see the header on the order the LDS unit applies them in)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t e = (uint32_t)wv * (ITEMS * WAVE) + (uint32_t)i * WAVE + lane;
            const uint32_t d = (rs_xf(key[i], a.sub) >> a.shift) & (RADIX - 1);
            rank[i] = 0;
            if (e < tile_n) rank[i] = __hip_atomic_fetch_add(&wave_hist[wv][d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t e = (uint32_t)wv * (ITEMS * WAVE) + (uint32_t)i * WAVE + lane;
            const bool valid = e < tile_n;
            const uint32_t d = (rs_xf(key[i], a.sub) >> a.shift) & (RADIX - 1);
            unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
            for (int b = 0; b < DB; ++b) {
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
        for (int j = 0; j < DPT; ++j)
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) {
                before[j] += sbuf[(w * 2 + 0) * RADIX + t * DPT + j];
                hcount[j] += sbuf[(w * 2 + 1) * RADIX + t * DPT + j];
            }
    }
    RS_STAMP(2);

    // ---- thread t owns digits t * DPT + j: per-wave exclusive prefixes, tile totals, position of the digit inside the tile
    uint32_t total[DPT], tsum = 0;
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            const uint32_t c = wave_hist[w][t * DPT + j];
            wave_hist[w][t * DPT + j] = run;
            run += c;
        }
        total[j] = run;
        tsum += run;
    }
    const uint32_t incl = wave_incl_scan_dpp(tsum);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t wadd = 0;
    for (int w = 0; w < wv; ++w) wadd += wsum[w];
    uint32_t dbase[DPT];
    {
        uint32_t run = wadd + incl - tsum;
#pragma unroll
        for (int j = 0; j < DPT; ++j) { dbase[j] = run; digit_base[t * DPT + j] = run; run += total[j]; }
    }
    uint32_t gdigit_base[DPT];
    if constexpr (SCANNED) {
        rs_digit_bases<DPT>(hcount, gdigit_base, wsum, lane, wv);
#pragma unroll
        for (int j = 0; j < DPT; ++j) gbase[t * DPT + j] = gdigit_base[j] + before[j] - dbase[j];
        __syncthreads();
    } else {
        // ---- publish, gather, publish: a two-level all-gather of the tiles' digit counts (one digit per thread).
        // Every tile publishes its counts; it reads the counts of the tiles of its group of 16 in ONE batch of loads; the
        // last tile of a group publishes the group's sum; every tile reads the group sums, 16 per batch.  Two dependent
        // round trips whatever the number of tiles.  (The first version walked back 16 predecessors at a time until it
        // met a tile that had already published an INCLUSIVE count: with every tile of a small sort starting together
        // nobody has one early, and tile T paid ~T/16 dependent round trips -- 4.4 us of a tile's 9.8 us at 40 tiles,
        // 7.7 of 12 us at 203, tools/ubench/sort_bench.)
        // A tile only waits for tiles with a smaller id, which were dispatched before it (or, past 256 tiles, took their
        // ticket before it): they are resident or done whatever else runs on the chip.  (Round 3 also tried reading ALL
        // tiles' counts, later ones included, so that the digit totals come out of the gather and no histogram kernel
        // runs: 4.8 us less per depth sort inside a replay -- but a tile that waits for a LATER tile needs the whole grid
        // resident, and the eight keyframes of a mapping window sort at the same time on eight streams: 8 x 203 tiles
        // of the VGA tile sort against 1536 slots can leave every resident tile waiting for one that cannot start.
        // Removed; the totals come from the histogram the key producer or rs_hist_kernel counted.)  Spins are bounded.
        static_assert(DPT == 1, "the one-sweep path ranks 256 digits");
        constexpr uint32_t G = RS_GROUP;
        const uint32_t live_tiles = (n_live + (uint32_t)TILE_PAIRS - 1) / (uint32_t)TILE_PAIRS;
        const uint32_t grp = tile / G, jin = tile % G;                  // wave-uniform
        const uint32_t members = min(G, live_tiles - grp * G);
        uint64_t* gstat = a.status + (size_t)a.tiles * RADIX;           // the group words follow the tiles' words
        rs_digit_bases<DPT>(hcount, gdigit_base, wsum, lane, wv);
        __hip_atomic_store(a.status + (size_t)tile * RADIX + t, RS_FLAG_LOCAL | (uint64_t)total[0], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0, pre = 0, gpre = 0;
        bool ok = rs_gather(a.status + (size_t)grp * G * RADIX + t, RADIX, jin, pre, spins, a.error + a.pass);
        if (jin == members - 1)      // the group's last live tile: `pre` covers the whole group but this tile
            __hip_atomic_store(gstat + (size_t)grp * RADIX + t, RS_FLAG_GLOBAL | (uint64_t)(pre + total[0]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        if (ok) ok = rs_gather(gstat + t, RADIX, grp, gpre, spins, a.error + a.pass);
        gbase[t] = gdigit_base[0] + gpre + pre - dbase[0];
        __syncthreads();
    }

    RS_STAMP(3);
    // ---- lay the tile out digit by digit in LDS (stable), then stream each run to its final place
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t e = (uint32_t)wv * (ITEMS * WAVE) + (uint32_t)i * WAVE + lane;
        const uint32_t d = (rs_xf(key[i], a.sub) >> a.shift) & (RADIX - 1);
        rank[i] = e < tile_n ? digit_base[d] + wave_hist[wv][d] + rank[i] : 0xFFFFFFFFu;      // slot inside the tile
    }
    rs_exchange<ITEMS>(sbuf, rank, key, t);
    __syncthreads();
    rs_exchange<ITEMS>(sbuf, rank, val, t);
    if constexpr (PAYLOAD) {
        __syncthreads();
        rs_exchange<ITEMS>(sbuf, rank, pay, t);
    }
    RS_STAMP(4);
    uint32_t g[ITEMS];
    bool live[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t p = (uint32_t)i * RS_THREADS + t;
        live[i] = p < tile_n;
        g[i] = gbase[(rs_xf(key[i], a.sub) >> a.shift) & (RADIX - 1)] + p;
    }
    if (!final_pass) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if (live[i]) {
                a.kout[g[i]] = key[i];
                a.vout[g[i]] = val[i];
                if constexpr (PAYLOAD) a.pout[g[i]] = pay[i];
            }
        RS_STAMP(5); RS_DRAIN(); RS_STAMP(6); RS_STAMP(7);
        return;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (live[i]) {
            if (a.kfinal) a.kfinal[g[i]] = key[i];
            a.vfinal[g[i]] = val[i];
        }
    if (a.ranges) {
        // The per-key ranges (the rasteriser's tile ranges), from the keys this thread holds in output order: equal keys are
        // neighbours in the tile's digit-ordered array (same digit; the earlier passes ordered the rest), so an element whose
        // LEFT neighbour carries another key opens its key's run inside this tile and closes the neighbour's; the tile's
        // first element opens, its last one closes.  A key's run may continue in other tiles: every tile offers the ends of
        // its piece (atomicMin / atomicMax on words preset to {~0, 0}).  The left neighbour of element (i, t) is element
        // (i, t - 1): the lane below, or -- lane 0 -- the previous wave's last lane, handed over through 4 x ITEMS words of
        // LDS.  (Done here, after the stores, on registers that are live anyway: kept across the exchanges the neighbour keys
        // cost 30 spilled VGPRs in the 72-register kernel of the C5 tile sort.)
        __shared__ uint32_t s_edge[RS_WAVES * ITEMS];
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) s_edge[wv * ITEMS + i] = key[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t p = (uint32_t)i * RS_THREADS + t;
            uint32_t l = (uint32_t)__shfl_up((int)key[i], 1, 64);
            if (lane == 0) l = wv > 0 ? s_edge[(wv - 1) * ITEMS + i] : (i > 0 ? s_edge[(RS_WAVES - 1) * ITEMS + i - 1] : 0xFFFFFFFFu);
            if (p < tile_n && (p == 0 || l != key[i])) {
                atomicMin(&a.ranges[key[i]].x, g[i]);
                if (p > 0)          // the piece that ends at p - 1: its last position follows from ITS digit's base
                    atomicMax(&a.ranges[l].y, gbase[(rs_xf(l, a.sub) >> a.shift) & (RADIX - 1)] + p);
            }
            if (p + 1 == tile_n) atomicMax(&a.ranges[key[i]].y, g[i] + 1u);
        }
    }
    RS_STAMP(5);
    RS_DRAIN();
    RS_STAMP(6);
    if (a.aux_out) {
        uint2* __restrict__ aout = a.aux_out;
        if constexpr (PAYLOAD) {
            // the rectangle that travelled with the pair: x0 | y0 << 8 | w << 16 | h << 24 -> {x0 | y0 << 16, w | h << 16}
#pragma unroll
            for (int i = 0; i < ITEMS; ++i)
                if (live[i]) {
                    const uint32_t r = pay[i];
                    aout[g[i]] = key[i] == 0xFFFFFFFFu ? make_uint2(0u, 0u)
                                                       : make_uint2((r & 0xFFu) | ((r & 0xFF00u) << 8), ((r >> 16) & 0xFFu) | ((r >> 24) << 16));
                }
        } else {
            // the gather through the sorted values: every load of the thread in flight before the first store; a culled
            // Gaussian (key of all ones) has the empty rectangle by construction and is not fetched.  (At 2 M keys the
            // gather costs ~25 us -- 1.5 M fetches of a 128-byte line for 8 bytes each, at the Infinity Cache's ~8.6 TB/s
            // for random rows -- against ~11 us for a plain pass: large sorts carry the rectangle as a payload instead.)
            const uint2* __restrict__ ain = a.aux_in;
            uint2 r[ITEMS];
#pragma unroll
            for (int i = 0; i < ITEMS; ++i)
                r[i] = (live[i] && !(a.aux_skip_ones && key[i] == 0xFFFFFFFFu)) ? ain[val[i]] : make_uint2(0u, 0u);
#pragma unroll
            for (int i = 0; i < ITEMS; ++i)
                if (live[i]) aout[g[i]] = r[i];
        }
    }
    RS_DRAIN();
    RS_STAMP(7);
}

// ------------------------------------------------------------------------------------------------
// host side

// A plain sort of n pairs on key bits [0, bits): `ka`/`va` hold the input; the passes ping-pong between (ka, va) and
// (kb, vb).  The sorted pairs end in (kb, vb) when the number of passes is odd and in (ka, va) when it is even
// (radix_result_in_b tells which).  With `n_dev` the live pair count is read on the device (min(n, *n_dev)) and n is only
// the capacity that sizes grids and scratch: no host-side size needed.
bool radix_result_in_b(int bits) { return (rs_plan_plain(bits).npasses & 1) != 0; }

// RADIX_ERROR_WORDS device words (one per pass) that a timed-out look-back spin sets to 1; any non-zero word = the
// sort's output is invalid (checked by the kernels that consume it and by the caller at its next sync point)
const uint32_t* radix_error_flag(void* temp, uint64_t n, int bits) {
    return rs_carve(temp, n ? n : 1, rs_plan_plain(bits), rs_scanned(n ? n : 1)).error;
}
const uint32_t* radix_depth_error_flag(void* temp, uint64_t n) {
    return rs_carve(temp, n ? n : 1, rs_plan_for_depth(n ? n : 1), rs_scanned(n ? n : 1)).error;
}

// The region of `temp` that must be zero when the sort starts (histograms, tickets, error and wide flags, status words /
// tree sums); a kernel that runs right before the sort can clear it with grid_zero() and pass temp_zeroed = true.
void radix_zero_region(void* temp, uint64_t n, int bits, uint32_t** ptr, size_t* words) {
    const RsTemp t = rs_carve(temp, n ? n : 1, rs_plan_plain(bits), rs_scanned(n ? n : 1));
    *ptr = t.hist;
    *words = t.zero_bytes / 4;
}
void radix_depth_zero_region(void* temp, uint64_t n, uint32_t** ptr, size_t* words) {
    const RsTemp t = rs_carve(temp, n ? n : 1, rs_plan_for_depth(n ? n : 1), rs_scanned(n ? n : 1));
    *ptr = t.hist;
    *words = t.zero_bytes / 4;
}

// does a sort of n pairs read a global digit histogram (one-sweep path) -- i.e. is it worth counting one while the keys are produced?
bool radix_wants_hist(uint64_t n) { return n > 0 && !rs_scanned(n); }
// does the depth sort of n Gaussians carry the rectangle as a payload (else it gathers it in its final pass)?
bool radix_depth_payload(uint64_t n) { return n > 0 && rs_scanned(n); }

template <int ITEMS, int DB, bool SCANNED, bool PAYLOAD>
static void rs_launch_pass(const RsPassArgs& a, uint32_t tiles, bool ballot, hipStream_t s) {
    if (ballot && DB == 8 && !PAYLOAD && a.vin != nullptr)
        hipLaunchKernelGGL((rs_pass_kernel<ITEMS, 8, SCANNED, false, true, false>), dim3(tiles), dim3(RS_THREADS), 0, s, a);
    else if (a.vin == nullptr && (DB == 9 || !SCANNED))
        hipLaunchKernelGGL((rs_pass_kernel<ITEMS, DB, SCANNED, PAYLOAD, false, (DB == 9 || !SCANNED)>), dim3(tiles), dim3(RS_THREADS), 0, s, a);
    else
        hipLaunchKernelGGL((rs_pass_kernel<ITEMS, DB, SCANNED, PAYLOAD, false, false>), dim3(tiles), dim3(RS_THREADS), 0, s, a);
}
template <int DB, bool PAYLOAD>
static void rs_launch_pass_items(const RsPassArgs& a, int items, bool scanned, uint32_t tiles, bool ballot, hipStream_t s) {
    if (scanned) {
        if (items == RS_ITEMS_WIDE) rs_launch_pass<RS_ITEMS_WIDE, DB, true, PAYLOAD>(a, tiles, ballot, s);
        else if (items == RS_ITEMS_MID) rs_launch_pass<RS_ITEMS_MID, DB, true, PAYLOAD>(a, tiles, ballot, s);
        else rs_launch_pass<RS_ITEMS, DB, true, PAYLOAD>(a, tiles, ballot, s);
    } else if constexpr (!PAYLOAD && DB == 8) {        // (payloads and 9-bit digits only exist on the counted-tiles path)
        if (items == RS_ITEMS_SMALL) rs_launch_pass<RS_ITEMS_SMALL, 8, false, false>(a, tiles, ballot, s);
        else if (items == RS_ITEMS_MID) rs_launch_pass<RS_ITEMS_MID, 8, false, false>(a, tiles, ballot, s);
        else rs_launch_pass<RS_ITEMS, 8, false, false>(a, tiles, ballot, s);
    }
}

struct RsBuffers {
    uint2* ranges;                 // see RsPassArgs::ranges (NULL: none)
    uint32_t *ka, *va, *pa;        // input (pa NULL: no payload)
    bool va_is_index;              // the first pass takes value = index instead of reading va (va is still written later)
    uint32_t *kb, *vb, *pb;        // ping-pong partners
    uint32_t *kfinal, *vfinal;     // depth plan: where the final pass writes (kfinal NULL: no keys)
    const uint2* aux_in;
    uint2* aux_out;
    bool aux_skip_ones;
};
static int rs_run(const RsPlan& pl, const RsBuffers& b, uint64_t n, void* temp, hipStream_t s, const uint32_t* n_dev,
                  bool temp_zeroed, const uint32_t* ext_hist, bool exclusive) {
    if (n == 0) return 0;
    if (n >= (1ull << 32)) { set_error("radix sort: more than 2^32-1 pairs"); return 1; }
    const bool scanned = rs_scanned(n);
    const RsTemp t = rs_carve(temp, n, pl, scanned);
    if (!temp_zeroed) MGS_HIP(zero_fill(t.hist, t.zero_bytes, s));
    const uint32_t tiles = rs_tiles(n, scanned);
    const int items = rs_tile_items(n, scanned);
    const bool payload = b.pa != nullptr;
    if (payload && !scanned) { set_error("radix sort: payloads need the counted-tiles path"); return 1; }
    const RsTree tree = rs_tree(tiles);
    RsTreeArgs ta;
    ta.levels = tree.levels;
    for (int l = 0; l < RS_MAX_LEVELS; ++l) ta.off[l] = l < tree.levels ? tree.off[l] : 0u;
    ta.top_rows = tree.rows[tree.levels - 1];
    if (!scanned && !ext_hist) {
        RsHistArgs h;
        h.npasses = pl.npasses; h.radix = pl.radix; h.sub = pl.sub; h.wide = pl.depth ? t.wide : nullptr;
        for (int p = 0; p < RS_MAX_PASSES; ++p) { h.shift[p] = pl.shift[p]; h.mask[p] = (1 << pl.db[p]) - 1; }
        hipLaunchKernelGGL(rs_hist_kernel, dim3(min(tiles, 1024u)), dim3(RS_THREADS), 0, s, b.ka, (uint32_t)n, n_dev, h, t.hist);
    }
    const uint32_t* ghist = (!scanned && ext_hist) ? ext_hist : t.hist;
    const int hstride = (!scanned && ext_hist) ? 256 : pl.radix;
    const uint32_t *kin = b.ka, *vin = b.va_is_index ? nullptr : b.va, *pin = b.pa;
    uint32_t *kout = b.kb, *vout = b.vb, *pout = b.pb;
    uint32_t *kalt = b.ka, *valt = b.va, *palt = b.pa;     // the side the NEXT pass writes to (the first input, once consumed)
    for (int p = 0; p < pl.npasses; ++p) {
        RsPassArgs a;
        a.kin = kin; a.vin = vin; a.pin = pin; a.kout = kout; a.vout = vout; a.pout = pout;
        a.n = (uint32_t)n; a.n_dev = n_dev; a.shift = pl.shift[p]; a.sub = pl.sub;
        a.hist = ghist + p * hstride;
        a.status = t.status + (size_t)p * rs_status_rows(tiles) * pl.radix;
        a.tiles = tiles;
        // Tile ids come from an atomic ticket: a tile then only waits for tiles that have STARTED, whatever order and place
        // the hardware gives the workgroups of this grid and of the grids running beside it (a mapping window's keyframes
        // sort on a stream each, MonoGS's three processes share the device; workgroups are dealt round-robin to per-XCD
        // dispatchers, so a higher block id can be resident and spinning while a lower one waits for a slot on a full XCD).
        // Only a caller that vouches for an otherwise idle device (MGS_FLAG_EXCLUSIVE_DEVICE) gets block ids for grids of at
        // most one workgroup per CU -- those are co-resident whatever the dispatch order -- and saves the atomic's round trip.
        a.ticket = (exclusive && tiles <= 256u) ? nullptr : t.tickets + p;
        a.error = t.error;
        a.pass = p;
        // (counted tiles only: no tile waits for another there.  One sweep with block ids as tile ids -- an exclusive device,
        //  every tile resident -- could take it too and was measured: a replayed tracking iteration 215.5 against 217.7 us,
        //  inside the noise: those passes are bound by their round trips, not by the memory side)
        a.xcd_band = (scanned && g_opt_radix_xcd_band != 0) ? 1 : 0;
        a.counts = t.counts; a.sums = t.sums + (size_t)p * tree.sum_rows * pl.radix;
        a.tree = ta;
        a.wide = pl.depth ? t.wide : nullptr;
        a.cond = pl.depth ? (p == 2 ? 1 : (p == 3 ? 2 : 0)) : 0;
        a.last = (!pl.depth && p == pl.npasses - 1) ? 1 : 0;
        // the final pass of a plain sort writes the ping-pong side; the depth sort has its own final arrays and no keys
        a.vfinal = b.vfinal ? b.vfinal : vout;
        a.kfinal = b.vfinal ? b.kfinal : kout;
        a.ranges = (!pl.depth && p == pl.npasses - 1) ? b.ranges : nullptr;
        if (a.ranges) a.kfinal = nullptr;       // whoever asks for the ranges does not read the sorted keys: 4 n bytes of stores less
        a.aux_in = b.aux_in; a.aux_out = b.aux_out; a.aux_skip_ones = b.aux_skip_ones ? 1 : 0;
        if (scanned) {
            RsTileHistArgs h;
            h.keys = kin; h.n = (uint32_t)n; h.n_dev = n_dev; h.shift = a.shift; h.sub = pl.sub; h.tiles = tiles; h.tr = ta;
            h.counts = t.counts; h.sums = const_cast<uint32_t*>(a.sums);
            h.wide = t.wide; h.detect = (pl.depth && p == 0) ? 1 : 0; h.only_if_wide = a.cond == 2 ? 1 : 0;
            if (items == RS_ITEMS_WIDE) {
                if (pl.db[p] == 9) rs_launch_tile_hist<RS_ITEMS_WIDE, 9>(h, s); else rs_launch_tile_hist<RS_ITEMS_WIDE, 8>(h, s);
            } else if (items == RS_ITEMS_MID) {
                if (pl.db[p] == 9) rs_launch_tile_hist<RS_ITEMS_MID, 9>(h, s); else rs_launch_tile_hist<RS_ITEMS_MID, 8>(h, s);
            } else {
                if (pl.db[p] == 9) rs_launch_tile_hist<RS_ITEMS, 9>(h, s); else rs_launch_tile_hist<RS_ITEMS, 8>(h, s);
            }
        }
        const bool ballot = g_opt_radix_ballot_rank != 0;
        if (pl.db[p] == 9) {
            if (payload) rs_launch_pass_items<9, true>(a, items, scanned, tiles, ballot, s);
            else rs_launch_pass_items<9, false>(a, items, scanned, tiles, ballot, s);
        } else {
            if (payload) rs_launch_pass_items<8, true>(a, items, scanned, tiles, ballot, s);
            else rs_launch_pass_items<8, false>(a, items, scanned, tiles, ballot, s);
        }
        kin = kout; vin = vout; pin = pout;
        uint32_t* tk = kalt; kalt = kout; kout = tk;          // (kin is now the side just written; the next pass writes the other)
        uint32_t* tv = valt; valt = vout; vout = tv;
        uint32_t* tp = palt; palt = pout; pout = tp;
    }
    MGS_HIP(hipGetLastError());
    return 0;
}

int radix_sort_pairs(uint32_t* ka, uint32_t* va, uint32_t* kb, uint32_t* vb, uint64_t n, int bits, void* temp,
                     hipStream_t s, const uint32_t* n_dev, bool temp_zeroed, const uint2* aux_in, uint2* aux_out,
                     const uint32_t* ext_hist, bool aux_empty_for_ones, bool exclusive, uint2* ranges) {
    const RsPlan pl = rs_plan_plain(bits);
    if (pl.npasses > RS_MAX_PASSES) { set_error("radix sort: more than 32 key bits"); return 1; }
    RsBuffers b;
    b.ranges = ranges;
    b.ka = ka; b.va = va; b.va_is_index = false; b.pa = nullptr; b.kb = kb; b.vb = vb; b.pb = nullptr; b.kfinal = nullptr; b.vfinal = nullptr;
    b.aux_in = aux_in; b.aux_out = aux_out; b.aux_skip_ones = aux_empty_for_ones;
    return rs_run(pl, b, n, temp, s, n_dev, temp_zeroed, ext_hist, exclusive);
}

// The depth sort of the forward (see the header): keys[P] = float bits of the depths (all ones: culled), destroyed;
// perm[P] receives the Gaussian indices in (depth, index) order, rect_sorted[P] their tile rectangles.  `rect` is either
// uint2[P] (gathered by the final pass) or, with `payload`, uint32[2 P]: the packed rectangles in the first half, the
// second half their ping-pong partner.  val_a / val_b and key_b: P words each of scratch.
int radix_sort_depth(uint32_t* keys, uint32_t* key_b, uint32_t* val_a, uint32_t* val_b, void* rect, bool payload,
                     uint32_t* perm, uint2* rect_sorted, uint64_t n, void* temp, hipStream_t s, bool temp_zeroed, bool exclusive) {
    const RsPlan pl = rs_plan_for_depth(n);
    RsBuffers b;
    b.ranges = nullptr;
    b.ka = keys; b.va = val_a; b.va_is_index = true; b.kb = key_b; b.vb = val_b;
    b.pa = payload ? (uint32_t*)rect : nullptr; b.pb = payload ? (uint32_t*)rect + n : nullptr;
    b.kfinal = nullptr; b.vfinal = perm;
    b.aux_in = payload ? nullptr : (const uint2*)rect; b.aux_out = rect_sorted; b.aux_skip_ones = true;
    return rs_run(pl, b, n, temp, s, nullptr, temp_zeroed, nullptr, exclusive);
}

int set_radix_spin_limit(uint32_t limit) {
    const uint32_t v = limit == 0xFFFFFFFFu ? RS_SPIN_LIMIT : limit;
    MGS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rs_spin_limit), &v, sizeof(v)));
    return 0;
}

}  // namespace mgs
