// Stable LSD radix sort of (uint32 key, uint32 value) pairs, hand-written for gfx950 (wave64).
//
// Used twice per forward (binning.hip): the P Gaussians by their 32 depth bits, then the R
// (tile id, Gaussian index) instances by tile id.  Replaces upstream's cub::DeviceRadixSort call
// (SURVEY.md section 2.1 K4).
//
// Two paths, bit-identical results, chosen by size (rs_scanned()):
//
// "One sweep" (<= 640 k pairs: every sort of a SLAM-sized map; fewest launches):
//   * ONE histogram kernel reads the keys once and counts every digit of every pass (each pass kernel
//     scans its 256 counts into exclusive global bases itself);
//   * per pass ONE kernel.  A workgroup (256 threads = 4 waves) takes a tile of 1024-4096 pairs, ranks
//     them stably (wave64 __ballot match per digit bit + per-wave LDS counters), publishes its 256
//     digit counts, obtains the sum of the counts of all EARLIER tiles by decoupled look-back, lays
//     the tile out digit-by-digit in LDS and writes each digit's run to its final place (coalesced
//     runs instead of a 256-way scatter).
// "Pre-scanned offsets" (> 640 k pairs): per pass rs_tile_hist_kernel (digit counts of every tile),
//   rs_row_scan_kernel (exclusive scan along the tiles of each digit + digit totals) and the same
//   ranking / scatter kernel reading its offsets from that table -- no waiting between workgroups.  With
//   hundreds of co-resident tiles the look-back's status traffic and round trips dominated the pass.
// The last pass can also gather an auxiliary array through the sorted values (aux_out[pos] = aux_in[value]).
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

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / WAVE;
constexpr int RS_ITEMS = 16;                       // pairs per thread for large sorts: 4096-pair tiles
constexpr int RS_ITEMS_MID = 8;
constexpr int RS_ITEMS_SMALL = 4;                  // small sorts are latency-bound: 1024-pair tiles rank 4x faster
// Measured on MI355X (depth sort of P keys): 40 k: 72 us (16) -> 50 us (4); 400 k: 122 us (32) / 83 us (8);
// 400 k: 108 us (4); 2 M: 158 us (16) / 208 us (8) / 152 us (32).  Pre-scanned path, round 2 (C5): 16 -> 8 items leaves the
// depth sort at 0.136 ms and takes the tile sort from 0.094 to 0.105 ms; 32 items: 0.168 / 0.116 ms.  A few hundred tiles is the sweet spot between the per-tile
// ranking latency and the length of the look-back chain.
// Above this many pairs the pre-scanned path takes over.  Round 2: 1 M keys (depth sort): one sweep 0.134 ms, pre-scanned
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

// large sorts take the SCANNED path (rs_pass_kernel); a test knob forces either one (no environment lookups on the launch path)
int g_opt_radix_scanned = -1;       // mgs_debug_set_option("radix_scanned", -1 | 0 | 1): -1 = by size
static inline bool rs_scanned(uint64_t n) {
    if (g_opt_radix_scanned >= 0) return g_opt_radix_scanned == 1 && rs_items(n) == RS_ITEMS;
    return rs_items(n) == RS_ITEMS;          // > 640 k pairs: hundreds of tiles
}
static inline int rs_passes(int bits) { return (bits + 7) / 8; }
static inline uint32_t rs_tiles(uint64_t n) {
    const uint64_t tile = (uint64_t)RS_THREADS * rs_items(n);
    return (uint32_t)((n + tile - 1) / tile);
}

// temp layout: [hist: RS_MAX_PASSES*256 u32][tickets: RS_MAX_PASSES u32][error: RS_MAX_PASSES u32][pad]
//              [status: passes * tiles * 256 u64]
struct RsTemp {
    uint32_t* hist;
    uint32_t* tickets;
    uint32_t* error;
    uint64_t* status;
    size_t zero_bytes;     // everything from `hist` that must be zero before the sort
};
static RsTemp rs_carve(void* temp, uint64_t n, int bits) {
    char* p = (char*)align_up((size_t)temp, 256);
    RsTemp t;
    t.hist = (uint32_t*)p;
    t.tickets = t.hist + RS_MAX_PASSES * RS_RADIX;
    t.error = t.tickets + RS_MAX_PASSES;      // one word per pass
    char* q = (char*)align_up((size_t)(t.error + RS_MAX_PASSES), 256);
    t.status = (uint64_t*)q;
    const size_t status_bytes = (size_t)rs_passes(bits) * rs_tiles(n) * RS_RADIX * sizeof(uint64_t);
    t.zero_bytes = (size_t)(q - p) + status_bytes;
    return t;
}
size_t radix_temp_bytes(uint64_t n, int bits) {
    if (n == 0) return 256;
    const RsTemp t = rs_carve(nullptr, n, bits);
    return t.zero_bytes + 512;
}

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

// ---- SCANNED path, step 1: digit counts of every tile for one pass, counts[digit][tile] -----------------------
template <int ITEMS>
__global__ void __launch_bounds__(RS_THREADS) rs_tile_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                  const uint32_t* __restrict__ n_dev, int shift,
                                                                  uint32_t tiles, uint32_t* __restrict__ counts) {
    constexpr uint32_t TILE_PAIRS = RS_THREADS * ITEMS;
    __shared__ uint32_t wh[RS_WAVES][RS_RADIX];
    const int t = threadIdx.x, wv = t >> 6;
    for (int i = t; i < RS_WAVES * RS_RADIX; i += RS_THREADS) (&wh[0][0])[i] = 0;
    __syncthreads();
    const uint32_t n_live = n_dev ? min(n, n_dev[0]) : n;
    const uint32_t tile = blockIdx.x, tile_start = tile * TILE_PAIRS;
    if (tile_start < n_live) {
        const uint32_t tile_n = min(TILE_PAIRS, n_live - tile_start);
        uint32_t k[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t e = (uint32_t)i * RS_THREADS + t;
            k[i] = e < tile_n ? keys[tile_start + e] : 0u;
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if ((uint32_t)i * RS_THREADS + t < tile_n) atomicAdd(&wh[wv][(k[i] >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    counts[(size_t)t * tiles + tile] = (wh[0][t] + wh[1][t]) + (wh[2][t] + wh[3][t]);     // dead tiles write zeros
}

// ---- SCANNED path, step 2: counts[digit][tile] -> exclusive scan along the tiles of each digit; the row total goes to
// totals[digit] (the scatter kernel turns the 256 totals into digit bases itself, as the one-sweep path does with
// the global histogram).  One wave per digit, 64 tiles per step, DPP scan (the first version used __shfl_up, i.e.
// ds_bpermute, seven times per step: 8.7 us for 125 k counts).
__global__ void __launch_bounds__(RS_THREADS) rs_row_scan_kernel(uint32_t* __restrict__ counts, uint32_t tiles,
                                                                 uint32_t* __restrict__ totals /* [256] */) {
    // one WORKGROUP per digit: wave w scans the w-th quarter of the row (64 tiles per step, DPP scan), the quarters are
    // joined through LDS (a wave per digit walked 1287 tiles in 21 dependent steps: 13 us for the tile sort at C5)
    __shared__ uint32_t s_part[RS_WAVES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t d = blockIdx.x;                                          // 0..255
    uint32_t* row = counts + (size_t)d * tiles;
    const uint32_t per = ((tiles + RS_WAVES - 1) / RS_WAVES + WAVE - 1) / WAVE * WAVE;   // quarter length, multiple of 64
    const uint32_t lo = min(tiles, (uint32_t)wv * per), hi = min(tiles, lo + per);
    // pass 1: total of the quarter
    uint32_t sum = 0;
    for (uint32_t i = lo + lane; i < hi; i += WAVE) sum += row[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) s_part[wv] = sum;
    __syncthreads();
    uint32_t carry = 0;                                                     // wave-uniform: tiles before this quarter
    for (int w = 0; w < wv; ++w) carry += s_part[w];
    if (threadIdx.x == 0) totals[d] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    // pass 2: exclusive scan of the quarter (the re-read hits the cache)
    uint32_t next = (lo + lane) < hi ? row[lo + lane] : 0u;
    for (uint32_t b0 = lo; b0 < hi; b0 += WAVE) {
        const uint32_t i = b0 + lane;
        const uint32_t v = next;
        next = (i + WAVE) < hi ? row[i + WAVE] : 0u;
        const uint32_t incl = wave_incl_scan_dpp(v);
        if (i < hi) row[i] = carry + incl - v;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
}

struct RsPassArgs {
    const uint32_t* kin;
    uint32_t* kout;
    const uint32_t* vin;
    uint32_t* vout;
    uint32_t n;               // number of pairs (capacity when n_dev is set)
    const uint32_t* n_dev;    // optional: live count on the device (<= n after clamping)
    int shift;
    const uint32_t* hist;     // [256] global count of each digit for this pass
    uint64_t* status;         // [tiles][256]
    uint32_t* ticket;
    uint32_t* error;          // [RS_MAX_PASSES] one word per pass: pass p raises error[p] when a look-back spin times out
    int pass;
    const uint2* aux_in;      // optional (last pass): aux_out[final position] = aux_in[value]
    uint2* aux_out;
    const uint32_t* scanned;  // SCANNED path: [256][tiles] exclusive scan (digit-major) of the per-tile digit counts
    uint32_t tiles;
};

// SCANNED = false: one sweep -- the tile publishes its digit counts and finds the sum over earlier tiles by decoupled
//                  look-back (fewest launches: right for the small sorts of SLAM-sized maps).
// SCANNED = true:  the global position of every (digit, tile) run was computed beforehand by rs_tile_hist_kernel +
//                  rs_scan_kernel.  With several hundred co-resident tiles starting together the look-back reads
//                  ~tiles^2/2 x 256 status words per pass -- more L2 traffic than the keys -- and its chain of
//                  round trips, not the data movement, set the pass time (35 us for 32 MB at 2 M pairs).
template <int ITEMS, bool SCANNED, int WINDOW = RS_WINDOW>
__global__ void __launch_bounds__(RS_THREADS) rs_pass_kernel(RsPassArgs a) {
    constexpr int TILE_PAIRS = RS_THREADS * ITEMS;
    __shared__ uint32_t wave_hist[RS_WAVES][RS_RADIX];
    __shared__ uint32_t digit_base[RS_RADIX];
    __shared__ int64_t gbase[RS_RADIX];                  // global position of LDS slot 0 for each digit (may be negative)
    __shared__ uint32_t skeys[TILE_PAIRS];
    __shared__ uint32_t svals[TILE_PAIRS];
    __shared__ uint32_t wsum[RS_WAVES];
    __shared__ uint32_t s_tile;

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint32_t hcount = a.hist[t];       // requested first: needed only after the ranking, one round trip hidden
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
    for (int i = t; i < RS_WAVES * RS_RADIX; i += RS_THREADS) (&wave_hist[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t tile_start = tile * (uint32_t)TILE_PAIRS;
    const uint32_t n_live = a.n_dev ? min(a.n, a.n_dev[0]) : a.n;
    if (tile_start >= n_live) return;        // capacity mode: tiles past the live count have nothing to do
                                             // (tickets are dense, so no live tile ever looks back at them)
    const uint32_t tile_n = min((uint32_t)TILE_PAIRS, n_live - tile_start);

    // ---- load (wave-striped: item i of lane l of wave w is element w*(ITEMS*64) + i*64 + l of the tile) and rank
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t e = (uint32_t)wv * (ITEMS * WAVE) + (uint32_t)i * WAVE + lane;
        const bool valid = e < tile_n;
        key[i] = valid ? a.kin[tile_start + e] : 0xFFFFFFFFu;
        val[i] = valid ? a.vin[tile_start + e] : 0u;
    }
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
    __syncthreads();

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
    // exclusive global base of digit t = scan of this pass's 256 digit totals (global histogram on the one-sweep path,
    // row totals of rs_row_scan_kernel on the SCANNED path): cheaper here than a separate launch
    const uint32_t hincl = wave_incl_scan_dpp(hcount);
    __syncthreads();                     // wsum is reused
    if (lane == 63) wsum[wv] = hincl;
    __syncthreads();
    uint32_t hadd = 0;
    for (int w = 0; w < wv; ++w) hadd += wsum[w];
    const uint32_t gdigit_base = hadd + hincl - hcount;
    if (SCANNED) {
        // position of this tile's run of digit t inside the digit: straight from the scanned table
        gbase[t] = (int64_t)gdigit_base + (int64_t)a.scanned[(size_t)t * a.tiles + tile] - (int64_t)dbase;
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

    // ---- lay the tile out digit by digit in LDS (stable), then stream each run to its final place
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t e = (uint32_t)wv * (ITEMS * WAVE) + (uint32_t)i * WAVE + lane;
        if (e < tile_n) {
            const uint32_t d = (key[i] >> a.shift) & 0xFFu;
            const uint32_t pos = digit_base[d] + wave_hist[wv][d] + rank[i];
            skeys[pos] = key[i];
            svals[pos] = val[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t p = (uint32_t)i * RS_THREADS + t;
        if (p < tile_n) {
            const uint32_t k = skeys[p];
            const int64_t g = gbase[(k >> a.shift) & 0xFFu] + (int64_t)p;
            const uint32_t v = svals[p];
            a.kout[g] = k;
            a.vout[g] = v;
            if (a.aux_out) a.aux_out[g] = a.aux_in[v];
        }
    }
}

// Sorts n pairs on key bits [0, bits).  With `n_dev` the live pair count is read on the device
// (min(n, *n_dev)) and n is only the capacity that sizes grids and scratch: no host-side size needed.
// Sorts n pairs on key bits [0, bits).  `ka`/`va` hold the input; the passes ping-pong between
// (ka, va) and (kb, vb).  The sorted pairs end in (kb, vb) when the number of passes is odd and in
// (ka, va) when it is even (radix_result_in_b tells which).
bool radix_result_in_b(int bits) { return (rs_passes(bits) & 1) != 0; }

// RADIX_ERROR_WORDS device words (one per pass) that a timed-out look-back spin sets to 1; any non-zero word = the
// sort's output is invalid (checked by the kernels that consume it and by the caller at its next sync point)
const uint32_t* radix_error_flag(void* temp, uint64_t n, int bits) { return rs_carve(temp, n ? n : 1, bits).error; }

// The region of `temp` that must be zero when the sort starts (histograms, tickets, error flag, status words); a
// kernel that runs right before the sort can clear it with grid_zero() and pass temp_zeroed = true.
void radix_zero_region(void* temp, uint64_t n, int bits, uint32_t** ptr, size_t* words) {
    const RsTemp t = rs_carve(temp, n ? n : 1, bits);
    *ptr = t.hist;
    // the SCANNED path never reads the status words (it reuses their storage for counts it fully overwrites)
    *words = rs_scanned(n) ? (size_t)((char*)t.status - (char*)t.hist) / 4 : t.zero_bytes / 4;
}

// does a sort of n pairs read a global digit histogram (one-sweep path) -- i.e. is it worth counting one while the keys are produced?
bool radix_wants_hist(uint64_t n) { return n > 0 && !rs_scanned(n); }

int radix_sort_pairs(uint32_t* ka, uint32_t* va, uint32_t* kb, uint32_t* vb, uint64_t n, int bits, void* temp,
                     hipStream_t s, const uint32_t* n_dev, bool temp_zeroed, const uint2* aux_in, uint2* aux_out,
                     const uint32_t* ext_hist) {
    if (n == 0) return 0;
    if (n >= (1ull << 32)) { set_error("radix sort: more than 2^32-1 pairs"); return 1; }
    const int npasses = rs_passes(bits);
    if (npasses > RS_MAX_PASSES) { set_error("radix sort: more than 32 key bits"); return 1; }
    const RsTemp t = rs_carve(temp, n, bits);
    if (!temp_zeroed) MGS_HIP(zero_fill(t.hist, rs_scanned(n) ? (size_t)((char*)t.status - (char*)t.hist) : t.zero_bytes, s));
    const uint32_t tiles = rs_tiles(n);
    const uint32_t hist_blocks = min(tiles, 1024u);
    const bool scanned = rs_scanned(n);
    if (!scanned && !ext_hist)
        hipLaunchKernelGGL(rs_hist_kernel, dim3(hist_blocks), dim3(RS_THREADS), 0, s, ka, (uint32_t)n, n_dev, npasses, t.hist);
    const uint32_t* ghist = (!scanned && ext_hist) ? ext_hist : t.hist;
    uint32_t* counts = reinterpret_cast<uint32_t*>(t.status);      // the status words are unused on the SCANNED path
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
        a.scanned = counts; a.tiles = tiles;
        const bool last = p == npasses - 1;
        a.aux_in = last ? aux_in : nullptr; a.aux_out = last ? aux_out : nullptr;
        if (scanned) {
            // t.hist[p] receives the 256 digit totals
            hipLaunchKernelGGL(rs_tile_hist_kernel<RS_ITEMS>, dim3(tiles), dim3(RS_THREADS), 0, s, kin, (uint32_t)n, n_dev,
                               a.shift, tiles, counts);
            hipLaunchKernelGGL(rs_row_scan_kernel, dim3(RS_RADIX), dim3(RS_THREADS), 0, s, counts, tiles,
                               t.hist + p * RS_RADIX);
            hipLaunchKernelGGL((rs_pass_kernel<RS_ITEMS, true>), dim3(tiles), dim3(RS_THREADS), 0, s, a);
        } else if (rs_items(n) == RS_ITEMS_SMALL)
            hipLaunchKernelGGL((rs_pass_kernel<RS_ITEMS_SMALL, false>), dim3(tiles), dim3(RS_THREADS), 0, s, a);
        else if (rs_items(n) == RS_ITEMS_MID)
            hipLaunchKernelGGL((rs_pass_kernel<RS_ITEMS_MID, false>), dim3(tiles), dim3(RS_THREADS), 0, s, a);
        else
            hipLaunchKernelGGL((rs_pass_kernel<RS_ITEMS, false>), dim3(tiles), dim3(RS_THREADS), 0, s, a);
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
