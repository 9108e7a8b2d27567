// tools/ubench/sort_bench.hip -- the library's radix sort (monogs_amd/csrc/radix_sort.hip, compiled into this program with
// RS_TRACE) on C5-shaped keys: whole-sort time, per-launch time, and where a tile of rs_pass_kernel spends its life.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I monogs_amd/csrc tools/ubench/sort_bench.hip -o tools/ubench/sort_bench
//   tools/ubench/sort_bench depth|tile|few [ballot_rank 0|1] [n] [radix_scanned -1|0|1]
//
// depth: n keys (default 2 M), 23 % of them 0xFFFFFFFF (culled), the rest float bits of a depth in [0.2, 12) -- with the
//        rectangle gather of the last pass.   tile: n (default 5.27 M) 13-bit tile ids emitted rectangle by rectangle.
#define RS_TRACE 1
#include "../../monogs_amd/csrc/radix_sort.hip"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

namespace mgs {
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
}  // namespace mgs

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const bool far = argc > 1 && !strcmp(argv[1], "depthfar");      // depth keys, a few of them beyond 13 107 units: the fourth pass runs
    const bool depth = argc < 2 || !strcmp(argv[1], "depth") || far;
    mgs::g_opt_radix_ballot_rank = argc > 2 ? atoi(argv[2]) : 0;
    mgs::g_opt_radix_scanned = argc > 4 ? atoi(argv[4]) : -1;
    const uint64_t n = argc > 3 && atoll(argv[3]) > 0 ? strtoull(argv[3], nullptr, 10) : (depth ? 2000000ull : 5271297ull);
    const bool few = argc > 1 && !strcmp(argv[1], "few");     // five distinct keys: every wave instruction full of equal digits
    const int bits = depth ? 32 : (few ? 16 : 13);
    std::mt19937 rng(7);
    std::vector<uint32_t> keys(n), vals(n);
    std::vector<uint2> aux(n);
    if (depth) {
        std::uniform_real_distribution<float> z(0.2f, 12.0f), u(0.f, 1.f);
        for (uint64_t i = 0; i < n; ++i) {
            float d = z(rng);
            if (far && (i % 1000) == 7) d *= 5000.f;
            uint32_t b;
            memcpy(&b, &d, 4);
            const bool culled = u(rng) < 0.23f;
            keys[i] = culled ? 0xFFFFFFFFu : b;
            vals[i] = (uint32_t)i;
            aux[i] = culled ? make_uint2(0u, 0u) : make_uint2((uint32_t)(i % 120) | ((uint32_t)(i % 67) << 16), (uint32_t)(1 + i % 5) | ((uint32_t)(1 + i % 3) << 16));
        }
    } else if (few) {
        const uint32_t pick[5] = {0u, 1u, 2u, 257u, 258u};
        std::uniform_int_distribution<int> k(0, 4);
        for (uint64_t i = 0; i < n; ++i) { keys[i] = pick[k(rng)]; vals[i] = (uint32_t)i; }
    } else {
        // rectangles of neighbouring tiles, row-major inside the rectangle, as duplicate_kernel emits them (120 x 68 tiles)
        std::uniform_int_distribution<int> cx(0, 119), cy(0, 67), w(1, 3), h(1, 3);
        uint64_t i = 0;
        uint32_t gi = 0;
        while (i < n) {
            const int x0 = cx(rng), y0 = cy(rng), ww = w(rng), hh = h(rng);
            for (int y = y0; y < std::min(68, y0 + hh) && i < n; ++y)
                for (int x = x0; x < std::min(120, x0 + ww) && i < n; ++x) {
                    keys[i] = (uint32_t)(y * 120 + x);
                    vals[i] = gi;
                    ++i;
                }
            ++gi;
        }
    }
    uint32_t *ka, *va, *kb, *vb, *perm;
    uint2 *aux_in, *aux_out;
    void* temp;
    const bool payload = depth && mgs::radix_depth_payload(n);
    const size_t tb = depth ? mgs::radix_depth_temp_bytes(n) : mgs::radix_temp_bytes(n, bits);
    CK(hipMalloc(&perm, n * 4));
    CK(hipMalloc(&ka, n * 4)); CK(hipMalloc(&va, n * 4)); CK(hipMalloc(&kb, n * 4)); CK(hipMalloc(&vb, n * 4));
    CK(hipMalloc(&aux_in, n * 8)); CK(hipMalloc(&aux_out, n * 8)); CK(hipMalloc(&temp, tb));
    std::vector<uint32_t> packed(n);
    for (uint64_t i = 0; i < n; ++i)
        packed[i] = (aux[i].x & 0xFFu) | ((aux[i].x >> 16) << 8) | ((aux[i].y & 0xFFu) << 16) | ((aux[i].y >> 16) << 24);
    const uint32_t tiles = mgs::rs_tiles(n);
    uint64_t* trace;
    CK(hipMalloc(&trace, (size_t)tiles * 8 * 8));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto upload = [&]() {
        (void)hipMemcpyAsync(ka, keys.data(), n * 4, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(va, vals.data(), n * 4, hipMemcpyHostToDevice, s);
        if (payload) (void)hipMemcpyAsync(aux_in, packed.data(), n * 4, hipMemcpyHostToDevice, s);     // (the sort consumes it)
        else if (depth) (void)hipMemcpyAsync(aux_in, aux.data(), n * 8, hipMemcpyHostToDevice, s);
    };
    auto sort = [&]() {
        if (depth) return mgs::radix_sort_depth(ka, kb, va, vb, aux_in, payload, perm, aux_out, n, temp, s, false);
        return mgs::radix_sort_pairs(ka, va, kb, vb, n, bits, temp, s, nullptr, false, nullptr, nullptr, nullptr, false);
    };
    // ---- correctness against std::stable_sort
    upload();
    if (sort()) return 1;
    CK(hipStreamSynchronize(s));
    const bool in_b = mgs::radix_result_in_b(bits);
    std::vector<uint32_t> gk(n), gv(n);
    if (!depth) CK(hipMemcpy(gk.data(), in_b ? kb : ka, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gv.data(), depth ? perm : (in_b ? vb : va), n * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i) bad += (!depth && gk[i] != keys[order[i]]) || (gv[i] != vals[order[i]]);
    if (depth) {
        std::vector<uint2> ga(n);
        CK(hipMemcpy(ga.data(), aux_out, n * 8, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; ++i) bad += ga[i].x != aux[order[i]].x || ga[i].y != aux[order[i]].y;
    }
    printf("%s sort of %llu pairs, %s ranking, %s: %u tiles, %s\n", depth ? (far ? "depth (some beyond the narrow range)" : (payload ? "depth (payload)" : "depth (gather)")) : (few ? "few-keys" : "tile"), (unsigned long long)n,
           mgs::g_opt_radix_ballot_rank ? "ballot" : "LDS-atomic", mgs::rs_scanned(n) ? "counted tiles" : "one sweep", tiles, bad ? "WRONG" : "matches std::stable_sort");
    if (bad) return 2;
    // ---- time (the input of an even/odd pass count ends where it started or not: re-upload outside the timed region is
    //      not needed for timing purposes -- the passes do the same work on any permutation of these keys -- but the first
    //      pass of a sorted input scatters differently, so the keys are restored every repetition, untimed)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float total = 0.f;
    const int reps = 30;
    for (int r = 0; r < reps + 3; ++r) {
        upload();
        CK(hipEventRecord(e0, s));
        if (sort()) return 1;
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) total += ms;
    }
    printf("whole sort: %.1f us (average of %d)\n", total / reps * 1e3, reps);
    // ---- trace of the LAST pass kernel of one more sort (every pass overwrites the stamps)
    uint64_t* htrace;
    CK(hipMalloc(&htrace, (size_t)tiles * 4 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(mgs::g_rs_trace), &trace, sizeof(trace)));
    CK(hipMemset(htrace, 0, (size_t)tiles * 32));
    if (mgs::rs_scanned(n)) CK(hipMemcpyToSymbol(HIP_SYMBOL(mgs::g_rs_htrace), &htrace, sizeof(htrace)));
    for (int which = 0; which < 2; ++which) {
        // which = 0: trace of the last pass; 1: of the first pass (sort on the low 8 bits only)
        upload();
        CK(hipMemsetAsync(trace, 0, (size_t)tiles * 64, s));
        CK(hipMemsetAsync(htrace, 0, (size_t)tiles * 32, s));
        if (which == 1 && depth) break;     // (the depth plan has no one-pass form)
        int rc = which == 0 ? sort()
                            : mgs::radix_sort_pairs(ka, va, kb, vb, n, 8, temp, s, nullptr, false, nullptr, nullptr, nullptr, false);
        if (rc) return 1;
        CK(hipStreamSynchronize(s));
        std::vector<uint64_t> tr((size_t)tiles * 8);
        CK(hipMemcpy(tr.data(), trace, tr.size() * 8, hipMemcpyDeviceToHost));
        uint64_t t0 = ~0ull, t1 = 0;
        for (uint32_t t = 0; t < tiles; ++t) { t0 = std::min(t0, tr[t * 8]); t1 = std::max(t1, tr[t * 8 + 7]); }
        double ph[8] = {0};
        std::vector<double> starts(tiles), ends(tiles);
        for (uint32_t t = 0; t < tiles; ++t) {
            for (int k = 1; k < 8; ++k) ph[k] += (double)(tr[t * 8 + k] - tr[t * 8 + k - 1]) * 0.01;
            starts[t] = (double)(tr[t * 8] - t0) * 0.01;
            ends[t] = (double)(tr[t * 8 + 7] - t0) * 0.01;
        }
        std::sort(starts.begin(), starts.end());
        std::sort(ends.begin(), ends.end());
        printf("%s pass (stamps cost a drain each): span %.1f us; tile life %.1f us = load %.1f | rank %.1f | digit scan + offsets %.1f | "
               "LDS layout %.1f | store issue %.1f | store drain %.1f | aux %.1f\n",
               which == 0 ? "last" : "first", (double)(t1 - t0) * 0.01,
               (ph[1] + ph[2] + ph[3] + ph[4] + ph[5] + ph[6] + ph[7]) / tiles, ph[1] / tiles, ph[2] / tiles, ph[3] / tiles,
               ph[4] / tiles, ph[5] / tiles, ph[6] / tiles, ph[7] / tiles);
        if (mgs::rs_scanned(n)) {
            std::vector<uint64_t> ht((size_t)tiles * 4);
            CK(hipMemcpy(ht.data(), htrace, ht.size() * 8, hipMemcpyDeviceToHost));
            uint64_t h0 = ~0ull, h1 = 0;
            double hp[4] = {0};
            uint32_t wgs = 0;
            for (uint32_t t = 0; t < tiles; ++t) {
                if (!ht[t * 4]) continue;            // (1024-thread workgroups stamp one slot per four tiles)
                ++wgs;
                h0 = std::min(h0, ht[t * 4]); h1 = std::max(h1, ht[t * 4 + 3]);
                for (int k = 1; k < 4; ++k) hp[k] += (double)(ht[t * 4 + k] - ht[t * 4 + k - 1]) * 0.01;
            }
            for (int k = 1; k < 4; ++k) hp[k] *= (double)tiles / wgs;
            printf("   its histogram kernel: span %.1f us; tile life = load %.1f | LDS count %.1f | store + adds drained %.1f; gap to the scatter kernel's first tile %.1f us\n",
                   (double)(h1 - h0) * 0.01, hp[1] / tiles, hp[2] / tiles, hp[3] / tiles, (double)((int64_t)t0 - (int64_t)h1) * 0.01);
        }
        printf("   tile starts: median %.1f, 90%% %.1f, last %.1f us; tile ends: first %.1f, median %.1f, last %.1f us\n",
               starts[tiles / 2], starts[tiles * 9 / 10], starts[tiles - 1], ends[0], ends[tiles / 2], ends[tiles - 1]);
    }
    return 0;
}
