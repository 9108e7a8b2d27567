// tools/ubench/depth_small_bench.hip -- the single-workgroup depth chain of small maps (depth_chain_small_kernel, binning.hip,
// compiled into this program with DS_TRACE) alone: checks perm / offsets against std::stable_sort, times the launch, and prints
// where the workgroup spends its life (100 MHz stamps at the phase boundaries).
//   tools/ubench/depth_small_bench [n = 20000]
#define DS_TRACE 1
#include "../../monogs_amd/csrc/radix_sort.hip"
#include "../../monogs_amd/csrc/binning.hip"

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
    const int n = argc > 1 ? atoi(argv[1]) : 20000;
    if (n < 1 || n > mgs::DS_SMALL_MAX) { fprintf(stderr, "n must be 1..%d\n", mgs::DS_SMALL_MAX); return 1; }
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> z(0.2f, 8.0f), u(0.f, 1.f);
    std::vector<uint32_t> keys(n);
    std::vector<uint2> rect(n);
    for (int i = 0; i < n; ++i) {
        float d = z(rng);
        uint32_t b;
        memcpy(&b, &d, 4);
        const bool culled = u(rng) < 0.2f;
        keys[i] = culled ? 0xFFFFFFFFu : b;
        rect[i] = culled ? make_uint2(0u, 0u) : make_uint2((uint32_t)(i % 40) | ((uint32_t)(i % 30) << 16), (uint32_t)(1 + i % 5) | ((uint32_t)(1 + i % 3) << 16));
    }
    uint32_t *dk, *perm, *off, *tot;
    uint2 *dr, *rs;
    unsigned long long* tr;
    CK(hipMalloc(&dk, n * 4)); CK(hipMalloc(&perm, n * 4)); CK(hipMalloc(&off, n * 4)); CK(hipMalloc(&tot, 4));
    CK(hipMalloc(&dr, n * 8)); CK(hipMalloc(&rs, n * 8)); CK(hipMalloc(&tr, 32 * 8));
    CK(hipMemcpy(dk, keys.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dr, rect.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 20; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mgs::depth_chain_small_kernel, dim3(1), dim3(mgs::DS_THREADS), 0, 0, dk, dr, perm, rs, off, tot, n, tr);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<uint32_t> hp(n), ho(n);
    std::vector<unsigned long long> ht(32);
    uint32_t htot;
    CK(hipMemcpy(hp.data(), perm, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ho.data(), off, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&htot, tot, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ht.data(), tr, 32 * 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> ref(n);
    std::iota(ref.begin(), ref.end(), 0u);
    std::stable_sort(ref.begin(), ref.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    uint32_t run = 0;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        run += (rect[ref[i]].y & 0xFFFFu) * (rect[ref[i]].y >> 16);
        if (hp[i] != ref[i] || ho[i] != run) ++bad;
    }
    printf("n = %d: %s (mismatches %d, total %u vs %u); launch %.1f us (best of 20, events)\n", n, bad || htot != run ? "WRONG" : "ok", bad, htot, run, best * 1e3);
    const char* names[4] = {"count+rank", "column scan", "scatter", ""};
    printf("load                      %6.2f us\n", (ht[0] - 0) * 0.0 + 0.0);
    for (int p = 0; p < 4; ++p)
        printf("pass %d: count+rank %6.2f  column scan %6.2f  scatter %6.2f us\n", p, (ht[1 + 4 * p] - ht[p ? 4 * p - 1 : 0]) * 0.01,
               (ht[2 + 4 * p] - ht[1 + 4 * p]) * 0.01, (ht[3 + 4 * p] - ht[2 + 4 * p]) * 0.01);
    printf("perm + rectangle gather   %6.2f us\nscan + stores             %6.2f us\n", (ht[17] - ht[15]) * 0.01, (ht[18] - ht[17]) * 0.01);
    (void)names;
    return bad ? 2 : 0;
}
