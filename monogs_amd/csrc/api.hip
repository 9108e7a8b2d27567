// extern "C" entry points of libmonogs_raster.so (see include/monogs_raster.h).
#include "common.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace mgs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- scratch carving -------------------------------------------------------------------------
static inline char* take(char*& p, size_t bytes) {
    char* r = p;
    p += align_up(bytes, 256);
    return r;
}

GeometryState GeometryState::carve(void* base, int P) {
    char* p = (char*)align_up((size_t)base, 256);
    GeometryState g;
    g.rec = (float*)take(p, (size_t)P * REC_FLOATS * sizeof(float));
    g.depth_key = (uint32_t*)take(p, (size_t)P * sizeof(uint32_t));
    g.depth_alt = (uint32_t*)take(p, (size_t)P * sizeof(uint32_t));
    g.iota = (uint32_t*)take(p, (size_t)P * sizeof(uint32_t));
    g.iota_alt = (uint32_t*)take(p, (size_t)P * sizeof(uint32_t));
    g.perm = (uint32_t*)take(p, (size_t)P * sizeof(uint32_t));
    g.point_offsets = (uint32_t*)take(p, (size_t)P * sizeof(uint32_t));
    g.scan_blocks = (uint32_t*)take(p, ((size_t)scan_nblocks(P) + 64) * sizeof(uint32_t));
    g.clamped = (uint8_t*)take(p, (size_t)P * 4);
    g.rect = (uint2*)take(p, (size_t)P * sizeof(uint2));
    g.rect_sorted = (uint2*)take(p, (size_t)P * sizeof(uint2));
    g.scan_status = (uint64_t*)take(p, (SCAN_SMALL_MAX_BLOCKS + 1) * sizeof(uint64_t));
    g.tile_hist = (uint32_t*)take(p, 4 * 256 * sizeof(uint32_t));
    g.sort_temp_bytes = radix_depth_temp_bytes((uint64_t)P);
    g.sort_temp = take(p, g.sort_temp_bytes);           // 256-aligned, directly behind tile_hist
    g.end = p;
    return g;
}
size_t GeometryState::bytes(int P) {
    const GeometryState g = carve(nullptr, P);
    return (size_t)g.end + 256;
}

size_t ImageState::bytes(int W, int H) {
    const size_t HW = (size_t)W * H, nt = (size_t)tiles_x(W) * tiles_y(H);
    return align_up(HW * 4, 256) * 2 + align_up(nt * sizeof(uint2), 256) + 256;
}
ImageState ImageState::carve(void* base, int W, int H) {
    const size_t HW = (size_t)W * H, nt = (size_t)tiles_x(W) * tiles_y(H);
    char* p = (char*)align_up((size_t)base, 256);
    ImageState s;
    s.final_T = (float*)take(p, HW * 4);
    s.n_contrib = (uint32_t*)take(p, HW * 4);
    s.ranges = (uint2*)take(p, nt * sizeof(uint2));
    return s;
}

size_t BinningState::bytes(uint64_t R, int W, int H) {
    const size_t r = (size_t)(R ? R : 1);
    return align_up(r * 4, 256) * 4 + align_up(mgs::sort_temp_bytes(R, tile_bits(W, H)), 256) + 256 + 256;
}
BinningState BinningState::carve(void* base, uint64_t R, int W, int H) {
    const size_t r = (size_t)(R ? R : 1);
    char* p = (char*)align_up((size_t)base, 256);
    BinningState b;
    b.keys_a = (uint32_t*)take(p, r * 4);
    b.keys_b = (uint32_t*)take(p, r * 4);
    b.vals_a = (uint32_t*)take(p, r * 4);
    b.vals_b = (uint32_t*)take(p, r * 4);
    const bool in_b = radix_result_in_b(tile_bits(W, H));
    b.keys_sorted = in_b ? b.keys_b : b.keys_a;
    b.vals_sorted = in_b ? b.vals_b : b.vals_a;
    b.sort_temp_bytes = mgs::sort_temp_bytes(R, tile_bits(W, H));
    b.sort_temp = take(p, b.sort_temp_bytes);
    b.count = (uint32_t*)take(p, 2 * sizeof(uint32_t));
    return b;
}

// ---- per-stage timing with HIP events on the launch stream -------------------------------
struct StageTimer {
    hipStream_t s;
    bool on;
    hipEvent_t ev[10];
    int n = 0;
    StageTimer(hipStream_t s_, bool on_) : s(s_), on(on_) {
        if (on) for (auto& e : ev) (void)hipEventCreate(&e);
    }
    ~StageTimer() {
        if (on) for (auto& e : ev) (void)hipEventDestroy(e);
    }
    void mark() {
        if (on && n < 10) (void)hipEventRecord(ev[n++], s);
    }
    float ms(int i) {   // time between mark i and mark i+1
        float t = 0.f;
        if (on && i + 1 < n) (void)hipEventElapsedTime(&t, ev[i], ev[i + 1]);
        return t;
    }
    void sync() {
        if (on && n) (void)hipEventSynchronize(ev[n - 1]);
    }
};

static int check_cam(const mgs_camera* cam) {
    if (!cam) { set_error("camera is NULL"); return 1; }
    if (cam->image_width <= 0 || cam->image_height <= 0) { set_error("image size must be positive"); return 1; }
    if (!cam->bg || !cam->viewmatrix || !cam->projmatrix || !cam->projmatrix_raw || !cam->campos) {
        set_error("camera tensors (bg, viewmatrix, projmatrix, projmatrix_raw, campos) must be non-NULL");
        return 1;
    }
    return 0;
}

// the blend backward forms `index * 64 + slot` in 32 bits (blend.hip, bt_flush): refuse maps it cannot address
// mgs_debug_set_blend_events: recorded around the blend forward / backward launch of every call while set
static hipEvent_t g_dbg_fwd_events[2] = {nullptr, nullptr}, g_dbg_bwd_events[2] = {nullptr, nullptr};

static int check_map_size(int32_t P) {
    if (P > MGS_MAX_GAUSSIANS) {
        set_error("P exceeds MGS_MAX_GAUSSIANS (2^26 - 1): the gradient-line offset of the blend backward is 32-bit");
        return 1;
    }
    return 0;
}

}  // namespace mgs

using namespace mgs;

extern "C" {

int mgs_abi_version(void) { return MGS_ABI_VERSION; }
const char* mgs_last_error(void) { return g_err; }

size_t mgs_geometry_bytes(int32_t P) { return GeometryState::bytes(P < 0 ? 0 : P); }
size_t mgs_image_bytes(int32_t W, int32_t H) { return ImageState::bytes(W, H); }
size_t mgs_binning_bytes(uint64_t R, int32_t W, int32_t H) { return BinningState::bytes(R, W, H); }
size_t mgs_backward_bytes(int32_t P) {
    return (size_t)(P < 0 ? 0 : P) * GRAD_FLOATS * sizeof(float) + (size_t)(TAU_SLOTS + 1) * 16 * sizeof(float) + 256;
}
float* mgs_backward_tau(void* backward_scratch, int32_t P) { return backward_tau_out(backward_scratch, P < 0 ? 0 : P); }

int mgs_forward_preprocess(const mgs_camera* cam, int32_t P, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, void* geometry, int32_t* radii,
                           void* prepare_backward, uint64_t* num_rendered, const uint32_t* prev_status,
                           uint32_t* prev_status_out, mgs_timing* timing, void* stream) {
    if (check_cam(cam)) return 1;
    if (P < 0) { set_error("P must be >= 0"); return 1; }
    if (check_map_size(P)) return 1;
    if (num_rendered) *num_rendered = 0;        // NULL = capacity mode: no read-back, no stream sync
    if (prev_status_out) *prev_status_out = 0;
    if (P == 0) {
        if (prev_status && prev_status_out) {   // nothing to render, but the caller still wants the earlier forward's verdict
            MGS_HIP(hipMemcpyAsync(prev_status_out, prev_status, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
            MGS_HIP(hipStreamSynchronize((hipStream_t)stream));
        }
        return 0;
    }
    if (!means3D || !opacities || !geometry || !radii) { set_error("means3D, opacities, geometry, radii must be non-NULL"); return 1; }
    if ((shs == nullptr) == (colors_precomp == nullptr)) {
        set_error("Please provide excatly one of either SHs or precomputed colors!");
        return 1;
    }
    const bool have_sr = scales != nullptr || rotations != nullptr;
    if ((have_sr && (!scales || !rotations || cov3D_precomp)) || (!have_sr && !cov3D_precomp)) {
        set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
        return 1;
    }
    if (shs && (cam->sh_degree < 0 || cam->sh_degree > 3 || cam->sh_coeffs < (cam->sh_degree + 1) * (cam->sh_degree + 1))) {
        set_error("sh_degree must be 0..3 and shs must hold at least (degree+1)^2 coefficients");
        return 1;
    }
    hipStream_t s = (hipStream_t)stream;
    GeometryState g = GeometryState::carve(geometry, P);
    StageTimer tm(s, timing != nullptr);
    tm.mark();
    if (int rc = launch_preprocess_forward(*cam, P, means3D, shs, colors_precomp, opacities, scales, rotations,
                                           cov3D_precomp, g, radii,
                                           prepare_backward ? backward_grad_acc(prepare_backward) : nullptr, s)) return rc;
    tm.mark();
    const bool exclusive = (cam->flags & MGS_FLAG_EXCLUSIVE_DEVICE) != 0;
    if (depth_chain_is_small(P)) {          // one single-workgroup launch: sort, rectangle gather and scan (timed as the depth sort)
        if (int rc = launch_depth_chain_small(g, P, s)) return rc;
        tm.mark();
    } else {
        if (int rc = launch_depth_sort(g, P, depth_sort_payload(P, cam->image_width, cam->image_height), s, exclusive)) return rc;
        tm.mark();
        if (int rc = launch_scan(g, P, s, exclusive)) return rc;
    }
    tm.mark();
    if (num_rendered) {
        uint32_t total = 0, sort_errors[RADIX_ERROR_WORDS] = {0, 0, 0, 0};
        MGS_HIP(hipMemcpyAsync(&total, g.scan_blocks + scan_nblocks(P), sizeof(uint32_t), hipMemcpyDeviceToHost, s));   // grand total
        MGS_HIP(hipMemcpyAsync(sort_errors, radix_depth_error_flag(g.sort_temp, (uint64_t)P), sizeof(sort_errors),
                               hipMemcpyDeviceToHost, s));
        // the status word of an EARLIER forward on this stream rides along: its kernels are done by the time this copy runs,
        // so a tile-sort timeout of the exact path is seen one forward later at no extra synchronisation
        if (prev_status && prev_status_out)
            MGS_HIP(hipMemcpyAsync(prev_status_out, prev_status, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        MGS_HIP(hipStreamSynchronize(s));
        if (sort_errors[0] | sort_errors[1] | sort_errors[2] | sort_errors[3]) { set_error("depth sort: a look-back spin timed out (results invalid)"); return 3; }
        *num_rendered = total;
    }
    if (timing) {
        if (!num_rendered) MGS_HIP(hipStreamSynchronize(s));
        timing->preprocess_ms = tm.ms(0);
        timing->depth_sort_ms = tm.ms(1);
        timing->scan_ms = tm.ms(2);
    }
    return 0;
}

static int forward_render_impl(const mgs_camera* cam, int32_t P, uint64_t R, bool capacity, void* geometry,
                               void* binning, void* image, float* out_color, float* out_depth, float* out_opacity,
                               int32_t* n_touched, uint32_t* overflow, mgs_timing* timing, void* stream) {
    if (check_cam(cam)) return 1;
    if (P < 0) { set_error("P must be >= 0"); return 1; }
    if (check_map_size(P)) return 1;
    if (!image || !out_color || !out_depth || !out_opacity) { set_error("image scratch and outputs must be non-NULL"); return 1; }
    if (P > 0 && (!geometry || !n_touched)) { set_error("geometry and n_touched must be non-NULL"); return 1; }
    if (R > 0 && !binning) { set_error("binning scratch is NULL"); return 1; }
    if (R >= (1ull << 32)) { set_error("num_rendered exceeds 2^32-1"); return 1; }
    // No Gaussians: nothing was preprocessed (geometry may be NULL, its digit table was never cleared), so nothing may be
    // sorted whatever capacity the caller passed -- the kernels below then only clear the ranges and render the background.
    if (P == 0) R = 0;
    hipStream_t s = (hipStream_t)stream;
    const int W = cam->image_width, H = cam->image_height;
    GeometryState g = GeometryState::carve(geometry, P);
    ImageState img = ImageState::carve(image, W, H);
    BinningState b = BinningState::carve(binning, R, W, H);
    const uint32_t* n_dev = nullptr;
    StageTimer tm(s, timing != nullptr);
    const bool cap = capacity && binning;
    if (cap) n_dev = b.count;
    tm.mark();
    // (also zeroes n_touched, the tile ranges and the scratch of the tile sort; in capacity mode it publishes the
    //  clamped live count and the overflow flag; with R == 0 it emits nothing)
    if (int rc = launch_duplicate(*cam, P, g, b, capacity ? R : (R > 0 ? 0xFFFFFFFFull : 0ull), n_touched, img, R,
                                  tile_bits(W, H), cap ? b.count : nullptr, overflow, s)) return rc;
    tm.mark();
    // (the sort's final pass writes the per-tile ranges: no ranges launch since round 4)
    if (int rc = launch_sort(g, b, R, tile_bits(W, H), s, n_dev, (cam->flags & MGS_FLAG_EXCLUSIVE_DEVICE) != 0, img.ranges)) return rc;
    tm.mark();
    if (g_dbg_fwd_events[0]) MGS_HIP(hipEventRecord(g_dbg_fwd_events[0], s));
    if (int rc = launch_blend_forward(*cam, g, b, img, out_color, out_depth, out_opacity, n_touched,
                                      R > 0 ? radix_error_flag(b.sort_temp, R, tile_bits(W, H)) : nullptr, overflow, s)) return rc;
    if (g_dbg_fwd_events[1]) MGS_HIP(hipEventRecord(g_dbg_fwd_events[1], s));
    tm.mark();
    if (timing) {
        tm.sync();
        timing->duplicate_ms = tm.ms(0);
        timing->sort_ms = tm.ms(1);
        timing->ranges_ms = 0.f;              // (no ranges launch since round 4: the tile sort's final pass writes them)
        timing->blend_fwd_ms = tm.ms(2);
    }
    return 0;
}

int mgs_forward_render(const mgs_camera* cam, int32_t P, uint64_t R, void* geometry, void* binning, void* image,
                       float* out_color, float* out_depth, float* out_opacity, int32_t* n_touched,
                       uint32_t* status, mgs_timing* timing, void* stream) {
    return forward_render_impl(cam, P, R, false, geometry, binning, image, out_color, out_depth, out_opacity, n_touched,
                               status, timing, stream);
}

int mgs_forward_render_capacity(const mgs_camera* cam, int32_t P, uint64_t capacity, void* geometry, void* binning,
                                void* image, float* out_color, float* out_depth, float* out_opacity,
                                int32_t* n_touched, uint32_t* overflow, mgs_timing* timing, void* stream) {
    if (capacity == 0) { set_error("capacity must be > 0"); return 1; }
    return forward_render_impl(cam, P, capacity, true, geometry, binning, image, out_color, out_depth, out_opacity,
                               n_touched, overflow, timing, stream);
}

int mgs_forward_capacity(const mgs_camera* cam, int32_t P, const float* means3D, const float* shs,
                         const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                         const float* cov3D_precomp, void* geometry, int32_t* radii, void* prepare_backward,
                         uint64_t capacity, void* binning, void* image, float* out_color, float* out_depth,
                         float* out_opacity, int32_t* n_touched, uint32_t* overflow, mgs_timing* timing, void* stream) {
    if (int rc = mgs_forward_preprocess(cam, P, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                        geometry, radii, prepare_backward, nullptr, nullptr, nullptr, timing, stream)) return rc;
    return mgs_forward_render_capacity(cam, P, capacity, geometry, binning, image, out_color, out_depth, out_opacity,
                                       n_touched, overflow, timing, stream);
}

int mgs_backward(const mgs_camera* cam, int32_t P, uint64_t R, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                 const float* cov3D_precomp, const int32_t* radii, const void* geometry, const void* binning,
                 const void* image, const float* dL_dcolor, const float* dL_ddepth, float* dL_dmeans2D,
                 float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscales, float* dL_drotations, float* dL_dtau, void* backward_scratch,
                 int32_t scratch_prepared, mgs_timing* timing, void* stream) {
    if (check_cam(cam)) return 1;
    if (P < 0) { set_error("P must be >= 0"); return 1; }
    if (check_map_size(P)) return 1;
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) {
        if (dL_dtau) MGS_HIP(zero_fill(dL_dtau, 6 * sizeof(float), s));
        return 0;
    }
    if (!means3D || !opacities || !radii || !geometry || !image || !dL_dcolor || !dL_ddepth || !backward_scratch) {
        set_error("means3D, opacities, radii, geometry, image, dL_dcolor, dL_ddepth, backward_scratch must be non-NULL");
        return 1;
    }
    if (R > 0 && !binning) { set_error("binning scratch is NULL"); return 1; }
    const int W = cam->image_width, H = cam->image_height;
    GeometryState g = GeometryState::carve(const_cast<void*>(geometry), P);
    ImageState img = ImageState::carve(const_cast<void*>(image), W, H);
    BinningState b = BinningState::carve(const_cast<void*>(binning), R, W, H);
    float* grad_acc = backward_grad_acc(backward_scratch);
    StageTimer tm(s, timing != nullptr);
    // (the pose-gradient slots sit right behind the accumulator: one clear covers both)
    float* tau_part = (dL_dtau && (P + 255) / 256 > TAU_DIRECT_MAX_BLOCKS) ? backward_tau_part(backward_scratch, P) : nullptr;
    if (scratch_prepared) {
        // the matching forward cleared the lines of the visible Gaussians, the slots and the six floats of dL/dtau
        if (dL_dtau && dL_dtau != backward_tau_out(backward_scratch, P)) {
            set_error("scratch_prepared: dL_dtau must be mgs_backward_tau(backward_scratch, P)");
            return 1;
        }
    } else {
        MGS_HIP(zero_fill2(grad_acc, (size_t)P * GRAD_FLOATS * sizeof(float) + (tau_part ? (size_t)TAU_SLOTS * 16 * sizeof(float) : 0),
                           dL_dtau, 6 * sizeof(float), s));
    }
    tm.mark();
    if (R > 0) {
        // colours / opacities take no gradient and do not feed the geometry (no SH): the lighter blend backward
        const bool pose_only = !dL_dcolors && !dL_dopacity && !dL_dsh && !shs;
        if (g_dbg_bwd_events[0]) MGS_HIP(hipEventRecord(g_dbg_bwd_events[0], s));
        if (int rc = launch_blend_backward(*cam, g, b, img, dL_dcolor, dL_ddepth, grad_acc, pose_only, s)) return rc;
        if (g_dbg_bwd_events[1]) MGS_HIP(hipEventRecord(g_dbg_bwd_events[1], s));
    }
    tm.mark();
    GeomBackwardArgs a;
    a.means3D = means3D; a.shs = shs; a.colors_precomp = colors_precomp; a.opacities = opacities;
    a.scales = scales; a.rotations = rotations; a.cov3D_precomp = cov3D_precomp; a.radii = radii;
    a.grad_acc = grad_acc;
    a.dL_dmeans2D = dL_dmeans2D; a.dL_dcolors = dL_dcolors; a.dL_dopacity = dL_dopacity;
    a.dL_dmeans3D = dL_dmeans3D; a.dL_dcov3D = dL_dcov3D; a.dL_dsh = dL_dsh; a.dL_dscales = dL_dscales;
    a.dL_drotations = dL_drotations; a.dL_dtau = dL_dtau; a.tau_part = tau_part;
    if (int rc = launch_geom_backward(*cam, P, g, a, s)) return rc;
    tm.mark();
    if (timing) {
        tm.sync();
        timing->blend_bwd_ms = tm.ms(0);
        timing->geom_bwd_ms = tm.ms(1);
    }
    return 0;
}

int mgs_debug_blend_stats(const mgs_camera* cam, int32_t P, uint64_t R, const void* geometry, const void* binning,
                           const void* image, uint64_t* stats_dev, void* stream) {
    if (check_cam(cam)) return 1;
    if (!geometry || !image || !stats_dev || (R > 0 && !binning)) { set_error("bad arguments"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    MGS_HIP(zero_fill(stats_dev, MGS_BLEND_STATS_WORDS * sizeof(uint64_t), s));
    if (P == 0 || R == 0) return 0;
    const int W = cam->image_width, H = cam->image_height;
    GeometryState g = GeometryState::carve(const_cast<void*>(geometry), P);
    ImageState img = ImageState::carve(const_cast<void*>(image), W, H);
    BinningState b = BinningState::carve(const_cast<void*>(binning), R, W, H);
    return launch_blend_backward_stats(*cam, g, b, img, (unsigned long long*)stats_dev, s);
}

int mgs_debug_valu_ceiling(float* out, int32_t iters, void* stream) {
    if (!out || iters < 1) { set_error("mgs_debug_valu_ceiling: bad arguments"); return 1; }
    return launch_valu_ceiling(out, iters, (hipStream_t)stream);
}

int mgs_debug_set_radix_spin_limit(uint32_t limit) { return set_radix_spin_limit(limit); }

int mgs_debug_set_blend_events(void* fwd_start, void* fwd_end, void* bwd_start, void* bwd_end) {
    if ((fwd_start == nullptr) != (fwd_end == nullptr) || (bwd_start == nullptr) != (bwd_end == nullptr)) {
        set_error("mgs_debug_set_blend_events: a pair is both events or neither");
        return 1;
    }
    g_dbg_fwd_events[0] = (hipEvent_t)fwd_start; g_dbg_fwd_events[1] = (hipEvent_t)fwd_end;
    g_dbg_bwd_events[0] = (hipEvent_t)bwd_start; g_dbg_bwd_events[1] = (hipEvent_t)bwd_end;
    return 0;
}

static int g_opt_debug_sort_exclusive = 0;   // mgs_debug_set_option("debug_sort_exclusive", 1): mgs_debug_sort_pairs sorts as under
                                             // MGS_FLAG_EXCLUSIVE_DEVICE (block ids as tile ids)
size_t mgs_debug_sort_temp_bytes(uint64_t n, int32_t bits) { return radix_temp_bytes(n, bits); }
int mgs_debug_sort_pairs(uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt, uint64_t n, int32_t bits,
                         void* temp, void* stream) {
    if (bits < 1 || bits > 32 || (n && (!keys || !vals || !keys_alt || !vals_alt || !temp))) {
        set_error("mgs_debug_sort_pairs: bad arguments");
        return 1;
    }
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = radix_sort_pairs(keys, vals, keys_alt, vals_alt, n, bits, temp, s, nullptr, false, nullptr, nullptr, nullptr, false,
                                  g_opt_debug_sort_exclusive != 0))
        return rc;
    if (radix_result_in_b(bits)) {
        MGS_HIP(hipMemcpyAsync(keys, keys_alt, n * 4, hipMemcpyDeviceToDevice, s));
        MGS_HIP(hipMemcpyAsync(vals, vals_alt, n * 4, hipMemcpyDeviceToDevice, s));
    }
    uint32_t err[RADIX_ERROR_WORDS];
    MGS_HIP(hipMemcpyAsync(err, radix_error_flag(temp, n, bits), sizeof(err), hipMemcpyDeviceToHost, s));
    MGS_HIP(hipStreamSynchronize(s));
    if (err[0] | err[1] | err[2] | err[3]) { set_error("mgs_debug_sort_pairs: a look-back spin timed out"); return 2; }
    return 0;
}

int mgs_debug_set_option(const char* name, int64_t value) {
    if (name && !strcmp(name, "radix_scanned")) { g_opt_radix_scanned = (int)value; return 0; }
    if (name && !strcmp(name, "radix_ballot_rank")) { g_opt_radix_ballot_rank = (int)value; return 0; }
    if (name && !strcmp(name, "radix_tile_items")) { g_opt_radix_tile_items = (int)value; return 0; }
    if (name && !strcmp(name, "radix_xcd_band")) { g_opt_radix_xcd_band = (int)value; return 0; }
    if (name && !strcmp(name, "debug_sort_exclusive")) { g_opt_debug_sort_exclusive = (int)value; return 0; }
    if (name && !strcmp(name, "dup_slot_major")) { g_opt_dup_slot_major = (int)value; return 0; }
    if (name && !strcmp(name, "blend_bwd_transposed")) { g_opt_blend_bwd_transposed = (int)value; return 0; }
    if (name && !strcmp(name, "scan_small")) { g_opt_scan_small = (int)value; return 0; }
    if (name && !strcmp(name, "depth_small")) { g_opt_depth_small = (int)value; return 0; }
    if (name && !strcmp(name, "blend_lds_pad_fwd")) { g_opt_blend_lds_pad_fwd = (int)value; return 0; }
    if (name && !strcmp(name, "blend_lds_pad_bwd")) { g_opt_blend_lds_pad_bwd = (int)value; return 0; }
    if (name && !strcmp(name, "knn_grid_min")) { g_opt_knn_grid_min = value > 0x7FFFFFFF ? 0x7FFFFFFF : (int)value; return 0; }
    set_error("mgs_debug_set_option: unknown option");
    return 1;
}

int mgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* visible, void* stream) {
    (void)projmatrix;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !visible))) { set_error("bad arguments"); return 1; }
    return launch_mark_visible(P, means3D, viewmatrix, visible, (hipStream_t)stream);
}

size_t mgs_knn_scratch_bytes(int32_t P) { return knn_scratch_bytes(P); }

int mgs_dist2_knn(int32_t P, const float* points, float* out, void* scratch, void* stream) {
    if (P < 0 || (P > 0 && (!points || !out))) { set_error("bad arguments"); return 1; }
    return launch_knn(P, points, out, scratch, (hipStream_t)stream);
}

}  // extern "C"
