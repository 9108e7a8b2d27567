/* monogs_raster.h -- C ABI of libmonogs_raster.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the two native operators MonoGS calls:
 *
 *   diff_gaussian_rasterization   imported at /root/reference/gaussian_splatting/gaussian_renderer/__init__.py:13-16,
 *                                 settings built at :70-84, called at :130-156
 *   simple_knn._C.distCUDA2       imported at /root/reference/gaussian_splatting/scene/gaussian_model.py:18,
 *                                 called at :294-302
 *
 * The reference binds these through pybind11 + torch::Tensor (there is no C plugin ABI upstream; the
 * sources are un-vendored submodules, /root/reference/.gitmodules:1-6).  This header is the C-level
 * contract of the same entry points: plain device pointers, sizes and a HIP stream; no torch types.
 * The entry points correspond 1:1 to the upstream extension functions listed in SURVEY.md section 8b:
 *
 *   rasterize_gaussians            -> mgs_forward_preprocess + mgs_forward_render
 *   rasterize_gaussians_backward   -> mgs_backward
 *   mark_visible                   -> mgs_mark_visible
 *   distCUDA2                      -> mgs_dist2_knn
 *
 * Conventions
 *   - every pointer is DEVICE memory on the current HIP device unless marked [host];
 *   - float tensors are contiguous float32; matrices are the 4x4 tensors MonoGS passes, i.e. the
 *     TRANSPOSE of the maths matrix (flat element 4*j+i is maths element (i,j));
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - every function returns 0 on success and a non-zero code on failure; mgs_last_error() gives the
 *     message of the calling thread's last failure;
 *   - the library keeps no state between calls: scratch lives in caller-owned buffers whose sizes come
 *     from the mgs_*_bytes functions, so several forwards may precede one backward
 *     (/root/reference/utils/slam_mapper.py:273-394) and several processes may share a GPU.
 */
#ifndef MONOGS_RASTER_H
#define MONOGS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGS_ABI_VERSION 9
#define MGS_TILE 16 /* tile edge in pixels; ranges are per 16x16 tile */

/* GaussianRasterizationSettings, minus `prefiltered` / `debug` which are call flags
 * (/root/reference/gaussian_splatting/gaussian_renderer/__init__.py:70-84). */
typedef struct mgs_camera {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t sh_degree;           /* active SH degree (0..3); ignored with colors_precomp */
    int32_t sh_coeffs;           /* M: coefficient triplets per Gaussian in `shs` (0 with colors_precomp) */
    int32_t scale_dim;           /* floats per Gaussian in `scales` / `dL_dscales`: 0 or 3 = (sx, sy, sz); 1 = isotropic --
                                    the expansion render() does with scales.repeat(1, 3)
                                    (/root/reference/gaussian_splatting/gaussian_renderer/__init__.py:101-104) and the
                                    sum of its backward happen inside the kernels */
    int32_t flags;               /* MGS_FLAG_* bits, 0 = none */
    const float* bg;             /* [3]  */
    const float* viewmatrix;     /* [16] transposed world->camera */
    const float* projmatrix;     /* [16] transposed P @ T_cw */
    const float* projmatrix_raw; /* [16] transposed P */
    const float* campos;         /* [3]  */
} mgs_camera;

/* mgs_camera.flags.
 * MGS_FLAG_EXCLUSIVE_DEVICE: the caller guarantees that NO other grid runs beside this call's kernels -- one process on the
 * device, one stream (a pose-tracking loop, a captured single-stream iteration).  The small radix sorts (<= 256 tiles) and
 * the single-launch scan then take their tile / block ids from the block index, which saves a returning atomic (~2 us) per
 * launch.  Without the flag -- the default, and what MonoGS's topology needs: tracker, mapper and viewer processes share one GPU
 * (/root/reference/slam.py:102-179), and a mapping window renders its keyframes on a stream each -- every such launch hands out
 * its ids by an atomic ticket, so a workgroup only ever waits for workgroups that have already started: forward progress does not
 * depend on where, or in which order, the hardware places workgroups of competing grids. */
#define MGS_FLAG_EXCLUSIVE_DEVICE 1

/* Per-stage device time in milliseconds, measured with HIP events on `stream`.
 * Passing a non-NULL mgs_timing makes the call synchronise `stream` before returning. */
typedef struct mgs_timing {
    float preprocess_ms;
    float depth_sort_ms;
    float scan_ms;
    float duplicate_ms;
    float sort_ms;
    float ranges_ms;      /* always 0 since ABI v8: the tile sort's final pass writes the ranges (kept for layout compatibility) */
    float blend_fwd_ms;
    float blend_bwd_ms;
    float geom_bwd_ms;
} mgs_timing;

int mgs_abi_version(void);
const char* mgs_last_error(void);

/* Largest map one call takes.  The blend backward addresses a Gaussian's 64-byte gradient line with a 32-bit byte offset
 * (index x 64 + slot), which wraps at 2^26 Gaussians; the entry points that take P return 1 ("P exceeds ...") beyond it
 * instead of wrapping.  (A 288 GB device could hold such a map; MonoGS maps are 10^4 .. 10^6.) */
#define MGS_MAX_GAUSSIANS ((1 << 26) - 1)

/* Scratch sizes (bytes).  geometry: per-Gaussian state carried from forward to backward;
 * image: per-pixel and per-tile state; binning: keys / values / sort temp for R = num_rendered.
 * LIFETIME (caller-owned, the library keeps nothing): geometry, image, binning and the backward scratch must stay
 * allocated and unmodified from the forward until the LAST backward through it has run; radii / n_touched / the three
 * images are plain outputs and need not outlive anything.  The Python binding (monogs_amd/rasterizer.py) carries the
 * scratch in the autograd context -- about 250 B per Gaussian + 8 B per pixel + 16 B per instance, alive exactly as long as
 * the graph of that forward -- and hands out radii / n_touched as views of their own 8 P-byte tensor and the images as views
 * of one 20 HW-byte tensor, so keeping an OUTPUT (render_pkg["radii"], a keyframe's n_touched) never pins the scratch. */
size_t mgs_geometry_bytes(int32_t P);
size_t mgs_image_bytes(int32_t width, int32_t height);
size_t mgs_binning_bytes(uint64_t num_rendered, int32_t width, int32_t height);
size_t mgs_backward_bytes(int32_t P);
/* The six floats inside a backward scratch that a PREPARED backward (below) accumulates dL/dtau into. */
float* mgs_backward_tau(void* backward_scratch, int32_t P);

/* Forward, stage 1: per-Gaussian projection (cull, covariance, radius, tile rectangle, colour), the depth
 * order of the Gaussians and the prefix sum of tiles touched in that order.  Writes radii[P] and the geometry scratch, then copies the total number
 * of (Gaussian, tile) instances to *num_rendered [host] -- this synchronises `stream`, exactly as the
 * upstream forward does, because the caller must size the binning scratch from it.
 * Exactly one of (shs, colors_precomp) and exactly one of ((scales, rotations), cov3D_precomp) is non-NULL.
 * `prepare_backward`: NULL, or the scratch (mgs_backward_bytes(P)) of the ONE backward that will follow this forward.
 * The per-Gaussian kernel then also clears what that backward accumulates into -- the 64-byte gradient line of every
 * VISIBLE Gaussian (the others are never touched), the pose-gradient slots and the six floats at
 * mgs_backward_tau(scratch, P) -- and mgs_backward may be called with scratch_prepared = 1: no clearing launch and no
 * 64 B x P fill per backward. */
int mgs_forward_preprocess(const mgs_camera* cam, int32_t P,
                           const float* means3D,        /* [P,3] */
                           const float* shs,            /* [P,M,3] or NULL */
                           const float* colors_precomp, /* [P,3] or NULL */
                           const float* opacities,      /* [P] */
                           const float* scales,         /* [P,3] or NULL */
                           const float* rotations,      /* [P,4] or NULL */
                           const float* cov3D_precomp,  /* [P,6] or NULL */
                           void* geometry, int32_t* radii /* [P] */,
                           void* prepare_backward /* backward scratch to clear, or NULL */,
                           uint64_t* num_rendered /* [host]; NULL = capacity mode, no sync */,
                           const uint32_t* prev_status /* device, optional: status word of an EARLIER forward on this stream */,
                           uint32_t* prev_status_out /* [host], optional: receives *prev_status at this call's own
                                                        synchronisation (num_rendered != NULL): a sort timeout of the exact
                                                        path cannot pass unseen, and costs no synchronisation of its own */,
                           mgs_timing* timing /* [host] or NULL */, void* stream);

/* Forward, stage 2: duplicate (in depth order), stable grouping by tile, per-tile ranges, front-to-back blend.
 * Outputs: color[3,H,W], depth[1,H,W] (sum z.alpha.T), opacity[1,H,W] (1 - T), n_touched[P]
 * (zeroed here, then incremented per pixel where the Gaussian is blended with T.(1-alpha) > 0.5). */
int mgs_forward_render(const mgs_camera* cam, int32_t P, uint64_t num_rendered,
                       void* geometry, void* binning, void* image,
                       float* out_color, float* out_depth, float* out_opacity, int32_t* n_touched,
                       uint32_t* status /* device, optional: MGS_STATUS_* bits of this forward */,
                       mgs_timing* timing, void* stream);

/* Status word of a forward (device uint32, written by the forward's own kernels; 0 = clean).  The hand-written radix
 * sort bounds every inter-workgroup wait (small sorts only: a tile waits for the digit counts of earlier tiles); a wait
 * that runs out raises a flag instead of hanging, and the sorted order (hence the blend order) is then invalid.  mgs_forward_preprocess reports a depth-sort timeout itself when it
 * synchronises (num_rendered != NULL); everything else arrives here and is read by the caller at a sync point of its
 * choice (monogs_amd.rasterizer.check_overflow). */
#define MGS_STATUS_CAPACITY_OVERFLOW 1u   /* capacity mode: instances were dropped */
#define MGS_STATUS_DEPTH_SORT_TIMEOUT 2u  /* a bounded wait of the depth sort ran out */
#define MGS_STATUS_TILE_SORT_TIMEOUT 4u   /* a bounded wait of the tile sort ran out */

/* ---- mgs_debug_*: TEST AND MEASUREMENT USE ONLY -------------------------------------------------------------------------
 * "The library keeps no state between calls" (above) holds for every entry point a caller of the rasteriser needs.  The three
 * setters below are the exception, on purpose: each writes a PROCESS-GLOBAL variable that later calls read unsynchronised.
 * They are NOT THREAD-SAFE (set them while no other thread is inside the library), they affect every later call of the process
 * whatever its device or stream, and nothing on MonoGS's path calls them: the test suite and the tools under tools/ do, to force
 * an algorithm path or to time a kernel.  Defaults are restored by the values given with each. */
/* TEST USE ONLY, process-global, not thread-safe: the bound of those waits, in polls (all later sorts); 0xFFFFFFFF restores the default. */
int mgs_debug_set_radix_spin_limit(uint32_t limit);
/* TEST USE ONLY, process-global, not thread-safe: knobs that force an algorithm path whatever the problem size (-1 restores the default):
 * "radix_scanned" (0 = one kernel per pass with a gather of the earlier tiles' counts, 1 = counted tiles: two kernels per
 * pass, no waiting between workgroups -- honoured from 64 k pairs), "radix_ballot_rank" (1 = rank with wave ballots instead of
 * returning LDS atomics: the reference the sort tests compare with), "scan_small" (0 = the two-launch scan at every size), "depth_small" (1 = maps <= 24 576 Gaussians run
 * depth sort + rectangle gather + scan as ONE single-workgroup launch: measured slower, off by default), "blend_lds_pad_fwd" / "blend_lds_pad_bwd" (dynamic
 * LDS bytes the blend kernels never touch: fewer workgroups per compute unit, a measurement knob),
 * "dup_slot_major" (0 / 1 = the duplicate kernel's emission balanced by Gaussians / by output slots at every size),
 * "knn_grid_min" (Morton-box kNN from this many points), "blend_bwd_transposed" (2 = default, 1 = round 3's transposed blend
 * backward fed by per-survivor scalar loads, 0 = the per-survivor wave reduction),
 * "radix_xcd_band" (0 = counted tiles in block-id order instead of one contiguous band of tiles per XCD), "radix_tile_items"
 * (8 | 12 | 16 pairs per thread on the counted-tiles path, 0 = by size; set it before any scratch is sized),
 * "debug_sort_exclusive" (1 = mgs_debug_sort_pairs sorts as under MGS_FLAG_EXCLUSIVE_DEVICE).  Nothing on the launch path
 * consults the environment. */
int mgs_debug_set_option(const char* name, int64_t value);

/* Test entry: the library's stable radix sort of n (key, value) pairs on key bits [0, bits) -- what the forward runs on
 * the depth keys and on the tile ids -- on caller-provided DEVICE buffers: keys / vals hold the input and receive the
 * sorted pairs, keys_alt / vals_alt (n words each) are the ping-pong partners, temp (mgs_debug_sort_temp_bytes(n, bits)
 * bytes) the scratch.  Returns non-zero with mgs_last_error() set if a look-back spin timed out. */
size_t mgs_debug_sort_temp_bytes(uint64_t n, int32_t bits);
int mgs_debug_sort_pairs(uint32_t* keys, uint32_t* vals, uint32_t* keys_alt, uint32_t* vals_alt, uint64_t n, int32_t bits,
                         void* temp, void* stream);

/* Forward, stage 2 without a host-side instance count ("capacity mode"): call mgs_forward_preprocess with
 * num_rendered = NULL (no read-back, no stream sync), size the binning scratch with
 * mgs_binning_bytes(capacity, W, H) for a caller-chosen capacity (e.g. 1.5x the previous frame's count), and the
 * kernels read the live count min(R, capacity) on the device.  If R > capacity the surplus instances are dropped
 * and MGS_STATUS_CAPACITY_OVERFLOW is set in *overflow (device uint32, optional; the forward's status word, which
 * also receives the sort-timeout bits): the caller must re-render with a larger capacity.  The
 * whole forward + backward then contains no host synchronisation and can be captured in a hipGraph.
 * Pass the same `capacity` as num_rendered to mgs_backward. */
int mgs_forward_render_capacity(const mgs_camera* cam, int32_t P, uint64_t capacity,
                                void* geometry, void* binning, void* image,
                                float* out_color, float* out_depth, float* out_opacity, int32_t* n_touched,
                                uint32_t* overflow, mgs_timing* timing, void* stream);
/* Both stages of a capacity-mode forward in ONE call (mgs_forward_preprocess with num_rendered = NULL, then
 * mgs_forward_render_capacity): what an eager caller pays per crossing of the FFI boundary matters at SLAM sizes. */
int mgs_forward_capacity(const mgs_camera* cam, int32_t P, const float* means3D, const float* shs,
                         const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                         const float* cov3D_precomp, void* geometry, int32_t* radii, void* prepare_backward,
                         uint64_t capacity, void* binning, void* image, float* out_color, float* out_depth,
                         float* out_opacity, int32_t* n_touched, uint32_t* overflow, mgs_timing* timing, void* stream);

/* Backward.  Consumes dL/dcolor[3,H,W] and dL/ddepth[1,H,W] (dL/dopacity is ignored, as upstream) and
 * the scratch of the matching forward.  Any output pointer may be NULL (that gradient is then not
 * stored); dL_dtau is [6] = (rho, theta), already summed over Gaussians.
 * scratch_prepared = 1: `backward_scratch` was handed to the matching mgs_forward_preprocess as prepare_backward and has
 * not been used by a backward since; dL_dtau must then be NULL or mgs_backward_tau(backward_scratch, P). */
int mgs_backward(const mgs_camera* cam, int32_t P, uint64_t num_rendered,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* opacities, const float* scales, const float* rotations,
                 const float* cov3D_precomp, const int32_t* radii,
                 const void* geometry, const void* binning, const void* image,
                 const float* dL_dcolor, const float* dL_ddepth,
                 float* dL_dmeans2D,  /* [P,3] NDC-scaled x,y; z = 0 */
                 float* dL_dcolors,   /* [P,3] (colors_precomp) */
                 float* dL_dopacity,  /* [P]   */
                 float* dL_dmeans3D,  /* [P,3] */
                 float* dL_dcov3D,    /* [P,6] (cov3D_precomp) */
                 float* dL_dsh,       /* [P,M,3] (shs) */
                 float* dL_dscales,   /* [P,3] */
                 float* dL_drotations,/* [P,4] */
                 float* dL_dtau,      /* [6]   */
                 void* backward_scratch, int32_t scratch_prepared, mgs_timing* timing, void* stream);

/* Diagnostic: MGS_VALU_CEILING_BLOCKS x 256 threads (8 waves per SIMD on all 256 compute units) run `iters` trips of 8
 * independent v_fma_f32 each and store one float per thread into out[MGS_VALU_CEILING_BLOCKS * 256].  Timed by the caller,
 * 8 * iters * MGS_VALU_CEILING_BLOCKS * 4 wave-instructions / time is the plain-FMA issue rate this chip sustains (the
 * ceiling bench.py quotes next to the 1228.8 G wave-inst/s of the data sheet). */
#define MGS_VALU_CEILING_BLOCKS 2048
int mgs_debug_valu_ceiling(float* out, int32_t iters, void* stream);

/* TEST / MEASUREMENT USE ONLY, process-global, not thread-safe.
 * Measurement hook: until switched off again, every forward records fwd_start right before and fwd_end right after its
 * blend-forward launch, every mgs_backward bwd_start / bwd_end around its blend-backward launch, on the call's stream
 * (hipEvent_t handles owned by the caller, created with timing enabled; each pair both set or both NULL; four NULLs switch
 * the hook off).  No synchronisation, nothing else changes: bench.py times the two blend kernels INSIDE its timed region
 * with it (an mgs_timing struct synchronises per call, and a device that idles between kernels clocks the issue-bound
 * blend kernels ~8 % slower than back-to-back steps do).  Process-wide, not thread-safe, not for use under stream capture. */
int mgs_debug_set_blend_events(void* fwd_start, void* fwd_end, void* bwd_start, void* bwd_end);

/* Diagnostic (not on the hot path): counts what the blend backward of the matching forward does, into
 * stats_dev[MGS_BLEND_STATS_WORDS] (device uint64): [0] 64-instance steps walked, [1] instances that pass the per-quadrant
 * cull and are fetched ("survivors"), [2] survivors with >= 1 active pixel (= wave reductions = atomic instructions),
 * [3] active (pixel, instance) pairs, [4] inactive survivors that are inactive only because of the depth order,
 * [5..7] active survivors with <= 2 / 4 / 8 active pixels.  bench.py divides the kernel's VALU instruction count
 * (rocprofv3 --pmc) by [1] and [2] to report instructions per survivor.
 * [8 + 3 d + {0, 1, 2}], d = 0..4 (round 5): what the walk would cost if the wave ran one survivor stream per GROUP of
 * pixels -- d = 0: two 8x4 halves (top / bottom), 1: two 4x8 halves (left / right), 2: four 4x4 blocks, 3: four 8x2 strips,
 * 4: eight 4x2 blocks -- with the cull run per group: {0} loop trips when the groups' survivor lists are paired step by step
 * (sum over steps of the longest group list), {1} (group, survivor) rows, {2} trips when every group runs down its own
 * list over the whole walk (sum over quadrant walks of the longest group total).  Compare with [1]. */
#define MGS_BLEND_STATS_WORDS 24
int mgs_debug_blend_stats(const mgs_camera* cam, int32_t P, uint64_t num_rendered, const void* geometry,
                          const void* binning, const void* image, uint64_t* stats_dev, void* stream);

/* visible[P] (1 byte each) = view-space z > 0.2 (upstream markVisible; unused by MonoGS). */
int mgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* visible, void* stream);

/* out[P] = mean squared distance to the 3 nearest other points (exact). */
size_t mgs_knn_scratch_bytes(int32_t P);
int mgs_dist2_knn(int32_t P, const float* points /* [P,3] */, float* out /* [P] */,
                  void* scratch, void* stream);

/* ---- Fused SLAM losses (caller-side widening) --------------------------------
 * Forward value + analytic gradients of get_loss_mapping (/root/reference/utils/slam_utils.py:101-146,
 * mode = 0) and get_loss_tracking (:58-98, mode & MGS_LOSS_TRACKING).  MGS_LOSS_INVERT_DEPTH selects the reference's
 * `invert_depth=True` branch (:83-88, :138-141): the depth term compares 1 / (depth + eps) with 1 / (gt_depth + eps),
 * eps = 1e-6 in the tracking loss and 0 in the mapping loss, exactly as the reference writes the two.  render[3,H,W], depth[1,H,W], opacity[1,H,W]
 * (tracking only), gt_rgb[3,H,W], gt_depth[H,W]; mask / grad_mask are [H,W] bytes (0 / non-zero; mask may be
 * NULL = all ones); exposure_a / exposure_b are device scalars (ignored when init != 0: rgb = render).
 * mgs_loss_forward writes the scalar loss to loss_out [device] and keeps its sums in `scratch`
 * (mgs_loss_scratch_bytes); mgs_loss_backward turns them into d_render[3,H,W], d_depth[1,H,W] and
 * d_exposure[2] = (dL/da, dL/db; may be NULL), all scaled by the device scalar grad_out (NULL = 1).
 * The opacity image gets no gradient (the rasteriser ignores dL/dopacity).
 * d_exposure is STORED (scale x the unscaled sums the forward kept in its per-workgroup partials: no atomics, no clear,
 * bitwise reproducible); it may be the two floats at scratch + MGS_LOSS_SCRATCH_DAB. */
#define MGS_LOSS_TRACKING 1
#define MGS_LOSS_INVERT_DEPTH 2
#define MGS_LOSS_SCRATCH_DAB 10
#define MGS_LOSS_SCRATCH_LOSS 12
size_t mgs_loss_scratch_bytes(void);
int mgs_loss_forward(int32_t width, int32_t height, int32_t mode, int32_t init, float lambda_rgb,
                     const float* render, const float* depth, const float* opacity, const float* gt_rgb,
                     const float* gt_depth, const uint8_t* mask, const uint8_t* grad_mask,
                     const float* exposure_a, const float* exposure_b, float* scratch, float* loss_out,
                     void* stream);
int mgs_loss_backward(int32_t width, int32_t height, int32_t mode, int32_t init, float lambda_rgb,
                      const float* render, const float* depth, const float* opacity, const float* gt_rgb,
                      const float* gt_depth, const uint8_t* mask, const uint8_t* grad_mask,
                      const float* exposure_a, const float* exposure_b, const float* scratch,
                      const float* grad_out, float* d_render, float* d_depth, float* d_exposure, void* stream);
/* Loss value AND gradients (for grad_out = 1) in two launches, for loops that drive the rasteriser's backward themselves
 * (torch.autograd.backward([color, depth], [d_render, d_depth])) instead of building an autograd node for the scalar:
 * no finalize kernel, no ones-fill.  Afterwards scratch[MGS_LOSS_SCRATCH_LOSS] holds the loss value and
 * scratch[MGS_LOSS_SCRATCH_DAB .. +1] hold dL/d(exposure_a), dL/d(exposure_b) (unless `init`). */
int mgs_loss_grads(int32_t width, int32_t height, int32_t mode, int32_t init, float lambda_rgb,
                   const float* render, const float* depth, const float* opacity, const float* gt_rgb,
                   const float* gt_depth, const uint8_t* mask, const uint8_t* grad_mask,
                   const float* exposure_a, const float* exposure_b, float* scratch, float* d_render, float* d_depth,
                   void* stream);

/* ---- Fused pose update (caller-side widening) -------------------------------
 * torch.optim.Adam.step() on (cam_rot_delta lr_rot, cam_trans_delta lr_trans, exposure_a/b lr_exposure)
 * followed by update_pose (/root/reference/utils/pose_utils.py:76-93): T_cw <- exp([rho; theta]^) T_cw,
 * deltas reset to zero.  R[3,3] (row-major), T[3], the parameters and the Adam moments adam_m[8], adam_v[8]
 * (order rot(3), trans(3), a, b) are device tensors updated in place; `step` is the 1-based Adam step, or, when
 * step_counter (device int32) is non-NULL, the counter is incremented on the device and used instead (so the call
 * can be captured in a hipGraph and replayed);
 * out[2] = {converged (|tau| < converged_threshold ? 1 : 0), |tau|}.  exposure pointers / any gradient may be NULL.
 * flags: MGS_POSE_STICKY makes the call a no-op once out[0] reports convergence (the tracker's early exit,
 * /root/reference/utils/slam_tracker.py:172-176, for loops replayed from a hipGraph); zero out[] to start over.
 * Camera refresh (optional, all four or none): with viewmatrix / projmatrix / campos non-NULL the kernel also recomputes
 * the viewpoint's camera tensors from the NEW R, T and projmatrix_raw -- bit-identical to mgs_camera_setup -- so the next
 * render of the loop needs no camera launch of its own. */
#define MGS_POSE_STICKY 1
int mgs_pose_step(float* R, float* T, float* rot_delta, float* trans_delta, float* exposure_a, float* exposure_b,
                  const float* grad_rot, const float* grad_trans, const float* grad_a, const float* grad_b,
                  float* adam_m, float* adam_v, int32_t step, float lr_rot, float lr_trans, float lr_exposure,
                  float beta1, float beta2, float eps, float converged_threshold, int32_t* step_counter, float* out,
                  int32_t flags, float* host_flag /* optional: pinned host word that also receives out[0] */,
                  const float* projmatrix_raw, float* viewmatrix, float* projmatrix, float* campos /* optional refresh */,
                  void* stream);
/* The same update for n <= 16 independent viewpoints in ONE launch (the keyframes of a mapping window,
 * /root/reference/utils/slam_mapper.py:486-496).  `ptrs`: HOST array of n x 18 device pointers in the order of mgs_pose_step's
 * pointer arguments (R, T, rot_delta, trans_delta, exposure_a, exposure_b, grad_rot, grad_trans, grad_a, grad_b, adam_m, adam_v,
 * step_counter [required], out, projmatrix_raw, viewmatrix, projmatrix, campos; NULL where optional); the scalars are shared. */
int mgs_pose_step_batch(int32_t n, void* const* ptrs, float lr_rot, float lr_trans, float lr_exposure, float beta1,
                        float beta2, float eps, float converged_threshold, int32_t flags, void* stream);

/* ---- Keyframe back-projection ------------------------------------------------
 * The per-point part of GaussianModel.create_viewpoint_pcd (/root/reference/gaussian_splatting/scene/gaussian_model.py:121-319)
 * for N selected pixels: gather rgb (with the tracked exposure exp(a)*rgb + b clamped to [0,1] when exposure_a is
 * non-NULL) and depth, unproject the pixel centre (x+0.5, y+0.5) with (fx, fy, cx, cy) and move it to the world
 * with the world->camera pose (R row-major, T): p_w = R^T (p_c - T).  selected[i] = x * H + y (the reference's
 * flattening order, gaussian_model.py:180-187).  segmentation / ids may be NULL. */
int mgs_backproject(int32_t N, int32_t W, int32_t H, const int64_t* selected, const float* rgb /* [3,H,W] */,
                    const float* depth /* [H,W] */, const int32_t* segmentation /* [H,W] or NULL */,
                    const float* exposure_a, const float* exposure_b, float fx, float fy, float cx, float cy,
                    const float* R, const float* T, float* points /* [N,3] */, float* features /* [N,3] */,
                    int32_t* ids /* [N] or NULL */, void* stream);

/* ---- Camera matrices of one viewpoint ---------------------------------
 * From the world->camera rotation R[3,3] (row-major) and translation T[3] and the transposed projection
 * projmatrix_raw[4,4]: viewmatrix = getWorld2View(R, T)^T (/root/reference/gaussian_splatting/utils/graphics_utils.py:33-42,
 * /root/reference/utils/camera_utils.py:171-174), projmatrix = viewmatrix @ projmatrix_raw
 * (/root/reference/utils/camera_utils.py:224-231), campos = viewmatrix^-1[3,:3] = -R^T T
 * (/root/reference/utils/camera_utils.py:176-178).  One launch; all pointers are device memory. */
int mgs_camera_setup(const float* R, const float* T, const float* projmatrix_raw, float* viewmatrix /* [16] */,
                     float* projmatrix /* [16] */, float* campos /* [3] */, void* stream);

/* ---- Fused Gaussian optimiser step + densification statistics ---------------
 * mgs_adam_step: torch.optim.Adam defaults over n_tensors <= 8 float tensors in one launch (the reference's five
 * groups: /root/reference/gaussian_splatting/scene/gaussian_model.py:398-442).  All tables are HOST arrays of
 * device pointers / sizes / learning rates; grads[t] may be NULL: that tensor is skipped (parameter, moments and step
 * count untouched), as torch.optim.Adam skips a parameter whose .grad is None.  `step` is the 1-based step, or
 * step_counter (device int32[n_tensors], one count per tensor like torch's state["step"]) is incremented on the
 * device for the tensors that have a gradient and used instead (hipGraph-capturable).
 * mgs_densify_stats: for the Gaussians with radii > 0,  xyz_gradient_accum += ||viewspace_grad[:, :2]||,
 * denom += 1 (gaussian_model.py:888-892), max_radii_2d = max(max_radii_2d, radii) (utils/slam_mapper.py:453-457);
 * any of the three outputs may be NULL. */
int mgs_adam_step(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const uint64_t* numel, const float* lr, double beta1, double beta2,
                  double eps, int32_t step, int32_t* step_counter,
                  const float* lr_device /* device float[n_tensors] used instead of lr[] when non-NULL: a schedule
                                            stepped on the device (mgs_lr_schedule_step), hipGraph-replayable */,
                  void* stream);
int mgs_densify_stats(int32_t P, const float* viewspace_grad /* [P,3] */, const int32_t* radii,
                      float* xyz_gradient_accum /* [P] */, float* denom /* [P] */, float* max_radii_2d /* [P] */,
                      void* stream);
/* The statistics of ONE mapping iteration over the n_keyframes <= 32 keyframes a rank rendered, in one launch -- what
 * /root/reference/utils/slam_mapper.py:400-404,453-460 does keyframe by keyframe after loss.backward():
 *   visibility_bits[k][w] bit i = (n_touched_k[64 w + i] > 0)   (occ_aware_visibility; uint64 words, ceil(P/64) per keyframe;
 *                                                                 may be NULL)
 *   over v = radii_k > 0:  grad_norm[v] += ||grad_means2D_k[v, :2]||,  visible[v] += 1,  max_radii[v] = max(., radii_k[v])
 * The keyframe tables are HOST arrays of device pointers (grad_means2D[k] may be NULL).  accumulate != 0: the three arrays
 * are the map's running statistics (xyz_gradient_accum, denom, max_radii_2d) and end up bit for bit as the reference's
 * loop over keyframes leaves them; accumulate == 0 (keyframe-sharded window): they receive this rank's share of the
 * iteration only, to be summed / maximised across ranks and folded in with mgs_window_apply. */
int mgs_window_stats(int32_t P, int32_t n_keyframes, const float* const* grad_means2D, const int32_t* const* radii,
                     const int32_t* const* n_touched, float* grad_norm /* [P] */, float* visible /* [P] */,
                     float* max_radii /* [P] */, int32_t accumulate, uint64_t* visibility_bits, void* stream);
int mgs_window_apply(int32_t P, const float* grad_norm, const float* visible, const float* max_radii,
                     float* xyz_gradient_accum, float* denom, float* max_radii_2d, void* stream);
/* GaussianModel.update_learning_rate (/root/reference/gaussian_splatting/scene/gaussian_model.py:451-465, schedule
 * /root/reference/gaussian_splatting/utils/general_utils.py:79-94) on the device: *iteration += 1, *lr_out = schedule of
 * the new value -- the mapper's `self.nr_iters` and the xyz group's learning rate, kept in device memory so that a
 * captured mapping iteration steps them itself. */
int mgs_lr_schedule_step(int32_t* iteration, float* lr_out, double lr_init, double lr_final, int32_t lr_delay_steps,
                         double lr_delay_mult, int32_t max_steps, void* stream);

/* ---- Fused map activations ------------------------------------------------
 * rotations = F.normalize(rot_raw), scales3 = exp(scale_raw) (isotropic [P,1] expanded to [P,3] as render() does),
 * opacities = sigmoid(opacity_raw): /root/reference/gaussian_splatting/scene/gaussian_model.py:84-106 and
 * gaussian_renderer/__init__.py:101-104.  One launch forward, one backward (any gradient / output may be NULL). */
int mgs_activate_forward(int32_t P, int32_t scale_dim /* 1 or 3 */, const float* rot_raw /* [P,4] */,
                         const float* scale_raw /* [P,scale_dim] */, const float* opacity_raw /* [P] */,
                         float* rotations /* [P,4] */, float* scales3 /* [P,3] */, float* opacities /* [P] */,
                         void* stream);
int mgs_activate_backward(int32_t P, int32_t scale_dim, const float* rot_raw, const float* scales3,
                          const float* opacities, const float* grad_rotations, const float* grad_scales3,
                          const float* grad_opacities, float* d_rot_raw, float* d_scale_raw, float* d_opacity_raw,
                          void* stream);

/* dst[i] = src_0[i] + src_1[i] + ... (1..16 device buffers of `count` floats, `src` a HOST array of device pointers; dst may
 * be one of the sources): the gradients that the N keyframe renders of a mapping window return for the same map tensor,
 * summed in ONE launch instead of the autograd engine's N - 1 pairwise adds per tensor (monogs_amd.window.fan_out). */
int mgs_sum_buffers(int32_t n_src, const float* const* src, float* dst, uint64_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MONOGS_RASTER_H */
