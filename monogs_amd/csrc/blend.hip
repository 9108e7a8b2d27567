// Per-tile alpha blending, forward and backward.
//
// Boundary replaced: upstream `renderCUDA` forward / backward of the un-vendored rasteriser
// (SURVEY.md section 2.1 K6, K7); per-fragment rule corroborated by
// /root/reference/viewer/gl_render/shaders/gau_frag.glsl:20-25; output semantics pinned by the
// consumers /root/reference/utils/slam_utils.py:71,91,143 and utils/slam_tracker.py:414.
//
// gfx950 mapping.  A 16x16 tile is one 256-thread workgroup, but its four wavefronts never
// synchronise: wave w owns the 8x8 pixel quadrant (w&1, w>>1), one pixel per lane, and walks the
// tile's instance list on its own, 64 instances per step:
//   1. lane l loads instance l of the step: the Gaussian index (coalesced) and float4 #0 of its
//      64-byte record, {px, py, ex, ey};
//   2. lane l tests the alpha >= 1/255 bounding box (px +- ex, py +- ey) against the wave's quadrant;
//      __ballot compacts the survivors into a 64-bit mask held in scalar registers;
//   3. the wave pops survivors off the mask; the survivor's record arrives -- forward: with broadcast reads from a
//      per-wave LDS queue that the lanes filled with the records they prefetched; backward: through the scalar cache
//      (v_readlane of the index, s_load_dwordx2 + s_load_dwordx8) -- and all 64 lanes (pixels) evaluate it.
// No barriers; a quadrant whose pixels are all saturated retires early.
// Skipping an instance for a whole quadrant never changes a pixel: every skipped pair has
// alpha < 1/255 and the per-pixel rule would have skipped it too.
#include "common.h"

#include <type_traits>


namespace mgs {

extern int g_opt_blend_lds_pad_fwd, g_opt_blend_lds_pad_bwd;
#ifdef BS_TRACE
// Diagnostic build only (tools/build_variant.sh trace blend.hip -DBS_TRACE; tools/wave_timeline.py): every wave of the two blend
// kernels leaves {start, end} (s_memrealtime, 100 MHz), its hardware id and the survivors it evaluated in a buffer of its own.
static unsigned long long* g_trace_fwd = nullptr;
static unsigned long long* g_trace_bwd = nullptr;
#define BS_STAMP(v) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#endif

__device__ __forceinline__ uint32_t bcast(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

// One 256-thread workgroup per 16x16 tile, wave w = quadrant w.  (Round 5 measured one 64-thread workgroup per (tile, quadrant),
// launched as one contiguous band of quadrants per XCD, so that wave slots refill one by one instead of four at a time: C5
// backward 0.335 / 0.343 against 0.345 / 0.333 ms, forward 0.190 / 0.192 against 0.194 / 0.189 -- nothing)
constexpr int MGS_WG_WAVES = 4;
#define MGS_TILE_WAVE(ntiles_, tile_, wave_, ws_)                                                   \
    const int tile_ = (int)blockIdx.x, wave_ = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), ws_ = wave_; \
    if (tile_ >= (ntiles_)) return;
struct BlendArgs {
#ifdef BS_TRACE
    unsigned long long* trace;
#endif
    const float4* __restrict__ rec;            // [P][4]
    const uint32_t* __restrict__ point_list;
    const uint2* __restrict__ ranges;
    const float* __restrict__ bg;
    int W, H, gx;
};

// Does the alpha >= 1/255 ellipse of an instance reach the 8x8 pixel quadrant whose first pixel is (qx0, qy0)?
// First the padded bounding box (rejects most), then the exact minimum of the ellipse's quadratic form over the
// quadrant rectangle: the centre is inside, or the minimum lies on one of the four edges, where the quadratic is
// one-dimensional and its clamped minimiser is closed-form.  Conservative: threshold 1.02 instead of 1 and the
// rectangle grown by 0.05 px, so a skipped instance has alpha < 1/255 at every pixel of the quadrant under any
// rounding.  Runs once per lane per 64 instances, so its ~40 instructions cost < 1 per instance.
//
// `alive` is the wave's 64-bit mask of pixels that can still use an instance of this step (forward: not yet
// saturated; backward: the pixel's last contributor is not in front of the step).  An instance whose cull box covers
// no alive pixel is skipped as well -- in a dense scene most of a tile's depth range is walked for the sake of a few
// unsaturated pixels, and most instances of those steps sit over pixels that finished long ago.  Exact: a pixel
// outside the box has alpha < 1/255, a dead pixel ignores every instance.
__device__ __forceinline__ bool quadrant_hit(const float4 c /* px, py, ex, ey */, const float4 n /* na, nb, nc, . */,
                                             float qx0, float qy0, unsigned long long alive) {
    const float x0 = qx0 - 0.05f - c.x, x1 = qx0 + (float)(SUB - 1) + 0.05f - c.x;   // rectangle relative to the centre
    const float y0 = qy0 - 0.05f - c.y, y1 = qy0 + (float)(SUB - 1) + 0.05f - c.y;
    if (c.z < 0.f) return false;                      // opacity < 1/255 (or an empty slot): contributes nowhere
    if (!((c.z >= x0) && (-c.z <= x1) && (c.w >= y0) && (-c.w <= y1))) return false;   // bounding boxes apart
    {   // pixels (qx0 + i, qy0 + j) inside the box: i in [ix0, ix1], j in [iy0, iy1] (clamped; the box may be huge or NaN-wide)
        const int ix0 = (int)fminf(8.f, fmaxf(0.f, ceilf(-x0 - c.z - 0.06f))), ix1 = (int)fminf(7.f, fmaxf(-1.f, floorf(-x0 + c.z + 0.06f)));
        const int iy0 = (int)fminf(8.f, fmaxf(0.f, ceilf(-y0 - c.w - 0.06f))), iy1 = (int)fminf(7.f, fmaxf(-1.f, floorf(-y0 + c.w + 0.06f)));
        if (ix0 > ix1 || iy0 > iy1) return false;
        const unsigned long long cols = (unsigned long long)((0xFFu >> (7 - ix1)) & (0xFFu << ix0) & 0xFFu) * 0x0101010101010101ull;
        const unsigned long long rows = (~0ull >> (8 * (7 - iy1))) & (~0ull << (8 * iy0));
        if ((cols & rows & alive) == 0ull) return false;
    }
    if (x0 <= 0.f && x1 >= 0.f && y0 <= 0.f && y1 >= 0.f) return true;                 // centre inside
    if (!(n.x > 0.f) || !(n.z > 0.f)) return true;                                     // no ellipse data: keep
    // v_rcp_f32 (1 ulp) instead of two IEEE divide sequences: the 2 % threshold margin dwarfs the error
    const float rby = -n.y * __builtin_amdgcn_rcpf(n.z), rbx = -n.y * __builtin_amdgcn_rcpf(n.x);
    auto edge_x = [&](float xe) {          // x = xe, y in [y0, y1]
        const float t = fminf(y1, fmaxf(y0, rby * xe));
        return n.x * xe * xe + 2.f * n.y * xe * t + n.z * t * t;
    };
    auto edge_y = [&](float ye) {          // y = ye, x in [x0, x1]
        const float t = fminf(x1, fmaxf(x0, rbx * ye));
        return n.x * t * t + 2.f * n.y * t * ye + n.z * ye * ye;
    };
    const float qmin = fminf(fminf(edge_x(x0), edge_x(x1)), fminf(edge_y(y0), edge_y(y1)));
    return qmin <= 1.02f;
}

// the survivor's record through the scalar cache (wave-uniform address -> s_load)
struct Rec {
    float px, py, ca, cb, cc, op, r, g, b, z;
};
// Read through the CONSTANT address space (common.h): the backend then always selects scalar loads for the wave-uniform
// address (s_load_dwordx2 + s_load_dwordx8).  Through the generic pointer that choice hangs on a no-clobber analysis that
// flips to per-lane vector loads with unrelated edits of the loop (an atomic moved, an LDS store added).  The records
// are written by preprocess, an earlier kernel.
__device__ __forceinline__ Rec fetch(const BlendArgs& a, uint32_t gid_uniform) {
    const const_float_p p = MGS_CONST(reinterpret_cast<const float*>(a.rec)) + (size_t)gid_uniform * REC_FLOATS;
    return Rec{p[R_X], p[R_Y], p[R_CA], p[R_CB], p[R_CC], p[R_OPAC], p[R_R], p[R_G], p[R_B], p[R_DEPTH]};
}

// One survivor against the 64 pixels of the quadrant.  Round 3: the per-pixel decisions narrow EXEC instead of building
// lane masks on the scalar unit -- the kernel was bound by the compute unit's ONE scalar ALU (24 scalar instructions per
// survivor from each of four SIMDs against 28 vector ones; r03 PMC: 86 M SALU + 124 M VALU per launch at C5):
//     EXEC = live;  v_cmpx (power <= 0);  v_cmpx (alpha >= 1/255)          -> EXEC = pixels this instance acts on
//     vcc = (T (1 - alpha) < 1e-4): those stop (NOT blended);  EXEC &= ~vcc -> contributors;  live &= ~vcc;
//     mask = live ? mask : 0          (s_cselect on the SCC of the line before: the walk leaves when the quadrant is done)
//     w = alpha T;  C += colour w;  D += depth w;  T = T (1 - alpha);  last = position      plain stores under EXEC:
//                                                                                            no v_cndmask, no mask algebra
//     [TOUCH] vcc = (T (1 - alpha) > 0.5);  lane j of touched_cnt = popcount(vcc)            (v_writelane ignores EXEC)
//     EXEC = all
// One asm block (the compiler must never see a narrowed EXEC): 17 scalar + 23 vector instructions per survivor, 19 + 25
// while a pixel of the quadrant is still in front of half its light (TOUCH: only then can T (1 - alpha) exceed 0.5).
#ifdef MGS_FWD_SCALAR
#define MGS_RECOP "s"          // experiment: the survivor's record through two scalar loads (SGPR offset), as the round-5 backward
#else
#define MGS_RECOP "v"          // the survivor's record comes back from the per-wave LDS queue in vector registers
#endif
template <bool TOUCH>
__device__ __forceinline__ void blend_one(unsigned long long& live, unsigned long long& mask, float& T, uint32_t& last, float& C0, float& C1, float& C2,
                                      float& D, const Rec& g, float power, float alpha, uint32_t pos, int j,
                                      int& touched_cnt) {
    const float test_T = T * (1.f - alpha);
    float w;
    int cnt;
    if (TOUCH) {
        asm volatile(
            "s_mov_b64 exec, %[live]\n\t"
            "v_cmpx_nlt_f32_e32 vcc, 0, %[power]\n\t"
            "v_cmpx_ngt_f32_e32 vcc, %[amin], %[alpha]\n\t"
            "v_cmp_gt_f32_e32 vcc, %[tmin], %[tt]\n\t"
            "s_andn2_b64 exec, exec, vcc\n\t"
            "s_andn2_b64 %[live], %[live], vcc\n\t"
            "s_cselect_b64 %[mask], %[mask], 0\n\t"
            "v_mul_f32_e32 %[w], %[alpha], %[T]\n\t"
            "v_mov_b32_e32 %[T], %[tt]\n\t"
            "v_mov_b32_e32 %[last], %[pos]\n\t"
            "v_cmp_lt_f32_e32 vcc, 0.5, %[tt]\n\t"
            "v_fmac_f32_e32 %[C0], %[cr], %[w]\n\t"
            "v_fmac_f32_e32 %[C1], %[cg], %[w]\n\t"
            "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
            "s_mov_b32 m0, %[j]\n\t"
            "v_fmac_f32_e32 %[C2], %[cb], %[w]\n\t"
            "v_fmac_f32_e32 %[D], %[cz], %[w]\n\t"
            "s_mov_b64 exec, -1\n\t"
            "v_writelane_b32 %[tc], %[cnt], m0\n\t"
            : [live] "+s"(live), [mask] "+s"(mask), [T] "+v"(T), [last] "+v"(last), [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [D] "+v"(D),
              [w] "=&v"(w), [cnt] "=&s"(cnt), [tc] "+v"(touched_cnt)
            : [power] "v"(power), [alpha] "v"(alpha), [tt] "v"(test_T), [amin] "s"(1.0f / 255.0f), [tmin] "s"(0.0001f),
              [pos] "s"(pos), [cr] MGS_RECOP(g.r), [cg] MGS_RECOP(g.g), [cb] MGS_RECOP(g.b), [cz] MGS_RECOP(g.z), [j] "s"(j)
            : "vcc", "scc", "m0");
    } else {
        asm volatile(
            "s_mov_b64 exec, %[live]\n\t"
            "v_cmpx_nlt_f32_e32 vcc, 0, %[power]\n\t"
            "v_cmpx_ngt_f32_e32 vcc, %[amin], %[alpha]\n\t"
            "v_cmp_gt_f32_e32 vcc, %[tmin], %[tt]\n\t"
            "s_andn2_b64 exec, exec, vcc\n\t"
            "s_andn2_b64 %[live], %[live], vcc\n\t"
            "s_cselect_b64 %[mask], %[mask], 0\n\t"
            "v_mul_f32_e32 %[w], %[alpha], %[T]\n\t"
            "v_mov_b32_e32 %[T], %[tt]\n\t"
            "v_mov_b32_e32 %[last], %[pos]\n\t"
            "v_fmac_f32_e32 %[C0], %[cr], %[w]\n\t"
            "v_fmac_f32_e32 %[C1], %[cg], %[w]\n\t"
            "v_fmac_f32_e32 %[C2], %[cb], %[w]\n\t"
            "v_fmac_f32_e32 %[D], %[cz], %[w]\n\t"
            "s_mov_b64 exec, -1\n\t"
            : [live] "+s"(live), [mask] "+s"(mask), [T] "+v"(T), [last] "+v"(last), [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [D] "+v"(D),
              [w] "=&v"(w)
            : [power] "v"(power), [alpha] "v"(alpha), [tt] "v"(test_T), [amin] "s"(1.0f / 255.0f), [tmin] "s"(0.0001f),
              [pos] "s"(pos), [cr] MGS_RECOP(g.r), [cg] MGS_RECOP(g.g), [cb] MGS_RECOP(g.b), [cz] MGS_RECOP(g.z)
            : "vcc", "scc");
    }
}

__global__ void __launch_bounds__(256) blend_forward_kernel(BlendArgs a, float* __restrict__ out_color,
                                                            float* __restrict__ out_depth,
                                                            float* __restrict__ out_opacity,
                                                            float* __restrict__ final_T,
                                                            uint32_t* __restrict__ n_contrib,
                                                            int32_t* __restrict__ n_touched,
                                                            uint2* ranges_rw /* = a.ranges: read AND written here, through this pointer only */,
                                                            const uint32_t* __restrict__ sort_err,
                                                            uint32_t* __restrict__ status) {
    MGS_TILE_WAVE(a.gx * ((a.H + TILE - 1) / TILE), tile, wave, ws)
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int lane = threadIdx.x & 63;
    const bool first_of_tile = wave == 0 && lane == 0;
    const int qx0i = tx * TILE + (wave & 1) * SUB, qy0i = ty * TILE + (wave >> 1) * SUB;
    const int pxi = qx0i + (lane & 7), pyi = qy0i + (lane >> 3);
    const bool inside = pxi < a.W && pyi < a.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const float qx0 = (float)qx0i, qy0 = (float)qy0i;
    // The tile's range as the tile sort's final pass left it: {first, last + 1} of the tile's instances, or the preset {~0, 0}
    // of a tile nothing landed in.  A tile sort whose look-back timed out left the instance list partly unwritten (arbitrary
    // indices): then every tile is empty, and the forward's status word says why.  Either way the table is rewritten in its
    // canonical form ({0, 0} for an empty tile) for the backward and for whoever reads it: this workgroup is its only reader here.
    // (read through the SAME pointer the normalised form is stored through below -- `a.ranges` is a restrict-qualified const view
    //  of this table for the kernels that only read it; the other waves of the workgroup may see the table before or after lane 0's
    //  store: both forms decode to the same range.  The backward relies on the canonical {0, 0} written here.)
    uint2 range = ranges_rw[tile];
    {
        const bool sort_bad = sort_err != nullptr && radix_failed(sort_err) != 0u;
        if (sort_bad && tile == 0 && first_of_tile && status) atomicOr(status, (uint32_t)MGS_STATUS_TILE_SORT_TIMEOUT);
        if (sort_bad || range.x >= range.y) {
            range = make_uint2(0u, 0u);
            if (first_of_tile) ranges_rw[tile] = range;
        }
    }

    // T is the running transmittance.  Which pixels are still blending is a 64-bit lane mask kept in scalar registers
    // (`live`): every per-pixel decision below is a v_cmp that writes a lane mask, the masks are combined on the scalar
    // unit, and __builtin_amdgcn_inverse_ballot_w64 hands a mask back to v_cndmask as its condition -- no predicate is
    // ever materialised in a VGPR (a ballot of a compound bool compiled to v_cndmask 0/1 + v_cmp_ne, twice per
    // survivor, and the "done" state cost three selects on T / Tf).  A pixel outside the image never blends.
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
    uint32_t last = 0;
    unsigned long long live = __builtin_amdgcn_ballot_w64(inside);
#ifdef BS_TRACE
    unsigned long long tr0;
    BS_STAMP(tr0);
#endif

    uint32_t gid_n = 0;
    float4 box_n = make_float4(0.f, 0.f, -1.f, -1.f), ell_n = make_float4(0.f, 0.f, 0.f, 0.f);
    // Round 4: the survivors' records come through a per-wave LDS queue instead of the scalar cache.  Every lane prefetches
    // the WHOLE 64-byte record of its instance of the next step (the cache line its cull data comes from anyway: no extra
    // line is touched) and parks {px, py, ex, ey | conic, opacity | colour, depth} in its queue entry; a survivor's record
    // then comes back with three broadcast ds_reads of a wave-uniform address.  Against the scalar fetch this removes a
    // dependent scalar-cache round trip per survivor (32 waves of different tiles share a 16 KB scalar cache: most of them
    // went to L2) and five scalar instructions of address arithmetic: 0.215 -> 0.191 ms at C5, 38.0 -> 34.8 us at 100 k / VGA.
    // (Rounds 2 built the same queue against the mask-algebra kernel and measured no gain: that kernel was bound by its
    // scalar ALU work, not by the fetch.)  No barrier: the queue belongs to one wave and a wave's LDS operations run in order.
#ifdef MGS_FWD_SCALAR
    typedef float fv2 __attribute__((ext_vector_type(2)));
    typedef float fv8 __attribute__((ext_vector_type(8)));
    const const_float_p recs = MGS_CONST(reinterpret_cast<const float*>(a.rec));
    auto prefetch = [&](uint32_t i) {
        gid_n = 0;
        box_n = make_float4(0.f, 0.f, -1.f, -1.f);
        if (i < range.y) {
            gid_n = a.point_list[i];
            box_n = a.rec[(size_t)gid_n * 4];
            ell_n = a.rec[(size_t)gid_n * 4 + 3];
        }
    };
    auto walk_step = [&](auto touch_tag, uint32_t base, uint32_t gid_l, unsigned long long mask) {
        int touched_cnt = 0;
        const uint32_t goff_l = gid_l << 6;
        while (mask) {
            int j;
            asm volatile("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0" : "=&s"(j), "+s"(mask));
            const uint32_t goff = (uint32_t)__builtin_amdgcn_readlane((int)goff_l, j);
            fv2 r0;
            fv8 r1;
            asm volatile("s_load_dwordx2 %0, %2, %3\n\ts_load_dwordx8 %1, %2, %3 offset:0x10\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(r0), "=&s"(r1) : "s"(recs), "s"(goff) : "memory");
            const Rec g{r0[0], r0[1], r1[0], r1[1], r1[2], r1[3], r1[4], r1[5], r1[6], r1[7]};
            const float dx = g.px - pxf, dy = g.py - pyf;
            const float power = dx * (g.ca * dx + g.cb * dy) + (g.cc * dy) * dy;
            const float alpha = fminf(0.99f, g.op * __builtin_amdgcn_exp2f(power));
            blend_one<decltype(touch_tag)::value>(live, mask, T, last, C0, C1, C2, D, g, power, alpha,
                                                  (base - range.x) + (uint32_t)j + 1u, j, touched_cnt);
        }
        if (decltype(touch_tag)::value && touched_cnt != 0) atomicAdd(n_touched + gid_l, touched_cnt);
    };
    if (range.x < range.y) prefetch(range.x + lane);
    for (uint32_t base = range.x; base < range.y && live != 0ull; base += WAVE) {
        const uint32_t gid_l = gid_n;
        const float4 c = box_n, el = ell_n;
        prefetch(base + WAVE + lane);
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(quadrant_hit(c, el, qx0, qy0, live));
        if ((live & __builtin_amdgcn_ballot_w64(T > 0.5f)) != 0ull) walk_step(std::true_type{}, base, gid_l, mask);
        else walk_step(std::false_type{}, base, gid_l, mask);
    }
#else
    __shared__ __attribute__((aligned(16))) float4 s_queue[MGS_WG_WAVES][WAVE][3];
    float4 c1_n = make_float4(0.f, 0.f, 0.f, 0.f), c2_n = c1_n;
    float4* const my_entry = &s_queue[ws][lane][0];
    // LDS byte address of this wave's queue, in a scalar register: an entry's address is then scalar arithmetic + one v_mov
    const uint32_t q_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)&s_queue[ws][0][0]);
    auto prefetch = [&](uint32_t i) {               // next step's index + cull data, issued one step ahead
        gid_n = 0;
        box_n = make_float4(0.f, 0.f, -1.f, -1.f);
        if (i < range.y) {
            gid_n = a.point_list[i];
            box_n = a.rec[(size_t)gid_n * 4];
            ell_n = a.rec[(size_t)gid_n * 4 + 3];
            c1_n = a.rec[(size_t)gid_n * 4 + 1];
            c2_n = a.rec[(size_t)gid_n * 4 + 2];
        }
    };
    auto walk_step = [&](auto touch_tag, uint32_t base, uint32_t gid_l, unsigned long long mask) {
        // n_touched of this step's instances, collected in lane j of one register (v_writelane: one instruction, no
        // branch) and added with ONE vector atomic per step instead of a guarded lane-0 atomic per touched survivor
        int touched_cnt = 0;
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= ~(1ull << j);
            // uniform address: three broadcast reads, each waited for where its values are first needed (LDS reads of a wave
            // return in order): the centre arrives first and dx, dy start while the conic and the colour are still on their way
            typedef float qv2 __attribute__((ext_vector_type(2)));
            typedef float qv4 __attribute__((ext_vector_type(4)));
            qv2 e0;
            qv4 e1, e2;
            {
                const uint32_t ad = q_base + (uint32_t)j * 48u;
                asm volatile("ds_read_b64 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32"
                             : "=&v"(e0), "=&v"(e1), "=&v"(e2) : "v"(ad) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(e0));
            float dx = e0[0] - pxf, dy = e0[1] - pyf;
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(e1), "+v"(dx), "+v"(dy));      // (dx, dy tied in: formed BEFORE this wait)
            const float power = dx * (e1[0] * dx + e1[1] * dy) + (e1[2] * dy) * dy;     // log2 of the Gaussian falloff
            float alpha = fminf(0.99f, e1[3] * __builtin_amdgcn_exp2f(power));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e2), "+v"(alpha));             // (alpha formed BEFORE the colour is waited for)
            const Rec g{e0[0], e0[1], e1[0], e1[1], e1[2], e1[3], e2[0], e2[1], e2[2], e2[3]};
            // (also: mask = 0 once no pixel of the quadrant is live -- the walk's only exit test stays `mask != 0`)
            blend_one<decltype(touch_tag)::value>(live, mask, T, last, C0, C1, C2, D, g, power, alpha,
                                                  (base - range.x) + (uint32_t)j + 1u, j, touched_cnt);
        }
        if (decltype(touch_tag)::value && touched_cnt != 0) atomicAdd(n_touched + gid_l, touched_cnt);
    };
    if (range.x < range.y) prefetch(range.x + lane);
    for (uint32_t base = range.x; base < range.y && live != 0ull; base += WAVE) {
        const uint32_t gid_l = gid_n;
        const float4 c = box_n, el = ell_n;
        my_entry[0] = c; my_entry[1] = c1_n; my_entry[2] = c2_n;      // (the wave's own queue: LDS ops of a wave run in order)
        prefetch(base + WAVE + lane);
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(quadrant_hit(c, el, qx0, qy0, live));
        // a pixel counts as "touched" by an instance when T (1 - alpha) > 0.5: impossible once every live pixel has T <= 0.5
        if ((live & __builtin_amdgcn_ballot_w64(T > 0.5f)) != 0ull) walk_step(std::true_type{}, base, gid_l, mask);
        else walk_step(std::false_type{}, base, gid_l, mask);
    }
#endif
#ifdef BS_TRACE
    if (a.trace) {
        unsigned long long tr1;
        BS_STAMP(tr1);
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        if (lane == 0) {
            unsigned long long* o = a.trace + ((size_t)tile * 4 + wave) * 4;
            o[0] = tr0; o[1] = tr1; o[2] = (unsigned long long)hw | ((unsigned long long)xcc << 32); o[3] = range.y - range.x;
        }
    }
#endif
    if (inside) {
        const size_t pix = (size_t)pyi * a.W + pxi, HW = (size_t)a.H * a.W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C0 + T * a.bg[0];
        out_color[HW + pix] = C1 + T * a.bg[1];
        out_color[2 * HW + pix] = C2 + T * a.bg[2];
        out_depth[pix] = D;
        out_opacity[pix] = 1.f - T;
    }
}

static BlendArgs make_args(const mgs_camera& cam, const GeometryState& g, const BinningState& b,
                           const ImageState& img) {
    BlendArgs a;
#ifdef BS_TRACE
    a.trace = nullptr;
#endif
    a.rec = reinterpret_cast<const float4*>(g.rec);
    a.point_list = b.vals_sorted;
    a.ranges = img.ranges;
    a.bg = cam.bg;
    a.W = cam.image_width;
    a.H = cam.image_height;
    a.gx = tiles_x(a.W);
    return a;
}

int launch_blend_forward(const mgs_camera& cam, const GeometryState& g, const BinningState& b,
                         const ImageState& img, float* out_color, float* out_depth, float* out_opacity,
                         int32_t* n_touched, const uint32_t* sort_err, uint32_t* status, hipStream_t s) {
    BlendArgs a = make_args(cam, g, b, img);
#ifdef BS_TRACE
    a.trace = g_trace_fwd;
#endif
    const int ntiles = a.gx * tiles_y(a.H);
    if (ntiles == 0) return 0;
    const dim3 grid(ntiles), block(256);
    hipLaunchKernelGGL(blend_forward_kernel, grid, block, (size_t)g_opt_blend_lds_pad_fwd, s, a, out_color, out_depth, out_opacity,
                       img.final_T, img.n_contrib, n_touched, img.ranges, sort_err, status);
    MGS_HIP(hipGetLastError());
    return 0;
}

// =================================================================================================
// backward: the same walk, back to front, from the pixel's last contributor
// =================================================================================================

// ---- packed reduction of 10 per-lane values over the 64 lanes --------------------------------
// stage 1: v_permlane32_swap pairs (x, y): lanes 0-31 then hold x[l]+x[l+32], lanes 32-63 y[l-32]+y[l]
// stage 2: v_permlane16_swap pairs those: each 16-lane row then holds 16 partials of ONE value
// stage 3: row-local butterfly sum through ds_swizzle -> every lane of each row
__device__ __forceinline__ float swap32_add(float x, float y) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float x, float y) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Sum over each 16-lane row, result in EVERY lane of the row.  The four exchange steps go through
// ds_swizzle (LDS crossbar, its own issue port) instead of DPP: measured on gfx950 a v_add_f32_dpp costs
// ~7.9 cycles of VALU issue against ~2.9 for a plain v_add_f32 (tools/ubench/valu_rate.hip), and this
// kernel is VALU-issue-bound, so the butterfly's data movement is moved off the VALU.
template <int XOR>
__device__ __forceinline__ float swz_add(float v) {
    // bit-mask mode: lane' = ((lane & 0x1f) | 0) ^ XOR within each half-wave
    return v + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x001F | (XOR << 10)));
}
__device__ __forceinline__ float row_sum_all(float v) {
    v = swz_add<1>(v);
    v = swz_add<2>(v);
    v = swz_add<4>(v);
    v = swz_add<8>(v);
    return v;
}
// Where reduce10 leaves value k (k = gradient slot G_SX..G_DDEPTH):
//   lane 15: a0   lane 31: a2   lane 47: a1   lane 63: a3      (register c0, row sums in every lane)
//   lane  0: a4   lane 16: a6   lane 32: a5   lane 48: a7      (register c1)
//   lane  1: a8   lane 33: a9                                  (register c2)
__device__ __forceinline__ int reduce10_slot(int lane) {
    const int row = lane >> 4, pos = lane & 15;
    if (pos == 15) return row == 0 ? 0 : row == 1 ? 2 : row == 2 ? 1 : 3;
    if (pos == 0) return row == 0 ? 4 : row == 1 ? 6 : row == 2 ? 5 : 7;
    if (pos == 1) return row == 0 ? 8 : row == 2 ? 9 : -1;
    return -1;
}
__device__ __forceinline__ float reduce10(float a0, float a1, float a2, float a3, float a4, float a5, float a6,
                                          float a7, float a8, float a9, int lane) {
    const float b0 = swap32_add(a0, a1), b1 = swap32_add(a2, a3), b2 = swap32_add(a4, a5);
    const float b3 = swap32_add(a6, a7), b4 = swap32_add(a8, a9);
    float c0 = swap16_add(b0, b1);     // rows: a0 a2 a1 a3
    float c1 = swap16_add(b2, b3);     // rows: a4 a6 a5 a7
    float c2 = swz_add<16>(b4);        // rows: a8 a8 a9 a9 (b4 paired with itself: one LDS-crossbar exchange instead of a
                                       //                    register copy + v_permlane16_swap on the saturated VALU)
    c0 = row_sum_all(c0);
    c1 = row_sum_all(c1);
    c2 = row_sum_all(c2);
    const int pos = lane & 15;
    return pos == 15 ? c0 : (pos == 0 ? c1 : c2);
}

// Pose-only variant (colours and opacities take no gradient -- tracking): six sums, slots {SX, SY, SXX, SXY, SYY, DDEPTH}
//   lane 15: a0   lane 31: a2   lane 47: a1   lane 63: a3      (register c0)
//   lane  0: a4   lane 32: a5                                  (register c1)
__device__ __forceinline__ int reduce6_slot(int lane) {
    const int row = lane >> 4, pos = lane & 15;
    if (pos == 15) return row == 0 ? G_SX : row == 1 ? G_SXX : row == 2 ? G_SY : G_SXY;
    if (pos == 0) return row == 0 ? G_SYY : row == 2 ? G_DDEPTH : -1;
    return -1;
}
__device__ __forceinline__ float reduce6(float a0, float a1, float a2, float a3, float a4, float a5, int lane) {
    const float b0 = swap32_add(a0, a1), b1 = swap32_add(a2, a3), b2 = swap32_add(a4, a5);
    float c0 = swap16_add(b0, b1);     // rows: a0 a2 a1 a3
    float c1 = swz_add<16>(b2);        // rows: a4 a4 a5 a5
    c0 = row_sum_all(c0);
    c1 = row_sum_all(c1);
    return (lane & 15) == 15 ? c0 : c1;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}

// POSE_ONLY: the caller wants no gradient for colours / opacities (pose tracking against a fixed map): the colour and
//            opacity sums are neither formed nor reduced -- 6 values instead of 10 through the wave reduction.
// (Round 2 measured a batched reduction staged through LDS -- permlane32 swap, one ds_write_b32 per register pair, a
//  flush every 3 survivors in which each lane sums a 16-float segment -- against this in-register one: 20 % fewer VALU
//  instructions, but the batched atomics (3 gradient lines per instruction, or 3 instructions back to back) were no
//  longer hidden: 0.60 ms against 0.51 ms at C5, 0.48 ms with the atomics removed.  Also measured: two survivors per
//  loop trip (independent fetch / alpha / reduction streams interleaved for ILP): 0.536 vs 0.508 ms at C5, 85.9 vs
//  86.9 us at 100 k / VGA -- the waves are not bound by their dependent chains.  Re-measured: stopping the row butterfly
//  one step early and handing the atomic unit two lanes per value: 0.523 vs 0.506 ms.  DESIGN.md section 4.)
template <bool POSE_ONLY>
__global__ void __launch_bounds__(256) blend_backward_kernel(BlendArgs a, int ntiles,
                                                             const float* __restrict__ final_T,
                                                             const uint32_t* __restrict__ n_contrib,
                                                             const float* __restrict__ dL_dcolor,
                                                             const float* __restrict__ dL_ddepth,
                                                             float* __restrict__ grad_acc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = (int)blockIdx.x;
    if (tile >= ntiles) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const uint2 range = a.ranges[tile];
    if (range.y <= range.x) return;
    const size_t HW = (size_t)a.H * a.W;
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];

    // per-pixel state
    const int qx0i = tx * TILE + (wave & 1) * SUB, qy0i = ty * TILE + (wave >> 1) * SUB;
    const int pxi = qx0i + (lane & 7), pyi = qy0i + (lane >> 3);
    const bool inside = pxi < a.W && pyi < a.H;
    const size_t pix = (size_t)pyi * a.W + pxi;
    const float pxf = (float)pxi, pyf = (float)pyi, qx0 = (float)qx0i, qy0 = (float)qy0i;
    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    const float g0 = inside ? dL_dcolor[pix] : 0.f;
    const float g1 = inside ? dL_dcolor[HW + pix] : 0.f;
    const float g2 = inside ? dL_dcolor[2 * HW + pix] : 0.f;
    const float gd = inside ? dL_ddepth[pix] : 0.f;
    const float bgT = -T_final * (bg0 * g0 + bg1 * g1 + bg2 * g2);   // background term, per pixel
    // T: transmittance in front of the instance being processed.  Bk: sum over channels of (colour blended BEHIND the
    // instance) * dL/dpixel -- the recursion  Bk <- alpha q + (1 - alpha) Bk  folded over the instances already
    // walked (upstream keeps last_alpha / last_color and applies the same update one instance late).  Both updates
    // are the identity for alpha = 0, so an inactive lane needs no select on its state.
    float T = T_final, Bk = 0.f;
    const uint32_t maxc = wave_max_u32(last);   // wave-uniform
    if (maxc == 0) return;
    const uint32_t end = range.x + maxc;

    const int slot = POSE_ONLY ? reduce6_slot(lane) : reduce10_slot(lane);
    const uint32_t slot_bytes = slot < 0 ? 0u : (uint32_t)slot * 4u;

    uint32_t gid_n = 0;
    float4 box_n = make_float4(0.f, 0.f, -1.f, -1.f), ell_n = make_float4(0.f, 0.f, 0.f, 0.f);
    auto prefetch = [&](int b) {                    // next step's index + cull data, issued one step ahead
        gid_n = 0;
        box_n = make_float4(0.f, 0.f, -1.f, -1.f);
        const uint32_t i = range.x + (uint32_t)b * WAVE + lane;
        if (b >= 0 && i < end) {
            gid_n = a.point_list[i];
            box_n = a.rec[(size_t)gid_n * 4];
            ell_n = a.rec[(size_t)gid_n * 4 + 3];
        }
    };
    prefetch((int)((maxc - 1) / WAVE));
    for (int b = (int)((maxc - 1) / WAVE); b >= 0; --b) {
        const uint32_t gid_l = gid_n;
        const float4 c = box_n, el = ell_n;
        prefetch(b - 1);
        // pixels whose last contributor is at or behind this step's first instance (1-based position b*64 + 1)
        const unsigned long long alive = __builtin_amdgcn_ballot_w64(last >= (uint32_t)b * WAVE + 1u);
        unsigned long long mask = __builtin_amdgcn_ballot_w64(quadrant_hit(c, el, qx0, qy0, alive));
        // instance j of this step has the 1-based list position k = b * 64 + j + 1; "k <= last" as ONE compare of the scalar
        // j with a per-step register (the add of j to a per-lane base was a VALU instruction per survivor)
        const int rel_last = (int)last - (int)((uint32_t)b * WAVE + 1u);
        while (mask) {
            const int j = 63 - __builtin_clzll(mask);
            mask &= ~(1ull << j);
            const uint32_t gid = bcast(gid_l, j);
            const Rec g = fetch(a, gid);
            const float dx = g.px - pxf, dy = g.py - pyf;
            const float power = dx * (g.ca * dx + g.cb * dy) + (g.cc * dy) * dy;   // log2 of the Gaussian falloff
            const float G = __builtin_amdgcn_exp2f(power);
            const float alpha = fminf(0.99f, g.op * G);
            const bool act = (j <= rel_last) && !(power > 0.f) && !(alpha < 1.0f / 255.0f);
            if (__builtin_amdgcn_ballot_w64(act) == 0ull) continue;
            // Everything below runs for all 64 lanes; an inactive lane has alpha = 0: it contributes exact zeros
            // (w = 0, h = 0) and its state passes through (T * rcp(1) = T, Bk + 0 * diff = Bk).
            const float a_eff = act ? alpha : 0.f;
            const float inv = __builtin_amdgcn_rcpf(1.f - a_eff);   // v_rcp_f32, not the IEEE divide sequence
            const float Tn = T * inv;
            const float qq = __builtin_fmaf(g.z, gd, __builtin_fmaf(g.b, g2, __builtin_fmaf(g.g, g1, g.r * g0)));   // one chain: 4 VALU, not 5
            const float diff = qq - Bk;
            const float dL_dalpha = diff * Tn + bgT * inv;
            Bk = Bk + a_eff * diff;
            T = Tn;
            const float w = a_eff * Tn;
            const float h = act ? G * dL_dalpha : 0.f;   // g.op and the conic are applied per Gaussian later
            const float hx = h * dx, hy = h * dy;
            // partial sums of this instance over the wave's pixels (see the G_S* slots in common.h)
            const float s_x = hx, s_y = hy, s_xx = hx * dx, s_xy = hx * dy, s_yy = hy * dy, s_h = h;
            const float s_r = w * g0, s_g = w * g1, s_b = w * g2, s_z = w * gd;
            // ---- 10 wave sums (two swap stages + a ds_swizzle butterfly), then ONE atomic instruction with
            //      10 active lanes covering the Gaussian's 64-byte gradient line
            const float m = POSE_ONLY ? reduce6(s_x, s_y, s_xx, s_xy, s_yy, s_z, lane)
                                      : reduce10(s_x, s_y, s_xx, s_xy, s_yy, s_h, s_r, s_g, s_b, s_z, lane);
            // The line address is wave-uniform, the slot a per-lane constant: the atomic takes the line as its SGPR base and
            // the slot as its 32-bit VGPR offset (written as asm: the compiler folds the slot into a per-lane 64-bit base
            // and pays a 64-bit VALU add per survivor for the uniform part).  No-return, relaxed: what atomicAdd emitted.
            float* const line = grad_acc + (size_t)gid * GRAD_FLOATS;
            if (slot >= 0) asm volatile("global_atomic_add_f32 %0, %1, %2" ::"v"(slot_bytes), "v"(m), "s"(line) : "memory");
        }
    }
}

// =================================================================================================
// backward, "transposed" variant (round 3): no 64-lane reduction per survivor.
//
// The walk, the cull and the per-pixel arithmetic are those of blend_backward_kernel.  What changes is where the ten
// per-Gaussian sums are formed.  There, every surviving (instance, quadrant) pair pays a packed 64-lane reduction of ten
// values: 7 v_permlane*_swap (3.5 ns of SIMD issue each, tools/ubench/valu_rate.hip), 20 adds, 13 LDS-crossbar exchanges,
// 2 selects -- 55-60 ns of the ~125 ns a survivor costs.  Here a survivor only leaves its two per-pixel factors
//     h = G dL/dalpha        (geometry: the six raw moments are h x {dx, dy, dx^2, dx dy, dy^2, 1})
//     w = alpha T            (colour / depth: w x dL/dpixel)
// in a 4-slot slab of LDS (two ds_write_b32, lane = pixel).  Every fourth survivor the wave turns round: row r (16 lanes)
// of the wave takes slot r, lane q of the row takes the 4 horizontally adjacent pixels 4q..4q+3 of the quadrant (one
// ds_read_b128 per factor), forms the ten sums over ITS four pixels -- dy is shared by the four, dx steps by one, so the
// moments cost 17 plain VALU and the colours 16 -- and the 16 lanes of a row are then added up by a packed DPP butterfly
// (21 full-rate v_add_f32_dpp: the first two stages halve the register count with bank masks, row_ror:8 / row_half_mirror;
// the last two are quad permutes), which leaves the ten sums of the row's Gaussian in ten lanes of the row.  ONE atomic
// instruction then covers the four 64-byte gradient lines of the batch (40 active lanes, one memory-side request per
// line, as before).  Per survivor: ~10 + 11 VALU instead of ~29 expensive ones, nothing on the LDS crossbar, and the
// transposition is exact (same products, different summation order).
// =================================================================================================
constexpr int BT_SLOTS = 4;                      // survivors per batch = rows of 16 lanes
constexpr int BS_STRIDE = 272;                   // slot stride of blend_backward_s_kernel's slab and metadata (bytes)
constexpr int BS_W_OFF = 1280;                   // w factors behind the h factors: a multiple of 256 (ds_write2st64 offset) >= 4 x 272
constexpr int BS_META_OFF = BS_W_OFF + BT_SLOTS * BS_STRIDE;      // 2368
// which gradient slot the lane (t = lane & 15) of a row holds after reduce_rows (see there); -1: none
__device__ __forceinline__ int bt_slot10(int t) {
    const int bank = t >> 2, m = t & 3;
    if (m == 0) return bank == 3 ? 0 : bank == 1 ? 1 : bank == 2 ? 2 : 3;
    if (m == 1) return bank == 3 ? 4 : bank == 1 ? 5 : bank == 2 ? 6 : 7;
    if (m == 2) return bank == 3 ? 8 : bank == 1 ? 9 : -1;
    return -1;
}
__device__ __forceinline__ int bt_slot6(int t) {          // pose-only: v0..v5 = {SX, SY, SXX, SXY, SYY, DDEPTH}
    const int bank = t >> 2, m = t & 3;
    if (m == 0) return bank == 3 ? G_SX : bank == 1 ? G_SY : bank == 2 ? G_SXX : G_SXY;
    if (m == 1) return bank == 3 ? G_SYY : bank == 1 ? G_DDEPTH : -1;
    return -1;
}
// Sum ten values over the 16 lanes of every row.  On return lane t of a row holds, in the returned register,
//   t & 3 == 0:  bank 3: v0, bank 1: v1, bank 2: v2, bank 0: v3     (bank = t >> 2)
//   t & 3 == 1:  bank 3: v4, bank 1: v5, bank 2: v6, bank 0: v7
//   t & 3 == 2:  bank 3: v8, bank 1: v9
// Stage A (partner t ^ 8) packs two values into one register with the DPP bank mask (banks 2, 3 = lanes with bit 3 set take
// the first value, banks 0, 1 the second); stage B (partner 7 - (t & 7) in the same half row: opposite bit 2) packs again
// (banks 1, 3 / 0, 2); stages C, D add up the quads.  One asm block: the instructions are ordered so that no DPP operand
// was written by either of the two preceding instructions (the VALU -> DPP read hazard needs two wait states and the
// compiler does not look inside inline asm); the s_nop covers the block's inputs.
__device__ __forceinline__ float reduce_rows10(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                               float v7, float v8, float v9, unsigned long long m1, unsigned long long m2) {
    float a0, a1, a2, a3, a4;
    asm volatile(
        "s_nop 1\n\t"
        // stage A: (v0, v1) -> a0, (v2, v3) -> a1, (v4, v5) -> a2, (v6, v7) -> a3, (v8, v9) -> a4
        "v_add_f32_dpp %0, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %0, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %14, %14 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        // stage B: (a0, a1) -> v0's register ... the inputs are dead: reuse %5 (b0), %6 (b1), %7 (b2)
        "v_add_f32_dpp %5, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %5, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %6, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %7, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        // stage C: quad_perm [2,3,0,1]
        "v_add_f32_dpp %0, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        // stage D: quad_perm [1,0,3,2]
        "v_add_f32_dpp %5, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5),
          "+v"(v6), "+v"(v7), "+v"(v8), "+v"(v9));
    // v0 = {v0..v3 by bank}, v1 = {v4..v7 by bank}, v2 = {v8 | v9}: lane t & 3 picks 0 -> v0, 1 -> v1, else v2
    const float r12 = __builtin_amdgcn_inverse_ballot_w64(m1) ? v1 : v2;
    return __builtin_amdgcn_inverse_ballot_w64(m2) ? v0 : r12;
}
// six values (pose-only): stage A (v0, v1) -> a0, (v2, v3) -> a1, (v4, v5) -> a2; stage B (a0, a1) -> b0, a2 -> b1
//   t & 3 == 0:  bank 3: v0, bank 1: v1, bank 2: v2, bank 0: v3;   t & 3 == 1:  banks 2, 3: v4, banks 0, 1: v5
__device__ __forceinline__ float reduce_rows6(float v0, float v1, float v2, float v3, float v4, float v5,
                                              unsigned long long m2) {
    float a0, a1, a2;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %4, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %3, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5));
    return __builtin_amdgcn_inverse_ballot_w64(m2) ? v0 : v1;
}

struct __attribute__((aligned(16))) BtMetaRec { float x, y; uint32_t g; };
// What a lane needs to turn a batch round (all by value: a by-reference lambda ended up as a closure object in scratch).
struct BtLane {
    const float* fac;               // this wave's factor slab [h | w][slot][pixel] (LDS)
    const float* pix;               // this wave's dL/dpixel block [(rgb,) depth][q][4 pixels] (LDS)
    const BtMetaRec* meta;          // this wave's metadata [slot] {centre x, centre y, Gaussian index} (LDS)
    int lane;
    float qx0, qy0;                 // first pixel of the quadrant (wave-uniform)
    uint32_t slot_bytes;
    unsigned long long m_out, m_q0, m_q1;
    float* grad_acc;
    int qxi = 0, qyi = 0;           // the same origin as integers (WIDE_META flush)
};
template <bool POSE_ONLY, bool WIDE_META = false>     // WIDE_META: metadata records 256 bytes apart, holding index << 6 (blend_backward_s_kernel)
__device__ __forceinline__ void bt_flush(const BtLane L, int n) {
    // Everything a lane derives from its number here (row, q, three LDS addresses, its first pixel) is RE-derived per batch,
    // ~8 VALU per four survivors: hoisted out of the walk by the compiler these values cost six more VGPRs over the whole
    // kernel, i.e. a wave per SIMD.  The empty asm keeps the loop-invariant code motion from undoing that.
    int lane = L.lane;
    asm volatile("" : "+v"(lane));
    const int q = lane & 15, row = lane >> 4;
    // [row][4 q .. 4 q + 3]; WIDE_META (blend_backward_s_kernel): rows BS_STRIDE bytes apart in all three arrays
    const float* const hp = WIDE_META ? L.fac + row * (BS_STRIDE / 4) + 4 * q : L.fac + 4 * lane;
    const float4 h4 = *reinterpret_cast<const float4*>(hp);
    const float4 w4 = *reinterpret_cast<const float4*>(hp + (WIDE_META ? BS_W_OFF / 4 : BT_SLOTS * WAVE));
    const BtMetaRec me = L.meta[WIDE_META ? row * (BS_STRIDE / 16) : row];
    // [channel][q][4]: the sixteen lanes of a row read 256 contiguous bytes per channel (conflict-free; laid out [q][channel]
    // the 64-byte lane stride put four lanes on every bank: 48 M conflict cycles per launch at C5, r03 PMC)
    const float* const fp = L.pix + q * 4;
    float u0, v0;
    if (WIDE_META) {                    // the origin stays in scalar INTEGER registers (see blend_backward_s_kernel); same values
        int oqx = L.qxi, oqy = L.qyi;
        asm volatile("" : "+s"(oqx), "+s"(oqy));
        u0 = (float)(oqx + 4 * (q & 1)), v0 = (float)(oqy + (q >> 1));
    } else {
        u0 = L.qx0 + (float)(4 * (q & 1)), v0 = L.qy0 + (float)(q >> 1);
    }
    const float dy = me.y - v0;
    const float dx0 = me.x - u0, dx1 = dx0 - 1.f, dx2 = dx0 - 2.f, dx3 = dx0 - 3.f;
    const float hx0 = h4.x * dx0, hx1 = h4.y * dx1, hx2 = h4.z * dx2, hx3 = h4.w * dx3;
    const float H0 = (h4.x + h4.y) + (h4.z + h4.w);
    const float H1 = (hx0 + hx1) + (hx2 + hx3);
    const float H2 = __builtin_fmaf(hx3, dx3, __builtin_fmaf(hx2, dx2, __builtin_fmaf(hx1, dx1, hx0 * dx0)));
    const float s_y = dy * H0, s_xy = dy * H1, s_yy = dy * s_y;
    float m;
    if (POSE_ONLY) {
        const float4 c3 = *reinterpret_cast<const float4*>(fp);
        const float s_z = __builtin_fmaf(w4.w, c3.w, __builtin_fmaf(w4.z, c3.z, __builtin_fmaf(w4.y, c3.y, w4.x * c3.x)));
        m = reduce_rows6(H1, s_y, H2, s_xy, s_yy, s_z, L.m_q0);
    } else {
        const float4 c0 = *reinterpret_cast<const float4*>(fp), c1 = *reinterpret_cast<const float4*>(fp + 64),
                     c2 = *reinterpret_cast<const float4*>(fp + 128), c3 = *reinterpret_cast<const float4*>(fp + 192);
        const float s_r = __builtin_fmaf(w4.w, c0.w, __builtin_fmaf(w4.z, c0.z, __builtin_fmaf(w4.y, c0.y, w4.x * c0.x)));
        const float s_g = __builtin_fmaf(w4.w, c1.w, __builtin_fmaf(w4.z, c1.z, __builtin_fmaf(w4.y, c1.y, w4.x * c1.x)));
        const float s_b = __builtin_fmaf(w4.w, c2.w, __builtin_fmaf(w4.z, c2.z, __builtin_fmaf(w4.y, c2.y, w4.x * c2.x)));
        const float s_z = __builtin_fmaf(w4.w, c3.w, __builtin_fmaf(w4.z, c3.z, __builtin_fmaf(w4.y, c3.y, w4.x * c3.x)));
        m = reduce_rows10(H1, s_y, H2, s_xy, s_yy, H0, s_r, s_g, s_b, s_z, L.m_q1, L.m_q0);
    }
    // rows >= n hold a stale slot: masked out.  One instruction, <= 40 active lanes, one 64-byte request per gradient line.
    const unsigned long long rows = n >= BT_SLOTS ? ~0ull : ((1ull << (16 * n)) - 1ull);
    const uint32_t off = (WIDE_META ? me.g : me.g * (uint32_t)(GRAD_FLOATS * sizeof(float))) + L.slot_bytes;
    if (__builtin_amdgcn_inverse_ballot_w64(L.m_out & rows))
        asm volatile("global_atomic_add_f32 %0, %1, %2" ::"v"(off), "v"(m), "s"(L.grad_acc) : "memory");
}

template <bool POSE_ONLY>
__global__ void __launch_bounds__(256) blend_backward_t_kernel(BlendArgs a, int ntiles,
                                                               const float* __restrict__ final_T,
                                                               const uint32_t* __restrict__ n_contrib,
                                                               const float* __restrict__ dL_dcolor,
                                                               const float* __restrict__ dL_ddepth,
                                                               float* __restrict__ grad_acc) {
    // [wave][factor h / w][slot][pixel]: 2 KB per wave
    __shared__ __attribute__((aligned(16))) float s_fac[4][2][BT_SLOTS][WAVE];
    __shared__ BtMetaRec s_meta[4][BT_SLOTS];                                            // [wave][slot] {cx, cy, index}
    __shared__ __attribute__((aligned(16))) float s_pix[4][POSE_ONLY ? 1 : 4][16][4];    // [wave][(rgb,) depth][q][4 pixels]
    // (the wave number is wave-uniform but computed from a per-lane register: readfirstlane tells the compiler, and the
    //  quadrant origin and the LDS bases derived from it then live in SGPRs -- five VGPRs less)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int tile = (int)blockIdx.x;
    if (tile >= ntiles) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const uint2 range = a.ranges[tile];
    if (range.y <= range.x) return;
    const size_t HW = (size_t)a.H * a.W;
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];

    // per-pixel state (lane = pixel of the 8x8 quadrant, as in blend_backward_kernel)
    const int qx0i = tx * TILE + (wave & 1) * SUB, qy0i = ty * TILE + (wave >> 1) * SUB;
    const int pxi = qx0i + (lane & 7), pyi = qy0i + (lane >> 3);
    const bool inside = pxi < a.W && pyi < a.H;
    const size_t pix = (size_t)pyi * a.W + pxi;
    const float pxf = (float)pxi, pyf = (float)pyi, qx0 = (float)qx0i, qy0 = (float)qy0i;
    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    const float g0 = inside ? dL_dcolor[pix] : 0.f;
    const float g1 = inside ? dL_dcolor[HW + pix] : 0.f;
    const float g2 = inside ? dL_dcolor[2 * HW + pix] : 0.f;
    const float gd = inside ? dL_ddepth[pix] : 0.f;
    const float bgT = -T_final * (bg0 * g0 + bg1 * g1 + bg2 * g2);
    float T = T_final, Bk = 0.f;
    const uint32_t maxc = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(last));   // wave-uniform, in an SGPR
    if (maxc == 0) return;
    const uint32_t end = range.x + maxc;

    // the turned-round view: row = slot of the batch, lane q of the row = pixels 4q .. 4q+3 of the quadrant
    const int q = lane & 15;
    const int fxi = qx0i + 4 * (q & 1), fyi = qy0i + (q >> 1);
    // dL/dpixel of the lane's four pixels, constants of the launch, live in LDS, [channel][q][pixel] (pose-only: depth only)
    // -- in registers they pushed the kernel past 64 VGPRs -- and come back with four (one) broadcast ds_read_b128 per batch:
    // the four rows read the same addresses.
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in_i = (fxi + i) < a.W && fyi < a.H;
            const size_t px_i = (size_t)fyi * a.W + fxi + i;
            if (!POSE_ONLY) {
                s_pix[wave][0][q][i] = in_i ? dL_dcolor[px_i] : 0.f;
                s_pix[wave][1][q][i] = in_i ? dL_dcolor[HW + px_i] : 0.f;
                s_pix[wave][2][q][i] = in_i ? dL_dcolor[2 * HW + px_i] : 0.f;
            }
            s_pix[wave][POSE_ONLY ? 0 : 3][q][i] = in_i ? dL_ddepth[px_i] : 0.f;
        }
    }
    const int slot = POSE_ONLY ? bt_slot6(q) : bt_slot10(q);
    const uint32_t slot_bytes = slot < 0 ? 0u : (uint32_t)slot * 4u;
    const unsigned long long m_out = __builtin_amdgcn_ballot_w64(slot >= 0);            // lanes that carry a sum
    const unsigned long long m_q0 = __builtin_amdgcn_ballot_w64((q & 3) == 0), m_q1 = __builtin_amdgcn_ballot_w64((q & 3) == 1);
    float* const fhl = &s_fac[wave][0][0][lane];          // this lane's column of the factor slab: [k * WAVE] = slot k
    BtMetaRec* const fmeta = &s_meta[wave][0];
    const BtLane bl{&s_fac[wave][0][0][0], &s_pix[wave][0][0][0], fmeta, lane, qx0, qy0, slot_bytes, m_out, m_q0, m_q1, grad_acc};
    int k = 0;                      // survivors in the open batch (wave-uniform)

    uint32_t gid_n = 0;
    float4 box_n = make_float4(0.f, 0.f, -1.f, -1.f), ell_n = make_float4(0.f, 0.f, 0.f, 0.f);
    auto prefetch = [&](int b) {
        gid_n = 0;
        box_n = make_float4(0.f, 0.f, -1.f, -1.f);
        const uint32_t i = range.x + (uint32_t)b * WAVE + lane;
        if (b >= 0 && i < end) {
            gid_n = a.point_list[i];
            box_n = a.rec[(size_t)gid_n * 4];
            ell_n = a.rec[(size_t)gid_n * 4 + 3];
        }
    };
    prefetch((int)((maxc - 1) / WAVE));
    for (int b = (int)((maxc - 1) / WAVE); b >= 0; --b) {
        const uint32_t gid_l = gid_n;
        const float4 c = box_n, el = ell_n;
        prefetch(b - 1);
        const unsigned long long alive = __builtin_amdgcn_ballot_w64(last >= (uint32_t)b * WAVE + 1u);
        unsigned long long mask = __builtin_amdgcn_ballot_w64(quadrant_hit(c, el, qx0, qy0, alive));
        const uint32_t k_first = (uint32_t)b * WAVE + 1u;          // 1-based list position of instance 0 of this step
        // one survivor: its two per-pixel factors into slot k of the batch (lane = pixel); whose they are -- centre and index
        // of the Gaussian -- is noted by lane j, which has held all three in registers since it loaded instance j's cull box
        auto apply = [&](const Rec& g, int j, float power, float G, float alpha) {
            const bool act = (k_first + (uint32_t)j <= last) && !(power > 0.f) && !(alpha < 1.0f / 255.0f);
            if (__builtin_amdgcn_ballot_w64(act) == 0ull) return;
            const float a_eff = act ? alpha : 0.f;
            const float inv = __builtin_amdgcn_rcpf(1.f - a_eff);    // (an IEEE divide changes no gradient digit: tools/grad_accuracy.py)
            const float Tn = T * inv;
            const float qq = __builtin_fmaf(g.z, gd, __builtin_fmaf(g.b, g2, __builtin_fmaf(g.g, g1, g.r * g0)));
            const float diff = qq - Bk;
            const float dL_dalpha = diff * Tn + bgT * inv;
            Bk = Bk + a_eff * diff;
            T = Tn;
            const float w = a_eff * Tn;
            const float h = act ? G * dL_dalpha : 0.f;
            fhl[k * WAVE] = h;
            fhl[(BT_SLOTS + k) * WAVE] = w;
            if (__builtin_amdgcn_inverse_ballot_w64(1ull << j)) fmeta[k] = BtMetaRec{c.x, c.y, gid_l};
            if (++k == BT_SLOTS) {
                bt_flush<POSE_ONLY>(bl, BT_SLOTS);
                k = 0;
            }
        };
        while (mask) {
            const int j = 63 - __builtin_clzll(mask);
            mask &= ~(1ull << j);
            const Rec g = fetch(a, bcast(gid_l, j));
            const float dx = g.px - pxf, dy = g.py - pyf;
            const float power = dx * (g.ca * dx + g.cb * dy) + (g.cc * dy) * dy;
            const float G = __builtin_amdgcn_exp2f(power);
            apply(g, j, power, G, fminf(0.99f, g.op * G));
        }
    }
    if (k) bt_flush<POSE_ONLY>(bl, k);
}

// =================================================================================================
// backward, round 5: the transposed accumulation of round 3 with its scalar side cut down and its per-pixel side under EXEC.
//
// What round 3 / 4 left per survivor in front of the flush: 31 vector and ~27 scalar-pipe instructions (s_flbit + s_xor for
// the next mask bit, v_readlane of the index, two 64-bit shifts, a 64-bit add, three s_load in two round trips, two waits,
// the mask algebra of the activity test, a saveexec for the one-lane metadata store, address arithmetic for it, the batch
// counter).  tools/ubench/valu_rate.hip (round 5 rows) prices the scalar pipe: a compute unit has ONE scalar ALU for its four
// SIMDs -- 2.2 ns of SIMD time per scalar instruction alone, ~0.65 ns when it rides behind vector work; kernel time
// ~= V + 0.3 S reproduces the measured 90 ns per survivor from 75 ns of vector and 48 ns of scalar issue.  (The forward's answer
// -- records through a per-wave LDS queue -- was built for this kernel too, compacted to 32 entries so that eight workgroups
// per compute unit stay: 0.383 against 0.339 ms at C5.  The slab and the flush already keep the CU's LDS pipe busy, three
// broadcast ds_read_b128 per survivor from four SIMDs are another 48 LDS cycles per survivor slot)
// So the record stays in scalar registers and everything around it goes:
//   * lane l holds instance 63 - l of the step: walking the step back to front is walking the mask from bit 0, `s_ff1` gives
//     the lane, `s_bitset0` clears it (2 scalar instructions instead of 5);
//   * the lane keeps index << 6 -- byte offset of the 64-byte record AND of the gradient line -- so the one v_readlane yields
//     the SGPR offset of two scalar loads {px, py}, {conic, opacity, colour, depth} off one base: one round trip, no address
//     arithmetic;
//   * "list position <= last contributor" is `lane >= 63 - (last - k_first)` against a per-step register; the three activity
//     tests narrow EXEC (v_cmpx) inside ONE asm block, state updates and factors are plain instructions under it (no v_cndmask,
//     no mask algebra), the slab slot is cleared by an LDS store under the full EXEC first (a wave's LDS operations run in
//     order), the batch slot advances only if a pixel was active (s_cselect on EXEC != 0), and lane j stores the slot's
//     metadata under EXEC = 1 << j.
// 29 vector + ~16 scalar-pipe instructions per survivor in front of the flush.  Same arithmetic in the same order as
// blend_backward_t_kernel (the A/B partner, option 1): gradients equal to the last bit of every sum's order.
// =================================================================================================
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v3f __attribute__((ext_vector_type(3)));
typedef float v8f __attribute__((ext_vector_type(8)));
template <bool POSE_ONLY>
__global__ void __launch_bounds__(256) blend_backward_s_kernel(BlendArgs a, int ntiles,
                                                               const float* __restrict__ final_T,
                                                               const uint32_t* __restrict__ n_contrib,
                                                               const float* __restrict__ dL_dcolor,
                                                               const float* __restrict__ dL_ddepth,
                                                               float* __restrict__ grad_acc) {
    // per wave: the factor slab [h | w][slot][pixel] and the slots' metadata {cx, cy, index << 6}, every array with the SAME slot
    // stride so that ONE scalar (slab base + open slot x stride) addresses all three -- 272 bytes, not 256, so that the four rows of
    // a flush find their metadata words on different LDS banks.  (SQ_LDS_BANK_CONFLICT reads 7.9 M cycles per launch at C5 -- two per
    // survivor -- with either stride, and with the slot-clearing or the factor store taken out: not isolated, and not worth more.)
    struct WaveLds { float h[BT_SLOTS][BS_STRIDE / 4]; float pad[(BS_W_OFF - BT_SLOTS * BS_STRIDE) / 4]; float w[BT_SLOTS][BS_STRIDE / 4];
                     BtMetaRec meta[BT_SLOTS][BS_STRIDE / 16]; };
    static_assert(sizeof(WaveLds) == BS_META_OFF + BT_SLOTS * BS_STRIDE, "layout");
    __shared__ __attribute__((aligned(16))) WaveLds s_w[MGS_WG_WAVES];
    __shared__ __attribute__((aligned(16))) float s_pix[MGS_WG_WAVES][POSE_ONLY ? 1 : 4][16][4];    // [wave][(rgb,) depth][q][4 pixels]
    MGS_TILE_WAVE(ntiles, tile, wave, ws)
    const int lane = threadIdx.x & 63;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const uint2 range = a.ranges[tile];
    if (range.y <= range.x) return;
    const size_t HW = (size_t)a.H * a.W;
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];

    const int qx0i = tx * TILE + (wave & 1) * SUB, qy0i = ty * TILE + (wave >> 1) * SUB;
    const int pxi = qx0i + (lane & 7), pyi = qy0i + (lane >> 3);
    const bool inside = pxi < a.W && pyi < a.H;
    const size_t pix = (size_t)pyi * a.W + pxi;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const float T_final = inside ? final_T[pix] : 0.f;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    const float g0 = inside ? dL_dcolor[pix] : 0.f;
    const float g1 = inside ? dL_dcolor[HW + pix] : 0.f;
    const float g2 = inside ? dL_dcolor[2 * HW + pix] : 0.f;
    const float gd = inside ? dL_ddepth[pix] : 0.f;
    const float bgT = -T_final * (bg0 * g0 + bg1 * g1 + bg2 * g2);
    float T = T_final, Bk = 0.f;
    const uint32_t maxc = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(last));
    if (maxc == 0) return;
    const uint32_t end = range.x + maxc;
#ifdef BS_TRACE
    unsigned long long tr0;
    uint32_t tr_n = 0;
    BS_STAMP(tr0);
#endif

    const int q = lane & 15;
    const int fxi = qx0i + 4 * (q & 1), fyi = qy0i + (q >> 1);
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in_i = (fxi + i) < a.W && fyi < a.H;
            const size_t px_i = (size_t)fyi * a.W + fxi + i;
            if (!POSE_ONLY) {
                s_pix[ws][0][q][i] = in_i ? dL_dcolor[px_i] : 0.f;
                s_pix[ws][1][q][i] = in_i ? dL_dcolor[HW + px_i] : 0.f;
                s_pix[ws][2][q][i] = in_i ? dL_dcolor[2 * HW + px_i] : 0.f;
            }
            s_pix[ws][POSE_ONLY ? 0 : 3][q][i] = in_i ? dL_ddepth[px_i] : 0.f;
        }
    }
    const int slot = POSE_ONLY ? bt_slot6(q) : bt_slot10(q);
    const uint32_t slot_bytes = slot < 0 ? 0u : (uint32_t)slot * 4u;
    const unsigned long long m_out = __builtin_amdgcn_ballot_w64(slot >= 0);
    const unsigned long long m_q0 = __builtin_amdgcn_ballot_w64((q & 3) == 0), m_q1 = __builtin_amdgcn_ballot_w64((q & 3) == 1);
    const BtLane bl{&s_w[ws].h[0][0], &s_pix[ws][0][0][0], &s_w[ws].meta[0][0], lane, 0.f, 0.f, slot_bytes, m_out, m_q0, m_q1, grad_acc, qx0i, qy0i};
    const uint32_t wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)&s_w[ws]);   // LDS address (SGPR)
    const const_float_p recs = MGS_CONST(reinterpret_cast<const float*>(a.rec));
    uint32_t koff = 0;              // open batch: survivors x 272 = byte offset of the open slot in the slab and in the metadata (SGPR)
    float zero = 0.f;
    asm volatile("" : "+v"(zero));  // (a VGPR holding 0: data operand of the slot-clearing LDS store)

    uint32_t gid_n = 0;
    float4 box_n = make_float4(0.f, 0.f, -1.f, -1.f), ell_n = make_float4(0.f, 0.f, 0.f, 0.f);
    auto prefetch = [&](int b) {                    // lane l takes instance 63 - l of step b
        gid_n = 0;
        box_n = make_float4(0.f, 0.f, -1.f, -1.f);
        const uint32_t i = range.x + (uint32_t)b * WAVE + (uint32_t)(63 - lane);
        if (b >= 0 && i < end) {
            gid_n = a.point_list[i];
            box_n = a.rec[(size_t)gid_n * 4];
            ell_n = a.rec[(size_t)gid_n * 4 + 3];
        }
    };
    prefetch((int)((maxc - 1) / WAVE));
    for (int b = (int)((maxc - 1) / WAVE); b >= 0; --b) {
        const float4 c = box_n, el = ell_n;
        v3f meta = {c.x, c.y, __uint_as_float(gid_n << 6)};        // what the lane notes for the flush if its instance survives
        prefetch(b - 1);
        const unsigned long long alive = __builtin_amdgcn_ballot_w64(last >= (uint32_t)b * WAVE + 1u);
        // (the quadrant origin as floats is re-derived per step from the scalar integers: kept across the walk the two floats
        //  sit in vector registers, and the kernel has exactly 64)
        int qxs = qx0i, qys = qy0i;
        asm volatile("" : "+s"(qxs), "+s"(qys));
        unsigned long long mask = __builtin_amdgcn_ballot_w64(quadrant_hit(c, el, (float)qxs, (float)qys, alive));
        // lane j holds the instance with list position k_first + 63 - j: a pixel takes it only if that is <= last,
        // i.e. j >= 63 + k_first - last
        const int thr = 63 + (int)((uint32_t)b * WAVE + 1u) - (int)last;
        // One survivor: pop the next set bit (the lane), fetch its record with two scalar loads off the lane's index << 6.
        // The fetch of survivor t + 1 is ISSUED before survivor t is evaluated and only waited for one trip later (two
        // register sets, the loop unrolled by two: no copies).  A scalar load that misses the 16 KB scalar cache takes ~240 ns
        // (tools/ubench/valu_rate.hip, "s_load_dwordx8 x2 + wait"), a survivor ~60 ns of this wave's own issue: with the wait
        // right behind the load, eight waves per SIMD are just enough to cover it and six (the average residency) are not.
        // (Scalar loads return out of order, so the only wait there is is lgkmcnt(0): it sits at the TOP of a trip, before the
        //  next fetch is issued.)
#define BS_POP(J, R0, R1)                                                                                                    \
        asm volatile("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0" : "=&s"(J), "+s"(mask));                                 \
        {                                                                                                                    \
            const uint32_t goff = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(meta[2]), J);                     \
            asm volatile("s_load_dwordx2 %0, %2, %3\n\ts_load_dwordx8 %1, %2, %3 offset:0x10"                                \
                         : "=&s"(R0), "=&s"(R1) : "s"(recs), "s"(goff) : "memory");                                          \
        }
        auto evaluate = [&](const v2f r0, const v8f r1, const int j) {
                float dx, dy, t1, t2, G, al, inv, qq;
                uint32_t adv, sa, tmp;
                asm volatile(
                    "v_sub_f32_e32 %[dx], %[px], %[pxf]\n\t"
                    "v_sub_f32_e32 %[dy], %[py], %[pyf]\n\t"
                    "v_mul_f32_e32 %[t1], %[cb], %[dy]\n\t"
                    "v_mul_f32_e32 %[t2], %[cc], %[dy]\n\t"
                    "v_fmac_f32_e32 %[t1], %[ca], %[dx]\n\t"
                    "v_mul_f32_e32 %[t2], %[dy], %[t2]\n\t"
                    "v_fmac_f32_e32 %[t2], %[dx], %[t1]\n\t"               // power (log2 of the falloff), as the other kernels form it
                    "v_exp_f32_e32 %[G], %[t2]\n\t"
                    "s_add_u32 %[tmp], %[wbase], %[koff]\n\t"               // the open slot
                    "v_lshl_add_u32 %[sa], %[lane], 2, %[tmp]\n\t"           // this lane's word of it
                    "ds_write2st64_b32 %[sa], %[zero], %[zero] offset1:5\n\t"   // the slot's h and w of every pixel <- 0 (full EXEC)
                    "v_cmpx_ge_i32_e32 vcc, %[j], %[thr]\n\t"              // EXEC: pixels whose last contributor is not in front of j
                    "v_cmpx_nlt_f32_e32 vcc, 0, %[t2]\n\t"                 //       and power <= 0
                    "v_mul_f32_e32 %[al], %[op], %[G]\n\t"
                    "v_min_f32_e32 %[al], 0x3f7d70a4, %[al]\n\t"           // min(0.99, opacity G)
                    "v_cmpx_ngt_f32_e32 vcc, %[amin], %[al]\n\t"           //       and alpha >= 1/255
                    "v_sub_f32_e32 %[inv], 1.0, %[al]\n\t"
                    "v_rcp_f32_e32 %[inv], %[inv]\n\t"
                    "v_mul_f32_e32 %[qq], %[cr], %[g0]\n\t"
                    "v_fmac_f32_e32 %[qq], %[cg], %[g1]\n\t"
                    "v_fmac_f32_e32 %[qq], %[cb2], %[g2]\n\t"
                    "v_fmac_f32_e32 %[qq], %[cz], %[gd]\n\t"
                    "v_mul_f32_e32 %[T], %[T], %[inv]\n\t"                 // T <- T / (1 - alpha): transmittance in front of j
                    "v_sub_f32_e32 %[qq], %[qq], %[Bk]\n\t"                // diff
                    "v_mul_f32_e32 %[dx], %[T], %[qq]\n\t"
                    "v_fmac_f32_e32 %[dx], %[bgT], %[inv]\n\t"             // dL/dalpha
                    "v_mul_f32_e32 %[dx], %[G], %[dx]\n\t"                 // h = G dL/dalpha
                    "v_mul_f32_e32 %[dy], %[al], %[T]\n\t"                 // w = alpha T
                    "v_fmac_f32_e32 %[Bk], %[al], %[qq]\n\t"               // Bk <- Bk + alpha diff
                    "ds_write2st64_b32 %[sa], %[dx], %[dy] offset1:5\n\t"  // (active pixels only)
                    "s_cmp_lg_u64 exec, 0\n\t"
                    "s_cselect_b32 %[adv], 272, 0\n\t"                     // the slot is taken only if some pixel was active
                    "s_lshl_b64 exec, 1, %[j]\n\t"                         // lane j: the slot's metadata {centre, index << 6}
                    "v_mov_b32_e32 %[sa], %[tmp]\n\t"
                    "ds_write_b96 %[sa], %[meta] offset:2368\n\t"
                    "s_mov_b64 exec, -1\n\t"
                    : [dx] "=&v"(dx), [dy] "=&v"(dy), [t1] "=&v"(t1), [t2] "=&v"(t2), [G] "=&v"(G), [al] "=&v"(al), [inv] "=&v"(inv),
                      [qq] "=&v"(qq), [sa] "=&v"(sa), [T] "+v"(T), [Bk] "+v"(Bk), [adv] "=&s"(adv), [tmp] "=&s"(tmp)
                    : [px] "s"(r0[0]), [py] "s"(r0[1]), [ca] "s"(r1[0]), [cb] "s"(r1[1]), [cc] "s"(r1[2]), [op] "s"(r1[3]),
                      [cr] "s"(r1[4]), [cg] "s"(r1[5]), [cb2] "s"(r1[6]), [cz] "s"(r1[7]), [pxf] "v"(pxf), [pyf] "v"(pyf), [thr] "v"(thr),
                      [j] "s"(j), [g0] "v"(g0), [g1] "v"(g1), [g2] "v"(g2), [gd] "v"(gd), [bgT] "v"(bgT), [lane] "v"(lane), [zero] "v"(zero),
                      [koff] "s"(koff), [wbase] "s"(wbase), [meta] "v"(meta), [amin] "s"(1.0f / 255.0f)
                    : "vcc", "scc", "memory");
                koff += adv;
            if (koff == BT_SLOTS * BS_STRIDE) {
                bt_flush<POSE_ONLY, true>(bl, BT_SLOTS);
                koff = 0;
            }
        };
#ifdef BS_TRACE
        tr_n += (uint32_t)__popcll(mask);
#endif
        if (mask) {
            int jA, jB = 0;
            v2f a0, b0;
            v8f a1, b1;
            BS_POP(jA, a0, a1)
            for (;;) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1) : : "memory");        // record A has arrived
                const bool moreB = mask != 0ull;
                if (moreB) { BS_POP(jB, b0, b1) }
                evaluate(a0, a1, jA);
                if (!moreB) break;
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1) : : "memory");        // record B has arrived
                const bool moreA = mask != 0ull;
                if (moreA) { BS_POP(jA, a0, a1) }
                evaluate(b0, b1, jB);
                if (!moreA) break;
            }
        }
#undef BS_POP
    }
    if (koff) bt_flush<POSE_ONLY, true>(bl, (int)(koff >> 8));
#ifdef BS_TRACE
    if (a.trace) {
        unsigned long long tr1;
        BS_STAMP(tr1);
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        if (lane == 0) {
            unsigned long long* o = a.trace + ((size_t)tile * 4 + wave) * 4;
            o[0] = tr0; o[1] = tr1; o[2] = (unsigned long long)hw | ((unsigned long long)xcc << 32); o[3] = tr_n;
        }
    }
#endif
}

// ---- diagnostic: what the backward walk does, counted (not on the hot path; mgs_debug_blend_stats) --------------
// The same walk, cull and per-pixel activity test as blend_backward_kernel<1, *>, with counters instead of gradient
// arithmetic: stats[0] steps of 64 instances, [1] instances that pass the quadrant cull and are fetched ("survivors"),
// [2] survivors with at least one active pixel ("active survivors" = wave reductions = atomic instructions),
// [3] active (pixel, instance) pairs, [4] inactive survivors whose cull box holds no pixel with k <= last (depth
// order, not geometry), [5..7] active survivors with <= 2 / <= 4 / <= 8 active pixels.
__global__ void __launch_bounds__(256) blend_backward_stats_kernel(BlendArgs a, int ntiles,
                                                                   const uint32_t* __restrict__ n_contrib,
                                                                   unsigned long long* __restrict__ stats) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = (int)blockIdx.x;
    if (tile >= ntiles) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const uint2 range = a.ranges[tile];
    if (range.y <= range.x) return;
    const int qx0i = tx * TILE + (wave & 1) * SUB, qy0i = ty * TILE + (wave >> 1) * SUB;
    const int pxi = qx0i + (lane & 7), pyi = qy0i + (lane >> 3);
    const bool inside = pxi < a.W && pyi < a.H;
    const float pxf = (float)pxi, pyf = (float)pyi, qx0 = (float)qx0i, qy0 = (float)qy0i;
    const uint32_t last = inside ? n_contrib[(size_t)pyi * a.W + pxi] : 0u;
    const uint32_t maxc = wave_max_u32(last);
    if (maxc == 0) return;
    const uint32_t end = range.x + maxc;
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = (int)((maxc - 1) / WAVE); b >= 0; --b) {
        const uint32_t i = range.x + (uint32_t)b * WAVE + lane;
        uint32_t gid_l = 0;
        float4 box = make_float4(0.f, 0.f, -1.f, -1.f), el = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < end) {
            gid_l = a.point_list[i];
            box = a.rec[(size_t)gid_l * 4];
            el = a.rec[(size_t)gid_l * 4 + 3];
        }
        const unsigned long long alive = __builtin_amdgcn_ballot_w64(last >= (uint32_t)b * WAVE + 1u);
        unsigned long long mask = __builtin_amdgcn_ballot_w64(quadrant_hit(box, el, qx0, qy0, alive));
        c[0] += 1;
        c[1] += __popcll(mask);
        while (mask) {
            const int j = 63 - __builtin_clzll(mask);
            mask &= ~(1ull << j);
            const uint32_t k = (uint32_t)b * WAVE + (uint32_t)j + 1u;
            const Rec g = fetch(a, bcast(gid_l, j));
            const float dx = g.px - pxf, dy = g.py - pyf;
            const float power = dx * (g.ca * dx + g.cb * dy) + (g.cc * dy) * dy;
            const float alpha = fminf(0.99f, g.op * __builtin_amdgcn_exp2f(power));
            const bool geo = !(power > 0.f) && !(alpha < 1.0f / 255.0f);
            const unsigned long long act = __builtin_amdgcn_ballot_w64(geo && k <= last);
            const int n = __popcll(act);
            if (n == 0) {
                if (__builtin_amdgcn_ballot_w64(geo) != 0ull) c[4] += 1;
                continue;
            }
            c[2] += 1;
            c[3] += n;
            c[5] += n <= 2;
            c[6] += n <= 4;
            c[7] += n <= 8;
        }
    }
    if (lane < 8) {
        unsigned long long v = c[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) v = lane == q ? c[q] : v;
        atomicAdd(stats + lane, v);
    }
}

// ---- diagnostic, round 5: how many loop trips would the walk take if the wave ran SEVERAL survivor streams side by side,
// one per group of pixels?  39 of a survivor's 64 lanes are active on average (stats[3] / stats[2]) and narrowing EXEC saves
// no time on gfx950, so the only way to use the idle lanes is to give them another survivor.  For five ways of cutting the
// 8x8 quadrant into groups (two 8x4 halves, two 4x8 halves, four 4x4 blocks, four 8x2 strips, eight 4x2 blocks) the cull is
// run per group (the same box + ellipse test against the group's rectangle and its alive pixels) and three sums are kept:
//   [8 + 3 d + 0]  sum over steps of max_g S_g      trips when the groups' lists are paired step by step
//   [8 + 3 d + 1]  sum over steps of sum_g S_g      (group, survivor) rows: what the backward would have to flush
//   [8 + 3 d + 2]  sum over walks of max_g sum_steps S_g   trips when every group runs down a list of its own
// to be read against stats[1] (survivors of the whole-quadrant cull = trips now).
__device__ __forceinline__ bool rect_hit(const float4 c, const float4 n, float qx0, float qy0, int ox, int oy, int w, int h,
                                         unsigned long long alive) {
    if (c.z < 0.f) return false;
    const float x0q = qx0 - 0.05f - c.x, y0q = qy0 - 0.05f - c.y;                      // quadrant origin relative to the centre
    const float x0 = x0q + (float)ox, x1 = qx0 + (float)(ox + w - 1) + 0.05f - c.x;   // the group's rectangle
    const float y0 = y0q + (float)oy, y1 = qy0 + (float)(oy + h - 1) + 0.05f - c.y;
    if (!((c.z >= x0) && (-c.z <= x1) && (c.w >= y0) && (-c.w <= y1))) return false;
    {
        const int ix0 = (int)fminf(8.f, fmaxf(0.f, ceilf(-x0q - c.z - 0.06f))), ix1 = (int)fminf(7.f, fmaxf(-1.f, floorf(-x0q + c.z + 0.06f)));
        const int iy0 = (int)fminf(8.f, fmaxf(0.f, ceilf(-y0q - c.w - 0.06f))), iy1 = (int)fminf(7.f, fmaxf(-1.f, floorf(-y0q + c.w + 0.06f)));
        if (ix0 > ix1 || iy0 > iy1) return false;
        const unsigned long long cols = (unsigned long long)((0xFFu >> (7 - ix1)) & (0xFFu << ix0) & 0xFFu) * 0x0101010101010101ull;
        const unsigned long long rows = (~0ull >> (8 * (7 - iy1))) & (~0ull << (8 * iy0));
        const unsigned long long gcols = (unsigned long long)(((1u << w) - 1u) << ox) * 0x0101010101010101ull;
        const unsigned long long grows = ((h >= 8 ? ~0ull : ((1ull << (8 * h)) - 1ull)) << (8 * oy));
        if ((cols & rows & gcols & grows & alive) == 0ull) return false;
    }
    if (x0 <= 0.f && x1 >= 0.f && y0 <= 0.f && y1 >= 0.f) return true;
    if (!(n.x > 0.f) || !(n.z > 0.f)) return true;
    const float rby = -n.y * __builtin_amdgcn_rcpf(n.z), rbx = -n.y * __builtin_amdgcn_rcpf(n.x);
    auto edge_x = [&](float xe) { const float t = fminf(y1, fmaxf(y0, rby * xe)); return n.x * xe * xe + 2.f * n.y * xe * t + n.z * t * t; };
    auto edge_y = [&](float ye) { const float t = fminf(x1, fmaxf(x0, rbx * ye)); return n.x * t * t + 2.f * n.y * t * ye + n.z * ye * ye; };
    return fminf(fminf(edge_x(x0), edge_x(x1)), fminf(edge_y(y0), edge_y(y1))) <= 1.02f;
}
__global__ void __launch_bounds__(256) blend_group_stats_kernel(BlendArgs a, int ntiles, const uint32_t* __restrict__ n_contrib,
                                                                unsigned long long* __restrict__ stats) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = (int)blockIdx.x;
    if (tile >= ntiles) return;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const uint2 range = a.ranges[tile];
    if (range.y <= range.x) return;
    const int qx0i = tx * TILE + (wave & 1) * SUB, qy0i = ty * TILE + (wave >> 1) * SUB;
    const int pxi = qx0i + (lane & 7), pyi = qy0i + (lane >> 3);
    const bool inside = pxi < a.W && pyi < a.H;
    const float qx0 = (float)qx0i, qy0 = (float)qy0i;
    const uint32_t last = inside ? n_contrib[(size_t)pyi * a.W + pxi] : 0u;
    const uint32_t maxc = wave_max_u32(last);
    if (maxc == 0) return;
    const uint32_t end = range.x + maxc;
    // decomposition d: NG[d] groups of GW[d] x GH[d] pixels, GX[d] groups per row of groups
    constexpr int ND = 5;
    constexpr int NG[ND] = {2, 2, 4, 4, 8}, GW[ND] = {8, 4, 4, 8, 4}, GH[ND] = {4, 8, 4, 2, 2}, GX[ND] = {1, 2, 2, 1, 2};
    unsigned long long trips_step[ND] = {0, 0, 0, 0, 0}, rows[ND] = {0, 0, 0, 0, 0};
    uint32_t walk[ND][8];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int g = 0; g < 8; ++g) walk[d][g] = 0;
    for (int b = (int)((maxc - 1) / WAVE); b >= 0; --b) {
        const uint32_t i = range.x + (uint32_t)b * WAVE + lane;
        float4 box = make_float4(0.f, 0.f, -1.f, -1.f), el = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < end) {
            const uint32_t gid_l = a.point_list[i];
            box = a.rec[(size_t)gid_l * 4];
            el = a.rec[(size_t)gid_l * 4 + 3];
        }
        const unsigned long long alive = __builtin_amdgcn_ballot_w64(last >= (uint32_t)b * WAVE + 1u);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            int mx = 0, sum = 0;
#pragma unroll
            for (int g = 0; g < NG[d]; ++g) {
                const int ox = (g % GX[d]) * GW[d], oy = (g / GX[d]) * GH[d];
                const int n = __popcll(__builtin_amdgcn_ballot_w64(rect_hit(box, el, qx0, qy0, ox, oy, GW[d], GH[d], alive)));
                mx = max(mx, n);
                sum += n;
                walk[d][g] += (uint32_t)n;
            }
            trips_step[d] += mx;
            rows[d] += sum;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            uint32_t mx = 0;
#pragma unroll
            for (int g = 0; g < NG[d]; ++g) mx = max(mx, walk[d][g]);
            atomicAdd(stats + 8 + 3 * d + 0, trips_step[d]);
            atomicAdd(stats + 8 + 3 * d + 1, rows[d]);
            atomicAdd(stats + 8 + 3 * d + 2, (unsigned long long)mx);
        }
    }
}

int launch_blend_backward_stats(const mgs_camera& cam, const GeometryState& g, const BinningState& b,
                                const ImageState& img, unsigned long long* stats, hipStream_t s) {
    const BlendArgs a = make_args(cam, g, b, img);
    const int ntiles = a.gx * tiles_y(a.H);
    if (ntiles == 0) return 0;
    hipLaunchKernelGGL(blend_backward_stats_kernel, dim3(ntiles), dim3(256), 0, s, a, ntiles, img.n_contrib, stats);
    hipLaunchKernelGGL(blend_group_stats_kernel, dim3(ntiles), dim3(256), 0, s, a, ntiles, img.n_contrib, stats);
    MGS_HIP(hipGetLastError());
    return 0;
}

// (The wave-per-tile and half-tile-per-wave variants of round 1 lost at every size and are gone; DESIGN.md section 4
//  keeps their measurements.)
int g_opt_blend_lds_pad_fwd = 0, g_opt_blend_lds_pad_bwd = 0;      // measurement knobs: dynamic LDS bytes the kernels never touch (fewer workgroups per CU)
int g_opt_blend_bwd_transposed = 2;     // mgs_debug_set_option("blend_bwd_transposed", 0 | 1 | 2): 2 = scalar side trimmed + EXEC (round 5), 1 = scalar-fetch transposed (round 3), 0 = the per-survivor wave reduction
int launch_blend_backward(const mgs_camera& cam, const GeometryState& g, const BinningState& b,
                          const ImageState& img, const float* dL_dcolor, const float* dL_ddepth, float* grad_acc,
                          bool pose_only, hipStream_t s) {
    BlendArgs a = make_args(cam, g, b, img);
#ifdef BS_TRACE
    a.trace = g_trace_bwd;
#endif
    const int ntiles = a.gx * tiles_y(a.H);
    if (ntiles == 0) return 0;
    if (g_opt_blend_bwd_transposed == 2) {
        const dim3 grid(ntiles), block(256);
        if (pose_only)
            hipLaunchKernelGGL((blend_backward_s_kernel<true>), grid, block, (size_t)g_opt_blend_lds_pad_bwd, s, a, ntiles, img.final_T,
                               img.n_contrib, dL_dcolor, dL_ddepth, grad_acc);
        else
            hipLaunchKernelGGL((blend_backward_s_kernel<false>), grid, block, (size_t)g_opt_blend_lds_pad_bwd, s, a, ntiles, img.final_T,
                               img.n_contrib, dL_dcolor, dL_ddepth, grad_acc);
    } else if (g_opt_blend_bwd_transposed != 0) {
        if (pose_only)
            hipLaunchKernelGGL((blend_backward_t_kernel<true>), dim3(ntiles), dim3(256), 0, s, a, ntiles, img.final_T,
                               img.n_contrib, dL_dcolor, dL_ddepth, grad_acc);
        else
            hipLaunchKernelGGL((blend_backward_t_kernel<false>), dim3(ntiles), dim3(256), 0, s, a, ntiles, img.final_T,
                               img.n_contrib, dL_dcolor, dL_ddepth, grad_acc);
    } else if (pose_only)
        hipLaunchKernelGGL((blend_backward_kernel<true>), dim3(ntiles), dim3(256), 0, s, a, ntiles, img.final_T,
                           img.n_contrib, dL_dcolor, dL_ddepth, grad_acc);
    else
        hipLaunchKernelGGL((blend_backward_kernel<false>), dim3(ntiles), dim3(256), 0, s, a, ntiles, img.final_T,
                           img.n_contrib, dL_dcolor, dL_ddepth, grad_acc);
    MGS_HIP(hipGetLastError());
    return 0;
}

// ---- diagnostic: the plain-FMA issue ceiling of this chip, measured by whoever asks (bench.py prints it next to the spec
// VALU peak).  8 waves per SIMD, every CU busy, 8 independent v_fma_f32 per trip: tools/ubench/valu_rate.hip's first row.
__global__ void __launch_bounds__(256) valu_ceiling_kernel(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float c = 1.0001f, d = 0.0001f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_fmaf(a0, c, d); a1 = __builtin_fmaf(a1, c, d); a2 = __builtin_fmaf(a2, c, d); a3 = __builtin_fmaf(a3, c, d);
        a4 = __builtin_fmaf(a4, c, d); a5 = __builtin_fmaf(a5, c, d); a6 = __builtin_fmaf(a6, c, d); a7 = __builtin_fmaf(a7, c, d);
    }
    out[blockIdx.x * 256 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}
int launch_valu_ceiling(float* out, int iters, hipStream_t s) {
    hipLaunchKernelGGL(valu_ceiling_kernel, dim3(MGS_VALU_CEILING_BLOCKS), dim3(256), 0, s, out, iters, 1.0f);
    MGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace mgs

#ifdef BS_TRACE
extern "C" int mgs_trace_set_blend_buffers(void* fwd, void* bwd) {      // diagnostic builds only: device buffers of 4 x tiles x 4 uint64, or NULL
    mgs::g_trace_fwd = (unsigned long long*)fwd;
    mgs::g_trace_bwd = (unsigned long long*)bwd;
    return 0;
}
#endif
