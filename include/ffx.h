/* ffx.h — C ABI of the MI355X-native Fireflies hot path.
 *
 * The reference (Henningson/Fireflies, pure Python) has no FFI: its hot path sits behind a
 * Python object protocol (torch ops + Mitsuba's `mi.render` / `scene.ray_intersect` /
 * `params.update()`).  Each entry point below cites the reference call it replaces
 * (paths relative to /root/reference).  The reference-side binding a maintainer would add
 * (ctypes) is shown in INTEGRATION.md.
 *
 * Two shared libraries implement exactly this header:
 *   fireflies_amd/csrc/libffx_hip.so   the product: hand-written HIP kernels for gfx950.
 *   oracle/_build/libffx_oracle.so      test infrastructure: scalar CPU restatement.
 *
 * Conventions
 *   - every function returns FFX_OK (0) or a negative error code; ffx_last_error() gives a
 *     thread-local message for the last failure on the calling thread.
 *   - the caller owns every buffer; the library allocates no device memory.  Pointers marked
 *     [dev] are device pointers for libffx_hip and host pointers for the oracle; pointers
 *     marked [host] are always host pointers (small parameter blocks, copied into kernel
 *     arguments at launch).
 *   - all calls are asynchronous on `stream` (a hipStream_t; the oracle ignores it).
 *   - arrays are dense, row-major, C order.  float = IEEE binary32.
 *   - no global state besides the error string; calls on distinct streams are re-entrant.
 *   - process environment read by libffx_hip.so at every ffx_trace_primary / ffx_render_* call.  None of these
 *     changes a result (tests/test_hip_parity.py runs every variant against the oracle); they select between
 *     equivalent kernels / launch shapes and exist for A/B measurements:
 *       FFX_TRAVERSAL=lane       per-lane kernels (LDS stack, apex vectors per ray) instead of the wave-packet kernels
 *       FFX_WIDE=0               wave-packet kernels on the binary walk only (default: 64-wide walk with binary fallback)
 *       FFX_PIXELS_PER_WAVE=1|2|4  pixels of a 2x2 tile one wavefront renders (default 1; 2 when the adjoint cache is written)
 *       FFX_TILE_BLOCK=0..8      log2 side of the square blocks in which tiles are enumerated (default 3)
 *       FFX_XCD_REMAP=0|1|B      workgroup -> tile mapping across the 8 XCDs: 0 round-robin, 1 one image band per XCD,
 *                                B >= 2: each XCD takes B consecutive work items of every 8 B (default 128)
 *       FFX_DUMMY_LDS=bytes      extra dynamic LDS per workgroup (occupancy experiments)
 *       FFX_K7_PPW_LOG2=0..6     cap on log2(pixels per wavefront) of ffx_trace_primary (default 4 at 1 spp, else 3)
 *       FFX_BINS=0               the packet render kernels walk the tree for every packet (default: tile bins first, ffx_bvh_info.off_bins)
 *       FFX_BIN_TILE=4..32       side of a camera tile of the bins in pixels (default 8)
 *       FFX_SHADOW_CLEAR=1|2|3   (off by default) the pre-pass also proves per triangle that nothing can shadow it from the projector (1) / the
 *                                spot (2) and marks it in its per-slot normal's flag word (ffx_bvh_info.off_gn); a pixel whose samples all
 *                                lie on such triangles skips that emitter's shadow stage — exact, tested, but the proof costs the loop more
 *                                than the skip gains on the default workload (fireflies_amd/csrc/ffx_trace.hip clear_enabled)
 *       FFX_ENVELOPE=0|1|2|3     (round 6; default 3) the emitters whose tile grid also gets an ENVELOPE from the pre-pass — bit 0 projector, bit 1
 *                                spot: per cell of the grid (7 x 7 per tile) one plane in front of every triangle the tile lists there; a shadow
 *                                packet whose segments all end in front of their cells' planes skips that emitter's any-hit stage (exact: the
 *                                image is the same bit for bit; 87 % of the vocal fold's shadow packets, K8 0.39 -> 0.34 ms).  0: none (the A/B
 *                                baseline).  A caller's hint in ffx_scene_desc.shadows (FFX_SHADOWS_PLAIN) leaves them out per pose.
 *       FFX_RFC_CAP=n            blocks of the filtered film's adjoint cache a forward may take, at most what the cache holds (a test knob: pixels
 *                                that find the arena full keep no records and are counted in `dropped` — the overflow path)
 *       FFX_BIN_CAP=n            capacity of each grid's entry list, at most the default 2 F + 16384 (a test knob: a grid whose lists do
 *                                not fit is marked not-ok by the pre-pass and its packets take the tree walks — the overflow path)
 *       FFX_RENDER_BLOCKS=0      ffx_render_fwd / ffx_render_fwd_filtered below 33 samples per pixel: a pixel per wave whatever the count, as
 *                                at 64 (default: compact blocks of up to 8 pixels per wave, k_render_fwd_blk — the same image bit for bit,
 *                                1.5 - 2.3x faster; FFX_RENDER_BLK_LOG2=n: at most 2^n pixels per wave of the box film, an experiment knob)
 *     ffx_bvh_build_host additionally reads, once per build (host side; the renders do not depend on them —
 *     tests/test_hip_parity.py::test_wide_overlay_builders_give_identical_images):
 *       FFX_WIDE_BUILD=area|count|layers   builder of the 64-wide overlay (default area: greedy SAH cut)
 *       FFX_WIDE_CLUSTER=4..64, FFX_WIDE_COST_EXP=x   experiment knobs of that builder (largest cluster; priority area * count^x)
 *     The Python host layer reads FFX_LIB (alternative BUILD of this library), FFX_ASYNC_UPDATE=0 (single BVH blob,
 *     refit on the caller's stream), FFX_CACHE_LIMIT_GB (adjoint cache budget), FFX_HOST_PHILOX=0 (sampler draws on the
 *     device instead of ffx_torch_rand_h) and then FFX_PREDRAW=0, FFX_SIDE_STREAMS=n (a side stream per BVH blob copy: a measured loss, default 1),
 *     FFX_PATTERN_STEP=0 (the optimiser's step with ffx_pattern_bwd_blur + ffx_pattern_fwd_blur instead of ffx_pattern_step), FFX_K9_L1=0 (ffx_l1_value_grad +
 *     ffx_render_bwd_cached instead of ffx_render_bwd_cached_l1), FFX_STEP_STREAMS=1 (the fused renders of a multi-sample step one after the other on the
 *     caller's stream instead of in turn on the scene's two render streams), FFX_DEFER_TOP=n (FFX_STEP_DEFER_TOP while the last render had at most n samples
 *     per pixel; default 8, 0: never); ffx_pattern_step itself reads FFX_ADAM_POW_CACHE=0 (pow() every step);
 *     bench.py reads FFX_DIST_BACKEND and FFX_BENCH_TIMED_STEPS.
 */
#ifndef FFX_H
#define FFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFX_ABI_VERSION 10
#define FFX_MAX_LEVELS 96

typedef void *ffx_stream; /* hipStream_t */

enum {
  FFX_OK = 0,
  FFX_ERR_ARG = -1,         /* bad argument (null pointer, non-positive size, ...) */
  FFX_ERR_LAUNCH = -2,      /* HIP launch / runtime failure */
  FFX_ERR_UNSUPPORTED = -3, /* valid request this build cannot serve */
  FFX_ERR_NOMEM = -4        /* caller-provided buffer too small */
};

enum { FFX_REDUCE_SUM = 0, FFX_REDUCE_SOFTOR = 1 };

const char *ffx_last_error(void);
int ffx_abi_version(void);
/* "hip-gfx950" for the product, "cpu-oracle" for the oracle. */
const char *ffx_backend(void);

/* ------------------------------------------------------------------------------------------
 * K1  pattern projection.
 * Replaces Laser.projectRaysToNDC (fireflies/projection/laser.py:262-275) =
 * transform_points(rays, K @ FLIP_Y) (fireflies/utils/math.py:220-228): q = KF·[r;1],
 * pts = q.xyz / q.w.  `KF` [host] is the 4x4 row-major product K @ FLIP_Y.
 * The backward is the autograd of the same expression w.r.t. rays.
 * ---------------------------------------------------------------------------------------- */
int ffx_project_rays_fwd(const float *rays /*[dev][n,3]*/, int n, const float *KF /*[host][16]*/,
                         float *pts /*[dev][n,3]*/, ffx_stream stream);
int ffx_project_rays_bwd(const float *rays /*[dev][n,3]*/, int n, const float *KF /*[host][16]*/,
                         const float *gpts /*[dev][n,3]*/, float *grays /*[dev][n,3]*/,
                         ffx_stream stream);

/* transform_points / transform_directions (fireflies/utils/math.py:220-235).
 * mode 0: homogeneous transform with perspective divide; mode 1: direction (w = 0, no divide). */
int ffx_transform_points(const float *pts /*[dev][n,3]*/, int n, const float *M /*[host][16]*/,
                         int mode, float *out /*[dev][n,3]*/, ffx_stream stream);

/* Constraint projection of the pattern after an optimiser step, in place:
 * Laser.clamp_to_fov (fireflies/projection/laser.py:199-206): ndc = transform_points(rays, KF);
 * ndc.xy = clamp(ndc.xy, lo, hi); rays = normalize(transform_points(ndc, KF_inv)); followed by
 * n_normalize - 1 further normalisations (Laser.normalize_rays, laser.py:254-255; the training loops
 * call both).  lo = 1 - clamp_val, hi = clamp_val. */
int ffx_clamp_to_fov(float *rays /*[dev][n,3] in/out*/, int n, const float *KF /*[host][16]*/,
                     const float *KF_inv /*[host][16]*/, float lo, float hi, int n_normalize, ffx_stream stream);

/* ------------------------------------------------------------------------------------------
 * K2  soft point splatting.
 * texture_size = (size0, size1); point n = (p0, p1) in [0,1]^2; outputs are indexed [i][j]
 * with i in [0,size1) paired with p1 and j in [0,size0) paired with p0:
 *     v(n,i,j) = exp(-(((j - p0*size0)^2 + (i - p1*size1)^2) / sigma)^2)
 * (fireflies/graphics/rasterization.py:7-37; note the d^4/sigma^2 falloff).
 * ---------------------------------------------------------------------------------------- */

/* rasterize_points (rasterization.py:7-37): dense [n,size1,size0] layers. */
int ffx_splat_dense_fwd(const float *pts /*[dev][n,2]*/, int n, float sigma, int size0, int size1,
                        float *out /*[dev][n,size1,size0]*/, ffx_stream stream);
/* autograd of rasterize_points w.r.t. points for an upstream gradient on the dense layers. */
int ffx_splat_dense_bwd(const float *pts /*[dev][n,2]*/, int n, float sigma, int size0, int size1,
                        const float *gout /*[dev][n,size1,size0]*/, float *gpts /*[dev][n,2]*/,
                        ffx_stream stream);

/* Fused splat + reduce over the point axis, never materialising [n,size1,size0]:
 *   reduce = FFX_REDUCE_SUM    : sum(rasterize_points(...), 0)        (rasterization.py:160-161)
 *   reduce = FFX_REDUCE_SOFTOR : 1 - prod(1 - rasterize_points(...))  (rasterization.py:156-157)
 * half_window < 0  : every term (terms that are exactly 0 in binary32 are skipped, so the
 *                    result equals the dense reduction up to summation order).
 * half_window >= 0 : the reference's footprint-limited variants baked_sum / baked_softor
 *                    (rasterization.py:164-237, 321-392): point n only touches the odd window
 *                    of half width `half_window` around floor(p*size), with the reference's
 *                    border clipping; `sigma` is then what those functions call sigma (the
 *                    caller passes the already squared value, rasterization.py:577).
 * Output orientation is that of baked_sum / sum(dense): tex[size1][size0].               */
int ffx_splat_fwd(const float *pts /*[dev][n,2]*/, int n, float sigma, int reduce, int half_window,
                  int size0, int size1, float *tex /*[dev][size1,size0]*/, ffx_stream stream);
/* gradient of the fused op w.r.t. points.  `tex` is the forward output (read for softor). */
int ffx_splat_bwd(const float *pts /*[dev][n,2]*/, int n, float sigma, int reduce, int half_window,
                  int size0, int size1, const float *tex /*[dev][size1,size0]*/,
                  const float *gtex /*[dev][size1,size0]*/, float *gpts /*[dev][n,2]*/,
                  ffx_stream stream);

/* rasterize_depth (rasterization.py:66-104): dense layers normalised by their own maximum and
 * scaled by depth[n].  Forward only plus gradient w.r.t. (pts, depth). */
int ffx_splat_depth_fwd(const float *pts /*[dev][n,2]*/, const float *depth /*[dev][n]*/, int n,
                        float sigma, int size0, int size1, float *out /*[dev][n,size1,size0]*/,
                        ffx_stream stream);

/* rasterize_lines (rasterization.py:107-153): soft line segments, lines [n,2,2] =
 * (start,end) x (c0,c1) in [0,1]^2, output [n,size1,size0] (the reference is only
 * self-consistent for size0 == size1; other sizes follow its broadcasting). */
int ffx_splat_lines_fwd(const float *lines /*[dev][n,2,2]*/, int n, float sigma, int size0,
                        int size1, float *out /*[dev][n,size1,size0]*/, ffx_stream stream);
/* autograd of rasterize_lines w.r.t. the segments for an upstream gradient on the layers (the reference
 * optimises segments through it: rasterization.py:645-743).  The three branches of the distance (before the
 * start, along the segment, past the end; :147-151) are differentiated where they are selected; inside the
 * segment the derivative of the projection parameter t0 is included, as torch.autograd does. */
int ffx_splat_lines_bwd(const float *lines /*[dev][n,2,2]*/, int n, float sigma, int size0, int size1,
                        const float *gout /*[dev][n,size1,size0]*/, float *glines /*[dev][n,2,2]*/,
                        ffx_stream stream);

/* Overlap regulariser of the reference's point-pattern loop, L1Loss(softor, sum)
 * (fireflies/graphics/rasterization.py:579,589-600), with its gradient:
 *   ws[0] = weight * mean(|a - b|);   g = weight * sign(a - b) / n   (= d ws[0] / d a;  d ws[0] / d b = -g)
 * ws: [dev][257] floats — [0] the value, [1..256] scratch for the two-pass reduction (fixed order). */
int ffx_l1_value_grad(const float *a /*[dev][n]*/, const float *b /*[dev][n]*/, long n, float weight,
                      float *ws /*[dev][257]*/, float *g /*[dev][n]*/, ffx_stream stream);
/* ... and acc[0] += ws[0] in the same launches (acc NULL: as above): the running loss of a step's scene samples when the task loss is an
 * L1 against a target image (optim.image_l1_loss) — one launch less per sample than adding the value afterwards */
int ffx_l1_value_grad_acc(const float *a /*[dev][n]*/, const float *b /*[dev][n]*/, long n, float weight,
                          float *ws /*[dev][257]*/, float *g /*[dev][n]*/, float *acc /*[dev][1] or NULL*/, ffx_stream stream);

/* ------------------------------------------------------------------------------------------
 * The pattern side of one optimisation step, fused (the loop of rasterization.py:583-607 with the projector in
 * front: Laser.projectRaysToNDC laser.py:262-275 -> rasterize_points + sum / softor rasterization.py:7-37,156-161
 * -> L1Loss(softor, sum) :589-600 -> backward -> Adam -> Laser.clamp_to_fov / normalize_rays laser.py:199-206,
 * 254-255).  Same arithmetic and summation order as the separate entry points above; a 64..1024-point pattern
 * makes every one of those launch-bound.
 *   ffx_pattern_fwd : pts = (KF rays).xy / w;  tsum = sum_n v_n;  tsor = 1 - prod_n (1 - v_n) (if want_softor),
 *                     ws[ffx_pattern_ws_floats(size0, size1)] = partial sums of |tsor - tsum| (any partition).
 *   ffx_pattern_bwd : grays_data = d/d rays of <gts, tsum>  (gts = upstream gradient on the SUM texture; NULL: skipped);
 *                     grays_reg  = d/d rays of reg_weight * mean|tsor - tsum|;  reg_value[0] = that value and, if
 *                     loss_in is given, reg_value[1] = sum(loss_in) / loss_div + reg_value[0] (the step's total loss) and
 *                     reg_value[2] = sum(loss_in).
 *   ffx_adam_clamp_step : g = grad / grad_div + grad_b (grad_b may be NULL; g is stored in grad_out, which may be NULL
 *                     if there is nothing to combine), then torch.optim.Adam's update of rays with (exp_avg,
 *                     exp_avg_sq, step [dev][1], incremented; lr / betas / eps as the doubles torch holds), then
 *                     ffx_clamp_to_fov(rays, ..., n_normalize) — one launch.
 * ---------------------------------------------------------------------------------------- */
size_t ffx_pattern_ws_floats(int size0, int size1);
int ffx_pattern_fwd(const float *rays /*[dev][n,3]*/, int n, const float *KF /*[host][16]*/, float sigma, int size0, int size1,
                    int want_softor, float *pts /*[dev][n,2]*/, float *tsum /*[dev][size1,size0]*/,
                    float *tsor /*[dev][size1,size0] or NULL*/, float *ws /*[dev] or NULL*/,
                    float *zero /*[dev][n_zero] or NULL: cleared by the same launch — the step's accumulation buffer
                                  (texture gradient + loss), which would otherwise cost a fill launch of its own*/,
                    long n_zero, ffx_stream stream);
int ffx_pattern_bwd(const float *rays /*[dev][n,3]*/, int n, const float *KF /*[host][16]*/, float sigma, int size0, int size1,
                    const float *tsum /*[dev]*/, const float *tsor /*[dev] or NULL*/, const float *gts /*[dev] or NULL*/,
                    float reg_weight, const float *ws /*[dev] or NULL*/, float *grays_data /*[dev][n,3] or NULL*/,
                    float *grays_reg /*[dev][n,3] or NULL*/, float *reg_value /*[dev][3] or NULL*/,
                    const float *loss_in /*[dev][loss_in_n] or NULL: partial sums of the data term (ffx_render_bwd_cached's dot slots, or one value)*/,
                    int loss_in_n, float loss_div, ffx_stream stream);
int ffx_adam_clamp_step(float *rays /*[dev][n,3] in/out*/, const float *grad /*[dev][n,3]*/, const float *grad_b /*[dev][n,3] or NULL*/,
                        float grad_div, float *grad_out /*[dev][n,3] or NULL*/, float *exp_avg /*[dev][n,3]*/,
                        float *exp_avg_sq /*[dev][n,3]*/, float *step /*[dev][1]*/, int n, double lr, double beta1, double beta2,
                        double eps, const float *KF /*[host][16]*/, const float *KF_inv /*[host][16]*/, float lo, float hi,
                        int n_normalize, const void *guard /*[dev] or NULL (ABI 7): as ffx_adam_args.guard — when the 32-bit word at byte 8 is not
                        zero the update is NOT applied (rays, both moments and the step count keep their values).  Either an adjoint cache's header
                        (its `dropped` count) or, behind a multi-rank exchange, the address of the third-last float of the all-reduced flat buffer whose
                        LAST float is the sum over ranks of their dropped counts: no rank then applies a poisoned update*/,
                        ffx_stream stream);
/* The same two launches carrying their neighbours along (round 3: every launch of the pattern side sits on the critical path of
 * a step, ~5 us each behind a 0.54 ms render):
 *   ffx_pattern_fwd_blur : ffx_pattern_fwd + tex = ffx_blur_fwd(tsum, blur_ksize, blur_sigma) in ONE launch (a11: the texture
 *                     finalise of examples/vocalfold_scene.py:59-63) — the workgroup of a 32x8 tile evaluates its halo too and blurs
 *                     from LDS.  tsum, tsor, ws, tex: bitwise the separate calls' (ksize 5; other sizes run the two launches).
 *   ffx_pattern_bwd_blur : ffx_pattern_bwd where gtex is d loss / d tex of the BLURRED texture: K3^T (ffx_blur_bwd) is applied inside
 *                     the gradient launch over the points' footprints only (bitwise the separate calls' gradient; blur_ksize 0: gtex
 *                     is the gradient on tsum, as ffx_pattern_bwd's gts).  gts_scratch [size1,size0] is only used when a footprint
 *                     does not fit the workgroup's LDS or the image is smaller than the kernel (then the transpose blur runs as its own
 *                     launch into it); may be NULL otherwise.
 *                     adam != NULL: the workgroup that finishes last applies ffx_adam_clamp_step(rays, grays_data, grays_reg,
 *                     grad_div, grad_out, ...) — the whole backward half of a single-process step is then ONE launch.  `counter`:
 *                     one device word, zero before the first call (the launch leaves it zero).  A multi-rank step exchanges the
 *                     gradient between the two and keeps them apart (adam = NULL).
 * ---------------------------------------------------------------------------------------- */
typedef struct ffx_adam_args {
  float *rays;        /* [dev][n,3] in/out: must be the `rays` argument of the call */
  float *exp_avg;     /* [dev][n,3]; NULL (with exp_avg_sq, step NULL): NO update — only the inner product below is evaluated (a multi-rank
                         step, which exchanges the gradient before it updates: ffx_adam_clamp_step afterwards) */
  float *exp_avg_sq;  /* [dev][n,3] */
  float *step;        /* [dev][1], incremented */
  float *grad_out;    /* [dev][n,3] or NULL (needed when there is anything to combine: grays_reg or grad_div != 1) */
  uint32_t *counter;  /* [dev][1], zero */
  double lr, beta1, beta2, eps;
  float KF_inv[16];
  float lo, hi;       /* Laser.clamp_to_fov's bounds */
  float grad_div;
  int32_t n_normalize;
  /* optional: the step's data term as an inner product <dot_a, dot_b> over dot_n floats (a loss linear in the image: the render and its
   * constant gradient), evaluated by the same launch — every workgroup sums a slice into dot_partial[n], the workgroup that applies the
   * update adds them up: reg_value[2] = the sum, reg_value[1] = sum / loss_div + reg_value[0].  Takes the place of loss_in (not both).
   * K8's own partial sums (ffx_render_fwd_adjoint's dot_out) cost a quarter of a million atomics per render: 27 us; this costs 3. */
  const float *dot_a; /* [dev][dot_n] or NULL */
  const float *dot_b; /* [dev][dot_b_n] */
  int64_t dot_n;
  float *dot_partial; /* [dev][n] scratch */
  int64_t dot_b_n;    /* period of dot_b: <a, b> = sum_i a[i] * b[i mod dot_b_n] (the S renders of a step stacked in dot_a against ONE
                         constant gradient); 0 = dot_n */
  /* optional (ABI 5): the adjoint cache the step's gradient came from (ffx_render_fwd_cache's `cache`: its header).  When that header's
   * `dropped` word is not zero — the arena of single-sample records overflowed, ffx_render_bwd_cached has poisoned gtex with NaN — the
   * update is NOT applied: rays, exp_avg, exp_avg_sq and step keep their values, and the caller, who finds out with
   * ffx_render_cache_status, can repeat the step with the re-tracing adjoint instead of finding its optimiser state full of NaN. */
  const void *guard;  /* [dev] or NULL */
} ffx_adam_args;
int ffx_pattern_fwd_blur(const float *rays /*[dev][n,3]*/, int n, const float *KF /*[host][16]*/, float sigma, int size0, int size1,
                         int want_softor, float *pts /*[dev][n,2]*/, float *tsum /*[dev][size1,size0]*/,
                         float *tsor /*[dev] or NULL*/, float *ws /*[dev] or NULL*/, float *zero /*[dev][n_zero] or NULL*/, long n_zero,
                         int blur_ksize, float blur_sigma, float *tex /*[dev][size1,size0]*/, ffx_stream stream);
int ffx_pattern_bwd_blur(const float *rays /*[dev][n,3]*/, int n, const float *KF /*[host][16]*/, float sigma, int size0, int size1,
                         const float *tsum /*[dev]*/, const float *tsor /*[dev] or NULL*/, const float *gtex /*[dev][size1,size0] or NULL*/,
                         float reg_weight, const float *ws /*[dev] or NULL*/, float *grays_data /*[dev][n,3] or NULL*/,
                         float *grays_reg /*[dev][n,3] or NULL*/, float *reg_value /*[dev][3] or NULL*/,
                         const float *loss_in /*[dev][loss_in_n] or NULL*/, int loss_in_n, float loss_div, int blur_ksize,
                         float blur_sigma, float *gts_scratch /*[dev][size1,size0] or NULL*/, const ffx_adam_args *adam /*[host] or NULL*/,
                         ffx_stream stream);

/* The pattern side of an optimisation step as ONE launch (ABI 10): ffx_pattern_bwd_blur(..., adam) of step s — K3^T, K2-bwd, K1-bwd, the
 * step's loss, torch.optim.Adam.step(), Laser.clamp_to_fov() + normalize_rays() — followed, inside the same launch, by ffx_pattern_fwd_blur of
 * step s + 1 on the UPDATED rays: pts, tsum, tsor, ws, tex are overwritten with the next step's texture and `zero` (the next step's accumulator,
 * normally gtex itself + the loss slots + the adjoint cache's header) is cleared — after the gradient has been taken from it.
 * Replaces, for a step of fireflies' optimisation loop (main.py:97-107: loss.backward(); optim.step(); laser.clamp_to_fov();
 * laser.normalize_rays(); and the next iteration's laser.generateTexture(sigma, size) + blur, fireflies/projection/laser.py:199-206,254-255,
 * fireflies/graphics/rasterization.py:583-607), the launches between two renders by one.
 * Every value is formed exactly as the two separate entry points form it (the same device functions); the bias corrections' beta^t are kept
 * as running products between launches (one multiply per step; pow() when the step count or the betas are not the last launch's).
 * Inside the launch the first n workgroups take the points' gradients (the last of them to arrive applies the update), the others slices of the
 * data term's inner product, then — once the update is published — a forward tile each.  A waiting workgroup only waits for workgroups with a
 * smaller index, which the hardware dispatches before it and which wait for nothing; the wait is bounded all the same: after ~0.2 s a helper
 * sets the `timeout` word and leaves (the texture is then incomplete: the caller must treat a non-zero word as a failed call).
 *   tsor, ws        : both given: the next step's soft-or texture and the regulariser's partial sums are written too (whatever reg_weight is).
 *   adam            : required, with state (exp_avg, exp_avg_sq, step); adam->counter is not used.
 *   blur_ksize      : 5 (the reference's kernel); a footprint that does not fit the workgroup's LDS: FFX_ERR_UNSUPPORTED (the caller then issues
 *                     the two launches).
 *   sync            : FFX_PATTERN_SYNC_BYTES device bytes (counters and flags on lines of their own), 8-byte aligned, zero before the first call;
 *                     the launch leaves its counters zero and its flags at `epoch`.  Word 5 (`timeout`): above.  Bytes 72..135: a
 *                     copy of the 64-byte header behind adam->guard as this step left it (the launch clears the header itself when `zero` covers
 *                     it); word 4 (`stale`): set when check_kept found the rays changed (below).
 *   epoch           : the previous call's on this sync buffer and these rays_kept + 1 (1 for the first): what the launch's flags are set to, and
 *                     its parity says which half of rays_kept is written.
 *   rays_kept       : [dev][2][n,3]: half (epoch & 1) receives the rays the new texture is made from (the forward part reads them from there);
 *                     with check_kept != 0 the launch first compares the incoming rays with the OTHER half, the previous call's — a caller that
 *                     skipped its own forward launch because the previous ffx_pattern_step had made the texture asks for the proof that nobody
 *                     edited the pattern in between; a mismatch sets sync's `stale` word (sticky).
 * ---------------------------------------------------------------------------------------- */
#define FFX_PATTERN_SYNC_BYTES 35840
int ffx_pattern_step(float *rays /*[dev][n,3] in/out*/, int n, const float *KF /*[host][16]*/, float sigma, int size0, int size1,
                     float *tsum /*[dev] in: step s, out: step s + 1*/, float *tsor /*[dev] or NULL*/, const float *gtex /*[dev][size1,size0] or NULL*/,
                     float reg_weight, float *ws /*[dev] or NULL*/, float *grays_data /*[dev][n,3] or NULL*/, float *grays_reg /*[dev][n,3] or NULL*/,
                     float *reg_value /*[dev][3]*/, const float *loss_in /*[dev][loss_in_n] or NULL*/, int loss_in_n, float loss_div, int blur_ksize,
                     float blur_sigma, const ffx_adam_args *adam /*[host]*/, float *pts /*[dev][n,2]*/, float *zero /*[dev][n_zero] or NULL*/, long n_zero,
                     float *tex /*[dev][size1,size0]*/, float *rays_kept /*[dev][2][n,3]*/, int check_kept, void *sync /*[dev]*/, uint32_t epoch,
                     ffx_stream stream);

/* ------------------------------------------------------------------------------------------
 * f1  sampler draws of a scene randomisation, on the host.
 * Replaces the `torch.rand(a.shape, device=a.device)` inside randomBetweenTensors (fireflies/utils/math.py:170-175),
 * the call behind every UniformSampler / UniformScalarToVec3Sampler draw (fireflies/sampling/uniform.py:16-19,
 * uniform_scalar_to_vec3.py) that Transformable.randomize / Mesh.randomize issue per entity
 * (fireflies/entity/base.py:220-234, entity/mesh.py:141-156), each followed in the reference by a `.tolist()` /
 * `.item()` device-to-host sync (fireflies/scene.py:258-274).  For a tensor of n <= 256 float32 elements on
 * PyTorch-ROCm's default CUDA generator with (seed = gen.initial_seed(), offset = gen.get_offset()) the call
 * writes the n values that torch.rand would have produced on the device (Philox4x32-10, element index =
 * subsequence, offset / 4 = counter; rocrand's (0, 1] float mapping with torch's 1 -> 0 fold) and the amount
 * by which the caller must advance the generator offset (gen.set_offset(offset + *offset_increment)).
 * No device work, no stream.  FFX_ERR_UNSUPPORTED for n > 256 or an offset that is not a multiple of 4
 * (the caller then draws on the device as the reference does).
 * ---------------------------------------------------------------------------------------- */
int ffx_torch_rand_h(uint64_t seed, uint64_t offset, int n, float *out /*[host][n]*/,
                     uint64_t *offset_increment /*[host]*/);
/* The k draws of a whole Scene.randomize() (fireflies/scene.py:360-371) — or of the S scene samples of an optimisation
 * step — in one call: draw i = ffx_torch_rand_h(seeds[i], offsets[i], counts[i]); values packed draw after draw.  The
 * caller advanced the generator by 4 per draw when it reserved the offsets. */
int ffx_torch_rand_batch_h(int k, const uint64_t *seeds /*[host][k]*/, const uint64_t *offsets /*[host][k]*/,
                           const int32_t *counts /*[host][k]*/, float *out /*[host][sum counts]*/);

/* The whole of Scene.randomize() on the host, for the S scene samples of a step, in ONE call (ABI 5).
 * Replaces the Python of fireflies/scene.py:344-384 (randomize_list over meshes, lights, materials, camera, projector),
 * Transformable.randomize (fireflies/entity/base.py:194-244: translation draw, rotation draw, attribute draws; world =
 * (T + centroid) @ R @ world with R = Z @ Y @ X and the reference's swapped axis names) and Mesh.randomize (fireflies/entity/mesh.py:
 * 141-165: translation, rotation, scale; (T + centroid) @ R @ S @ world), including the parent chains of Transformable.world()
 * (entity/base.py:239-244) and the un-centring translation Scene.update_meshes applies to a posed mesh (fireflies/scene.py:243-251).
 * `draws`: every sampler call of one randomisation in the reference's order — draw d is the torch.rand of draws[d].n <= 4 values at
 * generator offset offsets[s] + 4 d under seeds[s] (ffx_torch_rand_h), mapped to lo + u * (hi - lo) in float32 (multiply, then add).
 * `ents`: the entities in the same order (parents before children), each naming its draws.  Outputs per sample: the drawn values
 * (4 per draw, unused ones 0), each entity's randomised local matrix, its world matrix through the parent chain, and that matrix
 * times the translation by -centroid.  The float32 arithmetic is the reference's torch expressions' (an fma chain over k for the 3x3 /
 * 4x4 products, as torch's and numpy's sgemm kernels compute them; angles through double cos / sin): bit-identical to the Python
 * mirror, which the goldens g7 pin.  Samplers that are not plain uniform draws stay in Python (fireflies_amd/scene.py falls back). */
/* out = a @ b, row-major 4x4 float32, each element an fma chain over the inner index (the first product plain): the ONE definition of
 * a matrix product on the host side of the randomiser — the Python mirror (fireflies_amd/entity) calls it where the reference writes
 * torch.matmul (fireflies/entity/base.py:220-244), because sgemm libraries round a 4x4 product differently from CPU to CPU.
 * out may alias a or b. */
int ffx_mat4_mul_h(const float *a /*[host][16]*/, const float *b /*[host][16]*/, float *out /*[host][16]*/);
typedef struct ffx_rand_draw {
  int32_t n;          /* values of this torch.rand call: 1..4 */
  float lo[4], hi[4]; /* the sampler's bounds */
  int32_t pad[3];
} ffx_rand_draw;
typedef struct ffx_rand_entity {
  int32_t kind;    /* 0: no transform of its own (a Material: attributes only), 1: Transformable, 2: Mesh (with a scale draw) */
  int32_t parent;  /* row of the parent entity (< this row), -1: none */
  int32_t draw_t, draw_r, draw_s; /* rows of its translation / rotation / scale draws; draw_t < 0: not randomised — `world` is taken as it stands */
  int32_t pad[3];
  float world[16]; /* the entity's base world matrix (row-major); for an entity that is not randomised: its current local matrix */
  float centroid[3];
  float pad2;
} ffx_rand_entity;
int ffx_scene_randomize_h(int n_samples, const uint64_t *seeds /*[host][n_samples]*/, const uint64_t *offsets /*[host][n_samples]*/,
                          const ffx_rand_draw *draws /*[host][n_draws]*/, int n_draws, const ffx_rand_entity *ents /*[host][n_ents]*/, int n_ents,
                          float *values /*[host][n_samples, n_draws, 4]*/, float *local /*[host][n_samples, n_ents, 16]*/,
                          float *chain /*[host][n_samples, n_ents, 16]*/, float *chain_uncentred /*[host][n_samples, n_ents, 16]*/);

/* ------------------------------------------------------------------------------------------
 * K3  texture finalise: separable Gaussian blur, reflect border.
 * Replaces kornia.filters.gaussian_blur2d(tex, (k,k), (s,s)) at
 * examples/vocalfold_scene.py:61-63 and main.py:69-71 (k = 5, s = 3).  ksize odd, <= 15.
 * bwd is the exact transpose (gradient w.r.t. the input).
 * ---------------------------------------------------------------------------------------- */
int ffx_blur_fwd(const float *in /*[dev][h,w]*/, int h, int w, int ksize, float blur_sigma,
                 float *out /*[dev][h,w]*/, ffx_stream stream);
int ffx_blur_bwd(const float *gout /*[dev][h,w]*/, int h, int w, int ksize, float blur_sigma,
                 float *gin /*[dev][h,w]*/, ffx_stream stream);

/* ------------------------------------------------------------------------------------------
 * Dataset path: the post-processing steps of main.py:138-160 as single launches on device images (ABI 8).
 * The reference copies every render to the host and runs cv2 / kornia / numpy there; as torch expressions on the device the same
 * steps are ~25 small dependent launches per sample, which then set the pace of the loop (tools/datasetbench.py).
 *   ffx_rgb_to_gray     main.py:157 cv2.cvtColor(render, cv2.COLOR_RGB2GRAY):  out = (r wr + g wg) + b wb, every operation rounded to float32
 *   ffx_silhouette_fwd  fireflies/postprocessing/apply_silhouette.py:10-40: the image times a blurred filled circle (the endoscope's vignette).
 *                       disc(x, y) = 1 where (x - cx)^2 + (y - cy)^2 <= radius^2, else 0 (integer arithmetic); out = img * K3(disc) with
 *                       ffx_blur_fwd's arithmetic (ksize x ksize, reflect border) — bit for bit ffx_blur_fwd of the stored mask times the
 *                       image, without the mask ever existing.  out may be img.
 *   ffx_noise_clamp     fireflies/postprocessing/white_noise.py:5-20 with the noise drawn on the device: out = clamp(img + (noise std + mean),
 *                       lo, hi), NaN kept (torch.clamp's rule); every operation rounded to float32.  out may be img or noise.
 * ---------------------------------------------------------------------------------------- */
int ffx_rgb_to_gray(const void *img /*[dev][n_pixels,3] float32, or float16 with img_fp16*/, int img_fp16, size_t n_pixels,
                    float wr, float wg, float wb, float *out /*[dev][n_pixels]*/, ffx_stream stream);
int ffx_silhouette_fwd(const float *img /*[dev][h,w]*/, int h, int w, int cx, int cy, int radius, int ksize, float blur_sigma,
                       float *out /*[dev][h,w]*/, ffx_stream stream);
int ffx_noise_clamp(const float *img /*[dev][n]*/, const float *noise /*[dev][n]*/, size_t n, float mean, float std, float lo, float hi,
                    float *out /*[dev][n]*/, ffx_stream stream);

/* ------------------------------------------------------------------------------------------
 * K5 + K6  per-randomisation geometry update.
 * Replaces Mesh.get_randomized_vertices (fireflies/entity/mesh.py:158-165) +
 * Scene.update_meshes (fireflies/scene.py:243-251) + the acceleration-structure rebuild
 * inside mitsuba_params.update() (fireflies/scene.py:384).
 *
 * The BVH is an opaque blob in caller memory.  Topology is built once on the host from a
 * representative pose (ffx_bvh_build_host), uploaded by the caller, and re-fitted on the
 * device for every randomisation (ffx_scene_update), which also transforms the vertices:
 *   world vertex of triangle corner = xform[shape] · [src_verts[vert_off[shape] + idx]; 1]
 * so an animation frame is selected by pointing vert_off[shape] at that frame.
 * ---------------------------------------------------------------------------------------- */
typedef struct ffx_bvh_info {
  int32_t n_tris;
  int32_t n_nodes;
  int32_t n_levels;  /* refit groups (node heights), <= FFX_MAX_LEVELS */
  int32_t max_depth; /* traversal stack bound */
  uint64_t off_nodes; /* byte offsets inside the blob */
  uint64_t off_order;
  uint64_t off_refit;
  uint64_t off_recs;
  uint64_t total_bytes;
  int32_t level_start[FFX_MAX_LEVELS + 1]; /* ranges into the refit list, leaves-first */
  /* 64-wide overlay of the same tree for the wave-packet kernels: inner nodes of up to
   * 64 children, triangles grouped in clusters of up to 64 consecutive leaf slots; ONE array of 32-byte
   * elements {f32 lo[3], f32 hi[3], i32 ref, pad} (16 bytes with 16-bit boxes in a -DFFX_WIDE_F32=0 build).
   * All zero in a blob written by the oracle (which walks its own binary tree). */
  int32_t n_wide;     /* wide inner nodes (0: the whole scene is one cluster) */
  int32_t wide_depth; /* wide inner levels above the clusters */
  int32_t wide_root;  /* reference of the root: cluster << 31 | element << 6 | (count - 1), elements of 32 B from off_wnodes */
  int32_t wide_pad;
  uint64_t off_wnodes; /* n_wide x 64 x 32 B child records */
  uint64_t off_wsrc;   /* n_wide x 64 x int32: where each child's box lives in the binary tree */
  uint64_t off_tq;     /* n_tris (+ 64 of padding) x 32 B triangle boxes, leaf-slot order; = off_wnodes + n_wide * 2048 */
  uint64_t off_whdr;   /* 64 B header of the overlay (the quantisation grid of the current pose in a 16-bit build) */
  /* refit plan (ABI 4): the tree cut into TREELETS — maximal subtrees of at most ~1024 triangles, each re-fitted by ONE
   * workgroup from its triangle records up to its root (workgroup barriers only) — plus the TOP of the tree above them,
   * re-fitted by whichever workgroup finishes last; ffx_scene_update is ONE launch.  int32 tables:
   * (n_treelets + 1) headers of 8 ints {slot_first, slot_count, level_first, n_levels, wchild_first, wchild_count, 0, 0},
   * the level starts, the node list (by treelet, then by height), the wide children grouped by the treelet that owns the
   * binary node their box is copied from, and the arrival counter.  Zero in a blob written by the oracle. */
  uint64_t off_plan;
  int32_t n_treelets; /* workgroups of the update launch; the header at index n_treelets describes the top */
  int32_t plan_ints;  /* int32 words of the plan area (the arrival counter is the last one) */
  uint64_t off_nrec;  /* (n_tris + 4) x 48 B: per leaf slot the three vertex normals {n0, n1, n2} as float4, written by
                         ffx_scene_update for the shapes of ffx_smooth (below), read by the render kernels at the hit */
  uint64_t off_gn;    /* (n_tris + 4) x 16 B: per leaf slot the unit geometric normal {nx, ny, nz, bits}, written by ffx_scene_update with
                         IEEE cross / sqrt / divide in the oracle's order; the packet render kernels read it instead of re-deriving it
                         per sample.  The fourth word is NOT a float: raw bits 0 for a degenerate triangle (the normal is then
                         {0,0,0}), else (shape + 1) | smooth << 30 (smooth: the record is flagged by ffx_smooth; bits 28 / 29: scratch of the
                         render calls' pre-pass under FFX_SHADOW_CLEAR — "nothing can shadow this triangle from the projector / spot") — the kernels take
                         the hit's shape id and smooth flag from it, so a blob written to any other encoding renders with a wrong
                         material row.  0 in the oracle's blob (it has no such area). */
  /* tile bins (ABI 5, DESIGN.md 5.1 "round 4"): for each of the three ray origins of a render (camera, projector, spot) a
   * perspective grid of tiles over that origin's field of view and, per tile, the list of triangles whose projection overlaps it —
   * 64-byte entries {padded 2-D box, three edge equations, leaf slot} in the grid's tile units.  Written by the render calls' pre-pass
   * (ffx_apex_prepare / a render without FFX_RENDER_APEX_READY) next to the apex records, read by the packet render kernels: a
   * pixel's packet tests the entries of its tile against its own screen rectangle with the lanes on the entries and runs the exact
   * ray / triangle test on the survivors only — no tree walk.  Packets the bins cannot serve (outside a grid, more than four tiles,
   * an overflowed or disabled grid) take the tree walks as before; results are identical either way.
   * Layout: FFX_N_APEX areas of bins_stride bytes from off_bins: a 64-byte header {ok, entries, capacity}, (tiles + 1) uint32 list
   * starts, `tiles` uint32 fill cursors, the entries, then (round 6) the grid's envelope: 16 bytes per cell, 49 cells per tile (fireflies_amd/csrc/ffx_common.h
   * FFX_ENV_SUB; header word 4: this pose's pre-pass wrote it).  0 in the oracle's blob. */
  uint64_t off_bins;
  uint64_t bins_stride;
} ffx_bvh_info;

/* Interpolated shading normals (optional; ffx_scene_update's `smooth`).  Mitsuba shades a mesh that carries vertex normals
 * (an OBJ with `vn`, a PLY with nx/ny/nz) in the frame of the normal interpolated at the hit, and re-derives
 * angle-weighted vertex normals from the new positions after every `<mesh>.vertex_positions` update — which is what
 * Scene.update_meshes does per randomisation (fireflies/scene.py:243-251 -> params.update(), :384; OBJ frames:
 * fireflies/entity/mesh.py:167-181) [EXT mesh.cpp recompute_vertex_normals: for every face and corner
 * n_face * unit_angle(edge, edge) is added to the corner's vertex, then normalised].  Here: the update computes the
 * vertex normals of the CURRENT pose in world space (one launch, a lane per vertex over its incident corners in
 * ascending triangle order — deterministic), stores them per leaf slot in the blob (off_nrec) and flags the shape's
 * records; a render kernel that hits a flagged record interpolates  ns = normalize((1-u-v) n0 + u n1 + v n2),  faces ns
 * to the viewer (the exporter's `twosided` wrapper flips by the sign of cos(theta_i) in the SHADING frame), evaluates
 * the BSDF and the emitters' cosines with ns, and keeps the geometric normal for what is geometry: the side the shadow
 * ray's origin is lifted to and the requirement that an emitter lies on the viewer's geometric side.  A zero-length
 * interpolated normal falls back to the geometric one.  Shapes not flagged render exactly as before. */
typedef struct ffx_smooth {
  const int32_t *shape_smooth; /* [host][n_shapes] 1: interpolated normals, 0: flat */
  const int32_t *shape_vbase;  /* [host][n_shapes] row of the shape's vertex 0 in the tables below (local vertex i -> row vbase + i) */
  const int32_t *adj_start;    /* [dev][n_vn + 1] CSR: the corners incident to each vertex row (empty for flat shapes) */
  const int32_t *adj;          /* [dev][adj_start[n_vn]] triangle << 2 | corner, ascending per vertex */
  int32_t n_vn;                /* vertex rows (all shapes, one frame each) */
  float *vnormals;             /* [dev][n_vn, 3] scratch: written, then read, by the update on its stream */
} ffx_smooth;

/* upper bound of the blob size for n_tris triangles.  The blob ends with scratch areas ("apex records",
 * DESIGN.md 4.1) that ffx_trace_primary / ffx_render_* rewrite on every call on their stream: calls that
 * share a blob must be stream-ordered, and the blob they take as `const void *` is const in its
 * topology and triangle records only. */
size_t ffx_bvh_blob_bytes(int n_tris);
/* Host-side topology build (binned SAH).  tris are *global* vertex indices into verts. */
int ffx_bvh_build_host(const float *verts /*[host][n_verts,3]*/, int n_verts,
                       const int32_t *tris /*[host][n_tris,3]*/, int n_tris,
                       void *blob /*[host]*/, size_t blob_bytes, ffx_bvh_info *info /*[host] out*/);
int ffx_scene_update(void *bvh /*[dev] blob*/, const ffx_bvh_info *info /*[host]*/,
                     const float *src_verts /*[dev][*,3]*/, const int32_t *tris /*[dev][n_tris,3] shape-local*/,
                     const int32_t *tri_shape /*[dev][n_tris]*/, const int32_t *vert_off /*[dev][n_shapes]*/,
                     const float *xform /*[dev][n_shapes,16]*/, int n_shapes, const ffx_smooth *smooth /*[host] or NULL*/,
                     ffx_stream stream);

/* Same pass with the per-shape tables given as HOST arrays (n_shapes <= FFX_MAX_SHAPES_H): they
 * travel as kernel arguments, so a randomisation enqueues no host-to-device copy and never blocks
 * the host.  This is what Scene.randomize() uses; the device-pointer form above serves batched,
 * device-resident randomisers. */
#define FFX_MAX_SHAPES_H 32
int ffx_scene_update_h(void *bvh /*[dev] blob*/, const ffx_bvh_info *info /*[host]*/,
                       const float *src_verts /*[dev][*,3]*/, const int32_t *tris /*[dev][n_tris,3] shape-local*/,
                       const int32_t *tri_shape /*[dev][n_tris]*/, const int32_t *vert_off /*[host][n_shapes]*/,
                       const float *xform /*[host][n_shapes,16]*/, int n_shapes, const ffx_smooth *smooth /*[host] or NULL*/,
                       ffx_stream stream);

/* ------------------------------------------------------------------------------------------
 * K7  primary visibility.
 * Replaces sensor.sample_ray + scene.ray_intersect in fireflies/graphics/depth.py:
 *   from_camera_non_wrapped (:49-86)  -> jitter = 0
 *   from_camera             (:128-166) -> jitter = 1
 *   get_segmentation_from_camera (:89-125) -> shape_out
 *   cast_laser_id           (:33-46)  -> ffx_trace_rays
 * Sample index idx = (y*W + x)*spp + s, sample position ((x + jx)/W, (y + jy)/H) with no
 * half-pixel offset (depth.py:61-69).  t is the distance along the unit ray direction from
 * the near-plane origin; a miss writes t = 0 (depth.py:84), shape = -1, prim = -1.
 * ---------------------------------------------------------------------------------------- */
typedef struct ffx_camera {
  float to_world[16];         /* row-major, camera looks down +z of its local frame */
  float camera_to_sample[16]; /* mi.perspective_projection(...) matrix */
  float near_clip, far_clip;
  int32_t width, height;
} ffx_camera;

/* jitter: bit 0 = per-sample jitter on; | FFX_RENDER_APEX_READY (4, defined below): the blob's camera area already holds THIS camera's apex
 * records and tile bins — a render or trace call of this pose from this camera wrote them (ffx_apex_prepare's promise, camera part) — and
 * nothing is launched in front of the kernel.  Without it the call writes the camera's records and bins itself. */
int ffx_trace_primary(const void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/,
                      const ffx_camera *cam /*[host]*/, int spp, int jitter, uint32_t seed,
                      float *t_out /*[dev][H*W*spp]*/, int32_t *shape_out /*[dev] or NULL*/,
                      int32_t *prim_out /*[dev] or NULL*/, ffx_stream stream);
int ffx_trace_rays(const void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/,
                   const float *origins /*[dev][n,3]*/, const float *dirs /*[dev][n,3]*/, int n,
                   float tmax, float *t_out /*[dev][n]*/, int32_t *shape_out /*[dev] or NULL*/,
                   int32_t *prim_out /*[dev] or NULL*/, ffx_stream stream);

/* ------------------------------------------------------------------------------------------
 * K8 / K9  render and its adjoint.
 * Replaces mi.render(scene, spp=...) (examples/vocalfold_scene.py:102, main.py:156) and its
 * Dr.Jit backward w.r.t. `tex.data` (examples/vocalfold_scene.py:69).  Shading model
 * (DESIGN.md §4): direct illumination at the primary hit from two delta emitters
 * (projector with irradiance texture, spot light), Lambert or principled BSDF per shape, optional
 * shadow rays, box reconstruction filter, counter-based per-sample jitter.
 * The render is linear in the texture, so the adjoint needs no forward state: it replays
 * the same samples (same seed) and scatters d(loss)/d(img) through the bilinear weights.
 * ---------------------------------------------------------------------------------------- */
typedef struct ffx_projector {
  float to_world[16];
  float camera_to_sample[16]; /* K_PROJECTOR of examples/vocalfold_scene.py:31-38 */
  float scale;                /* Mitsuba projector `scale` */
  float color[3];             /* RGB weight applied to a 1-channel texture */
  int32_t tex_w, tex_h, tex_channels; /* tex_channels: 1 or 3 */
  int32_t enabled;
} ffx_projector;

typedef struct ffx_spot {
  float to_world[16];
  float intensity[3];
  float cutoff_deg, beam_width_deg;
  int32_t enabled;
} ffx_spot;

#define FFX_MAX_BASE_TEX 4
typedef struct ffx_scene_desc {
  ffx_camera cam;
  ffx_projector proj;
  ffx_spot spot;
  int32_t shadows;    /* bit 0 (FFX_SHADOWS_ON): trace shadow rays toward both emitters.  Bit 2: FFX_SHADOWS_CACHE_DENSE, below.  Bit 1 (FFX_SHADOWS_PLAIN, round 6) is a HINT of the caller to the pre-pass
                         (ffx_apex_prepare / ffx_scene_step_h): the renders of this pose are short (fewer than ~33 samples per pixel) — the emitters'
                         envelopes (DESIGN.md 5.1 "round 6"), which let most shadow packets skip their any-hit stage, are then not built, and the spot's
                         tile grid is coarser (1.2 instead of 2 tiles per degree of cutoff): a shorter chain for a short render to wait for.
                         Images do not depend on it. */
  int32_t n_shapes;   /* rows of shape_albedo */
  int32_t mat_stride; /* floats per row of shape_albedo: 0 or 3 = Lambert albedo only, FFX_MAT_STRIDE = material rows (below) */
  /* Texture-valued base colours (Mitsuba: `<mat>.brdf_0.base_color.data`, which the reference's dataset loop re-assigns every
   * iteration: main.py:120-153).  A material row selects one with FFX_MAT_BASE_TEX = 1 + index (0: the row's own base_color).
   * Lookup [EXT Mitsuba `bitmap` texture, defaults]: the hit's texture coordinates — slot_uv's three corners interpolated with
   * the hit's barycentrics — wrapped to [0,1) (repeat), bilinear between texel centres (u w - 0.5, v h - 0.5), rows top-down
   * (the OBJ loader's flip_tex_coords default, v -> 1 - v, is applied to slot_uv by the caller).  The sampled RGB takes the
   * place of base_color everywhere in the row's BSDF (diffuse, metallic Fresnel, tints).  The adjoint with respect to the
   * PROJECTOR texture stays available through ffx_render_bwd (re-tracing); ffx_render_fwd_cache refuses such a scene
   * (its per-pixel footprint folds ONE base colour per shape). */
  int32_t n_base_tex;
  int32_t base_tex_w[FFX_MAX_BASE_TEX], base_tex_h[FFX_MAX_BASE_TEX];
  const float *base_tex[FFX_MAX_BASE_TEX]; /* [dev][h, w, 3] */
  const float *slot_uv;                    /* [dev][n_tris + 4, 6] (u0 v0 u1 v1 u2 v2 per leaf slot: ffx_bvh_info.off_order gives slot -> triangle) */
  /* The material table as part of the call (round 3): with n_mat_h > 0, mat_h holds the n_shapes rows (n_mat_h = n_shapes x
   * stride floats <= FFX_MAX_MAT_H: 8 material rows or 42 albedos) and travels to the kernels as a KERNEL ARGUMENT; the
   * shape_albedo pointer of the render calls is then ignored (may be NULL).  The reference writes the randomised material
   * parameters one by one into Mitsuba's parameter map (fireflies/scene.py:324-342) and Mitsuba uploads them in update(); here
   * a randomisation then enqueues no host-to-device copy at all — like the per-shape transforms of ffx_scene_update_h. */
  int32_t n_mat_h;
  float mat_h[128];
  /* The film's reconstruction filter (ABI 6).  FFX_RFILTER_BOX: a sample counts for the pixel it was drawn in, weight 1 (what every
   * call below computes).  FFX_RFILTER_GAUSSIAN: Mitsuba's `gaussian` — the default of `hdrfilm`, which every scene the reference
   * loads gets because none declares an <rfilter> (examples/vocalfold_scene.py:20-22, main.py:26-29) [EXT Mitsuba 3.5
   * src/rfilters/gaussian.cpp, src/render/imageblock.cpp put(), src/films/hdrfilm.cpp develop(); not in /root/reference]:
   *   g(x) = max(0, exp(-x^2 / (2 stddev^2)) - exp(-radius^2 / (2 stddev^2))),  radius = 4 stddev,  stddev default 0.5;
   *   a sample at film position (x + jx, y + jy) adds  w = g(cx - x - jx) g(cy - y - jy)  times its radiance, and w itself, to every
   *   pixel whose centre (cx, cy) = (i + 0.5, j + 0.5) is closer than `radius` along both axes;  pixel = sum(w L) / sum(w)
   *   (0 where no weight arrived).  Samples are drawn inside the film only (no border samples).
   * Served by ffx_render_fwd_filtered / ffx_render_bwd_filtered only (stddev <= 0.5: a 5x5-pixel window); every other render call
   * answers FFX_ERR_UNSUPPORTED for rfilter != 0 rather than rendering another filter than the one asked for. */
  int32_t rfilter;
  float rfilter_stddev; /* 0: the default, 0.5 */
} ffx_scene_desc;
#define FFX_SHADOWS_ON 1
#define FFX_SHADOWS_PLAIN 2
#define FFX_SHADOWS_CACHE_DENSE 4 /* shadows bit 2 (ABI 10) — not about shadows: the filtered film's adjoint cache (ffx_render_fwd_cache_filtered) gets a block for
                                   * every pass of every pixel instead of the arena's share (a quarter beyond 2^18 blocks): ffx_render_cache_bytes_sd answers
                                   * the dense size (5.4 GB at 1024 x 1024 x 256), the cache cannot overflow.  A caller's answer to `dropped` != 0. */
#define FFX_RFILTER_BOX 0
#define FFX_RFILTER_GAUSSIAN 1
#define FFX_MAX_MAT_H 128

/* Material rows (mat_stride == FFX_MAT_STRIDE): shape_albedo is then [n_shapes, 16] floats.  Model 1 is the reflection
 * side of Mitsuba 3.5's `principled` BSDF [EXT: src/bsdfs/principled.cpp eval(), principledhelpers.h,
 * microfacet.h — Burley 2012/2015; not in /root/reference], evaluated for the two delta emitters:
 *   value = F_principled D G / (4 cos_i)                       GGX (ax, ay from roughness, anisotropic), Smith G,
 *                                                              F = (1-m)(1-tint) F_dielectric(eta) + m Schlick(base) + (1-m) tint Schlick(base/lum R0(eta))
 *         + clearcoat/4 Schlick(0.04) GTR1(lerp(.1,.001,gloss)) G_ggx(0.25) cos_o
 *         + (1-m)(1-spec_trans) base/pi cos_o lerp(f_diff + f_retro, f_ss, flatness)
 *         + (1-m) sheen Schlick_w(cos_d) lerp(1, base/lum, sheen_tint) cos_o
 * in a frame (s, t, n) with n the geometric normal faced to the viewer (two-sided, as the Blender exporter wraps
 * every material) and (s, t) = Mitsuba's coordinate_system(n) (meshes carry no uv tangents here).  The transmission
 * lobe of spec_trans is not evaluated: the reference's materials are principled BSDFs nested in `twosided` (their keys are
 * `<mat>.brdf_0.*`, main.py:99-107), which Mitsuba only accepts without a transmission component, so the plugin's lobe is
 * off for good in those scenes and an assigned spec_trans only scales the diffuse lobe — as here.  `eta` is what Mitsuba derives from `specular`:
 * eta = 2 / (1 - sqrt(0.08 specular)) - 1  (specular 0 -> eta 1: no specular lobe at all, as in Mitsuba).
 * The render stays linear in the texture and affine in base_color, so the adjoint keeps its form. */
#define FFX_MAT_STRIDE 16
#define FFX_MAT_BASE_COLOR 0      /* 3 floats; the Lambert albedo for model 0 */
#define FFX_MAT_MODEL 3           /* 0.0 = Lambert (base_color / pi), 1.0 = principled */
#define FFX_MAT_ROUGHNESS 4
#define FFX_MAT_ANISOTROPIC 5
#define FFX_MAT_METALLIC 6
#define FFX_MAT_SPEC_TRANS 7
#define FFX_MAT_ETA 8
#define FFX_MAT_SPEC_TINT 9
#define FFX_MAT_SHEEN 10
#define FFX_MAT_SHEEN_TINT 11
#define FFX_MAT_FLATNESS 12
#define FFX_MAT_CLEARCOAT 13
#define FFX_MAT_CLEARCOAT_GLOSS 14
#define FFX_MAT_BASE_TEX 15        /* 0.0: base_color is the row's; k + 1: ffx_scene_desc.base_tex[k] sampled at the hit's texture coordinates */

int ffx_render_fwd(const void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/,
                   const ffx_scene_desc *sd /*[host]*/, const float *shape_albedo /*[dev][n_shapes,3 | FFX_MAT_STRIDE]*/,
                   const float *tex /*[dev][tex_h,tex_w,tex_channels]*/, int spp, uint32_t seed,
                   int img_fp16, void *img /*[dev][H,W,3] fp32 or fp16*/, ffx_stream stream);
/* gtex is ACCUMULATED into (the caller zeroes it).  flags (ABI 7): FFX_RENDER_APEX_READY (below) or 0 — without it the call rewrites
 * the blob's apex records and tile bins in front of its kernel, like a render. */
int ffx_render_bwd(const void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/,
                   const ffx_scene_desc *sd /*[host]*/, const float *shape_albedo /*[dev][n_shapes,3 | FFX_MAT_STRIDE]*/,
                   int spp, uint32_t seed, int flags, const float *gimg /*[dev][H,W,3] fp32*/,
                   float *gtex /*[dev][tex_h,tex_w,tex_channels]*/, ffx_stream stream);

/* The re-tracing adjoint with DETERMINISTIC accumulation (ABI 7; SURVEY 5 "race detection", 7.4 "deterministic mode").  Every other adjoint
 * accumulates gtex with float atomics: the result depends on the order in which the samples' taps arrive (reassociation, ~1e-7 relative,
 * different from run to run and with the number of ranks).  This one is bitwise reproducible: pass 1 re-traces and finds the largest
 * |tap| (an integer atomicMax on the float's bits), the host derives a power-of-two scale from it (ONE 4-byte read + stream
 * synchronisation), pass 2 re-traces and adds every tap as a 64-bit fixed-point integer — integer additions commute — and a last launch
 * converts:  gtex[t] += (float)(sum[t] / scale).  Resolution 2^-(62 - b) of the largest tap, b = bits of (pixels x spp x 4 taps): 2^-36 at 512 x 512 x 64 spp (float32 carries 2^-24 of a value).  Two re-traces:
 * a cross-check of the atomic paths at any size (tests/test_hip_parity.py) and a debugging aid, not the fast path.  Box and gaussian film.
 * workspace: ffx_render_bwd_det_bytes(sd) bytes of device memory, 16-byte aligned, contents irrelevant. */
size_t ffx_render_bwd_det_bytes(const ffx_scene_desc *sd /*[host]*/);
int ffx_render_bwd_det(const void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/, const ffx_scene_desc *sd /*[host]*/,
                       const float *shape_albedo /*[dev] or NULL with sd->mat_h*/, int spp, uint32_t seed, int flags /* FFX_RENDER_APEX_READY or 0 */,
                       const float *gimg /*[dev][H,W,3] fp32*/, float *gtex /*[dev][tex_h,tex_w,tex_channels]*/, void *workspace /*[dev]*/,
                       ffx_stream stream);

/* The two passes of the deterministic accumulation on their own (ABI 9), for sums that run over SEVERAL calls — the scene samples of an optimisation
 * step, the ranks of a multi-GPU step (the reference has no such exchange; fireflies/graphics/rasterization.py:583-607 is its serial
 * loop): integer sums are the same whatever the order and the grouping, so a step's texture gradient — and with it the pattern, the Adam state,
 * the whole run — comes out BITWISE equal on 1, 2, 4 or 8 ranks.  part 1: the largest |tap| of this render into *acc (one uint32 word, the
 * float's bits, atomicMax: clear it once, call per sample, MAX-reduce the word over the ranks).  ffx_det_scale_log2(bits, taps) -> the power of
 * two for `taps` = 4 x pixels x spp x ALL samples of all ranks (INT_MIN: nothing lit, or a non-finite tap).  part 2: acc = int64 [tex_h x tex_w x
 * channels], every tap added as llrint(value x 2^scale_log2) (clear once, call per sample, SUM-reduce over the ranks as integers).
 * ffx_det_finish: gtex[t] += (float)(acc[t] x 2^-scale_log2).  workspace: the filtered film's scratch (ffx_render_filter_bytes(sd)) or NULL. */
int ffx_render_bwd_det_part(const void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/, const ffx_scene_desc *sd /*[host]*/,
                            const float *shape_albedo /*[dev] or NULL with sd->mat_h*/, int spp, uint32_t seed, int flags /* FFX_RENDER_APEX_READY or 0 */,
                            const float *gimg /*[dev][H,W,3] fp32*/, int part /* 1 | 2 */, int scale_log2 /* part 2 */, void *acc /*[dev]*/,
                            void *workspace /*[dev] or NULL*/, ffx_stream stream);
int ffx_det_scale_log2(uint32_t vmax_bits, uint64_t n_taps); /* host arithmetic only */
int ffx_det_finish(const void *acc /*[dev] int64 [n]*/, int scale_log2, size_t n, float *gtex /*[dev][n], accumulated into*/, ffx_stream stream);

/* ------------------------------------------------------------------------------------------
 * K8 + K9 with an adjoint cache (store instead of re-trace).
 * ffx_render_fwd_cache = ffx_render_fwd that additionally writes what the adjoint needs into `cache`, an opaque
 * caller-owned buffer of ffx_render_cache_bytes(...) bytes that only the library that wrote it can read.
 * libffx_hip: per pixel the FOOTPRINT of its samples in the projector texture — window origin, shape and 5x5
 * weights  W[T] = sum_s fac_s * bilinear weight_s(T)  (the render is linear in the texture, the samples of a pixel
 * land on a handful of texels of one shape) — plus an arena of 24-byte records for the few samples that do not
 * fit: 37.7 MB at 512x512x64, ~10 MB of it touched.  The oracle keeps one 16-byte record per
 * sample.  ffx_render_bwd_cached scatters  gimg . albedo (. colour) / spp * W  into gtex (accumulated; the caller
 * zeroes it).  It needs neither the BVH nor the camera: the geometry may be re-fitted between the forward and the
 * backward pass (shape_albedo must still hold the forward's values).  Limits: projector textures up to 4094^2,
 * up to 255 shapes (FFX_ERR_UNSUPPORTED otherwise: use ffx_render_bwd).
 * ---------------------------------------------------------------------------------------- */
#define FFX_RENDER_FP16 1           /* img_fp16 bit 0: fp16 film */
#define FFX_RENDER_SPARSE_ADJOINT 2 /* img_fp16 bit 1 (ffx_render_fwd_cache only): the caller needs d loss / d tex only at texels whose
                                       VALUE is not zero — what a pattern optimiser needs, whose gradient flows through the splat that
                                       produced the texture (rasterization.py:583-607): samples whose four bilinear taps are all exactly
                                       zero are then neither shadow-traced nor cached.  The image is the same; gtex at zero-valued texels
                                       is unspecified.  The oracle ignores the bit (it always computes the full gradient). */
#define FFX_RENDER_APEX_READY 4      /* img_fp16 bit 2 (ffx_render_fwd, ffx_render_fwd_cache): the blob's apex areas already hold the records
                                       of THIS call's camera / projector / spot positions — written by ffx_apex_prepare, or by an earlier
                                       render / trace call with the same positions, and neither ffx_scene_update nor a call with other
                                       positions has touched the blob since.  The call then launches no pre-pass (6 us in front of every
                                       render; a caller that re-fits on a side stream prepares there, off the critical path).  A promise
                                       the library cannot check: stale records give a wrong image.  The oracle ignores the bit. */
#define FFX_RENDER_CACHE_ZEROED 8    /* img_fp16 bit 3 (ffx_render_fwd_cache): the first 64 bytes of `cache` are zero (cleared by the caller,
                                       stream-ordered before this call — e.g. by the launch that clears its gradient buffer): the call
                                       does not reset the header itself.  The oracle ignores the bit. */
#define FFX_RENDER_CACHE_KEEP_DROPPED 16 /* img_fp16 bit 4 (ffx_render_fwd_cache, ABI 7): the call resets the arena of single-sample records but
                                       KEEPS the header's `dropped` count — a step that reuses one cache for several scene samples, one after the
                                       other, then finds at its end whether ANY of them overflowed (ffx_adam_args.guard, ffx_render_cache_status):
                                       the first sample of the step clears the word (FFX_RENDER_CACHE_ZEROED or a plain call), the others keep it.
                                       The oracle ignores the bit (its cache never drops). */
/* Writes the apex records (the triangles as seen from a fixed ray origin) of sd's camera and enabled emitters into
 * the blob's apex areas — what every packet render does in front of its kernel unless told FFX_RENDER_APEX_READY.  Only
 * sd->cam.to_world, sd->proj.{enabled,to_world} and sd->spot.{enabled,to_world} are read.  No reference counterpart (Mitsuba
 * builds its acceleration structure inside params.update(), /root/reference/fireflies/scene.py:384). */
int ffx_apex_prepare(void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/, const ffx_scene_desc *sd /*[host]*/, ffx_stream stream);

/* ------------------------------------------------------------------------------------------
 * One scene sample pushed to the device in ONE call (ABI 8): the native params.update().
 * Replaces, for a sample drawn by ffx_scene_randomize_h, what the reference does between the draws and the render
 * (fireflies/scene.py:243-342: update_meshes, update_camera, update_projector, update_lights, update_materials write the
 * randomised poses and attributes into Mitsuba's parameter map one key at a time; scene.py:384: params.update() pushes them and
 * rebuilds the acceleration structure): the randomiser's output tables go straight into the scene description of the next renders,
 * the per-shape transform table and the re-fit + pre-pass launches — no parameter map in between.
 * `ops` is the compiled form of those key writes (the caller works it out once per sampler configuration):
 *   FFX_STEP_POSE_SD    chain[src] (16 floats)                    -> sd float word `dst` .. dst + 15 (a to_world block)
 *   FFX_STEP_VALUE_SD   values[src][comp]                         -> sd float word `dst`   (spot intensity / cutoff / beam width)
 *   FFX_STEP_VALUE_MAT  values[src][comp], conv 1: eta = 2 / (1 - sqrt(0.08 v)) - 1 in double, Mitsuba's specular -> eta
 *                                                                 -> mat_rows[dst]        (row x stride + column of the material table)
 *   FFX_STEP_MESH       mode 0: chain_uncentred[src], mode 1: chain[src] -> xform[dst]     (shape dst's transform)
 * in the order given (a later op overwrites an earlier one, as a later key write would).  `frames`: per shape the animation frame
 * of this sample (vert_off[shape] = frame_base + frame x frame_stride, checked against n_frames) or -1: vert_off stays.
 * sd_out = *tmpl with the ops applied and, when tmpl->n_mat_h > 0, the whole of mat_rows copied into mat_h (the caller's table
 * stays the one source of the rows).  Then, unless geom == NULL (description and tables only — also the form the CPU tests call):
 * ffx_scene_update_h(geom..., vert_off, xform) and ffx_apex_prepare(geom->bvh, sd_out) on `stream`. */
#define FFX_STEP_POSE_SD 0
#define FFX_STEP_VALUE_SD 1
#define FFX_STEP_VALUE_MAT 2
#define FFX_STEP_MESH 3
typedef struct ffx_step_op {
  int32_t kind, src, comp, dst;
  int32_t conv, mode, pad[2];
} ffx_step_op;
typedef struct ffx_step_plan {
  const ffx_step_op *ops; /* [host][n_ops] */
  int32_t n_ops, n_shapes;
  int32_t n_draws, n_ents;           /* rows of values / chain / chain_uncentred (bounds of the ops' src) */
  const int32_t *frame_base;         /* [host][n_shapes] pool offset of each shape's frame 0 */
  const int32_t *frame_stride;       /* [host][n_shapes] vertices per frame */
  const int32_t *n_frames;           /* [host][n_shapes] */
  int32_t n_mat_floats;              /* floats of mat_rows (n_shapes x stride), 0: no material table on the host */
  int32_t pad;
} ffx_step_plan;
typedef struct ffx_step_geom {
  void *bvh;                  /* [dev] the blob this sample is re-fitted into */
  const ffx_bvh_info *info;   /* [host] */
  const float *src_verts;     /* [dev] */
  const int32_t *tris;        /* [dev] */
  const int32_t *tri_shape;   /* [dev] */
  const ffx_smooth *smooth;   /* [host] or NULL */
} ffx_step_geom;
int ffx_scene_step_h(const ffx_step_plan *plan /*[host]*/, const float *values /*[host][n_draws,4]*/, const float *chain /*[host][n_ents,16]*/,
                     const float *chain_uncentred /*[host][n_ents,16]*/, const int32_t *frames /*[host][n_shapes] or NULL*/,
                     const ffx_scene_desc *tmpl /*[host]*/, ffx_scene_desc *sd_out /*[host]*/, float *mat_rows /*[host] in/out, or NULL*/,
                     float *xform /*[host][n_shapes,16] in/out*/, int32_t *vert_off /*[host][n_shapes] in/out*/,
                     const ffx_step_geom *geom /*[host] or NULL*/, int prepare_apex /*bit 0: the pre-pass behind the re-fit; bit 1: FFX_STEP_DEFER_TOP*/,
                     ffx_stream stream);
/* FFX_STEP_DEFER_TOP (ABI 10): the re-fit stops below the top of the tree — the treelets' records, boxes and nodes, which is all the pre-pass reads — and
 * the caller owes ffx_scene_refit_top(bvh, info, stream) on a stream ordered behind this call before anything WALKS the tree of that blob (a render, a
 * trace, a re-tracing adjoint).  The chain of dependent launches a short render waits for — re-fit, re-fit of the top, count, scan, fill — is one launch
 * shorter; the top (one workgroup, ~10 us of latency) runs on the render's own stream, beside the previous render.  Same tree, bit for bit.  (Only the
 * default two-launch re-fit has a top of its own: with FFX_REFIT=fused / levels the flag changes nothing and ffx_scene_refit_top launches nothing.) */
#define FFX_STEP_DEFER_TOP 2
int ffx_scene_refit_top(void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/, ffx_stream stream);

size_t ffx_render_cache_bytes(int width, int height, int spp); /* Lambert scenes (mat_stride 0 / 3) */
/* the same for any scene: with material rows the cache holds a second footprint per pixel (the part of the BSDF
 * that does not scale with base_color): 67.1 MB at 512x512x64; with sd->rfilter != 0 the size of the filtered film's cache
 * (ffx_render_fwd_cache_filtered, below) */
size_t ffx_render_cache_bytes_sd(const ffx_scene_desc *sd /*[host]*/, int spp);
int ffx_render_fwd_cache(const void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/,
                         const ffx_scene_desc *sd /*[host]*/, const float *shape_albedo /*[dev][n_shapes,3 | FFX_MAT_STRIDE]*/,
                         const float *tex /*[dev]*/, int spp, uint32_t seed, int img_fp16,
                         void *img /*[dev][H,W,3]*/, void *cache /*[dev] ffx_render_cache_bytes*/, ffx_stream stream);
/* Optionally the same launch accumulates <gimg, img> (img: the forward's image, fp32 or fp16): for a loss that is linear in
 * the image — the coverage loss of a pattern optimiser, -mean(green) — gimg is constant and that inner product IS the loss
 * value, so a gradient step needs no separate reduction launch (the reference's loop evaluates the loss with torch
 * reductions: fireflies/graphics/rasterization.py:589-601).  dot_out is an array of ffx_render_dot_slots(W, H) partial sums
 * (8x8-pixel blocks folded onto at most 256 slots; each is ADDED to, the caller zeroes them once and sums them —
 * ffx_pattern_bwd's loss_in does): spread because thousands of float atomics on one address serialise (measured +40 us).
 * img = dot_out = NULL: off. */
size_t ffx_render_dot_slots(int width, int height);
/* Forward render and adjoint in ONE launch, for a loss whose gradient does not depend on the image (linear in it: the coverage loss
 * -mean(green) of the reference's pattern optimisation, fireflies/graphics/rasterization.py:583-607 with `loss = -img[..., 1].mean()` style
 * objectives; gimg is then a constant the caller knows before the render).  img = the render; gtex += its adjoint applied to gimg
 * (accumulated: the caller zeroes it), formed where the pixel's footprint is: no cache, nothing that can overflow, no launch behind the
 * render.  dot_out (optional): FFX_ADJOINT_DOT_SLOTS partial sums that <gimg, img> is ADDED to (their sum is the value of the loss).
 * Flags in img_fp16: FFX_RENDER_FP16, FFX_RENDER_SPARSE_ADJOINT, FFX_RENDER_APEX_READY.  Same result as ffx_render_fwd_cache +
 * ffx_render_bwd_cached up to the order of the float atomics.  FFX_ERR_UNSUPPORTED with textured base colours (use ffx_render_bwd). */
#define FFX_ADJOINT_DOT_SLOTS 4096
int ffx_render_fwd_adjoint(const void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/, const ffx_scene_desc *sd /*[host]*/,
                           const float *shape_albedo /*[dev] or NULL with sd->mat_h*/, const float *tex /*[dev]*/, int spp, uint32_t seed,
                           int img_fp16, void *img /*[dev][H,W,3]*/, const float *gimg /*[dev][H,W,3] fp32*/,
                           float *gtex /*[dev][tex_h,tex_w,tex_channels]*/, float *dot_out /*[dev][FFX_ADJOINT_DOT_SLOTS] or NULL*/,
                           ffx_stream stream);
int ffx_render_bwd_cached(const ffx_scene_desc *sd /*[host]*/, const float *shape_albedo /*[dev][n_shapes,3 | FFX_MAT_STRIDE]*/,
                          const void *cache /*[dev]*/, int spp, const float *gimg /*[dev][H,W,3] fp32*/,
                          float *gtex /*[dev][tex_h,tex_w,tex_channels]*/, const void *img /*[dev][H,W,3] or NULL*/, int img_fp16,
                          float *dot_out /*[dev][ffx_render_dot_slots] or NULL*/, ffx_stream stream);
/* K9 under the reference's own loss (ABI 10): loss = weight * torch.nn.L1Loss()(img, target) (fireflies/graphics/rasterization.py:579,596-602) and its
 * backward through the render, in the scatter launch itself.  Per pixel the launch forms gimg = sign(img - target) * weight / (3 W H) — the arithmetic of
 * ffx_l1_value_grad — instead of reading it, and adds weight / (3 W H) * sum |img - target| of every 16x16-pixel block to loss_slots[block mod
 * ffx_render_dot_slots] (the caller zeroes them and sums them, e.g. ffx_pattern_bwd_blur's loss_in): ffx_l1_value_grad (two launches, a [H,W,3]
 * gradient image written and read back) + ffx_render_bwd_cached in ONE launch.  gtex as ffx_render_bwd_cached's up to the order of the float atomics,
 * the loss value up to the order of its partial sums.  fp32 image, projector with a one-channel texture; FFX_ERR_UNSUPPORTED otherwise. */
int ffx_render_bwd_cached_l1(const ffx_scene_desc *sd /*[host]*/, const float *shape_albedo /*[dev] or NULL with sd->mat_h*/, const void *cache /*[dev]*/,
                             int spp, const float *img /*[dev][H,W,3] fp32*/, const float *target /*[dev][H,W,3] fp32*/, float weight,
                             float *gtex /*[dev][tex_h,tex_w,1]*/, float *loss_slots /*[dev][ffx_render_dot_slots]*/, ffx_stream stream);
/* The cache is lossy when its arena of single-sample records fills up (a projector texture much finer than the camera's
 * pixel footprint, grazing views: most samples then miss their pixel's 5x5 window).  Samples beyond the arena are counted
 * in `dropped` and ffx_render_bwd_cached then poisons gtex[0] with NaN instead of returning a gradient with silent holes.
 * ffx_render_cache_status reads {records used, arena capacity, dropped} of a cache written by ffx_render_fwd_cache;
 * it SYNCHRONISES `stream` (a 64-byte device-to-host read).  dropped != 0: fall back to ffx_render_bwd.
 * (The oracle's cache is one record per sample and never drops: {0, 0, 0}.) */
int ffx_render_cache_status(const void *cache /*[dev]*/, uint32_t *out3 /*[host][3]*/, ffx_stream stream);


/* ------------------------------------------------------------------------------------------
 * K8 / K9 through a reconstruction filter that spreads a sample over neighbouring pixels (ffx_scene_desc.rfilter, above).
 * Forward: the render kernel leaves, per pixel, the 25 x (r, g, b, weight) sums its own samples contribute to the pixels of its
 * 5x5 window (`scratch`: 400 bytes per pixel), and a second launch gathers each pixel's 25 incoming sums and divides by the weight.
 * Adjoint: with  G[p] = gimg[p] / weight[p],  a sample's radiance receives  sum_n w_n(sample) G[p + n]  — the filter's transpose —
 * and is scattered through its bilinear texture taps as in ffx_render_bwd (re-traced: same seed, same samples; gtex ACCUMULATED).
 * The weights depend on the jitter only, so the adjoint recomputes them (it needs no state of the forward call, only scratch).
 * scratch: ffx_render_filter_bytes(sd) bytes of device memory, contents irrelevant before and after either call.
 * ---------------------------------------------------------------------------------------- */
size_t ffx_render_filter_bytes(const ffx_scene_desc *sd /*[host]*/);
int ffx_render_fwd_filtered(const void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/, const ffx_scene_desc *sd /*[host]*/,
                            const float *shape_albedo /*[dev] or NULL with sd->mat_h*/, const float *tex /*[dev]*/, int spp, uint32_t seed,
                            int img_fp16 /* FFX_RENDER_FP16 | FFX_RENDER_APEX_READY */, void *img /*[dev][H,W,3]*/, void *scratch /*[dev]*/,
                            ffx_stream stream);
/* Filtered render and its adjoint for a loss that is LINEAR in the image (gimg known before the render), as ffx_render_fwd_adjoint is for
 * the box film: the weights first (two small launches: G = gimg / weight), then ONE render launch whose per-pixel texture footprints take
 * every sample's own gradient sum_n w_n G[pixel + n], then the gather of the image.  gtex ACCUMULATED.  1-channel projector textures,
 * no textured base colours (FFX_ERR_UNSUPPORTED otherwise: ffx_render_fwd_filtered + ffx_render_bwd_filtered serve every case).
 * Flags: FFX_RENDER_FP16, FFX_RENDER_SPARSE_ADJOINT, FFX_RENDER_APEX_READY.  Same result as that pair up to the order of the float atomics. */
int ffx_render_fwd_adjoint_filtered(const void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/, const ffx_scene_desc *sd /*[host]*/,
                                    const float *shape_albedo /*[dev] or NULL with sd->mat_h*/, const float *tex /*[dev]*/, int spp, uint32_t seed,
                                    int img_fp16, void *img /*[dev][H,W,3]*/, const float *gimg /*[dev][H,W,3] fp32*/,
                                    float *gtex /*[dev][tex_h,tex_w,1]*/, void *scratch /*[dev]*/, ffx_stream stream);
int ffx_render_bwd_filtered(const void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/, const ffx_scene_desc *sd /*[host]*/,
                            const float *shape_albedo /*[dev] or NULL with sd->mat_h*/, int spp, uint32_t seed, int flags /* FFX_RENDER_APEX_READY or 0 */,
                            const float *gimg /*[dev][H,W,3] fp32*/, float *gtex /*[dev][tex_h,tex_w,tex_channels]*/, void *scratch /*[dev]*/,
                            ffx_stream stream);
/* Store instead of re-trace for the filtered film (ABI 7): the configuration the reference actually runs — hdrfilm's default gaussian filter
 * (examples/vocalfold_scene.py:20-22, main.py:26-29) under a loss that is NOT linear in the image (torch.nn.L1Loss,
 * fireflies/graphics/rasterization.py:579,596-602), whose gradient is only known after the render.
 * ffx_render_fwd_cache_filtered = ffx_render_fwd_filtered that additionally writes into `cache` (ffx_render_cache_bytes_sd(sd, spp) bytes
 * for a filtered sd; opaque, read only by the library that wrote it) what the adjoint needs.  A sample spreads over the 25 pixels of its
 * window with weights of its own, so a pixel's samples do NOT fold into one footprint as under the box film (25 gradients x 25 texels):
 * libffx_hip keeps, for the pixels that have a lit sample only, one 16-byte record per sample {base texel + shape, the two bilinear
 * fractions, the factor} (+ 4 bytes with material rows), in an ARENA of 64-sample blocks (round 6): the first lit 64-sample pass of a pixel
 * takes a block for itself and one for every later pass, the dense array of 8-byte pixel headers says where (and which passes hold a lit
 * sample); plus the weight every pixel received (written by the gather).  The arena holds a block for every pass of every pixel up to 2^18
 * blocks — 344 MB at 512x512x64 with material rows, ~15 MB of it touched by a dot pattern's render with FFX_RENDER_SPARSE_ADJOINT; such a
 * cache cannot overflow — and a quarter of them beyond: 1.34 GB at 1024x1024x256 (rounds 4-5: 5.4 GB of address space; configs[4]'s own
 * 1 024-point pattern lights 17 % of that film).  A pixel that finds
 * the arena full keeps no records and is counted: ffx_render_cache_status reports {blocks taken from the arena's counters — 0 while it holds
 * a block for every pass of every pixel: each pixel then has its own —, capacity, dropped pixels}, the adjoint
 * poisons gtex[0] with NaN when anything was dropped (re-trace with ffx_render_bwd_filtered then), ffx_adam_args.guard skips the update.  The
 * counters must be zero when the kernel starts: the call clears them in front of its launch (with FFX_RENDER_APEX_READY: one tiny launch)
 * unless told FFX_RENDER_CACHE_ZEROED; FFX_RENDER_CACHE_KEEP_DROPPED keeps the count of a step's earlier scene samples, as for the box film.
 * ffx_render_bwd_cached_filtered: gtex += the adjoint applied to gimg — per lit pixel a wave recomputes the samples' filter weights from the
 * jitter (`seed`: the forward's), forms G = gimg / weight over the pixel's window, gives every lit sample its own gradient
 * sum_n w_n G[pixel + n] and scatters its four taps through an LDS tile shared by a 16x16-pixel block.  Needs neither the BVH nor the
 * camera pose (the geometry may be re-fitted in between; shape_albedo must hold the forward's values).  Same result as
 * ffx_render_bwd_filtered up to the order of the float atomics.  Flags of the forward: FFX_RENDER_FP16, FFX_RENDER_SPARSE_ADJOINT,
 * FFX_RENDER_APEX_READY, FFX_RENDER_CACHE_ZEROED, FFX_RENDER_CACHE_KEEP_DROPPED.  FFX_ERR_UNSUPPORTED with textured base colours, projector textures above 4094^2 or more than 255 shapes, spp > 1024. */
int ffx_render_fwd_cache_filtered(const void *bvh /*[dev]*/, const ffx_bvh_info *info /*[host]*/, const ffx_scene_desc *sd /*[host]*/,
                                  const float *shape_albedo /*[dev] or NULL with sd->mat_h*/, const float *tex /*[dev]*/, int spp, uint32_t seed,
                                  int img_fp16, void *img /*[dev][H,W,3]*/, void *cache /*[dev] ffx_render_cache_bytes_sd*/, void *scratch /*[dev] ffx_render_filter_bytes*/,
                                  ffx_stream stream);
int ffx_render_bwd_cached_filtered(const ffx_scene_desc *sd /*[host]*/, const float *shape_albedo /*[dev] or NULL with sd->mat_h*/, const void *cache /*[dev]*/,
                                   int spp, uint32_t seed, const float *gimg /*[dev][H,W,3] fp32*/, float *gtex /*[dev][tex_h,tex_w,tex_channels]*/,
                                   ffx_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* FFX_H */
