/* tgs.h -- C ABI of libtgs_hip.so: the MI355X (gfx950) Gaussian-splatting hot path of Touch-GS.
 *
 * This is the drop-in boundary of SURVEY.md section 8(b).  Nothing in the reference tree calls C
 * (the reference shells out to `ns-train depth-gaussian-splatting`, scripts/train_bunny_real.sh:52,
 * whose rasterizer lives in an un-vendored submodule, .gitmodules:7-9), so each entry point cites
 * the reference-side *operator* it stands behind: the gsplat-0.1-shaped ops that nerfstudio's
 * Splatfacto model calls and the INRIA-shaped GaussianRasterizer (App. A.1).
 * The ctypes binding a maintainer adds is shown in INTEGRATION.md and lives in
 * touch_gs_amd/_lib.py.
 *
 * Conventions
 *  - plain pointers + sizes only; all pointers are DEVICE pointers unless marked [host];
 *    the caller (PyTorch) owns every buffer; the library never allocates device memory.
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*); no hidden sync.
 *  - returns 0 (TGS_OK) or a negative TgsStatus; message via tgs_last_error() (thread-local).
 *  - fp32 / int32, contiguous row-major.  No float atomics anywhere: results are
 *    bit-reproducible run to run.
 */
#ifndef TGS_H
#define TGS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGS_VERSION 310         /* 0.3.1 -- 310 (additions only, nothing of 300 changed meaning): TgsRasterOpts gains a sixth field k7_blocks
                                   (callers that pass the struct must pass the new size), tgs_adam_sh_gathered_geom_project_next,
                                   tgs_calib_fma_stream.  300: every entry point that takes `tile_start` takes its length next to it (validated against
                                   tgs_tile_start_len: the rasterizer keeps 512 scratch ints behind the starts) and the rasterize calls declare it
                                   non-const (they write that scratch); tgs_rasterize_fwd / _bwd / _bwd_band take a per-call TgsRasterOpts.
                                   201: tile_start buffers are T+513 ints; adds tgs_set_k7_quad, tgs_set_k6_split.  200 broke the ABI of 100:
                                   tgs_rasterize_fwd / tgs_rasterize_bwd[_band] gained stop_pos; earlier (round 3, then unversioned): status is int32[4], the
                                   rasterize calls carry slot_ok, splat slots 0/1 are rect-relative */
#define TGS_BLOCK 16            /* tile edge in pixels */
#define TGS_SPLAT_FLOATS 12     /* floats per projected-splat record */
#define TGS_PARTIAL_FLOATS 12   /* floats per (tile,Gaussian) partial-gradient record */
#define TGS_GROUP 256           /* Gaussians per binning group */

typedef enum TgsStatus {
  TGS_OK = 0,
  TGS_E_ARG = -1,        /* bad argument */
  TGS_E_HIP = -2,        /* a HIP runtime call failed */
  TGS_E_CAPACITY = -3    /* intersection capacity too small (see tgs_bin_sort) */
} TgsStatus;

/* Camera + frame constants.  viewmat: row-major world->camera, OpenCV axes
 * (x right, y down, z forward). */
typedef struct TgsCamera {
  float viewmat[16];
  float fx, fy, cx, cy;
  int32_t W, H;
  float near_plane;   /* 0.01 */
  float pix_center;   /* 0.5  */
  float bg[3];
  float glob_scale;   /* 1.0  */
} TgsCamera;

/* Fused training loss evaluated inside the compositing backward:
 *   L = l1_weight * sum|C - gt_rgb|  +  sum_{gt_depth>0} depth_weight * w(U) * (D/alpha - gt_depth)^2
 *   w(U) = 1 (SIMPLE_LOSS) or 1/(uncertainty_weight*U + eps) (DEPTH_UNCERTAINTY_WEIGHTED_LOSS).
 * The caller folds the means in: l1_weight = (1-ssim_lambda)/(3*H*W), depth_weight =
 * depth_loss_mult / #valid.  Any pointer may be NULL to drop its term. */
typedef struct TgsLossSpec {
  const float* gt_rgb;      /* [H,W,3] */
  const float* gt_depth;    /* [H,W]   (0 = no supervision) */
  const float* uncertainty; /* [H,W]   (NULL => SIMPLE_LOSS) */
  float l1_weight;
  float depth_weight;
  float uncertainty_weight;
  float eps;                /* 1e-6 */
} TgsLossSpec;

/* Adam hyper-parameters for tgs_adam_step: the flat parameter buffer is
 *   means[3N] | log_scales[3N] | quats[4N] | opac_logit[N] | sh[N*K*3]
 * where every segment starts at the next multiple of 4 floats (16-byte aligned views; the total
 * is rounded up to a multiple of 4 too), with one learning rate per group, SH split into DC (k=0)
 * and the rest.  Pad elements must carry zero gradient. */
typedef struct TgsAdamSpec {
  float lr_means, lr_scales, lr_quats, lr_opac, lr_sh_dc, lr_sh_rest;
  float beta1, beta2, eps;
  float bias_corr1, bias_corr2;   /* 1-beta1^t, 1-beta2^t */
  const float* device_bias_corr;  /* NULL, or device {bias_corr1, bias_corr2, lr_means} read by the
                                     kernel at run time instead of the three host fields: lets a
                                     captured hipGraph of the step be replayed with each step's
                                     values (lr_means follows an exponential decay schedule) */
} TgsAdamSpec;

int tgs_version(void);
const char* tgs_last_error(void);

/* Box calibration (bench.py): enqueues a plain v_fma_f32 stream (16 independent accumulators per lane, 4 waves per SIMD
 * on every CU) of n_iter iterations; *n_wave_instr (host, may be NULL) = wave instructions issued.  Time it with events on
 * `stream`: wave instructions / second = what the vector pipes sustain on this box under its power governor.  `sink`:
 * one device float (never written in practice). */
int tgs_calib_fma_stream(int n_iter, float* sink, int64_t* n_wave_instr /*[host]*/, void* stream);

/* Number of binning groups / tiles for sizing the caller's buffers. */
int tgs_num_groups(int N);                 /* ceil(N/TGS_GROUP) */
int tgs_num_tiles(int W, int H);           /* ceil(W/16)*ceil(H/16) */
int tgs_num_bands(int W, int H);           /* image bands of tgs_rasterize_bwd_band */
int tgs_band_tiles(int W, int H, int band, int* tile0, int* tile1);   /* row-major tile range of a band */
int tgs_tile_order_len(int W, int H);      /* 8 * 8 * ceil(ceil(tiles / 8) / 8) (one entry per K6/K7 block) */
int tgs_tile_counter_len(int W, int H);    /* int32 entries of the tile_cursor scratch: per-XCD counter rows + sub-list starts */
int64_t tgs_tile_start_len(int W, int H);  /* int32 entries of a tile_start buffer: tiles + 1 + the rasterizer's 512 scratch ints.  Every call that
                                              takes `tile_start` takes `tile_start_len` = the entries the caller allocated and fails with TGS_E_ARG
                                              if that is less (the kernels write the scratch: a T + 1 allocation would be written out of bounds) */
/* Bytes of scratch tgs_bin_sort needs for a given intersection capacity. */
size_t tgs_sort_scratch_bytes(int64_t capacity);

/* K1  projection + 3D->2D covariance + SH colour  (stands behind gsplat `project_gaussians`
 *     + `spherical_harmonics`; INRIA preprocess, App. A.1; spec App. B.1-B.5).
 * in : means[N,3] log_scales[N,3] quats[N,4] (w,x,y,z un-normalised) opac_logit[N]
 *      sh[N,sh_stride,3] (NULL or sh_deg<0 => colours taken from `colors_in`[N,3] or zero)
 * out: splats[N,12] = {x - 16 x0, y - 16 y0, depth, opacity, conic a, b, c, r, g, b, rect, 0} where
 *      rect (uint32 bits) = x0 | y0<<8 | w<<16 | h<<24 is the Gaussian's tile rectangle: the App. B.4
 *      rect intersected with the tiles in which alpha can reach 1/255 (output preserving; 0 = none);
 *      culled Gaussians have conic 0.  Image sides are limited to 4080 px (255 tiles).
 *      Slots 0, 1 hold the screen position (App. B.2, pixel units) RELATIVE to the origin of the tile
 *      rect, evaluated in compensated fp32 arithmetic: ~1e-5 px instead of the 2.4e-4 px ulp of an
 *      absolute 4K coordinate (absolute position = slot + 16 * {x0, y0}).  Callers that build records
 *      themselves (tgs_bin_sort + tgs_rasterize_*) follow the same convention.
 *      radii[N] (may be NULL) = the App. B.3 3-sigma pixel radius, 0 if culled. */
int tgs_project_fwd(const TgsCamera* cam /*[host]*/, int N, const float* means,
                    const float* log_scales, const float* quats, const float* opac_logit,
                    const float* sh, int sh_stride, int sh_deg, const float* colors_in,
                    float* splats, int32_t* radii, void* stream);

/* Stand-alone SH evaluation (stands behind gsplat `spherical_harmonics`):
 *   colors[N,3] = sum_k Y_k(dirs / |dirs|) coeffs[N,k,:]  (no +0.5, no clamp);  backward w.r.t. the
 *   coefficients only (v_coeffs[N,sh_stride,3], rows k >= (sh_deg+1)^2 are zeroed). */
int tgs_sh_fwd(int N, int sh_deg, int sh_stride, const float* dirs, const float* coeffs,
               float* colors, void* stream);
int tgs_sh_bwd(int N, int sh_deg, int sh_stride, const float* dirs, const float* v_colors,
               float* v_coeffs, void* stream);

/* K2-K5  tile binning + per-tile depth sort  (stands behind gsplat `map_gaussian_to_intersects`,
 *     the CUB radix sort and `get_tile_bin_edges` inside `rasterize_gaussians`; spec App. B.4, B.6).
 * in : splats[N,12] (slot 11 is overwritten with the in-group intersection offset)
 * out: group_base[G]    start of each 256-Gaussian group's contiguous range in the pair index
 *                       space (G = tgs_num_groups; ranges are disjoint, their order is arbitrary)
 *      tile_start[tgs_tile_start_len] [start,end) of every tile's list; tile_start[T] = #intersections; the 512 ints behind it are
 *                       scratch of the rasterizer: a pair of words per XCD, 256 B apart, that tgs_rasterize_fwd folds the
 *                       frame's deepest walk and the sum of its walks into (zeroed here) and tgs_rasterize_bwd reads --
 *                       tgs_set_k7_quad -- and, 128 B behind each pair, a slot counter of tgs_rasterize_bwd.  Since TGS_VERSION 201 the buffer is T+513 ints
 *      sorted_gid[cap]  Gaussian ids, per tile, front to back, ties by id
 *      tile_order[L]    (may be NULL; L = tgs_tile_order_len) tile visited by block b of K6 / K7:
 *                       block b runs on XCD b % 8; XCD x owns every 8th granule of 8 consecutive
 *                       tiles (balanced for any view) and visits its tiles longest list first inside
 *                       chunks of <= 1024; entry [i * 8 + x] = i-th visit of XCD x, T = "no tile"
 *      status[4]        {#intersections, overflow flag, sufficient capacity, longest list}.  The pair index
 *                       space [0, capacity) is cut into 8 equal regions, one per XCD: every K1 workgroup
 *                       takes its 256-Gaussian group's contiguous pair range from the region of the XCD
 *                       it runs on, or from the first other region with room.  A frame can therefore
 *                       overflow slightly before #intersections reaches capacity (ranges are not split
 *                       across regions) but never when capacity >= status[2] = #intersections + 8 x the
 *                       largest group total -- whatever XCD the workgroups happen to run on.  Size
 *                       buffers from status[2] (written every frame), not from status[0].  On overflow
 *                       nothing past the scans is written, overflow = 1 and status[0] = status[2]
 *                       (caller grows to it and retries)
 *      sticky_overflow  (may be NULL) one persistent int32: set to 1 by an overflowing frame and
 *                       never cleared by the library; while it is 1 every frame starts with
 *                       status[1] = 1 (empty lists).  Together with the `skip_if_overflow` argument of
 *                       the optimizer entry points this lets a caller read the status words late,
 *                       without a per-frame host sync: after an overflow nothing touches the model
 *                       until the caller has cleared the word, grown the buffers and replayed.
 *      max_list_hint    < 0: no promise (every sort class is launched).  >= 0: the caller vouches that no tile's
 *                       list is longer than this (e.g. 1.25 x the largest status[3] its recent frames of this scene
 *                       reported): the launches for list classes beyond it -- (1024, 4096] and (4096, ..) entries,
 *                       each ~4.6 us of the stream even when it finds nothing to sort -- are not issued.  A frame
 *                       that breaks the promise is VOID like one that overflowed: the unsorted ids of the long
 *                       lists are copied through (valid indices: the compositing kernels do not fault), status[1]
 *                       = 1, the sticky word is raised, status[3] = the longest list -- raise the hint and replay.
 * tmp: tile_cursor[tgs_tile_counter_len(W,H)], scratch (tgs_sort_scratch_bytes(capacity)). */
int tgs_bin_sort(const TgsCamera* cam /*[host]*/, int N, float* splats, int32_t* group_base,
                 int32_t* tile_start, int64_t tile_start_len, int32_t* tile_cursor, int32_t* sorted_gid,
                 int32_t* tile_order, int64_t capacity, void* scratch, int32_t* status,
                 int32_t* sticky_overflow, int32_t max_list_hint, void* stream);

/* K1 + K2-K5 in one call (the fast path): the projection workgroup IS the 256-Gaussian binning
 *     group, so it also builds the group scan and counts its tile intersections -- the records are
 *     not re-read and the pair offset is stored with the record.  Arguments as in tgs_project_fwd
 *     (without colors_in) followed by those of tgs_bin_sort. */
int tgs_project_bin_sort(const TgsCamera* cam /*[host]*/, int N, const float* means,
                         const float* log_scales, const float* quats, const float* opac_logit,
                         const float* sh, int sh_stride, int sh_deg, float* splats, int32_t* radii,
                         int32_t* group_base, int32_t* tile_start, int64_t tile_start_len, int32_t* tile_cursor,
                         int32_t* sorted_gid, int32_t* tile_order, int64_t capacity, void* scratch,
                         int32_t* status, int32_t* sticky_overflow, int32_t max_list_hint, void* stream);

/* Colour prefetch (single-process training loop): the Gaussians' 3K SH coefficients are needed by K1
 *     only for the colour they give from the camera.  tgs_project_bwd_adam_next (below) evaluates
 *     that colour for the NEXT view inside the optimizer kernel, where the updated row is on chip,
 *     and this variant of tgs_project_bin_sort takes it:
 *       colors_in[N,3]  colours written by tgs_project_bwd_adam_next for THIS camera
 *       color_tag       the device word that call was given; colors_in is used iff
 *                       *color_tag == tag_expect (the optimizer kernel stores its tag_value last --
 *                       a step skipped on an overflowed frame leaves the old tag), otherwise the
 *                       colours are evaluated from `sh` exactly as tgs_project_bin_sort does.
 *     Results are bit-identical to tgs_project_bin_sort either way; K1 reads 56 B instead of
 *     44 + 12K B per Gaussian. */
int tgs_project_bin_sort_colors(const TgsCamera* cam /*[host]*/, int N, const float* means,
                                const float* log_scales, const float* quats, const float* opac_logit,
                                const float* sh, int sh_stride, int sh_deg, float* splats,
                                int32_t* radii, int32_t* group_base, int32_t* tile_start, int64_t tile_start_len,
                                int32_t* tile_cursor, int32_t* sorted_gid, int32_t* tile_order,
                                int64_t capacity, void* scratch, int32_t* status,
                                int32_t* sticky_overflow, int32_t max_list_hint, const float* colors_in,
                                const int32_t* color_tag, int32_t tag_expect, void* stream);

/* Per-call choice of the compositing kernels' forms (tgs_rasterize_fwd / _bwd / _bwd_band; NULL = every field -1).
 * A field < 0 takes the process-wide default: the value of the matching tgs_set_* call, else of the environment
 * variable (read once), else the built-in one.  A caller that needs re-entrancy passes the struct and never touches
 * the setters; the setters exist for A/B runs and tests and are process-wide by definition. */
typedef struct TgsRasterOpts {
  int32_t k6_blocks;         /* 1 / 0: forward in 4x4-block / quadrant form (bit-identical images)      [TGS_K6_BLOCKS, 1] */
  int32_t k6_split;          /* tile_is_split factor of the forward, 0 = never (tgs_set_k6_split)        [TGS_K6_SPLIT, 2]  */
  int32_t k7_front_to_back;  /* 1: backward in the front-to-back form of TGS_VERSION 100                 [TGS_K7_F2B, 0]    */
  int32_t k7_quad;           /* chain-bound factor of the backward, 0 = one wave per tile (tgs_set_k7_quad) [TGS_K7_QUAD, 8] */
  int32_t k7_quad_min_walk;  /* walks up to this many entries stay with the one-wave kernel              [TGS_K7_QUAD_MIN, 16] */
  int32_t k7_blocks;         /* 1: backward in 4x4-block form (TGS_VERSION 310; measured, not the default) [TGS_K7_BLOCKS, 0] */
} TgsRasterOpts;

/* K6  per-tile front-to-back compositing of RGB + depth in ONE pass  (stands behind gsplat
 *     `rasterize_gaussians` fwd, called twice by Splatfacto for rgb and depth; spec App. B.6).
 * out: out_rgb[H,W,3] (incl. background)  out_depth[H,W] (= sum w*depth, NOT divided by alpha)
 *      final_T[H,W]  final_idx[H,W] (list position of the last contributor, -1 if none; may be
 *      NULL -- the backward does not need it)
 * in : tile_order (may be NULL = spatial order) as produced by tgs_bin_sort: scheduling only, the
 *      results do not depend on it */
int tgs_rasterize_fwd(const TgsCamera* cam /*[host]*/, const float* splats,
                      const int32_t* sorted_gid, int32_t* tile_start, int64_t tile_start_len,
                      const int32_t* tile_order, float* out_rgb, float* out_depth, float* final_T,
                      int32_t* final_idx /*may be NULL*/, int32_t* stop_pos /*may be NULL*/,
                      uint64_t* slot_ok /*may be NULL*/, const TgsRasterOpts* opts /*[host], may be NULL*/,
                      void* stream);
/* tile_start is NOT const: the forward folds the frame's walk statistics into the scratch ints behind the starts and
 * the backward keeps a slot counter there (tgs_tile_start_len).  The starts themselves are only read.
 * The statistics (deepest walk: a maximum; sum of walks: a sum) are zeroed by the scan of tgs_*bin_sort* and ACCUMULATE
 * over the forwards that follow on the same lists: issue ONE forward with stop_pos != NULL per backward.  A forward
 * with stop_pos == NULL (render only) neither stores stop positions nor touches the statistics; pixels outside the
 * image count as walk 0.  (A doubled sum can only move the chain-bound verdict of tgs_rasterize_bwd towards "one wave
 * per tile": the two forms of K7 differ by the rounding of one four-term sum, never in which pairs they visit.) */
/* stop_pos[H,W] (optional; REQUIRED by tgs_rasterize_bwd*): per pixel the list position (relative to the
 * tile's first entry) of the Gaussian whose T' <= 1e-4 stopped the pixel (App. B.6) -- every earlier position
 * with alpha >= 1/255 contributed, nothing else did -- or 0x7fffffff if the pixel never stopped.  The
 * backward walks the lists back to front from there (since TGS_VERSION 200). */
/* slot_ok[4 * tgs_slot_ok_len(W, H, capacity)] (optional): per batch of 64 list positions of a tile, four
 * 64-bit maps (one per 8x8 quadrant) of the Gaussians that changed the quadrant's state in the forward.
 * Handed to tgs_rasterize_bwd* (same splats and lists) the backward skips the other (Gaussian, quadrant)
 * evaluations; the results are bit-identical with and without it. */
size_t tgs_slot_ok_len(int W, int H, int64_t capacity);

/* Developer switch (A/B runs, tests): k6_blocks_on 1 / 0 = K6 in 4x4-block / quadrant form,
 * k7_front_to_back 1 / 0 = K7 in the front-to-back form of TGS_VERSION 100 / back to front; -1 leaves a
 * setting as it is.  Defaults: environment TGS_K6_BLOCKS (1), TGS_K7_F2B (0), read once at first use.
 * Returns the settings in force: bit 0 = block-form K6, bit 1 = front-to-back K7.
 * PROCESS-WIDE (atomic words; they only set the defaults a NULL / -1 TgsRasterOpts falls back to): a caller that
 * runs several models in one process passes TgsRasterOpts per call instead.  Same for the two setters below. */
int tgs_set_raster_variant(int k6_blocks_on, int k7_front_to_back);

/* In a CHAIN-BOUND frame K7 gives every tile that walks more than min_walk list entries to a workgroup of FOUR waves
 * (one per 8x8 quadrant) instead of one wave: in an object-centric scene a few hundred tiles carry walks of 700 - 1600
 * entries while the whole frame would fit 100 - 200 per wave slot, and the launch lasted as long as its longest tile.
 * A walk = how far into its list a tile's pixels reach; tgs_rasterize_fwd leaves the frame's deepest walk and the sum
 * of all walks behind tile_start.  A frame is chain-bound if the deepest walk exceeds factor / 2 times the sum spread
 * evenly over K7's 4096 wave slots.  Defaults: factor 8 (object-centric scenes of 100 - 300 k Gaussians at 720p qualify:
 * K7 0.58 - 0.78 of its one-wave time; a uniform scene such as configs[2] or 1 M clustered Gaussians at 1080p do not),
 * min_walk 16 (48 until TGS_VERSION 310); environment TGS_K7_QUAD / TGS_K7_QUAD_MIN.  factor 0 = always one wave per tile; a negative argument
 * leaves that setting.  Returns factor | min_walk << 8 in effect.  Results of the two forms differ by the rounding
 * of one four-term sum per (tile, Gaussian). */
int tgs_set_k7_quad(int factor, int min_walk);

/* The LONGEST tiles of a chain-bound frame (TGS_VERSION 310; an EXPERIMENT, off by default: measured, the chain of the deepest
 * tile is 3x shorter and the step is not faster -- DESIGN.md 5.8, profiles/r6_ab_runs.txt).  Tiles that walk more than min_walk
 * entries, among the first `heads` entries of the tile_order schedule (longest lists first), are composited by
 * tgs_rasterize_bwd's scan form instead of the four-wave form: 16 waves per tile, one per 4x4 block; 16 entries that touch the
 * block in the lanes of a DPP row, transmittance and the sum behind as prefix scans.  min_walk 0 = off; negative arguments leave
 * a setting.  Environment TGS_K7_SCAN_MIN / TGS_K7_SCAN_HEADS; TGS_K7_SCAN_SIDE (default 1): the scan form's launch goes to an
 * internal stream beside the other launches of the call (fork / join by events on the caller's stream; one stream per process:
 * set 0 when several host threads call the backward concurrently).  Returns min_walk | heads << 16.  Same decisions; sums in
 * scan order (rounding only). */
int tgs_set_k7_scan(int min_walk, int heads);

/* Binning: a Gaussian whose tile rect holds more than `tiles` tiles is a LONG RUN (TGS_VERSION 310): its pairs stay outside its
 * binning group's aggregated counting box (counted with direct atomics) and its partial-gradient records are summed by the
 * whole workgroup in K8 instead of by its own thread.  Default 32 (environment TGS_LONG_RUN); the trainer lowers it to 8 for
 * object-centric models, where a few thousand table / background Gaussians hold most of the pairs (model.spatial_sort; the
 * row order `optim.balanced_order` deals exactly these Gaussians evenly over the groups).  A launch-shape parameter: lists,
 * images and gradients are the same up to the rounding of K8's sums, whose shape depends on (tiles covered, this value).
 * tiles < 1 leaves the setting; returns the value in effect (1 .. 256).  Process-wide, like the other tgs_set_* tuning calls. */
int tgs_set_long_run(int tiles);

/* The forward's counterpart for tiles with LONG lists: a tile whose list is longer than max(256, factor * I / 4096) --
 * factor (default 2 -- 4 until TGS_VERSION 310 --, environment TGS_K6_SPLIT) times the per-slot load of an even spread -- among the first 512 entries
 * of the tile_order schedule is composited by FOUR blocks of the same launch, one 8x8 quadrant each (pixels are
 * independent: nothing to exchange; same images, final_T and stop positions bit for bit).  Block-form forward with a
 * tile_order only.  factor 0 = never; negative leaves it.  Returns the factor in effect. */
int tgs_set_k6_split(int factor);
/* The shape of that rule (TGS_VERSION 310): `floor` = the shortest list it splits (default 256, environment TGS_K6_FLOOR, never
 * below 64) and `heads` = how many leading entries of the schedule get the three extra blocks (default 512, TGS_K6_HEADS).  An
 * object-centric 720p frame composites faster with factor 1, floor 128, 2048 heads (its mid-size lists are single waves on an
 * under-occupied GPU: -1.7 % of the step), uniform frames do not (-1 %): the trainer's re-sort switches between the two
 * (model.spatial_sort).  Negative arguments leave a setting; returns floor | heads << 16.  Bit-identical outputs. */
int tgs_set_k6_split_shape(int floor, int heads);

/* K7  compositing backward with the tactile depth/uncertainty loss fused in  (stands behind
 *     gsplat `rasterize_gaussians` bwd; spec App. B.7).
 * in : out_rgb, out_depth, final_T, stop_pos of the forward (required);
 *      v_rgb[H,W,3] v_depth[H,W] v_alpha[H,W] upstream grads (each may be NULL);
 *      loss (may be NULL) adds dL/d(out) of the fused loss computed from out_rgb/out_depth/final_T;
 * out: partials[#intersections,12] one record per (tile,Gaussian) pair, addressed by the pair's
 *      pre-sort index = group_base[g/256] + splat[g].slot11 + index of the tile in g's rect:
 *      {v_x, v_y, v_depth, v_opacity, v_a, v_b, v_c, v_r, v_g, v_b, 0, 0}
 *      tile_loss[T,2] (may be NULL) per-tile {sum|C-gt|*l1_weight, depth-term} of the fused loss */
int tgs_rasterize_bwd(const TgsCamera* cam /*[host]*/, const float* splats,
                      const int32_t* group_base, const int32_t* sorted_gid,
                      int32_t* tile_start, int64_t tile_start_len, const int32_t* tile_order /*may be NULL*/,
                      const float* out_rgb, const float* out_depth, const float* final_T,
                      const int32_t* stop_pos, const float* v_rgb, const float* v_depth, const float* v_alpha,
                      const TgsLossSpec* loss /*[host]*/, float* partials, float* tile_loss,
                      const uint64_t* slot_ok /*may be NULL*/, const TgsRasterOpts* opts /*[host], may be NULL*/,
                      void* stream);

/* K7 for ONE image band: the tiles [tile0, tile1) of tgs_band_tiles(W, H, band, ...) (a contiguous row-major
 *     range; tgs_num_bands(W, H) >= 4 bands cover the image).  Needs the tile_order of tgs_bin_sort.  The
 *     bands together write exactly what tgs_rasterize_bwd writes; a band only reads the v_rgb / v_depth /
 *     v_alpha rows of its own tiles, so the image gradient may be produced band by band on another stream
 *     while earlier bands composite (DepthGaussianSplattingModel: SSIM pipelined behind K7). */
int tgs_rasterize_bwd_band(const TgsCamera* cam /*[host]*/, const float* splats,
                      const int32_t* group_base, const int32_t* sorted_gid,
                      int32_t* tile_start, int64_t tile_start_len, const int32_t* tile_order /*may be NULL*/,
                      const float* out_rgb, const float* out_depth, const float* final_T,
                      const int32_t* stop_pos, const float* v_rgb, const float* v_depth, const float* v_alpha,
                      const TgsLossSpec* loss /*[host]*/, float* partials, float* tile_loss,
                      int band, const uint64_t* slot_ok /*may be NULL*/,
                      const TgsRasterOpts* opts /*[host], may be NULL*/, void* stream);

/* K8a segmented reduction of the partials to one gradient record per Gaussian
 *     out: v_splats[N,12] = {v_x, v_y, v_depth, v_opacity, v_a, v_b, v_c, v_r, v_g, v_b, 0, 0}. */
int tgs_reduce_partials(int N, const float* splats, const int32_t* group_base,
                        const TgsCamera* cam /*[host]*/, const float* partials, float* v_splats,
                        void* stream);

/* K8  projection + SH backward  (stands behind gsplat `project_gaussians` bwd +
 *     `spherical_harmonics` bwd; spec App. B.8).
 * in : either partials (+group_base) -> reduced on the fly, or v_splats[N,12] (partials NULL)
 * out: v_means[N,3] v_log_scales[N,3] v_quats[N,4] v_opac_logit[N] v_sh[N,sh_stride,3]
 *      (all overwritten; v_sh may be NULL), v_xy[N,2] (may be NULL) = screen-space mean gradient (the
 *      INRIA `means2D.grad`), used for densification statistics. */
int tgs_project_bwd(const TgsCamera* cam /*[host]*/, int N, const float* means,
                    const float* log_scales, const float* quats, const float* opac_logit,
                    const float* sh, int sh_stride, int sh_deg, const float* splats,
                    const int32_t* group_base, const float* partials, const float* v_splats,
                    float* v_means, float* v_log_scales, float* v_quats, float* v_opac_logit,
                    float* v_sh, float* v_xy,
                    const int32_t* skip_if_overflow /*status[2] of the frame, or NULL*/, void* stream);

/* K8+K9 fused (single-process training): projection/SH backward whose gradients go straight
 *     through the Adam update of the flat parameter buffer `params` (layout of TgsAdamSpec) and
 *     its moment buffers -- the 59-float gradient of a Gaussian never touches HBM.  Requires an
 *     SH tensor that stores 4 or 16 bases per Gaussian (sh_stride 4 or 16); sh_deg may be any
 *     degree the storage holds (rows above it receive a zero gradient).
 *     Not usable when gradients must first be all-reduced across ranks. */
int tgs_project_bwd_adam(const TgsCamera* cam /*[host]*/, int N, int sh_stride, int sh_deg,
                         float* params, float* exp_avg, float* exp_avg_sq,
                         const TgsAdamSpec* spec /*[host]*/, const float* splats,
                         const int32_t* group_base, const float* partials, float* v_xy,
                         const int32_t* skip_if_overflow /*status[2] of the frame, or NULL*/,
                         void* stream);

/* tgs_project_bwd_adam + the colour prefetch for the next view (see tgs_project_bin_sort_colors):
 *     after the update, colors_next[N,3] = max(sum_k Y_k(dir) c_k + 0.5, 0) of the UPDATED
 *     coefficients and means, seen from next_cam, at the same active degree; then
 *     *color_tag = tag_value.  On an overflowed frame (skip_if_overflow) neither is written. */
int tgs_project_bwd_adam_next(const TgsCamera* cam /*[host]*/, int N, int sh_stride, int sh_deg,
                              float* params, float* exp_avg, float* exp_avg_sq,
                              const TgsAdamSpec* spec /*[host]*/, const float* splats,
                              const int32_t* group_base, const float* partials, float* v_xy,
                              const int32_t* skip_if_overflow /*status[2] of the frame, or NULL*/,
                              const TgsCamera* next_cam /*[host]*/, float* colors_next,
                              int32_t* color_tag, int32_t tag_value, void* stream);

/* Front prefetch (single-process training loop): tgs_project_bwd_adam_next + the NEXT view's K1.  The fused
 *     optimizer kernel holds every updated parameter of its 256 Gaussians in registers and its workgroup IS the
 *     binning group, so it also projects them for next_cam, writes their records (splats_next[N,12], radii_next
 *     [N] or NULL), allocates the group's pair range (group_base_next) and counts its tile intersections
 *     (tile_cursor_next, ranks inside scratch_next) -- K1 of the next frame, with the arithmetic of the
 *     stand-alone kernel on the same values, hence bit-identical.  The call clears tile_cursor_next / status_next
 *     first (sticky_overflow as in tgs_bin_sort).  capacity_next / scratch_next as for tgs_bin_sort.
 *     tgs_project_bin_sort_front completes that frame (scan, fill, sort) on the SAME buffers: if *tag_word ==
 *     tag_expect the records and counts are there; otherwise (the optimizer kernel was voided by its overflow
 *     guard, which leaves the counters cleared and the tag unchanged) K1 runs now, from the SH rows. */
int tgs_project_bwd_adam_next_front(const TgsCamera* cam /*[host]*/, int N, int sh_stride, int sh_deg,
                                    float* params, float* exp_avg, float* exp_avg_sq,
                                    const TgsAdamSpec* spec /*[host]*/, const float* splats,
                                    const int32_t* group_base, const float* partials, float* v_xy,
                                    const int32_t* skip_if_overflow /*status[2] of the frame, or NULL*/,
                                    const TgsCamera* next_cam /*[host]*/, float* colors_next,
                                    int32_t* tag_word, int32_t tag_value, float* splats_next,
                                    int32_t* radii_next, int32_t* group_base_next, int32_t* tile_cursor_next,
                                    int64_t capacity_next, void* scratch_next, int32_t* status_next,
                                    int32_t* sticky_overflow,
                                    int counters_cleared /*1: tgs_project_bin_sort_front cleared them*/, void* stream);
/* Data-parallel counterpart (the optimizer is not fused with K8 there): Adam on the 11 geometry parameters of every
 *     Gaussian from the all-reduced flat gradient `grads` (layout of `params`, scaled by grad_scale) -- what
 *     tgs_adam_step does on [0, start of the SH segment) -- and, on the result, the next view's K1 with the colours
 *     from the (already stepped) SH rows, into the next frame's buffers; then *tag_word = tag_value.
 *     tgs_project_bin_sort_front completes that frame.  No-op (tag unchanged) if skip_if_overflow[1] != 0. */
int tgs_adam_geom_project_next(const TgsCamera* next_cam /*[host]*/, int N, int sh_stride, int sh_deg,
                               float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                               const TgsAdamSpec* spec /*[host]*/, float grad_scale,
                               const int32_t* skip_if_overflow, int32_t* tag_word, int32_t tag_value,
                               float* splats_next, int32_t* radii_next, int32_t* group_base_next,
                               int32_t* tile_cursor_next, int64_t capacity_next, void* scratch_next,
                               int32_t* status_next, int32_t* sticky_overflow, int counters_cleared, void* stream);
/* The whole tail of a data-parallel step in one launch (TGS_VERSION 310): tgs_adam_step_sh_gathered_rows for every row
 *     chunk of the pipelined exchange + tgs_adam_geom_project_next -- the updated SH rows stay on chip for the next
 *     camera's colours, the geometry gradients are read once.  chunk_begin[n_chunks + 1] (host; consecutive ranges from 0 to N,
 *     every begin a multiple of TGS_GROUP, n_chunks <= 8), chunk_blocks[n_chunks] (host array of DEVICE pointers to the
 *     all-gathered colour blocks [world][3 rows + 4] of each chunk).  Bit-identical to the unfused sequence.  It can
 *     only start once the geometry all-reduce has landed; callers with real links keep the chunked SH Adam, which hides
 *     under it (parallel.GradSync.fused_tail). */
int tgs_adam_sh_gathered_geom_project_next(const TgsCamera* next_cam /*[host]*/, int world, int N, int sh_stride, int sh_deg,
                                           float* params, const float* grads, int n_chunks,
                                           const int32_t* chunk_begin /*[host]*/, const float* const* chunk_blocks /*[host]*/,
                                           float* exp_avg, float* exp_avg_sq, const TgsAdamSpec* spec /*[host]*/,
                                           float grad_scale, const int32_t* skip_if_overflow, int32_t* tag_word,
                                           int32_t tag_value, float* splats_next, int32_t* radii_next,
                                           int32_t* group_base_next, int32_t* tile_cursor_next, int64_t capacity_next,
                                           void* scratch_next, int32_t* status_next, int32_t* sticky_overflow,
                                           int counters_cleared, void* stream);
int tgs_project_bin_sort_front(const TgsCamera* cam /*[host]*/, int N, const float* means,
                               const float* log_scales, const float* quats, const float* opac_logit,
                               const float* sh, int sh_stride, int sh_deg, float* splats, int32_t* radii,
                               int32_t* group_base, int32_t* tile_start, int64_t tile_start_len, int32_t* tile_cursor,
                               int32_t* sorted_gid, int32_t* tile_order, int64_t capacity, void* scratch,
                               int32_t* status, int32_t* sticky_overflow, int32_t max_list_hint,
                               const int32_t* tag_word, int32_t tag_expect,
                               const TgsCamera* next_cam /*[host] or NULL*/, int32_t* next_tile_cursor /*or NULL*/,
                               int32_t* next_status /*or NULL*/, void* stream);
/* A tag mismatch (the fused kernel was voided by its overflow guard) VOIDS the frame -- empty lists, status[1] = 1,
 * sticky raised -- instead of re-running K1 (TGS_VERSION < 300 did, in a launch of its own): the parameter pointers
 * are kept in the signature but no longer read.  next_tile_cursor != NULL: extra workgroups of this call's scan launch
 * also clear the counters and the status word of the frame AFTER this one (next_cam's size; sticky_overflow as usual)
 * -- pass counters_cleared = 1 to the optimizer call that fills them.  tgs_front_can_clear_next is always true now: */
int tgs_front_can_clear_next(int N, int W, int H);

/* Data-parallel step (one view per rank, SURVEY section 8 row e).  The SH gradient of a rank is the
 *     outer product Y_k(dir(g)) x v_color[g,:], so the ranks exchange v_color (all-gather, 3 floats
 *     per Gaussian and rank) instead of all-reducing 3*sh_stride floats per Gaussian:
 *   tgs_project_bwd_color      = tgs_project_bwd (partials path) that writes, instead of v_sh, the
 *                                block v_color[3N+4] = clamp-gated colour gradients [N,3] followed
 *                                by this camera's position [3] and one zero pad;
 *   tgs_adam_step_sh_gathered  = Adam on the SH segment of `params` with the gradient
 *                                grad_scale * sum_r Y_k(dir_r(g)) * v_color_r[g,:], rebuilt in rank
 *                                order from v_color_all[world][3N+4] (the all-gathered blocks).
 *                                The means inside `params` must still be the ones the forward
 *                                pass used (step the geometry segments with tgs_adam_step
 *                                afterwards).  Requires 3*sh_stride % 4 == 0.
 *   Sync-free intersection budget under data parallelism: with `skip_if_overflow` (the frame's
 *   status word) tgs_project_bwd_color does nothing on an overflowed frame except setting the pad of
 *   its block to 1; after the all-gather tgs_dp_agree_overflow writes status_out = {0, any pad != 0}
 *   (identical on every rank) and raises the caller's sticky word, and the optimizer entry points
 *   take status_out as their `skip_if_overflow`. */
int tgs_project_bwd_color(const TgsCamera* cam /*[host]*/, int N, const float* means,
                          const float* log_scales, const float* quats, const float* opac_logit,
                          const float* sh, int sh_stride, int sh_deg, const float* splats,
                          const int32_t* group_base, const float* partials, float* v_means,
                          float* v_log_scales, float* v_quats, float* v_opac_logit, float* v_color,
                          float* v_xy, const int32_t* skip_if_overflow /*status[2] of the frame, or NULL*/,
                          void* stream);
/* Row-range forms (one chunk of a pipelined exchange; results bit-identical to the whole-model calls):
 *   tgs_project_bwd_color_rows      rows [row_begin, row_end) of the model, row_begin a multiple of TGS_GROUP;
 *                                   all arrays are the whole model's except v_color_rows = the CHUNK's block
 *                                   [3 (row_end - row_begin) + 4] (colour gradients | camera position | pad);
 *   tgs_adam_step_sh_gathered_rows  the SH rows [row_begin, row_end) from the all-gathered chunk blocks
 *                                   v_color_rows_all[world][3 (row_end - row_begin) + 4]. */
int tgs_project_bwd_color_rows(const TgsCamera* cam /*[host]*/, int N, int row_begin, int row_end,
                               const float* means, const float* log_scales, const float* quats,
                               const float* opac_logit, const float* sh, int sh_stride, int sh_deg,
                               const float* splats, const int32_t* group_base, const float* partials,
                               float* v_means, float* v_log_scales, float* v_quats, float* v_opac_logit,
                               float* v_color_rows, float* v_xy, const int32_t* skip_if_overflow, void* stream);
int tgs_adam_step_sh_gathered_rows(int world, int N, int row_begin, int row_end, int sh_stride, int sh_deg,
                                   float* params, const float* v_color_rows_all, float* exp_avg,
                                   float* exp_avg_sq, const TgsAdamSpec* spec /*[host]*/, float grad_scale,
                                   const int32_t* skip_if_overflow, void* stream);
int tgs_dp_agree_overflow(int world, int N, const float* v_color_all, int32_t* status_out,
                          int32_t* sticky_overflow /*may be NULL*/, void* stream);
int tgs_adam_step_sh_gathered(int world, int N, int sh_stride, int sh_deg, float* params,
                              const float* v_color_all, float* exp_avg, float* exp_avg_sq,
                              const TgsAdamSpec* spec /*[host]*/, float grad_scale,
                              const int32_t* skip_if_overflow /*status_out of tgs_dp_agree_overflow, or NULL*/,
                              void* stream);

/* Stores n <= 8 host floats into device memory; the values travel as launch arguments, so the call is
 * stream ordered without any host staging buffer.  Used to refresh TgsAdamSpec.device_bias_corr
 * before a captured step graph is replayed. */
int tgs_store_small(float* dst, const float* host_vals /*[host]*/, int n, void* stream);

/* K9  fused Adam over the flat parameter buffer (torch.optim.Adam semantics, no weight decay).
 *     Updates elements [elem_begin, elem_end) of the flat buffers (multiples of 4; pass 0, -1 for
 *     everything) so that chunks can be stepped as their gradient all-reduce completes. */
int tgs_adam_step(int N, int sh_stride, float* params, const float* grads, float* exp_avg,
                  float* exp_avg_sq, const TgsAdamSpec* spec /*[host]*/, float grad_scale,
                  int64_t elem_begin, int64_t elem_end,
                  const int32_t* skip_if_overflow /*status[2] of the frame, or NULL*/, void* stream);

/* K10 SSIM (11x11 Gaussian window, sigma 1.5, zero padding) forward + gradient image; the
 *     (1-SSIM) term of the Splatfacto-style loss.
 * out: block_partials[ceil(H/16)*ceil(W/16)] per-tile sums of the SSIM map (deterministic;
 *      mean SSIM = sum / (3*H*W)),
 *      v_img[H,W,3] = weight * d(sum of SSIM map)/d(img)   (NULL => forward only)
 * tmp: scratch[9*H*W] floats (needed when v_img != NULL). */
int tgs_ssim_fwd_bwd(int W, int H, const float* img, const float* gt, float weight,
                     float* block_partials, float* v_img, float* scratch, void* stream);
/* The same for the image rows [y0, y1) only (one band of a pipelined step, see tgs_rasterize_bwd_band):
 *     v_img rows [y0, y1) are written; the SSIM map is summed over rows [count_y0, count_y1) (inside
 *     [y0, y1); the bands of an image partition its rows) into block_partials[n_partials] (first entries =
 *     workgroup sums, the rest zero); scratch rows [y0-5, y1+5) are overwritten.  Values are bit-identical to
 *     the whole-image call. */
int tgs_ssim_fwd_bwd_rows(int W, int H, const float* img, const float* gt, float weight,
                          float* block_partials, int n_partials, float* v_img, float* scratch,
                          int y0, int y1, int count_y0, int count_y1, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Peer exchange: the data-parallel gradient exchange by direct stores into IPC-mapped peer memory (all 7 xGMI
 * links of a rank at once, no collective launch) -- the alternative to RCCL that touch_gs_amd.parallel selects
 * with TGS_DP_TRANSPORT=ipc (csrc/peer.hip).  The reference has no counterpart (single GPU).
 * tgs_peer_alloc: `bytes` of zeroed device memory of the first kind in `allow_kinds` (a mask of TGS_PEER_MEM_*, tried
 *     in the order uncached, fine-grained, plain) the driver grants + its 64-byte IPC handle (send it to the other
 *     processes by any means); *kind_out (may be NULL) = the kind obtained; TGS_E_HIP if none of the allowed kinds
 *     could be had -- there is no silent downgrade.  The transport's ordering argument (csrc/peer.hip) holds for
 *     uncached and fine-grained memory only: touch_gs_amd.parallel.PeerExchange asks for exactly those.
 *     tgs_peer_open maps another process's buffer; _close / _free undo them.
 * Flags are int32 words inside such buffers; `seq` must grow from call to call (compared as seq - flag <= 0);
 * `ticket` = a zeroed device int32 private to each (call site, stream).
 * tgs_peer_push:        src[0, bytes) -> dsts[i][0, bytes) for i < n_dst, then flags[i] = seq (release, system scope)
 * tgs_peer_scatter:     src[q * slice_bytes, ...) -> dsts[q][0, ...) for q < n_dst, then flags
 * tgs_peer_reduce_push: sum over r < world of srcs[r] in rank order -> every dsts[i], then flags
 * tgs_peer_wait:        the stream waits until all n flags (local memory) have reached seq; after timeout_s
 *     seconds (<= 0: 20 s) it gives up, sets *err = 1 + index of the missing flag and, if given, *poison_a = *poison_b = 1
 *     (device int32 words, may be NULL): hand it the sticky overflow word and word [1] of the verdict the optimizer
 *     kernels are guarded by (tgs_dp_agree_overflow ORs the sticky word in) and nothing behind a timed-out wait
 *     touches the model; the host reads *err with its per-step status copy.
 * All pointer arrays are HOST arrays of device pointers; at most 8 receivers; sizes and pointers are multiples of
 * 16 bytes (tgs_peer_push also takes multiples of 4 bytes, on a slower scalar path). */
#define TGS_PEER_MEM_UNCACHED 1     /* hipDeviceMallocUncached: peers' stores land in HBM behind the owner's L2 */
#define TGS_PEER_MEM_FINEGRAINED 2  /* hipDeviceMallocFinegrained: coherent at system scope */
#define TGS_PEER_MEM_PLAIN 4        /* hipMalloc: cached; the owner may read stale lines -- not used by PeerExchange */
int tgs_peer_alloc(size_t bytes, int allow_kinds, void** dptr, unsigned char* handle64, int* kind_out /*may be NULL*/);
int tgs_peer_open(const unsigned char* handle64, void** dptr);
int tgs_peer_close(void* dptr);
int tgs_peer_free(void* dptr);
int tgs_peer_push(int n_dst, void* const* dsts, int32_t* const* flags, const void* src, size_t bytes,
                  int32_t seq, int32_t* ticket, void* stream);
int tgs_peer_scatter(int n_dst, void* const* dsts, int32_t* const* flags, const void* src,
                     size_t slice_bytes, size_t total_bytes, int32_t seq, int32_t* ticket, void* stream);
int tgs_peer_reduce_push(int world, const void* const* srcs, int n_dst, void* const* dsts,
                         int32_t* const* flags, size_t bytes, int32_t seq, int32_t* ticket, void* stream);
 /* tgs_peer_signal: flags[i] = seq (release, system scope) as a launch of its own -- the conservative way to publish
 *     after a push / scatter / reduce_push that was given flags = NULL (the kernel boundary orders the data). */
int tgs_peer_signal(int n, int32_t* const* flags, int32_t seq, void* stream);
int tgs_peer_wait(int n, const int32_t* const* flags, int32_t seq, int32_t* err, float timeout_s,
                  int32_t* poison_a /*may be NULL*/, int32_t* poison_b /*may be NULL*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TGS_H */
