/*
 * pnr.h -- C ABI of the MI355X-native Point-NeRF render hot path (libpnr_hip.so).
 *
 * Drop-in boundary for SHUzhekiNg/pointnerf2studio (citations relative to the reference repo):
 *
 *   reference interface                                              replaced by
 *   ---------------------------------------------------------------  -----------------------------
 *   query_worldcoords.cpp:33-78  woord_query_grid_point_index(...)   pnr_scene_build + pnr_query_raypos
 *     (pybind module JIT-loaded at studio_utils.py:77-82; called at  (same inputs, same three outputs in the
 *      studio_utils.py:172-188)                                       same layouts, compacted over kept rays)
 *   studio_utils.py:115-127      NeuralPoints.get_hyperparameters    host side (python), feeds pnr_grid_params_t
 *   studio_utils.py:190-207      w2pers / index_select gathers       pnr_points_pack + gather inside pnr_render
 *   studio_model.py:270-365      dists, weights, PE, 3 MLPs, K-agg   pnr_weights_pack + pnr_render (shade stage)
 *   studio_model.py:368-399      ray_dist, composite, fill_invalid   pnr_render (composite stage)
 *
 * Conventions
 *   - every function returns 0 on success, a negative pnr_status_t otherwise; the message of the
 *     last failure on the calling thread is returned by pnr_last_error();
 *   - all `d_` pointers are DEVICE pointers (HBM) owned by the caller and borrowed for the call;
 *     nothing is retained after return except inside the opaque handles;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); pnr_render and
 *     pnr_query_raypos only ENQUEUE work on it and never synchronise; pnr_scene_build,
 *     pnr_points_pack and pnr_weights_pack may synchronise the stream (they run once per
 *     point-cloud / weight version, not per ray batch);
 *   - no torch types cross this boundary; the Python side passes tensor.data_ptr() and
 *     torch.cuda.current_stream().cuda_stream.
 */
#ifndef PNR_H_
#define PNR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNR_VERSION 100          /* 0.1.0 */
#define PNR_FEAT_DIM 32          /* point_features_dim  (studio_model.py:79)            */
#define PNR_MAX_K 32             /* neighbours per sample; the reference breaks at K>8 (cu:14) */
#define PNR_MAX_D 512            /* coarse samples per ray (z_depth_dim, default 400)   */
#define PNR_POINT_ROW_FLOATS 48  /* packed point row (192 B): xyz conf | color dir pad (64 B) | emb[32] */
#define PNR_MAX_CAMS 16          /* cameras per pnr_render_views call                    */

typedef enum {
    PNR_OK = 0,
    PNR_ERR_INVALID = -1,   /* bad argument (null pointer, size out of range, ...)     */
    PNR_ERR_HIP = -2,       /* a HIP runtime call failed                               */
    PNR_ERR_STATE = -3,     /* handle not built / packed yet                           */
    PNR_ERR_WORKSPACE = -4  /* workspace too small for the requested capacity          */
} pnr_status_t;

typedef struct pnr_scene pnr_scene_t;     /* voxel structure + packed point table           */
typedef struct pnr_weights pnr_weights_t; /* MLP weights in MFMA operand order              */

/* Voxel-grid hyper-parameters: exactly what NeuralPoints.get_hyperparameters + config produce
 * (studio_utils.py:104-127) and what the reference op receives as tensors
 * (query_worldcoords.cpp:33-50). */
typedef struct {
    float ranges[6];        /* ranges_tensor: padded bbox, [0:3] is the grid origin        */
    float vox[3];           /* scaled_vsize = vsize * vscale                               */
    int32_t dims[3];        /* scaled_vdim                                                 */
    int32_t kernel_size[3]; /* neighbour-search extent: (kernel_size[0]+1)/2 layers, cu:256 */
    int32_t query_size[3];  /* occupancy dilation extent, cu:101-103                       */
    int32_t P;              /* max points kept per voxel                                   */
    int32_t max_o;          /* max occupied voxels of the reference (only flagged here)    */
    int32_t compat_drop_voxel0; /* reproduce `voxel_idx > 0` (cu:147): the voxel of the first
                                   in-grid point holds no points                           */
} pnr_grid_params_t;

/* Camera of one ray bundle (studio_utils.py:148-155). */
typedef struct {
    float campos[3];    /* ray_bundle.origins[0]                                           */
    float camrotc2w[9]; /* metadata["camrotc2w"], row-major 3x3                            */
    float near_plane, far_plane;
} pnr_camera_t;

/* Render options. */
typedef struct {
    int32_t SR;            /* max shading samples per ray                                  */
    int32_t K;             /* neighbours per sample                                        */
    int32_t D;             /* coarse samples per ray; d_tmid has D entries                 */
    float radius_limit;    /* neighbour radius (4 * vsize, studio_utils.py:110)            */
    float vsize_z;         /* config.vsize[2], the fallback segment length                 */
    int32_t eval_clamp;    /* 1: clamp rgb to [0,1] (nerfstudio RGBRenderer outside training) */
    float bg[3];           /* background colour (white in the reference)                   */
    int32_t precision;     /* PNR_PRECISION_FP32 (exact fp32 MFMA) or PNR_PRECISION_BF16X3 */
    float jitter;          /* coarse-sample jitter as a fraction of the step (the reference hard-codes 0.3,
                              studio_utils.py:166); 0 = the mid-points of the table                      */
    uint32_t seed;         /* seed of the counter-based uniforms u(seed, ray, sample) used when jitter > 0 */
    float early_stop_eps;  /* 0 (default): every selected sample with a neighbour is shaded, as the reference does.
                              > 0: early ray termination -- samples are shaded front to back in chunks (3, 3, 6, 12,
                              rest of a ray's samples) and a ray whose transmittance has fallen below eps is not
                              shaded further; the skipped samples would change the pixel by less than eps.        */
    void *d_tape;          /* NULL (default), or the TRAINING workspace (pnr_backward_workspace_bytes(cap_samples, K)
                              bytes of device memory) of the pnr_render_backward call that will follow this render: the
                              render then leaves the post-activation outputs of the four per-pair layers there as it
                              computes them -- with their LeakyReLU masks as bits and every row's density
                              pre-activation -- and the backward, given the same workspace and the same opts, does not
                              recompute them (four row GEMMs: 6 of its 21 ms at 65 536 rays).  fp32, K <= 10 or 16,
                              early_stop_eps = 0; otherwise ignored (the backward recomputes, as without it).       */
    size_t tape_bytes;     /* size of d_tape                                                                       */
} pnr_render_opts_t;

/* MLP arithmetic.  FP32: v_mfma_f32_32x32x2_f32, every product and sum in fp32.  BF16X3: every fp32 product is
 * evaluated as ah*bh + ah*bl + al*bh on bf16 hi/lo splits with fp32 accumulation (relative error ~2^-16 per
 * product; RGB stays within 2e-5 of the fp32 mode, inside the 1e-4 budget; 2.6x the fp32 mode's speed).  In both
 * modes mlp_base layer 0 is summed in two parts (per-point table + per-pair distances). */
#define PNR_PRECISION_FP32 0
#define PNR_PRECISION_BF16X3 1

/* Counters written by pnr_render / pnr_query_raypos into d_counters[PNR_NUM_COUNTERS] (int64). */
enum {
    PNR_CNT_RAYS_HIT = 0,       /* R'  rays with >= 1 coarse sample in dilated occupancy     */
    PNR_CNT_RAYS_KEPT = 1,      /* R'' rays with >= 1 neighbour                              */
    PNR_CNT_SAMPLES_SELECTED = 2, /* shading samples selected (<= SR per ray)                */
    PNR_CNT_SAMPLES_VALID = 3,  /* S   samples with >= 1 neighbour                           */
    PNR_CNT_PAIRS_VALID = 4,    /* M   valid (sample, neighbour) pairs                       */
    PNR_CNT_CANDIDATES = 5,     /* candidates distance-tested                                */
    PNR_CNT_OVERFLOW = 6,       /* != 0: cap_samples was too small, output incomplete        */
    PNR_CNT_POINTS_UNIQUE = 7,  /* U   distinct neighbour points of the call                 */
    PNR_CNT_SAMPLES_SHADED = 8, /* samples sent through the MLPs (= SAMPLES_VALID unless
                                   early_stop_eps > 0)                                       */
    PNR_CNT_RESERVED = 9,
    PNR_NUM_COUNTERS = 10
};

const char *pnr_last_error(void);
int pnr_version(void);
/* sizeof of the structs that cross this boundary, as the library was compiled: [0] pnr_grid_params_t, [1] pnr_camera_t,
 * [2] pnr_render_opts_t, [3] pnr_view_t, [4] pnr_grads_t, [5] pnr_probe_t, [6] pnr_render_taps_t, [7] offsetof(
 * pnr_render_opts_t, d_tape) -- a binding in another language checks its own declarations against these. */
int pnr_abi_sizes(int64_t out[8]);

/* ---- scene: built once per point-cloud version ------------------------------------------------ */
int pnr_scene_create(pnr_scene_t **out);
int pnr_scene_destroy(pnr_scene_t *scene);
/* Builds the voxel structure over d_xyz [N,3] (replaces claim_occ / map_coor2occ / fill_occ2pnts,
 * cu:18-162, which the reference re-runs for every ray chunk). Deterministic: per-voxel point lists
 * are in ascending point index, first P kept. */
int pnr_scene_build(pnr_scene_t *scene, const float *d_xyz, int64_t N, const pnr_grid_params_t *params,
                    void *stream);
/* The cloud changed (points pruned and / or grown: neural_points.py:341-393 of the reference, driven by
 * run/train_studio.py:676-735): rebuilds the structure over the NEW cloud d_xyz [N,3] inside the memory the scene
 * already holds (no allocation unless it outgrew its buffers).  d_old_index [N] int32: the index point i had in the
 * previous cloud, -1 for an added point.  When the grid (ranges[0:3], vox, dims) is unchanged, surviving points reuse
 * their cell code and only the added points are binned.  The result is the structure pnr_scene_build produces on
 * the same cloud, bit for bit (same order-independent steps).  The packed rows are invalidated: call pnr_points_pack. */
int pnr_scene_update(pnr_scene_t *scene, const float *d_xyz, int64_t N, const pnr_grid_params_t *params,
                     const int32_t *d_old_index, void *stream);
/* info[0]=full builds, [1]=updates, [2]=cell codes reused by the last update, [3]=bytes of build scratch kept */
int pnr_scene_update_info(const pnr_scene_t *scene, int64_t info[4]);
/* info[0]=occupied voxels, [1]=(occupied > max_o), [2]=points kept in voxel lists, [3]=bricks,
 * [4]=device bytes held, [5]=N, [6]=points inside the grid, [7]=voxel dropped by compat (-1 none) */
int pnr_scene_info(const pnr_scene_t *scene, int64_t info[8]);
/* Packs the per-point tensors (studio_utils.py:84-90 layouts: xyz [N,3], embedding [N,32], conf [N],
 * dir [N,3], color [N,3]) into 192-byte rows so one neighbour costs one contiguous gather. */
int pnr_points_pack(pnr_scene_t *scene, const float *d_xyz, const float *d_embedding, const float *d_conf,
                    const float *d_dir, const float *d_color, int64_t N, void *stream);

/* Re-packs ONLY the listed rows: what a training loop needs after an optimiser step, which changes the features of
 * the U points the batch's rays touched (about 60 k of 6 M at 4096 rays), not the whole cloud -- pnr_points_pack over
 * 6 M points moves 2.3 GB per step.  d_index [n_index] int32 point indices (any order; duplicates allowed; entries
 * outside [0, N) are skipped).  d_n_index (may be null): a DEVICE int64 -- only the first min(*d_n_index, n_index)
 * entries are used, so a list whose length only the device knows (pnr_render_touched) needs no host read.  The scene
 * must hold the packed rows of a cloud of the same N (pnr_points_pack once).  Replaces nothing in the reference: its
 * gathers read the parameter tensors directly (studio_utils.py:199-207); this keeps the packed copy equal to them. */
int pnr_points_pack_rows(pnr_scene_t *scene, const float *d_xyz, const float *d_embedding, const float *d_conf,
                         const float *d_dir, const float *d_color, int64_t N, const int32_t *d_index, int64_t n_index,
                         const int64_t *d_n_index, void *stream);

/* Binds the caller's LIVE point tensors to the scene (same layouts as pnr_points_pack; the pointers are retained until
 * re-bound or unbound with all five null, and must stay valid): every following pnr_render* call re-packs, between its
 * neighbour search and its shading stage, the rows of the distinct neighbour points it found -- the rows it is about to
 * read, U x 340 bytes -- from those tensors.  A training loop then never runs an O(N) re-pack: whatever the optimiser did
 * to the tensors, a render reads their current values, and pnr_render_backward (which follows a render) too.  Rows no
 * render has touched since the last full pnr_points_pack are stale; nothing reads them.  The scene must hold packed rows
 * of the same N.  Renders that share a bound scene must not overlap on different streams. */
int pnr_points_bind(pnr_scene_t *scene, const float *d_xyz, const float *d_embedding, const float *d_conf,
                    const float *d_dir, const float *d_color, int64_t N);

/* ---- MLP weights: packed once per weight version ------------------------------------------------ */
int pnr_weights_create(pnr_weights_t **out);
int pnr_weights_destroy(pnr_weights_t *w);
/* d_w[i] / d_b[i], i = 0..8, PyTorch nn.Linear layouts ([out,in] row-major, [out]):
 *   0 mlp_base.0 [256,284]  1 mlp_base.1 [256,256]  2 mlp_head.0 [256,263]  3 mlp_head.1 [256,256]
 *   4 density    [1,256]    5 mlp_color.0 [128,280] 6 mlp_color.1 [128,128] 7 mlp_color.2 [128,128]
 *   8 rgb        [3,128]                                    (modules: studio_model.py:193-221)
 * d_Rw2c: points_Rw2c [3,3] (studio_utils.py:90). */
int pnr_weights_pack(pnr_weights_t *w, const float *const d_w[9], const float *const d_b[9],
                     const float *d_Rw2c, void *stream);

/* The nine Linear layers changed (an optimiser step), points_Rw2c did not: re-packs the forms the given arithmetic mode
 * reads (precision = PNR_PRECISION_FP32 / PNR_PRECISION_BF16X3, or -1 for both) in ONE kernel launch, without the host
 * copy of Rw2c and without synchronising -- pnr_weights_pack synchronises the stream and takes about thirty launches,
 * which a 3-ms training step would pay every step.  Needs one pnr_weights_pack before (layout, padding, Rw2c). */
int pnr_weights_update(pnr_weights_t *w, const float *const d_w[9], const float *const d_b[9], int32_t precision,
                       void *stream);

/* ---- the drop-in op ----------------------------------------------------------------------------- */
size_t pnr_query_workspace_bytes(int64_t R, int32_t D, int32_t SR, int32_t K);
/* woord_query_grid_point_index with explicit ray positions d_raypos [R,D,3].
 * Outputs (caller-allocated, worst case): d_sample_pidx [R,SR,K] int32, d_sample_loc [R,SR,3] f32,
 * d_ray_mask [R] int8.  The first R'' = d_counters[PNR_CNT_RAYS_KEPT] rows of pidx/loc are valid
 * (compacted over kept rays in ray order, -1 / 0 in unfilled slots), exactly the tensors the
 * reference returns (cu:425-432) once sliced to R''. */
int pnr_query_raypos(const pnr_scene_t *scene, const float *d_raypos, int64_t R, int32_t D, int32_t SR,
                     int32_t K, float radius_limit, int32_t *d_sample_pidx, float *d_sample_loc,
                     int8_t *d_ray_mask, int64_t *d_counters, void *d_workspace, size_t workspace_bytes,
                     void *stream);

/* ---- fused render: NeuralPoints.forward + PointNerf.get_outputs for one ray bundle -------------- */
/* pnr_render_workspace_bytes: the scene-independent part of the render workspace (what pnr_render_taps addresses).
 * pnr_render_workspace_bytes_for: the workspace pnr_render / pnr_render_views need on the given (built) scene: the
 * part above plus the per-call table of the factorised first layer -- one 1-KiB row per distinct neighbour point (at
 * most min(points in voxel lists, cap_samples * K) rows) -- and two int32 per scene point.  Returns 0 (and sets
 * the error string) on invalid arguments. */
size_t pnr_render_workspace_bytes(int64_t R, int64_t cap_samples, int32_t K);
size_t pnr_render_workspace_bytes_for(const pnr_scene_t *scene, const pnr_render_opts_t *opts, int64_t R,
                                      int64_t cap_samples);
/* d_dirs [R,3] ray directions, d_tmid [2,D]: row 0 = coarse-sample mid-point ray parameters at jitter 0, row 1 =
 * segment lengths tvals[j+1] - tvals[j] (diff_ray_marching.py:307-323 evaluated on the host).  With
 * opts->jitter > 0 every ray draws u_j = pnr_jitter_uniform(seed, ray, j) and follows the reference's arithmetic:
 * seg_j * (1 + jitter * (u_j - 0.5)), running sum, + near, mid-points (diff_ray_marching.py:312-323).
 * Outputs: d_rgb [R,3] (coarse_raycolor, background-filled), d_depth [R], d_acc [R], d_ray_mask [R] int8, d_counters [PNR_NUM_COUNTERS] int64.  cap_samples bounds the number
 * of selected shading samples held in the workspace; if exceeded PNR_CNT_OVERFLOW is set. */
int pnr_render(const pnr_scene_t *scene, const pnr_weights_t *weights, const float *d_dirs, int64_t R,
               const pnr_camera_t *cam, const float *d_tmid, const pnr_render_opts_t *opts,
               float *d_rgb, float *d_depth, float *d_acc, int8_t *d_ray_mask, int64_t *d_counters,
               void *d_workspace, size_t workspace_bytes, int64_t cap_samples, void *stream);

/* pnr_render for a bundle whose camera POSE lives on the device: d_campos [3] and d_camrotc2w [9] -- ray_bundle.origins[0]
 * and metadata["camrotc2w"] where the datamanager left them (studio_datamanager.py:79) -- are read by the kernels; the
 * planes stay host values (they come from a collider's attributes or the datamanager's config, not from the device).
 * The reference reads four scalars / vectors back per call (studio_utils.py:148-155); with this entry a bundle is
 * rendered without any device-to-host read.  pnr_render_backward after it: pass cams = NULL (the camera the render left
 * in its workspace is used). */
int pnr_render_pose(const pnr_scene_t *scene, const pnr_weights_t *weights, const float *d_dirs, int64_t R,
                    const float *d_campos, const float *d_camrotc2w, float near_plane, float far_plane,
                    const float *d_tmid, const pnr_render_opts_t *opts, float *d_rgb, float *d_depth, float *d_acc,
                    int8_t *d_ray_mask, int64_t *d_counters, void *d_workspace, size_t workspace_bytes,
                    int64_t cap_samples, void *stream);

/* The counter-based uniform in [0,1) (24 random bits) the kernels draw for coarse sample `sample` of ray `ray` when
 * jitter > 0; host-callable so a caller can reproduce a frame.  `ray` is the index of the ray inside the call for
 * pnr_render / pnr_render_views / pnr_query_raypos, and view * H * W + pixel id for pnr_render_camera(_lists): a frame
 * rendered from cameras draws the same uniforms for a pixel however it is cut into calls, tile shards or ranks. */
float pnr_jitter_uniform(uint32_t seed, uint32_t ray, uint32_t sample);

/* Several ray bundles (cameras) in ONE call: ray r belongs to camera d_ray_cam[r] or, when d_ray_cam is NULL,
 * to camera r / rays_per_cam (bundles concatenated back to back).  `cams` is a HOST array of n_cams cameras
 * (<= PNR_MAX_CAMS), d_tmid holds n_cams tables of D coarse-sample parameters.  Lifts the reference's
 * one-camera-per-bundle assumption (studio_utils.py:152) and lets a multi-view step pay the per-call overheads
 * once.  Same outputs as pnr_render, in ray order. */
int pnr_render_views(const pnr_scene_t *scene, const pnr_weights_t *weights, const float *d_dirs, int64_t R,
                     const pnr_camera_t *cams, int32_t n_cams, const int32_t *d_ray_cam, int64_t rays_per_cam,
                     const float *d_tmid, const pnr_render_opts_t *opts, float *d_rgb, float *d_depth, float *d_acc,
                     int8_t *d_ray_mask, int64_t *d_counters, void *d_workspace, size_t workspace_bytes,
                     int64_t cap_samples, void *stream);

/* ---- rays from cameras: pose + intrinsics in, no direction tensor ----------------------------------------
 * Replaces the ray generator in front of the model (studio_datamanager.py:62-110 -> nerfstudio RayGenerator /
 * Cameras.generate_rays, which materialises origins [R,3] + directions [R,3] in HBM for every bundle, plus the
 * per-ray copy of the camera rotation at studio_datamanager.py:108).  A view is a pinhole camera as nerfstudio's
 * Cameras holds it; the ray of pixel (x, y) is, in fp32 and in exactly this order (no fused multiply-add):
 *     cxn = ((float)x + 0.5f - cx) / fx;   cyn = -(((float)y + 0.5f - cy) / fy);   czn = -1
 *     w_i = R[i][0] * cxn + R[i][1] * cyn + R[i][2] * czn          (i = 0..2, left to right)
 *     dir = w / sqrtf(w_0 * w_0 + w_1 * w_1 + w_2 * w_2)            (unit length, as nerfstudio's bundles)
 * with origin campos.  pnr_pinhole_ray is the host statement of the same arithmetic (bit-identical to what the
 * kernels compute), so a caller or a test can reproduce any ray without the device. */
typedef struct {
    float campos[3];      /* camera_to_worlds[:3, 3]                                             */
    float camrotc2w[9];   /* camera_to_worlds[:3, :3], row-major (= metadata["camrotc2w"])       */
    float near_plane, far_plane;
    float fx, fy, cx, cy; /* intrinsics in pixels                                                */
} pnr_view_t;

void pnr_pinhole_ray(const pnr_view_t *view, int32_t x, int32_t y, float dir[3]);

/* The fused render of n_views views of an H x W frame with the rays generated INSIDE the kernels: ray r of the call
 * is pixel d_pixels[r % n_pixels] (flat row-major id y * W + x; d_pixels NULL = pixel r % n_pixels, n_pixels = H * W)
 * of view r / n_pixels, so the call renders R = n_views * n_pixels rays and the outputs have R rows, view-major in
 * d_pixels order.  d_pixels is what a tile shard owns (pointnerf2studio_amd.distributed.make_shard).  The sample
 * selection probes every ray from its pixel id alone; only the rays that hit the occupancy (about one in five on the
 * metric's configuration) get their direction written to a compact scratch row inside the workspace for the shading
 * stage.  d_tmid holds n_views tables [2, D] as for pnr_render_views.  Everything else as pnr_render_views. */
int pnr_render_camera(const pnr_scene_t *scene, const pnr_weights_t *weights, const pnr_view_t *views, int32_t n_views,
                      int32_t H, int32_t W, const int32_t *d_pixels, int64_t n_pixels, const float *d_tmid,
                      const pnr_render_opts_t *opts, float *d_rgb, float *d_depth, float *d_acc, int8_t *d_ray_mask,
                      int64_t *d_counters, void *d_workspace, size_t workspace_bytes, int64_t cap_samples, void *stream);

/* The same with ONE PIXEL LIST PER VIEW: d_pixels [n_views, n_pixels], ray r of the call is pixel d_pixels[r] of view
 * r / n_pixels.  What a multi-GPU step uses to give every view a different tile owner (the views of a step look at
 * the same object: with one list for all of them a rank's load imbalance repeats in every view instead of averaging
 * out -- max / mean pairs per rank 1.05 -> 1.004 at 8 ranks, pointnerf2studio_amd.distributed.make_shard(rotate=True)). */
int pnr_render_camera_lists(const pnr_scene_t *scene, const pnr_weights_t *weights, const pnr_view_t *views,
                            int32_t n_views, int32_t H, int32_t W, const int32_t *d_pixels, int64_t n_pixels,
                            const float *d_tmid, const pnr_render_opts_t *opts, float *d_rgb, float *d_depth, float *d_acc,
                            int8_t *d_ray_mask, int64_t *d_counters, void *d_workspace, size_t workspace_bytes,
                            int64_t cap_samples, void *stream);

/* The same rays written out as a direction tensor d_dirs [n_views * n_pixels, 3] (for callers that need one: the
 * training step's pnr_render_backward, tests). */
int pnr_camera_rays(const pnr_view_t *views, int32_t n_views, int32_t H, int32_t W, const int32_t *d_pixels,
                    int64_t n_pixels, float *d_dirs, void *stream);

/* Debug/test taps into the last pnr_render workspace (device pointers, valid until the workspace is
 * reused): per selected sample s: loc+t float4, ray id, neighbour list [K], decoded (sigma,r,g,b). */
typedef struct {
    const float *smp_loc;      /* [S_sel,4] x y z t */
    const int32_t *smp_ray;    /* [S_sel]           */
    const int32_t *smp_pidx;   /* [S_sel,K]         */
    const float *smp_out;      /* [S_sel,4] sigma r g b (zero where no neighbour) */
    const int32_t *ray_cnt;    /* [R] selected samples per ray */
    const int32_t *ray_off;    /* [R] first sample of ray      */
    const float *ray_dirs;     /* [R,3] after pnr_render_camera: directions of the rays with samples (other rows are
                                  not written); what pnr_render_backward takes as d_dirs after such a render */
} pnr_render_taps_t;
int pnr_render_taps(void *d_workspace, size_t workspace_bytes, int64_t R, int64_t cap_samples, int32_t K,
                    pnr_render_taps_t *taps);

/* The distinct neighbour points of the LAST render in this workspace (ascending point index): the rows of the point
 * tensors a backward of that render can touch, hence the rows to re-pack after the optimiser step
 * (pnr_points_pack_rows) and the rows of a sparse gradient exchange.  Writes min(U, index_cap) indices to d_index
 * [index_cap] (entries beyond U repeat the first entry, 0 when the list is empty: harmless duplicates for a consumer of
 * fixed length) and U itself to d_count (DEVICE int64, may be null).  No host synchronisation. */
int pnr_render_touched(const pnr_scene_t *scene, const pnr_render_opts_t *opts, int64_t R, void *d_render_workspace,
                       size_t render_workspace_bytes, int64_t cap_samples, int32_t *d_index, int64_t index_cap,
                       int64_t *d_count, void *stream);

/* ---- training step: gradients of a render ----------------------------------------------------------
 * Replaces what torch autograd derives when `ns-train pointnerf-original` back-propagates
 * get_loss_dict (studio_model.py:415-431) through get_outputs (studio_model.py:263-399) and
 * NeuralPoints.forward's index_select gathers (studio_utils.py:199-207): d loss / d {points_embeding,
 * points_color, points_dir, the nine Linear weights and biases}.  points_xyz and points_Rw2c are frozen in the
 * reference (studio_utils.py:84-90) and points_conf does not enter the render (studio_model.py:285-292: it feeds
 * the loss directly), so none of them gets a gradient here.
 * Every pointer may be null (that gradient is skipped); gradients are ACCUMULATED (+=) as torch does with .grad:
 * zero the buffers first for plain gradients.  Shapes are those of pnr_points_pack / pnr_weights_pack inputs.
 * The point gradients are summed per point in a FIXED order (rows grouped by point, ascending row index: a segmented
 * sum, no float atomics): two calls on the same inputs return the same bits. */
typedef struct {
    float *d_embedding; /* [N,32] */
    float *d_color;     /* [N,3]  */
    float *d_dir;       /* [N,3]  */
    float *d_w[9];      /* nn.Linear weights, [out,in] row-major, order of pnr_weights_pack */
    float *d_b[9];      /* [out] */
    /* Sparse emission of the point gradients (instead of the three dense tensors above, which are then ignored): row u
     * of d_point_grads [point_cap, 40] = [d embedding (32) | d color (3) | d dir (3) | 0 0] of the u-th distinct
     * neighbour point of the render (ascending point index, U = d_counters[PNR_CNT_POINTS_UNIQUE] rows; rows are
     * WRITTEN, not accumulated) and d_point_index [point_cap] its point index.  A 4096-ray batch touches ~60 k of 6 M
     * points: 10 MB instead of a zero-filled 768-MB tensor. */
    float *d_point_grads;
    int32_t *d_point_index;
    int64_t point_cap;
} pnr_grads_t;

/* Zeroes the rows d_index[0 .. n) of dense point-gradient tensors (d_embedding [N,32], d_color [N,3], d_dir [N,3]; any
 * may be null), n = n_index or min(*d_n_index, n_index) with d_n_index a DEVICE int64 (pnr_render_touched).  A training
 * loop that lets pnr_render_backward accumulate into persistent dense buffers resets them with this in O(U) instead of
 * zero-filling 768 MB per step (torch's autograd would allocate and fill a dense gradient for the index_select of
 * studio_utils.py:199-207). */
int pnr_point_grads_clear(float *d_embedding, float *d_color, float *d_dir, int64_t N, const int32_t *d_index,
                          int64_t n_index, const int64_t *d_n_index, void *stream);

/* ~10 KB per (sample, neighbour) row of capacity: the activation tapes, the gradients at the four pre-activations of the
 * per-pair MLPs, the rows' point gradients, the mask bits (+ 250 MB of partial weight-gradient tiles). */
size_t pnr_backward_workspace_bytes(int64_t cap_samples, int32_t K);
/* Call after pnr_render / pnr_render_views with the SAME scene, rays, cameras, options, cap_samples and render
 * workspace (its sample lists and neighbour indices are reused; nothing else may have used that workspace in
 * between).  The MLP forward is recomputed (or, after a render given this workspace as opts->d_tape, taken from the tape
 * that render wrote) with a tape of activations in the arithmetic of opts->precision: FP32 =
 * every product in fp32 (gradients agree with fp32 autograd to ~1e-6 relative); BF16X3 = every GEMM on bf16 hi/lo
 * splits (3 products, 2^-16 relative) -- the gradient of the function that mode renders.  d_w / d_b are the
 * raw weights (as given to pnr_weights_pack; `weights` only supplies Rw2c), d_grad_rgb [R,3] is d loss / d rgb.
 * opts->early_stop_eps must be 0.  d_rgb_recomputed (may be null) receives the fp32 rgb [R,3] of the recomputed
 * forward.  No host synchronisation; everything is queued on `stream`. */
/* (cams may be NULL after pnr_render_pose: the camera that render left in the workspace is used; n_cams = 1 then) */
int pnr_render_backward(const pnr_scene_t *scene, const pnr_weights_t *weights, const float *const d_w[9],
                        const float *const d_b[9], const float *d_dirs, int64_t R, const pnr_camera_t *cams,
                        int32_t n_cams, const int32_t *d_ray_cam, int64_t rays_per_cam,
                        const pnr_render_opts_t *opts, const float *d_grad_rgb, void *d_render_workspace,
                        size_t render_workspace_bytes, int64_t cap_samples, void *d_train_workspace,
                        size_t train_workspace_bytes, const pnr_grads_t *grads, float *d_rgb_recomputed,
                        void *stream);

/* ---- the confidence regulariser of the training loss -------------------------------------------------------
 * studio_model.py:288-292 gathers conf_coefficient = clamp(points_conf, 1e-4, 1) (straight-through gradient) for every
 * neighbour slot of the kept rays -- a [1, R'', SR, K] tensor in which unfilled slots read point 0
 * (studio_utils.py:193-199) -- and studio_model.py:427-429 adds mean(log v + log(1 - v)), v = clamp(., eps, 1 - eps),
 * times a weight to the loss.  pnr_conf_loss computes that mean over the neighbour lists of the LAST render in the
 * workspace without materialising the tensor: d_out[0] = the mean (NaN when no ray was kept, as torch.mean of an empty
 * tensor), d_out[1] = the tensor's element count; sums are taken in a fixed order (repeatable bits).
 * pnr_conf_loss_backward adds d mean / d points_conf times *d_upstream (a device scalar: d loss / d mean) into
 * d_grad_conf [N] -- the addends of one point are identical floats, so the result is repeatable as well.  d_conf [N] =
 * points_conf; d_scratch: pnr_conf_loss_workspace_bytes() bytes, the SAME buffer for both calls.  No host
 * synchronisation. */
size_t pnr_conf_loss_workspace_bytes(void);
int pnr_conf_loss(const pnr_scene_t *scene, const pnr_render_opts_t *opts, int64_t R, void *d_render_workspace,
                  size_t render_workspace_bytes, int64_t cap_samples, const float *d_conf, float eps, void *d_scratch,
                  float *d_out, void *stream);
int pnr_conf_loss_backward(const pnr_scene_t *scene, const pnr_render_opts_t *opts, int64_t R, void *d_render_workspace,
                           size_t render_workspace_bytes, int64_t cap_samples, const float *d_conf, float eps,
                           void *d_scratch, const float *d_fwd_out, const float *d_upstream, float *d_grad_conf,
                           void *stream);

/* ---- the optimiser half of the training step: row-sparse Adam for the point tensors --------------------------------
 * The reference trains the `neural_points` parameter group with Adam at lr 2e-3 (studio_config.py:41-47; nerfstudio's
 * AdamOptimizerConfig -> torch.optim.Adam, eps 1e-8, no weight decay, no amsgrad).  torch's dense Adam sweeps all N rows
 * of every point tensor per step although a 4096-ray batch gives ~60 k of 6 M rows a gradient.  A row whose gradient has
 * been zero since the optimiser was created has exp_avg = exp_avg_sq = 0 and dense Adam moves it by
 * -step_size * 0 / (0 + eps) = 0: Adam over the rows that EVER had a gradient is exactly dense Adam.
 *
 * pnr_rows_merge keeps that set on the device: d_flags [num_rows] int32 (zero-initialised by the caller once),
 * d_ever [ever_cap >= num_rows] the rows in order of first appearance, *d_ever_count their number (int64, zeroed once).
 * d_rows / rows_cap / d_n_rows: this step's rows -- rows_cap entries, of which the first min(*d_n_rows, rows_cap) count
 * when d_n_rows is given (what pnr_render_touched returns); entries outside [0, num_rows) are ignored.
 *
 * pnr_adam_rows applies one Adam step to the listed rows of up to PNR_ADAM_MAX_TENSORS tensors that share their row
 * count (embedding [N,32], color [N,3], dir [N,3], conf [N,1]: ONE launch), in torch.optim.Adam's arithmetic:
 *   exp_avg    += (1 - beta1) (grad - exp_avg);   exp_avg_sq = exp_avg_sq beta2 + (1 - beta2) grad grad;
 *   param      -= step_size * exp_avg / (sqrt(exp_avg_sq) / bias_correction2_sqrt + eps)
 * with step_size = lr / (1 - beta1^step) and bias_correction2_sqrt = sqrt(1 - beta2^step) evaluated by the caller (in
 * double, as torch does) for the step count AFTER its increment; the scalars travel as doubles and are cast to float once,
 * where torch casts them (1 - beta1 is formed in double first).  d_rows == NULL: every row (rows_cap = num_rows).
 * Neither call synchronises or reads anything back. */
#define PNR_ADAM_MAX_TENSORS 8
typedef struct {
    float *d_param;          /* [num_rows, width] */
    const float *d_grad;     /* [num_rows, width] */
    float *d_exp_avg;        /* [num_rows, width] */
    float *d_exp_avg_sq;     /* [num_rows, width] */
    int32_t width;           /* floats per row */
} pnr_adam_tensor_t;
int pnr_rows_merge(int32_t *d_flags, int64_t num_rows, int32_t *d_ever, int64_t *d_ever_count, int64_t ever_cap,
                   const int32_t *d_rows, int64_t rows_cap, const int64_t *d_n_rows, void *stream);
int pnr_adam_rows(const pnr_adam_tensor_t *tensors, int32_t n_tensors, int64_t num_rows, const int32_t *d_rows,
                  int64_t rows_cap, const int64_t *d_n_rows, double beta1, double beta2, double eps, double step_size,
                  double bias_correction2_sqrt, void *stream);

/* ---- probing outputs (point growing) ------------------------------------------------------------------
 * What the reference's legacy model returns with `opt.prob == 1` (models/neural_points_volumetric_model.py:331-352)
 * and run/train_studio.py:335-444 turns into new points: per ray, the shading sample of largest opacity
 * 1 - exp(-sigma * ray_dist) (the first one on ties), its world position, the distance of its nearest neighbour, and
 * the K-averages of its neighbours' colour / dir / conf / embedding under the weights the legacy aggregator returns in
 * probe mode: normalised inverse-distance weight x clamp(conf, 1e-4, 1) (point_aggregators.py:816-830).
 * Call after pnr_render / pnr_render_views / pnr_render_camera with the same cameras, options, cap_samples and
 * render workspace (sample lists, neighbour lists and densities are read from it).  Rays that are not kept get zeros
 * and index -1.  Any output pointer may be null.  No host synchronisation. */
typedef struct {
    float *d_max_opacity;    /* [R]     ray_max_shading_opacity                                   */
    float *d_max_loc;        /* [R,3]   ray_max_sample_loc_w                                      */
    float *d_far_dist;       /* [R]     ray_max_far_dist: min over the FILLED neighbour slots (1e10 if none) */
    float *d_avg_color;      /* [R,3]   shading_avg_color                                         */
    float *d_avg_dir;        /* [R,3]   shading_avg_dir                                           */
    float *d_avg_conf;       /* [R]     shading_avg_conf                                          */
    float *d_avg_embedding;  /* [R,32]  shading_avg_embedding                                     */
    int32_t *d_max_index;    /* [R]     index of that sample among the ray's selected samples      */
} pnr_probe_t;
int pnr_render_probe(const pnr_scene_t *scene, const pnr_camera_t *cams, int32_t n_cams, const int32_t *d_ray_cam,
                     int64_t rays_per_cam, const pnr_render_opts_t *opts, int64_t R, void *d_render_workspace,
                     size_t render_workspace_bytes, int64_t cap_samples, const pnr_probe_t *out, void *stream);

/* ---- per-stage device timing (bench / roofline) ------------------------------------------------- */
/* When enabled, pnr_render records hipEvents on `stream` between its stages into a ring of
 * PNR_PROFILE_SLOTS slots, one slot per call, claimed atomically (calls from several host threads or on several
 * streams get distinct slots and each records on its own stream; no host sync is added to the render).
 * pnr_profile_calls() is
 * the number of calls recorded since pnr_profile_enable(1); pnr_profile_read(call, ms) synchronises on that
 * call's last event and returns the elapsed device time of each stage in milliseconds. */
#define PNR_PROFILE_SLOTS 256
enum {
    PNR_STAGE_SELECT = 0,      /* occupancy masking + sample selection + scan + expand     */
    PNR_STAGE_KNN = 1,         /* neighbour search + valid-sample compaction               */
    PNR_STAGE_SHADE_PAIRS = 2, /* gather + mlp_base + mlp_head + density + K-aggregation   */
    PNR_STAGE_SHADE_COLOR = 3, /* colour MLP                                               */
    PNR_STAGE_COMPOSITE = 4,   /* ray_dist + alpha composite + background fill             */
    PNR_STAGE_POINT_PART = 5,  /* first-layer partial products of the distinct neighbour points (runs between
                                  KNN and SHADE_PAIRS)                                        */
    PNR_NUM_STAGES = 6
};
int pnr_profile_enable(int enable);
int64_t pnr_profile_calls(void);
int pnr_profile_read(int64_t call, float ms[PNR_NUM_STAGES]);

#ifdef __cplusplus
}
#endif
#endif /* PNR_H_ */
