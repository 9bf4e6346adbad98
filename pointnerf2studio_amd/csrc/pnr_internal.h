// Internal declarations shared by the HIP translation units of libpnr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "pnr.h"

namespace pnr {

void set_error(const char *fmt, ...);

#define PNR_HIP_CHECK(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            ::pnr::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return PNR_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

#define PNR_REQUIRE(cond, ...)                                                                    \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            ::pnr::set_error(__VA_ARGS__);                                                        \
            return PNR_ERR_INVALID;                                                               \
        }                                                                                         \
    } while (0)

// ------------------------------------------------------------------------------------------------
// Voxel structure in HBM (built once per point cloud; read-only on the render path).
//
// The grid is cut into 4x4x4 bricks; one 64-bit word holds the occupancy of a brick
// (bit = (x&3)<<4 | (y&3)<<2 | (z&3)), bricks are linear in (bx, by, bz).  A 16-byte record per
// brick {bits, rank} turns "is cell c occupied, and which voxel is it" into ONE load plus a
// popcount: voxel id = rank + popc(bits below c) -- a collision-free minimal perfect hash of the
// occupied voxels that is ~1.5 bits per cell (1-2 MB for a nerf-synthetic scene: L2-resident),
// against two dense int32 grids (2 x 33 MB) in the reference (cu:314-319).
// Voxel v owns cand[vox_start[v] .. vox_start[v+1]): its first P points in ascending point index
// as float4 {x, y, z, bits(point index)} -- the distance test reads contiguous 16-byte records
// instead of the reference's occ_2_pnts -> in_data[pidx*3] double indirection (cu:266-270).
// ------------------------------------------------------------------------------------------------
struct BrickRec {
    unsigned long long bits;  // cells of this brick that hold points (after the compat drop)
    uint32_t rank;            // number of occupied cells in all earlier bricks
    uint32_t pad;
};

struct GridView {
    float shift[3];
    float vox[3];
    int dims[3];
    int bdims[3];   // bricks per axis
    int kernel_size[3];
    int nbricks;
    int nvox;
    const unsigned long long *occ_dil;  // [nbricks] dilated occupancy (sample selection)
    const BrickRec *rec;                // [nbricks]
    const int *vox_start;               // [nvox + 1]
    const float4 *cand;                 // [vox_start[nvox]]
};

// fp32 subtract, IEEE divide, floor -- the reference's voxel coordinate (cu:40-42); the library is
// compiled with -ffp-contract=off so nothing here is fused.
__device__ __forceinline__ bool cell_of(const GridView &g, float px, float py, float pz, int &cx, int &cy,
                                        int &cz)
{
    cx = (int)floorf((px - g.shift[0]) / g.vox[0]);
    cy = (int)floorf((py - g.shift[1]) / g.vox[1]);
    cz = (int)floorf((pz - g.shift[2]) / g.vox[2]);
    return !(cx < 0 || cx >= g.dims[0] || cy < 0 || cy >= g.dims[1] || cz < 0 || cz >= g.dims[2]);
}

__device__ __forceinline__ void brick_of(const GridView &g, int cx, int cy, int cz, int &brick, int &bit)
{
    brick = ((cx >> 2) * g.bdims[1] + (cy >> 2)) * g.bdims[2] + (cz >> 2);
    bit = ((cx & 3) << 4) | ((cy & 3) << 2) | (cz & 3);
}

struct Camera {
    float o[3];
    float R[9];  // camrotc2w row-major
    float fx, fy, cx, cy;  // pinhole intrinsics (pnr_render_camera only; 0 otherwise).  64 bytes per camera
};

// The cameras of one render call live in a small device array; ray r belongs to camera ray_cam[r] or, when
// ray_cam is null, to camera r / rays_per_cam (views concatenated back to back).  tmid holds one table of D
// coarse-sample parameters per camera.
struct CamRef {
    const Camera *cams;
    const int *ray_cam;
    long long rays_per_cam;
    const float *tmid;  // per camera [2, D]: mid-points at jitter 0, segment lengths
    int D;
    int n_cams;
    float jitter;
    unsigned seed;
    float nears[PNR_MAX_CAMS];
    // rays from cameras (pnr_render_camera): ray r = pixel pixels[r % n_pixels] (or r % n_pixels) of view r / n_pixels
    // of a frame W pixels wide; rays_per_cam == n_pixels then
    int gen_rays;
    int W;
    const int *pixels;
    long long n_pixels;
    int pix_per_view;   // pixels holds one list of n_pixels entries PER VIEW (pnr_render_camera_lists)
    long long frame_pixels;   // H * W of the frame (pnr_render_camera): the jitter stream of a ray is keyed on it
};

// Direction of pixel (x, y) of a pinhole view: the arithmetic include/pnr.h states for pnr_view_t, fp32, unfused (the
// library is built with -ffp-contract=off; the host build of pnr_pinhole_ray uses the same flags, IEEE divide and sqrt
// on both sides), so host and device agree bit for bit.
__host__ __device__ __forceinline__ void pinhole_dir(const Camera &c, int x, int y, float &dx, float &dy, float &dz)
{
    const float cxn = ((float)x + 0.5f - c.cx) / c.fx;
    const float cyn = -(((float)y + 0.5f - c.cy) / c.fy);
    const float czn = -1.0f;
    const float w0 = c.R[0] * cxn + c.R[1] * cyn + c.R[2] * czn;
    const float w1 = c.R[3] * cxn + c.R[4] * cyn + c.R[5] * czn;
    const float w2 = c.R[6] * cxn + c.R[7] * cyn + c.R[8] * czn;
    const float n = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
    dx = w0 / n;
    dy = w1 / n;
    dz = w2 / n;
}

// counter-based uniform in [0, 1) with 24 random bits: two rounds of a 32-bit mix over (seed, ray, sample);
// exported as pnr_jitter_uniform so hosts (and the test suite) can reproduce the stream
__host__ __device__ __forceinline__ float pnr_uniform(unsigned seed, unsigned ray, unsigned j)
{
    unsigned h = seed * 0x9E3779B1u + ray * 0x85EBCA77u + j * 0xC2B2AE3Du + 0x27D4EB2Fu;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    h *= 0x297A2D39u;
    h ^= h >> 15;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
struct CamSet {
    Camera c[PNR_MAX_CAMS];
};
// The key of ray r's jitter stream u(seed, key, sample).  From a direction tensor: the ray's index inside the call.
// From cameras (pnr_render_camera): view * H * W + pixel id, whatever pixel list the call was given -- a frame rendered
// as one call, as tile shards on 1 / 2 / 4 / 8 ranks, or with the owners rotated per view draws the SAME uniforms for a
// pixel (for a whole frame without a pixel list the two coincide).
__device__ __forceinline__ unsigned jitter_key(const CamRef &cr, long long r)
{
    if (!cr.gen_rays) return (unsigned)r;
    const unsigned v = (unsigned)r / (unsigned)cr.n_pixels;       // (a call renders fewer than 2^31 rays)
    const unsigned i = (unsigned)r - v * (unsigned)cr.n_pixels;
    const unsigned p = cr.pixels ? (unsigned)cr.pixels[cr.pix_per_view ? (unsigned)r : i] : i;
    return v * (unsigned)cr.frame_pixels + p;
}
__device__ __forceinline__ int cam_id(const CamRef &cr, long long r)
{
    if (cr.n_cams <= 1) return 0;
    return cr.ray_cam ? cr.ray_cam[r] : (int)((unsigned)r / (unsigned)cr.rays_per_cam);  // R < 2^31
}
__device__ __forceinline__ Camera load_cam(const CamRef &cr, int cid)
{
    Camera c;
    const float *p = reinterpret_cast<const float *>(cr.cams + cid);
#pragma unroll
    for (int i = 0; i < 3; ++i) c.o[i] = p[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) c.R[i] = p[3 + i];
    c.fx = c.fy = c.cx = c.cy = 0.f;   // the intrinsics are read by ray_dir only (load_cam_full)
    return c;
}
__device__ __forceinline__ Camera load_cam_full(const CamRef &cr, int cid)
{
    Camera c = load_cam(cr, cid);
    const float *p = reinterpret_cast<const float *>(cr.cams + cid);
    c.fx = p[12];
    c.fy = p[13];
    c.cx = p[14];
    c.cy = p[15];
    return c;
}
// direction of ray r (of camera `cam`, loaded with load_cam_full when cr.gen_rays): from the caller's tensor, or
// generated from the pixel id
__device__ __forceinline__ void ray_dir(const CamRef &cr, const Camera &cam, const float *__restrict__ dirs,
                                        long long r, float &dx, float &dy, float &dz)
{
    if (!cr.gen_rays) {
        dx = dirs[3 * r];
        dy = dirs[3 * r + 1];
        dz = dirs[3 * r + 2];
        return;
    }
    const unsigned i = (unsigned)((unsigned long long)r % (unsigned long long)cr.n_pixels);
    const unsigned p = cr.pixels ? (unsigned)cr.pixels[cr.pix_per_view ? (unsigned long long)r : (unsigned long long)i] : i;
    const unsigned y = p / (unsigned)cr.W, x = p - y * (unsigned)cr.W;
    pinhole_dir(cam, (int)x, (int)y, dx, dy, dz);
}
// camera of a per-lane ray: rays are ordered by camera, so a wavefront almost always sees ONE camera and can
// fetch it with wave-uniform (scalar) loads; only a wavefront straddling two bundles loads per lane
__device__ __forceinline__ Camera load_cam_lanes(const CamRef &cr, int cid)
{
    const int cid0 = __builtin_amdgcn_readfirstlane(cid);
    if (__all(cid == cid0)) return load_cam(cr, cid0);
    return load_cam(cr, cid);
}


// ------------------------------------------------------------------------------------------------
// device-wide exclusive scan of int32 (n read from device memory when n_dev != nullptr)
// ------------------------------------------------------------------------------------------------
size_t scan_temp_bytes(int64_t n_max);
// out[i] = sum_{j<i} in[j] for i < n; out[n] = total (out has n_max + 1 entries); if total64 != nullptr
// the total is also stored there as int64.  in == out is allowed.
int scan_exclusive_i32(const int *in, int *out, int64_t n_max, const int *n_dev, int64_t *total64,
                       void *temp, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------
}  // namespace pnr

// device buffer that is kept (and grown geometrically) across builds / updates of a scene
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
};

struct pnr_scene {
    bool built = false;
    bool packed = false;
    pnr_grid_params_t params{};
    pnr::GridView grid{};
    int64_t N = 0;
    int64_t info[8] = {0};
    // owned device buffers
    unsigned long long *occ_dil = nullptr;
    pnr::BrickRec *rec = nullptr;
    int *vox_start = nullptr;
    float4 *cand = nullptr;
    float *point_rows = nullptr;  // [N, PNR_POINT_ROW_FLOATS]
    int64_t packed_N = 0;
    size_t bytes = 0;
    // build scratch, kept so that pnr_scene_update neither allocates nor recomputes what survives:
    // cell code of every point of the current cloud ((brick << 6) | bit, 0xFFFFFFFF outside the grid) and of the
    // previous one, the raw occupancy, counters, and the per-voxel work arrays
    DevBuf<uint32_t> pt_cell[2];
    int cur_cell = 0;
    DevBuf<unsigned long long> occ_all, occ_pts, occ_dil_buf;
    DevBuf<pnr::BrickRec> rec_buf;
    DevBuf<int> popc, cnt, capped, full_start, cursor, full_list, vox_start_buf;
    DevBuf<float4> cand_buf;
    DevBuf<unsigned char> scan_tmp;
    DevBuf<int> first_valid;               // [1] smallest index of a point inside the grid
    DevBuf<unsigned long long> n_inside;   // [1]
    DevBuf<long long> dropped;             // [1] cell code of the compat-dropped voxel, -1 none
    size_t scratch_bytes = 0;
    int64_t builds = 0, updates = 0, cells_reused = 0;
    // pnr_points_bind: the caller's live tensors (xyz, embedding, conf, dir, color); every render re-packs the rows of
    // its distinct neighbour points from them
    const float *live[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t live_N = 0;
};

// Packed MLP weights.  Layer l, output tile m (32 features), k-step group g (4 k-steps), lane, 4 floats:
// exactly the A operand of v_mfma_f32_32x32x2_f32 for 4 consecutive k-steps, so a wave streams its
// weights with one coalesced 1-KiB dwordx4 load per 4 MFMAs.
struct pnr_weights {
    bool packed = false;
    float *buf = nullptr;  // all packed tensors, offsets below (in floats)
    size_t bytes = 0;
    // offsets into buf
    size_t w_off[9] = {0};    // fp32 A-operand order (layers) / plain (heads)
    size_t w16_off[9] = {0};  // bf16 hi/lo tiles (layers)
    size_t w16a_off = 0;      // mlp_base layer 0, point-only inputs [0:224]: 8 tiles x 14 k-steps
    size_t w16b_off = 0;
    size_t w8acc_off = 0;     // colour head weights in accumulator order (384 floats)
    size_t w32a_off = 0, w32b_off = 0;  // fp32-packed halves of mlp_base layer 0
    size_t w4acc_off = 0;     // density head weights in accumulator order (256 floats)      // mlp_base layer 0, pair inputs [224:284]: 4 ring tiles x (2 row blocks x 4 k-steps)
    size_t b_off[9] = {0};
    float Rw2c[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
};

namespace pnr {

// layer geometry (k-steps of 2 input features; padded to a multiple of 4 k-steps)
constexpr int KS_BASE0 = 142;  // 284 / 2
constexpr int KS_HID = 128;    // 256 / 2
constexpr int KS_HEAD0 = 132;  // 264 / 2 (263 + 1 zero pad)
constexpr int KS_COL0 = 140;   // 280 / 2
constexpr int KS_COLH = 64;    // 128 / 2
constexpr int kg_of(int ks) { return (ks + 3) / 4; }

// workspace carving shared by pnr_render / taps
struct RenderWs {
    int *ray_cnt;
    int *ray_off;   // [R+1]
    int *ray_flag;
    unsigned long long *ray_bits;  // [R, 8]
    float4 *smp_loc;
    int *smp_ray;
    int *smp_pidx;   // [cap, K]
    int *smp_valid;  // [cap]
    int *smp_voff;   // [cap+1]
    int *vs_list;    // [cap]
    float *smp_sigma;  // [cap] by valid index
    float *agg;        // [cap, 256] by valid index
    float4 *smp_out;   // [cap]
    // early ray termination (opts.early_stop_eps > 0): shade-only buffers behind agg
    float *smp_sig_s;  // [cap] density by SAMPLE index, 0 for samples not (yet) shaded
    int *vs_all;       // [cap] samples in shading order: the passes' lists back to back
    float *ray_T;      // [R] transmittance after the chunks shaded so far
    float *ray_cm;     // [R] running maximum of the camera-space depth (ray_dist state of the composite)
    int *ray_alive;    // [R]
    float *ray_dirs;   // [R,3] pnr_render_camera: directions of the rays that have samples (written by k_expand)
    float *smp_wgt;    // [cap, K] normalised inverse-distance weight of every neighbour slot (k_pair_weights; the pair
                       // kernel on dense units reads its row's weight instead of summing over the sample's rows)
    bool wgt_from_knn; // this call's neighbour search wrote smp_wgt itself (k_knn3<16, true>): no k_pair_weights pass
    bool pt_flag_cleared, out_cleared;   // this call's k_render_init zeroed pt_flag / smp_out (launch_select_expand)
    int *n_sel;        // device ints: [0]=S_sel (clamped to cap), [1]=S_valid, [2]=R, [3]=U unique neighbour points,
                       // [4],[5] = first / one-past-last position of the current shading pass in vs_all, [6] = 0
    unsigned long long *shards;  // statistics counters, SHARDS x 128-byte lines per counter (see shard_add)
    Camera *cams;                // [PNR_MAX_CAMS] cameras of the call
    void *scan_temp;
    size_t total;       // bytes without the point-part buffers below
    // bf16x3 mode, factorised first layer (see k_point_part): only when carved with N > 0
    int *pt_flag;       // [N]    1 = point is a neighbour of some sample of this call
    int *pt_rank;       // [N+1]  exclusive scan of pt_flag = row of the point in pt_table; [N] = U
    int *pt_list;       // [u_cap] unique neighbour points, ascending point index
    void *pt_scan_temp;
    float *pt_table;    // [u_cap, 256] W1[:, 0:224] . [emb, PE(emb)] + b1 in accumulator order
    int64_t u_cap;
    size_t total_pt;    // bytes including them
};
// N = 0: no point-part buffers (query-only and fp32 workspaces)
RenderWs carve_render_ws(void *base, int64_t R, int64_t cap, int K, int64_t N = 0, int64_t n_list = 0);

// Statistics counters (rays hit, pairs, candidates, rays kept) are summed with atomics that nothing waits for.
// One word saturates at ~88 atomics/us on MI355X: 128k waves adding to ONE counter cost 1.5 ms of a 20 ms frame.
// Each counter is therefore spread over 64 cache lines indexed by workgroup; the lines are summed when published.
constexpr int SHARDS = 64;
constexpr int SHARD_STRIDE = 16;  // u64 per 128-byte line
enum { SH_RAYS_HIT = 0, SH_PAIRS = 1, SH_CAND = 2, SH_KEPT = 3, SH_COUNT = 4 };
__device__ __forceinline__ void shard_add(unsigned long long *shards, int counter, unsigned long long v)
{
    atomicAdd(&shards[((size_t)counter * SHARDS + (blockIdx.x & (SHARDS - 1))) * SHARD_STRIDE], v);
}
__device__ __forceinline__ unsigned long long shard_sum(const unsigned long long *shards, int counter)
{
    unsigned long long s = 0;
    for (int i = 0; i < SHARDS; ++i) s += shards[((size_t)counter * SHARDS + i) * SHARD_STRIDE];
    return s;
}

// Raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) ONCE PER DEVICE -- a process may drive
// several devices, and the attribute is per device -- and reports a failure; also hands back the device's CU count.
int ensure_dynamic_lds(const void *kernel, int bytes, int *cus = nullptr);

// Zero-fill as a KERNEL (16-byte stores where pointer and size allow).  The calls of the render path issue no
// hipMemsetAsync: a call captured into a hipGraph then consists of kernel nodes only (a graph with memset nodes faulted
// on its second launch under ROCm 7.2: tools/graph_replay.py), and the clears of a render are one launch instead of six.
int zero_async(void *p, size_t bytes, hipStream_t stream);

// d_dirs: the caller's directions; with cr.gen_rays they are null and k_expand writes ws.ray_dirs instead.
// Starts with the call's ONE clearing launch (k_render_init): n_sel, the statistics shards, d_counters, ray_flag and --
// render_clears: a render call follows with the neighbour search and the shading stage -- pt_flag [N] (when carved) and
// smp_out [cap]; ws.pt_flag_cleared / ws.out_cleared tell launch_knn / launch_shade.
int launch_select_expand(const GridView &g, const CamRef &cr, const float *d_dirs, const float *d_raypos,
                         int64_t R, int D, int SR, int64_t cap, RenderWs &ws, int64_t *d_counters,
                         hipStream_t stream, bool render_clears = false, int64_t N = 0);
// R (the call's ray count) and P (the scene's points per voxel) let small batches take the cooperative search
int launch_knn(const GridView &g, int K, float radius_limit, RenderWs &ws, int64_t cap, int64_t *d_counters,
               hipStream_t stream, int64_t N = 0, int64_t R = 0, int P = 0);
int launch_shade(const pnr_scene *scene, const pnr_weights *w, const CamRef &cr, const float *d_dirs,
                 const pnr_render_opts_t &opts, int64_t R, RenderWs &ws, int64_t cap, int64_t *d_counters,
                 hipStream_t stream, hipEvent_t ev_points, hipEvent_t ev_between);
// pnr_train.hip: the four activation tapes (H1, H2, G1, G2) inside a training workspace of `bytes` bytes carved for
// (cap, K); false if the workspace is too small.  tape_supported: the configurations in which a render given
// opts.d_tape fills them -- pnr_render* and pnr_render_backward must agree on it.
bool train_tape_ptrs(void *d_train_workspace, size_t bytes, int64_t cap, int K, float *tape[4], size_t tape_bytes[4],
                     unsigned **tape_bits, size_t *tape_bits_rows, float **tape_rowz, float *ctape[3], void **tape_sg);
inline bool tape_supported(const pnr_render_opts_t &o)
{
    return o.d_tape != nullptr && o.precision == PNR_PRECISION_FP32 && o.early_stop_eps == 0.f &&
           (o.K <= 10 || o.K == 16);   // the K served by k_shade_pairs<8 / 16> (dense units and K > 16 recompute)
}
int launch_refresh_rows(const pnr_scene *scene, const int *pt_list, const int *n_unique, int64_t u_cap,
                        hipStream_t stream);
int launch_composite(const CamRef &cr, const pnr_render_opts_t &opts, int64_t R, RenderWs &ws, float *d_rgb,
                     float *d_depth, float *d_acc, int8_t *d_ray_mask, int64_t *d_counters,
                     hipStream_t stream);

}  // namespace pnr
