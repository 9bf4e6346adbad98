// Composite stage + the fused render entry point.
// Replaces studio_model.py:368-399 (ray_dist via cummax, opacity, cumprod transmittance, RGBRenderer,
// fill_invalid with its torch.nonzero sync) by one pass over each ray's compact sample list, and ties
// the stages together behind pnr_render (NeuralPoints.forward + PointNerf.get_outputs).
#include <stdarg.h>
#include <stddef.h>

#include <algorithm>
#include <atomic>
#include <mutex>

#include "pnr_internal.h"

namespace pnr {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

constexpr int TPB = 256;

// per-stage timing: events recorded on the CALLER's stream into a ring of slots (one slot per render call, claimed
// atomically: calls from several host threads / on several streams each get their own slot and their own stream's
// events), read back on request -- the timed loop itself never synchronises
static std::atomic<bool> g_prof{false};
static std::atomic<long long> g_prof_calls{0};
static std::mutex g_prof_mu;   // event creation / enable
static hipEvent_t g_evs[PNR_PROFILE_SLOTS][PNR_NUM_STAGES + 1] = {{nullptr}};

// one thread per ray: its samples are contiguous in the compact list, at most SR of them.
__global__ void k_set_cams(CamSet set, int n, Camera *__restrict__ dst)
{
    const int i = threadIdx.x;
    if (i < n) dst[i] = set.c[i];
}

// one camera whose pose lives on the device (pnr_render_pose): position [3], rotation [9] row-major
__global__ void k_set_cam_dev(const float *__restrict__ pos, const float *__restrict__ rot, Camera *__restrict__ dst)
{
    const int i = threadIdx.x;
    float *d = reinterpret_cast<float *>(dst);
    if (i < 3) d[i] = pos[i];
    else if (i < 12) d[i] = rot[i - 3];
    else if (i < 16) d[i] = 0.f;
}

__global__ void __launch_bounds__(TPB) k_composite(CamRef cr, pnr_render_opts_t opts, int64_t R,
                                                    const int *__restrict__ ray_cnt, const int *__restrict__ ray_off,
                                                    const int *__restrict__ ray_flag,
                                                    const float4 *__restrict__ smp_loc,
                                                    const float4 *__restrict__ smp_out, const int *__restrict__ n_sel,
                                                    float *__restrict__ rgb, float *__restrict__ depth,
                                                    float *__restrict__ acc_out, int8_t *__restrict__ ray_mask,
                                                    unsigned long long *__restrict__ shards)
{
    const int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (r >= R) return;
    const int S = n_sel[0];
    const int off = ray_off[r];
    int cnt = ray_cnt[r];
    if ((int64_t)off + cnt > S) cnt = max(0, S - off);  // capacity overflow: drop what did not fit
    const bool keep = ray_flag[r] != 0 && cnt > 0;
    float cr_ = 0.f, cg = 0.f, cb = 0.f, acc = 0.f, dsum = 0.f;
    if (keep) {
        const Camera cam = load_cam_lanes(cr, cam_id(cr, r));
        const float vs = opts.vsize_z;
        const float two_vs = 2.0f * vs;
        // camera-space z of a world point: sum_j (p - o)[j] * R[j][2]   (studio_utils.py:137-144)
        auto zc = [&](float x, float y, float z) {
            const float sx = x - cam.o[0], sy = y - cam.o[1], sz = z - cam.o[2];
            return sx * cam.R[2] + sy * cam.R[5] + sz * cam.R[8];
        };
        // unfilled slots of the reference's [R,SR,3] tensor hold world (0,0,0) (cu:383): their z
        const float z_unfilled = zc(0.f, 0.f, 0.f);
        float4 p = smp_loc[off];
        float cm = zc(p.x, p.y, p.z);  // running cummax
        float T = 1.0f;
        for (int i = 0; i < cnt; ++i) {
            const float4 o = smp_out[off + i];
            const float t_i = p.w;
            float delta;
            if (i == opts.SR - 1) {
                delta = vs;  // last slot: torch.full(..., vsize[2])
            } else {
                float z_next;
                if (i + 1 < cnt) {
                    p = smp_loc[off + i + 1];
                    z_next = zc(p.x, p.y, p.z);
                } else {
                    z_next = z_unfilled;
                }
                const float cm_next = fmaxf(cm, z_next);
                delta = cm_next - cm;
                cm = cm_next;
                if (delta < 1e-8f || delta > two_vs) delta = vs;
            }
            // samples without neighbours have sigma = 0 (decoded features zero, ray_dist * valid = 0)
            const float sigma = o.x;
            const float opacity = 1.0f - expf(-sigma * delta);
            const float w = opacity * T;
            T = T * (1.0f - opacity + 1e-10f);
            cr_ += w * o.y;
            cg += w * o.z;
            cb += w * o.w;
            acc += w;
            dsum += w * t_i;
        }
    }
    float o0 = cr_ + opts.bg[0] * (1.0f - acc);
    float o1 = cg + opts.bg[1] * (1.0f - acc);
    float o2 = cb + opts.bg[2] * (1.0f - acc);
    if (!keep) {
        o0 = opts.bg[0];
        o1 = opts.bg[1];
        o2 = opts.bg[2];
    } else if (opts.eval_clamp) {
        o0 = fminf(fmaxf(o0, 0.f), 1.f);
        o1 = fminf(fmaxf(o1, 0.f), 1.f);
        o2 = fminf(fmaxf(o2, 0.f), 1.f);
    }
    rgb[3 * r] = o0;
    rgb[3 * r + 1] = o1;
    rgb[3 * r + 2] = o2;
    if (depth) depth[r] = keep ? dsum / (acc + 1e-6f) : 0.f;
    if (acc_out) acc_out[r] = keep ? acc : 0.f;
    ray_mask[r] = (int8_t)(keep ? 1 : 0);
    if (keep) shard_add(shards, SH_KEPT, 1ull);
}

// Probing outputs (neural_points_volumetric_model.py:331-352): one thread per ray.  First pass = the composite's own
// ray_dist / opacity arithmetic to find the first sample of largest opacity; second = that sample's K neighbour rows.
__global__ void __launch_bounds__(TPB) k_probe(CamRef cr, pnr_render_opts_t opts, int64_t R,
                                                const int *__restrict__ ray_cnt, const int *__restrict__ ray_off,
                                                const int *__restrict__ ray_flag, const float4 *__restrict__ smp_loc,
                                                const float4 *__restrict__ smp_out, const int *__restrict__ n_sel,
                                                const int *__restrict__ smp_pidx, const float4 *__restrict__ point_rows,
                                                pnr_probe_t o)
{
    const int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (r >= R) return;
    const int S = n_sel[0];
    const int off = ray_off[r];
    int cnt = ray_cnt[r];
    if ((int64_t)off + cnt > S) cnt = max(0, S - off);
    const bool keep = ray_flag[r] != 0 && cnt > 0;
    float best = -1.0f;
    int best_i = -1;
    float4 best_loc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (keep) {
        const Camera cam = load_cam_lanes(cr, cam_id(cr, r));
        const float vs = opts.vsize_z, two_vs = 2.0f * vs;
        auto zc = [&](float x, float y, float z) {
            const float sx = x - cam.o[0], sy = y - cam.o[1], sz = z - cam.o[2];
            return sx * cam.R[2] + sy * cam.R[5] + sz * cam.R[8];
        };
        const float z_unfilled = zc(0.f, 0.f, 0.f);
        float4 p = smp_loc[off];
        float cm = zc(p.x, p.y, p.z);
        for (int i = 0; i < cnt; ++i) {
            const float4 here = p;
            float delta;
            if (i == opts.SR - 1) {
                delta = vs;
            } else {
                float z_next = z_unfilled;
                if (i + 1 < cnt) {
                    p = smp_loc[off + i + 1];
                    z_next = zc(p.x, p.y, p.z);
                }
                const float cm_next = fmaxf(cm, z_next);
                delta = cm_next - cm;
                cm = cm_next;
                if (delta < 1e-8f || delta > two_vs) delta = vs;
            }
            const float opacity = 1.0f - expf(-smp_out[off + i].x * delta);
            if (opacity > best) {   // strictly: the first maximum wins
                best = opacity;
                best_i = i;
                best_loc = here;
            }
        }
    }
    float far = 0.f, conf_avg = 0.f, col[3] = {0.f, 0.f, 0.f}, dir[3] = {0.f, 0.f, 0.f};
    float emb[PNR_FEAT_DIM];
#pragma unroll
    for (int i = 0; i < PNR_FEAT_DIM; ++i) emb[i] = 0.f;
    if (best_i >= 0) {
        const int K = opts.K;
        const int *list = smp_pidx + ((int64_t)off + best_i) * K;
        float wsum = 0.f;
        far = 1e10f;
        for (int k = 0; k < K; ++k) {
            const int pi = list[k];
            if (pi < 0) continue;
            const float4 a0 = point_rows[(int64_t)pi * 12];
            const float dx = a0.x - best_loc.x, dy = a0.y - best_loc.y, dz = a0.z - best_loc.z;
            const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
            wsum += 1.0f / fmaxf(nrm, 1e-6f);
            far = fminf(far, nrm);
        }
        const float inv = 1.0f / fmaxf(wsum, 1e-8f);
        for (int k = 0; k < K; ++k) {
            const int pi = list[k];
            if (pi < 0) continue;
            const float4 *row = point_rows + (int64_t)pi * 12;
            const float4 a0 = row[0], c0 = row[1], c1 = row[2];
            const float dx = a0.x - best_loc.x, dy = a0.y - best_loc.y, dz = a0.z - best_loc.z;
            const float w = (1.0f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f)) * inv *
                            fminf(fmaxf(a0.w, 0.0001f), 1.0f);
            conf_avg += w * a0.w;
            col[0] += w * c0.x;
            col[1] += w * c0.y;
            col[2] += w * c0.z;
            dir[0] += w * c0.w;
            dir[1] += w * c1.x;
            dir[2] += w * c1.y;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 e = row[4 + q];
                emb[4 * q] += w * e.x;
                emb[4 * q + 1] += w * e.y;
                emb[4 * q + 2] += w * e.z;
                emb[4 * q + 3] += w * e.w;
            }
        }
    }
    if (o.d_max_opacity) o.d_max_opacity[r] = best_i >= 0 ? best : 0.f;
    if (o.d_max_index) o.d_max_index[r] = best_i;
    if (o.d_far_dist) o.d_far_dist[r] = far;
    if (o.d_avg_conf) o.d_avg_conf[r] = conf_avg;
    if (o.d_max_loc) {
        o.d_max_loc[3 * r] = best_loc.x;
        o.d_max_loc[3 * r + 1] = best_loc.y;
        o.d_max_loc[3 * r + 2] = best_loc.z;
    }
    if (o.d_avg_color)
        for (int i = 0; i < 3; ++i) o.d_avg_color[3 * r + i] = col[i];
    if (o.d_avg_dir)
        for (int i = 0; i < 3; ++i) o.d_avg_dir[3 * r + i] = dir[i];
    if (o.d_avg_embedding) {
        float4 *dst = reinterpret_cast<float4 *>(o.d_avg_embedding + (int64_t)r * PNR_FEAT_DIM);
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[q] = make_float4(emb[4 * q], emb[4 * q + 1], emb[4 * q + 2], emb[4 * q + 3]);
    }
}

// The same composite for SMALL batches (a training step's 4096 rays, an eval chunk, BASELINE cfg[0]'s 64 x 64 frame):
// one thread per ray leaves all but a few wavefronts of the device idle while each walks up to SR dependent iterations
// with two global loads apiece (23 us at 4096 rays).  Here a WAVEFRONT takes a ray: its lanes fetch the ray's samples
// together (camera-space z, ray parameter, decoded sigma / rgb) into LDS, then lane 0 runs the very loop of k_composite
// over them -- the same expressions in the same order, so the outputs are the same bits as the one-thread form's (the
// chunk loop of 2304 rays and the whole frame in one call must agree bit for bit: tests/test_gpu_plugin_eval.py).
constexpr int CW_MAXS = 128;    // selected samples per ray the LDS staging holds (SR above it: the one-thread form)
__global__ void __launch_bounds__(TPB) k_composite_wave(CamRef cr, pnr_render_opts_t opts, int64_t R,
                                                         const int *__restrict__ ray_cnt, const int *__restrict__ ray_off,
                                                         const int *__restrict__ ray_flag,
                                                         const float4 *__restrict__ smp_loc,
                                                         const float4 *__restrict__ smp_out, const int *__restrict__ n_sel,
                                                         float *__restrict__ rgb, float *__restrict__ depth,
                                                         float *__restrict__ acc_out, int8_t *__restrict__ ray_mask,
                                                         unsigned long long *__restrict__ shards)
{
    __shared__ float4 s_out[TPB / 64][CW_MAXS];
    __shared__ float s_z[TPB / 64][CW_MAXS], s_t[TPB / 64][CW_MAXS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * (TPB / 64) + wave;
    const bool exists = r < R;                       // wave-uniform; no early return: the block meets at a barrier
    int off = 0, cnt = 0;
    bool keep = false;
    Camera cam{};
    if (exists) {
        const int S = n_sel[0];
        off = ray_off[r];
        cnt = ray_cnt[r];
        if ((int64_t)off + cnt > S) cnt = max(0, S - off);
        keep = ray_flag[r] != 0 && cnt > 0;
    }
    auto zc = [&](float x, float y, float z) {
        const float sx = x - cam.o[0], sy = y - cam.o[1], sz = z - cam.o[2];
        return sx * cam.R[2] + sy * cam.R[5] + sz * cam.R[8];
    };
    if (keep) {
        cam = load_cam_lanes(cr, cam_id(cr, r));
        for (int i = lane; i < cnt; i += 64) {
            const float4 p = smp_loc[off + i];
            s_z[wave][i] = zc(p.x, p.y, p.z);
            s_t[wave][i] = p.w;
            s_out[wave][i] = smp_out[off + i];
        }
    }
    __syncthreads();
    if (!exists || lane != 0) return;
    float cr_ = 0.f, cg = 0.f, cb = 0.f, acc = 0.f, dsum = 0.f;
    if (keep) {
        const float vs = opts.vsize_z;
        const float two_vs = 2.0f * vs;
        const float z_unfilled = zc(0.f, 0.f, 0.f);
        float cm = s_z[wave][0];
        float T = 1.0f;
        for (int i = 0; i < cnt; ++i) {
            const float4 o = s_out[wave][i];
            const float t_i = s_t[wave][i];
            float delta;
            if (i == opts.SR - 1) {
                delta = vs;
            } else {
                const float z_next = (i + 1 < cnt) ? s_z[wave][i + 1] : z_unfilled;
                const float cm_next = fmaxf(cm, z_next);
                delta = cm_next - cm;
                cm = cm_next;
                if (delta < 1e-8f || delta > two_vs) delta = vs;
            }
            const float sigma = o.x;
            const float opacity = 1.0f - expf(-sigma * delta);
            const float w = opacity * T;
            T = T * (1.0f - opacity + 1e-10f);
            cr_ += w * o.y;
            cg += w * o.z;
            cb += w * o.w;
            acc += w;
            dsum += w * t_i;
        }
    }
    float o0 = cr_ + opts.bg[0] * (1.0f - acc);
    float o1 = cg + opts.bg[1] * (1.0f - acc);
    float o2 = cb + opts.bg[2] * (1.0f - acc);
    if (!keep) {
        o0 = opts.bg[0];
        o1 = opts.bg[1];
        o2 = opts.bg[2];
    } else if (opts.eval_clamp) {
        o0 = fminf(fmaxf(o0, 0.f), 1.f);
        o1 = fminf(fmaxf(o1, 0.f), 1.f);
        o2 = fminf(fmaxf(o2, 0.f), 1.f);
    }
    rgb[3 * r] = o0;
    rgb[3 * r + 1] = o1;
    rgb[3 * r + 2] = o2;
    if (depth) depth[r] = keep ? dsum / (acc + 1e-6f) : 0.f;
    if (acc_out) acc_out[r] = keep ? acc : 0.f;
    ray_mask[r] = (int8_t)(keep ? 1 : 0);
    if (keep) shard_add(shards, SH_KEPT, 1ull);
}

__global__ void k_publish_kept(const unsigned long long *__restrict__ shards, int64_t *__restrict__ counters)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) counters[PNR_CNT_RAYS_KEPT] = (int64_t)shard_sum(shards, SH_KEPT);
}

int launch_composite(const CamRef &cr, const pnr_render_opts_t &opts, int64_t R, RenderWs &ws, float *d_rgb,
                     float *d_depth, float *d_acc, int8_t *d_ray_mask, int64_t *d_counters, hipStream_t stream)
{
    unsigned long long *n_kept = ws.shards;
    // small batches: a wavefront per ray (see k_composite_wave; the same bits).  PNR_COMPOSITE_WAVE_MAX_RAYS=0: never
    static const int64_t wave_max_rays = [] {
        const char *e = getenv("PNR_COMPOSITE_WAVE_MAX_RAYS");
        return e ? (int64_t)atoll(e) : (int64_t)16384;
    }();
    if (R <= wave_max_rays && opts.SR <= CW_MAXS)
        hipLaunchKernelGGL(k_composite_wave, dim3((unsigned)((R + TPB / 64 - 1) / (TPB / 64))), dim3(TPB), 0, stream, cr,
                           opts, R, ws.ray_cnt, ws.ray_off, ws.ray_flag, ws.smp_loc, ws.smp_out, ws.n_sel, d_rgb, d_depth,
                           d_acc, d_ray_mask, n_kept);
    else
        hipLaunchKernelGGL(k_composite, dim3((unsigned)((R + TPB - 1) / TPB)), dim3(TPB), 0, stream, cr, opts, R,
                           ws.ray_cnt, ws.ray_off, ws.ray_flag, ws.smp_loc, ws.smp_out, ws.n_sel, d_rgb, d_depth, d_acc,
                           d_ray_mask, n_kept);
    hipLaunchKernelGGL(k_publish_kept, dim3(1), dim3(64), 0, stream, n_kept, d_counters);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

}  // namespace pnr

using namespace pnr;

extern "C" const char *pnr_last_error(void) { return g_err; }
extern "C" int pnr_version(void) { return PNR_VERSION; }
extern "C" int pnr_abi_sizes(int64_t out[8])
{
    if (!out) return PNR_ERR_INVALID;
    out[0] = (int64_t)sizeof(pnr_grid_params_t);
    out[1] = (int64_t)sizeof(pnr_camera_t);
    out[2] = (int64_t)sizeof(pnr_render_opts_t);
    out[3] = (int64_t)sizeof(pnr_view_t);
    out[4] = (int64_t)sizeof(pnr_grads_t);
    out[5] = (int64_t)sizeof(pnr_probe_t);
    out[6] = (int64_t)sizeof(pnr_render_taps_t);
    out[7] = (int64_t)offsetof(pnr_render_opts_t, d_tape);
    return PNR_OK;
}
extern "C" float pnr_jitter_uniform(uint32_t seed, uint32_t ray, uint32_t sample) { return pnr_uniform(seed, ray, sample); }

extern "C" size_t pnr_render_workspace_bytes(int64_t R, int64_t cap_samples, int32_t K)
{
    if (R < 1) R = 1;
    if (cap_samples < 1) cap_samples = 1;
    return carve_render_ws(nullptr, R, cap_samples, K).total;
}

extern "C" size_t pnr_render_workspace_bytes_for(const pnr_scene_t *scene, const pnr_render_opts_t *opts, int64_t R,
                                                 int64_t cap_samples)
{
    if (!scene || !opts || !scene->built) {
        set_error("pnr_render_workspace_bytes_for: null argument or scene not built");
        return 0;
    }
    if (R < 1) R = 1;
    if (cap_samples < 1) cap_samples = 1;
    return carve_render_ws(nullptr, R, cap_samples, opts->K, scene->N, scene->info[2]).total_pt;
}

// rays from cameras (pnr_render_camera): frame width, pixel list, pixels per view
struct RayGen {
    int W = 0;
    int H = 0;
    const int *pixels = nullptr;
    int64_t n_pixels = 0;
    int per_view = 0;
};

static int render_views(const pnr_scene_t *scene, const pnr_weights_t *weights, const float *d_dirs, int64_t R,
                        const CamSet &set, const float *nears, int32_t n_cams, const int32_t *d_ray_cam,
                        int64_t rays_per_cam, const RayGen *gen, const float *d_tmid, const pnr_render_opts_t *opts,
                        float *d_rgb, float *d_depth, float *d_acc, int8_t *d_ray_mask, int64_t *d_counters,
                        void *d_workspace, size_t workspace_bytes, int64_t cap_samples, hipStream_t stream,
                        const char *who, const float *d_pose_pos = nullptr, const float *d_pose_rot = nullptr)
{
    PNR_REQUIRE(scene && weights && (d_dirs || gen) && d_tmid && opts && d_rgb && d_ray_mask && d_counters &&
                    d_workspace,
                "%s: null argument", who);
    if (!scene->built || !scene->packed) {
        set_error("%s: scene not built / points not packed", who);
        return PNR_ERR_STATE;
    }
    if (!weights->packed) {
        set_error("%s: weights not packed", who);
        return PNR_ERR_STATE;
    }
    PNR_REQUIRE(R >= 1 && R < (int64_t)0x7FFFFFF0, "%s: R=%lld out of range", who, (long long)R);
    PNR_REQUIRE(n_cams >= 1 && n_cams <= PNR_MAX_CAMS, "%s: n_cams=%d not in [1,%d]", who, n_cams, PNR_MAX_CAMS);
    PNR_REQUIRE(d_ray_cam != nullptr || (rays_per_cam >= 1 && rays_per_cam * n_cams >= R),
                "%s: rays_per_cam=%lld does not cover R=%lld rays with %d cameras", who, (long long)rays_per_cam,
                (long long)R, n_cams);
    PNR_REQUIRE(opts->D >= 1 && opts->D <= PNR_MAX_D, "%s: D=%d not in [1,%d]", who, opts->D, PNR_MAX_D);
    PNR_REQUIRE(opts->K >= 1 && opts->K <= PNR_MAX_K, "%s: K=%d not in [1,%d]", who, opts->K, PNR_MAX_K);
    PNR_REQUIRE(opts->SR >= 1, "%s: SR=%d", who, opts->SR);
    PNR_REQUIRE(opts->jitter >= 0.f && opts->jitter < 1.f, "%s: jitter=%g not in [0,1)", who, opts->jitter);
    PNR_REQUIRE(opts->early_stop_eps >= 0.f && opts->early_stop_eps <= 0.01f, "%s: early_stop_eps=%g not in [0, 0.01]",
                who, opts->early_stop_eps);
    PNR_REQUIRE(opts->precision == PNR_PRECISION_FP32 || opts->precision == PNR_PRECISION_BF16X3,
                "%s: unknown precision %d", who, opts->precision);
    PNR_REQUIRE(cap_samples >= 1 && cap_samples < (int64_t)0x7FFFFFF0 / std::max(opts->K, 1),
                "%s: cap_samples=%lld out of range", who, (long long)cap_samples);
    const size_t need = pnr_render_workspace_bytes_for(scene, opts, R, cap_samples);
    if (workspace_bytes < need) {
        set_error("%s: workspace of %zu bytes < %zu required (pnr_render_workspace_bytes_for)", who, workspace_bytes,
                  need);
        return PNR_ERR_WORKSPACE;
    }
    const bool factored = true;  // both arithmetic modes start mlp_base layer 0 from the per-point table
    RenderWs ws = carve_render_ws(d_workspace, R, cap_samples, opts->K, scene->N, scene->info[2]);
    if (d_pose_pos)
        hipLaunchKernelGGL(k_set_cam_dev, dim3(1), dim3(64), 0, stream, d_pose_pos, d_pose_rot, ws.cams);
    else
        hipLaunchKernelGGL(k_set_cams, dim3(1), dim3(64), 0, stream, set, n_cams, ws.cams);
    CamRef cr{};
    cr.cams = ws.cams;
    cr.ray_cam = d_ray_cam;
    cr.rays_per_cam = d_ray_cam ? 1 : rays_per_cam;
    cr.tmid = d_tmid;
    cr.D = opts->D;
    cr.n_cams = n_cams;
    cr.jitter = opts->jitter;
    cr.seed = opts->seed;
    for (int c = 0; c < n_cams; ++c) cr.nears[c] = nears[c];
    if (gen) {
        cr.gen_rays = 1;
        cr.W = gen->W;
        cr.frame_pixels = (long long)gen->W * gen->H;
        cr.pixels = gen->pixels;
        cr.pix_per_view = gen->per_view;
        cr.n_pixels = gen->n_pixels;
    }
    // the shading stage reads directions per hit ray: the caller's tensor, or the rows k_expand generated
    const float *shade_dirs = gen ? ws.ray_dirs : d_dirs;
    const bool prof = g_prof.load();
    const long long slot = prof ? g_prof_calls.fetch_add(1) : 0;
    hipEvent_t *g_ev = g_evs[slot % PNR_PROFILE_SLOTS];
    if (prof) PNR_HIP_CHECK(hipEventRecord(g_ev[0], stream));
    int rc = launch_select_expand(scene->grid, cr, d_dirs, nullptr, R, opts->D, opts->SR, cap_samples, ws, d_counters,
                                  stream, true, factored ? scene->N : 0);
    if (rc != PNR_OK) return rc;
    if (prof) PNR_HIP_CHECK(hipEventRecord(g_ev[1], stream));
    rc = launch_knn(scene->grid, opts->K, opts->radius_limit, ws, cap_samples, d_counters, stream,
                    factored ? scene->N : 0, R, scene->params.P);
    if (rc != PNR_OK) return rc;
    // bound point tensors (pnr_points_bind, a training loop): the rows this call reads are re-packed from them first
    rc = launch_refresh_rows(scene, ws.pt_list, ws.n_sel + 3, ws.u_cap, stream);
    if (rc != PNR_OK) return rc;
    if (prof) PNR_HIP_CHECK(hipEventRecord(g_ev[2], stream));
    // events in time order: 0 select 1 knn 2 point-part 6 shade-pairs 3 shade-colour 4 composite 5
    rc = launch_shade(scene, weights, cr, shade_dirs, *opts, R, ws, cap_samples, d_counters, stream,
                      prof ? g_ev[6] : nullptr, prof ? g_ev[3] : nullptr);
    if (rc != PNR_OK) return rc;
    if (prof) PNR_HIP_CHECK(hipEventRecord(g_ev[4], stream));
    rc = launch_composite(cr, *opts, R, ws, d_rgb, d_depth, d_acc, d_ray_mask, d_counters, stream);
    if (rc != PNR_OK) return rc;
    if (prof) PNR_HIP_CHECK(hipEventRecord(g_ev[5], stream));
    return PNR_OK;
}

static void camset_of(const pnr_camera_t *cams, int n_cams, CamSet &set, float *nears)
{
    for (int c = 0; c < n_cams && c < PNR_MAX_CAMS; ++c) {
        for (int i = 0; i < 3; ++i) set.c[c].o[i] = cams[c].campos[i];
        for (int i = 0; i < 9; ++i) set.c[c].R[i] = cams[c].camrotc2w[i];
        nears[c] = cams[c].near_plane;
    }
}

static Camera camera_of(const pnr_view_t &v)
{
    Camera c{};
    for (int i = 0; i < 3; ++i) c.o[i] = v.campos[i];
    for (int i = 0; i < 9; ++i) c.R[i] = v.camrotc2w[i];
    c.fx = v.fx;
    c.fy = v.fy;
    c.cx = v.cx;
    c.cy = v.cy;
    return c;
}

extern "C" int pnr_render(const pnr_scene_t *scene, const pnr_weights_t *weights, const float *d_dirs, int64_t R,
                          const pnr_camera_t *cam, const float *d_tmid, const pnr_render_opts_t *opts, float *d_rgb,
                          float *d_depth, float *d_acc, int8_t *d_ray_mask, int64_t *d_counters, void *d_workspace,
                          size_t workspace_bytes, int64_t cap_samples, void *stream)
{
    PNR_REQUIRE(cam && d_dirs, "pnr_render: null argument");
    CamSet set{};
    float nears[PNR_MAX_CAMS] = {0};
    camset_of(cam, 1, set, nears);
    return render_views(scene, weights, d_dirs, R, set, nears, 1, nullptr, R, nullptr, d_tmid, opts, d_rgb, d_depth, d_acc,
                        d_ray_mask, d_counters, d_workspace, workspace_bytes, cap_samples, (hipStream_t)stream,
                        "pnr_render");
}

extern "C" int pnr_render_views(const pnr_scene_t *scene, const pnr_weights_t *weights, const float *d_dirs, int64_t R,
                                const pnr_camera_t *cams, int32_t n_cams, const int32_t *d_ray_cam,
                                int64_t rays_per_cam, const float *d_tmid, const pnr_render_opts_t *opts, float *d_rgb,
                                float *d_depth, float *d_acc, int8_t *d_ray_mask, int64_t *d_counters,
                                void *d_workspace, size_t workspace_bytes, int64_t cap_samples, void *stream)
{
    PNR_REQUIRE(cams && d_dirs, "pnr_render_views: null argument");
    PNR_REQUIRE(n_cams >= 1 && n_cams <= PNR_MAX_CAMS, "pnr_render_views: n_cams=%d not in [1,%d]", n_cams,
                PNR_MAX_CAMS);
    CamSet set{};
    float nears[PNR_MAX_CAMS] = {0};
    camset_of(cams, n_cams, set, nears);
    return render_views(scene, weights, d_dirs, R, set, nears, n_cams, d_ray_cam, rays_per_cam, nullptr, d_tmid, opts,
                        d_rgb, d_depth, d_acc, d_ray_mask, d_counters, d_workspace, workspace_bytes, cap_samples,
                        (hipStream_t)stream, "pnr_render_views");
}

extern "C" int pnr_render_pose(const pnr_scene_t *scene, const pnr_weights_t *weights, const float *d_dirs, int64_t R,
                               const float *d_campos, const float *d_camrotc2w, float near_plane, float far_plane,
                               const float *d_tmid, const pnr_render_opts_t *opts, float *d_rgb, float *d_depth,
                               float *d_acc, int8_t *d_ray_mask, int64_t *d_counters, void *d_workspace,
                               size_t workspace_bytes, int64_t cap_samples, void *stream)
{
    PNR_REQUIRE(d_dirs && d_campos && d_camrotc2w, "pnr_render_pose: null argument");
    (void)far_plane;   // (the far plane is in d_tmid, as for pnr_render)
    CamSet set{};
    float nears[PNR_MAX_CAMS] = {0};
    nears[0] = near_plane;
    return render_views(scene, weights, d_dirs, R, set, nears, 1, nullptr, R, nullptr, d_tmid, opts, d_rgb, d_depth, d_acc,
                        d_ray_mask, d_counters, d_workspace, workspace_bytes, cap_samples, (hipStream_t)stream,
                        "pnr_render_pose", d_campos, d_camrotc2w);
}

static int check_views(const pnr_view_t *views, int32_t n_views, int32_t H, int32_t W, const int32_t *d_pixels,
                       int64_t n_pixels, const char *who)
{
    PNR_REQUIRE(views != nullptr, "%s: null argument", who);
    PNR_REQUIRE(n_views >= 1 && n_views <= PNR_MAX_CAMS, "%s: n_views=%d not in [1,%d]", who, n_views, PNR_MAX_CAMS);
    PNR_REQUIRE(H >= 1 && W >= 1 && (int64_t)H * W < (int64_t)0x7FFFFFF0, "%s: frame %d x %d out of range", who, H, W);
    PNR_REQUIRE(n_pixels >= 1 && (d_pixels != nullptr || n_pixels <= (int64_t)H * W),
                "%s: n_pixels=%lld does not fit the %d x %d frame", who, (long long)n_pixels, H, W);
    for (int v = 0; v < n_views; ++v)
        PNR_REQUIRE(views[v].fx > 0.f && views[v].fy > 0.f, "%s: view %d has focal lengths %g, %g", who, v, views[v].fx,
                    views[v].fy);
    return PNR_OK;
}

static int render_camera_impl(const pnr_scene_t *scene, const pnr_weights_t *weights, const pnr_view_t *views,
                              int32_t n_views, int32_t H, int32_t W, const int32_t *d_pixels, int64_t n_pixels,
                              int per_view, const float *d_tmid, const pnr_render_opts_t *opts, float *d_rgb,
                              float *d_depth, float *d_acc, int8_t *d_ray_mask, int64_t *d_counters, void *d_workspace,
                              size_t workspace_bytes, int64_t cap_samples, void *stream, const char *who)
{
    int rc = check_views(views, n_views, H, W, d_pixels, n_pixels, who);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(!per_view || d_pixels != nullptr, "%s: per-view pixel lists need d_pixels", who);
    CamSet set{};
    float nears[PNR_MAX_CAMS] = {0};
    for (int v = 0; v < n_views; ++v) {
        set.c[v] = camera_of(views[v]);
        nears[v] = views[v].near_plane;
    }
    RayGen gen;
    gen.W = W;
    gen.H = H;
    gen.pixels = d_pixels;
    gen.n_pixels = n_pixels;
    gen.per_view = per_view;
    return render_views(scene, weights, nullptr, (int64_t)n_views * n_pixels, set, nears, n_views, nullptr, n_pixels, &gen,
                        d_tmid, opts, d_rgb, d_depth, d_acc, d_ray_mask, d_counters, d_workspace, workspace_bytes,
                        cap_samples, (hipStream_t)stream, who);
}

extern "C" int pnr_render_camera(const pnr_scene_t *scene, const pnr_weights_t *weights, const pnr_view_t *views,
                                 int32_t n_views, int32_t H, int32_t W, const int32_t *d_pixels, int64_t n_pixels,
                                 const float *d_tmid, const pnr_render_opts_t *opts, float *d_rgb, float *d_depth,
                                 float *d_acc, int8_t *d_ray_mask, int64_t *d_counters, void *d_workspace,
                                 size_t workspace_bytes, int64_t cap_samples, void *stream)
{
    return render_camera_impl(scene, weights, views, n_views, H, W, d_pixels, n_pixels, 0, d_tmid, opts, d_rgb, d_depth,
                              d_acc, d_ray_mask, d_counters, d_workspace, workspace_bytes, cap_samples, stream,
                              "pnr_render_camera");
}

extern "C" int pnr_render_camera_lists(const pnr_scene_t *scene, const pnr_weights_t *weights, const pnr_view_t *views,
                                       int32_t n_views, int32_t H, int32_t W, const int32_t *d_pixels,
                                       int64_t n_pixels, const float *d_tmid, const pnr_render_opts_t *opts,
                                       float *d_rgb, float *d_depth, float *d_acc, int8_t *d_ray_mask,
                                       int64_t *d_counters, void *d_workspace, size_t workspace_bytes,
                                       int64_t cap_samples, void *stream)
{
    return render_camera_impl(scene, weights, views, n_views, H, W, d_pixels, n_pixels, 1, d_tmid, opts, d_rgb, d_depth,
                              d_acc, d_ray_mask, d_counters, d_workspace, workspace_bytes, cap_samples, stream,
                              "pnr_render_camera_lists");
}

extern "C" void pnr_pinhole_ray(const pnr_view_t *view, int32_t x, int32_t y, float dir[3])
{
    const Camera c = camera_of(*view);
    pinhole_dir(c, x, y, dir[0], dir[1], dir[2]);
}

namespace pnr {
__global__ void __launch_bounds__(TPB) k_camera_rays(CamSet set, CamRef cr, int64_t R, float *__restrict__ dirs)
{
    const int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (r >= R) return;
    float dx, dy, dz;
    ray_dir(cr, set.c[cam_id(cr, r)], nullptr, r, dx, dy, dz);
    dirs[3 * r] = dx;
    dirs[3 * r + 1] = dy;
    dirs[3 * r + 2] = dz;
}
}  // namespace pnr

extern "C" int pnr_camera_rays(const pnr_view_t *views, int32_t n_views, int32_t H, int32_t W, const int32_t *d_pixels,
                               int64_t n_pixels, float *d_dirs, void *stream)
{
    int rc = check_views(views, n_views, H, W, d_pixels, n_pixels, "pnr_camera_rays");
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(d_dirs != nullptr, "pnr_camera_rays: null argument");
    CamSet set{};
    for (int v = 0; v < n_views; ++v) set.c[v] = camera_of(views[v]);
    CamRef cr{};
    cr.n_cams = n_views;
    cr.rays_per_cam = n_pixels;
    cr.gen_rays = 1;
    cr.W = W;
    cr.pixels = d_pixels;
    cr.n_pixels = n_pixels;
    const int64_t R = (int64_t)n_views * n_pixels;
    hipLaunchKernelGGL(k_camera_rays, dim3((unsigned)((R + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)stream, set, cr, R,
                       d_dirs);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

extern "C" int pnr_render_probe(const pnr_scene_t *scene, const pnr_camera_t *cams, int32_t n_cams,
                                const int32_t *d_ray_cam, int64_t rays_per_cam, const pnr_render_opts_t *opts, int64_t R,
                                void *d_render_workspace, size_t render_workspace_bytes, int64_t cap_samples,
                                const pnr_probe_t *out, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    PNR_REQUIRE(scene && cams && opts && d_render_workspace && out, "pnr_render_probe: null argument");
    if (!scene->built || !scene->packed) {
        set_error("pnr_render_probe: scene not built / points not packed");
        return PNR_ERR_STATE;
    }
    PNR_REQUIRE(R >= 1 && R < (int64_t)0x7FFFFFF0, "pnr_render_probe: R=%lld out of range", (long long)R);
    PNR_REQUIRE(n_cams >= 1 && n_cams <= PNR_MAX_CAMS, "pnr_render_probe: n_cams=%d not in [1,%d]", n_cams,
                PNR_MAX_CAMS);
    PNR_REQUIRE(d_ray_cam != nullptr || (rays_per_cam >= 1 && rays_per_cam * n_cams >= R),
                "pnr_render_probe: rays_per_cam=%lld does not cover R=%lld rays", (long long)rays_per_cam, (long long)R);
    PNR_REQUIRE(opts->K >= 1 && opts->K <= PNR_MAX_K && opts->SR >= 1, "pnr_render_probe: K=%d SR=%d", opts->K,
                opts->SR);
    PNR_REQUIRE(opts->early_stop_eps == 0.f, "pnr_render_probe: needs a render with early_stop_eps = 0");
    const size_t need = pnr_render_workspace_bytes_for(scene, opts, R, cap_samples);
    if (render_workspace_bytes < need) {
        set_error("pnr_render_probe: workspace of %zu bytes < %zu of the render it follows", render_workspace_bytes, need);
        return PNR_ERR_WORKSPACE;
    }
    RenderWs ws = carve_render_ws(d_render_workspace, R, cap_samples, opts->K, scene->N, scene->info[2]);
    CamSet set{};
    float nears[PNR_MAX_CAMS] = {0};
    camset_of(cams, n_cams, set, nears);
    hipLaunchKernelGGL(k_set_cams, dim3(1), dim3(64), 0, stream, set, n_cams, ws.cams);
    CamRef cr{};
    cr.cams = ws.cams;
    cr.ray_cam = d_ray_cam;
    cr.rays_per_cam = d_ray_cam ? 1 : rays_per_cam;
    cr.n_cams = n_cams;
    hipLaunchKernelGGL(k_probe, dim3((unsigned)((R + TPB - 1) / TPB)), dim3(TPB), 0, stream, cr, *opts, R, ws.ray_cnt,
                       ws.ray_off, ws.ray_flag, ws.smp_loc, ws.smp_out, ws.n_sel, ws.smp_pidx,
                       reinterpret_cast<const float4 *>(scene->point_rows), *out);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

namespace pnr {
__global__ void __launch_bounds__(TPB) k_touched(const int *__restrict__ n_sel, const int *__restrict__ pt_list,
                                                  int *__restrict__ out, int64_t cap, long long *__restrict__ count)
{
    const int U = n_sel[3];
    const int pad = U > 0 ? pt_list[0] : 0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < cap; i += (int64_t)gridDim.x * TPB)
        out[i] = i < U ? pt_list[i] : pad;
    if (count && blockIdx.x == 0 && threadIdx.x == 0) *count = U;
}
}  // namespace pnr

extern "C" int pnr_render_touched(const pnr_scene_t *scene, const pnr_render_opts_t *opts, int64_t R,
                                  void *d_render_workspace, size_t render_workspace_bytes, int64_t cap_samples,
                                  int32_t *d_index, int64_t index_cap, int64_t *d_count, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    PNR_REQUIRE(scene && opts && d_render_workspace && d_index, "pnr_render_touched: null argument");
    if (!scene->built) {
        set_error("pnr_render_touched: scene not built");
        return PNR_ERR_STATE;
    }
    PNR_REQUIRE(index_cap >= 1, "pnr_render_touched: index_cap=%lld", (long long)index_cap);
    const size_t need = pnr_render_workspace_bytes_for(scene, opts, R, cap_samples);
    if (render_workspace_bytes < need) {
        set_error("pnr_render_touched: workspace of %zu bytes < %zu of the render it follows", render_workspace_bytes, need);
        return PNR_ERR_WORKSPACE;
    }
    RenderWs ws = carve_render_ws(d_render_workspace, R, cap_samples, opts->K, scene->N, scene->info[2]);
    const unsigned blocks = (unsigned)std::min<int64_t>((index_cap + TPB - 1) / TPB, 1024);
    hipLaunchKernelGGL(k_touched, dim3(blocks), dim3(TPB), 0, stream, ws.n_sel, ws.pt_list, d_index, index_cap,
                       reinterpret_cast<long long *>(d_count));
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

extern "C" int pnr_profile_enable(int enable)
{
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (enable && !g_evs[0][0])
        for (int s = 0; s < PNR_PROFILE_SLOTS; ++s)
            for (int i = 0; i <= PNR_NUM_STAGES; ++i) PNR_HIP_CHECK(hipEventCreate(&g_evs[s][i]));
    g_prof_calls.store(0);
    g_prof.store(enable != 0);
    return PNR_OK;
}

extern "C" int64_t pnr_profile_calls(void) { return g_prof_calls.load(); }

extern "C" int pnr_profile_read(int64_t call, float ms[PNR_NUM_STAGES])
{
    PNR_REQUIRE(ms != nullptr, "pnr_profile_read: null argument");
    const long long recorded = g_prof_calls.load();
    if (call < 0 || call >= recorded || call < recorded - PNR_PROFILE_SLOTS) {
        set_error("pnr_profile_read: call %lld not recorded (recorded %lld, ring of %d)", (long long)call, recorded,
                  PNR_PROFILE_SLOTS);
        return PNR_ERR_STATE;
    }
    hipEvent_t *g_ev = g_evs[call % PNR_PROFILE_SLOTS];
    PNR_HIP_CHECK(hipEventSynchronize(g_ev[5]));
    static const int first[PNR_NUM_STAGES] = {0, 1, 6, 3, 4, 2}, last[PNR_NUM_STAGES] = {1, 2, 3, 4, 5, 6};
    for (int i = 0; i < PNR_NUM_STAGES; ++i)
        PNR_HIP_CHECK(hipEventElapsedTime(&ms[i], g_ev[first[i]], g_ev[last[i]]));
    return PNR_OK;
}

extern "C" int pnr_render_taps(void *d_workspace, size_t workspace_bytes, int64_t R, int64_t cap_samples, int32_t K,
                               pnr_render_taps_t *taps)
{
    PNR_REQUIRE(d_workspace && taps, "pnr_render_taps: null argument");
    if (workspace_bytes < pnr_render_workspace_bytes(R, cap_samples, K)) {
        set_error("pnr_render_taps: workspace too small");
        return PNR_ERR_WORKSPACE;
    }
    RenderWs ws = carve_render_ws(d_workspace, R, cap_samples, K);
    taps->smp_loc = reinterpret_cast<const float *>(ws.smp_loc);
    taps->smp_ray = ws.smp_ray;
    taps->smp_pidx = ws.smp_pidx;
    taps->smp_out = reinterpret_cast<const float *>(ws.smp_out);
    taps->ray_cnt = ws.ray_cnt;
    taps->ray_off = ws.ray_off;
    taps->ray_dirs = ws.ray_dirs;
    return PNR_OK;
}
