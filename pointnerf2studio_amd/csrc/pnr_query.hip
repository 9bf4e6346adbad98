// Query stage: coarse-sample occupancy masking, first-SR sample selection, fixed-K neighbour search.
//
// Replaces mask_raypos / host slotting / get_shadingloc / query_neigh_along_ray_layered / host
// post-filter (query_worldcoords.cu:165-302,381-429).  What is different on MI355X:
//   * a 64-lane wavefront walks one ray at a time: the D coarse samples are probed 64 at a time against the
//     L2-resident dilated-occupancy bitmask, `__ballot` + popcount give each hit its slot (the
//     reference's cumsum over a [R,D] int tensor), and raypos[R,D,3] is never materialised: a sample
//     position is campos + dir * t_mid[j], evaluated where needed (mul then add, unfused, as torch does);
//   * shading samples of all rays live in ONE compact list (ray -> [off, off+cnt)), sized on the
//     device by a scan -- no `.item()` syncs, no masked_select copies;
//   * the neighbour search reads 16-byte {xyz, index} records that are contiguous per voxel.
// The neighbour lists are bit-exact against the sequential oracle: same traversal order
// (layer -> x -> y -> z -> slot), same replace-the-farthest rule, same fp32 distance expression.
#include <algorithm>

#include <stdlib.h>

#include "pnr_internal.h"

namespace pnr {

constexpr int TPB = 256;

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

RenderWs carve_render_ws(void *base, int64_t R, int64_t cap, int K, int64_t N, int64_t n_list)
{
    RenderWs ws{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *p = (void *)((uintptr_t)base + off);
        off += align_up(bytes);
        return p;
    };
    ws.n_sel = (int *)take(64 * sizeof(int));
    ws.shards = (unsigned long long *)take((size_t)SH_COUNT * SHARDS * SHARD_STRIDE * sizeof(unsigned long long));
    ws.cams = (Camera *)take(sizeof(Camera) * PNR_MAX_CAMS);
    ws.ray_cnt = (int *)take((size_t)(R + 1) * sizeof(int));
    ws.ray_off = (int *)take((size_t)(R + 1) * sizeof(int));
    ws.ray_flag = (int *)take((size_t)(R + 1) * sizeof(int));
    ws.ray_bits = (unsigned long long *)take((size_t)R * 8 * sizeof(unsigned long long));
    ws.smp_loc = (float4 *)take((size_t)cap * sizeof(float4));
    ws.smp_ray = (int *)take((size_t)cap * sizeof(int));
    ws.smp_pidx = (int *)take((size_t)cap * K * sizeof(int));
    ws.smp_valid = (int *)take((size_t)(cap + 1) * sizeof(int));
    ws.smp_voff = (int *)take((size_t)(cap + 1) * sizeof(int));
    ws.vs_list = (int *)take((size_t)cap * sizeof(int));
    ws.smp_out = (float4 *)take((size_t)cap * sizeof(float4));
    ws.scan_temp = take(scan_temp_bytes(std::max<int64_t>(R, cap) + 1));
    // shade-only buffers last so that the query-only workspace is a prefix
    ws.smp_sigma = (float *)take((size_t)cap * sizeof(float));
    ws.agg = (float *)take((size_t)(cap + 32) * 256 * sizeof(float));  // packed layout: whole 32-sample blocks
    ws.smp_sig_s = (float *)take((size_t)cap * sizeof(float));
    ws.vs_all = (int *)take((size_t)cap * sizeof(int));
    ws.ray_T = (float *)take((size_t)R * sizeof(float));
    ws.ray_cm = (float *)take((size_t)R * sizeof(float));
    ws.ray_alive = (int *)take((size_t)R * sizeof(int));
    ws.ray_dirs = (float *)take((size_t)R * 3 * sizeof(float));
    ws.smp_wgt = (float *)take((size_t)cap * K * sizeof(float));
    ws.total = off;
    if (N > 0) {
        ws.u_cap = std::max<int64_t>(1, std::min<int64_t>(n_list, cap * (int64_t)K));
        ws.pt_flag = (int *)take((size_t)N * sizeof(int));
        ws.pt_rank = (int *)take((size_t)(N + 1) * sizeof(int));
        ws.pt_list = (int *)take((size_t)ws.u_cap * sizeof(int));
        ws.pt_table = (float *)take((size_t)(ws.u_cap + 128) * 256 * sizeof(float));  // + one tile of padding rows
        ws.pt_scan_temp = take(scan_temp_bytes(N + 1));
    }
    ws.total_pt = off;
    return ws;
}

// ------------------------------------------------------------------------------------------------
// k_select: bits[r][w] = occupancy of coarse samples 64w..64w+63,
// cnt[r] = min(SR, hits).  Either explicit positions (d_raypos, the drop-in op) or o + d * t.
// ------------------------------------------------------------------------------------------------
// Ray parameter of coarse sample j = 64w + lane.  jitter == 0: the host table.  jitter > 0: the reference's
// arithmetic per ray (diff_ray_marching.py:312-323): seg_j * (1 + jitter * (u_j - 0.5)); running sum (torch's CPU
// cumsum accumulates float32 in double and rounds every prefix); + near; mid-points of consecutive end points.
// `carry` (running double sum) and `e_prev` (end point of sample 64w - 1) flow from word to word.
// jkey: jitter_key(cr, ray), computed ONCE per ray by the caller (it holds a division)
//
// The running sum over a wavefront's 64 segments: an inclusive scan of doubles on DPP moves (row shifts inside the rows
// of 16 lanes, then the two row broadcasts: the GCN wavefront scan) -- 14 register moves and 7 adds where six rounds of
// __shfl_up cost twelve ds_bpermute round trips through the LDS crossbar, per 64 samples of every hit ray, in k_select
// and again in k_expand.  Lanes without a source read 0.0 and add it.  The segments are float32 values of one magnitude:
// every partial sum of <= 512 of them is EXACT in double (24 + 9 significant bits), so the order of the additions does
// not change a bit of any prefix (the oracle adds sequentially).
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_incl_scan_f64(double s)
{
    constexpr int ROW_SHR = 0x110, ROW_BCAST15 = 0x142, ROW_BCAST31 = 0x143;
    double a = s + dpp_mov_f64<ROW_SHR + 1, 0xf, 0xf>(s);
    a += dpp_mov_f64<ROW_SHR + 2, 0xf, 0xf>(s);
    a += dpp_mov_f64<ROW_SHR + 3, 0xf, 0xf>(s);          // lanes' last four inside their row of 16
    a += dpp_mov_f64<ROW_SHR + 4, 0xf, 0xe>(a);          // lanes 4..15 of a row: + the four before
    a += dpp_mov_f64<ROW_SHR + 8, 0xf, 0xc>(a);          // lanes 8..15: + the eight before -- rows are scanned
    a += dpp_mov_f64<ROW_BCAST15, 0xa, 0xf>(a);          // rows 1, 3: + the total of the row before
    a += dpp_mov_f64<ROW_BCAST31, 0xc, 0xf>(a);          // rows 2, 3: + the total of the first 32 lanes
    return a;
}
__device__ __forceinline__ double bcast_f64(double v, int L)   // L wave-uniform
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), L), __builtin_amdgcn_readlane(__double2loint(v), L));
}
__device__ __forceinline__ float sample_t(const CamRef &cr, const float *__restrict__ tab, float near_plane,
                                          unsigned jkey, int D, int w, int lane, double &carry, float &e_prev)
{
    const int j = w * 64 + lane;
    if (cr.jitter == 0.0f) return j < D ? tab[j] : 0.f;
    float seg = 0.f;
    if (j < D) {
        const float u = pnr_uniform(cr.seed, jkey, (unsigned)j);
        seg = tab[D + j] * (1.0f + cr.jitter * (u - 0.5f));
    }
    double s = wave_incl_scan_f64((double)seg);
    s += carry;
    carry = bcast_f64(s, 63);
    const float e = near_plane + (float)s;          // end point of sample j
    float e_left = __shfl_up(e, 1, 64);
    if (lane == 0) e_left = e_prev;
    e_prev = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 63));
    return (e_left + e) / 2.0f;
}

__device__ __forceinline__ void sample_pos(const float *__restrict__ raypos, const float (&rd)[3],
                                           const Camera &cam, int64_t r, int D, int j, float t, float &px, float &py,
                                           float &pz)
{
    if (raypos) {
        const float *p = raypos + ((int64_t)r * D + j) * 3;
        px = p[0];
        py = p[1];
        pz = p[2];
    } else {
        float dx = rd[0], dy = rd[1], dz = rd[2];
        float mx = dx * t, my = dy * t, mz = dz * t;  // unfused: raydir * t, then campos + (.)
        px = cam.o[0] + mx;
        py = cam.o[1] + my;
        pz = cam.o[2] + mz;
    }
}

// uniform broadcast of lane L's value (L wave-uniform)
__device__ __forceinline__ float bcast_f(float v, int L)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L));
}
__device__ __forceinline__ int bcast_i(int v, int L) { return __builtin_amdgcn_readlane(v, L); }

// A wavefront owns RPW consecutive rays.  Phase 1, one ray per lane: camera, ray / grid-box clip, word range.  Phase 2:
// the rays whose range is not empty are probed one after the other by the whole wavefront, 64 coarse samples per
// step, the occupancy words of all steps of a ray in flight together.  (One wavefront per ray spent its time
// launching 640 k wavefronts, four out of five of them for a ray that misses the box; 64 rays per wavefront made the
// chain of dependent loads of its ~15 live rays the bottleneck instead.)
#ifndef PNR_RPW
#define PNR_RPW 16
#endif
constexpr int RPW = PNR_RPW;
constexpr int MAXW = PNR_MAX_D / 64;  // occupancy words per ray
template <int RW>
__global__ void __launch_bounds__(TPB) k_select(GridView g, CamRef cr, const float *__restrict__ dirs,
                                                 const float *__restrict__ raypos,
                                                 int64_t R, int D, int SR, int *__restrict__ ray_cnt,
                                                 unsigned long long *__restrict__ ray_bits,
                                                 unsigned long long *__restrict__ shards)
{
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6)) * RW;
    if (r0 >= R) return;
    const int64_t rl = r0 + lane;
    const bool live = lane < RW && rl < R;
    const int nwords = (D + 63) >> 6;
    const bool jittered = !raypos && cr.jitter != 0.0f;
    // ---- phase 1: this lane's ray ---------------------------------------------------------------------------
    int cid = 0, w_lo = 1, w_hi = 0;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (live) {
        w_lo = 0;
        w_hi = nwords - 1;
        if (!raypos) {
            cid = cam_id(cr, rl);
            const Camera cam = load_cam_full(cr, cid);
            const float *tmid = cr.tmid + (size_t)cid * 2 * D;
            ray_dir(cr, cam, dirs, rl, dx, dy, dz);
            // Conservative ray / grid-box clip: a coarse sample outside the voxel grid can never be occupied
            // (cu:182-187), so whole 64-sample words whose parameter range misses the box -- grown by two voxels
            // against rounding -- are skipped without probing.  Every probed sample still takes the exact test.
            float t_in = -3.0e38f, t_out = 3.0e38f;
            const float d[3] = {dx, dy, dz};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float lo = g.shift[a] - 2.0f * g.vox[a], hi = g.shift[a] + (float)(g.dims[a] + 2) * g.vox[a];
                if (fabsf(d[a]) > 1e-12f) {
                    const float ta = (lo - cam.o[a]) / d[a], tb = (hi - cam.o[a]) / d[a];
                    t_in = fmaxf(t_in, fminf(ta, tb));
                    t_out = fminf(t_out, fmaxf(ta, tb));
                } else if (cam.o[a] < lo || cam.o[a] > hi) {
                    t_out = -3.0e38f;  // parallel to the slab and outside it
                }
            }
            // word range that can intersect the box.  The table is monotonic and linear: the range comes from its end
            // points with 3 samples of slack, then the first/last kept word is verified against the table.
            // Jittered parameters drift from the table, but boundedly: sample j sits at near + f * (table_j - near)
            // with f in [1 - jitter/2, 1 + jitter/2) (every segment is scaled by 1 + jitter * (u - 0.5), u in [0, 1)),
            // so it can lie in [t_in, t_out] only if its TABLE value lies in
            // [near + (t_in - near) / (1 + jitter/2), near + (t_out - near) / (1 - jitter/2)].
            if (D > 1) {
                if (jittered) {
                    const float nr = cr.nears[cid];
                    const float f_hi = 1.0f + 0.5f * cr.jitter, f_lo = 1.0f - 0.5f * cr.jitter;
                    // (t - near) may be negative: the division by the wider / narrower factor then moves the bound
                    // the other way, so take the looser of the two on each side
                    const float a_in = t_in - nr, a_out = t_out - nr;
                    t_in = nr + fminf(a_in / f_hi, a_in / f_lo) - 1e-4f;
                    t_out = nr + fmaxf(a_out / f_lo, a_out / f_hi) + 1e-4f;
                }
                const float t0 = tmid[0], t1 = tmid[D - 1];
                const float inv_dt = (float)(D - 1) / (t1 - t0);
                const float jl = (t_in - t0) * inv_dt - 3.0f, jh = (t_out - t0) * inv_dt + 3.0f;
                if (!(jh >= 0.f) || !(jl <= (float)(D - 1))) {
                    w_lo = 1;
                    w_hi = 0;  // the ray misses the box: nothing to probe
                } else {
                    w_lo = (int)fmaxf(jl, 0.f) >> 6;
                    w_hi = (int)fminf(jh, (float)(D - 1)) >> 6;
                    while (w_lo > 0 && tmid[w_lo * 64 - 1] >= t_in) --w_lo;
                    while (w_hi < nwords - 1 && tmid[(w_hi + 1) * 64] <= t_out) ++w_hi;
                }
            }
        }
        if (w_lo > w_hi) ray_cnt[rl] = 0;  // (ray_bits of such a ray are never read: k_expand skips it)
    }
    // ---- phase 2: the rays with something to probe, one at a time, 64 samples per step ----------------------------
    unsigned long long todo = __ballot(live && w_lo <= w_hi);
    while (todo) {
        const int L = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int64_t r = r0 + L;
        const int wl = bcast_i(w_lo, L), wh = bcast_i(w_hi, L);
        Camera cam{};
        const float *tmid = nullptr;
        float near_plane = 0.f;
        float rd[3] = {0.f, 0.f, 0.f};
        if (!raypos) {
            const int c = bcast_i(cid, L);
            cam = load_cam(cr, c);
            tmid = cr.tmid + (size_t)c * 2 * D;
            near_plane = cr.nears[c];
            rd[0] = bcast_f(dx, L);
            rd[1] = bcast_f(dy, L);
            rd[2] = bcast_f(dz, L);
        }
        int total = 0;
        unsigned long long my_word = 0ull;  // lane w keeps word w
        double carry = 0.0;
        float e_prev = near_plane;
        // all probes of the ray first (one occupancy word per step, loads in flight together), then the ballots
        unsigned long long occ[MAXW];
        int bitpos[MAXW];
        // jittered parameters are a running sum from sample 0: the scan runs over every word up to wh, the probes
        // over [wl, wh] only
        const int w_first = jittered ? 0 : wl;
        const unsigned jkey = jittered ? jitter_key(cr, r) : 0u;
#pragma unroll
        for (int k = 0; k < MAXW; ++k) {
            occ[k] = 0ull;
            bitpos[k] = 0;
            const int w = w_first + k;
            if (w > wh) continue;  // wave-uniform
            const int j = w * 64 + lane;
            const float t = raypos ? 0.f : sample_t(cr, tmid, near_plane, jkey, D, w, lane, carry, e_prev);
            if (j < D && w >= wl) {
                float px, py, pz;
                sample_pos(raypos, rd, cam, r, D, j, t, px, py, pz);
                int cx, cy, cz;
                if (cell_of(g, px, py, pz, cx, cy, cz)) {
                    int brick;
                    brick_of(g, cx, cy, cz, brick, bitpos[k]);
                    occ[k] = g.occ_dil[brick];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < MAXW; ++k) {
            const int w = w_first + k;
            if (w > wh) continue;
            const unsigned long long m = __ballot((occ[k] >> bitpos[k]) & 1ull);
            if (lane == w) my_word = m;
            total += __popcll(m);
        }
        if (lane < 8) ray_bits[r * 8 + lane] = my_word;  // one 64-byte store per ray
        if (lane == 0) {
            ray_cnt[r] = min(total, SR);
            if (total > 0) shard_add(shards, SH_RAYS_HIT, 1ull);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_expand: a wavefront per RPW rays; the rays with samples are expanded one at a time by the whole wavefront:
// hit j with rank < SR becomes sample off[r] + rank.
// ------------------------------------------------------------------------------------------------
template <int RW>
__global__ void __launch_bounds__(TPB) k_expand(CamRef cr, const float *__restrict__ dirs,
                                                 const float *__restrict__ raypos,
                                                 int64_t R, int D, int SR, const int *__restrict__ ray_off,
                                                 const unsigned long long *__restrict__ ray_bits, int64_t cap,
                                                 float4 *__restrict__ smp_loc, int *__restrict__ smp_ray,
                                                 int *__restrict__ n_sel, int64_t *__restrict__ counters,
                                                 float *__restrict__ ray_dirs)
{
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6)) * RW;
    if (r0 == 0 && lane == 0) {
        // total selected samples, clamped to the workspace capacity
        int total = ray_off[R];
        n_sel[0] = (int)min((int64_t)total, cap);
        counters[PNR_CNT_SAMPLES_SELECTED] = total;
        if (total > cap) counters[PNR_CNT_OVERFLOW] = 1;
    }
    if (r0 >= R) return;
    const int64_t rl = r0 + lane;
    int my_off = 0, my_n = 0, cid = 0;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (lane < RW && rl < R) {
        my_off = ray_off[rl];
        my_n = ray_off[rl + 1] - my_off;
        if (my_n > 0 && !raypos) {
            cid = cam_id(cr, rl);
            if (cr.gen_rays) {
                // rays from cameras: the direction is generated here, and kept for the shading stage -- for the rays
                // that have samples only (the rest of [R,3] is never written or read)
                ray_dir(cr, load_cam_full(cr, cid), nullptr, rl, dx, dy, dz);
                ray_dirs[3 * rl] = dx;
                ray_dirs[3 * rl + 1] = dy;
                ray_dirs[3 * rl + 2] = dz;
            } else {
                dx = dirs[3 * rl];
                dy = dirs[3 * rl + 1];
                dz = dirs[3 * rl + 2];
            }
        }
    }
    unsigned long long todo = __ballot(my_n > 0);
    const int nwords = (D + 63) >> 6;
    while (todo) {
        const int L = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int64_t r = r0 + L;
        const int off = bcast_i(my_off, L);
        Camera cam{};
        const float *tmid = nullptr;
        float near_plane = 0.f;
        float rd[3] = {0.f, 0.f, 0.f};
        if (!raypos) {
            const int c = bcast_i(cid, L);
            cam = load_cam(cr, c);
            tmid = cr.tmid + (size_t)c * 2 * D;
            near_plane = cr.nears[c];
            rd[0] = bcast_f(dx, L);
            rd[1] = bcast_f(dy, L);
            rd[2] = bcast_f(dz, L);
        }
        int base = 0;
        double carry = 0.0;
        float e_prev = near_plane;
        // the ray's eight occupancy words: one 64-byte load, word w broadcast from lane w
        const unsigned long long mine = lane < 8 ? ray_bits[r * 8 + lane] : 0ull;
        const unsigned jkey = cr.jitter != 0.0f ? jitter_key(cr, r) : 0u;
        for (int w = 0; w < nwords && base < SR; ++w) {
            const unsigned long long m = __shfl(mine, w, 64);
            const float t = raypos ? 0.f : sample_t(cr, tmid, near_plane, jkey, D, w, lane, carry, e_prev);
            if ((m >> lane) & 1ull) {
                const int rank = base + __popcll(m & ((1ull << lane) - 1ull));
                const int64_t s = (int64_t)off + rank;
                if (rank < SR && s < cap) {
                    float px, py, pz;
                    sample_pos(raypos, rd, cam, r, D, w * 64 + lane, t, px, py, pz);
                    smp_loc[s] = make_float4(px, py, pz, t);
                    smp_ray[s] = (int)r;
                }
            }
            base += __popcll(m);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_knn: one thread per selected sample (query_neigh_along_ray_layered, cu:217-302).
// ------------------------------------------------------------------------------------------------
template <int KMAX>
__global__ void __launch_bounds__(TPB) k_knn(GridView g, int K, float radius_limit2,
                                              const float4 *__restrict__ smp_loc, const int *__restrict__ smp_ray,
                                              const int *__restrict__ n_sel, int *__restrict__ smp_pidx,
                                              int *__restrict__ smp_valid, int *__restrict__ ray_flag,
                                              unsigned long long *__restrict__ shards, int *__restrict__ pt_flag)
{
    const int S = n_sel[0];
    for (int64_t s = (int64_t)blockIdx.x * TPB + threadIdx.x; s < S; s += (int64_t)gridDim.x * TPB) {
        const float4 c = smp_loc[s];
        int fx, fy, fz;
        cell_of(g, c.x, c.y, c.z, fx, fy, fz);  // selected samples are inside the grid by construction
        int kid = 0, far_ind = 0;
        float far2 = 0.0f;
        float buf[KMAX];
        int out[KMAX];
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            buf[i] = 0.f;
            out[i] = -1;
        }
        unsigned tested = 0;
        const int nlayers = (g.kernel_size[0] + 1) / 2;
        for (int layer = 0; layer < nlayers; ++layer) {
            for (int x = max(-fx, -layer); x < min(g.dims[0] - fx, layer + 1); ++x) {
                for (int y = max(-fy, -layer); y < min(g.dims[1] - fy, layer + 1); ++y) {
                    for (int z = max(-fz, -layer); z < min(g.dims[2] - fz, layer + 1); ++z) {
                        if (max(abs(z), max(abs(x), abs(y))) != layer) continue;
                        int brick, bit;
                        brick_of(g, fx + x, fy + y, fz + z, brick, bit);
                        const BrickRec rec = g.rec[brick];
                        const unsigned long long mbit = 1ull << bit;
                        if (!(rec.bits & mbit)) continue;
                        const int v = (int)rec.rank + __popcll(rec.bits & (mbit - 1ull));
                        const int vs = g.vox_start[v], ve = g.vox_start[v + 1];
                        for (int q = vs; q < ve; ++q) {
                            const float4 p = g.cand[q];
                            const float xv = p.x - c.x, yv = p.y - c.y, zv = p.z - c.z;
                            const float d2 = xv * xv + yv * yv + zv * zv;  // left-to-right, unfused
                            ++tested;
                            if (radius_limit2 == 0.0f || d2 <= radius_limit2) {
                                const int pidx = __float_as_int(p.w);
                                if (kid++ < K) {
                                    const int slot = kid - 1;
#pragma unroll
                                    for (int i = 0; i < KMAX; ++i)
                                        if (i == slot) {
                                            out[i] = pidx;
                                            buf[i] = d2;
                                        }
                                    if (d2 > far2) {
                                        far2 = d2;
                                        far_ind = slot;
                                    }
                                } else if (d2 < far2) {
#pragma unroll
                                    for (int i = 0; i < KMAX; ++i)
                                        if (i == far_ind) {
                                            out[i] = pidx;
                                            buf[i] = d2;
                                        }
                                    far2 = d2;
#pragma unroll
                                    for (int i = 0; i < KMAX; ++i)
                                        if (i < K && buf[i] > far2) {
                                            far2 = buf[i];
                                            far_ind = i;
                                        }
                                }
                            }
                        }
                    }
                }
            }
            if (kid >= K) break;
        }
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
            if (i < K) smp_pidx[s * K + i] = out[i];
        if (pt_flag) {
            // the set of points this call's shading touches (every writer stores the same value)
#pragma unroll
            for (int i = 0; i < KMAX; ++i)
                if (i < K && out[i] >= 0) pt_flag[out[i]] = 1;
        }
        const int nn = min(kid, K);
        smp_valid[s] = nn > 0;
        if (nn > 0) {
            ray_flag[smp_ray[s]] = 1;  // every writer stores the same value
            shard_add(shards, SH_PAIRS, (unsigned long long)nn);
        }
        shard_add(shards, SH_CAND, (unsigned long long)tested);
    }
}

// The same search with the dependent loads batched (kernel sizes <= 3, i.e. every configuration the reference ships).
// k_knn above walks cell by cell and candidate by candidate: brick record -> voxel list bounds -> one candidate at a
// time, ~90 dependent L2 round trips per sample, and a wavefront pays the longest chain of its 64 samples.  Here a
// thread handles one x-slab (up to 9 cells) at a time: the 9 brick records are loaded together, then the list bounds
// of the occupied cells together, then each occupied cell's candidates twelve at a time, cells visited in the same
// order and candidates tested in the same order as the reference, so the lists are identical (slot order included).
// (occupancy is what this latency-bound kernel lives on: batches of 12 candidates at 133 VGPRs = 3 waves per SIMD ran
// 0.91 ms on cfg 1; batches of 4 with the allocation capped for 6 waves per SIMD -- a few spilled registers included --
// 0.68 ms; slab records fetched three cells at a time for 8 waves: 0.72)
// WGT (the pair kernel on dense units follows, K = 11..15): the search also leaves the sample's normalised
// inverse-distance weights (studio_model.py:285-286,467-475) -- it holds the K squared distances in registers; a separate
// pass (k_pair_weights) re-reads one point row per neighbour slot for them: 7.3 GB of scattered 16-byte reads per frame at
// BASELINE cfg[4].  Same expression, same order, same bits as that pass.
template <int KMAX, bool WGT = false>
__global__ void __launch_bounds__(TPB, KMAX <= 8 ? 6 : 4) k_knn3(GridView g, int K, float radius_limit2,
                                               const float4 *__restrict__ smp_loc, const int *__restrict__ smp_ray,
                                               const int *__restrict__ n_sel, int *__restrict__ smp_pidx,
                                               int *__restrict__ smp_valid, int *__restrict__ ray_flag,
                                               unsigned long long *__restrict__ shards, int *__restrict__ pt_flag,
                                               float *__restrict__ smp_wgt = nullptr)
{
    constexpr int CB = KMAX <= 8 ? 4 : 8;  // candidates fetched per batch
    const int S = n_sel[0];
    for (int64_t s = (int64_t)blockIdx.x * TPB + threadIdx.x; s < S; s += (int64_t)gridDim.x * TPB) {
        const float4 c = smp_loc[s];
        int fx, fy, fz;
        cell_of(g, c.x, c.y, c.z, fx, fy, fz);  // selected samples are inside the grid by construction
        int kid = 0, far_ind = 0;
        float far2 = 0.0f;
        float buf[KMAX];
        int out[KMAX];
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            buf[i] = 0.f;
            out[i] = -1;
        }
        unsigned tested = 0;
        const int nlayers = (g.kernel_size[0] + 1) / 2;  // 1 or 2
        for (int layer = 0; layer < nlayers; ++layer) {
            for (int x = -layer; x <= layer; ++x) {
                const bool in_x = fx + x >= 0 && fx + x < g.dims[0];
                // ---- level 1: brick records of the slab's cells (slot = 3 (y + 1) + (z + 1)) -----------------------
                unsigned long long bits[9];
                unsigned rank[9];
                unsigned act = 0;
#pragma unroll
                for (int sl = 0; sl < 9; ++sl) {
                    const int y = sl / 3 - 1, z = sl % 3 - 1;
                    const bool on = in_x && max(abs(z), max(abs(x), abs(y))) == layer && fy + y >= 0 &&
                                    fy + y < g.dims[1] && fz + z >= 0 && fz + z < g.dims[2];
                    bits[sl] = 0;
                    rank[sl] = 0;
                    if (on) {
                        int brick, bit;
                        brick_of(g, fx + x, fy + y, fz + z, brick, bit);
                        const BrickRec rec = g.rec[brick];
                        const unsigned long long mbit = 1ull << bit;
                        if (rec.bits & mbit) {
                            act |= 1u << sl;
                            bits[sl] = rec.bits & (mbit - 1ull);
                            rank[sl] = rec.rank;
                        }
                    }
                }
                // ---- level 2: list bounds of the occupied cells ----------------------------------------------------
                int vs[9], ve[9];
#pragma unroll
                for (int sl = 0; sl < 9; ++sl) {
                    vs[sl] = 0;
                    ve[sl] = 0;
                    if (act & (1u << sl)) {
                        const int v = (int)rank[sl] + __popcll(bits[sl]);
                        vs[sl] = g.vox_start[v];
                        ve[sl] = g.vox_start[v + 1];
                    }
                }
                // ---- level 3: occupied cells in slot order, candidates CB at a time --------------------------------
                while (act) {
                    const int sl = __ffs(act) - 1;
                    act &= act - 1;
                    int q0 = 0, q1 = 0;
#pragma unroll
                    for (int i = 0; i < 9; ++i)
                        if (i == sl) {
                            q0 = vs[i];
                            q1 = ve[i];
                        }
                    for (int base = q0; base < q1; base += CB) {
                        float4 p[CB];
#pragma unroll
                        for (int b = 0; b < CB; ++b) p[b] = g.cand[min(base + b, q1 - 1)];
#pragma unroll
                        for (int b = 0; b < CB; ++b) {
                            if (base + b >= q1) continue;
                            const float xv = p[b].x - c.x, yv = p[b].y - c.y, zv = p[b].z - c.z;
                            const float d2 = xv * xv + yv * yv + zv * zv;  // left-to-right, unfused
                            ++tested;
                            if (radius_limit2 == 0.0f || d2 <= radius_limit2) {
                                const int pidx = __float_as_int(p[b].w);
                                if (kid++ < K) {
                                    const int slot = kid - 1;
#pragma unroll
                                    for (int i = 0; i < KMAX; ++i)
                                        if (i == slot) {
                                            out[i] = pidx;
                                            buf[i] = d2;
                                        }
                                    if (d2 > far2) {
                                        far2 = d2;
                                        far_ind = slot;
                                    }
                                } else if (d2 < far2) {
#pragma unroll
                                    for (int i = 0; i < KMAX; ++i)
                                        if (i == far_ind) {
                                            out[i] = pidx;
                                            buf[i] = d2;
                                        }
                                    far2 = d2;
#pragma unroll
                                    for (int i = 0; i < KMAX; ++i)
                                        if (i < K && buf[i] > far2) {
                                            far2 = buf[i];
                                            far_ind = i;
                                        }
                                }
                            }
                        }
                    }
                }
            }
            if (kid >= K) break;
        }
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
            if (i < K) smp_pidx[s * K + i] = out[i];
        if (pt_flag) {
#pragma unroll
            for (int i = 0; i < KMAX; ++i)
                if (i < K && out[i] >= 0) pt_flag[out[i]] = 1;
        }
        if (WGT) {
            // w_k = mask_k / clamp(||p_k - s||, 1e-6), divided by clamp(sum_k w_k, 1e-8): buf[k] IS ||p_k - s||^2 as
            // k_pair_weights forms it (the same subtraction, squares summed left to right, unfused)
            float wsum = 0.f;
#pragma unroll
            for (int i = 0; i < KMAX; ++i)
                if (i < K) {
                    buf[i] = out[i] >= 0 ? 1.0f / fmaxf(sqrtf(buf[i]), 1e-6f) : 0.f;
                    wsum += buf[i];
                }
            const float den = fmaxf(wsum, 1e-8f);
#pragma unroll
            for (int i = 0; i < KMAX; ++i)
                if (i < K) smp_wgt[s * K + i] = buf[i] / den;
        }
        const int nn = min(kid, K);
        smp_valid[s] = nn > 0;
        if (nn > 0) {
            ray_flag[smp_ray[s]] = 1;  // every writer stores the same value
            shard_add(shards, SH_PAIRS, (unsigned long long)nn);
        }
        shard_add(shards, SH_CAND, (unsigned long long)tested);
    }
}

// The same search for SMALL batches (a training step draws 4096 rays: ~9 k samples, 150 waves -- k_knn3 then takes the
// 150 us of one thread's chain of ~30 dependent loads whatever the batch).  Here 32 lanes share a sample, lane l < 27
// owns cell l = 9 (x + 1) + 3 (y + 1) + (z + 1) of the 3 x 3 x 3 neighbourhood: the 27 brick records are loaded at once,
// then the 27 list bounds, then every cell's (at most COOP_P) candidates -- five dependent levels instead of thirty.  The
// candidates then pass through the reference's sequential insertion in the reference's order (layer 0 = the centre
// cell, layer 1 = the others by x, y, z; a list in slot order; the search stops behind layer 0 once K were found): the
// insertion state is replicated over the half wave, a candidate's distance and index are broadcast from its owner, so the
// lists -- and the tested-candidate counts -- are the ones k_knn3 and the oracle produce.
constexpr int COOP_P = 12;   // candidates a lane keeps (the scene's P must not exceed it)
template <int KMAX>
__global__ void __launch_bounds__(TPB) k_knn3_coop(GridView g, int K, float radius_limit2,
                                                   const float4 *__restrict__ smp_loc, const int *__restrict__ smp_ray,
                                                   const int *__restrict__ n_sel, int *__restrict__ smp_pidx,
                                                   int *__restrict__ smp_valid, int *__restrict__ ray_flag,
                                                   unsigned long long *__restrict__ shards, int *__restrict__ pt_flag)
{
    const int S = n_sel[0];
    const int lane = threadIdx.x & 63, half = lane >> 5, l = lane & 31;
    const int hbase = half << 5;                                    // first lane of this half wave
    const int64_t pair0 = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6, npairs = ((int64_t)gridDim.x * TPB) >> 6;
    const int nlayers = (g.kernel_size[0] + 1) / 2;                 // 1 or 2
    const int x = l / 9 - 1, sl9 = l - 9 * (l / 9), y = sl9 / 3 - 1, z = sl9 - 3 * (sl9 / 3) - 1;
    const int my_layer = l < 27 ? max(abs(z), max(abs(x), abs(y))) : 99;
    for (int64_t wp = pair0; 2 * wp < S; wp += npairs) {
        const int64_t s = 2 * wp + half;
        const bool live = s < S;
        const float4 c = smp_loc[live ? s : 0];
        int fx, fy, fz;
        cell_of(g, c.x, c.y, c.z, fx, fy, fz);
        // ---- this lane's cell: brick record -> list bounds -> candidates ----------------------------------------------
        const bool on = live && my_layer < nlayers && fx + x >= 0 && fx + x < g.dims[0] && fy + y >= 0 &&
                        fy + y < g.dims[1] && fz + z >= 0 && fz + z < g.dims[2];
        int q0 = 0, q1 = 0;
        if (on) {
            int brick, bit;
            brick_of(g, fx + x, fy + y, fz + z, brick, bit);
            const BrickRec rec = g.rec[brick];
            const unsigned long long mbit = 1ull << bit;
            if (rec.bits & mbit) {
                const int v = (int)rec.rank + __popcll(rec.bits & (mbit - 1ull));
                q0 = g.vox_start[v];
                q1 = g.vox_start[v + 1];
            }
        }
        const int cnt = min(q1 - q0, COOP_P);
        float dd[COOP_P];
        int pi[COOP_P];
        unsigned pass = 0;
        {
            // all of the lane's candidate loads in flight together (branch-free: a predicated load is its own basic block
            // and is waited for before the next one is issued); record 0 stands in for the ones that do not exist
            float4 pc[COOP_P];
#pragma unroll
            for (int b = 0; b < COOP_P; ++b) pc[b] = g.cand[b < cnt ? q0 + b : 0];
#pragma unroll
            for (int b = 0; b < COOP_P; ++b) {
                const float xv = pc[b].x - c.x, yv = pc[b].y - c.y, zv = pc[b].z - c.z;
                dd[b] = xv * xv + yv * yv + zv * zv;   // left-to-right, unfused
                pi[b] = __float_as_int(pc[b].w);
                if (b < cnt && (radius_limit2 == 0.0f || dd[b] <= radius_limit2)) pass |= 1u << b;
            }
        }
        // ---- the reference's insertion, cell by cell in its order; state replicated over the half wave ---------------
        int kid = 0, far_ind = 0;
        float far2 = 0.0f;
        float buf[KMAX];
        int out[KMAX];
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            buf[i] = 0.f;
            out[i] = -1;
        }
        unsigned tested = 0;
        for (int layer = 0; layer < nlayers; ++layer) {
            for (int cl = 0; cl < 27; ++cl) {
                if ((cl == 13) != (layer == 0)) continue;            // layer 0: the centre cell; layer 1: the 26 others
                const int n_c = __shfl(cnt, hbase + cl, 64);
                const unsigned pm = (unsigned)__shfl((int)pass, hbase + cl, 64);
                tested += (unsigned)n_c;
                if (pm == 0) continue;
#pragma unroll
                for (int b = 0; b < COOP_P; ++b) {
                    if (!(pm & (1u << b))) continue;
                    const float d2 = __shfl(dd[b], hbase + cl, 64);
                    const int pidx = __shfl(pi[b], hbase + cl, 64);
                    if (kid++ < K) {
                        const int slot = kid - 1;
#pragma unroll
                        for (int i = 0; i < KMAX; ++i)
                            if (i == slot) {
                                out[i] = pidx;
                                buf[i] = d2;
                            }
                        if (d2 > far2) {
                            far2 = d2;
                            far_ind = slot;
                        }
                    } else if (d2 < far2) {
#pragma unroll
                        for (int i = 0; i < KMAX; ++i)
                            if (i == far_ind) {
                                out[i] = pidx;
                                buf[i] = d2;
                            }
                        far2 = d2;
#pragma unroll
                        for (int i = 0; i < KMAX; ++i)
                            if (i < K && buf[i] > far2) {
                                far2 = buf[i];
                                far_ind = i;
                            }
                    }
                }
            }
            if (kid >= K) break;
        }
        if (!live) continue;
        // lane i < K of the half wave writes neighbour i
        int mine = -1;
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
            if (i == l) mine = out[i];
        if (l < K) {
            smp_pidx[s * K + l] = mine;
            if (pt_flag && mine >= 0) pt_flag[mine] = 1;
        }
        if (l == 0) {
            const int nn = min(kid, K);
            smp_valid[s] = nn > 0;
            if (nn > 0) {
                ray_flag[smp_ray[s]] = 1;  // every writer stores the same value
                shard_add(shards, SH_PAIRS, (unsigned long long)nn);
            }
            shard_add(shards, SH_CAND, (unsigned long long)tested);
        }
    }
}

// vs_list[voff[s]] = s for samples with >= 1 neighbour; publishes S_valid
__global__ void __launch_bounds__(TPB) k_compact_valid(const int *__restrict__ smp_valid,
                                                        const int *__restrict__ smp_voff, int *__restrict__ n_sel,
                                                        int *__restrict__ vs_list, int64_t *__restrict__ counters,
                                                        const unsigned long long *__restrict__ acc)
{
    const int S = n_sel[0];
    int64_t s0 = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (s0 == 0) {
        n_sel[1] = smp_voff[S];
        counters[PNR_CNT_SAMPLES_VALID] = smp_voff[S];
        counters[PNR_CNT_RAYS_HIT] = (int64_t)shard_sum(acc, SH_RAYS_HIT);
        counters[PNR_CNT_PAIRS_VALID] = (int64_t)shard_sum(acc, SH_PAIRS);
        counters[PNR_CNT_CANDIDATES] = (int64_t)shard_sum(acc, SH_CAND);
    }
    for (int64_t s = s0; s < S; s += (int64_t)gridDim.x * TPB)
        if (smp_valid[s]) vs_list[smp_voff[s]] = (int)s;
}

// ------------------------------------------------------------------------------------------------
// compat scatter: the reference's [R'',SR,K] / [R'',SR,3] / [R] int8 outputs (cu:425-432)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) k_scatter_compat(int64_t R, int SR, int K, const int *__restrict__ ray_cnt,
                                                         const int *__restrict__ ray_off,
                                                         const int *__restrict__ ray_flag,
                                                         const int *__restrict__ ray_rank,
                                                         const float4 *__restrict__ smp_loc,
                                                         const int *__restrict__ smp_pidx, int *__restrict__ out_pidx,
                                                         float *__restrict__ out_loc, int8_t *__restrict__ out_mask,
                                                         int64_t *__restrict__ counters)
{
    // one wavefront per ray; lanes stride over the SR*K (and SR*3) outputs of the ray's row
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (r == 0 && lane == 0) counters[PNR_CNT_RAYS_KEPT] = ray_rank[R];
    if (r >= R) return;
    const int keep = ray_flag[r];
    if (lane == 0) out_mask[r] = (int8_t)(keep ? 1 : 0);
    if (!keep) return;
    const int64_t row = ray_rank[r];
    const int cnt = ray_cnt[r], off = ray_off[r];
    for (int i = lane; i < SR * K; i += 64) {
        int slot = i / K;
        out_pidx[row * SR * K + i] = slot < cnt ? smp_pidx[((int64_t)off + slot) * K + (i - slot * K)] : -1;
    }
    for (int i = lane; i < SR * 3; i += 64) {
        int slot = i / 3, c = i - slot * 3;
        float v = 0.f;
        if (slot < cnt) {
            float4 p = smp_loc[(int64_t)off + slot];
            v = c == 0 ? p.x : (c == 1 ? p.y : p.z);
        }
        out_loc[row * SR * 3 + i] = v;
    }
}

static inline unsigned nblk(int64_t n, int per = TPB) { return (unsigned)std::max<int64_t>(1, (n + per - 1) / per); }

// accumulators (unsigned long long) live behind n_sel: [8..13] as 64-bit words
static inline unsigned long long *acc_ptr(RenderWs &ws) { return ws.shards; }

// ------------------------------------------------------------------------------------------------
// the clears of a call in ONE launch (they were six hipMemsetAsync: six dispatches in front of a 0.4-ms frame)
// ------------------------------------------------------------------------------------------------
struct InitArgs {
    int *n_sel;                   // [64]
    unsigned long long *shards;   // [SH_COUNT * SHARDS * SHARD_STRIDE]
    int64_t *counters;            // [PNR_NUM_COUNTERS] (the caller's: 8-byte aligned only)
    int *ray_flag;                // [R + 1]
    int64_t n_ray_flag;
    int4 *pt_flag4;               // [ceil(N / 4)] or null (the region is padded to 256 bytes)
    int64_t n_pt4;
    float4 *smp_out;              // [cap] or null
    int64_t n_out;
};
__global__ void __launch_bounds__(TPB) k_render_init(InitArgs a)
{
    const int64_t t0 = (int64_t)blockIdx.x * TPB + threadIdx.x, nt = (int64_t)gridDim.x * TPB;
    constexpr int64_t SHARD_INTS = (int64_t)SH_COUNT * SHARDS * SHARD_STRIDE * 2;
    for (int64_t i = t0; i < 64; i += nt) a.n_sel[i] = 0;
    for (int64_t i = t0; i < SHARD_INTS; i += nt) reinterpret_cast<int *>(a.shards)[i] = 0;
    for (int64_t i = t0; i < PNR_NUM_COUNTERS; i += nt) a.counters[i] = 0;
    for (int64_t i = t0; i < a.n_ray_flag; i += nt) a.ray_flag[i] = 0;
    const int4 z4 = make_int4(0, 0, 0, 0);
    for (int64_t i = t0; i < a.n_pt4; i += nt) a.pt_flag4[i] = z4;
    const float4 zf = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = t0; i < a.n_out; i += nt) a.smp_out[i] = zf;
}

__global__ void __launch_bounds__(TPB) k_zero(void *p, size_t bytes, int mode)
{
    const size_t t0 = (size_t)blockIdx.x * TPB + threadIdx.x, nt = (size_t)gridDim.x * TPB;
    if (mode == 16) {
        const int4 z = make_int4(0, 0, 0, 0);
        for (size_t i = t0; i < bytes / 16; i += nt) reinterpret_cast<int4 *>(p)[i] = z;
    } else if (mode == 4) {
        for (size_t i = t0; i < bytes / 4; i += nt) reinterpret_cast<int *>(p)[i] = 0;
    } else {
        for (size_t i = t0; i < bytes; i += nt) reinterpret_cast<unsigned char *>(p)[i] = 0;
    }
}

int zero_async(void *p, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return PNR_OK;
    const size_t both = (size_t)(uintptr_t)p | bytes;
    const int mode = (both & 15) == 0 ? 16 : ((both & 3) == 0 ? 4 : 1);
    const size_t units = bytes / (size_t)mode;
    const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>(1, (units + TPB - 1) / TPB), 256 * 16);
    hipLaunchKernelGGL(k_zero, dim3(grid), dim3(TPB), 0, stream, p, bytes, mode);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

int launch_select_expand(const GridView &g, const CamRef &cr, const float *d_dirs, const float *d_raypos,
                         int64_t R, int D, int SR, int64_t cap, RenderWs &ws, int64_t *d_counters,
                         hipStream_t stream, bool render_clears, int64_t N)
{
    InitArgs ia{};
    ia.n_sel = ws.n_sel;
    ia.shards = ws.shards;
    ia.counters = d_counters;
    ia.ray_flag = ws.ray_flag;
    ia.n_ray_flag = R + 1;
    ws.pt_flag_cleared = ws.out_cleared = false;
    if (render_clears) {
        if (N > 0 && ws.pt_flag) {
            ia.pt_flag4 = reinterpret_cast<int4 *>(ws.pt_flag);
            ia.n_pt4 = (N + 3) / 4;
            ws.pt_flag_cleared = true;
        }
        ia.smp_out = ws.smp_out;
        ia.n_out = cap;
        ws.out_cleared = true;
    }
    {
        const int64_t widest = std::max<int64_t>(std::max<int64_t>(ia.n_ray_flag, ia.n_pt4), std::max<int64_t>(ia.n_out, 4096));
        hipLaunchKernelGGL(k_render_init, dim3((unsigned)std::min<int64_t>(nblk(widest), 256 * 16)), dim3(TPB), 0, stream, ia);
    }
    unsigned long long *acc = acc_ptr(ws);
    // small batches: 4 rays per wavefront instead of 16 (a wavefront walks its rays one after the other, each a round of
    // dependent occupancy probes: at 4096 rays 256 wavefronts took 68 us for the selection and 34 for the expansion)
    const bool few = R <= 16384;
    if (few)
        hipLaunchKernelGGL(k_select<4>, dim3(nblk((R + 3) / 4, TPB / 64)), dim3(TPB), 0, stream, g, cr, d_dirs, d_raypos, R, D,
                           SR, ws.ray_cnt, ws.ray_bits, acc);
    else
        hipLaunchKernelGGL(k_select<RPW>, dim3(nblk((R + RPW - 1) / RPW, TPB / 64)), dim3(TPB), 0, stream, g, cr, d_dirs,
                           d_raypos, R, D, SR, ws.ray_cnt, ws.ray_bits, acc);
    int rc = scan_exclusive_i32(ws.ray_cnt, ws.ray_off, R, nullptr, nullptr, ws.scan_temp, stream);
    if (rc != PNR_OK) return rc;
    if (few)
        hipLaunchKernelGGL(k_expand<4>, dim3(nblk((R + 3) / 4, TPB / 64)), dim3(TPB), 0, stream, cr, d_dirs, d_raypos, R, D, SR,
                           ws.ray_off, ws.ray_bits, cap, ws.smp_loc, ws.smp_ray, ws.n_sel, d_counters, ws.ray_dirs);
    else
        hipLaunchKernelGGL(k_expand<RPW>, dim3(nblk((R + RPW - 1) / RPW, TPB / 64)), dim3(TPB), 0, stream, cr, d_dirs, d_raypos,
                           R, D, SR, ws.ray_off, ws.ray_bits, cap, ws.smp_loc, ws.smp_ray, ws.n_sel, d_counters, ws.ray_dirs);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

// vs_list-style compaction of the flagged points; publishes U
__global__ void __launch_bounds__(TPB) k_list_points(int64_t N, const int *__restrict__ pt_flag,
                                                      const int *__restrict__ pt_rank, int *__restrict__ pt_list,
                                                      int *__restrict__ n_sel, int64_t *__restrict__ counters)
{
    const int64_t p0 = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (p0 == 0) {
        n_sel[3] = pt_rank[N];
        counters[PNR_CNT_POINTS_UNIQUE] = pt_rank[N];
    }
    for (int64_t p = p0; p < N; p += (int64_t)gridDim.x * TPB)
        if (pt_flag[p]) pt_list[pt_rank[p]] = (int)p;
}

int launch_knn(const GridView &g, int K, float radius_limit, RenderWs &ws, int64_t cap, int64_t *d_counters,
               hipStream_t stream, int64_t N, int64_t R, int P)
{
    unsigned long long *acc = acc_ptr(ws);
    ws.wgt_from_knn = false;
    int *pt_flag = (N > 0) ? ws.pt_flag : nullptr;
    if (pt_flag && !ws.pt_flag_cleared) {
        const int rcz = zero_async(pt_flag, ((size_t)N * sizeof(int) + 15) / 16 * 16, stream);   // (the region is padded)
        if (rcz != PNR_OK) return rcz;
    }
    const float r2 = radius_limit * radius_limit;  // fp32, as cu:410
    // grid-stride over the device-side sample count; enough workgroups to fill the chip
    const unsigned grid = (unsigned)std::min<int64_t>(nblk(cap), 256 * 32);
    const bool batched = g.kernel_size[0] <= 3;
    // small batches (R: the call's rays, 0 = unknown): 32 lanes per sample, see k_knn3_coop
    static const int coop_max_rays = [] {
        const char *e = getenv("PNR_KNN_COOP_MAX_RAYS");
        return e ? atoi(e) : 12288;   // (measured at cfg 1: 53 / 85 / 138 us at 4096 / 8192 / 16 384 rays against k_knn3's 140-157)
    }();
    // (PNR_WGT_FROM_KNN=0 in the environment: the separate k_pair_weights pass, for A/B runs and the equality test)
    static const bool wgt_in_search = [] {
        const char *e = getenv("PNR_WGT_FROM_KNN");
        return !(e && atoi(e) == 0);
    }();
    const bool coop = batched && g.kernel_size[1] <= 3 && g.kernel_size[2] <= 3 && K <= 16 && P >= 1 && P <= COOP_P &&
                      R >= 1 && R <= coop_max_rays;
    if (coop) {
        const unsigned cgrid = (unsigned)std::min<int64_t>((cap + 7) / 8, 256 * 32);   // 8 samples per workgroup pass
        if (K <= 8)
            hipLaunchKernelGGL(k_knn3_coop<8>, dim3(cgrid), dim3(TPB), 0, stream, g, K, r2, ws.smp_loc, ws.smp_ray, ws.n_sel,
                               ws.smp_pidx, ws.smp_valid, ws.ray_flag, acc, pt_flag);
        else
            hipLaunchKernelGGL(k_knn3_coop<16>, dim3(cgrid), dim3(TPB), 0, stream, g, K, r2, ws.smp_loc, ws.smp_ray, ws.n_sel,
                               ws.smp_pidx, ws.smp_valid, ws.ray_flag, acc, pt_flag);
    } else if (batched && K <= 8)
        hipLaunchKernelGGL(k_knn3<8>, dim3(grid), dim3(TPB), 0, stream, g, K, r2, ws.smp_loc, ws.smp_ray, ws.n_sel,
                           ws.smp_pidx, ws.smp_valid, ws.ray_flag, acc, pt_flag, (float *)nullptr);
    else if (batched && K >= 11 && K <= 15 && ws.smp_wgt && wgt_in_search) {
        // (the K for which the fp32 pair kernel runs on dense units and reads per-slot weights: launch_shade)
        hipLaunchKernelGGL((k_knn3<16, true>), dim3(grid), dim3(TPB), 0, stream, g, K, r2, ws.smp_loc, ws.smp_ray, ws.n_sel,
                           ws.smp_pidx, ws.smp_valid, ws.ray_flag, acc, pt_flag, ws.smp_wgt);
        ws.wgt_from_knn = true;
    } else if (batched && K <= 16)
        hipLaunchKernelGGL(k_knn3<16>, dim3(grid), dim3(TPB), 0, stream, g, K, r2, ws.smp_loc, ws.smp_ray, ws.n_sel,
                           ws.smp_pidx, ws.smp_valid, ws.ray_flag, acc, pt_flag, (float *)nullptr);
    else if (K <= 8)
        hipLaunchKernelGGL(k_knn<8>, dim3(grid), dim3(TPB), 0, stream, g, K, r2, ws.smp_loc, ws.smp_ray, ws.n_sel,
                           ws.smp_pidx, ws.smp_valid, ws.ray_flag, acc, pt_flag);
    else if (K <= 16)
        hipLaunchKernelGGL(k_knn<16>, dim3(grid), dim3(TPB), 0, stream, g, K, r2, ws.smp_loc, ws.smp_ray, ws.n_sel,
                           ws.smp_pidx, ws.smp_valid, ws.ray_flag, acc, pt_flag);
    else
        hipLaunchKernelGGL(k_knn<32>, dim3(grid), dim3(TPB), 0, stream, g, K, r2, ws.smp_loc, ws.smp_ray, ws.n_sel,
                           ws.smp_pidx, ws.smp_valid, ws.ray_flag, acc, pt_flag);
    int rc = scan_exclusive_i32(ws.smp_valid, ws.smp_voff, cap, ws.n_sel, nullptr, ws.scan_temp, stream);
    if (rc != PNR_OK) return rc;
    hipLaunchKernelGGL(k_compact_valid, dim3(grid), dim3(TPB), 0, stream, ws.smp_valid, ws.smp_voff, ws.n_sel,
                       ws.vs_list, d_counters, acc);
    if (pt_flag) {
        rc = scan_exclusive_i32(ws.pt_flag, ws.pt_rank, N, nullptr, nullptr, ws.pt_scan_temp, stream);
        if (rc != PNR_OK) return rc;
        hipLaunchKernelGGL(k_list_points, dim3((unsigned)std::min<int64_t>(nblk(N), 256 * 32)), dim3(TPB), 0, stream, N,
                           ws.pt_flag, ws.pt_rank, ws.pt_list, ws.n_sel, d_counters);
    }
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

}  // namespace pnr

using namespace pnr;

extern "C" size_t pnr_query_workspace_bytes(int64_t R, int32_t D, int32_t SR, int32_t K)
{
    (void)D;
    if (R < 1) R = 1;
    RenderWs ws = carve_render_ws(nullptr, R, R * (int64_t)SR, K);
    // the shade-only tail (sigma, agg) is not needed by the query
    return (size_t)(uintptr_t)ws.smp_sigma;
}

extern "C" int pnr_query_raypos(const pnr_scene_t *scene, const float *d_raypos, int64_t R, int32_t D, int32_t SR,
                                int32_t K, float radius_limit, int32_t *d_sample_pidx, float *d_sample_loc,
                                int8_t *d_ray_mask, int64_t *d_counters, void *d_workspace, size_t workspace_bytes,
                                void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    PNR_REQUIRE(scene && d_raypos && d_sample_pidx && d_sample_loc && d_ray_mask && d_counters && d_workspace,
                "pnr_query_raypos: null argument");
    if (!scene->built) {
        set_error("pnr_query_raypos: scene not built");
        return PNR_ERR_STATE;
    }
    PNR_REQUIRE(R >= 1 && R * (int64_t)SR < (int64_t)0x7FFFFFF0, "pnr_query_raypos: R=%lld out of range", (long long)R);
    PNR_REQUIRE(D >= 1 && D <= PNR_MAX_D, "pnr_query_raypos: D=%d not in [1,%d]", D, PNR_MAX_D);
    PNR_REQUIRE(K >= 1 && K <= PNR_MAX_K, "pnr_query_raypos: K=%d not in [1,%d]", K, PNR_MAX_K);
    PNR_REQUIRE(SR >= 1, "pnr_query_raypos: SR=%d", SR);
    if (workspace_bytes < pnr_query_workspace_bytes(R, D, SR, K)) {
        set_error("pnr_query_raypos: workspace of %zu bytes < %zu required", workspace_bytes,
                  pnr_query_workspace_bytes(R, D, SR, K));
        return PNR_ERR_WORKSPACE;
    }
    const int64_t cap = R * (int64_t)SR;
    RenderWs ws = carve_render_ws(d_workspace, R, cap, K);
    ws.smp_wgt = nullptr;   // (shade-only: behind the query workspace's end -- the search must not write weights there)
    CamRef cr{};  // explicit positions: no camera involved
    cr.D = D;
    int rc = launch_select_expand(scene->grid, cr, nullptr, d_raypos, R, D, SR, cap, ws, d_counters, stream);
    if (rc != PNR_OK) return rc;
    rc = launch_knn(scene->grid, K, radius_limit, ws, cap, d_counters, stream, 0, R, scene->params.P);
    if (rc != PNR_OK) return rc;
    // rank of every kept ray = exclusive scan of the keep flags (reuses smp_voff as scratch: [R+1] <= [cap+1])
    int *ray_rank = ws.smp_voff;
    rc = scan_exclusive_i32(ws.ray_flag, ray_rank, R, nullptr, nullptr, ws.scan_temp, stream);
    if (rc != PNR_OK) return rc;
    hipLaunchKernelGGL(k_scatter_compat, dim3(nblk(R, TPB / 64)), dim3(TPB), 0, stream, R, SR, K, ws.ray_cnt,
                       ws.ray_off, ws.ray_flag, ray_rank, ws.smp_loc, ws.smp_pidx, d_sample_pidx, d_sample_loc,
                       d_ray_mask, d_counters);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}
