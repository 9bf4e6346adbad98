// Scene build: the persistent voxel structure + the packed point table.
//
// Replaces claim_occ / map_coor2occ / fill_occ2pnts (query_worldcoords.cu:18-162), which the reference
// re-runs over all N points for EVERY 2304-ray chunk (cu:314-365, ~130 MB of fills per call), by one
// build per point-cloud version.  Semantics are the canonical sequential ones of SURVEY.md 8a-note:
// per-voxel lists in ascending point index, first P kept; the voxel of the first in-grid point is
// emptied when compat_drop_voxel0 (the reference's `voxel_idx > 0`, cu:147); all occupied voxels are
// kept even beyond max_o (flagged in info[1]).  Every step is order-independent (bit-OR, integer
// counts, per-voxel selection of the P smallest indices), so the structure is bitwise deterministic.
#include <algorithm>

#include "pnr_internal.h"

namespace pnr {

constexpr int TPB = 256;

// ---- B1: cell of every point, raw occupancy bits, first in-grid point -------------------------------
// old_index (pnr_scene_update on an unchanged grid): point i of the new cloud was point old_index[i] of the previous
// one (-1: added) -- a surviving point keeps its cell code, only added points pay the three fp32 divisions.
__global__ void __launch_bounds__(TPB) k_point_cells(const float *__restrict__ xyz, int64_t N, GridView g,
                                                      const int *__restrict__ old_index,
                                                      const uint32_t *__restrict__ old_cell, int64_t N_old,
                                                      uint32_t *__restrict__ pt_cell,
                                                      unsigned long long *__restrict__ occ_all,
                                                      int *__restrict__ first_valid,
                                                      unsigned long long *__restrict__ n_inside,
                                                      unsigned long long *__restrict__ n_reused)
{
    int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= N) return;
    uint32_t code = 0xFFFFFFFFu;
    const int oi = old_index ? old_index[i] : -1;
    if (oi >= 0 && oi < N_old) {
        code = old_cell[oi];
        atomicAdd(n_reused, 1ull);
    } else {
        int cx, cy, cz;
        if (cell_of(g, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], cx, cy, cz)) {
            int brick, bit;
            brick_of(g, cx, cy, cz, brick, bit);
            code = ((uint32_t)brick << 6) | (uint32_t)bit;
        }
    }
    if (code != 0xFFFFFFFFu) {
        atomicOr(&occ_all[code >> 6], 1ull << (code & 63));
        atomicMin(first_valid, (int)i);
        atomicAdd(n_inside, 1ull);
    }
    pt_cell[i] = code;
}

// ---- B2: occupancy used for point lookup = raw occupancy minus the compat-dropped voxel --------------
__global__ void k_drop_voxel0(const uint32_t *__restrict__ pt_cell, const int *__restrict__ first_valid,
                              int64_t N, int compat, unsigned long long *__restrict__ occ_pts,
                              long long *__restrict__ dropped_code)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int fv = *first_valid;
    *dropped_code = -1;
    if (compat && fv >= 0 && fv < N) {
        uint32_t code = pt_cell[fv];
        occ_pts[code >> 6] &= ~(1ull << (code & 63));
        *dropped_code = code;
    }
}

// ---- B3: dilation of the raw occupancy over [c - k/2, c + (k+1)/2) per axis (cu:101-110) --------------
__global__ void __launch_bounds__(TPB) k_dilate(const unsigned long long *__restrict__ occ_all, GridView g,
                                                 int q0, int q1, int q2, unsigned long long *__restrict__ occ_dil)
{
    int b = blockIdx.x * TPB + threadIdx.x;
    if (b >= g.nbricks) return;
    unsigned long long bits = occ_all[b];
    if (!bits) return;
    int bz = b % g.bdims[2];
    int by = (b / g.bdims[2]) % g.bdims[1];
    int bx = b / (g.bdims[2] * g.bdims[1]);
    while (bits) {
        int bit = __builtin_ctzll(bits);
        bits &= bits - 1;
        int cx = bx * 4 + (bit >> 4), cy = by * 4 + ((bit >> 2) & 3), cz = bz * 4 + (bit & 3);
        for (int x = max(0, cx - q0 / 2); x < min(g.dims[0], cx + (q0 + 1) / 2); ++x)
            for (int y = max(0, cy - q1 / 2); y < min(g.dims[1], cy + (q1 + 1) / 2); ++y)
                for (int z = max(0, cz - q2 / 2); z < min(g.dims[2], cz + (q2 + 1) / 2); ++z) {
                    int nb, nbit;
                    brick_of(g, x, y, z, nb, nbit);
                    atomicOr(&occ_dil[nb], 1ull << nbit);
                }
    }
}

// ---- B4: per-brick popcount (scanned into ranks) and record assembly ----------------------------------
__global__ void __launch_bounds__(TPB) k_brick_popc(const unsigned long long *__restrict__ occ_pts, int nbricks,
                                                     int *__restrict__ popc)
{
    int b = blockIdx.x * TPB + threadIdx.x;
    if (b < nbricks) popc[b] = __popcll(occ_pts[b]);
}

__global__ void __launch_bounds__(TPB) k_make_recs(const unsigned long long *__restrict__ occ_pts,
                                                    const int *__restrict__ rank, int nbricks,
                                                    BrickRec *__restrict__ rec)
{
    int b = blockIdx.x * TPB + threadIdx.x;
    if (b >= nbricks) return;
    BrickRec r;
    r.bits = occ_pts[b];
    r.rank = (uint32_t)rank[b];
    r.pad = 0;
    rec[b] = r;
}

__device__ __forceinline__ int voxel_of_code(const BrickRec *__restrict__ rec, uint32_t code)
{
    if (code == 0xFFFFFFFFu) return -1;
    BrickRec r = rec[code >> 6];
    unsigned long long m = 1ull << (code & 63);
    if (!(r.bits & m)) return -1;
    return (int)r.rank + __popcll(r.bits & (m - 1));
}

// ---- B5: points per voxel ---------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) k_count_points(const uint32_t *__restrict__ pt_cell, int64_t N,
                                                       const BrickRec *__restrict__ rec, int *__restrict__ cnt)
{
    int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= N) return;
    int v = voxel_of_code(rec, pt_cell[i]);
    if (v >= 0) atomicAdd(&cnt[v], 1);
}

__global__ void __launch_bounds__(TPB) k_cap_counts(const int *__restrict__ cnt, int nvox, int P,
                                                     int *__restrict__ capped)
{
    int v = blockIdx.x * TPB + threadIdx.x;
    if (v < nvox) capped[v] = min(cnt[v], P);
}

// ---- B6: unordered fill of the full lists, then per-voxel selection of the P smallest indices ----------
__global__ void __launch_bounds__(TPB) k_fill_lists(const uint32_t *__restrict__ pt_cell, int64_t N,
                                                     const BrickRec *__restrict__ rec,
                                                     const int *__restrict__ full_start, int *__restrict__ cursor,
                                                     int *__restrict__ full_list)
{
    int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= N) return;
    int v = voxel_of_code(rec, pt_cell[i]);
    if (v < 0) return;
    int slot = atomicAdd(&cursor[v], 1);
    full_list[full_start[v] + slot] = (int)i;
}

// one thread per voxel: repeatedly extract the smallest index larger than the previous one (P passes over
// the voxel's list).  Lists are short (a few to a few hundred entries) and this runs once per scene.
__global__ void __launch_bounds__(TPB) k_select_first_p(const int *__restrict__ full_start,
                                                         const int *__restrict__ full_list,
                                                         const int *__restrict__ vox_start, int nvox,
                                                         const float *__restrict__ xyz, float4 *__restrict__ cand)
{
    int v = blockIdx.x * TPB + threadIdx.x;
    if (v >= nvox) return;
    const int fs = full_start[v], fe = full_start[v + 1];
    const int os = vox_start[v], oe = vox_start[v + 1];
    int prev = -1;
    for (int o = os; o < oe; ++o) {
        int best = 0x7FFFFFFF;
        for (int j = fs; j < fe; ++j) {
            int p = full_list[j];
            if (p > prev && p < best) best = p;
        }
        prev = best;
        float4 c;
        c.x = xyz[3 * (int64_t)best];
        c.y = xyz[3 * (int64_t)best + 1];
        c.z = xyz[3 * (int64_t)best + 2];
        c.w = __int_as_float(best);
        cand[o] = c;
    }
}

// ---- packed point rows --------------------------------------------------------------------------------
// row = [x y z conf | color[3] dir[3] 0 0 | 0 0 0 0 | emb[0:32]]  (48 floats, 192 bytes, 16-byte aligned pieces)
__global__ void __launch_bounds__(TPB) k_pack_points(const float *__restrict__ xyz, const float *__restrict__ emb,
                                                      const float *__restrict__ conf, const float *__restrict__ dir,
                                                      const float *__restrict__ color, int64_t N,
                                                      float *__restrict__ rows)
{
    // one thread per (point, float4 chunk): 12 chunks per 192-byte row, coalesced 16-byte stores.  The 48 bytes a
    // (sample, neighbour) pair needs in bf16x3 mode (position + conf, colour, direction) lead the row inside one
    // 64-byte segment, which never straddles a 128-byte line at this stride; the embedding (read per pair only in
    // fp32 mode, per distinct point by k_point_part) follows.
    int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x;
    int64_t i = t / 12;
    int c = (int)(t - i * 12);
    if (i >= N) return;
    float4 v;
    if (c == 0) {
        v = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], conf ? conf[i] : 1.0f);
    } else if (c == 1) {
        v = make_float4(color[3 * i], color[3 * i + 1], color[3 * i + 2], dir[3 * i]);
    } else if (c == 2) {
        v = make_float4(dir[3 * i + 1], dir[3 * i + 2], 0.f, 0.f);
    } else if (c == 3) {
        v = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const float *e = emb + i * PNR_FEAT_DIM + (c - 4) * 4;
        v = make_float4(e[0], e[1], e[2], e[3]);
    }
    reinterpret_cast<float4 *>(rows)[i * 12 + c] = v;
}

// the same for a LIST of rows: one thread per (list entry, float4 chunk); the number of entries may live on the device
__global__ void __launch_bounds__(TPB) k_pack_point_rows(const float *__restrict__ xyz, const float *__restrict__ emb,
                                                          const float *__restrict__ conf, const float *__restrict__ dir,
                                                          const float *__restrict__ color, int64_t N,
                                                          const int *__restrict__ index, int64_t n_index,
                                                          const long long *__restrict__ n_dev,
                                                          const int *__restrict__ n_dev32, float *__restrict__ rows)
{
    int64_t n = n_index;
    if (n_dev) n = min((int64_t)*n_dev, n_index);
    if (n_dev32) n = min((int64_t)*n_dev32, n_index);
    for (int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x; t < n * 12; t += (int64_t)gridDim.x * TPB) {
        const int64_t e = t / 12;
        const int c = (int)(t - e * 12);
        const int64_t i = index[e];
        if (i < 0 || i >= N) continue;
        float4 v;
        if (c == 0) {
            v = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], conf ? conf[i] : 1.0f);
        } else if (c == 1) {
            v = make_float4(color[3 * i], color[3 * i + 1], color[3 * i + 2], dir[3 * i]);
        } else if (c == 2) {
            v = make_float4(dir[3 * i + 1], dir[3 * i + 2], 0.f, 0.f);
        } else if (c == 3) {
            v = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const float *q = emb + i * PNR_FEAT_DIM + (c - 4) * 4;
            v = make_float4(q[0], q[1], q[2], q[3]);
        }
        reinterpret_cast<float4 *>(rows)[i * 12 + c] = v;
    }
}

static inline unsigned nblk(int64_t n) { return (unsigned)((n + TPB - 1) / TPB); }

// grow-only device buffer: reallocated (with 1/8 headroom) when the request exceeds its capacity
template <typename T>
static int ensure(DevBuf<T> &b, size_t count, size_t *bytes)
{
    if (count < 1) count = 1;
    if (b.cap >= count) return PNR_OK;
    if (b.p) {
        (void)hipFree(b.p);
        if (bytes) *bytes -= b.cap * sizeof(T);
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t cap = count + count / 8 + 16;
    PNR_HIP_CHECK(hipMalloc((void **)&b.p, cap * sizeof(T)));
    b.cap = cap;
    if (bytes) *bytes += cap * sizeof(T);
    return PNR_OK;
}

template <typename T>
static void release(DevBuf<T> &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

static void scene_free_grid(pnr_scene *s)
{
    release(s->occ_dil_buf);
    release(s->rec_buf);
    release(s->vox_start_buf);
    release(s->cand_buf);
    release(s->pt_cell[0]);
    release(s->pt_cell[1]);
    release(s->occ_all);
    release(s->occ_pts);
    release(s->popc);
    release(s->cnt);
    release(s->capped);
    release(s->full_start);
    release(s->cursor);
    release(s->full_list);
    release(s->scan_tmp);
    release(s->first_valid);
    release(s->n_inside);
    release(s->dropped);
    s->occ_dil = nullptr;
    s->rec = nullptr;
    s->vox_start = nullptr;
    s->cand = nullptr;
    s->scratch_bytes = 0;
    s->built = false;
}

static bool same_grid(const pnr_grid_params_t &a, const pnr_grid_params_t &b)
{
    for (int i = 0; i < 3; ++i)
        if (a.ranges[i] != b.ranges[i] || a.vox[i] != b.vox[i] || a.dims[i] != b.dims[i]) return false;
    return true;
}

// The build proper.  d_old_index == nullptr: every cell is computed (pnr_scene_build).  Otherwise (pnr_scene_update)
// surviving points reuse their cell code when the grid (origin, voxel size, dims) is unchanged.  All device memory is
// kept in the scene and reused; the content written is a pure function of (xyz, params): identical to a fresh build.
static int scene_build_impl(pnr_scene *scene, const float *d_xyz, int64_t N, const pnr_grid_params_t *p,
                            const int *d_old_index, hipStream_t stream, const char *who)
{
    PNR_REQUIRE(scene && d_xyz && p, "%s: null argument", who);
    PNR_REQUIRE(N > 0 && N < (int64_t)0x7FFFFFFF, "%s: N=%lld out of range", who, (long long)N);
    PNR_REQUIRE(p->P >= 1 && p->P <= 1024, "%s: P=%d out of range", who, p->P);
    for (int a = 0; a < 3; ++a) {
        PNR_REQUIRE(p->dims[a] >= 1, "%s: dims[%d]=%d", who, a, p->dims[a]);
        PNR_REQUIRE(p->vox[a] > 0.f, "%s: vox[%d]=%g", who, a, p->vox[a]);
        PNR_REQUIRE(p->kernel_size[a] >= 1 && p->kernel_size[a] <= 9 && p->query_size[a] >= 1 &&
                        p->query_size[a] <= 9,
                    "%s: kernel/query size out of range", who);
    }
    GridView g{};
    for (int a = 0; a < 3; ++a) {
        g.shift[a] = p->ranges[a];
        g.vox[a] = p->vox[a];
        g.dims[a] = p->dims[a];
        g.bdims[a] = (p->dims[a] + 3) / 4;
        g.kernel_size[a] = p->kernel_size[a];
    }
    const int64_t nbricks64 = (int64_t)g.bdims[0] * g.bdims[1] * g.bdims[2];
    PNR_REQUIRE(nbricks64 < (1ll << 25), "%s: grid of %lld bricks is too large", who, (long long)nbricks64);
    g.nbricks = (int)nbricks64;

    const bool reuse = d_old_index != nullptr && scene->built && same_grid(scene->params, *p);
    const int64_t N_old = scene->N;
    const int prev = scene->cur_cell, cur = reuse ? 1 - prev : prev;
    scene->built = false;   // until this build completes
    size_t *sb = &scene->scratch_bytes;
    int rc = PNR_OK;
#define TRY(x)                         \
    do {                               \
        rc = (x);                      \
        if (rc != PNR_OK) return rc;   \
    } while (0)
    const size_t scan_n = (size_t)std::max<int64_t>(nbricks64, N) + 1;
    TRY(ensure(scene->pt_cell[cur], (size_t)N, sb));
    TRY(ensure(scene->occ_all, (size_t)g.nbricks, sb));
    TRY(ensure(scene->occ_pts, (size_t)g.nbricks, sb));
    TRY(ensure(scene->occ_dil_buf, (size_t)g.nbricks, sb));
    TRY(ensure(scene->rec_buf, (size_t)g.nbricks, sb));
    TRY(ensure(scene->popc, (size_t)g.nbricks + 1, sb));
    TRY(ensure(scene->first_valid, 1, sb));
    TRY(ensure(scene->n_inside, 2, sb));
    TRY(ensure(scene->dropped, 1, sb));
    TRY(ensure(scene->scan_tmp, scan_temp_bytes((int64_t)scan_n), sb));
    uint32_t *pt_cell = scene->pt_cell[cur].p;
    unsigned long long *occ_all = scene->occ_all.p, *occ_pts = scene->occ_pts.p, *occ_dil = scene->occ_dil_buf.p;
    BrickRec *rec = scene->rec_buf.p;
    int *popc = scene->popc.p, *first_valid = scene->first_valid.p;
    unsigned long long *n_inside = scene->n_inside.p;
    long long *dropped = scene->dropped.p;
    void *scan_tmp = scene->scan_tmp.p;

    PNR_HIP_CHECK(hipMemsetAsync(occ_all, 0, sizeof(unsigned long long) * g.nbricks, stream));
    PNR_HIP_CHECK(hipMemsetAsync(occ_dil, 0, sizeof(unsigned long long) * g.nbricks, stream));
    PNR_HIP_CHECK(hipMemsetAsync(first_valid, 0x7F, sizeof(int), stream));
    PNR_HIP_CHECK(hipMemsetAsync(n_inside, 0, 2 * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(k_point_cells, dim3(nblk(N)), dim3(TPB), 0, stream, d_xyz, N, g, reuse ? d_old_index : nullptr,
                       reuse ? scene->pt_cell[prev].p : nullptr, N_old, pt_cell, occ_all, first_valid, n_inside,
                       n_inside + 1);
    PNR_HIP_CHECK(hipMemcpyAsync(occ_pts, occ_all, sizeof(unsigned long long) * g.nbricks, hipMemcpyDeviceToDevice,
                                 stream));
    hipLaunchKernelGGL(k_drop_voxel0, dim3(1), dim3(64), 0, stream, pt_cell, first_valid, N, p->compat_drop_voxel0,
                       occ_pts, dropped);
    hipLaunchKernelGGL(k_dilate, dim3(nblk(g.nbricks)), dim3(TPB), 0, stream, occ_all, g, p->query_size[0],
                       p->query_size[1], p->query_size[2], occ_dil);
    hipLaunchKernelGGL(k_brick_popc, dim3(nblk(g.nbricks)), dim3(TPB), 0, stream, occ_pts, g.nbricks, popc);
    TRY(scan_exclusive_i32(popc, popc, g.nbricks, nullptr, nullptr, scan_tmp, stream));
    hipLaunchKernelGGL(k_make_recs, dim3(nblk(g.nbricks)), dim3(TPB), 0, stream, occ_pts, popc, g.nbricks, rec);
    int h_nvox = 0;
    long long h_dropped = -1;
    unsigned long long h_inside[2] = {0, 0};
    PNR_HIP_CHECK(hipMemcpyAsync(&h_nvox, popc + g.nbricks, sizeof(int), hipMemcpyDeviceToHost, stream));
    PNR_HIP_CHECK(hipMemcpyAsync(&h_dropped, dropped, sizeof(long long), hipMemcpyDeviceToHost, stream));
    PNR_HIP_CHECK(hipMemcpyAsync(h_inside, n_inside, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    PNR_HIP_CHECK(hipStreamSynchronize(stream));
    const int64_t nvox64 = h_nvox;
    g.nvox = (int)nvox64;

    TRY(ensure(scene->cnt, (size_t)nvox64 + 1, sb));
    TRY(ensure(scene->capped, (size_t)nvox64 + 1, sb));
    TRY(ensure(scene->full_start, (size_t)nvox64 + 1, sb));
    TRY(ensure(scene->cursor, (size_t)nvox64 + 1, sb));
    TRY(ensure(scene->vox_start_buf, (size_t)nvox64 + 1, sb));
    int *cnt = scene->cnt.p, *capped = scene->capped.p, *full_start = scene->full_start.p, *cursor = scene->cursor.p,
        *vox_start = scene->vox_start_buf.p;
    PNR_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(int) * (nvox64 + 1), stream));
    PNR_HIP_CHECK(hipMemsetAsync(cursor, 0, sizeof(int) * (nvox64 + 1), stream));
    PNR_HIP_CHECK(hipMemsetAsync(capped, 0, sizeof(int) * (nvox64 + 1), stream));
    hipLaunchKernelGGL(k_count_points, dim3(nblk(N)), dim3(TPB), 0, stream, pt_cell, N, rec, cnt);
    if (nvox64 > 0)
        hipLaunchKernelGGL(k_cap_counts, dim3(nblk(nvox64)), dim3(TPB), 0, stream, cnt, (int)nvox64, p->P, capped);
    TRY(scan_exclusive_i32(cnt, full_start, nvox64, nullptr, nullptr, scan_tmp, stream));
    TRY(scan_exclusive_i32(capped, vox_start, nvox64, nullptr, nullptr, scan_tmp, stream));
    int h_full = 0, h_cap = 0;
    PNR_HIP_CHECK(hipMemcpyAsync(&h_full, full_start + nvox64, sizeof(int), hipMemcpyDeviceToHost, stream));
    PNR_HIP_CHECK(hipMemcpyAsync(&h_cap, vox_start + nvox64, sizeof(int), hipMemcpyDeviceToHost, stream));
    PNR_HIP_CHECK(hipStreamSynchronize(stream));
    const int64_t total_full = h_full, total_capped = h_cap;
    TRY(ensure(scene->full_list, (size_t)total_full, sb));
    TRY(ensure(scene->cand_buf, (size_t)total_capped, sb));
    hipLaunchKernelGGL(k_fill_lists, dim3(nblk(N)), dim3(TPB), 0, stream, pt_cell, N, rec, full_start, cursor,
                       scene->full_list.p);
    if (nvox64 > 0)
        hipLaunchKernelGGL(k_select_first_p, dim3(nblk(nvox64)), dim3(TPB), 0, stream, full_start, scene->full_list.p,
                           vox_start, (int)nvox64, d_xyz, scene->cand_buf.p);
    PNR_HIP_CHECK(hipGetLastError());
    PNR_HIP_CHECK(hipStreamSynchronize(stream));
#undef TRY

    scene->params = *p;
    scene->N = N;
    scene->cur_cell = cur;
    scene->occ_dil = occ_dil;
    scene->rec = rec;
    scene->vox_start = vox_start;
    scene->cand = scene->cand_buf.p;
    g.occ_dil = scene->occ_dil;
    g.rec = scene->rec;
    g.vox_start = scene->vox_start;
    g.cand = scene->cand;
    scene->grid = g;
    scene->built = true;
    // what the render path reads: dilated occupancy, brick records, list bounds, candidates (+ the packed rows)
    scene->bytes = (size_t)g.nbricks * (sizeof(unsigned long long) + sizeof(BrickRec)) +
                   (size_t)(nvox64 + 1) * sizeof(int) + (size_t)total_capped * sizeof(float4) +
                   (scene->point_rows ? (size_t)scene->packed_N * PNR_POINT_ROW_FLOATS * sizeof(float) : 0);
    // occupied voxels of the reference include the compat-dropped one
    scene->info[0] = nvox64 + (h_dropped >= 0 ? 1 : 0);
    scene->info[1] = scene->info[0] > p->max_o;
    scene->info[2] = total_capped;
    scene->info[3] = g.nbricks;
    scene->info[4] = (int64_t)scene->bytes;
    scene->info[5] = N;
    scene->info[6] = (int64_t)h_inside[0];
    scene->info[7] = h_dropped;
    if (d_old_index) {
        ++scene->updates;
        scene->cells_reused = reuse ? (int64_t)h_inside[1] : 0;
    } else {
        ++scene->builds;
        scene->cells_reused = 0;
    }
    return PNR_OK;
}

}  // namespace pnr

using namespace pnr;

extern "C" int pnr_scene_create(pnr_scene_t **out)
{
    PNR_REQUIRE(out != nullptr, "pnr_scene_create: out is null");
    *out = new pnr_scene();
    return PNR_OK;
}

extern "C" int pnr_scene_destroy(pnr_scene_t *scene)
{
    if (!scene) return PNR_OK;
    scene_free_grid(scene);
    if (scene->point_rows) (void)hipFree(scene->point_rows);
    delete scene;
    return PNR_OK;
}

extern "C" int pnr_scene_build(pnr_scene_t *scene, const float *d_xyz, int64_t N, const pnr_grid_params_t *p,
                               void *stream_)
{
    if (scene) {   // tensors bound to the previous cloud say nothing about this one
        for (auto &q : scene->live) q = nullptr;
        scene->live_N = 0;
    }
    return scene_build_impl(scene, d_xyz, N, p, nullptr, (hipStream_t)stream_, "pnr_scene_build");
}

extern "C" int pnr_scene_update(pnr_scene_t *scene, const float *d_xyz, int64_t N, const pnr_grid_params_t *p,
                                const int32_t *d_old_index, void *stream_)
{
    PNR_REQUIRE(scene != nullptr, "pnr_scene_update: null argument");
    if (!scene->built) {
        set_error("pnr_scene_update: scene not built (call pnr_scene_build first)");
        return PNR_ERR_STATE;
    }
    PNR_REQUIRE(d_old_index != nullptr, "pnr_scene_update: d_old_index is null (use pnr_scene_build for a new cloud)");
    for (auto &q : scene->live) q = nullptr;
    scene->live_N = 0;
    const int rc = scene_build_impl(scene, d_xyz, N, p, d_old_index, (hipStream_t)stream_, "pnr_scene_update");
    if (rc == PNR_OK) scene->packed = false;   // the packed rows describe the previous cloud
    return rc;
}

extern "C" int pnr_scene_update_info(const pnr_scene_t *scene, int64_t info[4])
{
    PNR_REQUIRE(scene && info, "pnr_scene_update_info: null argument");
    info[0] = scene->builds;
    info[1] = scene->updates;
    info[2] = scene->cells_reused;
    info[3] = (int64_t)scene->scratch_bytes;
    return PNR_OK;
}

extern "C" int pnr_scene_info(const pnr_scene_t *scene, int64_t info[8])
{
    PNR_REQUIRE(scene && info, "pnr_scene_info: null argument");
    if (!scene->built) {
        set_error("pnr_scene_info: scene not built");
        return PNR_ERR_STATE;
    }
    for (int i = 0; i < 8; ++i) info[i] = scene->info[i];
    info[4] = (int64_t)scene->bytes;
    return PNR_OK;
}

extern "C" int pnr_points_pack(pnr_scene_t *scene, const float *d_xyz, const float *d_embedding,
                               const float *d_conf, const float *d_dir, const float *d_color, int64_t N,
                               void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    PNR_REQUIRE(scene && d_xyz && d_embedding && d_dir && d_color, "pnr_points_pack: null argument");
    PNR_REQUIRE(N > 0 && N < (int64_t)0x7FFFFFFF, "pnr_points_pack: N=%lld out of range", (long long)N);
    if (scene->built && scene->N != N) {
        set_error("pnr_points_pack: N=%lld differs from the built scene (%lld)", (long long)N, (long long)scene->N);
        return PNR_ERR_INVALID;
    }
    if (scene->packed_N != N) {
        if (scene->point_rows) {
            (void)hipFree(scene->point_rows);
            scene->bytes -= (size_t)scene->packed_N * PNR_POINT_ROW_FLOATS * sizeof(float);
            scene->point_rows = nullptr;
        }
        PNR_HIP_CHECK(hipMalloc((void **)&scene->point_rows, (size_t)N * PNR_POINT_ROW_FLOATS * sizeof(float)));
        scene->bytes += (size_t)N * PNR_POINT_ROW_FLOATS * sizeof(float);
        scene->packed_N = N;
    }
    hipLaunchKernelGGL(k_pack_points, dim3(nblk(N * 12)), dim3(TPB), 0, stream, d_xyz, d_embedding, d_conf, d_dir,
                       d_color, N, scene->point_rows);
    PNR_HIP_CHECK(hipGetLastError());
    scene->packed = true;
    return PNR_OK;
}

extern "C" int pnr_points_pack_rows(pnr_scene_t *scene, const float *d_xyz, const float *d_embedding,
                                    const float *d_conf, const float *d_dir, const float *d_color, int64_t N,
                                    const int32_t *d_index, int64_t n_index, const int64_t *d_n_index, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    PNR_REQUIRE(scene && d_xyz && d_embedding && d_dir && d_color && d_index, "pnr_points_pack_rows: null argument");
    if (!scene->packed || scene->packed_N != N) {
        set_error("pnr_points_pack_rows: the scene holds no packed rows of a %lld-point cloud (call pnr_points_pack once)",
                  (long long)N);
        return PNR_ERR_STATE;
    }
    PNR_REQUIRE(n_index >= 0 && n_index < (int64_t)0x7FFFFFFF, "pnr_points_pack_rows: n_index=%lld out of range",
                (long long)n_index);
    if (n_index == 0) return PNR_OK;
    // a grid-stride launch: the list may be much shorter than n_index when its length lives on the device
    const unsigned blocks = (unsigned)std::min<int64_t>((n_index * 12 + TPB - 1) / TPB, 4096);
    hipLaunchKernelGGL(k_pack_point_rows, dim3(blocks), dim3(TPB), 0, stream, d_xyz, d_embedding, d_conf, d_dir, d_color, N,
                       d_index, n_index, reinterpret_cast<const long long *>(d_n_index), (const int *)nullptr,
                       scene->point_rows);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

extern "C" int pnr_points_bind(pnr_scene_t *scene, const float *d_xyz, const float *d_embedding, const float *d_conf,
                               const float *d_dir, const float *d_color, int64_t N)
{
    PNR_REQUIRE(scene != nullptr, "pnr_points_bind: null argument");
    if (!d_xyz && !d_embedding && !d_dir && !d_color) {   // unbind
        for (auto &p : scene->live) p = nullptr;
        scene->live_N = 0;
        return PNR_OK;
    }
    PNR_REQUIRE(d_xyz && d_embedding && d_dir && d_color, "pnr_points_bind: null tensor (pass all null to unbind)");
    if (!scene->packed || scene->packed_N != N) {
        set_error("pnr_points_bind: the scene holds no packed rows of a %lld-point cloud (call pnr_points_pack once)",
                  (long long)N);
        return PNR_ERR_STATE;
    }
    scene->live[0] = d_xyz;
    scene->live[1] = d_embedding;
    scene->live[2] = d_conf;
    scene->live[3] = d_dir;
    scene->live[4] = d_color;
    scene->live_N = N;
    return PNR_OK;
}

namespace pnr {
// bound tensors (pnr_points_bind): the rows of the call's distinct neighbour points are re-packed from them
int launch_refresh_rows(const pnr_scene *scene, const int *pt_list, const int *n_unique, int64_t u_cap, hipStream_t stream)
{
    if (!scene->live[0] || scene->live_N != scene->N || u_cap < 1) return PNR_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((u_cap * 12 + TPB - 1) / TPB, 2048);
    hipLaunchKernelGGL(k_pack_point_rows, dim3(blocks), dim3(TPB), 0, stream, scene->live[0], scene->live[1],
                       scene->live[2], scene->live[3], scene->live[4], scene->N, pt_list, u_cap,
                       (const long long *)nullptr, n_unique, scene->point_rows);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}
}  // namespace pnr
