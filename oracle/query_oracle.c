/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the neural-point query stage.
 *
 * A sequential (ascending thread index) restatement of the six CUDA kernels and the
 * host orchestration of the reference's only native op,
 *   pointnerf/models/neural_points/cuda/query_worldcoords.cu:18-433
 * under the canonical deterministic semantics of SURVEY.md section 8a-note / Appendix A.
 * Nothing under pointnerf2studio_amd/ may call into this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Deviations from the (racy, time-seeded) CUDA binary, all documented in DESIGN.md:
 *   - voxel ids are handed out in order of first point index (cu:49-62 run sequentially);
 *   - per-voxel point slots are filled in ascending point index, first P kept; the
 *     time-seeded reservoir replacement of cu:65-73 and cu:152-158 is NOT reproduced;
 *   - when more than max_o voxels are occupied every voxel is kept (flagged in stats[1]);
 *   - the `voxel_idx > 0` test of cu:147 (which drops every point of voxel id 0) is
 *     reproduced when compat_drop0 != 0.
 * All arithmetic is fp32 with IEEE division and no FMA contraction
 * (build with -ffp-contract=off; see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* cu:40-44 / cu:138-141 / cu:180-183 / cu:245-247: fp32 subtract, fp32 divide, floor. */
static inline int cell_of(const float *p, const float *shift, const float *vox,
                          const int *dims, int *c)
{
    c[0] = (int)floorf((p[0] - shift[0]) / vox[0]);
    c[1] = (int)floorf((p[1] - shift[1]) / vox[1]);
    c[2] = (int)floorf((p[2] - shift[2]) / vox[2]);
    return !(c[0] < 0 || c[0] >= dims[0] || c[1] < 0 || c[1] >= dims[1] ||
             c[2] < 0 || c[2] >= dims[2]);
}

typedef struct {
    int dims[3];
    int P;
    int n_occ;          /* number of occupied voxels (may exceed max_o)          */
    int32_t *coor_2_occ; /* [vol] voxel id or -1                     cu:337-348   */
    uint8_t *coor_occ;   /* [vol] dilated occupancy                  cu:101-110   */
    int32_t *occ_2_pnts; /* [n_occ, P] point ids, -1 unfilled        cu:117-162   */
    int32_t *occ_numpnts;/* [n_occ] uncapped counts                               */
} grid_t;

static void grid_free(grid_t *g)
{
    free(g->coor_2_occ); free(g->coor_occ); free(g->occ_2_pnts); free(g->occ_numpnts);
    memset(g, 0, sizeof(*g));
}

/* claim_occ (cu:18-78) + map_coor2occ (cu:80-115) + fill_occ2pnts (cu:117-162). */
static int grid_build(grid_t *g, const float *xyz, int N, const int *dims,
                      const int *query_size, int P, const float *shift,
                      const float *vox, int compat_drop0)
{
    const int64_t vol = (int64_t)dims[0] * dims[1] * dims[2];
    memcpy(g->dims, dims, sizeof(g->dims));
    g->P = P;
    g->coor_2_occ = (int32_t *)malloc(sizeof(int32_t) * vol);
    g->coor_occ = (uint8_t *)calloc(vol, 1);
    if (!g->coor_2_occ || !g->coor_occ) return -1;
    for (int64_t i = 0; i < vol; ++i) g->coor_2_occ[i] = -1;

    /* claim_occ: ids in order of first point index. */
    int n_occ = 0;
    int32_t *occ_2_coor = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)(N > 0 ? N : 1));
    if (!occ_2_coor) return -1;
    for (int i = 0; i < N; ++i) {
        int c[3];
        if (!cell_of(xyz + 3 * (size_t)i, shift, vox, dims, c)) continue;
        int64_t lin = (int64_t)c[0] * dims[1] * dims[2] + (int64_t)c[1] * dims[2] + c[2];
        if (g->coor_2_occ[lin] == -1) {
            g->coor_2_occ[lin] = n_occ;
            occ_2_coor[3 * n_occ + 0] = c[0];
            occ_2_coor[3 * n_occ + 1] = c[1];
            occ_2_coor[3 * n_occ + 2] = c[2];
            ++n_occ;
        }
    }
    g->n_occ = n_occ;

    /* map_coor2occ: dilation of occupancy over [c - k/2, c + (k+1)/2) per axis. */
    for (int v = 0; v < n_occ; ++v) {
        const int *c = occ_2_coor + 3 * v;
        for (int x = imax(0, c[0] - query_size[0] / 2); x < imin(dims[0], c[0] + (query_size[0] + 1) / 2); ++x)
            for (int y = imax(0, c[1] - query_size[1] / 2); y < imin(dims[1], c[1] + (query_size[1] + 1) / 2); ++y)
                for (int z = imax(0, c[2] - query_size[2] / 2); z < imin(dims[2], c[2] + (query_size[2] + 1) / 2); ++z)
                    g->coor_occ[(int64_t)x * dims[1] * dims[2] + (int64_t)y * dims[2] + z] = 1;
    }
    free(occ_2_coor);

    /* fill_occ2pnts: ascending point index, first P kept. */
    g->occ_2_pnts = (int32_t *)malloc(sizeof(int32_t) * (size_t)imax(n_occ, 1) * P);
    g->occ_numpnts = (int32_t *)calloc((size_t)imax(n_occ, 1), sizeof(int32_t));
    if (!g->occ_2_pnts || !g->occ_numpnts) return -1;
    for (int64_t i = 0; i < (int64_t)imax(n_occ, 1) * P; ++i) g->occ_2_pnts[i] = -1;
    for (int i = 0; i < N; ++i) {
        int c[3];
        if (!cell_of(xyz + 3 * (size_t)i, shift, vox, dims, c)) continue;
        int64_t lin = (int64_t)c[0] * dims[1] * dims[2] + (int64_t)c[1] * dims[2] + c[2];
        int v = g->coor_2_occ[lin];
        if (compat_drop0 ? (v > 0) : (v >= 0)) {   /* cu:147 */
            int tmp = g->occ_numpnts[v]++;
            if (tmp < P) g->occ_2_pnts[(int64_t)v * P + tmp] = i;
        }
    }
    return 0;
}

/* query_neigh_along_ray_layered (cu:217-302) for one shading sample. */
static void query_one(const grid_t *g, const float *xyz, const float *loc,
                      const float *shift, const float *vox, const int *kernel_size,
                      int K, float radius_limit2, int32_t *out /* [K], preset -1 */,
                      float *buf /* [K] scratch */)
{
    const int *dims = g->dims;
    const float cx = loc[0], cy = loc[1], cz = loc[2];
    const int fx = (int)floorf((cx - shift[0]) / vox[0]);
    const int fy = (int)floorf((cy - shift[1]) / vox[1]);
    const int fz = (int)floorf((cz - shift[2]) / vox[2]);
    int kid = 0, far_ind = 0;
    float far2 = 0.0f;
    for (int layer = 0; layer < (kernel_size[0] + 1) / 2; ++layer) {
        for (int x = imax(-fx, -layer); x < imin(dims[0] - fx, layer + 1); ++x) {
            for (int y = imax(-fy, -layer); y < imin(dims[1] - fy, layer + 1); ++y) {
                for (int z = imax(-fz, -layer); z < imin(dims[2] - fz, layer + 1); ++z) {
                    if (imax(abs(z), imax(abs(x), abs(y))) != layer) continue;
                    int64_t lin = (int64_t)(fx + x) * dims[1] * dims[2] + (int64_t)(fy + y) * dims[2] + (fz + z);
                    int v = g->coor_2_occ[lin];
                    if (v < 0) continue;
                    int cnt = imin(g->P, g->occ_numpnts[v]);
                    for (int s = 0; s < cnt; ++s) {
                        int p = g->occ_2_pnts[(int64_t)v * g->P + s];
                        float xv = xyz[3 * (size_t)p + 0] - cx;
                        float yv = xyz[3 * (size_t)p + 1] - cy;
                        float zv = xyz[3 * (size_t)p + 2] - cz;
                        float d2 = xv * xv + yv * yv + zv * zv;
                        if (radius_limit2 == 0.0f || d2 <= radius_limit2) {
                            if (kid++ < K) {
                                out[kid - 1] = p;
                                buf[kid - 1] = d2;
                                if (d2 > far2) { far2 = d2; far_ind = kid - 1; }
                            } else if (d2 < far2) {
                                out[far_ind] = p;
                                buf[far_ind] = d2;
                                far2 = d2;
                                for (int i = 0; i < K; ++i)
                                    if (buf[i] > far2) { far2 = buf[i]; far_ind = i; }
                            }
                        }
                    }
                }
            }
        }
        if (kid >= K) break;
    }
}

/*
 * woord_query_grid_point_index_cuda (cu:305-433) with B = 1.
 *
 * raypos [R,D,3]; xyz [N,3]; outputs are written COMPACTED over the kept rays at the
 * front of caller-allocated worst-case buffers:
 *   sample_pidx [R,SR,K] int32, sample_loc [R,SR,3] f32, ray_mask [R] int8.
 * stats[0] = occupied voxels, stats[1] = (occupied > max_o), stats[2] = R' (rays that hit
 * the dilated occupancy), stats[3] = R'' (rays kept), stats[4] = valid shading samples
 * (>= 1 neighbour), stats[5] = valid (sample, neighbour) pairs, stats[6] = selected
 * shading samples (slots filled), stats[7] = candidates distance-tested.
 * Returns R'' or a negative error.
 */
int pnr_oracle_query(const float *raypos, int R, int D, const float *xyz, int N,
                     const int *kernel_size, const int *query_size, int SR, int K,
                     const int *dims, int max_o, int P, float radius_limit,
                     const float *ranges, const float *vox, int compat_drop0,
                     int32_t *sample_pidx, float *sample_loc, int8_t *ray_mask,
                     int64_t *stats)
{
    const float *shift = ranges;
    grid_t g;
    memset(&g, 0, sizeof(g));
    if (grid_build(&g, xyz, N, dims, query_size, P, shift, vox, compat_drop0) != 0) {
        grid_free(&g);
        return -1;
    }
    stats[0] = g.n_occ;
    stats[1] = g.n_occ > max_o;

    const float radius_limit2 = radius_limit * radius_limit;   /* cu:410, fp32 */
    float *buf = (float *)malloc(sizeof(float) * (size_t)imax(K, 1));
    int32_t *slot = (int32_t *)malloc(sizeof(int32_t) * (size_t)imax(D, 1));
    int n_hit = 0, n_keep = 0;
    int64_t n_valid_samples = 0, n_pairs = 0, n_selected = 0;

    for (int r = 0; r < R; ++r) {
        /* mask_raypos (cu:165-189) + host slotting (cu:381-391). */
        int cum = 0, any = 0;
        for (int j = 0; j < D; ++j) {
            int c[3], m = 0;
            if (cell_of(raypos + 3 * ((size_t)r * D + j), shift, vox, dims, c))
                m = g.coor_occ[(int64_t)c[0] * dims[1] * dims[2] + (int64_t)c[1] * dims[2] + c[2]];
            cum += m;
            any |= m;
            slot[j] = (m && cum <= SR) ? cum - 1 : -1;
        }
        ray_mask[r] = 0;
        if (!any) continue;
        ++n_hit;

        /* get_shadingloc (cu:192-214) into the next compacted row. */
        int32_t *pid = sample_pidx + (size_t)n_keep * SR * K;
        float *loc = sample_loc + (size_t)n_keep * SR * 3;
        for (int i = 0; i < SR * K; ++i) pid[i] = -1;
        memset(loc, 0, sizeof(float) * (size_t)SR * 3);
        int n_slots = 0;
        for (int j = 0; j < D; ++j) {
            if (slot[j] < 0) continue;
            memcpy(loc + 3 * slot[j], raypos + 3 * ((size_t)r * D + j), 3 * sizeof(float));
            ++n_slots;
        }
        /* query_neigh_along_ray_layered for every filled slot. */
        int ray_has_neighbour = 0;
        int64_t s_valid = 0, s_pairs = 0;
        for (int s = 0; s < n_slots; ++s) {
            query_one(&g, xyz, loc + 3 * s, shift, vox, kernel_size, K, radius_limit2,
                      pid + (size_t)s * K, buf);
            int cnt = 0;
            for (int k = 0; k < K; ++k) cnt += pid[(size_t)s * K + k] >= 0;
            s_pairs += cnt;
            s_valid += cnt > 0;
        }
        ray_has_neighbour = s_pairs > 0;
        /* host post-filter (cu:425-429): rays without any neighbour are dropped. */
        if (ray_has_neighbour) {
            ray_mask[r] = 1;
            ++n_keep;
            n_valid_samples += s_valid;
            n_pairs += s_pairs;
            n_selected += n_slots;
        }
    }
    stats[2] = n_hit;
    stats[3] = n_keep;
    stats[4] = n_valid_samples;
    stats[5] = n_pairs;
    stats[6] = n_selected;
    stats[7] = 0;
    free(buf);
    free(slot);
    grid_free(&g);
    return n_keep;
}
