// Shade stage: neighbour gather -> dists / inverse-distance weights / positional encodings ->
// mlp_base -> mlp_head -> density head -> weighted K-aggregation, then the colour MLP per sample.
// Replaces studio_utils.py:190-207 (w2pers over ALL N points + five index_select gathers) and
// studio_model.py:270-365 (boolean compactions, [M,284] / [M,263] materialisations, rocBLAS GEMMs).
//
// MI355X design:
//   * The MLP is evaluated TRANSPOSED, H^T = W . X^T: the weights are the MFMA A operand (output
//     features on the 32 tile rows), the (sample, neighbour) rows sit on the 32 tile COLUMNS = lanes.
//     The 32x32 accumulator layout (col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5)) is then
//     exactly a B operand of the next layer, so a layer's output registers feed the next layer with
//     NO data movement: no LDS round trip for activations, no transposition; activations never leave
//     the VGPR file.  The k-order this implies is baked into the packed weights (pnr_weights_pack).
//   * One wavefront owns 32 rows (4 samples x K=8 neighbours) and all 256 features; one wave per SIMD,
//     four waves per CU, persistent grid of one workgroup per CU (fp32: a contiguous tile range per workgroup;
//     bf16x3: XCD-aware interleaved tiles, see k_shade_pairs_bf16).
//   * The gather reads one 192-byte packed row per neighbour; the two lanes that share a row (l, l+32)
//     split its features, so no positional encoding is computed twice.
//   * K-aggregation is a segmented butterfly over the 8 lanes of a sample, in registers.
// Two arithmetic modes (pnr_render_opts_t.precision):
//   PNR_PRECISION_FP32   v_mfma_f32_32x32x2_f32: every product and sum in fp32 (an fp32 fma chain per output).
//                        Weights stream L2 -> VGPR through a buffer descriptor (one 1-KiB load per 4 MFMAs).
//   PNR_PRECISION_BF16X3 v_mfma_f32_32x32x16_bf16 on hi/lo splits: a*b ~ ah*bh + ah*bl + al*bh with fp32
//                        accumulation (relative error ~2^-16 per product; RGB within 1e-5 of the fp32
//                        path on the parity scenes).  3 MFMAs of 32 cycles replace 8 of 64: the weights
//                        are consumed ~5x faster, so the four waves share them through LDS (LDS-DMA into a
//                        4-slot ring of 16..36 KiB tiles, one barrier per tile); density head and
//                        K-aggregation run inside the last layer's MFMA shadow.
// In both modes mlp_base layer 0 is factorised: k_point_part(_f32) contracts its 224 point-only inputs once per
// distinct neighbour point of the call, the pair kernel starts from that row (pt_table) and multiplies the 60 encoded
// distances.  DESIGN.md section 4.1 has the measurements.
#include <algorithm>

#include "pnr_internal.h"

namespace pnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// diagnostic builds only (never shipped; results are wrong by construction): bit0 cheap PE, bit1 no MFMA, bit3 no
// barrier + no DMA, bit4 no split, bit5 no DMA issue, bit6 no barrier (tools/build_ablate.sh)
#ifndef PNR_ABLATE
#define PNR_ABLATE 0
#endif

// diagnostic builds only: -DPNR_STAMPS=1 accumulates s_memtime deltas of the phases of k_shade_pairs_bf16 per wave
// and writes them (never into an output) to the tail of the smp_sigma buffer
#ifndef PNR_STAMPS
#define PNR_STAMPS 0
#endif
__device__ __forceinline__ unsigned long long stamp()
{
#if PNR_STAMPS
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
#else
    return 0;
#endif
}

constexpr int WAVES = 4;
constexpr int TPB = WAVES * 64;
constexpr int PF = 6;  // fp32 path: weight loads (1 KiB each per wave) kept in flight

// LeakyReLU(0.1): max(x, 0.1 x) (identical to the select form for finite x, one instruction shorter)
// LeakyReLU(0.1) = max(x, 0.1x); fmaxf costs an extra instruction (hipcc canonicalises the operand first:
// v_max_f32 v, v, v)
// (kept as one v_mul + one v_max through inline asm: any builtin form is turned back into canonicalise + max)
__device__ __forceinline__ float leaky(float x)
{
    float r;
    const float y = 0.1f * x;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

struct ShadeParams {
    const float4 *point_rows;  // [N, 12] float4: a0 | c0 | c1 | pad | emb[8]
    const float *wbuf;         // packed weights (fp32 A-operand order, bf16x3 tiles, plain heads, biases)
    size_t wbytes;
    size_t w_off[9];    // float offsets: fp32-packed layers / plain heads
    size_t w16_off[9];  // float offsets: bf16x3-packed layers (0 for the heads)
    size_t b_off[9];
    float Rw2c[9];
    CamRef cr;
    const float *dirs;
    const float4 *smp_loc;
    const int *smp_ray;
    const int *smp_pidx;
    const int *vs_list;
    const int *n_sel;  // [1] = S_valid
    float *smp_sigma;  // [S_valid]
    float *agg;        // [S_valid, 256]
    float4 *smp_out;   // [S_sel]
    int K;
    long long dbg_off;  // PNR_STAMPS builds: float offset into smp_sigma of the stamp area
    // bf16x3 mode: factorised first layer
    int i_v0, i_v1;        // the kernel works on positions [n_sel[i_v0], n_sel[i_v1]) of vs_list
    float *smp_sig_s;      // [S_sel] density by sample index (early ray termination), may be null
    size_t w16a_off, w16b_off, w4acc_off, w8acc_off;
    size_t w32a_off, w32b_off;   // fp32-packed halves of mlp_base layer 0 (point-only k-steps 0..111, pair 112..143)
    const int *pt_rank;     // [N+1] point index -> row of pt_table
    const int *pt_list;     // [U] rows -> point index
    float4 *pt_table;       // [u_cap, 8 row blocks, 2 lane halves, 4] float4
    int u_cap;
};

__device__ __forceinline__ float4 load_w(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ void rot_rows(const float (&M)[9], float x, float y, float z, float &ox, float &oy,
                                         float &oz)
{
    // v @ M^T : out[i] = sum_j v[j] * M[i][j]
    ox = x * M[0] + y * M[1] + z * M[2];
    oy = x * M[3] + y * M[4] + z * M[5];
    oz = x * M[6] + y * M[7] + z * M[8];
}

__device__ __forceinline__ void to_cam(const Camera &cam, float x, float y, float z, float &cx, float &cy, float &cz)
{
    // (p - o) @ Rc2w : out[i] = sum_j s[j] * R[j][i]      (studio_utils.py:129-144)
    const float sx = x - cam.o[0], sy = y - cam.o[1], sz = z - cam.o[2];
    cx = sx * cam.R[0] + sy * cam.R[3] + sz * cam.R[6];
    cy = sx * cam.R[1] + sy * cam.R[4] + sz * cam.R[7];
    cz = sx * cam.R[2] + sy * cam.R[5] + sz * cam.R[8];
}

// sin and cos for the bf16x3 mode: Cody-Waite reduction by pi/2 (two constants, exact for |x| < ~800) + the
// single-precision minimax polynomials on [-pi/4, pi/4] (max error ~1e-7, below the mode's 2^-16 products).
// Branch-free and ~25 instructions against ~60 for sincosf with its large-argument path; arguments beyond the
// reduction's range (never produced by trained embeddings or voxel-sized distances) fall back to sincosf.
__device__ __forceinline__ void fast_sincos(float x, float &sn, float &cs)
{
    if (__builtin_expect(fabsf(x) > 512.0f, 0)) {
        sincosf(x, &sn, &cs);
        return;
    }
    const float k = rintf(x * 0.636619772367581343f);           // x * 2/pi
    float r = fmaf(-k, 1.5707962512969970703125f, x);           // pi/2 high part
    r = fmaf(-k, 7.54978995489188216e-8f, r);                   // pi/2 low part
    const float z = r * r;
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    const float s0 = fmaf(r * z, ps, r);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    const float c0 = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? c0 : s0, cc = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -ss : ss;
    cs = ((q + 1) & 2) ? -cc : cc;
}

// v + (v of the lane DPP control CTRL selects): 0xB1 / 0x4E = quad_perm xor 1 / xor 2, 0x141 = row_half_mirror,
// 0x140 = row_mirror
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Lane segment of one sample inside each 32-lane half.  SEG = 8 (K <= 8) / 16 (K <= 16): the sample's K rows sit in
// the first K lanes of an 8- / 16-lane segment aligned to the DPP rows (the other lanes of the segment idle: pidx -1,
// weight 0), so that sums over a sample are DPP steps.  SEG = 0 (K > 16): segments of exactly K lanes, summed with
// K cross-lane reads.
template <int SEG>
__device__ __forceinline__ int seg_len(int K)
{
    return SEG ? SEG : K;
}

// sum over the lanes of one sample's segment (idle lanes must hold 0)
template <int SEG>
__device__ __forceinline__ float seg_sum(float v, int K, int lane)
{
    if (SEG == 8 || SEG == 16) {
        // all on the VALU (DPP): xor-1 and xor-2 inside each quad, the mirrored quad of the 8-lane half row (lane i <-
        // lane 7 - i), and for 16 lanes the mirrored half row -- no LDS crossbar (ds_bpermute) involved
        v = dpp_add<0xB1>(v);
        v = dpp_add<0x4E>(v);
        v = dpp_add<0x141>(v);
        if (SEG == 16) v = dpp_add<0x140>(v);
        return v;
    } else {
        const int j = lane & 31;
        const int base = (lane & 32) + (j / K) * K;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += __shfl(v, min(base + k, 63), 64);
        return s;
    }
}

// ------------------------------------------------------------------------------------------------
// rows of a tile: gather + per-row features (shared by both arithmetic modes)
// ------------------------------------------------------------------------------------------------
struct RowCtx {
    int s;        // sample index of this lane's row
    int v_idx;    // valid-sample index of this lane's row
    int slot;     // neighbour slot of the row
    bool row_ok;  // the row maps to a real (sample, slot)
    bool smp_ok;  // the lane's segment maps to a real sample (the lane may still be an idle slot >= K)
    float wgt;    // normalised inverse-distance weight (0 for unfilled slots)
    float ex[4];  // this lane half's share of [color(3), dir - view (3), <dir, view>, 0]
};

// The lane's 144 layer-1 input values: value i = 8s + j is element j of k-step s in the bf16 path and k-step t = i in
// the fp32 path.  Lane half h = 0 carries emb[0:16], their encodings (values 0..111: point_inputs, computed once per
// distinct point by k_point_part) and the rotated world distances (values 112..143: pair_inputs), h = 1 carries
// emb[16:32], their encodings and the camera-space distances.
// Gathered inputs of one lane's (sample, neighbour) row.  The three dependent load levels are separate functions
// so that the bf16x3 kernel can issue them for the NEXT tile between the layers of the current one (one wave per
// SIMD cannot hide a vs_list -> smp_pidx -> point-row chain of three HBM/L2 latencies any other way).
struct RowFetch {
    int v_idx, slot, s, pidx, ray, urow, cid;
    bool row_ok, smp_ok;
    float4 a0, c0, c1, loc;
    float dirx, diry, dirz;
};

template <int SEG>
__device__ __forceinline__ void fetch_a(const ShadeParams &P, int tile, int lane, int wave, int V0, int S_valid,
                                        RowFetch &f)
{
    const int j = lane & 31;
    const int L = seg_len<SEG>(P.K);
    const int SPW = 32 / L;
    const int SPT = SPW * WAVES;
    const int sl = j / L;
    f.v_idx = V0 + tile * SPT + wave * SPW + sl;
    f.slot = j - sl * L;
    f.smp_ok = (j < SPW * L) && (f.v_idx < S_valid);
    f.row_ok = f.smp_ok && f.slot < P.K;
    // unconditional loads at clamped indices: a branch here would end the basic block, and hipcc then sinks the
    // hi/lo split of the previous layer out of the MFMA shadows into the block behind the branch
    // (the select on row_ok happens in fetch_b: here it would put a wait for this load right behind its issue)
    f.s = P.vs_list[f.row_ok ? f.v_idx : 0];
}

template <int SEG>
__device__ __forceinline__ void fetch_b(const ShadeParams &P, RowFetch &f)
{
    const int K = P.K;
    f.s = f.row_ok ? f.s : 0;
    const int pv = P.smp_pidx[(int64_t)f.s * K + (f.row_ok ? f.slot : 0)];
    f.pidx = f.row_ok ? pv : -1;
    f.loc = P.smp_loc[f.s];
    f.ray = P.smp_ray[f.s];
}

// Camera of a wavefront whose rays all belong to camera cid0, through the scalar cache.  (hipcc emits VECTOR loads
// for load_cam even at a uniform address -- the kernel stores to global memory -- and vector loads return in
// order: behind the 32 pt_table gathers of the tile they would expose the whole gather latency.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Camera load_cam_scalar(const CamRef &cr, int cid0)
{
    const float *p = reinterpret_cast<const float *>(cr.cams + cid0);
    i32x4 a, b, c;
    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c)
                 : "s"(p)
                 : "memory");
    Camera cam;
    cam.o[0] = __int_as_float(a.x);
    cam.o[1] = __int_as_float(a.y);
    cam.o[2] = __int_as_float(a.z);
    cam.R[0] = __int_as_float(a.w);
    cam.R[1] = __int_as_float(b.x);
    cam.R[2] = __int_as_float(b.y);
    cam.R[3] = __int_as_float(b.z);
    cam.R[4] = __int_as_float(b.w);
    cam.R[5] = __int_as_float(c.x);
    cam.R[6] = __int_as_float(c.y);
    cam.R[7] = __int_as_float(c.z);
    cam.R[8] = __int_as_float(c.w);
    return cam;
}
__device__ __forceinline__ Camera load_cam_wave(const CamRef &cr, int cid)
{
    const int cid0 = __builtin_amdgcn_readfirstlane(cid);
    if (__all(cid == cid0)) return load_cam_scalar(cr, cid0);
    // a wavefront straddling two ray bundles (rare): per-lane loads, retired inside this branch so that the join
    // carries no pending vector load (hipcc would wait vmcnt(0) there on every tile)
    Camera c = load_cam(cr, cid);
    asm volatile("" ::"v"(c.o[0]), "v"(c.o[1]), "v"(c.o[2]), "v"(c.R[0]), "v"(c.R[1]), "v"(c.R[2]), "v"(c.R[3]),
                 "v"(c.R[4]), "v"(c.R[5]), "v"(c.R[6]), "v"(c.R[7]), "v"(c.R[8]));
    return c;
}

// branch-free camera index of a ray (cam_id() branches; a branch between the layers would split their basic block)
__device__ __forceinline__ int cam_id_flat(const CamRef &cr, const int *valid_ints, int ray)
{
    const int *src = cr.ray_cam ? cr.ray_cam + ray : valid_ints;
    const int listed = *src;
    const unsigned rpc = (unsigned)max((long long)1, (long long)cr.rays_per_cam);
    const int by_div = (int)((unsigned)ray / rpc);
    const int cid = cr.ray_cam ? listed : by_div;
    return cr.n_cams <= 1 ? 0 : cid;
}

// the embedding is not needed per pair (its first-layer contribution comes from pt_table)
__device__ __forceinline__ void fetch_c_pair(const ShadeParams &P, RowFetch &f)
{
    const int p = max(f.pidx, 0);
    const float4 *row = P.point_rows + (int64_t)p * 12;
    f.a0 = row[0];
    f.c0 = row[1];
    f.c1 = row[2];
    f.urow = min(P.pt_rank[p], P.u_cap - 1);
    f.cid = cam_id_flat(P.cr, P.n_sel, f.ray);
    f.dirx = P.dirs[3 * (int64_t)f.ray];
    f.diry = P.dirs[3 * (int64_t)f.ray + 1];
    f.dirz = P.dirs[3 * (int64_t)f.ray + 2];
}

// the lane's point-only layer-1 inputs: 16 embedding channels and their encodings (x0[0:112])
template <bool FAST_PE>
__device__ __forceinline__ void point_inputs(const float (&e)[16], float *x0)
{
#pragma unroll
    for (int d = 0; d < 16; ++d) x0[d] = e[d];
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        float sn = 0.f, cs = 1.f;
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            if (PNR_ABLATE & 1) {
                sn = e[d] * (float)(1 << f);
                cs = 1.0f - sn;
            } else if (FAST_PE && f > 0) {
                // double angle from the previous octave: sin 2a = 2 sin a cos a, cos 2a = (cos a - sin a)(cos a + sin a)
                const float s2 = 2.0f * sn * cs, c2 = (cs - sn) * (cs + sn);
                sn = s2;
                cs = c2;
            } else if (FAST_PE) {
                fast_sincos(e[d], sn, cs);
            } else {
                sincosf(e[d] * (float)(1 << f), &sn, &cs);
            }
            x0[16 + (d * 3 + f) * 2 + 0] = sn;
            x0[16 + (d * 3 + f) * 2 + 1] = cs;
        }
    }
}

// the lane's pair inputs: weight, encoded distances (xq[0:32] = x0[112:144]) and the extra head inputs
template <int SEG, bool FAST_PE>
__device__ __forceinline__ void pair_inputs(const ShadeParams &P, const RowFetch &f, const Camera &cam, int lane,
                                            float *xq, RowCtx &ctx)
{
    const int h = lane >> 5;
    const int K = P.K;
    ctx.s = f.s;
    ctx.v_idx = f.v_idx;
    ctx.row_ok = f.row_ok;
    ctx.smp_ok = f.smp_ok;
    ctx.slot = f.slot;
    const bool valid = f.pidx >= 0;
    const float4 a0 = f.a0, c0 = f.c0, c1 = f.c1, loc = f.loc;
    const float dirx = f.dirx, diry = f.diry, dirz = f.dirz;

    // dists + inverse-distance weight (studio_model.py:270-286,467-475)
    const float dwx = a0.x - loc.x, dwy = a0.y - loc.y, dwz = a0.z - loc.z;
    const float nrm = sqrtf(dwx * dwx + dwy * dwy + dwz * dwz);
    float wgt = valid ? 1.0f / fmaxf(nrm, 1e-6f) : 0.f;
    const float wsum = seg_sum<SEG>(wgt, K, lane);
    ctx.wgt = wgt / fmaxf(wsum, 1e-8f);

    float dd[3];
    if (h == 0) {
        rot_rows(P.Rw2c, dwx, dwy, dwz, dd[0], dd[1], dd[2]);  // dists[:3] @ Rw2c^T   (studio_model.py:313)
    } else {
        float pcx, pcy, pcz, scx, scy, scz;
        to_cam(cam, a0.x, a0.y, a0.z, pcx, pcy, pcz);
        to_cam(cam, loc.x, loc.y, loc.z, scx, scy, scz);
        const float ppx = pcx / pcz, ppy = pcy / pcz, spx = scx / scz, spy = scy / scz;
        dd[0] = ppx * pcz - spx * scz;
        dd[1] = ppy * pcz - spy * scz;
        dd[2] = pcz - scz;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float sn = 0.f, cs = 1.f;
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            if (PNR_ABLATE & 1) {
                sn = dd[d] * (float)(1 << f);
                cs = 1.0f - sn;
            } else if (FAST_PE && f > 0) {
                const float s2 = 2.0f * sn * cs, c2 = (cs - sn) * (cs + sn);
                sn = s2;
                cs = c2;
            } else if (FAST_PE) {
                fast_sincos(dd[d], sn, cs);
            } else {
                sincosf(dd[d] * (float)(1 << f), &sn, &cs);
            }
            xq[(d * 5 + f) * 2 + 0] = sn;
            xq[(d * 5 + f) * 2 + 1] = cs;
        }
    }
    xq[30] = 0.f;
    xq[31] = 0.f;

    // [color(3), dir @ Rw2c^T - view (3), <dir @ Rw2c^T, view> (1)]   (studio_model.py:322-335)
    float sdx, sdy, sdz, vx, vy, vz;
    rot_rows(P.Rw2c, c0.w, c1.x, c1.y, sdx, sdy, sdz);
    rot_rows(P.Rw2c, dirx, diry, dirz, vx, vy, vz);
    const float dv0 = sdx - vx, dv1 = sdy - vy, dv2 = sdz - vz;
    const float dot = sdx * vx + sdy * vy + sdz * vz;
    ctx.ex[0] = h ? c0.y : c0.x;
    ctx.ex[1] = h ? dv0 : c0.z;
    ctx.ex[2] = h ? dv2 : dv1;
    ctx.ex[3] = h ? 0.f : dot;
}

// Bias-initialised accumulator of an output tile.  The tile's 32 biases are wave-uniform: they are fetched through
// the SCALAR cache (s_load_dwordx16 x2, counted on lgkmcnt, issued a few k-steps ahead) and selected per lane
// half.  Not from LDS: hipcc cannot tell an LDS read from the LDS-DMA destinations in flight and guards it with
// s_waitcnt vmcnt(0), draining the DMA once per tile; not by VMEM either: vector memory returns in order, behind
// the DMA.  (hipcc emits vector loads for a plain `bias[i]`, hence the inline asm.)
typedef int i32x16 __attribute__((ext_vector_type(16)));
struct BiasRegs {
    i32x16 a, b;
};

__device__ __forceinline__ void bias_issue(const float *bias32, BiasRegs &r)
{
    // early-clobber outputs: a destination tuple must not overlap the address pair the second load still reads
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(r.a), "=&s"(r.b) : "s"(bias32) : "memory");
}

__device__ __forceinline__ f32x16 bias_finish(BiasRegs &r, int h)
{
    // also retires the (at most four) fragment reads in flight: one LDS latency per tile
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.a), "+s"(r.b)::"memory");
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lo = 8 * q + i, hi = 8 * q + 4 + i;
            const float flo = __int_as_float(lo < 16 ? r.a[lo] : r.b[lo - 16]);
            const float fhi = __int_as_float(hi < 16 ? r.a[hi] : r.b[hi - 16]);
            acc[4 * q + i] = h ? fhi : flo;
        }
    return acc;
}

__device__ __forceinline__ void bias_wait(BiasRegs &r)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.a), "+s"(r.b)::"memory");
}

// accumulator values 4q .. 4q+3 of the next tile (rows 8q + 4h + i): every lane takes the h = 0 row from the
// SGPRs, then the upper lane half is overwritten under a half exec mask -- 2 VALU per value instead of the
// 2 v_mov + v_cndmask a select on two SGPRs costs (one constant-bus operand per VALU on gfx9).
// The wave runs the dense layers with all 64 lanes active, so exec_lo is restored to -1.
__device__ __forceinline__ void bias_quarter(const BiasRegs &r, int q4, f32x16 &acc)
{
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        const int lo = 8 * q4 + i, hi = lo + 4;
        const int l0 = lo < 16 ? r.a[lo] : r.b[lo - 16], l1 = lo + 1 < 16 ? r.a[lo + 1] : r.b[lo + 1 - 16];
        const int h0 = hi < 16 ? r.a[hi] : r.b[hi - 16], h1 = hi + 1 < 16 ? r.a[hi + 1] : r.b[hi + 1 - 16];
        float a0, a1;
        // straight into accumulator registers (hipcc keeps MFMA accumulators of this kernel in AGPRs: a VGPR result
        // would cost a v_accvgpr_write per value at the tile boundary)
        asm volatile("v_accvgpr_write_b32 %0, %2\n\tv_accvgpr_write_b32 %1, %3\n\ts_mov_b32 exec_lo, 0\n\t"
                     "v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\ts_mov_b32 exec_lo, -1"
                     : "=&a"(a0), "=&a"(a1)
                     : "s"(l0), "s"(l1), "s"(h0), "s"(h1));
        acc[4 * q4 + i] = a0;
        acc[4 * q4 + i + 1] = a1;
    }
}

// bf16x3 mode: layout of the aggregated features between the pair and the colour kernel.  The colour kernel's lane
// (j, h) of the wave that owns samples 32b .. 32b+31 needs, for k-step k, features 16k + 8h + {0..7} of sample
// 32b + j: stored as two float4 (hp = 0, 1) at float4 index ((b*16 + k)*2 + hp)*64 + j + 32h, so each of its 32
// loads is one contiguous KiB per wave (row-major rows cost 32 scattered 16-byte loads per lane: ~7k cycles of the
// CU's texture-address unit per tile, tools/ub_gather.hip).  The pair kernel's lane holding features
// 32t + 8q + 4hp + {0..3} writes chunk (k = 2t + (q>>1), hp, h = q&1).
__device__ __forceinline__ int64_t agg_idx4(int v, int k, int hp, int h)
{
    return ((((int64_t)(v >> 5) * 16 + k) * 2 + hp) * 64) + (v & 31) + 32 * h;
}

// density head + weighted K-aggregation + stores (studio_model.py:337-353)
template <int SEG, bool PACKED>
__device__ __forceinline__ void finish_rows(const ShadeParams &P, int lane, const float (&hC)[128],
                                            const RowCtx &ctx)
{
    const int h = lane >> 5;
    const int K = P.K;
    const float *w4 = P.wbuf + P.w_off[4];
    const float b4 = P.wbuf[P.b_off[4]];
    // The 256 head weights are wave-uniform: 32 at a time through the scalar cache (hipcc turned the per-lane
    // float4 loads of an earlier version into 32 load -> vmcnt(0) -> use round trips, 12k cycles per tile).
    // Both lane halves multiply with uniform weights (SGPR operand) and pick their own sum at the end.
    float part_lo = 0.f, part_hi = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        BiasRegs wr;
        bias_issue(w4 + 32 * m, wr);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wr.a), "+s"(wr.b)::"memory");
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lo = 8 * q + i, hi = 8 * q + 4 + i;
                const float wlo = __int_as_float(lo < 16 ? wr.a[lo] : wr.b[lo - 16]);
                const float whi = __int_as_float(hi < 16 ? wr.a[hi] : wr.b[hi - 16]);
                part_lo += hC[m * 16 + 4 * q + i] * wlo;
                part_hi += hC[m * 16 + 4 * q + i] * whi;
            }
    }
    float part = h ? part_hi : part_lo;
    part += __shfl_xor(part, 32, 64);
    const float alpha = fmaxf(part + b4, 0.f);
    const float sigma = seg_sum<SEG>(alpha * ctx.wgt, K, lane);
    const bool writer = ctx.row_ok && ctx.slot == 0;
    if (writer && h == 0) {
        P.smp_sigma[ctx.v_idx] = sigma;
        if (P.smp_sig_s) P.smp_sig_s[ctx.s] = sigma;
    }
    float *dst = P.agg + (int64_t)ctx.v_idx * 256;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 o;
            o.x = seg_sum<SEG>(hC[m * 16 + 4 * q + 0] * ctx.wgt, K, lane);
            o.y = seg_sum<SEG>(hC[m * 16 + 4 * q + 1] * ctx.wgt, K, lane);
            o.z = seg_sum<SEG>(hC[m * 16 + 4 * q + 2] * ctx.wgt, K, lane);
            o.w = seg_sum<SEG>(hC[m * 16 + 4 * q + 3] * ctx.wgt, K, lane);
            if (writer) {
                if (PACKED)
                    reinterpret_cast<float4 *>(P.agg)[agg_idx4(ctx.v_idx, 2 * m + (q >> 1), h, q & 1)] = o;
                else
                    *reinterpret_cast<float4 *>(dst + 32 * m + 8 * q + 4 * h) = o;
            }
        }
}

// colour head on the last hidden layer: 128 -> 3, sigmoid, widen (studio_model.py:357-359)
__device__ __forceinline__ void color_head(const ShadeParams &P, int lane, const float (&hA)[64], float (&rgb)[3])
{
    const int h = lane >> 5;
    const float *w8 = P.wbuf + P.w_off[8];
    const float *b8 = P.wbuf + P.b_off[8];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float part = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 wv = *reinterpret_cast<const float4 *>(w8 + c * 128 + 32 * m + 8 * q + 4 * h);
                part += hA[m * 16 + 4 * q + 0] * wv.x;
                part += hA[m * 16 + 4 * q + 1] * wv.y;
                part += hA[m * 16 + 4 * q + 2] * wv.z;
                part += hA[m * 16 + 4 * q + 3] * wv.w;
            }
        part += __shfl_xor(part, 32, 64);
        const float z = part + b8[c];
        const float sg = 1.0f / (1.0f + expf(-z));
        rgb[c] = sg * (1.0f + 2.0f * 0.001f) - 0.001f;
    }
}

// the same with the weights from the LDS table w8tab[((c * 4 + t) * 2 + h) * 16 + r] (accumulator order).  Inline asm
// reads (hipcc would guard plain ones with s_waitcnt vmcnt(0) while the next tile's weight DMA is in flight), all
// twelve per colour in flight at once.
__device__ __forceinline__ void color_head_lds(const float (&b8)[3], const u32x4 *w8tab, int lane,
                                               const float (&hA)[64], float (&rgb)[3])
{
    const int h = lane >> 5;
    const unsigned base = (unsigned)(uintptr_t)w8tab + 64u * h;
    float part[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f32x4 wv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wv[i]) : "v"(base), "n"(512 * c + 128 * (i >> 2) + 16 * (i & 3)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // the opaque "+v" keeps the multiplies behind the wait above
            asm volatile("" : "+v"(wv[i]));
            acc += hA[4 * i + 0] * wv[i].x;
            acc += hA[4 * i + 1] * wv[i].y;
            acc += hA[4 * i + 2] * wv[i].z;
            acc += hA[4 * i + 3] * wv[i].w;
        }
        part[c] = acc;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float pc = part[c] + __shfl_xor(part[c], 32, 64);
        const float z = pc + b8[c];
        const float sg = 1.0f / (1.0f + expf(-z));
        rgb[c] = sg * (1.0f + 2.0f * 0.001f) - 0.001f;
    }
}

// ================================================================================================
// fp32 mode
// ================================================================================================
// One dense layer.  in[KSP] are this lane's B-operand registers (k-step t: the lane supplies one input
// feature of its row), out[MT*16] the accumulators.  The layer's packed A operands ([MT][KSP/4][64 lanes]
// float4) start at byte offset wbase of the weight buffer: buffer loads with the wave-uniform offset in an
// SGPR, so the ~1000 loads of an unrolled layer share ONE address VGPR (with 64-bit global addresses hipcc
// hoists a distinct address pair per load out of the tile loop and spills ~2000 VGPRs).
// `init` (per lane: float4 index 8m + q = accumulator values 4q..4q+3 of output tile m, the pt_table row of the
// lane's pair) replaces the bias when the layer continues a sum started elsewhere.
template <int KSP, int MT>
__device__ __forceinline__ void dense_layer(__amdgpu_buffer_rsrc_t rsrc, int wbase, const float *__restrict__ bias,
                                            int lane, const float (&in)[KSP], float (&out)[MT * 16],
                                            const float4 *__restrict__ init = nullptr)
{
    static_assert(KSP % 4 == 0, "k-steps are packed in groups of 4");
    constexpr int KG = KSP / 4;
    constexpr int NG = MT * KG;
    const int h = lane >> 5;
    const int voff = lane * 16;
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = init ? init[8 * m + q] : *reinterpret_cast<const float4 *>(bias + 32 * m + 8 * q + 4 * h);
            acc[m][4 * q + 0] = b.x;
            acc[m][4 * q + 1] = b.y;
            acc[m][4 * q + 2] = b.z;
            acc[m][4 * q + 3] = b.w;
        }
    }
    float4 wq[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) wq[p] = load_w(rsrc, voff, wbase + p * 1024);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int m = g / KG, kg = g % KG;
        const float4 w = wq[g % PF];
        if (g + PF < NG) wq[g % PF] = load_w(rsrc, voff, wbase + (g + PF) * 1024);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, in[4 * kg + 0], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, in[4 * kg + 1], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, in[4 * kg + 2], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, in[4 * kg + 3], acc[m], 0, 0, 0);
        // pin the schedule: keep the rolling window of PF loads in flight, nothing hoisted further
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[m * 16 + r] = acc[m][r];
}

template <int SEG>
__global__ void __launch_bounds__(TPB, 1) k_shade_pairs(ShadeParams P)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int SPT = (32 / seg_len<SEG>(P.K)) * WAVES;  // samples per workgroup tile
    const int V0 = P.n_sel[P.i_v0], S_valid = P.n_sel[P.i_v1];
    const int ntiles = (S_valid - V0 + SPT - 1) / SPT;
    // XCD-aware tile order (see k_shade_pairs_bf16): 32 consecutive tiles per XCD and round, so that the pt_table rows
    // neighbouring rays share are fetched into that XCD's L2 once (12.7 GB beyond L2 per launch with one contiguous
    // tile range per workgroup)
    const int G = gridDim.x;
    const int t_begin = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int t_end = ntiles;

    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wbuf), 0, (int)P.wbytes, 0x00020000);
    const int w0_ = (int)(P.w32b_off * 4), w1_ = (int)(P.w_off[1] * 4), w2_ = (int)(P.w_off[2] * 4),
              w3_ = (int)(P.w_off[3] * 4);
    const float *b0 = P.wbuf + P.b_off[0], *b1 = P.wbuf + P.b_off[1], *b2 = P.wbuf + P.b_off[2],
                *b3 = P.wbuf + P.b_off[3];

    for (int tile = t_begin; tile < t_end; tile += G) {
        // opaque per iteration: otherwise the ~1000 scalar load offsets are hoisted out of this loop and
        // spilled to VGPR lanes
        int w0 = w0_, w1 = w1_, w2 = w2_, w3 = w3_;
        asm volatile("" : "+s"(w0), "+s"(w1), "+s"(w2), "+s"(w3));
        // mlp_base layer 0 is factorised as in the bf16x3 mode: the 224 point-only inputs were contracted once per
        // distinct neighbour point by k_point_part_f32 (pt_table row, accumulator order); here the row starts the
        // accumulators and only the 60 encoded distances (k-steps 112..143 of the lane's inputs) are multiplied
        float xq[32];
        RowCtx ctx;
        const float4 *trow;
        {
            RowFetch f;
            fetch_a<SEG>(P, tile, lane, wave, V0, S_valid, f);
            fetch_b<SEG>(P, f);
            fetch_c_pair(P, f);
            trow = P.pt_table + (int64_t)f.urow * 64 + 4 * (lane >> 5);
            const Camera cam = load_cam_lanes(P.cr, f.cid);
            pair_inputs<SEG, false>(P, f, cam, lane, xq, ctx);
        }
        float hA[128];
        dense_layer<32, 8>(rsrc, w0, b0, lane, xq, hA, trow);
#pragma unroll
        for (int i = 0; i < 128; ++i) hA[i] = leaky(hA[i]);
        float hB[132];
        {
            float tmp[128];
            dense_layer<128, 8>(rsrc, w1, b1, lane, hA, tmp);
#pragma unroll
            for (int i = 0; i < 128; ++i) hB[i] = leaky(tmp[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) hB[128 + i] = ctx.ex[i];
        dense_layer<132, 8>(rsrc, w2, b2, lane, hB, hA);
#pragma unroll
        for (int i = 0; i < 128; ++i) hA[i] = leaky(hA[i]);
        float hC[128];
        dense_layer<128, 8>(rsrc, w3, b3, lane, hA, hC);
#pragma unroll
        for (int i = 0; i < 128; ++i) hC[i] = leaky(hC[i]);
        finish_rows<SEG, false>(P, lane, hC, ctx);
    }
}

// Colour MLP: one lane-column per valid sample, 32 samples per wavefront.
// input 280 = [agg(256) | sin(view*2^f) (12) | cos(...) (12)] -> 128 -> 128 -> 128 -> 3, sigmoid, widen.
__global__ void __launch_bounds__(TPB, 1) k_shade_color(ShadeParams P)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int V0 = P.n_sel[P.i_v0], S_valid = P.n_sel[P.i_v1];
    constexpr int SPT = 32 * WAVES;
    const int ntiles = (S_valid - V0 + SPT - 1) / SPT;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wbuf), 0, (int)P.wbytes, 0x00020000);
    const int w5_ = (int)(P.w_off[5] * 4), w6_ = (int)(P.w_off[6] * 4), w7_ = (int)(P.w_off[7] * 4);
    const float *b5 = P.wbuf + P.b_off[5], *b6 = P.wbuf + P.b_off[6], *b7 = P.wbuf + P.b_off[7];

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int w5 = w5_, w6 = w6_, w7 = w7_;
        asm volatile("" : "+s"(w5), "+s"(w6), "+s"(w7));
        const int v_idx = V0 + tile * SPT + wave * 32 + j;  // colour kernels are launched with V0 = 0
        const bool ok = v_idx < S_valid;
        const int s = ok ? P.vs_list[v_idx] : 0;
        const int ray = P.smp_ray[s];
        const float *src = P.agg + (int64_t)(ok ? v_idx : 0) * 256;
        float x[140];
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const float4 a = *reinterpret_cast<const float4 *>(src + 8 * c + 4 * h);
            x[4 * c + 0] = a.x;
            x[4 * c + 1] = a.y;
            x[4 * c + 2] = a.z;
            x[4 * c + 3] = a.w;
        }
        float vx, vy, vz;
        rot_rows(P.Rw2c, P.dirs[3 * (int64_t)ray], P.dirs[3 * (int64_t)ray + 1], P.dirs[3 * (int64_t)ray + 2], vx, vy,
                 vz);
        const float vv[3] = {vx, vy, vz};
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                float sn, cs;
                sincosf(vv[d] * (float)(1 << f), &sn, &cs);
                x[128 + d * 4 + f] = h ? cs : sn;
            }
        float hA[64], hB[64];
        dense_layer<140, 4>(rsrc, w5, b5, lane, x, hA);
#pragma unroll
        for (int i = 0; i < 64; ++i) hA[i] = leaky(hA[i]);
        dense_layer<64, 4>(rsrc, w6, b6, lane, hA, hB);
#pragma unroll
        for (int i = 0; i < 64; ++i) hB[i] = leaky(hB[i]);
        dense_layer<64, 4>(rsrc, w7, b7, lane, hB, hA);
#pragma unroll
        for (int i = 0; i < 64; ++i) hA[i] = leaky(hA[i]);
        float rgb[3];
        color_head(P, lane, hA, rgb);
        if (ok && h == 0) P.smp_out[s] = make_float4(P.smp_sigma[v_idx], rgb[0], rgb[1], rgb[2]);
    }
}

// ================================================================================================
// bf16x3 mode
// ================================================================================================
constexpr int STAGE_U4 = 9 * 256;        // one LDS weight tile: up to 18 k-steps x {hi, lo} x 64 lanes x 16 B = 36 KiB
constexpr int RING = 4;                  // weight tiles in LDS: one being multiplied, up to three landed / in flight
constexpr int LDS_U4 = RING * STAGE_U4;

__device__ __forceinline__ void split8(const float *v, bf16x8 &hi, bf16x8 &lo)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 hb = (__bf16)v[j];
        hi[j] = hb;
        lo[j] = (__bf16)(v[j] - (float)hb);
    }
}

// LDS-DMA (buffer_load_dwordx4 ... lds): ROUNDS x 4 KiB of the weight tile at byte offset `off` go straight
// from L2 into ring slot `slot`, no VGPRs; each wave moves 1 KiB per instruction (lane-linear image).
template <int ROUNDS>
__device__ __forceinline__ void stage_dma(__amdgpu_buffer_rsrc_t rsrc, int off, int tid, int wave_u, u32x4 *lds,
                                          int slot)
{
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + slot * STAGE_U4 + i * 256 + wave_u * 64), 16,
                                                 tid * 16, off + i * 4096, 0, 0);
}

__device__ __forceinline__ void stage_dma_one(__amdgpu_buffer_rsrc_t rsrc, int off, int tid, int wave_u, u32x4 *lds,
                                              int slot, int i)
{
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + slot * STAGE_U4 + i * 256 + wave_u * 64), 16, tid * 16,
                                             off + i * 4096, 0, 0);
}

// per-wave state of the weight-tile ring: slot of the tile being multiplied, the A fragments of the next two
// k-steps (already read from LDS) and the bias-initialised accumulator of the next tile
struct Ring {
    int cur;
    u32x4 ah, al, bh, bl;
    f32x16 acc0;
    unsigned long long stall_bar, stall_bias;  // PNR_STAMPS builds only
};

// One dense layer on bf16 hi/lo splits.  MT output tiles of 32 features.  Weight tiles
// ([KS][{hi,lo}][64 lanes][8 bf16], KS * 2 KiB) travel L2 -> LDS by LDS-DMA into a 4-slot ring, issued three
// tiles ahead of use.  KS_NX is the k-step count of the NEXT layer's tiles (byte offset wnx, biases at
// bias_nx_off): this layer's last tiles prefetch across the layer boundary, so the k-loop of the whole MLP chain
// is one continuous stream of MFMAs.
//
// The instruction order is pinned by hand (left alone hipcc serialises `ds_read -> lgkmcnt(0) -> mfma` through
// one register quad and sinks loads down to their first use):
//   * A-operand fragments are read from LDS two k-steps ahead, ACROSS tile and layer boundaries, and the next
//     tile's accumulator is initialised from the LDS bias table during the last k-step (a VMEM bias load
//     issued behind the DMA would wait for the whole DMA: VMEM returns in order);
//   * ONE raw s_barrier per tile, in the MIDDLE of the tile: it publishes tile T+1 (whose DMA was issued two
//     tiles earlier; a COUNTED vmcnt keeps tile T+2's DMA in flight -- __syncthreads() would drain it) and
//     frees the slot of tile T-1 for the DMA of tile T+3, issued right behind it.  The fragment stream never
//     stops at a barrier;
//   * with SPLIT_OUT the activation + hi/lo split of the PREVIOUS output tile (16 values -> two k-steps of the
//     next layer's operands) is cut in three and placed BETWEEN the three MFMAs of the first 8 k-steps (an
//     in-order wave cannot issue VALU work placed behind an MFMA that waits for the matrix pipe).
// wait until at most N of the wave's vector-memory operations (LDS-DMA pieces included) are outstanding
template <int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct NoHook {
    __device__ __forceinline__ void operator()(int, int) const {}
};
// default sink of a layer without SPLIT_OUT: value r (accumulator register) of output tile `tile` -> out[tile*16 + r]
struct StoreOut {
    float *out;
    __device__ __forceinline__ void operator()(int tile, int r, float v) const { out[tile * 16 + r] = v; }
};

// `hook(m, s)` runs at the end of k-step s of tile m, inside that k-step's scheduling region: the place for loads
// that must be issued a few at a time between MFMAs (a burst of scattered loads blocks the wave at issue).
// Without SPLIT_OUT every finished accumulator value goes through `sink(tile, r, value)` (after LeakyReLU with
// OUT_LEAKY), 16 / KS values per k-step of the NEXT tile, behind that k-step's third MFMA: whatever the sink does
// runs in the MFMA shadow instead of in an epilogue.
template <int KS, int MT, int KS_NX, bool SPLIT_OUT, bool NX_BIAS = true, bool OUT_LEAKY = false,
          typename Sink = StoreOut, typename Hook = NoHook>
__device__ __forceinline__ void dense_layer_bf16(__amdgpu_buffer_rsrc_t rsrc, int wbase, int wnx,
                                                 const float *__restrict__ bias, const float *__restrict__ bias_nx,
                                                 int lane, int tid, int wave_u, u32x4 *lds,
                                                 Ring &ring, const bf16x8 *xh, const bf16x8 *xl, bf16x8 *yh,
                                                 bf16x8 *yl, Sink sink, Hook hook = Hook())
{
    static_assert(KS >= 8, "the split of the previous tile is spread over 8 k-steps");
    static_assert(MT >= 3, "the DMA runs three tiles ahead");
    const int h = lane >> 5;
    constexpr int R_SAME = (KS + 1) / 2, R_NX = (KS_NX + 1) / 2;
    f32x16 prev;
    BiasRegs breg;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const u32x4 *cur = lds + ring.cur * STAGE_U4;
        const int nxs = ring.cur + 1 >= RING ? ring.cur + 1 - RING : ring.cur + 1;
        const u32x4 *nxt = lds + nxs * STAGE_U4;
        f32x16 acc = ring.acc0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // the scalar bias loads of the next tile (issued S_BI) are awaited BEFORE this k-step's fragment reads
            // are issued: SMEM returns out of order, so the wait is lgkmcnt(0) and would otherwise expose the
            // LDS latency of the reads just issued
            constexpr int S_BF = KS - 4, S_BI = KS >= 12 ? KS - 8 : 0;
            // NX_BIAS = false: the next layer initialises its accumulators itself (pt_table rows)
            const bool want_bias = NX_BIAS || m + 1 < MT;
            if (s == S_BI && want_bias) bias_issue((m + 1 < MT) ? bias + 32 * (m + 1) : bias_nx, breg);
            if (s == S_BF && want_bias) {
                const unsigned long long tb0 = stamp();
                bias_wait(breg);
                ring.stall_bias += stamp() - tb0;
            }
            __builtin_amdgcn_sched_barrier(0);
            // fragments of k-step s+2: of this tile, or of the next tile (published by the mid-tile barrier)
            u32x4 ch, cl;
            if (s + 2 < KS) {
                ch = cur[(2 * (s + 2)) * 64 + lane];
                cl = cur[(2 * (s + 2) + 1) * 64 + lane];
            } else {
                ch = nxt[(2 * (s + 2 - KS)) * 64 + lane];
                cl = nxt[(2 * (s + 2 - KS) + 1) * 64 + lane];
            }
            const bf16x8 wh = __builtin_bit_cast(bf16x8, ring.ah);
            const bf16x8 wl = __builtin_bit_cast(bf16x8, ring.al);
            // the previous tile's accumulators are read two k-steps into this tile at the earliest: its last MFMA
            // needs ~64 cycles to retire
            constexpr int S0 = KS >= 10 ? 2 : 0;
            const bool do_split = SPLIT_OUT && m > 0 && s >= S0 && s < S0 + 8 && !(PNR_ABLATE & 16);
            const int sp = s - S0;
            float v0 = 0.f, v1 = 0.f, r0 = 0.f, r1 = 0.f;
            __bf16 h0, h1;
            if (PNR_ABLATE & 2)
                asm volatile("" ::"v"(wh), "v"(wl), "v"(xh[s]), "v"(xl[s]));
            else
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[s], acc, 0, 0, 0);
            if (do_split) {
                v0 = leaky(prev[2 * sp]);
                v1 = leaky(prev[2 * sp + 1]);
                h0 = (__bf16)v0;
                h1 = (__bf16)v1;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(PNR_ABLATE & 2)) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[s], acc, 0, 0, 0);
            if (do_split) {
                r0 = v0 - (float)h0;
                r1 = v1 - (float)h1;
            }
            if (s >= S_BF && want_bias) bias_quarter(breg, s - S_BF, ring.acc0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(PNR_ABLATE & 2)) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[s], acc, 0, 0, 0);
            if (do_split) {
                const int kk = 2 * (m - 1) + sp / 4, j0 = (2 * sp) % 8;
                yh[kk][j0] = h0;
                yh[kk][j0 + 1] = h1;
                yl[kk][j0] = (__bf16)r0;
                yl[kk][j0 + 1] = (__bf16)r1;
            }
            if (!SPLIT_OUT && m > 0) {
                // the previous tile's last MFMA was issued >= 96 cycles ago: its accumulators have retired
#pragma unroll
                for (int r = (s * 16) / KS; r < ((s + 1) * 16) / KS; ++r)
                    sink(m - 1, r, OUT_LEAKY ? leaky(prev[r]) : prev[r]);
            }
            ring.ah = ring.bh;
            ring.al = ring.bl;
            ring.bh = ch;
            ring.bl = cl;
            constexpr int S_MID = KS / 2 - 1;
            if (s == S_MID && !(PNR_ABLATE & 8)) {
                // ---- mid-tile: tile T+1 has landed everywhere, slot of tile T-1 is free ---------------------------
                if (m + 2 < MT)
                    wait_vm<R_SAME>();
                else
                    wait_vm<R_NX>();
                const unsigned long long tb0 = stamp();
                if (!(PNR_ABLATE & 64)) __builtin_amdgcn_s_barrier();
                ring.stall_bar += stamp() - tb0;
            }
            // ---- DMA of tile T+3 into the freed slot: two 1-KiB pieces per k-step behind the barrier, so the
            //      scalar address arithmetic hides between MFMAs instead of stalling the matrix pipe in one burst
            if (s > S_MID && !(PNR_ABLATE & (8 | 32))) {
                constexpr int R3 = 0;
                (void)R3;
                const int rounds = (m + 3 < MT) ? R_SAME : R_NX;
                constexpr int STEPS = KS - 1 - S_MID;               // k-steps left behind the barrier
                const int per = (rounds + STEPS - 1) / STEPS;       // pieces per k-step (1..3)
                const int first = per * (s - S_MID - 1);
                int slot3 = ring.cur + 3;
                slot3 = slot3 >= RING ? slot3 - RING : slot3;
                const int off3 = (m + 3 < MT) ? wbase + (m + 3) * KS * 2048 : wnx + (m + 3 - MT) * KS_NX * 2048;
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (q < per && first + q < rounds) stage_dma_one(rsrc, off3, tid, wave_u, lds, slot3, first + q);
            }
            hook(m, s);
            __builtin_amdgcn_sched_barrier(0);
        }
        prev = acc;
        ring.cur = nxs;
    }
    if (!SPLIT_OUT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sink(MT - 1, r, OUT_LEAKY ? leaky(prev[r]) : prev[r]);
    }
    if (SPLIT_OUT && !(PNR_ABLATE & 16)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float v0 = leaky(prev[2 * s]), v1 = leaky(prev[2 * s + 1]);
            const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
            const int kk = 2 * (MT - 1) + s / 4, j0 = (2 * s) % 8;
            yh[kk][j0] = h0;
            yh[kk][j0 + 1] = h1;
            yl[kk][j0] = (__bf16)(v0 - (float)h0);
            yl[kk][j0 + 1] = (__bf16)(v1 - (float)h1);
        }
    }
}

// The pair half of mlp_base layer 0 (bf16x3 mode): inputs [224:284] = the 60 encoded distances (4 k-steps), the
// point half W1[:, 0:224] . [emb, PE(emb)] + b1 arrives as the initial accumulator (`pin`, gathered from
// pt_table).  A ring tile holds TWO row blocks of 32 features x 4 k-steps (8 fragment pairs, 16 KiB), so that the
// barrier / DMA cadence stays at one per 8 k-steps.  The hi/lo split of row block B-1 runs between the MFMAs of
// k-steps 1..3 of block B (6 + 6 + 4 values): this layer is VALU-paced, not MFMA-paced.
template <int KS_NX>
__device__ __forceinline__ void dense_layer1b_bf16(__amdgpu_buffer_rsrc_t rsrc, int wbase, int wnx,
                                                   const float *__restrict__ bias_nx, int lane, int tid, int wave_u,
                                                   u32x4 *lds, Ring &ring, const bf16x8 *xh, const bf16x8 *xl,
                                                   const f32x16 *pin, bf16x8 *yh, bf16x8 *yl)
{
    constexpr int KS = 8, MT = 4;
    constexpr int R_SAME = 4, R_NX = (KS_NX + 1) / 2;
    f32x16 prev, acc;
    BiasRegs breg;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const u32x4 *cur = lds + ring.cur * STAGE_U4;
        const int nxs = ring.cur + 1 >= RING ? ring.cur + 1 - RING : ring.cur + 1;
        const u32x4 *nxt = lds + nxs * STAGE_U4;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int xs = s & 3, B = 2 * m + (s >> 2);
            const bool last = m + 1 == MT;
            if (last && s == 0) bias_issue(bias_nx, breg);
            if (last && s == 4) bias_wait(breg);
            __builtin_amdgcn_sched_barrier(0);
            u32x4 ch, cl;
            if (s + 2 < KS) {
                ch = cur[(2 * (s + 2)) * 64 + lane];
                cl = cur[(2 * (s + 2) + 1) * 64 + lane];
            } else {
                ch = nxt[(2 * (s + 2 - KS)) * 64 + lane];
                cl = nxt[(2 * (s + 2 - KS) + 1) * 64 + lane];
            }
            if (xs == 0) acc = pin[B];
            const bf16x8 wh = __builtin_bit_cast(bf16x8, ring.ah);
            const bf16x8 wl = __builtin_bit_cast(bf16x8, ring.al);
            // value pairs of the previous row block handled in this k-step
            const bool do_split = B > 0 && xs >= 1 && !(PNR_ABLATE & 16);
            const int p0 = 3 * (xs - 1), np = xs == 3 ? 2 : 3;
            float v0[3], v1[3], r0[3], r1[3];
            __bf16 h0[3], h1[3];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[xs], acc, 0, 0, 0);
            if (do_split) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (q < np) {
                        v0[q] = leaky(prev[2 * (p0 + q)]);
                        v1[q] = leaky(prev[2 * (p0 + q) + 1]);
                        h0[q] = (__bf16)v0[q];
                        h1[q] = (__bf16)v1[q];
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[xs], acc, 0, 0, 0);
            if (do_split) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (q < np) {
                        r0[q] = v0[q] - (float)h0[q];
                        r1[q] = v1[q] - (float)h1[q];
                    }
            }
            if (last && s >= 4) bias_quarter(breg, s - 4, ring.acc0);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[xs], acc, 0, 0, 0);
            if (do_split) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (q < np) {
                        const int sp = p0 + q;
                        const int kk = 2 * (B - 1) + sp / 4, j0 = (2 * sp) % 8;
                        yh[kk][j0] = h0[q];
                        yh[kk][j0 + 1] = h1[q];
                        yl[kk][j0] = (__bf16)r0[q];
                        yl[kk][j0 + 1] = (__bf16)r1[q];
                    }
            }
            ring.ah = ring.bh;
            ring.al = ring.bl;
            ring.bh = ch;
            ring.bl = cl;
            if (s == 3 && !(PNR_ABLATE & 8)) {
                if (m + 2 < MT)
                    wait_vm<R_SAME>();
                else
                    wait_vm<R_NX>();
                if (!(PNR_ABLATE & 64)) __builtin_amdgcn_s_barrier();
            }
            if (s > 3 && !(PNR_ABLATE & (8 | 32))) {
                const int rounds = (m + 3 < MT) ? R_SAME : R_NX;
                const int per = (rounds + 3) / 4;  // pieces per k-step (1..3)
                const int first = per * (s - 4);
                int slot3 = ring.cur + 3;
                slot3 = slot3 >= RING ? slot3 - RING : slot3;
                const int off3 = (m + 3 < MT) ? wbase + (m + 3) * KS * 2048 : wnx + (m + 3 - MT) * KS_NX * 2048;
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (q < per && first + q < rounds) stage_dma_one(rsrc, off3, tid, wave_u, lds, slot3, first + q);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (xs == 3) prev = acc;
        }
        ring.cur = nxs;
    }
    if (!(PNR_ABLATE & 16)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float v0 = leaky(prev[2 * s]), v1 = leaky(prev[2 * s + 1]);
            const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
            const int kk = 2 * 7 + s / 4, j0 = (2 * s) % 8;
            yh[kk][j0] = h0;
            yh[kk][j0 + 1] = h1;
            yl[kk][j0] = (__bf16)(v0 - (float)h0);
            yl[kk][j0 + 1] = (__bf16)(v1 - (float)h1);
        }
    }
}

// first three tiles of a chain into slots 0..2, fragments of k-steps 0 and 1 and the first accumulator
template <int KS0>
__device__ __forceinline__ void ring_start(__amdgpu_buffer_rsrc_t rsrc, int w_first, const float *__restrict__ bias0,
                                           int lane, int tid, int wave_u, u32x4 *lds, Ring &ring)
{
    constexpr int R0 = (KS0 + 1) / 2;
    stage_dma<R0>(rsrc, w_first, tid, wave_u, lds, 0);
    stage_dma<R0>(rsrc, w_first + KS0 * 2048, tid, wave_u, lds, 1);
    stage_dma<R0>(rsrc, w_first + 2 * KS0 * 2048, tid, wave_u, lds, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ring.cur = 0;
    ring.ah = lds[0 * 64 + lane];
    ring.al = lds[1 * 64 + lane];
    ring.bh = lds[2 * 64 + lane];
    ring.bl = lds[3 * 64 + lane];
    if (bias0) {
        BiasRegs breg;
        bias_issue(bias0, breg);
        ring.acc0 = bias_finish(breg, lane >> 5);
    }
}

// fp32 mode of k_point_part below: the same table with v_mfma_f32_32x32x2_f32 (exact fp32), exact sincosf encodings
__global__ void __launch_bounds__(TPB, 1) k_point_part_f32(ShadeParams P)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int U = min(P.n_sel[3], P.u_cap);
    constexpr int PPT = 32 * WAVES;
    const int ntiles = (U + PPT - 1) / PPT;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wbuf), 0, (int)P.wbytes, 0x00020000);
    const int wa_ = (int)(P.w32a_off * 4);
    const float *b0 = P.wbuf + P.b_off[0];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int wa = wa_;
        asm volatile("" : "+s"(wa));
        const int u = tile * PPT + wave * 32 + j;
        const int pidx = P.pt_list[u < U ? u : 0];
        const float4 *row = P.point_rows + (int64_t)pidx * 12 + 4 + 4 * h;
        const float4 e0 = row[0], e1 = row[1], e2 = row[2], e3 = row[3];
        const float e[16] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w,
                             e2.x, e2.y, e2.z, e2.w, e3.x, e3.y, e3.z, e3.w};
        float x0[112];
        point_inputs<false>(e, x0);
        float o[128];
        dense_layer<112, 8>(rsrc, wa, b0, lane, x0, o);
        float4 *dst = P.pt_table + (int64_t)u * 64 + 4 * h;   // rows beyond U: the table's padding rows
#pragma unroll
        for (int B = 0; B < 8; ++B)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[8 * B + q] = make_float4(o[16 * B + 4 * q], o[16 * B + 4 * q + 1], o[16 * B + 4 * q + 2],
                                             o[16 * B + 4 * q + 3]);
    }
}

// Point half of mlp_base layer 0 for the U distinct neighbour points of the call (bf16x3 mode):
//   pt_table[u] = W1[:, 0:224] . [emb_u, PE(emb_u, 3)] + b1          (studio_model.py:309-317, inputs [0:224])
// The 224 point-only inputs of the 284 are the same for every sample that has the point as a neighbour (~10 pairs
// per point and frame at BASELINE configs[1]), so this contraction is done once per point and call instead of
// once per pair; k_shade_pairs_bf16 starts its first layer from the gathered row and multiplies only the 60
// encoded distances.  Rows are stored in accumulator order [row block][lane half][16] so that a lane picks up its
// 16 values of a row block with four 16-byte loads.  One wave = 32 points on the MFMA columns, as in the pair kernel.
__global__ void __launch_bounds__(TPB, 1) k_point_part(ShadeParams P)
{
    __shared__ u32x4 lds[LDS_U4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int j = lane & 31, h = lane >> 5;
    const int U = min(P.n_sel[3], P.u_cap);
    constexpr int PPT = 32 * WAVES;
    const int ntiles = (U + PPT - 1) / PPT;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wbuf), 0, (int)P.wbytes, 0x00020000);
    const int wa_ = (int)(P.w16a_off * 4);
    if ((int)blockIdx.x >= ntiles) return;
    const float *__restrict__ b0 = P.wbuf + P.b_off[0];
    Ring ring;
    ring.stall_bar = 0;
    ring.stall_bias = 0;
    ring_start<14>(rsrc, wa_, b0, lane, tid, wave_u, lds, ring);
    // embeddings of the first tile; those of the next tile are fetched while this one is multiplied
    float4 en[4];
    int pidx_nx;
    {
        const int u0 = blockIdx.x * PPT + wave * 32 + j;
        const float4 *row = P.point_rows + (int64_t)P.pt_list[u0 < U ? u0 : 0] * 12 + 4 + 4 * h;
#pragma unroll
        for (int i = 0; i < 4; ++i) en[i] = row[i];
        const int u1 = u0 + (int)gridDim.x * PPT;
        pidx_nx = P.pt_list[u1 < U ? u1 : 0];
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int wa = wa_;
        asm volatile("" : "+s"(wa));
        const int u = tile * PPT + wave * 32 + j;
        const bool ok = u < U;
        const float4 e0 = en[0], e1 = en[1], e2 = en[2], e3 = en[3];
        {
            const float4 *row = P.point_rows + (int64_t)pidx_nx * 12 + 4 + 4 * h;
#pragma unroll
            for (int i = 0; i < 4; ++i) en[i] = row[i];
            const int u2 = u + 2 * (int)gridDim.x * PPT;
            pidx_nx = P.pt_list[u2 < U ? u2 : 0];
        }
        bf16x8 xh[14], xl[14];
        {
            const float e[16] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w,
                                 e2.x, e2.y, e2.z, e2.w, e3.x, e3.y, e3.z, e3.w};
            float x0[112];
            point_inputs<true>(e, x0);
#pragma unroll
            for (int s = 0; s < 14; ++s) split8(&x0[8 * s], xh[s], xl[s]);
        }
        // Rows leave through the layer's sink, one 16-byte store per four finished values, between the MFMAs of the
        // following output tile: a burst of 32 scattered stores per lane behind the layer kept the texture-address
        // unit busy for as long as the layer's MFMAs take, and the next tile's first counted vmcnt waited for them
        // (vector memory retires in order).  Lanes beyond U write into the table's 128 padding rows (no branch:
        // a branch inside the layer would split its basic block).
        float4 *dst = P.pt_table + (int64_t)u * 64 + 4 * h;
        float q4[3];
        auto sink = [&](int t, int r, float v) {
            if ((r & 3) < 3)
                q4[r & 3] = v;
            else
                dst[8 * t + (r >> 2)] = make_float4(q4[0], q4[1], q4[2], v);
        };
        (void)ok;
        dense_layer_bf16<14, 8, 14, false>(rsrc, wa, wa, b0, b0, lane, tid, wave_u, lds, ring, xh, xl, nullptr, nullptr,
                                           sink);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int SEG>
__global__ void __launch_bounds__(TPB, 1) k_shade_pairs_bf16(ShadeParams P)
{
    __shared__ u32x4 lds[LDS_U4 + 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int SPT = (32 / seg_len<SEG>(P.K)) * WAVES;
    const int V0 = P.n_sel[P.i_v0], S_valid = P.n_sel[P.i_v1];
    const int ntiles = (S_valid - V0 + SPT - 1) / SPT;
    // XCD-aware tile order.  Workgroup b runs on XCD b % 8 (round-robin dispatch), one workgroup per CU.  In every
    // round of gridDim.x tiles XCD x takes the 32 CONSECUTIVE tiles [x * G/8, (x+1) * G/8) of the round, one per
    // CU: the CUs behind one L2 then work on ~45 neighbouring rays at the same time, and the pt_table / point rows
    // those rays share (a point serves ~7 pairs) are fetched into that L2 once instead of once per pair.  With a
    // contiguous tile range per workgroup the 32 CUs of an XCD stream 4 MB of unrelated rows through the 4 MB L2 per
    // tile time and nearly every gather misses (rocprofv3 FETCH_SIZE: 14.9 GB per launch for 10.8 GB gathered).
#ifdef PNR_AB_CONTIG_TILES  // diagnostic A/B builds only: one contiguous tile range per workgroup
    const int G = 1;
    const int t_begin = (int)(((int64_t)ntiles * blockIdx.x) / gridDim.x);
    const int t_end = (int)(((int64_t)ntiles * (blockIdx.x + 1)) / gridDim.x);
#else
    const int G = gridDim.x;
    const int pos = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int t_begin = pos, t_end = ntiles;
#endif
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wbuf), 0, (int)P.wbytes, 0x00020000);
    const int wb_ = (int)(P.w16b_off * 4), w1_ = (int)(P.w16_off[1] * 4), w2_ = (int)(P.w16_off[2] * 4),
              w3_ = (int)(P.w16_off[3] * 4);
    if (t_begin >= t_end) return;  // uniform per workgroup
    const float *__restrict__ b1 = P.wbuf + P.b_off[1];
    const float *__restrict__ b2 = P.wbuf + P.b_off[2];
    const float *__restrict__ b3 = P.wbuf + P.b_off[3];
    // density-head weights in accumulator order (pnr_weights_pack: w4acc[(t * 2 + h) * 16 + r] =
    // w4[32t + 8(r>>2) + 4h + (r&3)]) behind the ring, fetched by LDS-DMA like everything else in LDS: ONE plain LDS
    // store anywhere in the kernel makes hipcc guard every fragment read with s_waitcnt vmcnt(0)
    u32x4 *w4tab = lds + LDS_U4;
    if (wave_u == 0) {
        typedef __attribute__((address_space(3))) void *lds_ptr_t;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)w4tab, 16, lane * 16, (int)(P.w4acc_off * 4), 0, 0);
    }
    const float b4 = P.wbuf[P.b_off[4]];
    Ring ring;
    ring.stall_bar = 0;
    ring.stall_bias = 0;
    ring_start<8>(rsrc, wb_, nullptr, lane, tid, wave_u, lds, ring);
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    RowFetch cur, nxt;
    fetch_a<SEG>(P, t_begin, lane, wave, V0, S_valid, cur);
    fetch_b<SEG>(P, cur);
    fetch_c_pair(P, cur);
    for (int tile = t_begin; tile < t_end; tile += G) {
        int wb = wb_, w1 = w1_, w2 = w2_, w3 = w3_;
        asm volatile("" : "+s"(wb), "+s"(w1), "+s"(w2), "+s"(w3));
        const unsigned long long ts0 = stamp();
        const Camera cam = load_cam_wave(P.cr, cur.cid);
        // first level of the next tile's gather chain (a tile past the end loads row 0: harmless); the other two
        // levels follow at the layer boundaries
        fetch_a<SEG>(P, tile + G, lane, wave, V0, S_valid, nxt);
        __builtin_amdgcn_sched_barrier(0);
        // Point halves of layer 1 (pt_table rows, accumulator order).  A lane reads 512 B in 32 scattered 16-byte
        // loads; 4 waves x 32 of them keep the CU's texture-address unit busy for ~7k cycles (tools/ub_gather.hip)
        // and block the issuing wave meanwhile, wherever they are issued: spreading half of them between the MFMAs of
        // the previous tile's last layer moved the cost there, cycle for cycle.  They land while the distances are
        // encoded.
        f32x16 pin[8];
        {
            const float4 *trow = P.pt_table + (int64_t)cur.urow * 64 + 4 * (lane >> 5);
#pragma unroll
            for (int B = 0; B < 8; ++B)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = trow[8 * B + q];
                    pin[B][4 * q] = v.x;
                    pin[B][4 * q + 1] = v.y;
                    pin[B][4 * q + 2] = v.z;
                    pin[B][4 * q + 3] = v.w;
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long tsg = stamp();
        RowCtx ctx;
        bf16x8 xqh[4], xql[4];
        {
            float xq[32];
            pair_inputs<SEG, true>(P, cur, cam, lane, xq, ctx);
#pragma unroll
            for (int s = 0; s < 4; ++s) split8(&xq[8 * s], xqh[s], xql[s]);
        }
        const unsigned long long ts1 = stamp();
        bf16x8 xh[17], xl[17], yh[17], yl[17];
        dense_layer1b_bf16<16>(rsrc, wb, w1, b1, lane, tid, wave_u, lds, ring, xqh, xql, pin, yh, yl);
        const unsigned long long ts2 = stamp();
        fetch_b<SEG>(P, nxt);
        // layer 2's output (+ the 7 extra head inputs as k-step 16) goes to xh/xl
        dense_layer_bf16<16, 8, 17, true>(rsrc, w1, w2, b1, b2, lane, tid, wave_u, lds, ring, yh, yl, xh, xl, StoreOut{nullptr});
        const unsigned long long ts3 = stamp();
        fetch_c_pair(P, nxt);
        {
            float v[8] = {ctx.ex[0], ctx.ex[1], ctx.ex[2], ctx.ex[3], 0.f, 0.f, 0.f, 0.f};
            split8(v, xh[16], xl[16]);
        }
        dense_layer_bf16<17, 8, 16, true>(rsrc, w2, w3, b2, b3, lane, tid, wave_u, lds, ring, xh, xl, yh, yl, StoreOut{nullptr});
        const unsigned long long ts4 = stamp();
        // Last layer.  The chain wraps around: the next pair tile starts again with the pair half of layer 0, whose
        // accumulators come from pt_table (no bias prefetch).
        if (SEG != 0) {
            // K <= 16: density head and K-aggregation (studio_model.py:337-353) run inside the layer, one finished
            // value per k-step in the MFMA shadow.  Value r of output tile t (feature 32t + 8(r>>2) + 4h + (r&3)) is
            // multiplied with its head weight (w4tab: accumulator order, 16 per lane half and tile, fetched four at a
            // time three k-steps ahead), weighted, summed over the lanes of the sample's segment, and kept by the lane
            // whose slot equals t: afterwards lanes 0..7 of a segment store one 32-feature tile each of the sample's 256.
            constexpr int NS = SEG == 16 ? 4 : 3;   // DPP steps of the segment sum
            float part = 0.f;
            float mine[16];
            f32x4 wv[4];
            // (inline asm: hipcc guards a plain LDS read with s_waitcnt vmcnt(0) while LDS-DMA is in flight, which
            // would drain the weight pipeline four times per tile.  The read is consumed 13 k-steps = 26 younger
            // fragment reads later; LDS returns in order and every fragment read is awaited by the compiler.)
            const unsigned w4a = (unsigned)(uintptr_t)w4tab + 64u * (lane >> 5);  // LDS byte address
            auto hook = [&](int m, int s) {
                if ((s & 3) == 3)
                    asm volatile("ds_read_b128 %0, %1 offset:%2"
                                 : "=v"(wv[s >> 2])
                                 : "v"(w4a), "n"(128 * m + 16 * (s >> 2)));
            };
            // The three DPP steps of the 8-lane sum form a pipeline over consecutive values (p1..p3): a DPP operand
            // written by the instruction just before it costs two wait states (s_nop), here every DPP reads a
            // register written one k-step earlier.
            float p1 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f;
            auto stage = [&](int L) {  // value L leaves the pipeline
                const float a = SEG == 16 ? dpp_add<0x140>(p4) : dpp_add<0x141>(p3);
                if (L >= 0) mine[L & 15] = (ctx.slot == (L >> 4)) ? a : ((L >> 4) == 0 ? 0.f : mine[L & 15]);
                if (SEG == 16) p4 = dpp_add<0x141>(p3);
                p3 = dpp_add<0x4E>(p2);
                p2 = dpp_add<0xB1>(p1);
            };
            auto sink = [&](int t, int r, float v) {
                const f32x4 w = wv[r >> 2];
                const float wr = (r & 3) == 0 ? w.x : (r & 3) == 1 ? w.y : (r & 3) == 2 ? w.z : w.w;
                part += v * wr;
                stage(16 * t + r - NS);
                p1 = v * ctx.wgt;
                // (an opaque use: hipcc otherwise sinks the whole chain into the block of the stores behind the layer)
                asm volatile("" : "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(part));
                if (16 * t + r >= NS) asm volatile("" : "+v"(mine[(16 * t + r - NS) & 15]));
            };
            dense_layer_bf16<16, 8, 8, false, false, true>(rsrc, w3, wb, b3, nullptr, lane, tid, wave_u, lds, ring, yh,
                                                           yl, nullptr, nullptr, sink, hook);
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                stage(128 - NS + i);
                p1 = 0.f;
            }
            const unsigned long long ts5 = stamp();
            ph[4] += ts5 - ts4;
            // Retire the next tile's prefetched loads HERE, ahead of the stores (vector memory returns in order and
            // the last of them was issued a layer ago: the wait is free).  Left to the first use at the top of the
            // next iteration, hipcc waits vmcnt(0) across the back edge: for the stores just issued.
            asm volatile("" ::"v"(nxt.dirz), "v"(nxt.urow));
            part += __shfl_xor(part, 32, 64);
            const float alpha = fmaxf(part + b4, 0.f);
            const float sigma = seg_sum<SEG>(alpha * ctx.wgt, P.K, lane);
            if (ctx.row_ok && ctx.slot == 0 && lane < 32) {
                P.smp_sigma[ctx.v_idx] = sigma;
                if (P.smp_sig_s) P.smp_sig_s[ctx.s] = sigma;
            }
            if (ctx.smp_ok && ctx.slot < 8) {   // (idle lanes of the segment store too: K < 8 leaves slots K..7 idle)
                float4 *agg4 = reinterpret_cast<float4 *>(P.agg);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    agg4[agg_idx4(ctx.v_idx, 2 * ctx.slot + (q >> 1), lane >> 5, q & 1)] =
                        make_float4(mine[4 * q], mine[4 * q + 1], mine[4 * q + 2], mine[4 * q + 3]);
            }
            ph[5] += stamp() - ts5;
        } else {
            float o[128];
            dense_layer_bf16<16, 8, 8, false, false, true>(rsrc, w3, wb, b3, nullptr, lane, tid, wave_u, lds, ring, yh,
                                                           yl, nullptr, nullptr, StoreOut{o});
            const unsigned long long ts5 = stamp();
            ph[4] += ts5 - ts4;
            asm volatile("" ::"v"(nxt.dirz), "v"(nxt.urow));
            finish_rows<SEG, true>(P, lane, o, ctx);  // o: LeakyReLU already applied inside the layer
            ph[5] += stamp() - ts5;
        }
        ph[0] += ts1 - ts0;
        ph[1] += ts2 - ts1;
        ph[2] += ts3 - ts2;
        ph[3] += ts4 - ts3;
        ph[7] += 1;
        ph[6] += tsg - ts0;  // (PNR_STAMPS builds) issue time of the gathers; ring.stall_bar holds the barrier stalls
        cur = nxt;
    }
#if PNR_STAMPS
    if (lane == 0) {
        // debug tail of the sigma buffer: [cap - 8192 .. cap) floats hold 8 x u64 per wave for the first 512 waves
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(P.smp_sigma + P.dbg_off);
        const int wid = blockIdx.x * WAVES + wave;
        if (wid < 256)
            for (int i = 0; i < 8; ++i) dbg[wid * 8 + i] = ph[i];
    }
#endif
    // the two tiles prefetched for a pair tile that does not exist are simply dropped
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void __launch_bounds__(TPB, 1) k_shade_color_bf16(ShadeParams P)
{
    __shared__ u32x4 lds[LDS_U4 + 128];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int j = lane & 31, h = lane >> 5;
    const int V0 = P.n_sel[P.i_v0], S_valid = P.n_sel[P.i_v1];
    constexpr int SPT = 32 * WAVES;
    const int ntiles = (S_valid - V0 + SPT - 1) / SPT;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wbuf), 0, (int)P.wbytes, 0x00020000);
    const int w5_ = (int)(P.w16_off[5] * 4), w6_ = (int)(P.w16_off[6] * 4), w7_ = (int)(P.w16_off[7] * 4);
    if ((int)blockIdx.x >= ntiles) return;
    const float *__restrict__ b5 = P.wbuf + P.b_off[5];
    const float *__restrict__ b6 = P.wbuf + P.b_off[6];
    const float *__restrict__ b7 = P.wbuf + P.b_off[7];
    // colour head weights (3 x 128) in accumulator order behind the ring, by LDS-DMA (see w4tab in the pair kernel)
    u32x4 *w8tab = lds + LDS_U4;
    if (wave_u < 2) {
        typedef __attribute__((address_space(3))) void *lds_ptr_t;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(w8tab + wave_u * 64), 16, tid * 16,
                                                 (int)(P.w8acc_off * 4), 0, 0);
    }
    Ring ring;
    ring_start<18>(rsrc, w5_, b5, lane, tid, wave_u, lds, ring);
    const float4 *agg4 = reinterpret_cast<const float4 *>(P.agg);
    // loads behind the last MFMA of a tile would each cost a full vmcnt(0) round trip: head biases once, the sample's
    // density with the tile's other loads
    const float b8[3] = {P.wbuf[P.b_off[8]], P.wbuf[P.b_off[8] + 1], P.wbuf[P.b_off[8] + 2]};
    // sample -> ray -> direction of the first tile; the chain of the next tile is issued while this one computes
    int s_nx, ray_nx;
    float dnx[3];
    {
        const int v0 = V0 + blockIdx.x * SPT + wave * 32 + j;
        s_nx = P.vs_list[v0 < S_valid ? v0 : 0];
        ray_nx = P.smp_ray[s_nx];
#pragma unroll
        for (int d = 0; d < 3; ++d) dnx[d] = P.dirs[3 * (int64_t)ray_nx + d];
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int w5 = w5_, w6 = w6_, w7 = w7_;
        asm volatile("" : "+s"(w5), "+s"(w6), "+s"(w7));
        const int v_idx = V0 + tile * SPT + wave * 32 + j;  // colour kernels are launched with V0 = 0
        const bool ok = v_idx < S_valid;
        const int s = s_nx;
        const float sigma = P.smp_sigma[ok ? v_idx : 0];
        const float dir[3] = {dnx[0], dnx[1], dnx[2]};
        const int v_nx = v_idx + (int)gridDim.x * SPT;
        s_nx = P.vs_list[v_nx < S_valid ? v_nx : 0];
        bf16x8 xh[18], xl[18];
        const int64_t a_base = (((int64_t)(tile * WAVES + wave) * 16) * 2) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            // k-step k: this lane half supplies features 16k + 8h .. 16k + 8h + 7 (agg_idx4: one contiguous KiB per load)
            const float4 a = agg4[a_base + (2 * k) * 64];
            const float4 b = agg4[a_base + (2 * k + 1) * 64];
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            split8(v, xh[k], xl[k]);
        }
        ray_nx = P.smp_ray[s_nx];
        {
            float vx, vy, vz;
            rot_rows(P.Rw2c, dir[0], dir[1], dir[2], vx, vy, vz);
            // encoded view direction, input order [sin(d*4+f) (12) | cos (12)]: k-step 16 = values 0..7 (h = 0) /
            // 8..15 (h = 1), k-step 17 = values 16..23 (h = 0) / zero.  Selected value by value between scalars: a
            // select between two elements of one array becomes an indexed read of the array through SCRATCH, whose
            // s_waitcnt vmcnt(0) also drains the weight DMA in flight
            float sn0, cs0, sn1, cs1, sn2, cs2;
            fast_sincos(vx, sn0, cs0);
            fast_sincos(vy, sn1, cs1);
            fast_sincos(vz, sn2, cs2);
            float v16[8], v17[8];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (f > 0) {
                    const float a0 = 2.0f * sn0 * cs0, b0 = (cs0 - sn0) * (cs0 + sn0);
                    const float a1 = 2.0f * sn1 * cs1, b1 = (cs1 - sn1) * (cs1 + sn1);
                    const float a2 = 2.0f * sn2 * cs2, b2 = (cs2 - sn2) * (cs2 + sn2);
                    sn0 = a0, cs0 = b0, sn1 = a1, cs1 = b1, sn2 = a2, cs2 = b2;
                }
                v16[f] = h ? sn2 : sn0;        // values 8 + f (sin of component 2) / f (sin of component 0)
                v16[4 + f] = h ? cs0 : sn1;    // values 12 + f (cos of component 0) / 4 + f (sin of component 1)
                v17[f] = h ? 0.f : cs1;        // values 16 + f (cos of component 1)
                v17[4 + f] = h ? 0.f : cs2;    // values 20 + f (cos of component 2)
            }
            split8(v16, xh[16], xl[16]);
            split8(v17, xh[17], xl[17]);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) dnx[d] = P.dirs[3 * (int64_t)ray_nx + d];
        bf16x8 yh[8], yl[8];
        dense_layer_bf16<18, 4, 8, true>(rsrc, w5, w6, b5, b6, lane, tid, wave_u, lds, ring, xh, xl, yh, yl, StoreOut{nullptr});
        dense_layer_bf16<8, 4, 8, true>(rsrc, w6, w7, b6, b7, lane, tid, wave_u, lds, ring, yh, yl, xh, xl, StoreOut{nullptr});
        float o[64];
        dense_layer_bf16<8, 4, 18, false, true, true>(rsrc, w7, w5, b7, b5, lane, tid, wave_u, lds, ring, xh, xl, nullptr,
                                                      nullptr, StoreOut{o});
        float rgb[3];
        color_head_lds(b8, w8tab, lane, o, rgb);
        if (ok && h == 0) P.smp_out[s] = make_float4(sigma, rgb[0], rgb[1], rgb[2]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// weight packing: PyTorch [out,in] -> MFMA A-operand order
// ------------------------------------------------------------------------------------------------
enum LayerKind { L_BASE0 = 0, L_HIDDEN = 1, L_HEAD0 = 2, L_COLOR0 = 3 };

__device__ __forceinline__ int hidden_feat(int t, int h)
{
    const int m = t >> 4, r = t & 15;
    return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// fp32 path: input feature of k-step t for lane half h
__device__ int feat_of(int kind, int t, int h, int n_in)
{
    switch (kind) {
    case L_BASE0:
        if (t < 16) return 16 * h + t;
        if (t < 112) {
            const int u = t - 16, sc = u & 1, df = u >> 1, d = df / 3 + 16 * h, f = df % 3;
            return 32 + 2 * (d * 3 + f) + sc;
        }
        if (t < 142) {
            const int u = t - 112, sc = u & 1, df = u >> 1, d = df / 5 + 3 * h, f = df % 5;
            return 224 + 2 * (d * 5 + f) + sc;
        }
        return -1;
    case L_HIDDEN:
        return t < n_in / 2 ? hidden_feat(t, h) : -1;
    case L_HEAD0:
        if (t < 128) return hidden_feat(t, h);
        if (t == 128) return h ? 257 : 256;
        if (t == 129) return h ? 259 : 258;
        if (t == 130) return h ? 261 : 260;
        if (t == 131) return h ? -1 : 262;
        return -1;
    case L_COLOR0:
        if (t < 128) return 8 * (t >> 2) + 4 * h + (t & 3);
        if (t < 140) return (h ? 268 : 256) + (t - 128);
        return -1;
    }
    return -1;
}

// bf16 path: input feature of element j of k-step s (16 features) for lane half h.  Hidden layers: the
// accumulator registers 8s'..8s'+7 of output tile m ARE k-step 2m+s' (row = 16s' + 8(j>>2) + 4h + (j&3)).
__device__ int feat16_of(int kind, int s, int h, int j, int n_in)
{
    const int hid = 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
    switch (kind) {
    case L_BASE0:
        return feat_of(L_BASE0, 8 * s + j, h, n_in);  // the lane's value i = 8s + j, same order as the fp32 path
    case L_HIDDEN:
        return s < n_in / 16 ? hid : -1;
    case L_HEAD0:
        if (s < 16) return hid;
        if (s == 16 && j < 4) {
            const int lo[4] = {256, 258, 260, 262}, hi[4] = {257, 259, 261, -1};
            return h ? hi[j] : lo[j];
        }
        return -1;
    case L_COLOR0: {
        if (s < 16) return 16 * s + 8 * h + j;
        const int f = 256 + (s - 16) * 16 + 8 * h + j;
        return f < 280 ? f : -1;
    }
    }
    return -1;
}

__global__ void k_pack_layer(const float *__restrict__ W, int n_out, int n_in, int kind, int ksp, int t0,
                             float *__restrict__ dst)
{
    // dst[((m*KG + g)*64 + lane)*4 + q] = W[32m + (lane&31)][feat(t0 + 4g+q, lane>>5)]
    const int kg = ksp / 4, mt = n_out / 32;
    const int64_t total = (int64_t)mt * kg * 64 * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i & 3), lane = (int)((i >> 2) & 63);
        const int64_t mg = i >> 8;
        const int g = (int)(mg % kg), m = (int)(mg / kg);
        const int f = feat_of(kind, t0 + 4 * g + q, lane >> 5, n_in);
        dst[i] = (f >= 0 && f < n_in) ? W[(int64_t)(32 * m + (lane & 31)) * n_in + f] : 0.f;
    }
}

__global__ void k_pack_layer_bf16(const float *__restrict__ W, int n_out, int n_in, int kind, int ks, int kx, int s0,
                                  unsigned short *__restrict__ dst)
{
    // dst[(((m*KS + s)*2 + plane)*64 + lane)*8 + j] = {hi, lo}(W[32 rb + (lane&31)][feat16(s0 + s % kx, lane>>5, j)])
    // a ring tile m holds ks / kx row blocks rb = m * (ks / kx) + s / kx of kx k-steps each (kx == ks: one)
    const int mt = n_out / 32 / (ks / kx);
    const int64_t total = (int64_t)mt * ks * 64 * 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const int64_t ms = i >> 9;
        const int s = (int)(ms % ks), m = (int)(ms / ks);
        const int f = feat16_of(kind, s0 + s % kx, lane >> 5, j, n_in);
        const int rb = m * (ks / kx) + s / kx;
        const float w = (f >= 0 && f < n_in) ? W[(int64_t)(32 * rb + (lane & 31)) * n_in + f] : 0.f;
        const __bf16 hb = (__bf16)w;
        const __bf16 lb = (__bf16)(w - (float)hb);
        const int64_t base = ((ms * 2) * 64 + lane) * 8 + j;
        dst[base] = __builtin_bit_cast(unsigned short, hb);
        dst[base + 64 * 8] = __builtin_bit_cast(unsigned short, lb);
    }
}

// density head weights in accumulator order: dst[(t * 2 + h) * 16 + r] = w4[32t + 8(r>>2) + 4h + (r&3)]
__global__ void k_pack_head_acc(const float *__restrict__ w4, float *__restrict__ dst)
{
    const int i = threadIdx.x, t = i >> 5, h = (i >> 4) & 1, r = i & 15;
    dst[i] = w4[32 * t + 8 * (r >> 2) + 4 * h + (r & 3)];
}

// colour head weights [3,128] in accumulator order: dst[((c * 4 + t) * 2 + h) * 16 + r] = w8[c][32t + 8(r>>2) + 4h + (r&3)]
__global__ void k_pack_color_head_acc(const float *__restrict__ w8, float *__restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 384) return;
    const int c = i >> 7, t = (i >> 5) & 3, h = (i >> 4) & 1, r = i & 15;
    dst[i] = w8[c * 128 + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3)];
}

__global__ void k_copy(const float *__restrict__ src, int n, float *__restrict__ dst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// early ray termination (opts.early_stop_eps > 0): samples are shaded front to back in chunks of a ray's sample
// index; between the chunks a ray whose transmittance fell below eps leaves.  Everything stays on the device.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) k_pass_init(int64_t R, float *__restrict__ ray_T, int *__restrict__ ray_alive,
                                                   int *__restrict__ n_sel)
{
    const int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (r == 0) {
        n_sel[4] = 0;
        n_sel[5] = 0;
        n_sel[6] = 0;
    }
    if (r >= R) return;
    ray_T[r] = 1.0f;
    ray_alive[r] = 1;
}

// flag[v] = valid sample v belongs to a live ray and its index inside the ray is in [lo, hi)
__global__ void __launch_bounds__(TPB) k_pass_flag(const int *__restrict__ n_sel, const int *__restrict__ vs_list,
                                                   const int *__restrict__ smp_ray, const int *__restrict__ ray_off,
                                                   const int *__restrict__ ray_alive, int lo, int hi,
                                                   int *__restrict__ flag)
{
    const int S_valid = n_sel[1];
    for (int64_t v = (int64_t)blockIdx.x * TPB + threadIdx.x; v < S_valid; v += (int64_t)gridDim.x * TPB) {
        const int s = vs_list[v];
        const int r = smp_ray[s];
        const int i = s - ray_off[r];
        flag[v] = (ray_alive[r] && i >= lo && i < hi) ? 1 : 0;
    }
}

__global__ void __launch_bounds__(TPB) k_pass_scatter(const int *__restrict__ n_sel, const int *__restrict__ vs_list,
                                                      const int *__restrict__ flag, const int *__restrict__ pos,
                                                      int *__restrict__ vs_all)
{
    const int S_valid = n_sel[1], base = n_sel[5];
    for (int64_t v = (int64_t)blockIdx.x * TPB + threadIdx.x; v < S_valid; v += (int64_t)gridDim.x * TPB)
        if (flag[v]) vs_all[base + pos[v]] = vs_list[v];
}

// the pass just listed occupies positions [n_sel[4], n_sel[5]) of vs_all
__global__ void k_pass_advance(int *__restrict__ n_sel, const int *__restrict__ pos)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int count = pos[n_sel[1]];
        n_sel[4] = n_sel[5];
        n_sel[5] = n_sel[5] + count;
    }
}

// transmittance of the live rays after samples [lo, hi): the composite's own arithmetic (k_composite, same
// ray_dist quirks), so that "T < eps" means what it means there
__global__ void __launch_bounds__(TPB) k_pass_update(CamRef cr, pnr_render_opts_t opts, int64_t R,
                                                     const int *__restrict__ ray_cnt, const int *__restrict__ ray_off,
                                                     const float4 *__restrict__ smp_loc,
                                                     const float *__restrict__ smp_sig_s, const int *__restrict__ n_sel,
                                                     int lo, int hi, float *__restrict__ ray_T,
                                                     float *__restrict__ ray_cm, int *__restrict__ ray_alive)
{
    const int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (r >= R || !ray_alive[r]) return;
    const int S = n_sel[0];
    const int off = ray_off[r];
    int cnt = ray_cnt[r];
    if ((int64_t)off + cnt > S) cnt = max(0, S - off);
    if (lo >= cnt) {
        ray_alive[r] = 0;
        return;
    }
    const Camera cam = load_cam_lanes(cr, cam_id(cr, r));
    const float vs = opts.vsize_z, two_vs = 2.0f * vs;
    auto zc = [&](float x, float y, float z) {
        const float sx = x - cam.o[0], sy = y - cam.o[1], sz = z - cam.o[2];
        return sx * cam.R[2] + sy * cam.R[5] + sz * cam.R[8];
    };
    const float z_unfilled = zc(0.f, 0.f, 0.f);
    float cm;
    if (lo == 0) {
        const float4 p = smp_loc[off];
        cm = zc(p.x, p.y, p.z);
    } else {
        cm = ray_cm[r];
    }
    float T = ray_T[r];
    const int end = min(hi, cnt);
    for (int i = lo; i < end; ++i) {
        float delta;
        if (i == opts.SR - 1) {
            delta = vs;
        } else {
            float z_next = z_unfilled;
            if (i + 1 < cnt) {
                const float4 p = smp_loc[off + i + 1];
                z_next = zc(p.x, p.y, p.z);
            }
            const float cm_next = fmaxf(cm, z_next);
            delta = cm_next - cm;
            cm = cm_next;
            if (delta < 1e-8f || delta > two_vs) delta = vs;
        }
        const float opacity = 1.0f - expf(-smp_sig_s[off + i] * delta);
        T = T * (1.0f - opacity + 1e-10f);
    }
    ray_T[r] = T;
    ray_cm[r] = cm;
    ray_alive[r] = (T > opts.early_stop_eps && hi < cnt) ? 1 : 0;
}

__global__ void k_publish_shaded(const int *__restrict__ n_sel, int idx, int64_t *__restrict__ counters)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) counters[PNR_CNT_SAMPLES_SHADED] = n_sel[idx];
}

int launch_shade(const pnr_scene *scene, const pnr_weights *w, const CamRef &cr, const float *d_dirs,
                 const pnr_render_opts_t &opts, int64_t R, RenderWs &ws, int64_t cap, int64_t *d_counters,
                 hipStream_t stream, hipEvent_t ev_points, hipEvent_t ev_between)
{
    const int K = opts.K, precision = opts.precision;
    const bool early = opts.early_stop_eps > 0.f;
    ShadeParams P{};
    P.point_rows = reinterpret_cast<const float4 *>(scene->point_rows);
    P.wbuf = w->buf;
    P.wbytes = w->bytes;
    for (int i = 0; i < 9; ++i) {
        P.w_off[i] = w->w_off[i];
        P.w16_off[i] = w->w16_off[i];
        P.b_off[i] = w->b_off[i];
        P.Rw2c[i] = w->Rw2c[i];
    }
    P.cr = cr;
    P.dirs = d_dirs;
    P.smp_loc = ws.smp_loc;
    P.smp_ray = ws.smp_ray;
    P.smp_pidx = ws.smp_pidx;
    P.vs_list = early ? ws.vs_all : ws.vs_list;
    P.n_sel = ws.n_sel;
    P.i_v0 = early ? 4 : 6;  // n_sel[6] == 0
    P.i_v1 = early ? 5 : 1;
    P.smp_sig_s = early ? ws.smp_sig_s : nullptr;
    P.smp_sigma = ws.smp_sigma;
    P.agg = ws.agg;
    P.smp_out = ws.smp_out;
    P.K = K;
    P.dbg_off = cap - 8192;
    P.w16a_off = w->w16a_off;
    P.w16b_off = w->w16b_off;
    P.w4acc_off = w->w4acc_off;
    P.w8acc_off = w->w8acc_off;
    P.w32a_off = w->w32a_off;
    P.w32b_off = w->w32b_off;
    P.pt_rank = ws.pt_rank;
    P.pt_list = ws.pt_list;
    P.pt_table = reinterpret_cast<float4 *>(ws.pt_table);
    P.u_cap = (int)ws.u_cap;
    int dev = 0, cus = 256;
    PNR_HIP_CHECK(hipGetDevice(&dev));
    PNR_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    // decoded features of samples without neighbours (or not shaded) are zero (studio_model.py:361-362)
    PNR_HIP_CHECK(hipMemsetAsync(ws.smp_out, 0, (size_t)cap * sizeof(float4), stream));
    const int seg = K <= 8 ? 8 : (K <= 16 ? 16 : 0);   // lanes per sample segment (0: exactly K)
    const int spt = (32 / (seg ? seg : K)) * WAVES;
    const int64_t max_tiles = (cap + spt - 1) / spt;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cus, max_tiles));
    const bool bf = precision == PNR_PRECISION_BF16X3;
    {
        if (!ws.pt_table) {
            set_error("launch_shade: the workspace lacks the point-part buffers (pnr_render_workspace_bytes_for)");
            return PNR_ERR_WORKSPACE;
        }
        const int64_t ptiles = (ws.u_cap + 32 * WAVES - 1) / (32 * WAVES);
        const dim3 pgrid((unsigned)std::max<int64_t>(1, std::min<int64_t>(cus, ptiles)));
        if (bf)
            hipLaunchKernelGGL(k_point_part, pgrid, dim3(TPB), 0, stream, P);
        else
            hipLaunchKernelGGL(k_point_part_f32, pgrid, dim3(TPB), 0, stream, P);
    }
    if (ev_points) PNR_HIP_CHECK(hipEventRecord(ev_points, stream));
    auto launch_pairs = [&]() {
        if (seg == 8) {
            if (bf)
                hipLaunchKernelGGL(k_shade_pairs_bf16<8>, dim3(grid), dim3(TPB), 0, stream, P);
            else
                hipLaunchKernelGGL(k_shade_pairs<8>, dim3(grid), dim3(TPB), 0, stream, P);
        } else if (seg == 16) {
            if (bf)
                hipLaunchKernelGGL(k_shade_pairs_bf16<16>, dim3(grid), dim3(TPB), 0, stream, P);
            else
                hipLaunchKernelGGL(k_shade_pairs<16>, dim3(grid), dim3(TPB), 0, stream, P);
        } else {
            if (bf)
                hipLaunchKernelGGL(k_shade_pairs_bf16<0>, dim3(grid), dim3(TPB), 0, stream, P);
            else
                hipLaunchKernelGGL(k_shade_pairs<0>, dim3(grid), dim3(TPB), 0, stream, P);
        }
    };
    if (!early) {
        launch_pairs();
    } else {
        const unsigned rgrid = (unsigned)((R + TPB - 1) / TPB);
        const unsigned sgrid = (unsigned)std::min<int64_t>((cap + TPB - 1) / TPB, 256 * 32);
        PNR_HIP_CHECK(hipMemsetAsync(ws.smp_sig_s, 0, (size_t)cap * sizeof(float), stream));
        hipLaunchKernelGGL(k_pass_init, dim3(rgrid), dim3(TPB), 0, stream, R, ws.ray_T, ws.ray_alive, ws.n_sel);
        int *flag = ws.smp_valid, *pos = ws.smp_voff;  // free again once the valid samples are listed
#ifndef PNR_ES_BOUNDS
#define PNR_ES_BOUNDS 0, 3, 6, 12, 24
#endif
        static const int bounds[] = {PNR_ES_BOUNDS};
        constexpr int NP = (int)(sizeof(bounds) / sizeof(bounds[0]));
        for (int p = 0; p < NP; ++p) {
            const int lo = bounds[p], hi = (p + 1 < NP) ? std::min(bounds[p + 1], opts.SR) : opts.SR;
            if (lo >= opts.SR) break;
            hipLaunchKernelGGL(k_pass_flag, dim3(sgrid), dim3(TPB), 0, stream, ws.n_sel, ws.vs_list, ws.smp_ray,
                               ws.ray_off, ws.ray_alive, lo, hi, flag);
            int rc = scan_exclusive_i32(flag, pos, cap, ws.n_sel + 1, nullptr, ws.scan_temp, stream);
            if (rc != PNR_OK) return rc;
            hipLaunchKernelGGL(k_pass_scatter, dim3(sgrid), dim3(TPB), 0, stream, ws.n_sel, ws.vs_list, flag, pos,
                               ws.vs_all);
            hipLaunchKernelGGL(k_pass_advance, dim3(1), dim3(64), 0, stream, ws.n_sel, pos);
            launch_pairs();
            hipLaunchKernelGGL(k_pass_update, dim3(rgrid), dim3(TPB), 0, stream, cr, opts, R, ws.ray_cnt, ws.ray_off,
                               ws.smp_loc, ws.smp_sig_s, ws.n_sel, lo, hi, ws.ray_T, ws.ray_cm, ws.ray_alive);
        }
        P.i_v0 = 6;  // the colour MLP runs once over everything that was shaded
    }
    hipLaunchKernelGGL(k_publish_shaded, dim3(1), dim3(64), 0, stream, ws.n_sel, P.i_v1, d_counters);
    if (ev_between) PNR_HIP_CHECK(hipEventRecord(ev_between, stream));
    const int64_t ctiles = (cap + 32 * WAVES - 1) / (32 * WAVES);
    const unsigned cgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cus, ctiles));
    if (bf)
        hipLaunchKernelGGL(k_shade_color_bf16, dim3(cgrid), dim3(TPB), 0, stream, P);
    else
        hipLaunchKernelGGL(k_shade_color, dim3(cgrid), dim3(TPB), 0, stream, P);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

}  // namespace pnr

using namespace pnr;

extern "C" int pnr_weights_create(pnr_weights_t **out)
{
    PNR_REQUIRE(out != nullptr, "pnr_weights_create: out is null");
    *out = new pnr_weights();
    return PNR_OK;
}

extern "C" int pnr_weights_destroy(pnr_weights_t *w)
{
    if (!w) return PNR_OK;
    if (w->buf) (void)hipFree(w->buf);
    delete w;
    return PNR_OK;
}

extern "C" int pnr_weights_pack(pnr_weights_t *w, const float *const d_w[9], const float *const d_b[9],
                                const float *d_Rw2c, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    PNR_REQUIRE(w && d_w && d_b && d_Rw2c, "pnr_weights_pack: null argument");
    for (int i = 0; i < 9; ++i) PNR_REQUIRE(d_w[i] && d_b[i], "pnr_weights_pack: tensor %d is null", i);
    static const int n_out[9] = {256, 256, 256, 256, 1, 128, 128, 128, 3};
    static const int n_in[9] = {284, 256, 263, 256, 256, 280, 128, 128, 128};
    static const int ksp[9] = {144, 128, 132, 128, 0, 140, 64, 64, 0};    // fp32 k-steps (2 features each)
    static const int ks16[9] = {18, 16, 17, 16, 0, 18, 8, 8, 0};          // bf16 k-steps (16 features each)
    static const int kind[9] = {L_BASE0, L_HIDDEN, L_HEAD0, L_HIDDEN, -1, L_COLOR0, L_HIDDEN, L_HIDDEN, -1};
    size_t off = 0;
    auto pad = [](size_t n) { return (n + 1023) / 1024 * 1024; };  // 4 KiB granules (LDS staging rounds)
    for (int i = 0; i < 9; ++i) {
        w->w_off[i] = off;
        off += pad(ksp[i] ? (size_t)(n_out[i] / 32) * (ksp[i] / 4) * 256 : (size_t)n_out[i] * n_in[i]);
    }
    for (int i = 0; i < 9; ++i) {
        w->w16_off[i] = off;
        // [mt][ks][2 planes][64 lanes][8 bf16] = mt * ks * 2048 bytes = mt * ks * 512 floats
        off += ks16[i] ? pad((size_t)(n_out[i] / 32) * ks16[i] * 512) : 0;
    }
    // mlp_base layer 0 split for the bf16x3 mode: point-only k-steps 0..13 and pair k-steps 14..17
    w->w16a_off = off;
    off += pad((size_t)8 * 14 * 512);
    w->w16b_off = off;
    off += pad((size_t)4 * 8 * 512);
    w->w32a_off = off;
    off += pad((size_t)8 * (112 / 4) * 256);
    w->w32b_off = off;
    off += pad((size_t)8 * (32 / 4) * 256);
    w->w4acc_off = off;
    off += pad(256);
    w->w8acc_off = off;
    off += pad(512);
    for (int i = 0; i < 9; ++i) {
        w->b_off[i] = off;
        off += pad((size_t)n_out[i]);
    }
    off += 2 * 1024 * 9;  // tail pad: a staged tile may read up to 36 KiB from its start
    if (!w->buf || w->bytes != off * sizeof(float)) {
        if (w->buf) (void)hipFree(w->buf);
        w->buf = nullptr;
        PNR_HIP_CHECK(hipMalloc((void **)&w->buf, off * sizeof(float)));
        w->bytes = off * sizeof(float);
    }
    PNR_HIP_CHECK(hipMemsetAsync(w->buf, 0, w->bytes, stream));
    for (int i = 0; i < 9; ++i) {
        if (ksp[i]) {
            hipLaunchKernelGGL(k_pack_layer, dim3(256), dim3(256), 0, stream, d_w[i], n_out[i], n_in[i], kind[i],
                               ksp[i], 0, w->buf + w->w_off[i]);
            hipLaunchKernelGGL(k_pack_layer_bf16, dim3(256), dim3(256), 0, stream, d_w[i], n_out[i], n_in[i], kind[i],
                               ks16[i], ks16[i], 0, reinterpret_cast<unsigned short *>(w->buf + w->w16_off[i]));
        } else {
            int n = n_out[i] * n_in[i];
            hipLaunchKernelGGL(k_copy, dim3((n + 255) / 256), dim3(256), 0, stream, d_w[i], n, w->buf + w->w_off[i]);
        }
        hipLaunchKernelGGL(k_copy, dim3((n_out[i] + 255) / 256), dim3(256), 0, stream, d_b[i], n_out[i],
                           w->buf + w->b_off[i]);
    }
    hipLaunchKernelGGL(k_pack_layer_bf16, dim3(256), dim3(256), 0, stream, d_w[0], 256, 284, (int)L_BASE0, 14, 14, 0,
                       reinterpret_cast<unsigned short *>(w->buf + w->w16a_off));
    hipLaunchKernelGGL(k_pack_layer, dim3(256), dim3(256), 0, stream, d_w[0], 256, 284, (int)L_BASE0, 112, 0,
                       w->buf + w->w32a_off);
    hipLaunchKernelGGL(k_pack_layer, dim3(256), dim3(256), 0, stream, d_w[0], 256, 284, (int)L_BASE0, 32, 112,
                       w->buf + w->w32b_off);
    hipLaunchKernelGGL(k_pack_head_acc, dim3(1), dim3(256), 0, stream, d_w[4], w->buf + w->w4acc_off);
    hipLaunchKernelGGL(k_pack_color_head_acc, dim3(2), dim3(256), 0, stream, d_w[8], w->buf + w->w8acc_off);
    hipLaunchKernelGGL(k_pack_layer_bf16, dim3(256), dim3(256), 0, stream, d_w[0], 256, 284, (int)L_BASE0, 8, 4, 14,
                       reinterpret_cast<unsigned short *>(w->buf + w->w16b_off));
    PNR_HIP_CHECK(hipGetLastError());
    PNR_HIP_CHECK(hipMemcpyAsync(w->Rw2c, d_Rw2c, 9 * sizeof(float), hipMemcpyDeviceToHost, stream));
    PNR_HIP_CHECK(hipStreamSynchronize(stream));
    w->packed = true;
    return PNR_OK;
}
