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
//
// Files: pnr_shade_common.h (this: parameters, per-row inputs, segment sums, shared epilogues), pnr_shade_fp32.hip
// (exact mode kernels), pnr_shade_bf16.hip (bf16x3 machinery and kernels), pnr_shade.hip (weight packing, early
// termination passes, launch_shade).
#ifndef PNR_SHADE_COMMON_H_
#define PNR_SHADE_COMMON_H_
#include <algorithm>

#include "pnr_internal.h"

namespace pnr {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

// Packed fp32 forms in the fp32 MLP kernels (default on; -DPNR_NO_PK_LEAKY builds the scalar forms for A/B runs): the
// LeakyReLU multiply of two values as ONE v_pk_mul_f32 (forced through inline asm: left alone hipcc turns the vector
// multiply back into two v_mul_f32 -- 20 v_pk_mul_f32 against 864 v_mul_f32 in the pair kernel's ISA) and, in the pair
// kernel's layer-4 sink, the density product as v_pk_fma_f32 and the neighbour weight as v_pk_mul_f32 on pairs of
// values: 487 fewer VALU instructions per 32-pair tile (2 270 -> 1 780), 33.10 -> 33.00 ms on cfg 1 (A/B on one device,
// three rounds; the kernel is NOT limited by its VALU count -- see DESIGN.md section 4.1).
#ifndef PNR_NO_PK_LEAKY
#define PNR_PK_LEAKY 1
#endif
// two LeakyReLUs with one packed multiply (v_pk_mul_f32: the same IEEE product as two v_mul_f32)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void leaky2(float a, float b, float &ra, float &rb)
{
    const f32x2 x = {a, b};
#ifdef PNR_PK_LEAKY
    // forced: hipcc turns the vector multiply below back into two v_mul_f32 (20 v_pk_mul_f32 against 864 v_mul_f32 in the
    // fp32 pair kernel's ISA)
    f32x2 y;
    const f32x2 k = {0.1f, 0.1f};
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y) : "v"(x), "v"(k));
#else
    const f32x2 y = x * 0.1f;
#endif
    asm("v_max_f32 %0, %1, %2" : "=v"(ra) : "v"(a), "v"(y.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(rb) : "v"(b), "v"(y.y));
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
    float *smp_wgt;    // [S_sel, K] normalised inverse-distance weights (dense units only)
    const int *n_sel;  // [1] = S_valid
    float *smp_sigma;  // [S_valid]
    float *agg;        // [S_valid, 256]
    float4 *smp_out;   // [S_sel]
    int K;
    // bf16x3 mode: factorised first layer
    int i_v0, i_v1;        // the kernel works on positions [n_sel[i_v0], n_sel[i_v1]) of vs_list
    float *smp_sig_s;      // [S_sel] density by sample index (early ray termination), may be null
    size_t w16a_off, w16b_off, w4acc_off, w8acc_off;
    size_t w32a_off, w32b_off;   // fp32-packed halves of mlp_base layer 0 (point-only k-steps 0..111, pair 112..143)
    const int *pt_rank;     // [N+1] point index -> row of pt_table
    const int *pt_list;     // [U] rows -> point index
    float4 *pt_table;       // [u_cap, 8 row blocks, 2 lane halves, 4] float4
    int u_cap;
    // training renders (k_shade_pairs<SEG, true>): the backward's activation tapes H1 [rows,256], H2 [rows,264],
    // G1, G2 [rows,256] and their sizes in bytes
    float *tape[4];
    size_t tape_bytes[4];
    // ... and the LeakyReLU masks of the same four layers as BITS, [layer][row][lane half][4 words]: word w of a row's
    // half holds [activation > 0] of the features of output tiles 2 w, 2 w + 1 in accumulator order, the first at bit 31
    // (what k_train_pairs_bwd shifts out as it walks the layers backwards); tape_bits_rows = rows per layer
    unsigned *tape_bits;
    size_t tape_bits_rows;
    float *tape_rowz;   // [rows] the density head's pre-activation of every row (the backward's [z > 0])
    // ... and of the colour MLP (k_shade_color<true>): its three hidden activations C1, C2, C3 [S, 128] and the sigmoid
    // outputs of the colour head [S] (null: not taped)
    float *ctape[3];
    float4 *tape_sg;
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

// The same without any fallback (no branch, nothing if-converted): three-constant Cody-Waite reduction, exact for
// |k| < 2^15 (|x| < ~5e4); beyond that the result degrades gracefully (it stays in [-1, 1]) where fp32 arguments no
// longer resolve the period anyway.  Used by the fp32 pair kernel, whose arguments are voxel-sized distances times
// 2^0..2^4.
__device__ __forceinline__ void fast_sincos_nb(float x, float &sn, float &cs)
{
    const float k = rintf(x * 0.636619772367581343f);           // x * 2/pi
    float r = fmaf(-k, 1.5703125f, x);                          // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188216e-8
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188216e-8f, r);
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

// v of the lane DPP control CTRL selects (every lane has a source for the quad_perm controls)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Lane segment of one sample inside each 32-lane half.  SEG = 8 (K <= 8) / 16 (K <= 16): the sample's K rows sit in
// the first K lanes of an 8- / 16-lane segment aligned to the DPP rows (the other lanes of the segment idle: pidx -1,
// weight 0), so that sums over a sample are DPP steps.  SEG = 0 (K > 16): segments of exactly K lanes, summed with
// K cross-lane reads.
// Tile order of the persistent pair kernels.  Workgroup b runs on XCD b % 8 (round-robin dispatch), one workgroup per
// CU.  The tiles are cut into strips of G/8 consecutive tiles (one per CU of an XCD) and an XCD takes XCD_CHUNK
// consecutive strips before it jumps over the strips of the other seven: rays that are neighbours in the image share
// their neural points, and the pt_table rows they gather stay in the L2 of the XCD that fetched them.  Measured on
// cfg 1 (rocprofv3 FETCH_SIZE of the fp32 pair kernel, per launch): chunk 1 4.75 GB, 4 3.65, 8 3.43, 16 3.37, 32 3.41
// -- for 1.9 GB of distinct rows; kernel time unchanged (+-0.2 %) once the last, partial super-round is walked strip
// by strip (without that the XCDs finish up to a chunk apart: chunk 64 ran 10 % longer).
#ifndef PNR_XCD_CHUNK
#define PNR_XCD_CHUNK 16
#endif
struct TileWalk {
    int x, j, g8, plain, G, n_full, t_full;
    __device__ __forceinline__ TileWalk(int block, int grid, int ntiles)
        : x(block % 8), j(block / 8), g8(grid / 8), plain(grid % 8 != 0 ? block : -1), G(grid)
    {
        // whole super-rounds (8 XCDs x XCD_CHUNK strips) are walked chunk by chunk, the rest strip by strip, so that
        // the XCDs finish within one strip of each other
        const int per_super = PNR_XCD_CHUNK * grid;
        const int full = per_super > 0 ? ntiles / per_super : 0;
        n_full = full * PNR_XCD_CHUNK;
        t_full = full * per_super;
    }
    // the n-th tile of this workgroup; increasing in n
    __device__ __forceinline__ int at(int n) const
    {
        if (plain >= 0) return plain + n * G;
        constexpr int C = PNR_XCD_CHUNK;
        if (n >= n_full) return t_full + (n - n_full) * G + x * g8 + j;
        return ((n / C) * 8 + x) * (C * g8) + (n % C) * g8 + j;
    }
};

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
    float wgt;   // dense units: the row's normalised weight (smp_wgt)
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
template <bool FAST_PE, bool DOUBLE_ANGLE = true>
__device__ __forceinline__ void point_inputs(const float (&e)[16], float *x0)
{
#pragma unroll
    for (int d = 0; d < 16; ++d) x0[d] = e[d];
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        float sn = 0.f, cs = 1.f;
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            if (FAST_PE && DOUBLE_ANGLE && f > 0) {
                // double angle from the previous octave: sin 2a = 2 sin a cos a, cos 2a = (cos a - sin a)(cos a + sin a)
                const float s2 = 2.0f * sn * cs, c2 = (cs - sn) * (cs + sn);
                sn = s2;
                cs = c2;
            } else if (FAST_PE && !DOUBLE_ANGLE) {
                fast_sincos_nb(e[d] * (float)(1 << f), sn, cs);   // fp32 mode: every octave from its own argument
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

// the lane's pair inputs: weight, encoded distances (xq[0:32] = x0[112:144]) and the extra head inputs.
// FAST_PE: the branch-free Cody-Waite sincos (1e-7 absolute) instead of libm's; with DOUBLE_ANGLE the octaves above the
// first follow by the double-angle identities (the bf16x3 mode: their error stays below that mode's 2^-16 products),
// without it every octave is evaluated from its own argument x * 2^f (exact scaling).
template <int SEG, bool FAST_PE, bool DOUBLE_ANGLE = true>
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
    if constexpr (SEG < 0) {
        ctx.wgt = valid ? f.wgt : 0.f;   // dense units (k_shade_pairs_dense): normalised by k_pair_weights
    } else {
        const float wsum = seg_sum<SEG>(wgt, K, lane);
        ctx.wgt = wgt / fmaxf(wsum, 1e-8f);
    }

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
            if (FAST_PE && DOUBLE_ANGLE && f > 0) {
                const float s2 = 2.0f * sn * cs, c2 = (cs - sn) * (cs + sn);
                sn = s2;
                cs = c2;
            } else if (FAST_PE && !DOUBLE_ANGLE) {
                fast_sincos_nb(dd[d] * (float)(1 << f), sn, cs);
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

// kernels of the two arithmetic modes (pnr_shade_fp32.hip / pnr_shade_bf16.hip), launched by launch_shade;
// seg = lanes per sample segment (8, 16, or 0 for exactly K)
void launch_point_part_fp32(dim3 grid, hipStream_t stream, const ShadeParams &P);
void launch_pairs_fp32(int seg, dim3 grid, hipStream_t stream, const ShadeParams &P);
bool launch_pairs_fp32_tape(int seg, dim3 grid, hipStream_t stream, const ShadeParams &P);   // false: no such kernel
// dense units (any K): su consecutive samples fill tu 32-row tiles of a wave, see k_shade_pairs_dense
int launch_pairs_fp32_dense(int su, int tu, dim3 grid, hipStream_t stream, const ShadeParams &P);   // PNR_OK or a status
void launch_color_fp32(dim3 grid, hipStream_t stream, const ShadeParams &P);
void launch_point_part_bf16(dim3 grid, hipStream_t stream, const ShadeParams &P);
void launch_pairs_bf16(int seg, dim3 grid, hipStream_t stream, const ShadeParams &P);
void launch_color_bf16(dim3 grid, hipStream_t stream, const ShadeParams &P);

}  // namespace pnr
#endif  // PNR_SHADE_COMMON_H_
