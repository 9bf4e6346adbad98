// Shade stage, exact mode (PNR_PRECISION_FP32): v_mfma_f32_32x32x2_f32, weights streamed L2 -> VGPR.  See
// pnr_shade_common.h for the design overview.
#include "pnr_shade_common.h"

namespace pnr {

// ================================================================================================
// fp32 mode
// ================================================================================================
// ================================================================================================
// The pair kernel as ONE stream of weight groups.
//
// The first version of this kernel (round 1: four separate layer loops)
// ran its layers as four separate loops: every layer start refilled the weight pipeline (first load -> first MFMA: an
// L2 round trip with the matrix pipe idle), loaded its 32 bias float4 and waited for them, and every layer end read
// 128 accumulators back, applied LeakyReLU and only then started the next layer -- 430 instructions with no MFMA in
// flight, three times per tile, plus 1800 instructions of epilogue (density head, K-aggregation, stores) and a
// prologue whose fifteen libm sincosf calls cost 2200 instructions (their Payne-Hanek argument reduction is computed
// unconditionally).  35.1 ms per 800x800 frame, 0.81 of the fp32 MFMA peak.
//
// Here the 840 weight groups of a tile (64 + 256 + 264 + 256 groups of 4 k-steps) form one sequence with a rolling
// window of PFS loads in flight that never drains -- not at a layer boundary and not at a tile boundary (the window of
// the next tile's first groups fills during this tile's last groups).
//
// What shapes everything else (tools/ub_mfma_dep.hip): a SIMD does NOT overlap vector-ALU work
// with MFMA work -- not within a wave, not across two waves.  A dependent chain of v_mfma_f32_32x32x2_f32 with a
// 1-KiB weight load per 4 MFMAs and its B operands in AGPRs runs at 65.3 cycles per MFMA (0.98 of the pipe), scalar
// and memory instructions between the MFMAs are free, but every VALU instruction costs its own issue time, and every
// MFMA-to-MFMA gap that holds VALU work costs ~18 cycles on top.  So the kernel minimises the NUMBER of VALU
// instructions and the number of places they sit in:
//   * MFMA results live in VGPRs (this file is built with -amdgpu-mfma-vgpr-form): LeakyReLU reads them directly,
//     no v_accvgpr_read; the activations of every other layer are pinned in AGPRs (to_a), where the MFMA reads its B
//     operand directly -- the two 128-value activation sets never compete for the 256 architectural VGPRs;
//   * the activation of output tile m (16 values: packed multiply, max, AGPR write) is ONE block behind the first MFMA
//     of tile m + 1 -- of the NEXT layer's tile 0 for the last tile (its values are that layer's k-steps 112..127);
//   * the accumulator initial values (bias / pt_table rows, 16 per lane and output tile) are loaded one output tile
//     ahead, straight into the registers the MFMA accumulates in;
//   * the last layer's outputs never form an array: the 16 finished values of an output tile go through LeakyReLU, the
//     density-head product, the neighbour weight and the K-sum in one block, 7 instructions per value: the three DPP
//     adds are fused (v_add_f32_dpp) and run as a pipeline over consecutive values, so none reads a register written
//     by its predecessor; every fourth value stores a float4 of the aggregated feature through a buffer descriptor
//     (non-writing lanes carry an out-of-range offset: no branch);
//   * the gather chain of the NEXT tile is issued one dependent level per layer boundary;
//   * positional encodings use the branch-free Cody-Waite sincos (1e-7 absolute), each octave from its own argument.
// 9 700 instructions per tile for 3 360 MFMAs, ~2 600 of them VALU.  33.1 ms per 800x800 frame (round 1: 35.3).
// Sharing the weights of the four waves through an LDS ring (LDS-DMA, one barrier per 32 groups) was built and
// measured 6 % SLOWER (the DMA pieces cost the issuing wave more than the loads they replace).
#ifndef PNR_PFS
#define PNR_PFS 6   // weight groups (1 KiB per wave each) in flight; divides NG_TILE (5 / 6 / 8 / 12 measured equal)
#endif
constexpr int PFS = PNR_PFS;
#ifndef PNR_SINK_PER
#define PNR_SINK_PER 16   // layer-4 output values sunk in one place (1 / 2 / 4 / 16: 33.2 / 33.1 / 33.1 / 33.06 ms)
#endif
#ifndef PNR_FPM
#define PNR_FPM 16        // activation values of layers 1-3 behind one MFMA (even: LeakyReLU runs on pairs)
#endif
constexpr int FPM = PNR_FPM;
static_assert(FPM % 2 == 0 && 16 % FPM == 0, "");
static_assert(FPM == 16 && PNR_SINK_PER == 16, "the TAPE mode of k_shade_pairs collects an output tile's 16 values in one block");
constexpr int NG_L1 = 8 * (32 / 4), NG_L2 = 8 * (128 / 4), NG_L3 = 8 * (132 / 4), NG_L4 = 8 * (128 / 4);
constexpr int NG_TILE = NG_L1 + NG_L2 + NG_L3 + NG_L4;   // 840
static_assert(NG_TILE % PFS == 0, "the window slot of a group must not depend on the tile");

struct WBase {
    int l1, l2, l3, l4;   // byte offsets of the four layers' packed weights
};

// byte offset of weight group G of the tile stream (G >= NG_TILE: the next tile, same weights)
__device__ __forceinline__ int group_off(const WBase &wb, int G)
{
    G = G >= NG_TILE ? G - NG_TILE : G;
    if (G < NG_L1) return wb.l1 + G * 1024;
    if (G < NG_L1 + NG_L2) return wb.l2 + (G - NG_L1) * 1024;
    if (G < NG_L1 + NG_L2 + NG_L3) return wb.l3 + (G - NG_L1 - NG_L2) * 1024;
    return wb.l4 + (G - NG_L1 - NG_L2 - NG_L3) * 1024;
}

struct Ini {
    float4 v[4];   // accumulator initial values 4q .. 4q+3 (q = 0..3) of one output tile
};

__device__ __forceinline__ f32x16 acc_from(const Ini &b)
{
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a[4 * q + 0] = b.v[q].x;
        a[4 * q + 1] = b.v[q].y;
        a[4 * q + 2] = b.v[q].z;
        a[4 * q + 3] = b.v[q].w;
    }
    return a;
}

// bias of output tile m for this lane half: values 4q..4q+3 = bias[32m + 8q + 4h + 0..3]
__device__ __forceinline__ Ini bias_ini(const float *__restrict__ bias, int m, int h)
{
    Ini b;
#pragma unroll
    for (int q = 0; q < 4; ++q) b.v[q] = *reinterpret_cast<const float4 *>(bias + 32 * m + 8 * q + 4 * h);
    return b;
}

// One layer of a stream.  G0: index of its first weight group in the tile stream; `off(G)`: byte offset of group G
// (G up to one window past the tile's last group: the next tile, same weights); PF: loads in flight.  `in`: the lane's
// B operands.  acc[0] must hold the initial values of output tile 0 on entry -- tile m + 1's are fetched by
// `next_ini(m + 1)` during tile m; `fill(m, i)` runs behind MFMA i (0 .. 4 KG - 1) of output tile m.
template <int KSP, int MT, int G0, int PF, typename Off, typename NextIni, typename Fill>
__device__ __forceinline__ void layer_stream(__amdgpu_buffer_rsrc_t rsrc, int voff, Off off, float4 (&wq)[PF],
                                             const float (&in)[KSP], f32x16 (&acc)[MT], NextIni next_ini, Fill fill)
{
    static_assert(KSP % 4 == 0, "k-steps are packed in groups of 4");
    constexpr int KG = KSP / 4;
    Ini nx;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m > 0) acc[m] = acc_from(nx);
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int G = G0 + m * KG + kg;
            const float4 w = wq[G % PF];
            wq[G % PF] = load_w(rsrc, voff, off(G + PF));
            if (kg == 0 && m + 1 < MT) nx = next_ini(m + 1);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, in[4 * kg + 0], acc[m], 0, 0, 0);
            fill(m, 4 * kg + 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, in[4 * kg + 1], acc[m], 0, 0, 0);
            fill(m, 4 * kg + 1);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, in[4 * kg + 2], acc[m], 0, 0, 0);
            fill(m, 4 * kg + 2);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, in[4 * kg + 3], acc[m], 0, 0, 0);
            fill(m, 4 * kg + 3);
            // one scheduling region per group: the loads of a whole layer must not be hoisted to its top
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// pins a value in the accumulator half of the register file.  MFMA B operands may come from there directly, so the
// activations of every other layer live in AGPRs for their whole life and the two 128-value activation sets never
// compete for the 256 architectural VGPRs.
__device__ __forceinline__ float to_a(float v)
{
    asm("" : "+a"(v));
    return v;
}

// off: a constant once the caller is unrolled and inlined (an immediate of the instruction)
__device__ __forceinline__ void lds_add(unsigned addr, float v, int off)
{
    asm volatile("ds_add_f32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(off));
}
__device__ __forceinline__ void lds_write4(unsigned addr, f32x4 v, int off)
{
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(off));
}
__device__ __forceinline__ f32x4 lds_read4(unsigned addr, int off)
{
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
    return v;
}
__device__ __forceinline__ void lds_wait()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// TAPE (training renders, ShadeParams.tape_*): the post-activation outputs of the four layers -- H1, H2, G1, G2 of
// pnr_render_backward's tape, row (valid sample v) * K + slot, row-major -- leave the kernel as it computes them, so
// that the backward does not recompute the MLP chain with four row GEMMs (6 of its 21 ms at 65 536 rays).  The 16
// values a lane holds of an output tile are features 32 m + 8 q + 4 h + i of ITS row: written as they are they would be
// 16-byte pieces 1 KiB apart.  Each output tile therefore passes through a 4.5-KiB wave-private LDS block (32 rows x
// 32 features, rows 144 B apart: conflict-free), written in the activation gap of the tile and read back transposed in
// the next gap -- 8 lanes per row, 128 contiguous bytes -- so every store instruction writes whole 128-byte lines of
// 8 rows.  Two blocks alternate.  The LDS latency of the read-back is exposed once per gap (~2 % of a tile).
constexpr int TAPE_ROW_B = 144, TAPE_BLK_B = 32 * TAPE_ROW_B, TAPE_WAVE_B = 2 * TAPE_BLK_B;

template <int SEG, bool TAPE = false>
__global__ void __launch_bounds__(TPB, 1) k_shade_pairs(ShadeParams P)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int SPT = (32 / seg_len<SEG>(P.K)) * WAVES;  // samples per workgroup tile
    const int V0 = P.n_sel[P.i_v0], S_valid = P.n_sel[P.i_v1];
    const int ntiles = (S_valid - V0 + SPT - 1) / SPT;
    const TileWalk walk((int)blockIdx.x, (int)gridDim.x, ntiles);   // XCD-aware tile order
    const int t_begin = walk.at(0);

    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wbuf), 0, (int)P.wbytes, 0x00020000);
    const int voff = lane * 16;
    const WBase wb_ = {(int)(P.w32b_off * 4), (int)(P.w_off[1] * 4), (int)(P.w_off[2] * 4), (int)(P.w_off[3] * 4)};
    const float *b1 = P.wbuf + P.b_off[1], *b2 = P.wbuf + P.b_off[2], *b3 = P.wbuf + P.b_off[3];
    const float *w4t = P.wbuf + P.w4acc_off;   // density head in accumulator order: [(tile * 2 + h) * 16 + r]
    const float b4 = P.wbuf[P.b_off[4]];
    const int K = P.K;

    // ---- TAPE: staging blocks, the four tapes, the rows this lane stores ------------------------------------------
    extern __shared__ float tape_lds[];
    const unsigned tblk = TAPE ? (unsigned)(uintptr_t)tape_lds + (unsigned)(wave * TAPE_WAVE_B) : 0u;
    const unsigned t_wr = tblk + (unsigned)((lane & 31) * TAPE_ROW_B + 16 * h);            // + block + 32 q
    const unsigned t_rd = tblk + (unsigned)((lane >> 3) * TAPE_ROW_B + (lane & 7) * 16);   // + block + 8 i rows
    // descriptors of the tile's 32 tape rows (set per tile: offsets inside them stay small whatever the tape's size)
    __amdgpu_buffer_rsrc_t trsrc[4], brsrc[4], zrsrc;
    unsigned tg[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // rows 8 i + l / 8 of the tile: tape row - the tile's first
    // the mask bits of the lane's own row (ShadeParams.tape_bits): one running word, four finished words per layer
    unsigned sbw = 0u, sbq[4] = {0u, 0u, 0u, 0u};
    unsigned brow = 0xFFFFFFFFu;   // the lane's tape row - the tile's first, 0xFFFFFFFF = none
    // block (layer pl, output tile pt), written one gap earlier, goes out: 8 lanes per row, whole 128-byte lines
    auto tape_flush = [&](int pl, int pt) {
        const int ld = pl == 1 ? 264 : 256;   // H2 carries the seven extra head inputs behind its 256 columns
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = lds_read4(t_rd, (pt & 1) * TAPE_BLK_B + i * 8 * TAPE_ROW_B);
        lds_wait();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 o;
            o.x = __float_as_uint(v[i].x);
            o.y = __float_as_uint(v[i].y);
            o.z = __float_as_uint(v[i].z);
            o.w = __float_as_uint(v[i].w);
            // row tg[i] (0xFFFFFFFF: no such row -> beyond the descriptor's range, dropped), features 32 pt + 4 (l & 7)
            const unsigned off =
                tg[i] == 0xFFFFFFFFu ? 0xFFFFFFF0u : tg[i] * (unsigned)(ld * 4) + 16u * (lane & 7) + 128u * (unsigned)pt;
            __builtin_amdgcn_raw_buffer_store_b128(o, trsrc[pl], (int)off, 0, 0);
        }
    };
    // the 16 activations of output tile `tile` of layer `layer` (0..3) into block tile & 1; the block before it in the
    // sequence (layer 0 tile 0, ..., layer 3 tile 7) goes out first
    auto tape_tile = [&](int layer, int tile, const float (&t)[16]) {
        if (tile > 0)
            tape_flush(layer, tile - 1);
        else if (layer > 0)
            tape_flush(layer - 1, 7);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
            lds_write4(t_wr, v, (tile & 1) * TAPE_BLK_B + 32 * q);
        }
        // [value > 0] shifted into the running word (compare into vcc, add-with-carry: word = 2 word + bit)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(sbw) : "v"(t[r]) : "vcc");
        if (tile & 1) sbq[tile >> 1] = sbw;
        if (tile == 7) {
            u32x4 o;
            o.x = sbq[0];
            o.y = sbq[1];
            o.z = sbq[2];
            o.w = sbq[3];
            const unsigned off = brow == 0xFFFFFFFFu ? 0xFFFFFFF0u : brow * 32u + 16u * (unsigned)h;
            __builtin_amdgcn_raw_buffer_store_b128(o, brsrc[layer], (int)off, 0, 0);
        }
    };

    // the weight window of the first tile
    float4 wq[PFS];
    if (t_begin < ntiles) {
#pragma unroll
        for (int p = 0; p < PFS; ++p) wq[p] = load_w(rsrc, voff, group_off(wb_, p));
    }

    // The gather chain of a tile (vs_list -> smp_pidx / smp_loc -> point row, three dependent latencies) is issued one
    // tile ahead, one level per layer boundary: vector loads return in order, so by the time a level's consumer runs
    // its load is older than the whole weight window and costs no wait.
    RowFetch cur, nxt;
    if (t_begin < ntiles) {
        fetch_a<SEG>(P, t_begin, lane, wave, V0, S_valid, cur);
        fetch_b<SEG>(P, cur);
        fetch_c_pair(P, cur);
    }
    for (int n = 0, tile = t_begin; tile < ntiles; tile = walk.at(++n)) {
        // opaque per iteration: otherwise the scalar load offsets are hoisted out of this loop and spilled
        WBase wb = wb_;
        asm volatile("" : "+s"(wb.l1), "+s"(wb.l2), "+s"(wb.l3), "+s"(wb.l4));
        const auto goff = [&](int G) { return group_off(wb, G); };
        float xq[32];
        RowCtx ctx;
        const float4 *trow = P.pt_table + (int64_t)cur.urow * 64 + 4 * h;
        {
            const Camera cam = load_cam_wave(P.cr, cur.cid);   // scalar loads: no wait on the vector-load queue
            pair_inputs<SEG, true, false>(P, cur, cam, lane, xq, ctx);
        }
        if (TAPE) {
            // tape rows of the tile rows 8 i + l / 8 this lane stores: (valid sample) * K + slot
            constexpr int L = SEG ? SEG : 32;
            const int v_wave0 = __builtin_amdgcn_readfirstlane(V0 + tile * SPT + wave * (SPT / WAVES));
            const int64_t row_base = (int64_t)(v_wave0 - V0) * K;   // uniform: the tile's first tape row
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int ld = l == 1 ? 264 : 256;
                trsrc[l] = __builtin_amdgcn_make_buffer_rsrc(P.tape[l] + row_base * ld, 0, 32 * ld * 4, 0x00020000);
                brsrc[l] = __builtin_amdgcn_make_buffer_rsrc(
                    P.tape_bits + ((int64_t)l * (int64_t)P.tape_bits_rows + row_base) * 8, 0, 32 * 32, 0x00020000);
            }
            zrsrc = __builtin_amdgcn_make_buffer_rsrc(P.tape_rowz + row_base, 0, 32 * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 8 * i + (lane >> 3), sl = r / L, slot = r - sl * L, v = v_wave0 + sl;
                tg[i] = (slot < K && v < S_valid) ? (unsigned)(sl * K + slot) : 0xFFFFFFFFu;
            }
            brow = ctx.row_ok ? (unsigned)((ctx.v_idx - v_wave0) * K + ctx.slot) : 0xFFFFFFFFu;
        }
        fetch_a<SEG>(P, walk.at(n + 1), lane, wave, V0, S_valid, nxt);   // a tile past the end loads row 0: harmless
        f32x16 acc[8];
        float X[128], Y[132];
        // ---- layer 1: the 60 encoded distances on top of the point's pt_table row ------------------------------
        {
            Ini r0;
#pragma unroll
            for (int q = 0; q < 4; ++q) r0.v[q] = trow[q];
            acc[0] = acc_from(r0);
        }
        layer_stream<32, 8, 0, PFS>(
            rsrc, voff, goff, wq, xq, acc,
            [&](int m) {
                Ini r;
#pragma unroll
                for (int q = 0; q < 4; ++q) r.v[q] = trow[8 * m + q];
                return r;
            },
            [&](int m, int i) {
                if (m > 0 && i < 16 / FPM) {
                    float t16[16];
#pragma unroll
                    for (int q = 0; q < FPM; q += 2) {
                        float u, v;
                        leaky2(acc[m - 1][FPM * i + q], acc[m - 1][FPM * i + q + 1], u, v);
                        X[16 * (m - 1) + FPM * i + q] = to_a(u);
                        X[16 * (m - 1) + FPM * i + q + 1] = to_a(v);
                        t16[(FPM * i + q) & 15] = u;
                        t16[(FPM * i + q + 1) & 15] = v;
                    }
                    if (TAPE) tape_tile(0, m - 1, t16);
                }
            });
        // ---- layer 2 ---------------------------------------------------------------------------------------------
        {
            const f32x16 last = acc[7];
            acc[0] = acc_from(bias_ini(b1, 0, h));
            layer_stream<128, 8, NG_L1, PFS>(
                rsrc, voff, goff, wq, X, acc, [&](int m) { return bias_ini(b1, m, h); },
                [&](int m, int i) {
                    if (i < 16 / FPM) {
                        float t16[16];
#pragma unroll
                        for (int q = 0; q < FPM; q += 2) {
                            const int r = FPM * i + q;
                            float u, v;
                            if (m == 0) {
                                leaky2(last[r], last[r + 1], u, v);
                                X[112 + r] = to_a(u);
                                X[113 + r] = to_a(v);
                            } else {
                                leaky2(acc[m - 1][r], acc[m - 1][r + 1], u, v);
                                Y[16 * (m - 1) + r] = u;
                                Y[16 * (m - 1) + r + 1] = v;
                            }
                            t16[r & 15] = u;
                            t16[(r + 1) & 15] = v;
                        }
                        if (TAPE) tape_tile(m == 0 ? 0 : 1, m == 0 ? 7 : m - 1, t16);
                    }
                });
        }
        fetch_b<SEG>(P, nxt);
        // ---- layer 3: + the seven extra head inputs --------------------------------------------------------------
        {
            const f32x16 last = acc[7];
#pragma unroll
            for (int i = 0; i < 4; ++i) Y[128 + i] = ctx.ex[i];
            acc[0] = acc_from(bias_ini(b2, 0, h));
            layer_stream<132, 8, NG_L1 + NG_L2, PFS>(
                rsrc, voff, goff, wq, Y, acc, [&](int m) { return bias_ini(b2, m, h); },
                [&](int m, int i) {
                    if (i < 16 / FPM) {
                        float t16[16];
#pragma unroll
                        for (int q = 0; q < FPM; q += 2) {
                            const int r = FPM * i + q;
                            float u, v;
                            if (m == 0) {
                                leaky2(last[r], last[r + 1], u, v);
                                Y[112 + r] = u;
                                Y[113 + r] = v;
                            } else {
                                leaky2(acc[m - 1][r], acc[m - 1][r + 1], u, v);
                                X[16 * (m - 1) + r] = to_a(u);
                                X[16 * (m - 1) + r + 1] = to_a(v);
                            }
                            t16[r & 15] = u;
                            t16[(r + 1) & 15] = v;
                        }
                        if (TAPE) tape_tile(m == 0 ? 1 : 2, m == 0 ? 7 : m - 1, t16);
                    }
                });
        }
        fetch_c_pair(P, nxt);
        // ---- layer 4 + density head + K-aggregation in its shadows ----------------------------------------------
        float part = 0.f;   // this lane's share of <head, w4>
        const bool writer = ctx.row_ok && ctx.slot == 0;
        // the sample's aggregated features leave through a buffer store: lanes that do not write carry an offset
        // beyond the descriptor's range and the hardware drops them -- a branch around a plain store would split the
        // layer's basic block 32 times per tile.  Layout: agg_idx4 (whole blocks of 32 samples, one contiguous KiB per
        // load of the colour kernel).
        const int v_wave = __builtin_amdgcn_readfirstlane(V0 + tile * SPT + wave * (SPT / WAVES));
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
            P.agg + (int64_t)(v_wave >> 5) * 8192, 0, 65536, 0x00020000);
        // (a wave's samples are consecutive: they lie in the block of the first one or, when a pass of the
        // early-termination loop starts at an odd sample, in the next)
        const int ooff =
            writer ? 32768 * ((ctx.v_idx >> 5) - (v_wave >> 5)) + 16 * (ctx.v_idx & 31) + 1024 * h : 0x40000000;
        float4 hw[4], hw_nx[4];   // density-head weights of the output tile being sunk / of the next one
        float o4[4];
        // One output value of the finished tile per 8 MFMAs, ALL of its work behind one MFMA: every gap between two
        // MFMAs that holds vector-ALU work costs ~18 cycles plus ~4.5 per further instruction (tools/ub_mfma_dep.hip:
        // one wave per SIMD does not overlap its own VALU and MFMA work), so the sink is 7 instructions in one place:
        // LeakyReLU (2), density product (1), neighbour weight (1) and the three fused DPP adds of the K-sum, which run
        // as a pipeline over consecutive values (p1 -> p2 -> p3): each add reads a register written one value earlier,
        // so no wait states are needed between them.  Value L = 16 t + r leaves the pipeline NS values later; every
        // fourth one completes a float4 that is stored.
        constexpr int NS = SEG == 16 ? 4 : 3;
        float p1 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f;
        auto store4 = [&](int L) {
            u32x4 v;
            v.x = __float_as_uint(o4[0]);
            v.y = __float_as_uint(o4[1]);
            v.z = __float_as_uint(o4[2]);
            v.w = __float_as_uint(o4[3]);
            // features 32 t + 8 q + 4 h + {0..3} -> chunk (k = 2 t + (q >> 1), hp = h, half q & 1) of the sample's block
            const int t = L >> 4, q = (L & 15) >> 2;
            __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ooff, 2048 * (2 * t + (q >> 1)) + 512 * (q & 1), 0);
        };
        auto stage = [&](int L) {   // value L leaves the pipeline, the others advance one step
            float s, n3, n2, n4 = 0.f;
            if (SEG == 16) {
                asm volatile("s_nop 1\n\t"
                             "v_add_f32_dpp %0, %4, %4 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %2, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : "=&v"(s), "=&v"(n4), "=&v"(n3), "=&v"(n2)
                             : "v"(p4), "v"(p3), "v"(p2), "v"(p1));
            } else {
                asm volatile("s_nop 1\n\t"
                             "v_add_f32_dpp %0, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %1, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %2, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : "=&v"(s), "=&v"(n3), "=&v"(n2)
                             : "v"(p3), "v"(p2), "v"(p1));
            }
            p4 = n4;
            p3 = n3;
            p2 = n2;
            if (L >= 0) {
                o4[L & 3] = s;
                if ((L & 3) == 3) store4(L);
            }
        };
        float sv = 0.f;
        float g2t[16];   // TAPE: the activated values of the output tile being sunk
#ifdef PNR_PK_LEAKY
        // two values at a time: LeakyReLU with one packed multiply, the density product as one v_pk_fma_f32, the neighbour
        // weight as one v_pk_mul_f32 -- 3.5 instructions per pair instead of 8 in front of the DPP adds
        f32x2 part2 = {0.f, 0.f};
        const f32x2 wgt2 = {ctx.wgt, ctx.wgt};
        auto sink2 = [&](int t, int r, float a0, float a1) {
            float u, v;
            leaky2(a0, a1, u, v);
            if (TAPE) {
                g2t[r] = u;
                g2t[r + 1] = v;
            }
            const float4 w = hw[r >> 2];
            const f32x2 uv = {u, v};
            const f32x2 wv = (r & 2) ? f32x2{w.z, w.w} : f32x2{w.x, w.y};
            f32x2 p;
            asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(part2) : "v"(uv), "v"(wv));
            asm("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(uv), "v"(wgt2));
            stage(16 * t + r - NS);
            p1 = p.x;
            stage(16 * t + r + 1 - NS);
            p1 = p.y;
        };
#endif
        auto sink = [&](int t, int r, float a) {
            const float v = leaky(a);
            if (TAPE) g2t[r] = v;
            const float4 w = hw[r >> 2];
            part += v * ((r & 3) == 0 ? w.x : (r & 3) == 1 ? w.y : (r & 3) == 2 ? w.z : w.w);
            if (SEG != 0) {
                stage(16 * t + r - NS);
                p1 = v * ctx.wgt;
            } else {
                // K > 16: the sample's rows are not a DPP segment; shuffle sum per value
                sv = v * ctx.wgt;
                o4[r & 3] = seg_sum<SEG>(sv, K, lane);
                if ((r & 3) == 3) store4(16 * t + r);
            }
        };
        {
            const f32x16 last = acc[7];
            acc[0] = acc_from(bias_ini(b3, 0, h));
#pragma unroll
            for (int q = 0; q < 4; ++q) hw_nx[q] = *reinterpret_cast<const float4 *>(w4t + (0 * 2 + h) * 16 + 4 * q);
            layer_stream<128, 8, NG_L1 + NG_L2 + NG_L3, PFS>(
                rsrc, voff, goff, wq, X, acc, [&](int m) { return bias_ini(b3, m, h); },
                [&](int m, int i) {
                    if (m == 0 && i < 16 / FPM) {
                        float t16[16];
#pragma unroll
                        for (int q = 0; q < FPM; q += 2) {
                            float u, v;
                            leaky2(last[FPM * i + q], last[FPM * i + q + 1], u, v);
                            X[112 + FPM * i + q] = to_a(u);
                            X[113 + FPM * i + q] = to_a(v);
                            t16[(FPM * i + q) & 15] = u;
                            t16[(FPM * i + q + 1) & 15] = v;
                        }
                        if (TAPE) tape_tile(2, 7, t16);
                    }
                    if (m > 0 && i == 0) {
                        // the head weights of tile m - 1 become current, those of tile m are fetched
#pragma unroll
                        for (int q = 0; q < 4; ++q) hw[q] = hw_nx[q];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            hw_nx[q] = *reinterpret_cast<const float4 *>(w4t + (m * 2 + h) * 16 + 4 * q);
                    }
                    // value r of output tile m - 1 behind MFMA 8r + 4 of tile m (16 values, 128 MFMAs)
#if PNR_SINK_PER == 1
                    if (m > 0 && (i & 7) == 4) sink(m - 1, i >> 3, acc[m - 1][i >> 3]);
#else
                    if (m > 0 && (i % (8 * PNR_SINK_PER)) == 4) {
#ifdef PNR_PK_LEAKY
                        if (SEG != 0) {
#pragma unroll
                            for (int e = 0; e < 16; e += 2) sink2(m - 1, e, acc[m - 1][e], acc[m - 1][e + 1]);
                        } else
#endif
                        {
#pragma unroll
                            for (int e = 0; e < PNR_SINK_PER; ++e) {
                                const int r = (i / (8 * PNR_SINK_PER)) * PNR_SINK_PER + e;
                                sink(m - 1, r, acc[m - 1][r]);
                            }
                        }
                        if (TAPE) tape_tile(3, m - 1, g2t);
                    }
#endif
                });
        }
        // behind the tile's last MFMA: the sink of output tile 7, then the pipeline drains
#pragma unroll
        for (int q = 0; q < 4; ++q) hw[q] = hw_nx[q];
#ifdef PNR_PK_LEAKY
        if (SEG != 0) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) sink2(7, e, acc[7][e], acc[7][e + 1]);
            part += part2.x + part2.y;
        } else
#endif
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) sink(7, r, acc[7][r]);
        }
        if (TAPE) {
            tape_tile(3, 7, g2t);
            tape_flush(3, 7);
        }
        if (SEG != 0) {
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                stage(128 - NS + e);
                p1 = 0.f;
            }
        }
        part += __shfl_xor(part, 32, 64);
        if (TAPE)   // the row's density pre-activation (lane half 0 of an existing row; the others out of range)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(part + b4), zrsrc,
                                                  (brow == 0xFFFFFFFFu || h) ? 0x7FFFFFF0 : (int)(brow * 4u), 0, 0);
        const float alpha = fmaxf(part + b4, 0.f);
        const float sigma = seg_sum<SEG>(alpha * ctx.wgt, K, lane);
        if (writer && h == 0) {
            P.smp_sigma[ctx.v_idx] = sigma;
            if (P.smp_sig_s) P.smp_sig_s[ctx.s] = sigma;
        }
        cur = nxt;
    }
}


// ================================================================================================
// Dense units: the pair kernel for K that does not divide a 32-row tile.
//
// k_shade_pairs<16> gives a sample with 9..16 neighbours a 16-lane segment: at K = 12 (BASELINE cfg[4], the reference's
// ScanNet scripts) a quarter of every MFMA multiplies padding (0.66 of the peak).  Here SU consecutive samples fill TU
// consecutive tiles of ONE wave with no gap between them -- K = 12: 8 samples x 12 rows = 96 rows = 3 tiles exactly -- so
// a sample's rows sit at any multiple-of-K offset and may straddle two tiles.  Sums over a sample's rows therefore
// leave the register file, in two steps that cost the vector ALU next to nothing (a SIMD does not overlap VALU and
// MFMA work, but LDS instructions between two MFMAs are as free as loads; the 16-lane segment spends four DPP adds per
// value, 512 per tile):
//   1. every finished layer-4 value, times its row's normalised weight (k_pair_weights), goes into a wave-private
//      ROW BUFFER in LDS: 32 rows x 256 features, one ds_write_b128 per four values (rows 1040 bytes apart: conflict-free);
//   2. during the NEXT tile's layers 2 and 3 the wave reads the buffer back TRANSPOSED -- lane l takes features
//      4l .. 4l+3 of row r -- and adds them into the slot of row r's sample with four ds_add_f32: every lane owns its
//      addresses (no conflict, a fixed order row after row), one address add per row is all the VALU does;
//   3. a sample whose last row was in that tile is complete: its slot is read, stored in the colour kernel's layout and
//      cleared (four candidates per tile, predicated: no branch inside the stream).  A slot is reused every fourth
//      sample, which is safe for K >= 11 (the rows of samples s and s + 4 never share a tile).
// The first version added every value straight into the sample's slot (rows of a sample meet in one address): twelve-way
// same-address float atomics cost ~450 cycles per instruction, 0.71 of the peak.
// ================================================================================================
constexpr int DENSE_NS = 4;                      // sample slots per wave
constexpr int DENSE_ROW_B = 1040;                // bytes between rows of the row buffer (1024 + 16: bank skew)
constexpr int DENSE_ROWBUF_B = 32 * DENSE_ROW_B;
constexpr int DENSE_SLOTS_B = DENSE_NS * 1024;
constexpr int DENSE_SIG_B = 64;                  // density sums of the slots
constexpr int DENSE_DUMMY_B = 1024 + 64;         // where predicated-off LDS writes go
constexpr int DENSE_WAVE_B = DENSE_ROWBUF_B + DENSE_SLOTS_B + DENSE_SIG_B + DENSE_DUMMY_B;   // 38 528

__global__ void __launch_bounds__(TPB, 1) k_shade_pairs_dense(ShadeParams P, int SU, int TU)
{
    extern __shared__ float dense_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int K = P.K;
    const int V0 = P.n_sel[P.i_v0], S_valid = P.n_sel[P.i_v1];
    const int SPU = SU * WAVES;                        // samples per workgroup unit
    const int nunits = (S_valid - V0 + SPU - 1) / SPU;
    const TileWalk walk((int)blockIdx.x, (int)gridDim.x, nunits);   // XCD-aware order of the units
    const unsigned rowbuf = (unsigned)(uintptr_t)dense_lds + (unsigned)(wave * DENSE_WAVE_B);
    const unsigned slots = rowbuf + DENSE_ROWBUF_B, sig = slots + DENSE_SLOTS_B, dummy = sig + DENSE_SIG_B;
    const unsigned wr_base = rowbuf + (unsigned)(j * DENSE_ROW_B + 16 * h);   // + 128 t4 + 32 q: features 32 t4 + 8 q + 4 h
    const unsigned rd_base = rowbuf + 16u * lane;                             // + 1040 r: features 4 l .. 4 l + 3 of row r
    const unsigned lane16 = 16u * lane;

    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wbuf), 0, (int)P.wbytes, 0x00020000);
    const int voff = lane * 16;
    const WBase wb_ = {(int)(P.w32b_off * 4), (int)(P.w_off[1] * 4), (int)(P.w_off[2] * 4), (int)(P.w_off[3] * 4)};
    const float *b1 = P.wbuf + P.b_off[1], *b2 = P.wbuf + P.b_off[2], *b3 = P.wbuf + P.b_off[3];
    const float *w4t = P.wbuf + P.w4acc_off;
    const float b4 = P.wbuf[P.b_off[4]];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // slots, density sums and the dummy area start at zero (the row buffer is rewritten completely by every tile)
    for (int s = 0; s < DENSE_NS; ++s) asm volatile("ds_write_b128 %0, %1" ::"v"(slots + 1024u * s + lane16), "v"(zero4));
    asm volatile("ds_write_b128 %0, %1" ::"v"(dummy + lane16), "v"(zero4));
    if (lane < 8) asm volatile("ds_write_b128 %0, %1" ::"v"(sig + lane16), "v"(zero4));   // sig (64 B) + dummy tail (64 B)
    lds_wait();

    // rows of tile t of workgroup unit wu: row u = 32 t + j of the wave's unit = slot (u mod K) of its sample u / K
    auto fetch_a_dense = [&](int wu, int t, RowFetch &f, int &sl) {
        const int u = 32 * t + j;
        sl = u / K;
        f.slot = u - sl * K;
        f.v_idx = V0 + (wu * WAVES + wave) * SU + sl;
        f.smp_ok = sl < SU && f.v_idx < S_valid;
        f.row_ok = f.smp_ok;
        f.s = P.vs_list[f.row_ok ? f.v_idx : 0];
    };
    auto fetch_b_dense = [&](RowFetch &f) {
        fetch_b<-1>(P, f);
        f.wgt = P.smp_wgt[(int64_t)f.s * K + (f.row_ok ? f.slot : 0)];
    };

    // ---- the previous tile's rows -> sample slots (steps 2 and 3 of the header) ---------------------------------------
    // uniform state of the tile whose rows are in the row buffer
    int p_have = 0, p_si0 = 0, p_rem0 = 0, p_end = 0, p_vfirst = 0;
    f32x4 pend[2];
    // slot base of row r of the previous tile (wave-uniform: scalar ALU)
    auto row_slot_base = [&](int r) -> unsigned {
        const int x = p_rem0 + r;
        const int si = p_si0 + (x >= K ? 1 : 0) + (x >= 2 * K ? 1 : 0) + (x >= 3 * K ? 1 : 0);
        return p_have ? slots + 1024u * (unsigned)(si & (DENSE_NS - 1)) : dummy;
    };
    auto reduce_issue = [&](int r0) {   // reads of rows r0, r0 + 1
        pend[0] = lds_read4(rd_base, r0 * DENSE_ROW_B);
        pend[1] = lds_read4(rd_base, (r0 + 1) * DENSE_ROW_B);
    };
    auto reduce_add = [&](int r0) {     // adds of rows r0, r0 + 1 (read one step earlier)
        lds_wait();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const unsigned a = lane16 + row_slot_base(r0 + e);
            lds_add(a, pend[e].x, 0);
            lds_add(a, pend[e].y, 4);
            lds_add(a, pend[e].z, 8);
            lds_add(a, pend[e].w, 12);
        }
    };
    float sg_carry[4] = {0.f, 0.f, 0.f, 0.f};
    int sg_v[4] = {-1, -1, -1, -1};
    // samples whose last row was in the previous tile: out and cleared.  Four candidates, everything predicated.
    auto complete_prev = [&]() {
        lds_wait();
        const int fk = 2 * (lane >> 3) + (((lane >> 1) & 3) >> 1), fhp = lane & 1, fh = (lane >> 1) & 1;
        f32x4 sum[4];
        float sg[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int sc = p_si0 + c;
            const unsigned sb = slots + 1024u * (unsigned)(sc & (DENSE_NS - 1));
            sum[c] = lds_read4(sb + lane16, 0);
            asm volatile("ds_read_b32 %0, %1" : "=v"(sg[c]) : "v"(sig + 4u * (unsigned)(sc & (DENSE_NS - 1))));
        }
        lds_wait();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int sc = p_si0 + c;
            const int v = p_vfirst + sc;
            const bool done = p_have && (sc + 1) * K <= p_end && sc < SU && v < S_valid;
            const unsigned sb = slots + 1024u * (unsigned)(sc & (DENSE_NS - 1));
            lds_write4((done ? sb : dummy) + lane16, zero4, 0);
            asm volatile("ds_write_b32 %0, %1" ::"v"(done ? sig + 4u * (unsigned)(sc & (DENSE_NS - 1)) : dummy + 1024u),
                         "v"(0.f));
            // the sample's block of 32 in the colour kernel's layout (agg_idx4); lanes of a sample that is not complete
            // carry an offset beyond the descriptor's range: the hardware drops the store
            const int vb = done ? v : 0;
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                P.agg + (int64_t)(vb >> 5) * 8192, 0, 32768, 0x00020000);
            const int ooff = done ? 2048 * fk + 1024 * fhp + 512 * fh + 16 * (vb & 31) : 0x40000000;
            u32x4 o;
            o.x = __float_as_uint(sum[c].x);
            o.y = __float_as_uint(sum[c].y);
            o.z = __float_as_uint(sum[c].z);
            o.w = __float_as_uint(sum[c].w);
            __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, ooff, 0, 0);
            sg_carry[c] = sg[c];
            sg_v[c] = done ? v : -1;
        }
    };
    // the densities of the samples completed by complete_prev() (plain stores: called where branches cost nothing)
    auto store_sigmas = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (sg_v[c] >= 0 && lane == 0) {
                P.smp_sigma[sg_v[c]] = sg_carry[c];
                if (P.smp_sig_s) P.smp_sig_s[P.vs_list[sg_v[c]]] = sg_carry[c];
            }
            sg_v[c] = -1;
        }
    };

    int n_unit = 0, t_cur = 0;
    int wu_cur = walk.at(0);
    float4 wq[PFS];
    RowFetch cur, nxt;
    int sl_cur = 0, sl_nxt = 0;
    if (wu_cur < nunits) {
#pragma unroll
        for (int p = 0; p < PFS; ++p) wq[p] = load_w(rsrc, voff, group_off(wb_, p));
        fetch_a_dense(wu_cur, 0, cur, sl_cur);
        fetch_b_dense(cur);
        fetch_c_pair(P, cur);
    }
    while (wu_cur < nunits) {
        WBase wb = wb_;
        asm volatile("" : "+s"(wb.l1), "+s"(wb.l2), "+s"(wb.l3), "+s"(wb.l4));
        const auto goff = [&](int G) { return group_off(wb, G); };
        // the tile after this one: the unit's next tile, or tile 0 of the workgroup's next unit
        int t_nxt = t_cur + 1, n_nxt = n_unit, wu_nxt = wu_cur;
        if (t_nxt == TU) {
            t_nxt = 0;
            n_nxt = n_unit + 1;
            wu_nxt = walk.at(n_nxt);
        }
        float xq[32];
        RowCtx ctx;
        const float4 *trow = P.pt_table + (int64_t)cur.urow * 64 + 4 * h;
        {
            const Camera cam = load_cam_wave(P.cr, cur.cid);
            pair_inputs<-1, true, false>(P, cur, cam, lane, xq, ctx);
        }
        fetch_a_dense(wu_nxt, t_nxt, nxt, sl_nxt);            // a unit past the end loads entry 0: harmless
        f32x16 acc[8];
        float X[128], Y[132];
        // ---- layer 1 -----------------------------------------------------------------------------------------------
        {
            Ini r0;
#pragma unroll
            for (int q = 0; q < 4; ++q) r0.v[q] = trow[q];
            acc[0] = acc_from(r0);
        }
        layer_stream<32, 8, 0, PFS>(
            rsrc, voff, goff, wq, xq, acc,
            [&](int m) {
                Ini r;
#pragma unroll
                for (int q = 0; q < 4; ++q) r.v[q] = trow[8 * m + q];
                return r;
            },
            [&](int m, int i) {
                if (m > 0 && i < 16 / FPM) {
#pragma unroll
                    for (int q = 0; q < FPM; q += 2) {
                        float u, v;
                        leaky2(acc[m - 1][FPM * i + q], acc[m - 1][FPM * i + q + 1], u, v);
                        X[16 * (m - 1) + FPM * i + q] = to_a(u);
                        X[16 * (m - 1) + FPM * i + q + 1] = to_a(v);
                    }
                }
                if (m == 7 && i == 0) reduce_issue(0);   // the previous tile's rows 0, 1
            });
        // ---- layer 2: + rows 0..15 of the previous tile into their slots -------------------------------------------
        {
            const f32x16 last = acc[7];
            acc[0] = acc_from(bias_ini(b1, 0, h));
            layer_stream<128, 8, NG_L1, PFS>(
                rsrc, voff, goff, wq, X, acc, [&](int m) { return bias_ini(b1, m, h); },
                [&](int m, int i) {
                    if (i < 16 / FPM) {
#pragma unroll
                        for (int q = 0; q < FPM; q += 2) {
                            const int r = FPM * i + q;
                            float u, v;
                            if (m == 0) {
                                leaky2(last[r], last[r + 1], u, v);
                                X[112 + r] = to_a(u);
                                X[113 + r] = to_a(v);
                            } else {
                                leaky2(acc[m - 1][r], acc[m - 1][r + 1], Y[16 * (m - 1) + r], Y[16 * (m - 1) + r + 1]);
                            }
                        }
                    }
                    if (i == 0) {
                        reduce_add(2 * m);
                        reduce_issue(2 * m + 2);
                    }
                });
        }
        fetch_b_dense(nxt);
        // ---- layer 3: + rows 16..31 ----------------------------------------------------------------------------------
        {
            const f32x16 last = acc[7];
#pragma unroll
            for (int i = 0; i < 4; ++i) Y[128 + i] = ctx.ex[i];
            acc[0] = acc_from(bias_ini(b2, 0, h));
            layer_stream<132, 8, NG_L1 + NG_L2, PFS>(
                rsrc, voff, goff, wq, Y, acc, [&](int m) { return bias_ini(b2, m, h); },
                [&](int m, int i) {
                    if (i < 16 / FPM) {
#pragma unroll
                        for (int q = 0; q < FPM; q += 2) {
                            const int r = FPM * i + q;
                            float u, v;
                            if (m == 0) {
                                leaky2(last[r], last[r + 1], Y[112 + r], Y[113 + r]);
                            } else {
                                leaky2(acc[m - 1][r], acc[m - 1][r + 1], u, v);
                                X[16 * (m - 1) + r] = to_a(u);
                                X[16 * (m - 1) + r + 1] = to_a(v);
                            }
                        }
                    }
                    if (i == 0) {
                        reduce_add(16 + 2 * m);
                        if (m < 7) reduce_issue(16 + 2 * m + 2);
                    }
                });
        }
        fetch_c_pair(P, nxt);
        complete_prev();   // the row buffer is free again, the completed samples are on their way out
        // ---- layer 4: every finished value -> LeakyReLU, density product, weight; four values -> one ds_write_b128 --
        float part = 0.f;
        float4 hw[4], hw_nx[4];
        f32x4 o4;
        // value r of output tile t4 is feature 32 t4 + 8 (r >> 2) + 4 h + (r & 3) of the row
        auto sink = [&](int t4, int r, float a) {
            const float v = leaky(a);
            const float4 w = hw[r >> 2];
            part += v * ((r & 3) == 0 ? w.x : (r & 3) == 1 ? w.y : (r & 3) == 2 ? w.z : w.w);
            o4[r & 3] = v * ctx.wgt;
            if ((r & 3) == 3) lds_write4(wr_base, o4, 128 * t4 + 32 * (r >> 2));
        };
        {
            const f32x16 last = acc[7];
            acc[0] = acc_from(bias_ini(b3, 0, h));
#pragma unroll
            for (int q = 0; q < 4; ++q) hw_nx[q] = *reinterpret_cast<const float4 *>(w4t + (0 * 2 + h) * 16 + 4 * q);
            layer_stream<128, 8, NG_L1 + NG_L2 + NG_L3, PFS>(
                rsrc, voff, goff, wq, X, acc, [&](int m) { return bias_ini(b3, m, h); },
                [&](int m, int i) {
                    if (m == 0 && i < 16 / FPM) {
#pragma unroll
                        for (int q = 0; q < FPM; q += 2) {
                            float u, v;
                            leaky2(last[FPM * i + q], last[FPM * i + q + 1], u, v);
                            X[112 + FPM * i + q] = to_a(u);
                            X[113 + FPM * i + q] = to_a(v);
                        }
                    }
                    if (m > 0 && i == 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) hw[q] = hw_nx[q];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            hw_nx[q] = *reinterpret_cast<const float4 *>(w4t + (m * 2 + h) * 16 + 4 * q);
                    }
                    // the 16 values of output tile m - 1 in one block behind MFMA 4 of tile m
                    if (m > 0 && i == 4) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sink(m - 1, r, acc[m - 1][r]);
                    }
                });
        }
        // behind the tile's last MFMA: output tile 7, the row's density, the densities of the samples completed above
#pragma unroll
        for (int q = 0; q < 4; ++q) hw[q] = hw_nx[q];
#pragma unroll
        for (int r = 0; r < 16; ++r) sink(7, r, acc[7][r]);
        part += __shfl_xor(part, 32, 64);
        const float alpha = fmaxf(part + b4, 0.f);
        // (rows of one sample meet in one address here: one instruction per tile)
        lds_add(sig + 4u * (unsigned)(sl_cur & (DENSE_NS - 1)), h ? 0.f : alpha * ctx.wgt, 0);
        store_sigmas();
        // this tile becomes the previous one
        p_have = 1;
        p_si0 = (32 * t_cur) / K;
        p_rem0 = 32 * t_cur - p_si0 * K;
        p_end = 32 * (t_cur + 1);
        p_vfirst = V0 + (wu_cur * WAVES + wave) * SU;
        cur = nxt;
        sl_cur = sl_nxt;
        t_cur = t_nxt;
        n_unit = n_nxt;
        wu_cur = wu_nxt;
    }
    // the last tile's rows
    if (p_have) {
        lds_wait();
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 2) {
            reduce_issue(r0);
            reduce_add(r0);
        }
        complete_prev();
        store_sigmas();
    }
}

// Colour MLP: one lane-column per valid sample, 32 samples per wavefront.
// input 280 = [agg(256) | sin(view*2^f) (12) | cos(...) (12)] -> 128 -> 128 -> 128 -> 3, sigmoid, widen.
// The same stream structure as the pair kernel: the 268 weight groups of a tile (140 + 64 + 64) behind one rolling
// window that drains neither between the layers nor between the tiles, one activation block per output tile, biases a
// tile ahead; the 128 aggregated features of the NEXT tile are loaded (straight into AGPRs, where the first layer's
// MFMAs read them) as soon as the first layer has consumed the current ones, the sample -> ray -> direction chain of
// the next tile one level per layer boundary.
constexpr int PFC = 4;                         // divides 268
constexpr int NGC1 = 4 * 35, NGC2 = 4 * 16;    // groups of the first / of each of the other two layers
constexpr int NGC = NGC1 + 2 * NGC2;           // 268

// colour head on the last hidden layer: 128 -> 3, sigmoid, widen (studio_model.py:357-359).  The 16 weight quads of a
// colour are issued together, one colour ahead, and awaited once (left to hipcc, every quad is loaded and awaited on
// its own -- behind the weight window still in flight that is 48 drained queues per tile).
struct HeadW {
    float4 v[16];
};
__device__ __forceinline__ HeadW head_w(__amdgpu_buffer_rsrc_t rsrc, int w8, int h, int c)
{
    HeadW w;
#pragma unroll
    for (int i = 0; i < 16; ++i) w.v[i] = load_w(rsrc, 16 * h, w8 + 512 * c + 128 * (i >> 2) + 32 * (i & 3));
    return w;
}
__device__ __forceinline__ void color_head_f32(__amdgpu_buffer_rsrc_t rsrc, int w8, const float *__restrict__ b8, int h,
                                               HeadW w0, const float (&hC)[64], float (&rgb)[3], float (&sgo)[3])
{
    HeadW wc = w0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        HeadW wn;
        if (c < 2) wn = head_w(rsrc, w8, h, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            part += hC[4 * i + 0] * wc.v[i].x;
            part += hC[4 * i + 1] * wc.v[i].y;
            part += hC[4 * i + 2] * wc.v[i].z;
            part += hC[4 * i + 3] * wc.v[i].w;
        }
        part += __shfl_xor(part, 32, 64);
        const float z = part + b8[c];
        const float sg = 1.0f / (1.0f + expf(-z));
        sgo[c] = sg;
        rgb[c] = sg * (1.0f + 2.0f * 0.001f) - 0.001f;
        if (c < 2) wc = wn;
    }
}

struct ColorFetch {
    int v_idx, s, ray;
    bool ok;
    float sigma, dx, dy, dz;
};

// TAPE (training renders): the three hidden activations leave row-major ([S, 128] each: the colour MLP's part of
// pnr_render_backward's tape) through the same wave-private LDS blocks as the pair kernel's tape, and the colour head's
// sigmoid outputs with them -- the backward then recomputes nothing of the colour MLP (three row GEMMs and the head:
// a dozen launches of latency at the training batch size).
template <bool TAPE = false>
__global__ void __launch_bounds__(TPB, 1) k_shade_color(ShadeParams P)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int V0 = P.n_sel[P.i_v0], S_valid = P.n_sel[P.i_v1];
    constexpr int SPT = 32 * WAVES;
    const int ntiles = (S_valid - V0 + SPT - 1) / SPT;
    extern __shared__ float ctape_lds[];
    const unsigned tblk = TAPE ? (unsigned)(uintptr_t)ctape_lds + (unsigned)(wave * TAPE_WAVE_B) : 0u;
    const unsigned t_wr = tblk + (unsigned)(j * TAPE_ROW_B + 16 * h);
    const unsigned t_rd = tblk + (unsigned)((lane >> 3) * TAPE_ROW_B + (lane & 7) * 16);
    __amdgpu_buffer_rsrc_t crs[3];
    // block (layer pl, output tile pt) goes out: 8 lanes per row, whole 128-byte lines
    auto tape_flush = [&](int pl, int pt) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = lds_read4(t_rd, (pt & 1) * TAPE_BLK_B + i * 8 * TAPE_ROW_B);
        lds_wait();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 o;
            o.x = __float_as_uint(v[i].x);
            o.y = __float_as_uint(v[i].y);
            o.z = __float_as_uint(v[i].z);
            o.w = __float_as_uint(v[i].w);
            __builtin_amdgcn_raw_buffer_store_b128(o, crs[pl], (8 * i + (lane >> 3)) * 512 + 16 * (lane & 7) + 128 * pt, 0, 0);
        }
    };
    auto tape_tile = [&](int layer, int tile, const float *t) {
        if (tile > 0)
            tape_flush(layer, tile - 1);
        else if (layer > 0)
            tape_flush(layer - 1, 3);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
            lds_write4(t_wr, v, (tile & 1) * TAPE_BLK_B + 32 * q);
        }
    };
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wbuf), 0, (int)P.wbytes, 0x00020000);
    const int voff = lane * 16;
    const int w5_ = (int)(P.w_off[5] * 4), w6_ = (int)(P.w_off[6] * 4), w7_ = (int)(P.w_off[7] * 4);
    const float *b5 = P.wbuf + P.b_off[5], *b6 = P.wbuf + P.b_off[6], *b7 = P.wbuf + P.b_off[7], *b8 = P.wbuf + P.b_off[8];
    const int w8 = (int)(P.w_off[8] * 4);
    const float4 *agg4 = reinterpret_cast<const float4 *>(P.agg);
    if ((int)blockIdx.x >= ntiles) return;

    // first level of a tile's chain: the valid-sample index -> sample (colour kernels are launched with V0 = 0)
    auto fetch_s = [&](int tile, ColorFetch &f) {
        f.v_idx = V0 + tile * SPT + wave * 32 + j;
        f.ok = f.v_idx < S_valid;
        f.s = P.vs_list[f.ok ? f.v_idx : 0];
        f.sigma = P.smp_sigma[f.ok ? f.v_idx : 0];
    };
    auto fetch_ray = [&](ColorFetch &f) {
        f.s = f.ok ? f.s : 0;
        f.ray = P.smp_ray[f.s];
    };
    auto fetch_dir = [&](ColorFetch &f) {
        f.dx = P.dirs[3 * (int64_t)f.ray];
        f.dy = P.dirs[3 * (int64_t)f.ray + 1];
        f.dz = P.dirs[3 * (int64_t)f.ray + 2];
    };
    // features 8 c + 4 h + {0..3} of the lane's sample: one contiguous 512 B per lane half and load (agg_idx4)
    auto load_agg = [&](const ColorFetch &f, float *x) {
        const int vs = f.ok ? f.v_idx : 0;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const float4 a = agg4[agg_idx4(vs, 2 * (c >> 2) + ((c & 3) >> 1), h, c & 1)];
            x[4 * c + 0] = to_a(a.x);
            x[4 * c + 1] = to_a(a.y);
            x[4 * c + 2] = to_a(a.z);
            x[4 * c + 3] = to_a(a.w);
        }
    };

    ColorFetch cur, nxt;
    float x[140];
    fetch_s(blockIdx.x, cur);
    fetch_ray(cur);
    fetch_dir(cur);
    load_agg(cur, x);
    float4 wq[PFC];
    {
        const int w5 = w5_;
#pragma unroll
        for (int p = 0; p < PFC; ++p) wq[p] = load_w(rsrc, voff, w5 + p * 1024);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int w5 = w5_, w6 = w6_, w7 = w7_;
        asm volatile("" : "+s"(w5), "+s"(w6), "+s"(w7));
        const auto goff = [&](int G) {
            G = G >= NGC ? G - NGC : G;
            return G < NGC1 ? w5 + G * 1024 : G < NGC1 + NGC2 ? w6 + (G - NGC1) * 1024 : w7 + (G - NGC1 - NGC2) * 1024;
        };
        // view-direction encodings of this tile (its direction was fetched a tile ago)
        {
            float vx, vy, vz;
            rot_rows(P.Rw2c, cur.dx, cur.dy, cur.dz, vx, vy, vz);
            const float vv[3] = {vx, vy, vz};
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    float sn, cs;
                    fast_sincos_nb(vv[d] * (float)(1 << f), sn, cs);
                    x[128 + d * 4 + f] = to_a(h ? cs : sn);
                }
        }
        fetch_s(tile + gridDim.x, nxt);   // a tile past the end reads entry 0: harmless
        if (TAPE) {
            // the wave's 32 consecutive tape rows; the descriptors end behind the last existing one
            const int v0w = __builtin_amdgcn_readfirstlane(tile * SPT + wave * 32);
            const int nv = __builtin_amdgcn_readfirstlane(max(0, min(32, S_valid - V0 - v0w)));
#pragma unroll
            for (int l = 0; l < 3; ++l)
                crs[l] = __builtin_amdgcn_make_buffer_rsrc(P.ctape[l] + (int64_t)v0w * 128, 0, nv * 512, 0x00020000);
        }
        float hA[64], hB[64];
        Ini ini_next;
        f32x16 acc[4];
        // ---- layer 1 -----------------------------------------------------------------------------------------------
        acc[0] = acc_from(bias_ini(b5, 0, h));
        layer_stream<140, 4, 0, PFC>(
            rsrc, voff, goff, wq, x, acc, [&](int m) { return bias_ini(b5, m, h); },
            [&](int m, int i) {
                if (i != 0) return;
                if (m > 0) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) leaky2(acc[m - 1][r], acc[m - 1][r + 1], hA[16 * (m - 1) + r], hA[16 * (m - 1) + r + 1]);
                    if (TAPE) tape_tile(0, m - 1, &hA[16 * (m - 1)]);
                }
                if (m == 3) ini_next = bias_ini(b6, 0, h);
            });
        // the current tile's aggregated features are consumed: fetch the next tile's into the same registers
        fetch_ray(nxt);
        load_agg(nxt, x);
        // ---- layer 2 -----------------------------------------------------------------------------------------------
        {
            const f32x16 last = acc[3];
            acc[0] = acc_from(ini_next);
            layer_stream<64, 4, NGC1, PFC>(
                rsrc, voff, goff, wq, hA, acc, [&](int m) { return bias_ini(b6, m, h); },
                [&](int m, int i) {
                    if (i != 0) return;
                    if (m == 0) {
                        // values 48..63 are this layer's k-steps 48..63: long after MFMA 0
#pragma unroll
                        for (int r = 0; r < 16; r += 2) leaky2(last[r], last[r + 1], hA[48 + r], hA[49 + r]);
                        if (TAPE) tape_tile(0, 3, &hA[48]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; r += 2) leaky2(acc[m - 1][r], acc[m - 1][r + 1], hB[16 * (m - 1) + r], hB[16 * (m - 1) + r + 1]);
                        if (TAPE) tape_tile(1, m - 1, &hB[16 * (m - 1)]);
                    }
                    if (m == 3) ini_next = bias_ini(b7, 0, h);
                });
        }
        fetch_dir(nxt);
        // ---- layer 3 -----------------------------------------------------------------------------------------------
        float hC[64];
        {
            const f32x16 last = acc[3];
            acc[0] = acc_from(ini_next);
            layer_stream<64, 4, NGC1 + NGC2, PFC>(
                rsrc, voff, goff, wq, hB, acc, [&](int m) { return bias_ini(b7, m, h); },
                [&](int m, int i) {
                    if (i != 0) return;
                    if (m == 0) {
#pragma unroll
                        for (int r = 0; r < 16; r += 2) leaky2(last[r], last[r + 1], hB[48 + r], hB[49 + r]);
                        if (TAPE) tape_tile(1, 3, &hB[48]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; r += 2) leaky2(acc[m - 1][r], acc[m - 1][r + 1], hC[16 * (m - 1) + r], hC[16 * (m - 1) + r + 1]);
                        if (TAPE) tape_tile(2, m - 1, &hC[16 * (m - 1)]);
                    }
                });
        }
        const HeadW hw0 = head_w(rsrc, w8, h, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; r += 2) leaky2(acc[3][r], acc[3][r + 1], hC[48 + r], hC[49 + r]);
        if (TAPE) {
            tape_tile(2, 3, &hC[48]);
            tape_flush(2, 3);
        }
        float rgb[3], sgv[3];
        color_head_f32(rsrc, w8, b8, h, hw0, hC, rgb, sgv);
        if (cur.ok && h == 0) P.smp_out[cur.s] = make_float4(cur.sigma, rgb[0], rgb[1], rgb[2]);
        if (TAPE && cur.ok && h == 0) P.tape_sg[cur.v_idx - V0] = make_float4(sgv[0], sgv[1], sgv[2], 0.f);
        cur = nxt;
    }
}

// First-layer partial products of the listed points (pt_table, one 1-KiB row per point in accumulator order): the
// 224 point-only inputs [emb (32) | PE(emb, 3) (192)] of mlp_base layer 0, once per distinct neighbour point.
// One layer of 224 weight groups as a stream that does not drain between tiles; the finished output tile leaves as
// four 16-byte stores per lane behind the first MFMA of the next one (no VALU work at all), the next tile's embedding
// rows are fetched a tile ahead.
constexpr int PFP = 8;          // divides 224
constexpr int NGP = 8 * 28;     // 8 output tiles x 112 / 4 groups

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
    const int voff = lane * 16;
    const int wa_ = (int)(P.w32a_off * 4);
    const float *b0 = P.wbuf + P.b_off[0];
    if ((int)blockIdx.x >= ntiles) return;

    auto fetch_p = [&](int tile, int &pidx) {
        const int u = tile * PPT + wave * 32 + j;
        pidx = P.pt_list[u < U ? u : 0];
    };
    auto fetch_e = [&](int pidx, float4 (&e4)[4]) {
        const float4 *row = P.point_rows + (int64_t)pidx * 12 + 4 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) e4[q] = row[q];
    };
    int pidx_n;
    float4 e_cur[4], e_nxt[4];
    fetch_p(blockIdx.x, pidx_n);
    fetch_e(pidx_n, e_cur);
    fetch_p(blockIdx.x + gridDim.x, pidx_n);
    float4 wq[PFP];
#pragma unroll
    for (int p = 0; p < PFP; ++p) wq[p] = load_w(rsrc, voff, wa_ + p * 1024);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int wa = wa_;
        asm volatile("" : "+s"(wa));
        const auto goff = [&](int G) { return wa + (G >= NGP ? G - NGP : G) * 1024; };
        const float e[16] = {e_cur[0].x, e_cur[0].y, e_cur[0].z, e_cur[0].w, e_cur[1].x, e_cur[1].y, e_cur[1].z, e_cur[1].w,
                             e_cur[2].x, e_cur[2].y, e_cur[2].z, e_cur[2].w, e_cur[3].x, e_cur[3].y, e_cur[3].z, e_cur[3].w};
        float x0[112];
        point_inputs<true, false>(e, x0);   // branch-free sincos, 9e-8 absolute (libm's costs 150 instructions a call)
        // the next tile's embedding rows (their point index was fetched a tile ago), the point index of the one after
        fetch_e(pidx_n, e_nxt);
        fetch_p(tile + 2 * gridDim.x, pidx_n);
        const int u = tile * PPT + wave * 32 + j;
        float4 *dst = P.pt_table + (int64_t)u * 64 + 4 * h;   // rows beyond U: the table's padding rows
        auto store_tile = [&](int B, const f32x16 &a) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[8 * B + q] = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
        };
        f32x16 acc[8];
        acc[0] = acc_from(bias_ini(b0, 0, h));
        layer_stream<112, 8, 0, PFP>(
            rsrc, voff, goff, wq, x0, acc, [&](int m) { return bias_ini(b0, m, h); },
            [&](int m, int i) {
                if (m > 0 && i == 0) store_tile(m - 1, acc[m - 1]);
            });
        store_tile(7, acc[7]);
#pragma unroll
        for (int q = 0; q < 4; ++q) e_cur[q] = e_nxt[q];
    }
}

void launch_point_part_fp32(dim3 grid, hipStream_t stream, const ShadeParams &P)
{
    hipLaunchKernelGGL(k_point_part_f32, grid, dim3(TPB), 0, stream, P);
}

// the training render: the same kernel writing the backward's activation tape (ShadeParams.tape) as it goes
bool launch_pairs_fp32_tape(int seg, dim3 grid, hipStream_t stream, const ShadeParams &P)
{
    const size_t lds = (size_t)WAVES * TAPE_WAVE_B;
    if (seg == 8)
        hipLaunchKernelGGL((k_shade_pairs<8, true>), grid, dim3(TPB), lds, stream, P);
    else if (seg == 16)
        hipLaunchKernelGGL((k_shade_pairs<16, true>), grid, dim3(TPB), lds, stream, P);
    else
        return false;
    return true;
}

void launch_pairs_fp32(int seg, dim3 grid, hipStream_t stream, const ShadeParams &P)
{
    if (seg == 8)
        hipLaunchKernelGGL(k_shade_pairs<8>, grid, dim3(TPB), 0, stream, P);
    else if (seg == 16)
        hipLaunchKernelGGL(k_shade_pairs<16>, grid, dim3(TPB), 0, stream, P);
    else
        hipLaunchKernelGGL(k_shade_pairs<0>, grid, dim3(TPB), 0, stream, P);
}

int launch_pairs_fp32_dense(int su, int tu, dim3 grid, hipStream_t stream, const ShadeParams &P)
{
    const size_t lds = (size_t)WAVES * DENSE_WAVE_B;    // 154 KB of the CU's 160
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(k_shade_pairs_dense), (int)lds);
    if (rc != PNR_OK) return rc;
    hipLaunchKernelGGL(k_shade_pairs_dense, grid, dim3(TPB), lds, stream, P, su, tu);
    return PNR_OK;
}

void launch_color_fp32(dim3 grid, hipStream_t stream, const ShadeParams &P)
{
    if (P.ctape[0])
        hipLaunchKernelGGL(k_shade_color<true>, grid, dim3(TPB), (size_t)WAVES * TAPE_WAVE_B, stream, P);
    else
        hipLaunchKernelGGL(k_shade_color<false>, grid, dim3(TPB), 0, stream, P);
}

}  // namespace pnr

