// Shade stage, default mode (PNR_PRECISION_BF16X3): v_mfma_f32_32x32x16_bf16 on hi/lo splits, weights shared by the
// four waves through an LDS ring filled by LDS-DMA.  See pnr_shade_common.h for the design overview.
#include <type_traits>

#include "pnr_shade_common.h"

namespace pnr {

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

// A sink callable as sink(tile, s, const f32x16 &prev) is a STEP sink: instead of single values it is handed the whole
// finished accumulator of output tile `tile` at every k-step s >= 2 of the following tile and decides itself what to
// do at which step (k_point_part: a 4x4 transpose across the lanes of a quad spread over the k-steps).
template <typename Sink>
constexpr bool kStepSink = std::is_invocable_v<Sink &, int, int, const f32x16 &>;

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
                bias_wait(breg);
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
            const bool do_split = SPLIT_OUT && m > 0 && s >= S0 && s < S0 + 8;
            const int sp = s - S0;
            float v0 = 0.f, v1 = 0.f, r0 = 0.f, r1 = 0.f;
            __bf16 h0, h1;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[s], acc, 0, 0, 0);
            if (do_split) {
                v0 = leaky(prev[2 * sp]);
                v1 = leaky(prev[2 * sp + 1]);
                h0 = (__bf16)v0;
                h1 = (__bf16)v1;
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[s], acc, 0, 0, 0);
            if (do_split) {
                r0 = v0 - (float)h0;
                r1 = v1 - (float)h1;
            }
            if (s >= S_BF && want_bias) bias_quarter(breg, s - S_BF, ring.acc0);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[s], acc, 0, 0, 0);
            if (do_split) {
                const int kk = 2 * (m - 1) + sp / 4, j0 = (2 * sp) % 8;
                yh[kk][j0] = h0;
                yh[kk][j0 + 1] = h1;
                yl[kk][j0] = (__bf16)r0;
                yl[kk][j0 + 1] = (__bf16)r1;
            }
            if constexpr (!SPLIT_OUT && kStepSink<Sink>) {
                // a step sink sees the whole previous tile and schedules its own work over the k-steps (from 2 on)
                if (m > 0 && s >= 2) sink(m - 1, s, prev);
            } else if (!SPLIT_OUT && m > 0) {
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
            if (s == S_MID) {
                // ---- mid-tile: tile T+1 has landed everywhere, slot of tile T-1 is free ---------------------------
                if (m + 2 < MT)
                    wait_vm<R_SAME>();
                else
                    wait_vm<R_NX>();
                __builtin_amdgcn_s_barrier();
            }
            // ---- DMA of tile T+3 into the freed slot: two 1-KiB pieces per k-step behind the barrier, so the
            //      scalar address arithmetic hides between MFMAs instead of stalling the matrix pipe in one burst
            if (s > S_MID) {
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
    if constexpr (!SPLIT_OUT && kStepSink<Sink>) {
#pragma unroll
        for (int s = 2; s < KS; ++s) sink(MT - 1, s, prev);
    } else if (!SPLIT_OUT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sink(MT - 1, r, OUT_LEAKY ? leaky(prev[r]) : prev[r]);
    }
    if (SPLIT_OUT) {
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
            const bool do_split = B > 0 && xs >= 1;
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
            if (s == 3) {
                if (m + 2 < MT)
                    wait_vm<R_SAME>();
                else
                    wait_vm<R_NX>();
                __builtin_amdgcn_s_barrier();
            }
            if (s > 3) {
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
    {
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
        // Rows leave through the layer's step sink, between the MFMAs of the following output tile, as QUAD-COALESCED
        // stores.  In accumulator order lane (j, h) owns chunks q = 0..3 (16 B each) of row block t of ITS row, so a
        // plain store instruction would touch 64 different rows; scattered 16-byte accesses keep the texture-address
        // unit busy for ~100 cycles per instruction (tools/ub_gather.hip), 4 waves x 32 stores x 8 row blocks made
        // this kernel TA-bound at 2.5x its MFMA time.  A 4x4 transpose across the four lanes of a quad (rows 4a..4a+3,
        // same h) turns "my four chunks" into "chunk p of the quad's four rows": store i then writes 64 contiguous
        // bytes of row 4a+i per quad.  Two DPP stages (quad_perm xor 1, xor 2), one exchange unit per k-step.
        // Lanes beyond U write into the table's 128 padding rows (no branch: it would split the layer's basic block).
        const int pq = lane & 3;
        float4 *dst = P.pt_table + (int64_t)(u - pq) * 64 + 4 * h + pq;
        const bool odd = lane & 1, up = lane & 2;
        float t1[16];
        auto sink = [&](int t, int s, const f32x16 &pv) {
            // value r = 4q + c of the row block: chunk q, component c.  Stage 1 pairs chunks (0,1) and (2,3):
            // t1[4q + c], unit n = 0..7 = (pair n >> 2, component n & 3) at k-steps 2..9
            if (s >= 2 && s < 10) {
                const int n = s - 2, q0 = 2 * (n >> 2), c = n & 3;
                const float a = pv[4 * q0 + c], b = pv[4 * (q0 + 1) + c];
                const float ax = dpp_mov<0xB1>(a), bx = dpp_mov<0xB1>(b);
                t1[4 * q0 + c] = odd ? bx : a;
                t1[4 * (q0 + 1) + c] = odd ? b : ax;
            }
            // stage 2 pairs chunks (0,2) and (1,3): two components per k-step, the two finished rows leave at once
            if (s >= 10 && s < 14) {
                const int i = (s - 10) >> 1;
                if (((s - 10) & 1) == 0) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float a = t1[4 * i + c], b = t1[4 * (i + 2) + c];
                        const float ax = dpp_mov<0x4E>(a), bx = dpp_mov<0x4E>(b);
                        t1[4 * i + c] = up ? bx : a;
                        t1[4 * (i + 2) + c] = up ? b : ax;
                    }
                } else {
#pragma unroll
                    for (int c = 2; c < 4; ++c) {
                        const float a = t1[4 * i + c], b = t1[4 * (i + 2) + c];
                        const float ax = dpp_mov<0x4E>(a), bx = dpp_mov<0x4E>(b);
                        t1[4 * i + c] = up ? bx : a;
                        t1[4 * (i + 2) + c] = up ? b : ax;
                    }
                    dst[(int64_t)i * 64 + 8 * t] = make_float4(t1[4 * i], t1[4 * i + 1], t1[4 * i + 2], t1[4 * i + 3]);
                    dst[(int64_t)(i + 2) * 64 + 8 * t] =
                        make_float4(t1[4 * (i + 2)], t1[4 * (i + 2) + 1], t1[4 * (i + 2) + 2], t1[4 * (i + 2) + 3]);
                }
            }
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
    // XCD-aware tile order (TileWalk).  With a contiguous tile range per WORKGROUP instead, the 32 CUs of an XCD stream
    // 4 MB of unrelated rows through the 4 MB L2 per tile time and nearly every gather misses (rocprofv3 FETCH_SIZE:
    // 14.9 GB per launch for 10.8 GB gathered).
    const TileWalk walk((int)blockIdx.x, (int)gridDim.x, ntiles);
    const int t_begin = walk.at(0), t_end = ntiles;
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
    ring_start<8>(rsrc, wb_, nullptr, lane, tid, wave_u, lds, ring);
    RowFetch cur, nxt;
    fetch_a<SEG>(P, t_begin, lane, wave, V0, S_valid, cur);
    fetch_b<SEG>(P, cur);
    fetch_c_pair(P, cur);
    for (int n = 0, tile = t_begin; tile < t_end; tile = walk.at(++n)) {
        int wb = wb_, w1 = w1_, w2 = w2_, w3 = w3_;
        asm volatile("" : "+s"(wb), "+s"(w1), "+s"(w2), "+s"(w3));
        const Camera cam = load_cam_wave(P.cr, cur.cid);
        // first level of the next tile's gather chain (a tile past the end loads row 0: harmless); the other two
        // levels follow at the layer boundaries
        fetch_a<SEG>(P, walk.at(n + 1), lane, wave, V0, S_valid, nxt);
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
        RowCtx ctx;
        bf16x8 xqh[4], xql[4];
        {
            float xq[32];
            pair_inputs<SEG, true>(P, cur, cam, lane, xq, ctx);
#pragma unroll
            for (int s = 0; s < 4; ++s) split8(&xq[8 * s], xqh[s], xql[s]);
        }
        bf16x8 xh[17], xl[17], yh[17], yl[17];
        dense_layer1b_bf16<16>(rsrc, wb, w1, b1, lane, tid, wave_u, lds, ring, xqh, xql, pin, yh, yl);
        fetch_b<SEG>(P, nxt);
        // layer 2's output (+ the 7 extra head inputs as k-step 16) goes to xh/xl
        dense_layer_bf16<16, 8, 17, true>(rsrc, w1, w2, b1, b2, lane, tid, wave_u, lds, ring, yh, yl, xh, xl, StoreOut{nullptr});
        fetch_c_pair(P, nxt);
        {
            float v[8] = {ctx.ex[0], ctx.ex[1], ctx.ex[2], ctx.ex[3], 0.f, 0.f, 0.f, 0.f};
            split8(v, xh[16], xl[16]);
        }
        dense_layer_bf16<17, 8, 16, true>(rsrc, w2, w3, b2, b3, lane, tid, wave_u, lds, ring, xh, xl, yh, yl, StoreOut{nullptr});
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
        } else {
            float o[128];
            dense_layer_bf16<16, 8, 8, false, false, true>(rsrc, w3, wb, b3, nullptr, lane, tid, wave_u, lds, ring, yh,
                                                           yl, nullptr, nullptr, StoreOut{o});
            asm volatile("" ::"v"(nxt.dirz), "v"(nxt.urow));
            finish_rows<SEG, true>(P, lane, o, ctx);  // o: LeakyReLU already applied inside the layer
        }
        cur = nxt;
    }
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

void launch_point_part_bf16(dim3 grid, hipStream_t stream, const ShadeParams &P)
{
    hipLaunchKernelGGL(k_point_part, grid, dim3(TPB), 0, stream, P);
}

void launch_pairs_bf16(int seg, dim3 grid, hipStream_t stream, const ShadeParams &P)
{
    if (seg == 8)
        hipLaunchKernelGGL(k_shade_pairs_bf16<8>, grid, dim3(TPB), 0, stream, P);
    else if (seg == 16)
        hipLaunchKernelGGL(k_shade_pairs_bf16<16>, grid, dim3(TPB), 0, stream, P);
    else
        hipLaunchKernelGGL(k_shade_pairs_bf16<0>, grid, dim3(TPB), 0, stream, P);
}

void launch_color_bf16(dim3 grid, hipStream_t stream, const ShadeParams &P)
{
    hipLaunchKernelGGL(k_shade_color_bf16, grid, dim3(TPB), 0, stream, P);
}

}  // namespace pnr
