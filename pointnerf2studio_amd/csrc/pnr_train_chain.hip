// Training step, exact mode: the DATA gradients of the per-pair MLPs (mlp_head and mlp_base, eight F.linear backward
// halves of studio_model.py:319-353 under autograd) as ONE kernel -- the transposed counterpart of k_shade_pairs
// (pnr_shade_fp32.hip).
//
// Until round 3 every layer's data gradient was a row GEMM (k_gemm<false, false, EPI_MASK>: 128x128 tiles through LDS,
// the gradient written in place over the taped activation it masks): 0.55-0.6 of the fp32 MFMA peak, 6.8 of the
// backward's 16 ms at 65 536 rays, and 2 KiB of HBM traffic per row and layer.  Here a wave keeps 32 rows in registers
// for the whole chain, exactly as the render does:
//     dZ3 = w (dAGG + d sigma [z > 0] w4) * L'(G2)        (prologue: rank-2 per row, no GEMM)
//     dZ2 = (W3^T dZ3) * L'(G1),   [d extras | dZ1] = (W2^T dZ2) * [1 | L'(H2)],   dZ0 = (W1^T dZ1) * L'(H1),
//     dX0 = W0^T dZ0  (the 224 embedding columns only: positions are frozen, studio_utils.py:84-103)
// with the packed W^T streamed L2 -> VGPR as the MFMA A operand (k_pack_chain; one rolling window of loads that never
// drains), the gradients of a layer feeding the next layer's B operands straight from the accumulators (AGPR / VGPR
// alternating), the LeakyReLU masks taken from BITS (16 bytes per row, lane half and layer: written by the render's tape
// writer or by k_tape_bits, loaded once per tile, shifted into vcc value by value), and every dZ leaving once, row-major,
// through the wave-private LDS blocks of the render's tape writer (whole 128-byte lines) -- they are the A operands of
// the four weight-gradient GEMMs, which now read tapes nobody overwrites.
// The last layer's output rows are ORDERED so that a lane ends up with everything one embedding channel needs
// (d e, d sin / d cos of its three octaves: 7 of 8 consecutive accumulator registers): the chain rule through the
// positional encoding (k_train_rowgrad before) is 14 multiply-adds on registers, and the 7 extra head inputs
// (d colour, d dir) come out of a ninth output tile of the W2^T layer.  The kernel also counts the rows of every touched
// point (integer atomics) for the ordered point sums that follow.
// 4 224 MFMAs per 32-row tile (the render, whose first layer is factorised per point: 3 360).
#include <mutex>

#include "pnr_shade_common.h"
#include "pnr_train_chain.h"

namespace pnr {

#ifndef PNR_CPFS
#define PNR_CPFS 6
#endif
constexpr int CPFS = PNR_CPFS;
static_assert(CNG_TILE % CPFS == 0, "the window slot of a group must not depend on the tile");
constexpr int CT_ROW_B = 144, CT_BLK_B = 32 * CT_ROW_B, CT_WAVE_B = 2 * CT_BLK_B;   // as the render's tape writer

namespace {
__device__ __forceinline__ float chain_to_a(float v)
{
    asm("" : "+a"(v));
    return v;
}
__device__ __forceinline__ void c_lds_write4(unsigned addr, f32x4 v, int off)
{
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(off));
}
__device__ __forceinline__ f32x4 c_lds_read4(unsigned addr, int off)
{
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
    return v;
}
__device__ __forceinline__ f32x2 c_lds_read2(unsigned addr, int off)
{
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
    return v;
}
__device__ __forceinline__ void c_lds_wait()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// One layer of the chain.  G0: its first weight group in the tile stream; `in`: the lane's 128 B operands; accumulators
// start at zero (the first MFMA of an output tile takes the constant).  `pre(m)` runs in front of output tile m (loads
// for one tile ahead), `fill(m, i)` behind MFMA i (0..127) of tile m.
template <int MT, int G0, typename Off, typename Pre, typename Fill>
__device__ __forceinline__ void chain_layer(__amdgpu_buffer_rsrc_t rsrc, int voff, Off off, float4 (&wq)[CPFS],
                                            const float (&in)[128], f32x16 (&acc)[9], Pre pre, Fill fill)
{
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int kg = 0; kg < 32; ++kg) {
            const int G = G0 + m * 32 + kg;
            const float4 w = wq[G % CPFS];
            wq[G % CPFS] = load_w(rsrc, voff, off(G + CPFS));
            if (kg == 0) pre(m);
            if (kg == 0)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, in[0], zero, 0, 0, 0);
            else
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
}  // namespace

// LDS of a wave: the tape writer's two staging blocks | the EMBEDDINGS of the tile's rows (X0 columns 0..31: 128 bytes per
// row, rows 144 bytes apart) | the rows' point gradients (160 bytes per row, rows 176 apart).
// (Round 3 staged the taped (sin, cos) pairs of the positional encoding instead -- X0 columns 32..223, 768 bytes per row,
// one load + LDS write per row: 0.50 of the kernel's 5.3 ms at 65 536 rays.  The chain rule through the encoding needs
// sin / cos of e 2^f for the lane's two channels per output tile: one branch-free sincos per channel and two double-angle
// steps on registers (~35 instructions against 14 multiply-adds + the staging of 48 bytes) from a sixth of the bytes.)
constexpr int CX_ROW_B = 144, CX_BLK_B = 32 * CX_ROW_B;
constexpr int CG_ROW_B = 176, CG_BLK_B = 32 * CG_ROW_B;
constexpr int CHAIN_WAVE_B = CT_WAVE_B + CX_BLK_B + CG_BLK_B;   // 39 936
constexpr int CHAIN_LDS_B = WAVES * CHAIN_WAVE_B + 1024;        // + the density head in accumulator order, shared

__global__ void __launch_bounds__(TPB, 1) k_train_pairs_bwd(ChainParams P)
{
    extern __shared__ float chain_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int h = lane >> 5, j = lane & 31;
    const int n_rows = __builtin_amdgcn_readfirstlane(P.cnt[0]);
    const int ntiles = (n_rows + 127) >> 7;
    const int K = P.K;

    const unsigned tblk = (unsigned)(uintptr_t)chain_lds + (unsigned)(wave * CHAIN_WAVE_B);
    const unsigned t_wr = tblk + (unsigned)(j * CT_ROW_B + 16 * h);                      // + block + 32 q
    const unsigned t_rd = tblk + (unsigned)((lane >> 3) * CT_ROW_B + (lane & 7) * 16);   // + block + 8 i rows
    const unsigned xblk = tblk + CT_WAVE_B, gblk = xblk + CX_BLK_B;
    const unsigned w4lds = (unsigned)(uintptr_t)chain_lds + (unsigned)(WAVES * CHAIN_WAVE_B) + 64u * (unsigned)h;   // + 128 m + 16 q
    const unsigned x_rd = xblk + (unsigned)(j * CX_ROW_B + 8 * h);                       // + 16 m: the lane's two channels
    const unsigned g_wr = gblk + (unsigned)(j * CG_ROW_B);

    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.wchain), 0, CNG_TILE * 1024, 0x00020000);
    const int voff = lane * 16;
    const auto goff = [](int G) { return (G >= CNG_TILE ? G - CNG_TILE : G) * 1024; };
    float4 wq[CPFS];
    if ((int)blockIdx.x < ntiles) {
#pragma unroll
        for (int p = 0; p < CPFS; ++p) wq[p] = load_w(rsrc, voff, goff(p));
    }
    const f32x2 k01 = {0.1f, 0.1f};
    // the density head (1 KiB, [(tile * 2 + h) * 16 + r]) into LDS once: read per output tile of the prologue as LDS
    // reads, which do not queue behind the prologue's memory loads
    if (threadIdx.x < 64) {
        const float4 w = *reinterpret_cast<const float4 *>(P.w4acc + 4 * lane);
        const f32x4 wv = {w.x, w.y, w.z, w.w};
        asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(uintptr_t)chain_lds + (unsigned)(WAVES * CHAIN_WAVE_B) + 16u * (unsigned)lane),
                     "v"(wv));
    }
    __syncthreads();

    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        // ---- the wave's 32 rows: rows beyond the call's last one read the last row and store nothing ------------------
        // (uniform values, and told so: a descriptor the compiler believes to vary per lane turns every buffer store into a
        // readfirstlane loop with a branch -- in the middle of the MFMA stream)
        const int row0 = __builtin_amdgcn_readfirstlane(tile * 128 + wave * 32);
        const int nv = __builtin_amdgcn_readfirstlane(max(0, min(32, n_rows - row0)));   // rows of this wave that exist
        const int row0c = __builtin_amdgcn_readfirstlane(min(row0, n_rows - 1));
        const int rel = min(row0 + j, n_rows - 1) - row0c;
        const int row = row0c + rel;
        const bool live = j < nv;
        // LeakyReLU masks of the four layers as bits (written by the render, or by k_tape_bits after a recompute): one
        // coalesced 16-byte load per layer.  The first version read the taped activations themselves, 16-byte pieces
        // of 32 different rows per load instruction: such a load is slow to complete, vector loads return in order,
        // and every weight group behind it waited -- 33 stalls per tile, 0.51 of the peak.
        u32x4 sb[4];
#pragma unroll
        for (int l = 0; l < 4; ++l)
            sb[l] = *reinterpret_cast<const u32x4 *>(P.tape_bits + (((size_t)l * P.bits_rows + (size_t)row) * 2 + h) * 4);
        // the rows' embeddings (X0 columns 0..31) -> LDS: 8 lanes per row, 8 rows per instruction
        {
            const int r8 = lane >> 3, c4 = lane & 7;
            const unsigned xw = xblk + (unsigned)(r8 * CX_ROW_B + 16 * c4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rrc = min(row0 + r8 + 8 * i, n_rows - 1) - row0c;
                const float4 x = *reinterpret_cast<const float4 *>(P.X0 + (int64_t)(row0c + rrc) * 288 + 4 * c4);
                const f32x4 xv = {x.x, x.y, x.z, x.w};
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(xw), "v"(xv), "n"(i * 8 * CX_ROW_B));
            }
        }
        const float wk = live ? P.row_w[row] : 0.f;
        const int v = row / K;
        const float dsig = P.d_out[v].x;
        const float coef = P.row_z[row] > 0.f ? wk * dsig : 0.f;
        float vx, vy, vz;
        {
            const int64_t ray = P.smp_ray[P.vs_list[v]];
            rot_rows(P.Rw2c, P.dirs[3 * ray], P.dirs[3 * ray + 1], P.dirs[3 * ray + 2], vx, vy, vz);
        }
        {
            const int pidx = P.row_pidx[row];
            if (live && h == 0 && pidx >= 0) atomicAdd(&P.pt_cnt[P.pt_rank[pidx]], 1);
        }
        // stores go through descriptors that end behind the wave's last existing row: no predication anywhere
        __amdgpu_buffer_rsrc_t rD[4];
        rD[0] = __builtin_amdgcn_make_buffer_rsrc(P.D3 + (int64_t)row0c * 256, 0, nv * 1024, 0x00020000);
        rD[1] = __builtin_amdgcn_make_buffer_rsrc(P.D2 + (int64_t)row0c * 256, 0, nv * 1024, 0x00020000);
        rD[2] = __builtin_amdgcn_make_buffer_rsrc(P.D1 + (int64_t)row0c * 256, 0, nv * 1024, 0x00020000);
        rD[3] = __builtin_amdgcn_make_buffer_rsrc(P.D0 + (int64_t)row0c * 256, 0, nv * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t rG =
            __builtin_amdgcn_make_buffer_rsrc(P.rowgrad + (int64_t)row0c * 40, 0, nv * 160, 0x00020000);
        const float *xc = P.XC + (int64_t)v * 256 + 4 * h;

        // block (layer pl, output tile pt), written one gap earlier, goes out: 8 lanes per row, whole 128-byte lines
        auto tape_flush = [&](int pl, int pt) {
            f32x4 t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = c_lds_read4(t_rd, (pt & 1) * CT_BLK_B + i * 8 * CT_ROW_B);
            c_lds_wait();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4 o;
                o.x = __float_as_uint(t[i].x);
                o.y = __float_as_uint(t[i].y);
                o.z = __float_as_uint(t[i].z);
                o.w = __float_as_uint(t[i].w);
                const int off = (8 * i + (lane >> 3)) * 1024 + 16 * (lane & 7) + 128 * pt;
#ifndef PNR_CHAIN_EXP_NO_ST
                __builtin_amdgcn_raw_buffer_store_b128(o, rD[pl], off, 0, 0);
#else
                if (n_rows < 0) __builtin_amdgcn_raw_buffer_store_b128(o, rD[pl], off, 0, 0);
#endif
            }
        };
        auto tape_tile = [&](int layer, int tt, const float (&t)[16]) {
            if (tt > 0)
                tape_flush(layer, tt - 1);
            else if (layer > 0)
                tape_flush(layer - 1, 7);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v4 = {t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
                c_lds_write4(t_wr, v4, (tt & 1) * CT_BLK_B + 32 * q);
            }
        };
        // t = a * LeakyReLU'(taped activation): one packed multiply per pair of values, and per value the next mask bit
        // shifted out of its word into vcc and a select
        auto mask16 = [&](const f32x16 &a, unsigned &bits, float (&t)[16]) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 x = {a[r], a[r + 1]};
                f32x2 y;
                asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y) : "v"(x), "v"(k01));
                asm("v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, %2, %3, vcc"
                    : "+v"(bits), "=&v"(t[r])
                    : "v"(y.x), "v"(a[r])
                    : "vcc");
                asm("v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, %2, %3, vcc"
                    : "+v"(bits), "=&v"(t[r + 1])
                    : "v"(y.y), "v"(a[r + 1])
                    : "vcc");
            }
        };
        unsigned sbA[4] = {sb[3].x, sb[3].y, sb[3].z, sb[3].w};   // G2
        unsigned sbB[4] = {sb[2].x, sb[2].y, sb[2].z, sb[2].w};   // G1
        unsigned sbC[4] = {sb[1].x, sb[1].y, sb[1].z, sb[1].w};   // H2
        unsigned sbD[4] = {sb[0].x, sb[0].y, sb[0].z, sb[0].w};   // H1

        float X[128], Y[128];
        f32x16 acc[9];
        // ---- dZ3 = w (dAGG + d sigma [z > 0] w4) * L'(G2): the lane's 128 features of its row --------------------------
        // the row's 128 values of dAGG in ONE burst of loads (the staging asm of the loop below is a compiler barrier for
        // memory operations: loads left inside it were issued and waited for one at a time, ~16 us per tile)
        float4 dagg[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
#ifndef PNR_CHAIN_EXP_NO_XC
            dagg[i] = *reinterpret_cast<const float4 *>(xc + 32 * (i >> 2) + 8 * (i & 3));
#else
            dagg[i] = make_float4(dsig, wk, coef, vx);   // (timing experiment)
#endif
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            f32x4 w4v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w4v[q] = c_lds_read4(w4lds + 128u * (unsigned)m, 16 * q);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w4v[0]), "+v"(w4v[1]), "+v"(w4v[2]), "+v"(w4v[3])::"memory");
            f32x16 pre;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 da = dagg[4 * m + q];
                const f32x4 wv = w4v[q];
                pre[4 * q + 0] = wk * da.x + coef * wv.x;
                pre[4 * q + 1] = wk * da.y + coef * wv.y;
                pre[4 * q + 2] = wk * da.z + coef * wv.z;
                pre[4 * q + 3] = wk * da.w + coef * wv.w;
            }
            float t16[16];
            mask16(pre, sbA[m >> 1], t16);
#pragma unroll
            for (int r = 0; r < 16; ++r) X[16 * m + r] = chain_to_a(t16[r]);
            tape_tile(0, m, t16);
        }
        // ---- W3^T: dZ3 -> dZ2 (masks: G1) ---------------------------------------------------------------------------------
        chain_layer<8, 0>(
            rsrc, voff, goff, wq, X, acc, [&](int) {},
            [&](int m, int i) {
                if (i == 0 && m > 0) {
                    float t16[16];
                    mask16(acc[m - 1], sbB[(m - 1) >> 1], t16);
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[16 * (m - 1) + r] = t16[r];
                    tape_tile(1, m - 1, t16);
                }
            });
        // ---- W2^T: dZ2 -> [d extras | dZ1] (output tile 0: the seven extra head inputs; tiles 1..8 masked by H2) -----------
        float ex[4];
        {
            const f32x16 last = acc[7];
            chain_layer<9, CNG_B>(
                rsrc, voff, goff, wq, Y, acc, [&](int) {},
                [&](int m, int i) {
                    if (i != 0) return;
                    float t16[16];
                    if (m == 0) {
                        mask16(last, sbB[3], t16);
#pragma unroll
                        for (int r = 0; r < 16; ++r) Y[112 + r] = t16[r];
                        tape_tile(1, 7, t16);
                    } else if (m == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) ex[r] = acc[0][r];
                    } else {
                        mask16(acc[m - 1], sbC[(m - 2) >> 1], t16);
#pragma unroll
                        for (int r = 0; r < 16; ++r) X[16 * (m - 2) + r] = chain_to_a(t16[r]);
                        tape_tile(2, m - 2, t16);
                    }
                });
        }
        // ---- W1^T: dZ1 -> dZ0 (masks: H1) ---------------------------------------------------------------------------------
        {
            const f32x16 last = acc[8];
            chain_layer<8, CNG_B + CNG_C>(
                rsrc, voff, goff, wq, X, acc, [&](int) {},
                [&](int m, int i) {
                    if (i != 0) return;
                    float t16[16];
                    if (m == 0) {
                        mask16(last, sbC[3], t16);
#pragma unroll
                        for (int r = 0; r < 16; ++r) X[112 + r] = chain_to_a(t16[r]);
                        tape_tile(2, 7, t16);
                    } else {
                        mask16(acc[m - 1], sbD[(m - 1) >> 1], t16);
#pragma unroll
                        for (int r = 0; r < 16; ++r) Y[16 * (m - 1) + r] = t16[r];
                        tape_tile(3, m - 1, t16);
                    }
                });
        }
        // ---- W0^T: dZ0 -> d embedding.  Output tile m, lane half h: channels 4 m + 2 h and + 1, eight registers each
        //      [d e | d sin, d cos of octave 0 | 1 | 2 | -]; the taped (sin, cos) pairs come from the LDS block ------------
        f32x2 xe;     // the embedding values of the lane's two channels of the output tile in flight
        auto emb_store = [&](int tt, const f32x16 &a) {
            float g[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // sin / cos of e 2^f, f = 0..2: the base octave by the forward's branch-free sincos, the others by the
                // double-angle identities (1e-7: far below what a gradient is held to)
                float sn[3], cs[3];
                fast_sincos_nb(xe[e], sn[0], cs[0]);
#pragma unroll
                for (int f = 1; f < 3; ++f) {
                    sn[f] = 2.0f * sn[f - 1] * cs[f - 1];
                    cs[f] = fmaf(-2.0f * sn[f - 1], sn[f - 1], 1.0f);
                }
                // d/de [e, sin(e 2^f), cos(e 2^f)] = [1, 2^f cos, -2^f sin]
                float gg = a[8 * e];
#pragma unroll
                for (int f = 0; f < 3; ++f)
                    gg += (float)(1 << f) * (cs[f] * a[8 * e + 1 + 2 * f] - sn[f] * a[8 * e + 2 + 2 * f]);
                g[e] = gg;
            }
            const f32x2 g2 = {g[0], g[1]};
            asm volatile("ds_write_b64 %0, %1" ::"v"(g_wr + (unsigned)(16 * tt + 8 * h)), "v"(g2));
        };
        // (the wait names the registers the read fills: arithmetic on them must not be scheduled in front of it)
        auto xs_wait = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xe)::"memory");
        };
        auto xs_issue = [&](int tt) {
            xe = c_lds_read2(x_rd + (unsigned)(16 * tt), 0);
        };
        {
            const f32x16 last = acc[7];
            chain_layer<8, CNG_B + CNG_C + CNG_D>(
                rsrc, voff, goff, wq, Y, acc, [&](int) {},
                [&](int m, int i) {
                    if (i == 0) {
                        if (m == 0) {
                            float t16[16];
                            mask16(last, sbD[3], t16);
#pragma unroll
                            for (int r = 0; r < 16; ++r) Y[112 + r] = t16[r];
                            tape_tile(3, 7, t16);
                        } else {
                            if (m == 1) tape_flush(3, 7);
                            xs_wait();
                            emb_store(m - 1, acc[m - 1]);
                        }
                    }
                    if (i == 120) xs_issue(m);   // the embedding values of output tile m, a few MFMAs before they are used
                });
        }
        xs_wait();
        emb_store(7, acc[7]);
        // ---- d colour (lane half 0: extras 0..2) and d dir (lane half 1: extras 4..6, extra 3 from its partner) ---------
        {
            const float p3 = __shfl_xor(ex[3], 32, 64);
            // sdir = dir @ Rw2c^T; the head sees sdir - view and <sdir, view>
            const float gd = ex[2];
            const float gs0 = p3 + gd * vx, gs1 = ex[0] + gd * vy, gs2 = ex[1] + gd * vz;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float gdir = gs0 * P.Rw2c[c] + gs1 * P.Rw2c[3 + c] + gs2 * P.Rw2c[6 + c];
                const float o = h ? gdir : ex[c];
                asm volatile("ds_write_b32 %0, %1" ::"v"(g_wr + (unsigned)(4 * (32 + 3 * h + c))), "v"(o));
            }
        }
        // the rows' point gradients leave: 32 rows x 160 bytes are contiguous in memory (5 x 64 float4)
        c_lds_wait();
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int e = i * 64 + lane, rr = e / 10, c = e - rr * 10;
            const f32x4 g4 = c_lds_read4(gblk + (unsigned)(rr * CG_ROW_B + c * 16), 0);
            c_lds_wait();
            u32x4 o;
            o.x = __float_as_uint(g4.x);
            o.y = __float_as_uint(g4.y);
            o.z = __float_as_uint(g4.z);
            o.w = __float_as_uint(g4.w);
            __builtin_amdgcn_raw_buffer_store_b128(o, rG, e * 16, 0, 0);
        }
    }
}

// LeakyReLU masks as bits from ROW-MAJOR tapes (after a recompute: the render did not write the tape): one thread per
// (layer, row, lane half), the same words the render's tape writer produces
__global__ void __launch_bounds__(256) k_tape_bits(const int *__restrict__ cnt, const float *__restrict__ H1,
                                                   const float *__restrict__ H2, const float *__restrict__ G1,
                                                   const float *__restrict__ G2, size_t bits_rows,
                                                   unsigned *__restrict__ bits)
{
    const int rows = cnt[0];
    const int64_t total = (int64_t)rows * 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int hh = (int)(i & 1), l = (int)((i >> 1) & 3);
        const int64_t row = i >> 3;
        const float *src = l == 0 ? H1 + row * 256 : (l == 1 ? H2 + row * 264 : (l == 2 ? G1 + row * 256 : G2 + row * 256));
        unsigned w[4];
        for (int wd = 0; wd < 4; ++wd) {
            unsigned b = 0u;
            for (int t = 32 * wd; t < 32 * wd + 32; ++t) {
                const int m = t >> 4, r = t & 15;
                b = (b << 1) | (src[32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh] > 0.f ? 1u : 0u);
            }
            w[wd] = b;
        }
        u32x4 o;
        o.x = w[0];
        o.y = w[1];
        o.z = w[2];
        o.w = w[3];
        *reinterpret_cast<u32x4 *>(bits + (((size_t)l * bits_rows + (size_t)row) * 2 + hh) * 4) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// the chain's weight stream: dst[(G * 64 + lane) * 4 + q] = W[c][o], W the PyTorch [256 out, n_in] matrix of the layer,
// c = the forward OUTPUT feature that k-step 4 g + q carries in lane half lane >> 5 (accumulator order of the layer above),
// o = the forward INPUT feature that row lane & 31 of output tile m stands for
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pack_chain(const float *__restrict__ w0, const float *__restrict__ w1,
                                                    const float *__restrict__ w2, const float *__restrict__ w3,
                                                    const float *__restrict__ w4, float *__restrict__ dst)
{
    const int total = CNG_TILE * 256;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total + 256; idx += gridDim.x * blockDim.x) {
        if (idx >= total) {   // density head in accumulator order
            const int i = idx - total, t = i >> 5, hh = (i >> 4) & 1, r = i & 15;
            dst[idx] = w4[32 * t + 8 * (r >> 2) + 4 * hh + (r & 3)];
            continue;
        }
        const int q = idx & 3, lane = (idx >> 2) & 63, G = idx >> 8;
        const int i = lane & 31, hh = lane >> 5;
        const int layer = G < CNG_B ? 0 : (G < CNG_B + CNG_C ? 1 : (G < CNG_B + CNG_C + CNG_D ? 2 : 3));
        const int Gl = G - (layer == 0 ? 0 : (layer == 1 ? CNG_B : (layer == 2 ? CNG_B + CNG_C : CNG_B + CNG_C + CNG_D)));
        const int m = Gl >> 5, g = Gl & 31;
        const int t = 4 * g + q, mt = t >> 4, r = t & 15;
        const int c = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float *W;
        int n_in, o;
        if (layer == 0) {
            W = w3, n_in = 256, o = 32 * m + i;
        } else if (layer == 1) {
            W = w2, n_in = 263, o = m == 0 ? 256 + i : 32 * (m - 1) + i;
        } else if (layer == 2) {
            W = w1, n_in = 256, o = 32 * m + i;
        } else {
            W = w0, n_in = 284;
            const int ra = (i & 3) + 4 * (i >> 3), ha = (i >> 2) & 1;   // accumulator register / lane half of row i
            const int d = 4 * m + 2 * ha + (ra >> 3), comp = ra & 7;
            o = comp == 0 ? d : (comp == 7 ? -1 : 32 + 2 * (3 * d + ((comp - 1) >> 1)) + ((comp - 1) & 1));
        }
        dst[idx] = (o >= 0 && o < n_in) ? W[(int64_t)c * n_in + o] : 0.f;
    }
}

void launch_pack_chain(const float *w0, const float *w1, const float *w2, const float *w3, const float *w4, float *dst,
                       hipStream_t st)
{
    hipLaunchKernelGGL(k_pack_chain, dim3(264), dim3(256), 0, st, w0, w1, w2, w3, w4, dst);
}

int launch_pairs_bwd(const ChainParams &P, int64_t rows_max, hipStream_t st)
{
    // per DEVICE: the CU count, and the kernel's dynamic-LDS limit raised once (a process may drive several devices)
    int cus = 256;
    const int rca = ensure_dynamic_lds(reinterpret_cast<const void *>(k_train_pairs_bwd), CHAIN_LDS_B, &cus);
    if (rca != PNR_OK) return rca;
    const int64_t tiles = (rows_max + 127) / 128;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cus, tiles));
    hipLaunchKernelGGL(k_train_pairs_bwd, dim3(grid), dim3(TPB), (size_t)CHAIN_LDS_B, st, P);
    return PNR_OK;
}

void launch_tape_bits(const int *cnt, const float *H1, const float *H2, const float *G1, const float *G2, size_t bits_rows,
                      unsigned *bits, hipStream_t st)
{
    hipLaunchKernelGGL(k_tape_bits, dim3(2048), dim3(256), 0, st, cnt, H1, H2, G1, G2, bits_rows, bits);
}

}  // namespace pnr
