// Training step: gradients of a render with respect to the trainable point tensors and the MLP weights
// (SURVEY.md section 8f rank 1).  Replaces what torch autograd derives for studio_model.py:263-399 when
// `ns-train pointnerf-original` back-propagates the loss of studio_model.py:415-431 through get_outputs:
// index_select backward (dense [N,32] scatter), nine F.linear backward pairs, the boolean-compaction scatters,
// the K-aggregation and the cumprod composite.
//
// First HIP version of this row: correct and matrix-core bound, not yet fused.  The query products of the
// preceding pnr_render call (sample lists, neighbour indices) are taken from its workspace; the MLP forward is
// recomputed (fp32 by default, or bf16x3 after a render in that opt-in mode) with every activation kept ROW-MAJOR in HBM (the "tape"),
// then walked backwards:
//   * every Linear is one of three shapes of ONE hand-written fp32 MFMA GEMM (k_gemm: 128x128 tiles,
//     v_mfma_f32_32x32x2_f32, LDS double buffer): forward X.W^T (+bias, LeakyReLU), data gradient dZ.W (times
//     LeakyReLU' from the taped activation, written IN PLACE over that activation), weight gradient dZ^T.X
//     (split over the rows into partial tiles that a reducer sums: no float atomics);
//   * small row-parallel kernels for what is not a GEMM: inputs / encodings, density head + K-aggregation,
//     colour head, composite (reverse scan per ray), and the point gradients (rows grouped by point, one ordered sum
//     per point).  No float atomic anywhere: every gradient is bitwise repeatable.
// Frozen tensors (xyz, Rw2c) and the sample positions get no gradient, as in the reference
// (studio_utils.py:84-103: only embedding / conf / dir / color are Parameters with requires_grad; conf does not
// enter the render, studio_model.py:285-292).
#include <algorithm>
#include <type_traits>

#include "pnr_shade_common.h"
#include "pnr_train_chain.h"

namespace pnr {

// ------------------------------------------------------------------------------------------------
// fp32 MFMA GEMM   C[M,N] = op(A) . op(B)      (all matrices row-major, leading dimensions multiples of 4)
//   TA = false: A is [M, K] (tile rows m, contiguous k)     TA = true: A is [K, M] (A'[m][k] = A[k][m])
//   TB = false: B is [K, N]                                 TB = true: B is [N, K] (B'[k][n] = B[n][k])
// The row count that comes from the device (pairs / samples of the call) is read from *dev_rows: it is M when
// TA = false and the reduction length K when TA = true (weight gradients, split over gridDim.z).
// ------------------------------------------------------------------------------------------------
enum { EPI_STORE = 0, EPI_BIAS_LEAKY = 1, EPI_MASK = 2, EPI_PARTIAL = 3 };

struct GemmArgs {
    const float *A;
    const float *B;
    float *C;
    int lda, ldb, ldc;
    int M, N, K;            // static extents (M or K replaced by *dev_rows, see above)
    const int *dev_rows;
    const float *bias;      // EPI_BIAS_LEAKY
    const float *mask;      // EPI_MASK: taped post-activation, same leading dimension as C
    int mask_cols;          // columns >= mask_cols pass unmasked
    // LeakyReLU masks as SIGN BITS: the forward epilogue ballots (activation > 0) per stored float4 component, the masked
    // epilogue reads the four 64-bit words of its step back (2 KiB per 128x128 tile instead of the 64-KiB tile of floats)
    unsigned long long *sign_out;        // EPI_BIAS_LEAKY, may be null
    const unsigned long long *sign_in;   // EPI_MASK, null = read the float mask
    int sign_nt;                         // column tiles per row tile in the sign buffer
    float *colsum;          // TA only, may be null: colsum[z][m] = sum over split z's rows of A[k][m]  (bias gradient)
};

// accumulator register r of lane (j, h): row (r & 3) + 8 (r >> 2) + 4 h, column j of the 32x32 block
// Epilogue of a 128x128 tile.  In accumulator order (register r of lane (j, h): row (r & 3) + 8 (r >> 2) + 4 h, column j
// of a 32x32 block) a store instruction writes 4 bytes per lane into two rows; 64 of them per lane (plus 64 loads in
// the masked shape) made the epilogue 40-60 % of a GEMM.  Each wave therefore passes its 64x64 sub-tile through a
// private 8-KiB LDS region, 32 rows at a time: written in accumulator order (conflict-free: the lanes of a half wave
// hold consecutive columns), read back as float4 along the rows -- 16 lanes cover 256 contiguous bytes of a row, one
// instruction 4 rows, 16 store (and 16 load) instructions of 16 B per lane for the whole sub-tile.
// `lds_wave`: 32 x 64 floats private to the wave; every wave of the workgroup must have finished with the operand
// buffers before the call, and the caller synchronises before reusing them.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, const f32x16 (&acc)[2][2], int M, int N, int m0, int n0,
                                              int wm, int wn, int lane, float *lds_wave)
{
    const int j = lane & 31, h = lane >> 5;
    const int rr = lane >> 4, c4 = (lane & 15) * 4;   // read phase: row rr + 4 i, columns c4 .. c4 + 3
    const int col = n0 + wn * 64 + c4;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == EPI_BIAS_LEAKY && col < N) bias = *reinterpret_cast<const float4 *>(g.bias + col);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                lds_wave[((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + b * 32 + j] = acc[a][b][r];
        __builtin_amdgcn_wave_barrier();
        const int row0 = m0 + wm * 64 + a * 32 + rr;
        const bool use_bits = EPI == EPI_MASK && g.sign_in != nullptr;
        const bool tile_has_bits = n0 / 128 < g.sign_nt;   // 128 = the tile size (TM = TN, defined below)
        // word index of step (a, i), component c: ((((row tile * sign_nt + column tile) * 4 + wave) * 2 + a) * 8 + i) * 4 + c
        const int64_t wbase = ((((int64_t)(m0 / 128) * g.sign_nt + n0 / 128) * 4 + (wm * 2 + wn)) * 2 + a) * 32;
        // float masks (the taped activation this GEMM overwrites, mask == C): all loads before the first store
        float4 mk[8];
        if (EPI == EPI_MASK && !use_bits) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = row0 + 4 * i;
                mk[i] = (row < M && col < g.mask_cols) ? *reinterpret_cast<const float4 *>(g.mask + (int64_t)row * g.ldc + col)
                                                       : make_float4(1.f, 1.f, 1.f, 1.f);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + 4 * i;
            float4 v = *reinterpret_cast<const float4 *>(lds_wave + (rr + 4 * i) * 64 + c4);
            const bool ok = row < M && col < N;
            if (EPI == EPI_BIAS_LEAKY) {
                v.x += bias.x;
                v.y += bias.y;
                v.z += bias.z;
                v.w += bias.w;
                v.x = v.x > 0.f ? v.x : 0.1f * v.x;
                v.y = v.y > 0.f ? v.y : 0.1f * v.y;
                v.z = v.z > 0.f ? v.z : 0.1f * v.z;
                v.w = v.w > 0.f ? v.w : 0.1f * v.w;
                if (g.sign_out && tile_has_bits) {   // wave-uniform
                    const unsigned long long b0 = __ballot(ok && v.x > 0.f), b1 = __ballot(ok && v.y > 0.f);
                    const unsigned long long b2 = __ballot(ok && v.z > 0.f), b3 = __ballot(ok && v.w > 0.f);
                    if (lane < 4) g.sign_out[wbase + 4 * i + lane] = lane == 0 ? b0 : (lane == 1 ? b1 : (lane == 2 ? b2 : b3));
                }
            } else if (EPI == EPI_MASK) {
                if (use_bits) {
                    if (tile_has_bits) {
                        const unsigned long long *wp = g.sign_in + wbase + 4 * i;
                        v.x *= ((wp[0] >> lane) & 1ull) ? 1.0f : 0.1f;
                        v.y *= ((wp[1] >> lane) & 1ull) ? 1.0f : 0.1f;
                        v.z *= ((wp[2] >> lane) & 1ull) ? 1.0f : 0.1f;
                        v.w *= ((wp[3] >> lane) & 1ull) ? 1.0f : 0.1f;
                    }
                } else {
                    v.x *= mk[i].x > 0.f ? 1.0f : 0.1f;
                    v.y *= mk[i].y > 0.f ? 1.0f : 0.1f;
                    v.z *= mk[i].z > 0.f ? 1.0f : 0.1f;
                    v.w *= mk[i].w > 0.f ? 1.0f : 0.1f;
                }
            }
            if (ok) *reinterpret_cast<float4 *>(g.C + (int64_t)row * g.ldc + col) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

#ifndef PNR_GEMM_WGS
#define PNR_GEMM_WGS 2
#endif
#ifndef PNR_GEMM_TK
#define PNR_GEMM_TK 16
#endif
constexpr int TM = 128, TN = 128, TK = PNR_GEMM_TK, LDT = TM + 4;
constexpr int NLD = TK / 8;   // float4 loads per thread and operand per chunk

// weight gradients: the grid offers gridDim.z splits of the rows (sized for the workspace capacity); a split is worth
// its fixed costs (prologue, a 64-KB partial tile) only with >= 256 rows, so the kernel and the reducer agree on
// this number from the row count found on the device
#ifndef PNR_MIN_ROWS_PER_SPLIT
#define PNR_MIN_ROWS_PER_SPLIT 256
#endif
constexpr int MIN_ROWS_PER_SPLIT = PNR_MIN_ROWS_PER_SPLIT;
__device__ __forceinline__ int active_splits(int rows, int nz_grid)
{
    return min(nz_grid, max(1, (rows + MIN_ROWS_PER_SPLIT - 1) / MIN_ROWS_PER_SPLIT));
}

// the body of k_gemm for workgroup (bx, by, bz) of a (gx, ., gz) grid: also run by the batched weight-gradient launch
template <bool TA, bool TB, int EPI>
__device__ __forceinline__ void gemm_body(const GemmArgs &g, int bx, int by, int bz, int gx, int gz,
                                          float (&As)[2][TK][LDT], float (&Bs)[2][TK][LDT])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int rows = *g.dev_rows;
    const int M = TA ? g.M : rows;
    const int N = g.N;
    // row GEMMs (TA = false): the column tile is the FAST grid dimension, so the workgroups that share a row tile of A
    // run together and its second read comes from L2 (the tapes are 1 GB each: launched row-tile-major per column
    // tile, A was fetched from HBM once per column tile)
    // Row GEMMs are PERSISTENT: a fixed grid walks the (row tile, column tile) pairs of the row count found on the
    // device -- a grid sized for the capacity of the workspace would launch ~10x more workgroups than have work
    // (an empty workgroup still costs ~2 us of a CU slot: 0.4 ms per GEMM at 114 k of them).
    const int n_nt = (N + TN - 1) / TN;
    const int total = TA ? 1 : n_nt * ((M + TM - 1) / TM);
    for (int it = TA ? 0 : bx; it < total; it += TA ? 1 : gx) {
    const int m0 = (TA ? bx : it / n_nt) * TM, n0 = (TA ? by : it % n_nt) * TN;
    if (m0 >= M || n0 >= N) return;
    int k_begin = 0, k_end = TA ? rows : g.K;
    if (TA) {
        // split of the reduction over gz, in multiples of TK
        const int nz = active_splits(rows, gz);
        if (bz >= nz) return;
        const int chunk = ((rows + nz - 1) / nz + TK - 1) / TK * TK;
        k_begin = min(rows, bz * chunk);
        k_end = min(rows, k_begin + chunk);
        if (k_begin >= k_end) return;
    }
    const int wm = wave & 1, wn = wave >> 1;

    // Two register stages (chunk c + 2 is requested before the MFMAs of chunk c: one 16-deep chunk is ~2 k cycles of
    // fp32 MFMA, a memory round trip ~4.5 k), loads branch-free (indices clamped into the matrix, zeros selected when
    // the chunk is written to LDS -- a predicated load is its own basic block and hipcc then waits vmcnt(0) at the
    // merges), barriers that wait for LDS only.  Two chunks per loop iteration; chunks beyond k_end are all-zero.
    float4 ra[2][NLD], rb[2][NLD];
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool want_csum = TA && g.colsum && by == 0;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_tiles = [&](int k0, auto par) {
        constexpr int P = decltype(par)::value;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (TA) {  // natural: TK rows (k) x 128 contiguous m
                const int kr = (tid >> 5) + 8 * i, c4 = (tid & 31) * 4;
                const int k = min(k0 + kr, k_end - 1), m = min(m0 + c4, M - 4);
                ra[P][i] = *reinterpret_cast<const float4 *>(g.A + (int64_t)k * g.lda + m);
            } else {  // transposing: 128 rows (m) x TK contiguous k
                const int mr = (tid / (TK / 4)) + (1024 / TK) * i, kq = (tid % (TK / 4)) * 4;
                const int m = min(m0 + mr, M - 1), k = min(k0 + kq, k_end - 4);
                ra[P][i] = *reinterpret_cast<const float4 *>(g.A + (int64_t)m * g.lda + k);
            }
            if (!TB) {
                const int kr = (tid >> 5) + 8 * i, c4 = (tid & 31) * 4;
                const int k = min(k0 + kr, k_end - 1), n = min(n0 + c4, N - 4);
                rb[P][i] = *reinterpret_cast<const float4 *>(g.B + (int64_t)k * g.ldb + n);
            } else {
                const int nr = (tid / (TK / 4)) + (1024 / TK) * i, kq = (tid % (TK / 4)) * 4;
                const int n = min(n0 + nr, N - 1), k = min(k0 + kq, k_end - 4);
                rb[P][i] = *reinterpret_cast<const float4 *>(g.B + (int64_t)n * g.ldb + k);
            }
        }
    };
    // chunk starting at row / column k0 of the reduction, from register stage `par`, into LDS buffer `buf`
    auto store_tiles = [&](int buf, int k0, auto par) {
        constexpr int P = decltype(par)::value;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (TA) {
                const int kr = (tid >> 5) + 8 * i, c4 = (tid & 31) * 4;
                const float4 v = (k0 + kr < k_end && m0 + c4 < M) ? ra[P][i] : z4;
                *reinterpret_cast<float4 *>(&As[buf][kr][c4]) = v;
                if (want_csum) {
                    csum.x += v.x;
                    csum.y += v.y;
                    csum.z += v.z;
                    csum.w += v.w;
                }
            } else {
                const int mr = (tid / (TK / 4)) + (1024 / TK) * i, kq = (tid % (TK / 4)) * 4;
                const float4 v = (m0 + mr < M && k0 + kq < k_end) ? ra[P][i] : z4;
                As[buf][kq + 0][mr] = v.x;
                As[buf][kq + 1][mr] = v.y;
                As[buf][kq + 2][mr] = v.z;
                As[buf][kq + 3][mr] = v.w;
            }
            if (!TB) {
                const int kr = (tid >> 5) + 8 * i, c4 = (tid & 31) * 4;
                const float4 v = (k0 + kr < k_end && n0 + c4 < N) ? rb[P][i] : z4;
                *reinterpret_cast<float4 *>(&Bs[buf][kr][c4]) = v;
            } else {
                const int nr = (tid / (TK / 4)) + (1024 / TK) * i, kq = (tid % (TK / 4)) * 4;
                const float4 v = (n0 + nr < N && k0 + kq < k_end) ? rb[P][i] : z4;
                Bs[buf][kq + 0][nr] = v.x;
                Bs[buf][kq + 1][nr] = v.y;
                Bs[buf][kq + 2][nr] = v.z;
                Bs[buf][kq + 3][nr] = v.w;
            }
        }
    };
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    auto mfma_chunk = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            const float a0 = As[buf][kk + h][wm * 64 + j], a1 = As[buf][kk + h][wm * 64 + 32 + j];
            const float b0 = Bs[buf][kk + h][wn * 64 + j], b1 = Bs[buf][kk + h][wn * 64 + 32 + j];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, 1> P1;
    const int nchunks = (k_end - k_begin + TK - 1) / TK;
    load_tiles(k_begin, P0());
    load_tiles(k_begin + TK, P1());
    store_tiles(0, k_begin, P0());
    lds_barrier();
    for (int c = 0; c < nchunks; c += 2) {
        const int k0 = k_begin + c * TK;
        load_tiles(k0 + 2 * TK, P0());
        mfma_chunk(0);
        store_tiles(1, k0 + TK, P1());
        lds_barrier();
        load_tiles(k0 + 3 * TK, P1());
        mfma_chunk(1);
        store_tiles(0, k0 + 2 * TK, P0());
        lds_barrier();
    }

    if (want_csum) {
        // bias gradient: every thread summed the A elements it loaded (columns m0 + 4 (tid & 31) .. + 3, eight threads
        // per column group): the eight partial rows go to LDS and are added in a FIXED order (LDS float atomics would
        // add them in whatever order the waves arrive: the gradient must be bitwise repeatable), one partial row per
        // split (global atomics here meant 6 k adds per address)
        float *cs = &As[0][0][0];
        __syncthreads();
        *reinterpret_cast<float4 *>(&cs[(tid >> 5) * TM + (tid & 31) * 4]) = csum;
        __syncthreads();
        if (tid < TM && m0 + tid < M) {
            float t = cs[tid];
#pragma unroll
            for (int q = 1; q < 8; ++q) t += cs[q * TM + tid];
            g.colsum[(int64_t)bz * g.M + m0 + tid] = t;
        }
        __syncthreads();   // cs lies in the epilogue region of wave 0
    }
    // (the loop's last barrier is behind every read of the operand buffers: they are free for the epilogue)
    float *lds_wave = (wave < 2 ? &As[0][0][0] : &Bs[0][0][0]) + (wave & 1) * 2048;
    if (EPI == EPI_PARTIAL) {
        // weight gradients: split z writes its partial [M, ldc] block; k_reduce_parts sums the blocks (12.6 M float
        // atomics per GEMM -- 768 workgroups x 16 K -- cost more than the GEMM itself at training-batch sizes)
        GemmArgs gp = g;
        gp.C = g.C + (int64_t)bz * g.M * g.ldc;
        gemm_epilogue<EPI>(gp, acc, M, N, m0, n0, wm, wn, lane, lds_wave);
    } else {
        gemm_epilogue<EPI>(g, acc, M, N, m0, n0, wm, wn, lane, lds_wave);
    }
    __syncthreads();   // before the next tile's operands overwrite the epilogue regions
    }
}


template <bool TA, bool TB, int EPI>
__global__ void __launch_bounds__(256, PNR_GEMM_WGS) k_gemm(GemmArgs g)
{
    __shared__ float As[2][TK][LDT];
    static_assert(2 * TK * LDT >= 4096, "two 8-KiB epilogue regions per operand buffer");
    __shared__ float Bs[2][TK][LDT];
    gemm_body<TA, TB, EPI>(g, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.z, As, Bs);
}

// Weight gradients of the 256-wide layers (the four pair-level Linears): C[256, N] (partial of split z) = sum over the
// split's rows k of dZ[k][m] X[k][n].  k_gemm<true> tiles N in 128 columns: N = 264 and 288 (mlp_head.0 with its seven
// extra inputs, mlp_base.0 with the 60 encoded distances) then pay a third column tile for 8 / 32 columns -- 1.8 ms
// against 1.2 for N = 256 at 65 536 rays.  Here a workgroup owns all 256 rows of dW and NB x 32 columns (3 x 32 = 96:
// 288 = 3 x 96 exactly, 264 in three tiles with 9 % padding instead of 45 %; 4 x 32 for N = 256), a wave 64 rows:
// 2 x NB accumulators, 2 + NB LDS reads per 2 NB MFMAs.  Same pipeline as k_gemm: two register stages, two LDS
// buffers, branch-free loads, barriers that wait for LDS only; partial tiles in k_reduce_parts' layout, stored straight
// from the accumulators (a row of 32 columns is one 128-byte segment; once per ~350 chunks).
template <int NB>
__device__ __forceinline__ void wgrad256_body(const GemmArgs &g, int by, int bz, int gz, float (&As)[2][TK][256 + 4],
                                              float (&Bs)[2][TK][32 * NB + 4])
{
    constexpr int WN = 32 * NB;
    constexpr int NLB = (TK * WN / 4 + 255) / 256;   // float4 loads per thread for the B tile (the last one partial)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int rows = *g.dev_rows, N = g.N;
    const int n0 = by * WN;
    const int nz = active_splits(rows, gz);
    if (bz >= nz) return;
    const int chunk = ((rows + nz - 1) / nz + TK - 1) / TK * TK;
    const int k_begin = min(rows, bz * chunk), k_end = min(rows, k_begin + chunk);
    if (k_begin >= k_end) return;
    float4 ra[2][TK / 4], rb[2][NLB];
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool want_csum = g.colsum && by == 0;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int ar = tid >> 6, ac = (tid & 63) * 4;                    // A: rows ar + 4 i, columns ac .. ac + 3
    auto load_tiles = [&](int k0, auto par) {
        constexpr int P = decltype(par)::value;
#pragma unroll
        for (int i = 0; i < TK / 4; ++i) {
            const int k = min(k0 + ar + 4 * i, k_end - 1);
            ra[P][i] = *reinterpret_cast<const float4 *>(g.A + (int64_t)k * g.lda + ac);
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const int idx = min(tid + 256 * i, TK * WN / 4 - 1);
            const int br = idx / (WN / 4), bc = (idx - br * (WN / 4)) * 4;
            const int k = min(k0 + br, k_end - 1), n = min(n0 + bc, N - 4);
            rb[P][i] = *reinterpret_cast<const float4 *>(g.B + (int64_t)k * g.ldb + n);
        }
    };
    auto store_tiles = [&](int buf, int k0, auto par) {
        constexpr int P = decltype(par)::value;
#pragma unroll
        for (int i = 0; i < TK / 4; ++i) {
            const float4 v = (k0 + ar + 4 * i < k_end) ? ra[P][i] : z4;
            *reinterpret_cast<float4 *>(&As[buf][ar + 4 * i][ac]) = v;
            if (want_csum) {
                csum.x += v.x;
                csum.y += v.y;
                csum.z += v.z;
                csum.w += v.w;
            }
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const int idx = tid + 256 * i;
            if (idx < TK * WN / 4) {
                const int br = idx / (WN / 4), bc = (idx - br * (WN / 4)) * 4;
                const float4 v = (k0 + br < k_end && n0 + bc < N) ? rb[P][i] : z4;
                *reinterpret_cast<float4 *>(&Bs[buf][br][bc]) = v;
            }
        }
    };
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    f32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    auto mfma_chunk = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            const float a0 = As[buf][kk + h][wave * 64 + j], a1 = As[buf][kk + h][wave * 64 + 32 + j];
            float bv[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) bv[b] = Bs[buf][kk + h][32 * b + j];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[b], acc[0][b], 0, 0, 0);
                acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[b], acc[1][b], 0, 0, 0);
            }
        }
    };
    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, 1> P1;
    const int nchunks = (k_end - k_begin + TK - 1) / TK;
    load_tiles(k_begin, P0());
    load_tiles(k_begin + TK, P1());
    store_tiles(0, k_begin, P0());
    lds_barrier();
    for (int c = 0; c < nchunks; c += 2) {
        const int k0 = k_begin + c * TK;
        load_tiles(k0 + 2 * TK, P0());
        mfma_chunk(0);
        store_tiles(1, k0 + TK, P1());
        lds_barrier();
        load_tiles(k0 + 3 * TK, P1());
        mfma_chunk(1);
        store_tiles(0, k0 + 2 * TK, P0());
        lds_barrier();
    }
    if (want_csum) {
        // bias gradient: the four row groups' partial rows through LDS, added in a fixed order (bitwise repeatable)
        float *cs = &As[0][0][0];
        __syncthreads();
        *reinterpret_cast<float4 *>(&cs[ar * 256 + ac]) = csum;
        __syncthreads();
        g.colsum[(int64_t)bz * 256 + tid] = ((cs[tid] + cs[256 + tid]) + cs[512 + tid]) + cs[768 + tid];
    }
    // partial tile of split z: [256, ldc] block, columns n0 .. n0 + WN
    float *cp = g.C + (int64_t)bz * 256 * g.ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int col = n0 + 32 * b + j;
            if (col < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                    cp[(int64_t)row * g.ldc + col] = acc[a][b][r];
                }
            }
        }
}


template <int NB>
__global__ void __launch_bounds__(256, 2) k_wgrad256(GemmArgs g)
{
    __shared__ float As[2][TK][256 + 4];
    __shared__ float Bs[2][TK][32 * NB + 4];
    wgrad256_body<NB>(g, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.z, As, Bs);
}

// Every weight gradient of a training step in TWO launches (and one reducer launch): at the batch ns-train draws (4096
// rays, ~68 k rows) a weight-gradient GEMM is 8.9 GFLOP -- 57 us of the matrix pipes -- and seven of them one after the
// other, each followed by its reducer, cost 0.6 ms of launches, half-empty rounds and tails.  The jobs of a batch are
// independent (exact mode: the data-gradient chain leaves every dZ in its own buffer); a workgroup finds its job from
// the prefix of workgroup counts.
constexpr int MAX_WGRAD_JOBS = 8;
struct WgradBatch {
    GemmArgs g[MAX_WGRAD_JOBS];
    int wg0[MAX_WGRAD_JOBS + 1];   // first workgroup of job j; [n] = all
    int tx[MAX_WGRAD_JOBS], ty[MAX_WGRAD_JOBS], nz[MAX_WGRAD_JOBS];
    int n;
};
__device__ __forceinline__ int wgrad_job(const WgradBatch &b, int wg)
{
    int j = 0;
#pragma unroll
    for (int i = 1; i < MAX_WGRAD_JOBS; ++i) j += (i < b.n && wg >= b.wg0[i]) ? 1 : 0;
    return j;
}
__global__ void __launch_bounds__(256, PNR_GEMM_WGS) k_wgrad_batch128(WgradBatch b)
{
    __shared__ float As[2][TK][LDT];
    __shared__ float Bs[2][TK][LDT];
    const int j = wgrad_job(b, (int)blockIdx.x), l = (int)blockIdx.x - b.wg0[j];
    const int bx = l % b.tx[j], by = (l / b.tx[j]) % b.ty[j], bz = l / (b.tx[j] * b.ty[j]);
    gemm_body<true, false, EPI_PARTIAL>(b.g[j], bx, by, bz, b.tx[j], b.nz[j], As, Bs);
}
__global__ void __launch_bounds__(256, 2) k_wgrad_batch96(WgradBatch b)
{
    __shared__ float As[2][TK][256 + 4];
    __shared__ float Bs[2][TK][96 + 4];
    const int j = wgrad_job(b, (int)blockIdx.x), l = (int)blockIdx.x - b.wg0[j];
    wgrad256_body<3>(b.g[j], l % b.ty[j], l / b.ty[j], b.nz[j], As, Bs);
}

// dst[m][n] += sum over the active splits of part[z][m][n]  (part: [nz, M, ld], dst: [M, ld]);
// db[m] += sum over the active splits of csum[z][m]
__global__ void __launch_bounds__(256) k_reduce_parts(const float *__restrict__ part, const float *__restrict__ csum,
                                                      int nz_grid, int gran, int M, int N, int ld,
                                                      const int *__restrict__ dev_rows, float *__restrict__ dst,
                                                      float *__restrict__ db)
{
    const int rows = *dev_rows;
    const int nz = active_splits(rows, nz_grid);
    const int chunk = ((rows + nz - 1) / nz + gran - 1) / gran * gran;   // gran: the GEMM kernel's chunk depth
    const int used = chunk > 0 ? min(nz, (rows + chunk - 1) / chunk) : 0;   // splits with k_begin < rows
    const int n = M * ld;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + M; i += gridDim.x * blockDim.x) {
        const bool is_b = i >= n;
        if (!is_b && i % ld >= N) continue;
        const float *src = is_b ? csum + (i - n) : part + i;
        const int stride = is_b ? M : n;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int z = 0;
        for (; z + 4 <= used; z += 4) {   // four independent loads in flight
            a0 += src[(int64_t)(z + 0) * stride];
            a1 += src[(int64_t)(z + 1) * stride];
            a2 += src[(int64_t)(z + 2) * stride];
            a3 += src[(int64_t)(z + 3) * stride];
        }
        for (; z < used; ++z) a0 += src[(int64_t)z * stride];
        const float t = (a0 + a1) + (a2 + a3);
        if (is_b)
            db[i - n] += t;
        else
            dst[i] += t;
    }
}

// k_reduce_parts for every job of a batched weight-gradient launch: blockIdx.y = job
struct ReduceBatch {
    const float *part[MAX_WGRAD_JOBS], *csum[MAX_WGRAD_JOBS];
    const int *rows[MAX_WGRAD_JOBS];
    float *dst[MAX_WGRAD_JOBS], *db[MAX_WGRAD_JOBS];
    int nz[MAX_WGRAD_JOBS], M[MAX_WGRAD_JOBS], N[MAX_WGRAD_JOBS], ld[MAX_WGRAD_JOBS];
};
__global__ void __launch_bounds__(256) k_reduce_parts_batch(ReduceBatch b)
{
    const int j = blockIdx.y;
    const float *__restrict__ part = b.part[j];
    const float *__restrict__ csum = b.csum[j];
    float *__restrict__ dst = b.dst[j];
    float *__restrict__ db = b.db[j];
    const int M = b.M[j], N = b.N[j], ld = b.ld[j];
    const int rows = *b.rows[j];
    const int nz = active_splits(rows, b.nz[j]);
    const int chunk = ((rows + nz - 1) / nz + TK - 1) / TK * TK;
    const int used = chunk > 0 ? min(nz, (rows + chunk - 1) / chunk) : 0;
    const int n = M * ld;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + M; i += gridDim.x * blockDim.x) {
        const bool is_b = i >= n;
        if (!is_b && i % ld >= N) continue;
        const float *src = is_b ? csum + (i - n) : part + i;
        const int stride = is_b ? M : n;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int z = 0;
        for (; z + 4 <= used; z += 4) {   // the same expression as k_reduce_parts
            a0 += src[(int64_t)(z + 0) * stride];
            a1 += src[(int64_t)(z + 1) * stride];
            a2 += src[(int64_t)(z + 2) * stride];
            a3 += src[(int64_t)(z + 3) * stride];
        }
        for (; z < used; ++z) a0 += src[(int64_t)z * stride];
        const float t = (a0 + a1) + (a2 + a3);
        if (is_b)
            db[i - n] += t;
        else
            dst[i] += t;
    }
}

// The same GEMM for the two row shapes (forward, data gradient: both C[M,N] = A[M,K] . B[N,K]^T once the data
// gradient is given W^T) in the bf16x3 arithmetic of the render's default mode: every fp32 product is
// ah*bh + ah*bl + al*bh on bf16 hi/lo splits with fp32 accumulation (v_mfma_f32_32x32x16_bf16, relative error
// ~2^-16 per product).  Operands are split ONCE, when a chunk is written to LDS (each element then feeds 128
// products); LDS holds hi and lo planes in k-group-major order [k / 8][row][8 bf16], so a lane's 16-byte fragment
// read is conflict-free and IS the MFMA operand.  3 MFMAs of 32 cycles replace 8 of 64: these GEMMs become
// HBM-bound (1 GB of tape in, 1 GB out per layer).
constexpr int BK = 32;  // k per chunk = two MFMA steps

template <int EPI>
__global__ void __launch_bounds__(256, 2) k_gemm_nt_bf16x3(GemmArgs g)
{
    // [buffer][plane: A hi, A lo, B hi, B lo][k group][row]
    __shared__ u32x4 planes[2][4][BK / 8][TM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int M = *g.dev_rows, N = g.N, K = g.K;
    const int wm = wave & 1, wn = wave >> 1;
    constexpr int NL = BK / 8;  // float4 loads per thread and operand per chunk
    // persistent over the (row tile, column tile) pairs, column tile fastest (see k_gemm)
    const int n_nt = (N + TN - 1) / TN;
    const int total = n_nt * ((M + TM - 1) / TM);
    for (int it = blockIdx.x; it < total; it += gridDim.x) {
    const int m0 = (it / n_nt) * TM, n0 = (it % n_nt) * TN;
    float4 ra[NL], rb[NL];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int r = (tid >> 3) + 32 * i, k = k0 + (tid & 7) * 4;
            ra[i] = (m0 + r < M && k < K) ? *reinterpret_cast<const float4 *>(g.A + (int64_t)(m0 + r) * g.lda + k)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = (n0 + r < N && k < K) ? *reinterpret_cast<const float4 *>(g.B + (int64_t)(n0 + r) * g.ldb + k)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto split4 = [](const float4 &v, uint2 &hi, uint2 &lo) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const float x[4] = {v.x, v.y, v.z, v.w};
        bf16x4 hv, lv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __bf16 hb = (__bf16)x[i];
            hv[i] = hb;
            lv[i] = (__bf16)(x[i] - (float)hb);
        }
        hi = __builtin_bit_cast(uint2, hv);
        lo = __builtin_bit_cast(uint2, lv);
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int r = (tid >> 3) + 32 * i, kq = (tid & 7) * 4;
            const int kg = kq >> 3, half = (kq >> 2) & 1;
            uint2 hi, lo;
            split4(ra[i], hi, lo);
            reinterpret_cast<uint2 *>(&planes[buf][0][kg][r])[half] = hi;
            reinterpret_cast<uint2 *>(&planes[buf][1][kg][r])[half] = lo;
            split4(rb[i], hi, lo);
            reinterpret_cast<uint2 *>(&planes[buf][2][kg][r])[half] = hi;
            reinterpret_cast<uint2 *>(&planes[buf][3][kg][r])[half] = lo;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int nchunks = (K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_tiles((c + 1) * BK);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int kg = 2 * s + h;
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ah[q] = __builtin_bit_cast(bf16x8, planes[buf][0][kg][wm * 64 + q * 32 + j]);
                al[q] = __builtin_bit_cast(bf16x8, planes[buf][1][kg][wm * 64 + q * 32 + j]);
                bh[q] = __builtin_bit_cast(bf16x8, planes[buf][2][kg][wn * 64 + q * 32 + j]);
                bl[q] = __builtin_bit_cast(bf16x8, planes[buf][3][kg][wn * 64 + q * 32 + j]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                }
        }
        if (c + 1 < nchunks) store_tiles(buf ^ 1);
        __syncthreads();
    }
    gemm_epilogue<EPI>(g, acc, M, N, m0, n0, wm, wn, lane, reinterpret_cast<float *>(&planes[0][0][0][0]) + wave * 2048);
    __syncthreads();   // before the next tile's operands overwrite the epilogue regions
    }
}

// The row GEMM with its chunk loop FULLY UNROLLED (NCH = ceil(K / 32) is 4, 8 or 9 for this network) and THREE register
// stages: the global loads of chunk c + 3 are issued before the MFMAs of chunk c.  Stamps of the rolled kernel above
// showed a chunk waiting ~3 k cycles for loads issued one chunk (~2 k cycles) earlier -- a memory round trip is ~4.5 k
// cycles here; in straight-line code hipcc counts vmcnt exactly (a loop back-edge makes it wait for everything), and
// the barriers wait for LDS only.
template <int EPI, int NCH>
__global__ void __launch_bounds__(256, 2) k_gemm_nt_bf16x3_u(GemmArgs g)
{
    __shared__ u32x4 planes[2][4][BK / 8][TM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int M = *g.dev_rows, N = g.N, K = g.K;
    const int wm = wave & 1, wn = wave >> 1;
    constexpr int NL = BK / 8;
    const int n_nt = (N + TN - 1) / TN;
    const int total = n_nt * ((M + TM - 1) / TM);
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    for (int it = blockIdx.x; it < total; it += gridDim.x) {
        const int m0 = (it / n_nt) * TM, n0 = (it % n_nt) * TN;
        float4 ra[3][NL], rb[3][NL];
        const int lr = tid >> 3, lk = (tid & 7) * 4;
        // branch-free loads (a predicated load is its own basic block, and hipcc then waits vmcnt(0) at every merge):
        // rows clamped into the matrix, k beyond K read from the row's tail / the next row (every buffer has slack
        // behind it) -- both replaced by zeros with selects when the chunk is written to LDS
        const float *pa[NL], *pb[NL];
        bool oka[NL], okb[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int r = lr + 32 * i;
            oka[i] = m0 + r < M;
            okb[i] = n0 + r < N;
            pa[i] = g.A + (int64_t)min(m0 + r, M - 1) * g.lda + lk;
            pb[i] = g.B + (int64_t)min(n0 + r, N - 1) * g.ldb + lk;
        }
        auto load_tiles = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int P = c % 3;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                ra[P][i] = *reinterpret_cast<const float4 *>(pa[i] + c * BK);
                rb[P][i] = *reinterpret_cast<const float4 *>(pb[i] + c * BK);
            }
        };
        auto split4 = [](const float4 &v, uint2 &hi, uint2 &lo) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            const float x[4] = {v.x, v.y, v.z, v.w};
            bf16x4 hv, lv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const __bf16 hb = (__bf16)x[i];
                hv[i] = hb;
                lv[i] = (__bf16)(x[i] - (float)hb);
            }
            hi = __builtin_bit_cast(uint2, hv);
            lo = __builtin_bit_cast(uint2, lv);
        };
        auto store_tiles = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int P = c % 3, buf = c & 1;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int r = lr + 32 * i;
                const int kg = lk >> 3, half = (lk >> 2) & 1;
                uint2 hi, lo;
                const bool kin = c * BK + lk < K;
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                split4((oka[i] && kin) ? ra[P][i] : z4, hi, lo);
                reinterpret_cast<uint2 *>(&planes[buf][0][kg][r])[half] = hi;
                reinterpret_cast<uint2 *>(&planes[buf][1][kg][r])[half] = lo;
                split4((okb[i] && kin) ? rb[P][i] : z4, hi, lo);
                reinterpret_cast<uint2 *>(&planes[buf][2][kg][r])[half] = hi;
                reinterpret_cast<uint2 *>(&planes[buf][3][kg][r])[half] = lo;
            }
        };
        // one float4 of chunk c's staged registers (q = 0..7: operand q & 1, row group q >> 1): split + two LDS stores
        auto store_piece = [&](auto cc, int q) {
            constexpr int c = decltype(cc)::value;
            constexpr int P = c % 3, buf = c & 1;
            const int i = q >> 1;
            const int r = lr + 32 * i;
            const int kg = lk >> 3, half = (lk >> 2) & 1;
            const bool kin = c * BK + lk < K;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            uint2 hi, lo;
            if ((q & 1) == 0) {
                split4((oka[i] && kin) ? ra[P][i] : z4, hi, lo);
                reinterpret_cast<uint2 *>(&planes[buf][0][kg][r])[half] = hi;
                reinterpret_cast<uint2 *>(&planes[buf][1][kg][r])[half] = lo;
            } else {
                split4((okb[i] && kin) ? rb[P][i] : z4, hi, lo);
                reinterpret_cast<uint2 *>(&planes[buf][2][kg][r])[half] = hi;
                reinterpret_cast<uint2 *>(&planes[buf][3][kg][r])[half] = lo;
            }
        };
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        // the MFMAs of the chunk in LDS buffer `buf`; with NEXT, the split + LDS store of the following chunk is cut into
        // eight pieces and placed behind each block's three MFMAs: an in-order wave cannot issue VALU work that stands
        // behind a whole block of MFMAs, and two waves per SIMD did not overlap the two phases on their own
        auto mfma_chunk = [&](int buf, auto next, auto has_next) {
#pragma unroll
            for (int s = 0; s < BK / 16; ++s) {
                const int kg = 2 * s + h;
                bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    ah[q] = __builtin_bit_cast(bf16x8, planes[buf][0][kg][wm * 64 + q * 32 + j]);
                    al[q] = __builtin_bit_cast(bf16x8, planes[buf][1][kg][wm * 64 + q * 32 + j]);
                    bh[q] = __builtin_bit_cast(bf16x8, planes[buf][2][kg][wn * 64 + q * 32 + j]);
                    bl[q] = __builtin_bit_cast(bf16x8, planes[buf][3][kg][wn * 64 + q * 32 + j]);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        __builtin_amdgcn_sched_barrier(0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                        if constexpr (decltype(has_next)::value) store_piece(next, 4 * s + 2 * a + b);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
        };
        // prologue: three chunks in flight, the first one into LDS
        load_tiles(std::integral_constant<int, 0>());
        if constexpr (NCH > 1) load_tiles(std::integral_constant<int, 1>());
        if constexpr (NCH > 2) load_tiles(std::integral_constant<int, 2>());
        store_tiles(std::integral_constant<int, 0>());
        lds_barrier();
        auto step = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (c + 3 < NCH) load_tiles(std::integral_constant<int, c + 3>());   // into the stage chunk c left
            mfma_chunk(c & 1, std::integral_constant<int, (c + 1 < NCH ? c + 1 : c)>(), std::bool_constant<(c + 1 < NCH)>());
            lds_barrier();
        };
        step(std::integral_constant<int, 0>());
        if constexpr (NCH > 1) step(std::integral_constant<int, 1>());
        if constexpr (NCH > 2) step(std::integral_constant<int, 2>());
        if constexpr (NCH > 3) step(std::integral_constant<int, 3>());
        if constexpr (NCH > 4) step(std::integral_constant<int, 4>());
        if constexpr (NCH > 5) step(std::integral_constant<int, 5>());
        if constexpr (NCH > 6) step(std::integral_constant<int, 6>());
        if constexpr (NCH > 7) step(std::integral_constant<int, 7>());
        if constexpr (NCH > 8) step(std::integral_constant<int, 8>());
        static_assert(NCH <= 9, "unrolled for K <= 288");
        gemm_epilogue<EPI>(g, acc, M, N, m0, n0, wm, wn, lane, reinterpret_cast<float *>(&planes[0][0][0][0]) + wave * 2048);
        lds_barrier();
    }
}

// Weight gradients in the same arithmetic: C[M, N] (partial of split z) = sum over the split's rows k of A[k][m] B[k][n],
// both operands K-MAJOR in memory (dZ and the taped input, [rows, features]).  The MFMA wants 8 consecutive k per
// lane, so the loader transposes in registers: a thread fetches an 8 (k) x 4 (m) block -- eight float4, 512
// contiguous bytes per row over the 32 threads of a k group -- splits it and writes, for each of its four columns, the
// 8 k values as ONE 16-byte hi and ONE lo fragment into the k-group-major planes of k_gemm_nt_bf16x3.  Threads
// 0..127 serve the A operand, 128..255 the B operand.  Splits, partial tiles and partial bias rows as in k_gemm<TA>.
__global__ void __launch_bounds__(256, 2) k_gemm_tn_bf16x3(GemmArgs g)
{
    __shared__ u32x4 planes[2][4][BK / 8][TM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int rows = *g.dev_rows, M = g.M, N = g.N;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    const int nz = active_splits(rows, gridDim.z);
    if ((int)blockIdx.z >= nz) return;
    const int chunk = ((rows + nz - 1) / nz + BK - 1) / BK * BK;
    const int k_begin = min(rows, (int)blockIdx.z * chunk), k_end = min(rows, k_begin + chunk);
    if (k_begin >= k_end) return;
    const int wm = wave & 1, wn = wave >> 1;
    const bool isB = tid >= 128;
    const int t = tid & 127, kg = t >> 5, q4 = (t & 31) * 4;      // this thread's k group and column quad
    const float *src = isB ? g.B : g.A;
    const int ld = isB ? g.ldb : g.lda;
    const int c0 = (isB ? n0 : m0) + q4;
    const bool col_ok = c0 < (isB ? N : M);
    const bool want_csum = g.colsum && blockIdx.y == 0 && !isB;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    // Two register stages (chunk c + 2 is requested before the MFMAs of chunk c), loads branch-free (rows clamped into
    // the split, zeros selected when the chunk is written to LDS: a predicated load is its own basic block and hipcc then
    // waits vmcnt(0) at every merge), LDS-only barriers, and the transpose + split + LDS store of the next chunk cut
    // into four column pieces placed behind MFMA blocks -- the measures that paid in k_gemm_nt_bf16x3_u.
    float4 rg[2][8];
    const int c0c = min(c0, (isB ? N : M) - 4);            // a valid column quad for threads beyond the matrix
    const float *srcc = src + c0c;
    auto load_tiles = [&](int k0, auto par) {
        constexpr int P = decltype(par)::value;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = min(k0 + kg * 8 + i, k_end - 1);
            rg[P][i] = *reinterpret_cast<const float4 *>(srcc + (int64_t)k * ld);
        }
    };
    // column c of the thread's 8 x 4 block of chunk k0: 8 k values -> one hi and one lo fragment
    auto store_piece = [&](int buf, int k0, auto par, int c) {
        constexpr int P = decltype(par)::value;
        const int plane = isB ? 2 : 0;
        bf16x8 hv, lv;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 r4 = rg[P][i];
            float x = c == 0 ? r4.x : (c == 1 ? r4.y : (c == 2 ? r4.z : r4.w));
            x = (col_ok && k0 + kg * 8 + i < k_end) ? x : 0.f;
            const __bf16 hb = (__bf16)x;
            hv[i] = hb;
            lv[i] = (__bf16)(x - (float)hb);
            if (want_csum) {
                if (c == 0) csum.x += x;
                if (c == 1) csum.y += x;
                if (c == 2) csum.z += x;
                if (c == 3) csum.w += x;
            }
        }
        planes[buf][plane][kg][q4 + c] = __builtin_bit_cast(u32x4, hv);
        planes[buf][plane + 1][kg][q4 + c] = __builtin_bit_cast(u32x4, lv);
    };
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // MFMAs of the chunk in LDS buffer `buf`; the next chunk (first row k_next, register stage `par`) goes to buf ^ 1
    auto mfma_chunk = [&](int buf, int k_next, auto par) {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int kgs = 2 * s + h;
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ah[q] = __builtin_bit_cast(bf16x8, planes[buf][0][kgs][wm * 64 + q * 32 + j]);
                al[q] = __builtin_bit_cast(bf16x8, planes[buf][1][kgs][wm * 64 + q * 32 + j]);
                bh[q] = __builtin_bit_cast(bf16x8, planes[buf][2][kgs][wn * 64 + q * 32 + j]);
                bl[q] = __builtin_bit_cast(bf16x8, planes[buf][3][kgs][wn * 64 + q * 32 + j]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    __builtin_amdgcn_sched_barrier(0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                    if (b == 0) store_piece(buf ^ 1, k_next, par, 2 * s + a);   // four pieces per chunk
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };
    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, 1> P1;
    const int nchunks = (k_end - k_begin + BK - 1) / BK;
    load_tiles(k_begin, P0());
    load_tiles(k_begin + BK, P1());
#pragma unroll
    for (int c = 0; c < 4; ++c) store_piece(0, k_begin, P0(), c);
    lds_barrier();
    // two chunks per iteration; chunks beyond the split are all-zero (their loads are clamped, their values selected away)
    for (int c = 0; c < nchunks; c += 2) {
        const int k0 = k_begin + c * BK;
        load_tiles(k0 + 2 * BK, P0());                 // chunk c + 2 into the stage chunk c left
        mfma_chunk(0, k0 + BK, P1());                  // chunk c; chunk c + 1 -> LDS buffer 1
        lds_barrier();
        load_tiles(k0 + 3 * BK, P1());
        mfma_chunk(1, k0 + 2 * BK, P0());              // chunk c + 1; chunk c + 2 -> LDS buffer 0
        lds_barrier();
    }
    float *ldsf = reinterpret_cast<float *>(&planes[0][0][0][0]);
    if (g.colsum && blockIdx.y == 0) {   // uniform over the workgroup
        float *cs = ldsf + 4 * 2048;     // behind the four epilogue regions
        __syncthreads();
        // the four k groups' partial rows, added in a fixed order (no LDS float atomics: bitwise repeatable)
        if (!isB) *reinterpret_cast<float4 *>(&cs[kg * TM + q4]) = csum;
        __syncthreads();
        if (tid < TM && m0 + tid < M)
            g.colsum[(int64_t)blockIdx.z * g.M + m0 + tid] = ((cs[tid] + cs[TM + tid]) + cs[2 * TM + tid]) + cs[3 * TM + tid];
    }
    GemmArgs gp = g;
    gp.C = g.C + (int64_t)blockIdx.z * g.M * g.ldc;
    gemm_epilogue<EPI_PARTIAL>(gp, acc, M, N, m0, n0, wm, wn, lane, ldsf + wave * 2048);
}

// ------------------------------------------------------------------------------------------------
// workspace of a backward call
// ------------------------------------------------------------------------------------------------
constexpr int LD_X0 = 288, LD_H = 256, LD_H2 = 264, LD_XC = 288, LD_C = 128;
// padded weights / weight gradients: [out, ld] with the reference's [out, in] in the leading columns
static const int W_OUT[9] = {256, 256, 256, 256, 1, 128, 128, 128, 3};
static const int W_IN[9] = {284, 256, 263, 256, 256, 280, 128, 128, 128};
static const int W_LD[9] = {288, 256, 264, 256, 256, 288, 128, 128, 128};

#ifndef PNR_SPLIT_WGS
#define PNR_SPLIT_WGS 768
#endif
constexpr int MAX_SPLIT_WGS = PNR_SPLIT_WGS;                        // workgroups of one weight-gradient GEMM
constexpr size_t PART_FLOATS = (size_t)MAX_SPLIT_WGS * TM * TN;     // their partial tiles
constexpr size_t CSUM_FLOATS = (size_t)MAX_SPLIT_WGS * 256;         // + one partial bias-gradient row per split
// the batched launch keeps the partial tiles of all seven GEMMs at once: 2 x 192 x 256 x 256 + 170 x 256 x (264 + 288) +
// 2 x 256 x 128 x 128 + 85 x 128 x 288 floats at the split counts of a large workspace, + the bias rows
constexpr size_t POOL_FLOATS = (size_t)62 << 20;

struct TrainWs {
    int *cnt;        // [0] rows = S * K, [1] S valid samples
    int *s2v;        // [cap] sample index -> valid index or -1
    int *row_pidx;   // [cap * K]
    float *row_w;    // [cap * K] normalised inverse-distance weight
    float *row_z;    // [cap * K] density head before the ReLU
    float *sig;      // [cap] density per valid sample
    float4 *sg;      // [cap] sigmoid outputs of the colour head
    float4 *d_out;   // [cap] (d sigma, d rgb) per valid sample
    float *tmpT;     // [cap] transmittance in front of the sample
    float *tmpD;     // [cap] segment length
    float *X0, *H1, *H2, *G1, *G2;  // [cap * K, ld]
    float *XC, *C1, *C2, *C3;       // [cap, ld]
    unsigned long long *sgH1, *sgH2, *sgG1, *sgC1, *sgC2;   // LeakyReLU sign bits of the taped activations (gemm_epilogue)
    float *Wp[9], *dWp[9], *dbp[9];
    float *WT[9];                   // W^T, [W_LD (inputs, zero rows beyond in) , out]: B operand of the bf16x3 data GEMMs
    float *part;                    // partial tiles of the weight-gradient GEMM in flight: [splits, M, ld]
    float *dw_begin;                // dWp / dbp are contiguous: [dw_begin, dw_begin + dw_floats)
    size_t dw_floats;
    // point gradients: rows grouped by their point (a segmented, order-fixed sum instead of float atomics)
    int *pt_cnt;      // [rows + 1] rows per touched point (indexed by the point's rank in the call's pt_list)
    int *pt_start;    // [rows + 1] exclusive scan of pt_cnt
    int *pt_cursor;   // [rows]
    int *pt_rows;     // [rows] row ids grouped by point
    void *pt_scan;    // scan scratch
    // exact mode (k_train_pairs_bwd, pnr_train_chain.hip): the gradients at the four pre-activations of the pair MLPs
    // [rows, 256] each, the per-row point gradients [rows, 40] and the chain's packed W^T stream
    float *D3, *D2, *D1, *D0;
    float *rowgrad;
    float *chainW;
    float *D7, *D6, *D5;   // [cap, 128] gradients at the colour MLP's three pre-activations (exact mode: no taped
                           // activation is overwritten, so a backward may be repeated on the same taped render)
    float *DAGG;           // [cap, 256] gradient of the aggregated features
    unsigned *tape_bits;   // [4 layers: H1, H2, G1, G2][rows][2 lane halves][4]: LeakyReLU masks as bits (ShadeParams.tape_bits)
    size_t bits_rows;      // rows per layer
    size_t total;
};

static TrainWs carve_train_ws(void *base, int64_t cap, int K)
{
    TrainWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *p = base ? (void *)((char *)base + off) : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    const size_t rows = (size_t)cap * K + TM;  // a tile of slack behind the last row
    const size_t smp = (size_t)cap + TM;
    w.cnt = (int *)take(64);
    w.s2v = (int *)take(smp * 4);
    w.row_pidx = (int *)take(rows * 4);
    w.row_w = (float *)take(rows * 4);
    w.row_z = (float *)take(rows * 4);
    w.sig = (float *)take(smp * 4);
    w.sg = (float4 *)take(smp * 16);
    w.d_out = (float4 *)take(smp * 16);
    w.tmpT = (float *)take(smp * 4);
    w.tmpD = (float *)take(smp * 4);
    w.X0 = (float *)take(rows * LD_X0 * 4);
    w.H1 = (float *)take(rows * LD_H * 4);
    w.H2 = (float *)take(rows * LD_H2 * 4);
    w.G1 = (float *)take(rows * LD_H * 4);
    w.G2 = (float *)take(rows * LD_H * 4);
    w.XC = (float *)take(smp * LD_XC * 4);
    w.C1 = (float *)take(smp * LD_C * 4);
    w.C2 = (float *)take(smp * LD_C * 4);
    w.C3 = (float *)take(smp * LD_C * 4);
    {
        const size_t mt_rows = (rows + TM - 1) / TM, mt_smp = (smp + TM - 1) / TM;
        w.sgH1 = (unsigned long long *)take(mt_rows * 2 * 2048);
        w.sgH2 = (unsigned long long *)take(mt_rows * 2 * 2048);
        w.sgG1 = (unsigned long long *)take(mt_rows * 2 * 2048);
        w.sgC1 = (unsigned long long *)take(mt_smp * 2048);
        w.sgC2 = (unsigned long long *)take(mt_smp * 2048);
    }
    for (int i = 0; i < 9; ++i) w.Wp[i] = (float *)take((size_t)W_OUT[i] * W_LD[i] * 4);
    for (int i = 0; i < 9; ++i) w.WT[i] = (float *)take((size_t)W_OUT[i] * W_LD[i] * 4);
    w.part = (float *)take(std::max((size_t)(PART_FLOATS + CSUM_FLOATS), POOL_FLOATS) * 4);
    const size_t dw0 = off;
    for (int i = 0; i < 9; ++i) w.dWp[i] = (float *)take((size_t)W_OUT[i] * W_LD[i] * 4);
    for (int i = 0; i < 9; ++i) w.dbp[i] = (float *)take((size_t)W_OUT[i] * 4);
    w.dw_begin = base ? (float *)((char *)base + dw0) : nullptr;
    w.dw_floats = (off - dw0) / 4;
    w.pt_cnt = (int *)take((rows + 1) * 4);
    w.pt_start = (int *)take((rows + 1) * 4);
    w.pt_cursor = (int *)take(rows * 4);
    w.pt_rows = (int *)take(rows * 4);
    w.pt_scan = take(scan_temp_bytes((int64_t)rows + 1));
    w.D3 = (float *)take(rows * LD_H * 4);
    w.D2 = (float *)take(rows * LD_H * 4);
    w.D1 = (float *)take(rows * LD_H * 4);
    w.D0 = (float *)take(rows * LD_H * 4);
    w.rowgrad = (float *)take(rows * 40 * 4);
    w.chainW = (float *)take((CHAIN_W_FLOATS + 256) * 4);
    w.D6 = (float *)take(smp * LD_C * 4);
    w.D5 = (float *)take(smp * LD_C * 4);
    w.DAGG = (float *)take(smp * 256 * 4);
    w.tape_bits = (unsigned *)take(rows * 128);
    w.bits_rows = rows;
    w.D7 = (float *)take(smp * LD_C * 4);
    w.total = off;
    return w;
}

bool train_tape_ptrs(void *d_train_workspace, size_t bytes, int64_t cap, int K, float *tape[4], size_t tape_bytes[4],
                     unsigned **tape_bits, size_t *tape_bits_rows, float **tape_rowz, float *ctape[3], void **tape_sg)
{
    if (!d_train_workspace || carve_train_ws(nullptr, cap, K).total > bytes) return false;
    const TrainWs w = carve_train_ws(d_train_workspace, cap, K);
    const size_t rows = (size_t)cap * K + TM;
    *tape_bits = w.tape_bits;
    *tape_bits_rows = w.bits_rows;
    *tape_rowz = w.row_z;
    ctape[0] = w.C1;
    ctape[1] = w.C2;
    ctape[2] = w.C3;
    *tape_sg = w.sg;
    tape[0] = w.H1;
    tape[1] = w.H2;
    tape[2] = w.G1;
    tape[3] = w.G2;
    tape_bytes[0] = tape_bytes[2] = tape_bytes[3] = rows * LD_H * 4;
    tape_bytes[1] = rows * LD_H2 * 4;
    return true;
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_train_set_cams(CamSet set, int n, Camera *__restrict__ dst)
{
    const int i = threadIdx.x;
    if (i < n) dst[i] = set.c[i];
}

// The start of a backward in ONE launch: the row / sample counts, and the clears every later kernel relies on -- sized by the
// call's DEVICE-side counts, not by the workspace capacity (a plugin step's workspace is sized for the worst case R x SR:
// five memsets over capacity-sized arrays were 28 MB per step at 4096 rays and 440 MB at 65 536, besides their launches):
//   s2v [selected samples] = -1, d_out [valid samples] = 0, the padded weight-gradient buffers = 0,
//   pt_cnt [U + 1] = 0, pt_cursor [U] = 0  (U distinct neighbour points: everything that indexes them uses a rank < U)
__global__ void __launch_bounds__(256) k_train_init(const int *__restrict__ n_sel, int cap, int K, TrainWs w)
{
    const int S_sel = min(n_sel[0], cap), S = min(n_sel[1], cap), U = n_sel[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        w.cnt[0] = S * K;
        w.cnt[1] = S;
    }
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = t0; i < S_sel; i += nt) w.s2v[i] = -1;
    for (int64_t i = t0; i < S; i += nt) w.d_out[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = t0; i < (int64_t)w.dw_floats; i += nt) w.dw_begin[i] = 0.f;
    for (int64_t i = t0; i <= U; i += nt) w.pt_cnt[i] = 0;
    for (int64_t i = t0; i < U; i += nt) w.pt_cursor[i] = 0;
}

__global__ void k_train_s2v(const int *__restrict__ vs_list, const int *__restrict__ cnt, int *__restrict__ s2v)
{
    const int S = cnt[1];
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < S; v += gridDim.x * blockDim.x) s2v[vs_list[v]] = v;
}

// the nine Linear layers in one launch (blockIdx.y = layer): dst[out, ld] = src[out, in] zero-padded
struct NineMats {
    const float *src[9];
    float *dst[9];
    float *dstT[9];   // k_pad_weights only: transposed copy [ld, out]
    int n_out[9], n_in[9], ld[9];
};
__global__ void k_pad_weights(NineMats m)
{
    const int l = blockIdx.y;
    const int ld = m.ld[l], n_in = m.n_in[l], n = m.n_out[l] * ld;
    const float *__restrict__ src = m.src[l];
    float *__restrict__ dst = m.dst[l];
    float *__restrict__ dstT = m.dstT[l];
    const int n_out = m.n_out[l];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int o = i / ld, c = i - o * ld;
        const float v = c < n_in ? src[(int64_t)o * n_in + c] : 0.f;
        dst[i] = v;
        if (dstT) dstT[(int64_t)c * n_out + o] = v;
    }
}

// dst[out, in] += src[out, ld]  (dst null: skipped)
__global__ void k_unpad_add(NineMats m)
{
    const int l = blockIdx.y;
    if (!m.dst[l]) return;
    const int ld = m.ld[l], n_in = m.n_in[l], n = m.n_out[l] * n_in;
    const float *__restrict__ src = m.src[l];
    float *__restrict__ dst = m.dst[l];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int o = i / n_in, c = i - o * n_in;
        dst[i] += src[(int64_t)o * ld + c];
    }
}

struct TrainParams {
    const float4 *point_rows;
    float Rw2c[9];
    CamRef cr;
    const float *dirs;
    const float4 *smp_loc;
    const int *smp_ray;
    const int *smp_pidx;
    const int *vs_list;
    const int *n_sel;
    int K;
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wavefront per (sample, neighbour slot) row: the 284 inputs of mlp_base layer 0
// [emb(32) | (sin, cos)(emb * 2^f), f < 3 (192) | (sin, cos)(dist * 2^f), f < 5 (60)]  (studio_model.py:309-317,
// studio_utils.py:58-68), the 7 extra inputs of mlp_head layer 0 and the row's aggregation weight
// (studio_model.py:270-286).  Unfilled slots (pidx < 0) get zero inputs and weight 0: their gradient vanishes.
template <bool FAST_PE>
__global__ void __launch_bounds__(256) k_train_rows(TrainParams P, TrainWs w)
{
    // one wavefront per SAMPLE: the sample's position, ray, camera and its K neighbour indices / positions are loaded
    // once (lane k < K holds slot k) and handed to the K rows with cross-lane reads -- a wavefront per row paid the
    // four dependent load levels vs_list -> sample -> neighbour list -> point row for every row
    const int lane = threadIdx.x & 63;
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwv = (gridDim.x * blockDim.x) >> 6;
    const int S = w.cnt[1], K = P.K;
    // a row is assembled in wave-private LDS and leaves as 72 float4 (two store instructions of contiguous KiB) instead
    // of five partial-line store instructions
    __shared__ float rbuf[4][LD_X0];
    for (int v = wv; v < S; v += nwv) {
        const int s = P.vs_list[v];
        const float4 loc = P.smp_loc[s];
        const int ray = P.smp_ray[s];
        const int pk = lane < K ? P.smp_pidx[(int64_t)s * K + lane] : -1;
        float4 ak = make_float4(0.f, 0.f, 0.f, 0.f);
        float wl = 0.f;
        if (pk >= 0) {
            ak = P.point_rows[(int64_t)pk * 12];
            const float dx = ak.x - loc.x, dy = ak.y - loc.y, dz = ak.z - loc.z;
            wl = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f);
        }
        const float wsum = fmaxf(wave_sum(wl), 1e-8f);
        const Camera cam = load_cam(P.cr, cam_id(P.cr, ray));
        float vx, vy, vz, scx, scy, scz;
        rot_rows(P.Rw2c, P.dirs[3 * (int64_t)ray], P.dirs[3 * (int64_t)ray + 1], P.dirs[3 * (int64_t)ray + 2], vx, vy, vz);
        to_cam(cam, loc.x, loc.y, loc.z, scx, scy, scz);
        const float spx = scx / scz, spy = scy / scz;
        for (int k = 0; k < K; ++k) {
            const int row = v * K + k;
            const int pidx = __shfl(pk, k, 64);
            const bool valid = pidx >= 0;
            const float a0x = __shfl(ak.x, k, 64), a0y = __shfl(ak.y, k, 64), a0z = __shfl(ak.z, k, 64);
            const float wk = __shfl(wl, k, 64) / wsum;
            const float4 *prow = P.point_rows + (int64_t)max(pidx, 0) * 12;
            const float *emb = reinterpret_cast<const float *>(prow + 4);
            const float dwx = a0x - loc.x, dwy = a0y - loc.y, dwz = a0z - loc.z;
            float dd[6];
            rot_rows(P.Rw2c, dwx, dwy, dwz, dd[0], dd[1], dd[2]);
            {
                float pcx, pcy, pcz;
                to_cam(cam, a0x, a0y, a0z, pcx, pcy, pcz);
                const float ppx = pcx / pcz, ppy = pcy / pcz;
                dd[3] = ppx * pcz - spx * scz;
                dd[4] = ppy * pcz - spy * scz;
                dd[5] = pcz - scz;
            }
            float *x0 = rbuf[threadIdx.x >> 6];
            __builtin_amdgcn_wave_barrier();
            // columns 32 + 2u, 32 + 2u + 1 = (sin, cos) pair u: u < 96 embedding channel u / 3 at octave u % 3,
            // u >= 96 distance component (u - 96) / 5 at octave (u - 96) % 5 -- one sincosf per pair
            if (lane < 32) x0[lane] = valid ? emb[lane] : 0.f;
            if (lane < 4) x0[284 + lane] = 0.f;
            if (FAST_PE) {
                // the bf16x3 mode's encodings (as the render's point_inputs / pair_inputs): one reduced-range sincos per
                // channel, the higher octaves by double-angle steps; lane d < 32 owns embedding channel d (3 octaves),
                // lanes 32..37 the six distance components (5 octaves)
                if (lane < 38) {
                    const bool is_e = lane < 32;
                    float dv = dd[0];
                    dv = lane == 33 ? dd[1] : dv;
                    dv = lane == 34 ? dd[2] : dv;
                    dv = lane == 35 ? dd[3] : dv;
                    dv = lane == 36 ? dd[4] : dv;
                    dv = lane == 37 ? dd[5] : dv;
                    const float arg = is_e ? emb[min(lane, 31)] : dv;
                    float *dst = is_e ? x0 + 32 + 6 * lane : x0 + 224 + 10 * (lane - 32);
                    const int nf = is_e ? 3 : 5;
                    float sn, cs;
                    fast_sincos(arg, sn, cs);
#pragma unroll
                    for (int f = 0; f < 5; ++f) {
                        if (f < nf) *reinterpret_cast<float2 *>(dst + 2 * f) = valid ? make_float2(sn, cs) : make_float2(0.f, 0.f);
                        const float s2 = 2.0f * sn * cs, c2 = (cs - sn) * (cs + sn);
                        sn = s2;
                        cs = c2;
                    }
                }
            } else {
                for (int u = lane; u < 126; u += 64) {
                    float arg;
                    if (u < 96) {
                        const int d = u / 3, f = u - 3 * d;
                        arg = emb[d] * (float)(1 << f);
                    } else {
                        const int q = u - 96, d = q / 5, f = q - 5 * d;
                        float dv = dd[0];
                        dv = d == 1 ? dd[1] : dv;
                        dv = d == 2 ? dd[2] : dv;
                        dv = d == 3 ? dd[3] : dv;
                        dv = d == 4 ? dd[4] : dv;
                        dv = d == 5 ? dd[5] : dv;
                        arg = dv * (float)(1 << f);
                    }
                    // the exact mode's render encodes with the branch-free Cody-Waite sincos, every octave from its own
                    // argument (point_inputs / pair_inputs<.., true, false>): the same function here, so the taped inputs
                    // are the ones the render multiplied (libm's sincosf: ~150 instructions a call, half this kernel)
                    float sn, cs;
                    fast_sincos_nb(arg, sn, cs);
                    *reinterpret_cast<float2 *>(x0 + 32 + 2 * u) = valid ? make_float2(sn, cs) : make_float2(0.f, 0.f);
                }
            }
            __builtin_amdgcn_wave_barrier();
            {
                float4 *gx = reinterpret_cast<float4 *>(w.X0 + (int64_t)row * LD_X0);
                gx[lane] = reinterpret_cast<const float4 *>(x0)[lane];
                if (lane < LD_X0 / 4 - 64) gx[64 + lane] = reinterpret_cast<const float4 *>(x0)[64 + lane];
            }
            if (lane < 8) {
                const float4 c0 = prow[1], c1 = prow[2];
                float sdx, sdy, sdz;
                rot_rows(P.Rw2c, c0.w, c1.x, c1.y, sdx, sdy, sdz);
                float e = 0.f;
                e = lane == 0 ? c0.x : e;
                e = lane == 1 ? c0.y : e;
                e = lane == 2 ? c0.z : e;
                e = lane == 3 ? sdx - vx : e;
                e = lane == 4 ? sdy - vy : e;
                e = lane == 5 ? sdz - vz : e;
                e = lane == 6 ? sdx * vx + sdy * vy + sdz * vz : e;
                w.H2[(int64_t)row * LD_H2 + 256 + lane] = valid ? e : 0.f;
            }
            if (lane == 0) {
                w.row_pidx[row] = pidx;
                w.row_w[row] = valid ? wk : 0.f;
            }
        }
    }
}

// k_train_rows<false> for K <= 32, with the per-row scalar work done ONCE per sample, lane k for row k: the six encoded
// distance components, the seven extra head inputs, the weight -- where k_train_rows has all 64 lanes recompute them for
// every row (275 vector instructions per row, 0.9 ms at 65 536 rays: the kernel is bound by its instruction count).  The
// row loop is left with the embedding, the 126 (sin, cos) pairs -- two passes of one pair per lane, every lane with fixed
// source and destination offsets into the wave's LDS row -- and the row's store.  Same expressions, same values.
__global__ void __launch_bounds__(256) k_train_rows_x(TrainParams P, TrainWs w)
{
    const int lane = threadIdx.x & 63, wl_ = threadIdx.x >> 6;
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwv = (gridDim.x * blockDim.x) >> 6;
    const int S = w.cnt[1], K = P.K;
    // a wave's LDS: the row being assembled (288 floats) and behind it the sample's distance components [K][8]
    __shared__ float rbuf[4][LD_X0 + 32 * 8];
    float *x0 = rbuf[wl_];
    float *ddl = x0 + LD_X0;
    // pass 0: pair u = lane (embedding channel u / 3, octave u % 3); pass 1: pair u = 64 + lane (< 126): embedding for
    // u < 96, else distance component (u - 96) / 5 at octave (u - 96) % 5
    const int u1 = 64 + lane;
    const bool act1 = u1 < 126, e1 = u1 < 96;
    const int d0 = lane / 3, f0 = lane - 3 * d0;
    const int q1 = e1 ? u1 : u1 - 96, d1 = e1 ? q1 / 3 : q1 / 5, f1 = e1 ? q1 - 3 * d1 : q1 - 5 * d1;
    const float s0 = (float)(1 << f0), s1 = (float)(1 << f1);
    const int src1 = e1 ? d1 : LD_X0 + d1, kmul1 = e1 ? 0 : 8;   // + k * kmul1: the row's line of the distance table
    for (int v = wv; v < S; v += nwv) {
        const int s = P.vs_list[v];
        const float4 loc = P.smp_loc[s];
        const int ray = P.smp_ray[s];
        const int pk = lane < K ? P.smp_pidx[(int64_t)s * K + lane] : -1;
        const bool valid = pk >= 0;
        const float4 *prow = P.point_rows + (int64_t)max(pk, 0) * 12;
        float4 ak = make_float4(0.f, 0.f, 0.f, 0.f);
        float wl = 0.f;
        if (valid) {
            ak = prow[0];
            const float dx = ak.x - loc.x, dy = ak.y - loc.y, dz = ak.z - loc.z;
            wl = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f);
        }
        const float wsum = fmaxf(wave_sum(wl), 1e-8f);
        const Camera cam = load_cam(P.cr, cam_id(P.cr, ray));
        float vx, vy, vz, scx, scy, scz;
        rot_rows(P.Rw2c, P.dirs[3 * (int64_t)ray], P.dirs[3 * (int64_t)ray + 1], P.dirs[3 * (int64_t)ray + 2], vx, vy, vz);
        to_cam(cam, loc.x, loc.y, loc.z, scx, scy, scz);
        const float spx = scx / scz, spy = scy / scz;
        __builtin_amdgcn_wave_barrier();
        if (lane < K) {
            // this lane's row: distances, extras, weight
            const float dwx = ak.x - loc.x, dwy = ak.y - loc.y, dwz = ak.z - loc.z;
            float dd[6];
            rot_rows(P.Rw2c, dwx, dwy, dwz, dd[0], dd[1], dd[2]);
            {
                float pcx, pcy, pcz;
                to_cam(cam, ak.x, ak.y, ak.z, pcx, pcy, pcz);
                const float ppx = pcx / pcz, ppy = pcy / pcz;
                dd[3] = ppx * pcz - spx * scz;
                dd[4] = ppy * pcz - spy * scz;
                dd[5] = pcz - scz;
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) ddl[lane * 8 + c] = dd[c];
            const float4 c0 = prow[1], c1 = prow[2];
            float sdx, sdy, sdz;
            rot_rows(P.Rw2c, c0.w, c1.x, c1.y, sdx, sdy, sdz);
            const int row = v * K + lane;
            float4 ea = make_float4(c0.x, c0.y, c0.z, sdx - vx);
            float4 eb = make_float4(sdy - vy, sdz - vz, sdx * vx + sdy * vy + sdz * vz, 0.f);
            if (!valid) ea = eb = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 *he = reinterpret_cast<float4 *>(w.H2 + (int64_t)row * LD_H2 + 256);
            he[0] = ea;
            he[1] = eb;
            w.row_pidx[row] = pk;
            w.row_w[row] = valid ? wl / wsum : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = 0; k < K; ++k) {
            const int pidx = __shfl(pk, k, 64);
            const bool vk = pidx >= 0;
            const float *emb = reinterpret_cast<const float *>(P.point_rows + (int64_t)max(pidx, 0) * 12 + 4);
            if (lane < 32) x0[lane] = vk ? emb[lane] : 0.f;
            if (lane < 4) x0[284 + lane] = 0.f;
            __builtin_amdgcn_wave_barrier();
            {
                float sn, cs;
                fast_sincos_nb(x0[d0] * s0, sn, cs);
                *reinterpret_cast<float2 *>(x0 + 32 + 2 * lane) = vk ? make_float2(sn, cs) : make_float2(0.f, 0.f);
            }
            if (act1) {
                float sn, cs;
                fast_sincos_nb(x0[src1 + k * kmul1] * s1, sn, cs);
                *reinterpret_cast<float2 *>(x0 + 32 + 2 * u1) = vk ? make_float2(sn, cs) : make_float2(0.f, 0.f);
            }
            __builtin_amdgcn_wave_barrier();
            {
                float4 *gx = reinterpret_cast<float4 *>(w.X0 + (int64_t)(v * K + k) * LD_X0);
                gx[lane] = reinterpret_cast<const float4 *>(x0)[lane];
                if (lane < LD_X0 / 4 - 64) gx[64 + lane] = reinterpret_cast<const float4 *>(x0)[64 + lane];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// One wavefront per valid sample: density head, weighted K-aggregation (studio_model.py:337-353) and the colour
// MLP's input row [agg(256) | sin(view * 2^f) (12) | cos(...) (12)] (studio_model.py:304-308,355).
__global__ void __launch_bounds__(256) k_train_head_agg(TrainParams P, TrainWs w, const float *__restrict__ w4,
                                                        const float *__restrict__ b4)
{
    const int lane = threadIdx.x & 63;
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwv = (gridDim.x * blockDim.x) >> 6;
    const int S = w.cnt[1], K = P.K;
    const float4 wq = *reinterpret_cast<const float4 *>(w4 + 4 * lane);
    const float bias = b4[0];
    for (int v = wv; v < S; v += nwv) {
        float4 agg = make_float4(0.f, 0.f, 0.f, 0.f);
        float sigma = 0.f;
        for (int k = 0; k < K; ++k) {
            const int row = v * K + k;
            const float4 gq = *reinterpret_cast<const float4 *>(w.G2 + (int64_t)row * LD_H + 4 * lane);
            const float z = wave_sum(gq.x * wq.x + gq.y * wq.y + gq.z * wq.z + gq.w * wq.w) + bias;
            const float wk = w.row_w[row];
            if (lane == 0) w.row_z[row] = z;
            sigma += wk * fmaxf(z, 0.f);
            agg.x += wk * gq.x;
            agg.y += wk * gq.y;
            agg.z += wk * gq.z;
            agg.w += wk * gq.w;
        }
        float *xc = w.XC + (int64_t)v * LD_XC;
        *reinterpret_cast<float4 *>(xc + 4 * lane) = agg;
        if (lane < 32) {
            const int ray = P.smp_ray[P.vs_list[v]];
            float vv[3];
            rot_rows(P.Rw2c, P.dirs[3 * (int64_t)ray], P.dirs[3 * (int64_t)ray + 1], P.dirs[3 * (int64_t)ray + 2], vv[0],
                     vv[1], vv[2]);
            float val = 0.f;
            if (lane < 24) {
                const int q = lane % 12, d = q >> 2, f = q & 3;
                const float dv = d == 0 ? vv[0] : (d == 1 ? vv[1] : vv[2]);
                float sn, cs;
                sincosf(dv * (float)(1 << f), &sn, &cs);
                val = lane < 12 ? sn : cs;
            }
            xc[256 + lane] = val;
        }
        if (lane == 0) w.sig[v] = sigma;
    }
}

// The same products after a render that wrote the tape: the render already holds them -- the aggregated features (its
// colour kernel's blocked layout, agg_idx4), the densities, and (written with the tape) every row's density
// pre-activation.  One wavefront per valid sample copies its 256 aggregated features into the colour MLP's input row and
// encodes the view direction; the 1-KiB-per-row read of G2 that k_train_head_agg needs is gone.
__global__ void __launch_bounds__(256) k_train_head_taped(TrainParams P, TrainWs w, const float4 *__restrict__ agg,
                                                          const float *__restrict__ smp_sigma)
{
    const int lane = threadIdx.x & 63;
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwv = (gridDim.x * blockDim.x) >> 6;
    const int S = w.cnt[1];
    // float4 `lane` of a row = features 4 lane .. 4 lane + 3 = accumulator tile t, quarter q, lane half h
    const int t = lane >> 3, q = (lane & 7) >> 1, h = lane & 1;
    const int in_block = (2 * t + (q >> 1)) * 128 + h * 64 + (q & 1) * 32;
    for (int v = wv; v < S; v += nwv) {
        float *xc = w.XC + (int64_t)v * LD_XC;
        *reinterpret_cast<float4 *>(xc + 4 * lane) = agg[(int64_t)(v >> 5) * 2048 + in_block + (v & 31)];
        if (lane < 32) {
            const int ray = P.smp_ray[P.vs_list[v]];
            float vv[3];
            rot_rows(P.Rw2c, P.dirs[3 * (int64_t)ray], P.dirs[3 * (int64_t)ray + 1], P.dirs[3 * (int64_t)ray + 2], vv[0],
                     vv[1], vv[2]);
            float val = 0.f;
            if (lane < 24) {
                const int qq = lane % 12, d = qq >> 2, f = qq & 3;
                const float dv = d == 0 ? vv[0] : (d == 1 ? vv[1] : vv[2]);
                float sn, cs;
                sincosf(dv * (float)(1 << f), &sn, &cs);
                val = lane < 12 ? sn : cs;
            }
            xc[256 + lane] = val;
        }
        if (lane == 0) w.sig[v] = smp_sigma[v];
    }
}

// colour head: Linear 128 -> 3, sigmoid (the widening happens where the value is used) (studio_model.py:357-359)
__global__ void __launch_bounds__(256) k_train_color_head(TrainWs w, const float *__restrict__ w8,
                                                          const float *__restrict__ b8)
{
    const int lane = threadIdx.x & 63;
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwv = (gridDim.x * blockDim.x) >> 6;
    const int S = w.cnt[1];
    for (int v = wv; v < S; v += nwv) {
        const float c0 = w.C3[(int64_t)v * LD_C + lane], c1 = w.C3[(int64_t)v * LD_C + 64 + lane];
        float sg[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float z = wave_sum(c0 * w8[c * 128 + lane] + c1 * w8[c * 128 + 64 + lane]) + b8[c];
            sg[c] = 1.0f / (1.0f + expf(-z));
        }
        if (lane == 0) w.sg[v] = make_float4(sg[0], sg[1], sg[2], 0.f);
    }
}

// One thread per ray: the composite of k_composite (studio_model.py:368-390) forwards, then its reverse scan.
//   w_i = o_i T_i,  T_{i+1} = T_i (1 - o_i + 1e-10),  o_i = 1 - exp(-sigma_i delta_i),  out = sum w c + bg (1 - sum w)
__global__ void __launch_bounds__(256) k_train_composite_bwd(CamRef cr, pnr_render_opts_t opts, int64_t R,
                                                             const int *__restrict__ ray_cnt,
                                                             const int *__restrict__ ray_off,
                                                             const int *__restrict__ ray_flag,
                                                             const float4 *__restrict__ smp_loc,
                                                             const int *__restrict__ n_sel, TrainWs w,
                                                             const float *__restrict__ g_rgb,
                                                             float *__restrict__ rgb_out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int S = n_sel[0];
    const int off = ray_off[r];
    int cnt = ray_cnt[r];
    if ((int64_t)off + cnt > S) cnt = max(0, S - off);
    const bool keep = ray_flag[r] != 0 && cnt > 0;
    float o0 = opts.bg[0], o1 = opts.bg[1], o2 = opts.bg[2];
    if (keep) {
        const Camera cam = load_cam(cr, cam_id(cr, r));
        const float vs = opts.vsize_z, two_vs = 2.0f * vs;
        auto zc = [&](float x, float y, float z) {
            const float sx = x - cam.o[0], sy = y - cam.o[1], sz = z - cam.o[2];
            return sx * cam.R[2] + sy * cam.R[5] + sz * cam.R[8];
        };
        const float z_unfilled = zc(0.f, 0.f, 0.f);
        float4 p = smp_loc[off];
        float cm = zc(p.x, p.y, p.z);
        float T = 1.0f, cr_ = 0.f, cg = 0.f, cb = 0.f, acc = 0.f;
        for (int i = 0; i < cnt; ++i) {
            float delta;
            if (i == opts.SR - 1) {
                delta = vs;
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
            const int v = w.s2v[off + i];
            const float sigma = v >= 0 ? w.sig[v] : 0.f;
            const float opacity = 1.0f - expf(-sigma * delta);
            const float wi = opacity * T;
            w.tmpT[off + i] = T;
            w.tmpD[off + i] = delta;
            T = T * (1.0f - opacity + 1e-10f);
            if (v >= 0) {
                const float4 sg = w.sg[v];
                cr_ += wi * (sg.x * 1.002f - 0.001f);
                cg += wi * (sg.y * 1.002f - 0.001f);
                cb += wi * (sg.z * 1.002f - 0.001f);
            }
            acc += wi;
        }
        o0 = cr_ + opts.bg[0] * (1.0f - acc);
        o1 = cg + opts.bg[1] * (1.0f - acc);
        o2 = cb + opts.bg[2] * (1.0f - acc);
        float g0 = g_rgb[3 * r], g1 = g_rgb[3 * r + 1], g2 = g_rgb[3 * r + 2];
        if (opts.eval_clamp) {  // torch.clamp passes the gradient inside [min, max]
            g0 = (o0 < 0.f || o0 > 1.f) ? 0.f : g0;
            g1 = (o1 < 0.f || o1 > 1.f) ? 0.f : g1;
            g2 = (o2 < 0.f || o2 > 1.f) ? 0.f : g2;
            o0 = fminf(fmaxf(o0, 0.f), 1.f);
            o1 = fminf(fmaxf(o1, 0.f), 1.f);
            o2 = fminf(fmaxf(o2, 0.f), 1.f);
        }
        float GT = 0.f;  // gradient with respect to the transmittance behind sample i
        for (int i = cnt - 1; i >= 0; --i) {
            const int v = w.s2v[off + i];
            const float Ti = w.tmpT[off + i], delta = w.tmpD[off + i];
            const float sigma = v >= 0 ? w.sig[v] : 0.f;
            const float ex = expf(-sigma * delta);
            const float opacity = 1.0f - ex;
            float c_r = 0.f, c_g = 0.f, c_b = 0.f;
            if (v >= 0) {
                const float4 sg = w.sg[v];
                c_r = sg.x * 1.002f - 0.001f;
                c_g = sg.y * 1.002f - 0.001f;
                c_b = sg.z * 1.002f - 0.001f;
            }
            const float gw = g0 * (c_r - opts.bg[0]) + g1 * (c_g - opts.bg[1]) + g2 * (c_b - opts.bg[2]);
            const float go = gw * Ti - GT * Ti;
            GT = gw * opacity + GT * (1.0f - opacity + 1e-10f);
            if (v >= 0) {
                const float wi = opacity * Ti;
                w.d_out[v] = make_float4(go * delta * ex, wi * g0, wi * g1, wi * g2);
            }
        }
    }
    if (rgb_out) {
        rgb_out[3 * r] = o0;
        rgb_out[3 * r + 1] = o1;
        rgb_out[3 * r + 2] = o2;
    }
}

// The same for SMALL batches (the 4096 rays of a training step: 53 us of one-thread-per-ray serial loops over up to SR
// samples, two to four dependent global loads per iteration, on sixteen workgroups): a WAVEFRONT takes a ray, its lanes
// fetch the ray's samples together into LDS (camera-space z, valid index, density, sigmoids), then lane 0 runs the very
// loops of k_train_composite_bwd over them -- the same expressions in the same order, the same bits.
constexpr int CBW_MAXS = 128;
__global__ void __launch_bounds__(256) k_train_composite_bwd_wave(CamRef cr, pnr_render_opts_t opts, int64_t R,
                                                                  const int *__restrict__ ray_cnt,
                                                                  const int *__restrict__ ray_off,
                                                                  const int *__restrict__ ray_flag,
                                                                  const float4 *__restrict__ smp_loc,
                                                                  const int *__restrict__ n_sel, TrainWs w,
                                                                  const float *__restrict__ g_rgb,
                                                                  float *__restrict__ rgb_out)
{
    __shared__ float4 s_sg[4][CBW_MAXS];
    __shared__ float s_z[4][CBW_MAXS], s_sig[4][CBW_MAXS], s_T[4][CBW_MAXS], s_D[4][CBW_MAXS];
    __shared__ int s_v[4][CBW_MAXS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    const bool exists = r < R;                    // wave-uniform; no early return before the barrier
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
        cam = load_cam(cr, cam_id(cr, r));
        for (int i = lane; i < cnt; i += 64) {
            const float4 p = smp_loc[off + i];
            s_z[wave][i] = zc(p.x, p.y, p.z);
            const int v = w.s2v[off + i];
            s_v[wave][i] = v;
            s_sig[wave][i] = v >= 0 ? w.sig[v] : 0.f;
            s_sg[wave][i] = v >= 0 ? w.sg[v] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    if (!exists || lane != 0) return;
    float o0 = opts.bg[0], o1 = opts.bg[1], o2 = opts.bg[2];
    if (keep) {
        const float vs = opts.vsize_z, two_vs = 2.0f * vs;
        const float z_unfilled = zc(0.f, 0.f, 0.f);
        float cm = s_z[wave][0];
        float T = 1.0f, cr_ = 0.f, cg = 0.f, cb = 0.f, acc = 0.f;
        for (int i = 0; i < cnt; ++i) {
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
            const int v = s_v[wave][i];
            const float sigma = s_sig[wave][i];
            const float opacity = 1.0f - expf(-sigma * delta);
            const float wi = opacity * T;
            s_T[wave][i] = T;
            s_D[wave][i] = delta;
            T = T * (1.0f - opacity + 1e-10f);
            if (v >= 0) {
                const float4 sg = s_sg[wave][i];
                cr_ += wi * (sg.x * 1.002f - 0.001f);
                cg += wi * (sg.y * 1.002f - 0.001f);
                cb += wi * (sg.z * 1.002f - 0.001f);
            }
            acc += wi;
        }
        o0 = cr_ + opts.bg[0] * (1.0f - acc);
        o1 = cg + opts.bg[1] * (1.0f - acc);
        o2 = cb + opts.bg[2] * (1.0f - acc);
        float g0 = g_rgb[3 * r], g1 = g_rgb[3 * r + 1], g2 = g_rgb[3 * r + 2];
        if (opts.eval_clamp) {
            g0 = (o0 < 0.f || o0 > 1.f) ? 0.f : g0;
            g1 = (o1 < 0.f || o1 > 1.f) ? 0.f : g1;
            g2 = (o2 < 0.f || o2 > 1.f) ? 0.f : g2;
            o0 = fminf(fmaxf(o0, 0.f), 1.f);
            o1 = fminf(fmaxf(o1, 0.f), 1.f);
            o2 = fminf(fmaxf(o2, 0.f), 1.f);
        }
        float GT = 0.f;
        for (int i = cnt - 1; i >= 0; --i) {
            const int v = s_v[wave][i];
            const float Ti = s_T[wave][i], delta = s_D[wave][i];
            const float sigma = s_sig[wave][i];
            const float ex = expf(-sigma * delta);
            const float opacity = 1.0f - ex;
            float c_r = 0.f, c_g = 0.f, c_b = 0.f;
            if (v >= 0) {
                const float4 sg = s_sg[wave][i];
                c_r = sg.x * 1.002f - 0.001f;
                c_g = sg.y * 1.002f - 0.001f;
                c_b = sg.z * 1.002f - 0.001f;
            }
            const float gw = g0 * (c_r - opts.bg[0]) + g1 * (c_g - opts.bg[1]) + g2 * (c_b - opts.bg[2]);
            const float go = gw * Ti - GT * Ti;
            GT = gw * opacity + GT * (1.0f - opacity + 1e-10f);
            if (v >= 0) {
                const float wi = opacity * Ti;
                w.d_out[v] = make_float4(go * delta * ex, wi * g0, wi * g1, wi * g2);
            }
        }
    }
    if (rgb_out) {
        rgb_out[3 * r] = o0;
        rgb_out[3 * r + 1] = o1;
        rgb_out[3 * r + 2] = o2;
    }
}

// colour head backwards: dz8 = d rgb * 1.002 * sg (1 - sg); dW8 += dz8 (x) C3; db8 += dz8;
// dz7 <- (dz8 . W8) * LeakyReLU'(C3)   (the gradient at the colour MLP's last pre-activation; dz7 == C3: in place -- the
// bf16x3 recompute chain -- or a buffer of its own, which leaves a render's tape intact)
__global__ void __launch_bounds__(256) k_train_color_head_bwd(TrainWs w, const float *__restrict__ w8, float *dz7)
{
    const int lane = threadIdx.x & 63;
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwv = (gridDim.x * blockDim.x) >> 6;
    const int S = w.cnt[1];
    float dw[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, db[3] = {0.f, 0.f, 0.f};
    float wr[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        wr[c][0] = w8[c * 128 + lane];
        wr[c][1] = w8[c * 128 + 64 + lane];
    }
    for (int v = wv; v < S; v += nwv) {
        const float4 sg = w.sg[v], go = w.d_out[v];
        const float dz[3] = {go.y * 1.002f * sg.x * (1.0f - sg.x), go.z * 1.002f * sg.y * (1.0f - sg.y),
                             go.w * 1.002f * sg.z * (1.0f - sg.z)};
        const float *c3 = w.C3 + (int64_t)v * LD_C;
        float *d7 = dz7 + (int64_t)v * LD_C;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float a = c3[64 * q + lane];
            float g = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dw[c][q] += dz[c] * a;
                g += dz[c] * wr[c][q];
            }
            d7[64 * q + lane] = g * (a > 0.f ? 1.0f : 0.1f);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) db[c] += dz[c];
    }
    // one partial row [3 x 128 weights | 3 biases] per WAVE, summed by k_reduce_rows (atomics: 1024 adds per address)
    float *dst = w.part + (int64_t)wv * 388;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dst[c * 128 + lane] = dw[c][0];
        dst[c * 128 + 64 + lane] = dw[c][1];
        if (lane == 0) dst[384 + c] = db[c];
    }
}

// dstA[c] += sum_b part[b * stride + c] for c < na;  dstB[c - na] likewise for na <= c < n.
// A block owns 16 columns; its 16 row groups sum every 16th row each (four independent partial sums), LDS combines the
// groups in a fixed order: the same expression on every run (bitwise repeatable), 64 dependent loads per thread instead
// of the 1024 a column-per-thread loop needs (73 -> ~10 us at the training batch size).
__global__ void __launch_bounds__(256) k_reduce_rows(const float *__restrict__ part, int nrows, int stride, int n, int na,
                                                     float *__restrict__ dstA, float *__restrict__ dstB)
{
    __shared__ float sm[16][17];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < n) {
        int b = g;
        for (; b + 48 < nrows; b += 64) {
            a0 += part[(int64_t)(b + 0) * stride + c];
            a1 += part[(int64_t)(b + 16) * stride + c];
            a2 += part[(int64_t)(b + 32) * stride + c];
            a3 += part[(int64_t)(b + 48) * stride + c];
        }
        for (; b < nrows; b += 16) a0 += part[(int64_t)b * stride + c];
    }
    sm[g][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && c < n) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sm[i][cl];
        if (c < na)
            dstA[c] += t;
        else
            dstB[c - na] += t;
    }
}

// density head + K-aggregation backwards (dAGG arrives in XC[:, 0:256], d sigma in d_out.x):
//   dG2 = w (dAGG + d sigma * [z > 0] * w4);  G2 <- dG2 * LeakyReLU'(G2) (in place);  dw4, db4 accumulated
template <bool WRITE>
__global__ void __launch_bounds__(256) k_train_head_agg_bwd(TrainWs w, int K, const float *__restrict__ w4)
{
    const int lane = threadIdx.x & 63;
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwv = (gridDim.x * blockDim.x) >> 6;
    const int S = w.cnt[1];
    const float4 wq = *reinterpret_cast<const float4 *>(w4 + 4 * lane);
    float4 dw = make_float4(0.f, 0.f, 0.f, 0.f);
    float db = 0.f;
    for (int v = wv; v < S; v += nwv) {
        const float4 da = *reinterpret_cast<const float4 *>(w.XC + (int64_t)v * LD_XC + 4 * lane);
        const float dsig = w.d_out[v].x;
        for (int k = 0; k < K; ++k) {
            const int row = v * K + k;
            float4 *gp = reinterpret_cast<float4 *>(w.G2 + (int64_t)row * LD_H + 4 * lane);
            const float4 gq = *gp;
            const float wk = w.row_w[row];
            const float coef = w.row_z[row] > 0.f ? wk * dsig : 0.f;
            dw.x += coef * gq.x;
            dw.y += coef * gq.y;
            dw.z += coef * gq.z;
            dw.w += coef * gq.w;
            db += coef;
            if (WRITE) {   // (exact mode: k_train_pairs_bwd forms dZ4 in registers, G2 stays the tape)
                float4 o;
                o.x = (wk * da.x + coef * wq.x) * (gq.x > 0.f ? 1.0f : 0.1f);
                o.y = (wk * da.y + coef * wq.y) * (gq.y > 0.f ? 1.0f : 0.1f);
                o.z = (wk * da.z + coef * wq.z) * (gq.z > 0.f ? 1.0f : 0.1f);
                o.w = (wk * da.w + coef * wq.w) * (gq.w > 0.f ? 1.0f : 0.1f);
                *gp = o;
            }
        }
    }
    // per-workgroup partial rows, summed by k_reduce_head (8192 waves x 257 atomics on 257 addresses took 0.8 ms whatever
    // the batch; one set of atomics per workgroup still 1024 adds per address)
    __shared__ float4 red[4][64];
    __shared__ float redb[4];
    const int wave = threadIdx.x >> 6;
    red[wave][lane] = dw;
    if (lane == 0) redb[wave] = db;
    __syncthreads();
    if (wave == 0) {
        float4 t = red[0][lane];
        float tb = redb[0];
        for (int q = 1; q < 4; ++q) {
            t.x += red[q][lane].x;
            t.y += red[q][lane].y;
            t.z += red[q][lane].z;
            t.w += red[q][lane].w;
            tb += redb[q];
        }
        float *dst = w.part + (int64_t)blockIdx.x * 260;
        *reinterpret_cast<float4 *>(dst + 4 * lane) = t;
        if (lane == 0) dst[256] = tb;
    }
}

// Gradients of the point tensors (index_select backward), WITHOUT float atomics: a point is the neighbour of ~7 samples
// of a batch, and the order in which atomics would add its rows changes from run to run.  Instead:
//   k_train_rowgrad    one wavefront per row: the row's 38 gradient values (d embedding 32 | d colour 3 | d dir 3) into
//                      rowgrad [rows, 40]; counts the rows of every touched point (integer atomics: order-free)
//   (scan)             pt_start = exclusive scan of the counts, over the U touched points of the call (their rank in
//                      the render's pt_list, ascending point index)
//   k_train_fill       groups the row ids by point (slot order inside a group is arbitrary here ...)
//   k_train_point_sum  ... one wavefront per touched point SORTS its group and adds its rows in ascending row order:
//                      the sum is a fixed expression, bitwise repeatable.  Writes either the dense tensors (+=, rows of
//                      touched points only) or one compact row per touched point (sparse emission).
//   dX0 [row, 0:224] is in G2, the taped encodings in X0, d[color | sdir - view | <sdir, view>] in H2[:, 256:263]
constexpr int LD_RG = 40;
__global__ void __launch_bounds__(256) k_train_rowgrad(TrainParams P, TrainWs w, const int *__restrict__ pt_rank,
                                                       float *__restrict__ rowgrad)
{
    const int lane = threadIdx.x & 63;
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwv = (gridDim.x * blockDim.x) >> 6;
    const int rows = w.cnt[0], K = P.K;
    __shared__ float sbuf[4][2][224];
    for (int row = wv; row < rows; row += nwv) {
        const int pidx = w.row_pidx[row];
        if (pidx < 0) continue;
        if (lane == 0) atomicAdd(&w.pt_cnt[pt_rank[pidx]], 1);
        const float *dh = w.H2 + (int64_t)row * LD_H2 + 256;
        // the row of dX0 (224 floats) and of the taped encodings (columns 0..223 of X0) through wave-private LDS: two
        // coalesced float4 loads per lane instead of thirteen 4-byte loads at a stride of 24 bytes
        float *dx = &sbuf[threadIdx.x >> 6][0][0], *x0 = &sbuf[threadIdx.x >> 6][1][0];
        __builtin_amdgcn_wave_barrier();
        if (lane < 56) {
            reinterpret_cast<float4 *>(dx)[lane] = reinterpret_cast<const float4 *>(w.G2 + (int64_t)row * LD_H)[lane];
            reinterpret_cast<float4 *>(x0)[lane] = reinterpret_cast<const float4 *>(w.X0 + (int64_t)row * LD_X0)[lane];
        }
        __builtin_amdgcn_wave_barrier();
        float g = 0.f;
        if (lane < 32) {
            // d/de [e, sin(e 2^f), cos(e 2^f)] = [1, 2^f cos, -2^f sin]
            g = dx[lane];
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const int i = 32 + 2 * (3 * lane + f);
                g += (float)(1 << f) * (x0[i + 1] * dx[i] - x0[i] * dx[i + 1]);
            }
        } else if (lane < 35) {
            g = dh[lane - 32];
        } else if (lane < 38) {
            const int jd = lane - 35;
            const int ray = P.smp_ray[P.vs_list[row / K]];
            float vx, vy, vz;
            rot_rows(P.Rw2c, P.dirs[3 * (int64_t)ray], P.dirs[3 * (int64_t)ray + 1], P.dirs[3 * (int64_t)ray + 2], vx, vy,
                     vz);
            // sdir = dir @ Rw2c^T; the head sees sdir - view and <sdir, view>
            const float gd = dh[6];
            const float gs0 = dh[3] + gd * vx, gs1 = dh[4] + gd * vy, gs2 = dh[5] + gd * vz;
            // sdir[i] = sum_j dir[j] M[i][j]  =>  d dir[j] = sum_i d sdir[i] M[i][j]
            g = gs0 * P.Rw2c[jd] + gs1 * P.Rw2c[3 + jd] + gs2 * P.Rw2c[6 + jd];
        }
        if (lane < LD_RG) rowgrad[(int64_t)row * LD_RG + lane] = lane < 38 ? g : 0.f;
    }
}

__global__ void __launch_bounds__(256) k_train_fill(TrainWs w, const int *__restrict__ pt_rank)
{
    const int rows = w.cnt[0];
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += gridDim.x * blockDim.x) {
        const int pidx = w.row_pidx[row];
        if (pidx < 0) continue;
        const int u = pt_rank[pidx];
        w.pt_rows[w.pt_start[u] + atomicAdd(&w.pt_cursor[u], 1)] = row;
    }
}

__global__ void __launch_bounds__(256) k_train_point_sum(TrainWs w, const int *__restrict__ n_sel,
                                                         const int *__restrict__ pt_list,
                                                         const float *__restrict__ rowgrad, float *__restrict__ d_emb,
                                                         float *__restrict__ d_color, float *__restrict__ d_dir,
                                                         float *__restrict__ sp_rows, int *__restrict__ sp_index,
                                                         long long sp_cap)
{
    const int lane = threadIdx.x & 63;
    const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwv = (gridDim.x * blockDim.x) >> 6;
    const int U = n_sel[3];
    auto emit = [&](int u, float acc) {
        const int pidx = pt_list[u];
        if (sp_rows) {
            if (u < sp_cap) {
                if (lane < LD_RG) sp_rows[(int64_t)u * LD_RG + lane] = lane < 38 ? acc : 0.f;
                if (lane == 0) sp_index[u] = pidx;
            }
        } else if (lane < 32) {
            if (d_emb) d_emb[(int64_t)pidx * 32 + lane] += acc;
        } else if (lane < 35) {
            if (d_color) d_color[(int64_t)pidx * 3 + (lane - 32)] += acc;
        } else if (lane < 38) {
            if (d_dir) d_dir[(int64_t)pidx * 3 + (lane - 35)] += acc;
        }
    };
    // a point shared by more than 64 rows (rare): repeated selection of the next larger row id
    auto crowded = [&](int b, int c) {
        float acc = 0.f;
        int prev = -1;
        for (int j = 0; j < c; ++j) {
            int best = 0x7FFFFFFF;
            for (int i = lane; i < c; i += 64) {
                const int r = w.pt_rows[b + i];
                if (r > prev && r < best) best = r;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) best = min(best, __shfl_xor(best, d, 64));
            prev = best;
            if (lane < 38) acc += rowgrad[(int64_t)best * LD_RG + lane];
        }
        return acc;
    };
    // Two points per iteration: a point's work is a chain of four dependent memory latencies (group bounds -> row ids ->
    // gradient rows -> the accumulated tensor) and a handful of rows, so a wavefront that walks its points one by one
    // mostly waits (0.39 ms at 65 536 rays with every wave slot of the device taken).  The two chains are independent;
    // each point's rows are still added in ascending row order (the same expression, bitwise repeatable).
    for (int u0 = wv; u0 < U; u0 += 2 * nwv) {
        const int u1 = u0 + nwv;
        const bool ok1 = u1 < U;
        const int b0 = w.pt_start[u0], c0 = w.pt_start[u0 + 1] - b0;
        const int b1 = ok1 ? w.pt_start[u1] : 0, c1 = ok1 ? w.pt_start[u1 + 1] - b1 : 0;
        if (c0 <= 64 && c1 <= 64) {
            // rank sort inside the wavefront (row ids are distinct), then the rows in ascending order
            const int m0 = lane < c0 ? w.pt_rows[b0 + lane] : 0x7FFFFFFF;
            const int m1 = lane < c1 ? w.pt_rows[b1 + lane] : 0x7FFFFFFF;
            int r0 = 0, r1 = 0;
            for (int j = 0; j < c0; ++j) r0 += __shfl(m0, j, 64) < m0 ? 1 : 0;
            for (int j = 0; j < c1; ++j) r1 += __shfl(m1, j, 64) < m1 ? 1 : 0;
            // lane `rank` must end up holding `mine`: a gather by inverse permutation through ds_bpermute's twin
            const int s0 = __builtin_amdgcn_ds_permute(r0 << 2, m0);
            const int s1 = __builtin_amdgcn_ds_permute(r1 << 2, m1);
            float a0 = 0.f, a1 = 0.f;
            const int cm = max(c0, c1);
            for (int j = 0; j < cm; ++j) {
                const int row0 = __shfl(s0, j, 64), row1 = __shfl(s1, j, 64);
                float g0 = 0.f, g1 = 0.f;
                if (lane < 38) {
                    if (j < c0) g0 = rowgrad[(int64_t)row0 * LD_RG + lane];
                    if (j < c1) g1 = rowgrad[(int64_t)row1 * LD_RG + lane];
                }
                if (j < c0) a0 += g0;
                if (j < c1) a1 += g1;
            }
            emit(u0, a0);
            if (ok1) emit(u1, a1);
        } else {
            emit(u0, crowded(b0, c0));
            if (ok1) emit(u1, crowded(b1, c1));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
constexpr int PERSISTENT_WGS = 256 * 2 * 4;  // 256 CUs x 2 resident workgroups x 4 rounds of slack for balance

template <bool TA, bool TB, int EPI>
static void gemm(hipStream_t st, const GemmArgs &g, int m_max, int nsplit = 1)
{
    const unsigned mt = (unsigned)((m_max + TM - 1) / TM), nt = (unsigned)((g.N + TN - 1) / TN);
    const dim3 grid(TA ? mt : std::min(mt * nt, (unsigned)PERSISTENT_WGS), TA ? nt : 1u, (unsigned)nsplit);
    hipLaunchKernelGGL((k_gemm<TA, TB, EPI>), grid, dim3(256), 0, st, g);
}

// C[rows, N] = leaky(A[rows, K] . W[N, K]^T + b)
template <int EPI>
static void gemm_bf(hipStream_t st, const GemmArgs &g, int m_max)
{
    const unsigned tiles = (unsigned)((g.N + TN - 1) / TN) * (unsigned)((m_max + TM - 1) / TM);
    const dim3 grid(std::min(tiles, (unsigned)PERSISTENT_WGS));
    const int nch = (g.K + BK - 1) / BK;
    if (nch == 4) {
        hipLaunchKernelGGL((k_gemm_nt_bf16x3_u<EPI, 4>), grid, dim3(256), 0, st, g);
        return;
    }
    if (nch == 8) {
        hipLaunchKernelGGL((k_gemm_nt_bf16x3_u<EPI, 8>), grid, dim3(256), 0, st, g);
        return;
    }
    if (nch == 9) {
        hipLaunchKernelGGL((k_gemm_nt_bf16x3_u<EPI, 9>), grid, dim3(256), 0, st, g);
        return;
    }
    hipLaunchKernelGGL((k_gemm_nt_bf16x3<EPI>), grid, dim3(256), 0, st, g);
}

static void gemm_forward(hipStream_t st, bool bf, const float *A, int lda, const float *W, int ldw, const float *b,
                         float *C, int ldc, int N, int K, const int *dev_rows, int64_t rows_max,
                         unsigned long long *sign_out = nullptr)
{
    GemmArgs g{};
    g.sign_out = sign_out;
    g.sign_nt = (N + TN - 1) / TN;
    g.A = A; g.B = W; g.C = C; g.lda = lda; g.ldb = ldw; g.ldc = ldc; g.M = 0; g.N = N; g.K = K;
    g.dev_rows = dev_rows; g.bias = b;
    if (bf)
        gemm_bf<EPI_BIAS_LEAKY>(st, g, (int)rows_max);
    else
        gemm<false, true, EPI_BIAS_LEAKY>(st, g, (int)rows_max);
}

// C[rows, N] = (dZ[rows, K] . W[K, N]) * leaky'(C) for columns < mask_cols (in place over the taped activation)
// (bf16x3: the B operand is WT = W^T [N, K] with leading dimension K)
static void gemm_data(hipStream_t st, bool bf, const float *dZ, int lda, const float *W, int ldw, const float *WT,
                      float *C, int ldc, int N, int K, int mask_cols, const int *dev_rows, int64_t rows_max,
                      const unsigned long long *sign_in = nullptr, const float *mask_src = nullptr)
{
    GemmArgs g{};
    g.sign_in = sign_in;
    g.sign_nt = (mask_cols + TN - 1) / TN;
    g.A = dZ; g.B = W; g.C = C; g.lda = lda; g.ldb = ldw; g.ldc = ldc; g.M = 0; g.N = N; g.K = K;
    g.dev_rows = dev_rows; g.mask = mask_src ? mask_src : C; g.mask_cols = mask_cols;
    if (bf) {
        g.B = WT;
        g.ldb = K;
        if (mask_cols > 0)
            gemm_bf<EPI_MASK>(st, g, (int)rows_max);
        else
            gemm_bf<EPI_STORE>(st, g, (int)rows_max);
        return;
    }
    if (mask_cols > 0)
        gemm<false, false, EPI_MASK>(st, g, (int)rows_max);
    else
        gemm<false, false, EPI_STORE>(st, g, (int)rows_max);
}

// dW[M, N] += dZ[rows, M]^T . X[rows, N];  db[M] += column sums of dZ
static void gemm_weight(hipStream_t st, bool bf, const float *dZ, int lda, const float *X, int ldx, float *dW, int ldw, int M,
                        int N, const int *dev_rows, int64_t rows_max, float *db, float *part)
{
    GemmArgs g{};
    g.A = dZ; g.B = X; g.C = part; g.lda = lda; g.ldb = ldx; g.ldc = ldw; g.M = M; g.N = N; g.K = 0;
    g.dev_rows = dev_rows; g.colsum = part + PART_FLOATS;
    const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    int nsplit = (int)std::min<int64_t>(std::max<int64_t>(1, MAX_SPLIT_WGS / tiles), (rows_max + 4 * TK - 1) / (4 * TK));
    nsplit = std::max(nsplit, 1);
    // the partial blocks are [M, ldw] each: nsplit * M * ldw <= PART_FLOATS by construction (ldw <= tiles-per-row * TN)
    while ((size_t)nsplit * M * ldw > PART_FLOATS && nsplit > 1) --nsplit;
    if (bf) {
        const dim3 grid((unsigned)((M + TM - 1) / TM), (unsigned)((N + TN - 1) / TN), (unsigned)nsplit);
        hipLaunchKernelGGL(k_gemm_tn_bf16x3, grid, dim3(256), 0, st, g);
    } else if (M == 256 && N % 128 != 0) {
        // mlp_head.0 (N = 264) and mlp_base.0 (N = 288): whole-M tiles of 96 columns (k_wgrad256<3>: 1.34 ms against the
        // 1.8 of three 128-column tiles at 65 536 rays; at N = 256 the 128 x 128 tiles of k_gemm are the faster ones, 1.22
        // against 1.50 for k_wgrad256<4>), about one round of 512 workgroups
        const int nt = (N + 95) / 96;
        nsplit = (int)std::min<int64_t>(std::max<int64_t>(1, 512 / nt), (rows_max + 4 * TK - 1) / (4 * TK));
        while ((size_t)nsplit * M * ldw > PART_FLOATS && nsplit > 1) --nsplit;
        hipLaunchKernelGGL(k_wgrad256<3>, dim3(1u, (unsigned)nt, (unsigned)nsplit), dim3(256), 0, st, g);
    } else {
        gemm<true, false, EPI_PARTIAL>(st, g, M, nsplit);
    }
    hipLaunchKernelGGL(k_reduce_parts, dim3(288), dim3(256), 0, st, part, part + PART_FLOATS, nsplit, bf ? BK : TK, M, N,
                       ldw, dev_rows, dW, db);
}

// the weight-gradient GEMMs of a step, collected and launched together (k_wgrad_batch128 / k_wgrad_batch96 +
// k_reduce_parts_batch); exact mode only
struct WgradQueue {
    WgradBatch a{}, b{};
    ReduceBatch r{};
    int nr = 0;
    float *pool;
    size_t pool_floats, used = 0;
    WgradQueue(float *p, size_t n) : pool(p), pool_floats(n) {}
    // dW[M, N] += dZ[rows, M]^T . X[rows, N];  db[M] += column sums of dZ
    bool overflow = false;   // the pool cannot hold even one split of a job (never with POOL_FLOATS; checked by the caller)
    void add(const float *dZ, int lda, const float *X, int ldx, float *dW, int ldw, int M, int N, const int *dev_rows,
             int64_t rows_max, float *db)
    {
        const bool narrow = M == 256 && N % 128 != 0;
        WgradBatch &q = narrow ? b : a;
        const int tx = narrow ? 1 : (M + TM - 1) / TM, ty = narrow ? (N + 95) / 96 : (N + TN - 1) / TN;
        const int budget = narrow ? 512 : (M == 256 ? MAX_SPLIT_WGS : 256);
        int nsplit = (int)std::min<int64_t>(std::max<int64_t>(1, budget / (tx * ty)), (rows_max + 4 * TK - 1) / (4 * TK));
        const size_t per = (size_t)M * ldw + M;   // a split's partial tile and bias row
        const size_t left = pool_floats - used;
        if (left < per || (narrow ? b.n : a.n) >= MAX_WGRAD_JOBS) {
            overflow = true;
            return;
        }
        nsplit = (int)std::max<size_t>(1, std::min<size_t>((size_t)nsplit, left / per));
        float *part = pool + used, *csum = part + (size_t)nsplit * M * ldw;
        used += (size_t)nsplit * per;
        const int j = q.n++;
        GemmArgs &g = q.g[j];
        g = GemmArgs{};
        g.A = dZ; g.B = X; g.C = part; g.lda = lda; g.ldb = ldx; g.ldc = ldw; g.M = M; g.N = N; g.K = 0;
        g.dev_rows = dev_rows; g.colsum = csum;
        q.tx[j] = tx;
        q.ty[j] = ty;
        q.nz[j] = nsplit;
        q.wg0[j + 1] = q.wg0[j] + tx * ty * nsplit;
        r.part[nr] = part; r.csum[nr] = csum; r.rows[nr] = dev_rows; r.dst[nr] = dW; r.db[nr] = db;
        r.nz[nr] = nsplit; r.M[nr] = M; r.N[nr] = N; r.ld[nr] = ldw;
        ++nr;
    }
    void launch(hipStream_t st)
    {
        if (a.n) hipLaunchKernelGGL(k_wgrad_batch128, dim3((unsigned)a.wg0[a.n]), dim3(256), 0, st, a);
        if (b.n) hipLaunchKernelGGL(k_wgrad_batch96, dim3((unsigned)b.wg0[b.n]), dim3(256), 0, st, b);
        if (nr) hipLaunchKernelGGL(k_reduce_parts_batch, dim3(288, (unsigned)nr), dim3(256), 0, st, r);
    }
};

}  // namespace pnr

using namespace pnr;

namespace pnr {
__global__ void __launch_bounds__(256) k_clear_point_rows(float *__restrict__ d_emb, float *__restrict__ d_color,
                                                          float *__restrict__ d_dir, int64_t N,
                                                          const int *__restrict__ index, int64_t n_index,
                                                          const long long *__restrict__ n_dev)
{
    int64_t n = n_index;
    if (n_dev) n = min((int64_t)*n_dev, n_index);
    // 38 floats per listed point: one thread per (entry, float)
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n * 64; t += (int64_t)gridDim.x * 256) {
        const int64_t e = t >> 6;
        const int c = (int)(t & 63);
        const int64_t i = index[e];
        if (i < 0 || i >= N || c >= 38) continue;
        if (c < 32) {
            if (d_emb) d_emb[i * 32 + c] = 0.f;
        } else if (c < 35) {
            if (d_color) d_color[i * 3 + (c - 32)] = 0.f;
        } else if (d_dir) {
            d_dir[i * 3 + (c - 35)] = 0.f;
        }
    }
}
}  // namespace pnr

extern "C" int pnr_point_grads_clear(float *d_embedding, float *d_color, float *d_dir, int64_t N, const int32_t *d_index,
                                     int64_t n_index, const int64_t *d_n_index, void *stream_)
{
    PNR_REQUIRE(d_index != nullptr && N >= 1, "pnr_point_grads_clear: null index / N=%lld", (long long)N);
    PNR_REQUIRE(n_index >= 0 && n_index < (int64_t)0x7FFFFFFF, "pnr_point_grads_clear: n_index=%lld out of range",
                (long long)n_index);
    if (n_index == 0) return PNR_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((n_index * 64 + 255) / 256, 2048);
    hipLaunchKernelGGL(pnr::k_clear_point_rows, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, d_embedding, d_color,
                       d_dir, N, d_index, n_index, reinterpret_cast<const long long *>(d_n_index));
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

// ------------------------------------------------------------------------------------------------
// The confidence regulariser (studio_model.py:288-292,427-429) over the neighbour slots of a render.
// term(c) = log v + log(1 - v), v = clamp(clamp(c, 1e-4, 1), eps, 1 - eps); d term / d c = 1 / v - 1 / (1 - v) where
// neither clamp of v is active (the inner clamp passes its gradient straight through).
// ------------------------------------------------------------------------------------------------
namespace pnr {
__device__ __forceinline__ float conf_term(float c, float eps, float &dterm)
{
    const float cc = fminf(fmaxf(c, 0.0001f), 1.0f);
    const float v = fminf(fmaxf(cc, eps), 1.0f - eps);
    dterm = (cc > eps && cc < 1.0f - eps) ? (1.0f / v - 1.0f / (1.0f - v)) : 0.f;
    return logf(v) + logf(1.0f - v);
}

constexpr int CONF_BLOCKS = 512;   // partial sums (double) + filled-slot counts, one per block

// one thread per (selected sample, slot): partial sums per block in a FIXED tree, so that the total is repeatable
__global__ void __launch_bounds__(256) k_conf_partial(const int *__restrict__ n_sel, int K, const int *__restrict__ smp_pidx,
                                                      const float *__restrict__ conf, float eps,
                                                      double *__restrict__ part, long long *__restrict__ filled)
{
    __shared__ double sh[256];
    __shared__ int shn[256];
    const long long n = (long long)n_sel[0] * K;
    double acc = 0.0;
    int cnt = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int p = smp_pidx[i];
        if (p >= 0) {
            float d;
            acc += (double)conf_term(conf[p], eps, d);
            ++cnt;
        }
    }
    sh[threadIdx.x] = acc;
    shn[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[threadIdx.x] += sh[threadIdx.x + o];
            shn[threadIdx.x] += shn[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[blockIdx.x] = sh[0];
        filled[blockIdx.x] = shn[0];
    }
}

// out[0] = mean over the reference's [1, R'', SR, K] tensor (unfilled slots read point 0), out[1] = its element count
__global__ void k_conf_final(const double *__restrict__ part, const long long *__restrict__ filled, int nblocks,
                             const unsigned long long *__restrict__ shards, int SR, int K,
                             const float *__restrict__ conf, float eps, float *__restrict__ out,
                             long long *__restrict__ slots_out)
{
    // one wavefront: lane l adds the partials l, l + 64, ... in that order, a butterfly adds the lanes (every lane ends
    // with the same bits: a fixed expression) -- one thread walking 512 dependent loads took 36 us
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    double s = 0.0;
    long long nf = 0;
    for (int i = lane; i < nblocks; i += 64) {
        s += part[i];
        nf += filled[i];
    }
    unsigned long long kept = lane < SHARDS ? shards[((size_t)SH_KEPT * SHARDS + lane) * SHARD_STRIDE] : 0ull;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        nf += __shfl_xor(nf, o, 64);
        kept += __shfl_xor(kept, o, 64);
    }
    if (lane != 0) return;
    const long long slots = (long long)kept * SR * K;
    float d;
    s += (double)(slots - nf) * (double)conf_term(conf[0], eps, d);
    out[0] = (float)(s / (double)slots);     // (no kept ray: 0 / 0 = NaN, as torch.mean of an empty tensor)
    out[1] = (float)slots;
    *slots_out = slots;                      // exact, for the backward
}

// d_conf[p] += upstream / slots * dterm(conf[p]) per filled slot.  The addends of one point are IDENTICAL floats, so the
// order in which the atomics land does not matter: repeatable bits.  Point 0 also stands for every unfilled slot: its
// slots are only counted here and added once, by k_conf_bwd_zero.
__global__ void __launch_bounds__(256) k_conf_bwd(const int *__restrict__ n_sel, int K, const int *__restrict__ smp_pidx,
                                                  const float *__restrict__ conf, float eps,
                                                  const float *__restrict__ upstream, const long long *__restrict__ slots,
                                                  float *__restrict__ d_conf, unsigned long long *__restrict__ zero_cnt)
{
    const long long n = (long long)n_sel[0] * K;
    const float g = (float)((double)upstream[0] / (double)*slots);
    unsigned long long mine = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int p = smp_pidx[i];
        if (p > 0) {
            float d;
            conf_term(conf[p], eps, d);
            if (d != 0.f) unsafeAtomicAdd(&d_conf[p], g * d);
        } else if (p == 0) {
            ++mine;
        }
    }
    if (mine) atomicAdd(zero_cnt, mine);
}
__global__ void k_conf_bwd_zero(const double *__restrict__ part, const long long *__restrict__ filled, int nblocks,
                                const float *__restrict__ conf, float eps, const float *__restrict__ upstream,
                                const long long *__restrict__ slots_in, float *__restrict__ d_conf,
                                unsigned long long *__restrict__ zero_cnt)
{
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    long long nf = 0;
    for (int i = lane; i < nblocks; i += 64) nf += filled[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) nf += __shfl_xor(nf, o, 64);
    if (lane != 0) return;
    const long long slots = *slots_in;
    float d;
    conf_term(conf[0], eps, d);
    const float g = (float)((double)upstream[0] / (double)slots);
    d_conf[0] += (float)((double)((slots - nf) + (long long)*zero_cnt) * (double)g * (double)d);
    *zero_cnt = 0;
}
}  // namespace pnr

extern "C" size_t pnr_conf_loss_workspace_bytes(void) { return (size_t)pnr::CONF_BLOCKS * 16 + 64; }   // + zero count, slots

static int conf_common(const char *who, const pnr_scene_t *scene, const pnr_render_opts_t *opts, int64_t R,
                       void *d_render_workspace, size_t render_workspace_bytes, int64_t cap_samples, const float *d_conf,
                       void *d_scratch, RenderWs &ws)
{
    PNR_REQUIRE(scene && opts && d_render_workspace && d_conf && d_scratch, "%s: null argument", who);
    if (!scene->built) {
        set_error("%s: scene not built", who);
        return PNR_ERR_STATE;
    }
    const size_t need = pnr_render_workspace_bytes_for(scene, opts, R, cap_samples);
    if (render_workspace_bytes < need) {
        set_error("%s: workspace of %zu bytes < %zu of the render it follows", who, render_workspace_bytes, need);
        return PNR_ERR_WORKSPACE;
    }
    ws = carve_render_ws(d_render_workspace, R, cap_samples, opts->K, scene->N, scene->info[2]);
    return PNR_OK;
}

extern "C" int pnr_conf_loss(const pnr_scene_t *scene, const pnr_render_opts_t *opts, int64_t R, void *d_render_workspace,
                             size_t render_workspace_bytes, int64_t cap_samples, const float *d_conf, float eps,
                             void *d_scratch, float *d_out, void *stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    RenderWs ws{};
    const int rc = conf_common("pnr_conf_loss", scene, opts, R, d_render_workspace, render_workspace_bytes, cap_samples,
                               d_conf, d_scratch, ws);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(d_out != nullptr && eps >= 0.f && eps < 0.5f, "pnr_conf_loss: d_out null or eps=%g", eps);
    double *part = (double *)d_scratch;
    long long *filled = (long long *)(part + CONF_BLOCKS);
    hipLaunchKernelGGL(k_conf_partial, dim3(CONF_BLOCKS), dim3(256), 0, st, ws.n_sel, opts->K, ws.smp_pidx, d_conf, eps, part,
                       filled);
    hipLaunchKernelGGL(k_conf_final, dim3(1), dim3(64), 0, st, part, filled, CONF_BLOCKS, ws.shards, opts->SR, opts->K,
                       d_conf, eps, d_out, filled + CONF_BLOCKS + 1);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

extern "C" int pnr_conf_loss_backward(const pnr_scene_t *scene, const pnr_render_opts_t *opts, int64_t R,
                                      void *d_render_workspace, size_t render_workspace_bytes, int64_t cap_samples,
                                      const float *d_conf, float eps, void *d_scratch, const float *d_fwd_out,
                                      const float *d_upstream, float *d_grad_conf, void *stream_)
{
    hipStream_t st = (hipStream_t)stream_;
    RenderWs ws{};
    const int rc = conf_common("pnr_conf_loss_backward", scene, opts, R, d_render_workspace, render_workspace_bytes,
                               cap_samples, d_conf, d_scratch, ws);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(d_fwd_out && d_upstream && d_grad_conf, "pnr_conf_loss_backward: null argument");
    double *part = (double *)d_scratch;
    long long *filled = (long long *)(part + CONF_BLOCKS);
    unsigned long long *zero_cnt = (unsigned long long *)(filled + CONF_BLOCKS);
    {
        const int rcz = zero_async(zero_cnt, 8, st);
        if (rcz != PNR_OK) return rcz;
    }
    const long long *slots = filled + CONF_BLOCKS + 1;
    hipLaunchKernelGGL(k_conf_bwd, dim3(CONF_BLOCKS), dim3(256), 0, st, ws.n_sel, opts->K, ws.smp_pidx, d_conf, eps, d_upstream,
                       slots, d_grad_conf, zero_cnt);
    hipLaunchKernelGGL(k_conf_bwd_zero, dim3(1), dim3(64), 0, st, part, filled, CONF_BLOCKS, d_conf, eps, d_upstream, slots,
                       d_grad_conf, zero_cnt);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

extern "C" size_t pnr_backward_workspace_bytes(int64_t cap_samples, int32_t K)
{
    if (cap_samples < 1) cap_samples = 1;
    if (K < 1) K = 1;
    return carve_train_ws(nullptr, cap_samples, K).total;
}

extern "C" int pnr_render_backward(const pnr_scene_t *scene, const pnr_weights_t *weights, const float *const d_w[9],
                                   const float *const d_b[9], const float *d_dirs, int64_t R, const pnr_camera_t *cams,
                                   int32_t n_cams, const int32_t *d_ray_cam, int64_t rays_per_cam,
                                   const pnr_render_opts_t *opts, const float *d_grad_rgb, void *d_render_workspace,
                                   size_t render_workspace_bytes, int64_t cap_samples, void *d_train_workspace,
                                   size_t train_workspace_bytes, const pnr_grads_t *grads, float *d_rgb_recomputed,
                                   void *stream_)
{
    const char *who = "pnr_render_backward";
    hipStream_t st = (hipStream_t)stream_;
    PNR_REQUIRE(scene && weights && d_w && d_b && d_dirs && opts && d_grad_rgb && d_render_workspace &&
                    d_train_workspace && grads,
                "%s: null argument", who);
    for (int i = 0; i < 9; ++i) PNR_REQUIRE(d_w[i] && d_b[i], "%s: null weight pointer %d", who, i);
    if (!scene->built || !scene->packed) {
        set_error("%s: scene not built / points not packed", who);
        return PNR_ERR_STATE;
    }
    PNR_REQUIRE(R >= 1 && R < (int64_t)0x7FFFFFF0, "%s: R=%lld out of range", who, (long long)R);
    PNR_REQUIRE(n_cams >= 1 && n_cams <= PNR_MAX_CAMS, "%s: n_cams=%d not in [1,%d]", who, n_cams, PNR_MAX_CAMS);
    PNR_REQUIRE(d_ray_cam != nullptr || (rays_per_cam >= 1 && rays_per_cam * n_cams >= R),
                "%s: rays_per_cam=%lld does not cover R=%lld rays with %d cameras", who, (long long)rays_per_cam,
                (long long)R, n_cams);
    PNR_REQUIRE(opts->K >= 1 && opts->K <= PNR_MAX_K, "%s: K=%d not in [1,%d]", who, opts->K, PNR_MAX_K);
    PNR_REQUIRE(opts->early_stop_eps == 0.f, "%s: early ray termination skips samples; train with early_stop_eps = 0",
                who);
    PNR_REQUIRE(cap_samples >= 1 && cap_samples * (int64_t)opts->K < (int64_t)0x7FFFFF00 / 8,
                "%s: cap_samples=%lld out of range", who, (long long)cap_samples);
    const size_t need_r = pnr_render_workspace_bytes_for(scene, opts, R, cap_samples);
    if (render_workspace_bytes < need_r) {
        set_error("%s: render workspace of %zu bytes < %zu: pass the workspace of the preceding pnr_render call", who,
                  render_workspace_bytes, need_r);
        return PNR_ERR_WORKSPACE;
    }
    const size_t need_t = pnr_backward_workspace_bytes(cap_samples, opts->K);
    if (train_workspace_bytes < need_t) {
        set_error("%s: training workspace of %zu bytes < %zu required (pnr_backward_workspace_bytes)", who,
                  train_workspace_bytes, need_t);
        return PNR_ERR_WORKSPACE;
    }
    const int K = opts->K;
    const bool bf = opts->precision == PNR_PRECISION_BF16X3;
    RenderWs ws = carve_render_ws(d_render_workspace, R, cap_samples, K, scene->N, scene->info[2]);
    TrainWs tw = carve_train_ws(d_train_workspace, cap_samples, K);

    CamRef cr{};
    cr.cams = ws.cams;
    cr.ray_cam = d_ray_cam;
    cr.rays_per_cam = d_ray_cam ? 1 : rays_per_cam;
    cr.tmid = nullptr;
    cr.D = opts->D;
    cr.n_cams = n_cams;
    cr.jitter = opts->jitter;
    cr.seed = opts->seed;
    if (cams) {   // (cams == NULL: the cameras the render left in its workspace -- after pnr_render_pose)
        CamSet set{};
        for (int c = 0; c < n_cams; ++c) {
            for (int i = 0; i < 3; ++i) set.c[c].o[i] = cams[c].campos[i];
            for (int i = 0; i < 9; ++i) set.c[c].R[i] = cams[c].camrotc2w[i];
        }
        hipLaunchKernelGGL(k_train_set_cams, dim3(1), dim3(64), 0, st, set, n_cams, ws.cams);
    }

    TrainParams P{};
    P.point_rows = reinterpret_cast<const float4 *>(scene->point_rows);
    for (int i = 0; i < 9; ++i) P.Rw2c[i] = weights->Rw2c[i];
    P.cr = cr;
    P.dirs = d_dirs;
    P.smp_loc = ws.smp_loc;
    P.smp_ray = ws.smp_ray;
    P.smp_pidx = ws.smp_pidx;
    P.vs_list = ws.vs_list;
    P.n_sel = ws.n_sel;
    P.K = K;

    const int64_t rows_max = cap_samples * K, smp_max = cap_samples;
    const int *n_rows = tw.cnt, *n_smp = tw.cnt + 1;
    const dim3 eg(2048), eb(256);

    hipLaunchKernelGGL(k_train_init, dim3(512), eb, 0, st, ws.n_sel, (int)cap_samples, K, tw);
    hipLaunchKernelGGL(k_train_s2v, dim3(256), eb, 0, st, ws.vs_list, tw.cnt, tw.s2v);
    {
        NineMats m{};
        for (int i = 0; i < 9; ++i) {
            m.src[i] = d_w[i];
            m.dst[i] = tw.Wp[i];
            m.dstT[i] = tw.WT[i];
            m.n_out[i] = W_OUT[i];
            m.n_in[i] = W_IN[i];
            m.ld[i] = W_LD[i];
        }
        hipLaunchKernelGGL(k_pad_weights, dim3(32, 9), eb, 0, st, m);
    }

    // ---- forward with tape -----------------------------------------------------------------------
    // the render filled H1, H2, G1, G2 itself when it was given this workspace as opts->d_tape (k_shade_pairs<SEG, true>);
    // the LeakyReLU masks of the data gradients are then read from the taped activations (no sign-bit words)
    const bool taped = tape_supported(*opts) && opts->d_tape == d_train_workspace && opts->tape_bytes == train_workspace_bytes;
    if (bf)
        hipLaunchKernelGGL(k_train_rows<true>, eg, eb, 0, st, P, tw);
    else if (K <= 32)
        hipLaunchKernelGGL(k_train_rows_x, eg, eb, 0, st, P, tw);
    else
        hipLaunchKernelGGL(k_train_rows<false>, eg, eb, 0, st, P, tw);
    if (!taped) {
        gemm_forward(st, bf, tw.X0, LD_X0, tw.Wp[0], 288, d_b[0], tw.H1, LD_H, 256, 288, n_rows, rows_max, tw.sgH1);
        gemm_forward(st, bf, tw.H1, LD_H, tw.Wp[1], 256, d_b[1], tw.H2, LD_H2, 256, 256, n_rows, rows_max, tw.sgH2);
        gemm_forward(st, bf, tw.H2, LD_H2, tw.Wp[2], 264, d_b[2], tw.G1, LD_H, 256, 264, n_rows, rows_max, tw.sgG1);
        gemm_forward(st, bf, tw.G1, LD_H, tw.Wp[3], 256, d_b[3], tw.G2, LD_H, 256, 256, n_rows, rows_max);
    }
    const unsigned long long *sgH1 = taped ? nullptr : tw.sgH1, *sgH2 = taped ? nullptr : tw.sgH2,
                             *sgG1 = taped ? nullptr : tw.sgG1;
    if (taped)
        hipLaunchKernelGGL(k_train_head_taped, eg, eb, 0, st, P, tw, reinterpret_cast<const float4 *>(ws.agg), ws.smp_sigma);
    else
        hipLaunchKernelGGL(k_train_head_agg, eg, eb, 0, st, P, tw, d_w[4], d_b[4]);
    if (!taped) {   // (a taped render's colour kernel left C1, C2, C3 and the head's sigmoids as well)
        gemm_forward(st, bf, tw.XC, LD_XC, tw.Wp[5], 288, d_b[5], tw.C1, LD_C, 128, 288, n_smp, smp_max, tw.sgC1);
        gemm_forward(st, bf, tw.C1, LD_C, tw.Wp[6], 128, d_b[6], tw.C2, LD_C, 128, 128, n_smp, smp_max, tw.sgC2);
        gemm_forward(st, bf, tw.C2, LD_C, tw.Wp[7], 128, d_b[7], tw.C3, LD_C, 128, 128, n_smp, smp_max);
        hipLaunchKernelGGL(k_train_color_head, eg, eb, 0, st, tw, d_w[8], d_b[8]);
    }
    // LeakyReLU masks of the colour MLP's data gradients: the sign words of the recompute's epilogues, or the taped floats
    const unsigned long long *sgC1 = taped ? nullptr : tw.sgC1, *sgC2 = taped ? nullptr : tw.sgC2;

    // ---- backward --------------------------------------------------------------------------------
    {
        static const int64_t wave_max_rays = [] {   // (PNR_COMPOSITE_WAVE_MAX_RAYS=0: always one thread per ray)
            const char *e = getenv("PNR_COMPOSITE_WAVE_MAX_RAYS");
            return e ? (int64_t)atoll(e) : (int64_t)16384;
        }();
        if (R <= wave_max_rays && opts->SR <= CBW_MAXS)
            hipLaunchKernelGGL(k_train_composite_bwd_wave, dim3((unsigned)((R + 3) / 4)), eb, 0, st, cr, *opts, R,
                               ws.ray_cnt, ws.ray_off, ws.ray_flag, ws.smp_loc, ws.n_sel, tw, d_grad_rgb,
                               d_rgb_recomputed);
        else
            hipLaunchKernelGGL(k_train_composite_bwd, dim3((unsigned)((R + 255) / 256)), eb, 0, st, cr, *opts, R,
                               ws.ray_cnt, ws.ray_off, ws.ray_flag, ws.smp_loc, ws.n_sel, tw, d_grad_rgb,
                               d_rgb_recomputed);
    }
    // colour MLP
    // exact mode: every data gradient goes to its own buffer (nothing is written over an activation a weight gradient
    // still reads -- or over anything a taped render left: a second pnr_render_backward on the same render finds the tape
    // as the first did), the seven weight-gradient GEMMs are queued and leave together at the end
    const bool batched = !bf;
    hipLaunchKernelGGL(k_train_color_head_bwd, dim3(256), eb, 0, st, tw, d_w[8], batched ? tw.D7 : tw.C3);  // dZ7
    hipLaunchKernelGGL(k_reduce_rows, dim3((387 + 15) / 16), eb, 0, st, tw.part, 1024, 388, 387, 384, tw.dWp[8], tw.dbp[8]);
    WgradQueue wq(tw.part, POOL_FLOATS);
    if (batched) {
        gemm_data(st, bf, tw.D7, LD_C, tw.Wp[7], 128, tw.WT[7], tw.D6, LD_C, 128, 128, 128, n_smp, smp_max, sgC2, tw.C2);   // dZ6
        gemm_data(st, bf, tw.D6, LD_C, tw.Wp[6], 128, tw.WT[6], tw.D5, LD_C, 128, 128, 128, n_smp, smp_max, sgC1, tw.C1);   // dZ5
        gemm_data(st, bf, tw.D5, LD_C, tw.Wp[5], 288, tw.WT[5], tw.DAGG, 256, 256, 128, 0, n_smp, smp_max);            // dAGG
        wq.add(tw.D7, LD_C, tw.C2, LD_C, tw.dWp[7], 128, 128, 128, n_smp, smp_max, tw.dbp[7]);
        wq.add(tw.D6, LD_C, tw.C1, LD_C, tw.dWp[6], 128, 128, 128, n_smp, smp_max, tw.dbp[6]);
        wq.add(tw.D5, LD_C, tw.XC, LD_XC, tw.dWp[5], 288, 128, 288, n_smp, smp_max, tw.dbp[5]);
    } else {
        gemm_weight(st, bf, tw.C3, LD_C, tw.C2, LD_C, tw.dWp[7], 128, 128, 128, n_smp, smp_max, tw.dbp[7], tw.part);
        gemm_data(st, bf, tw.C3, LD_C, tw.Wp[7], 128, tw.WT[7], tw.C2, LD_C, 128, 128, 128, n_smp, smp_max, tw.sgC2);  // C2 <- dZ6
        gemm_weight(st, bf, tw.C2, LD_C, tw.C1, LD_C, tw.dWp[6], 128, 128, 128, n_smp, smp_max, tw.dbp[6], tw.part);
        gemm_data(st, bf, tw.C2, LD_C, tw.Wp[6], 128, tw.WT[6], tw.C1, LD_C, 128, 128, 128, n_smp, smp_max, tw.sgC1);  // C1 <- dZ5
        gemm_weight(st, bf, tw.C1, LD_C, tw.XC, LD_XC, tw.dWp[5], 288, 128, 288, n_smp, smp_max, tw.dbp[5], tw.part);
        gemm_data(st, bf, tw.C1, LD_C, tw.Wp[5], 288, tw.WT[5], tw.XC, LD_XC, 256, 128, 0, n_smp, smp_max);   // XC[:, :256] <- dAGG
    }
    // density head + aggregation
    const bool want_points = grads->d_embedding || grads->d_color || grads->d_dir || grads->d_point_grads;
    PNR_REQUIRE(!grads->d_point_grads || (grads->d_point_index && grads->point_cap >= 1),
                "%s: sparse point gradients need d_point_index and point_cap", who);
    float *rowgrad = nullptr;
    if (batched) {
        // exact mode (any K: the chain walks tape rows, 32 per wave, and a row finds its sample as row / K; the mask bits
        // are indexed by tape row): the whole data-gradient chain of the pair MLPs in one kernel (pnr_train_chain.hip).  dw4 / db4
        // first (the tape G2 is read, nothing is written over it), then the chain: D3..D0 = the gradients at the four
        // pre-activations, rowgrad = the rows' point gradients, pt_cnt = rows per touched point
        if (!taped) launch_tape_bits(tw.cnt, tw.H1, tw.H2, tw.G1, tw.G2, tw.bits_rows, tw.tape_bits, st);
        hipLaunchKernelGGL(k_train_head_agg_bwd<false>, dim3(1024), eb, 0, st, tw, K, d_w[4]);
        hipLaunchKernelGGL(k_reduce_rows, dim3((257 + 15) / 16), eb, 0, st, tw.part, 1024, 260, 257, 256, tw.dWp[4], tw.dbp[4]);
        launch_pack_chain(d_w[0], d_w[1], d_w[2], d_w[3], d_w[4], tw.chainW, st);
        ChainParams C{};
        C.wchain = tw.chainW;
        C.w4acc = tw.chainW + CHAIN_W_FLOATS;
        C.cnt = tw.cnt;
        C.row_pidx = tw.row_pidx;
        C.row_w = tw.row_w;
        C.row_z = tw.row_z;
        C.d_out = tw.d_out;
        C.XC = tw.DAGG;
        C.tape_bits = tw.tape_bits;
        C.bits_rows = tw.bits_rows;
        C.X0 = tw.X0;
        C.D3 = tw.D3;
        C.D2 = tw.D2;
        C.D1 = tw.D1;
        C.D0 = tw.D0;
        C.rowgrad = tw.rowgrad;
        C.pt_rank = ws.pt_rank;
        C.pt_cnt = tw.pt_cnt;
        C.vs_list = ws.vs_list;
        C.smp_ray = ws.smp_ray;
        C.dirs = d_dirs;
        for (int i = 0; i < 9; ++i) C.Rw2c[i] = weights->Rw2c[i];
        C.K = K;
        {
            const int rcc = launch_pairs_bwd(C, rows_max, st);
            if (rcc != PNR_OK) return rcc;
        }
        wq.add(tw.D3, LD_H, tw.G1, LD_H, tw.dWp[3], 256, 256, 256, n_rows, rows_max, tw.dbp[3]);
        wq.add(tw.D2, LD_H, tw.H2, LD_H2, tw.dWp[2], 264, 256, 264, n_rows, rows_max, tw.dbp[2]);
        wq.add(tw.D1, LD_H, tw.H1, LD_H, tw.dWp[1], 256, 256, 256, n_rows, rows_max, tw.dbp[1]);
        wq.add(tw.D0, LD_H, tw.X0, LD_X0, tw.dWp[0], 288, 256, 288, n_rows, rows_max, tw.dbp[0]);
        PNR_REQUIRE(!wq.overflow, "%s: the weight-gradient pool is too small for this step's GEMMs", who);
        wq.launch(st);
        rowgrad = tw.rowgrad;
    } else {
        hipLaunchKernelGGL(k_train_head_agg_bwd<true>, dim3(1024), eb, 0, st, tw, K, d_w[4]);          // G2 <- dZ4
        hipLaunchKernelGGL(k_reduce_rows, dim3((257 + 15) / 16), eb, 0, st, tw.part, 1024, 260, 257, 256, tw.dWp[4], tw.dbp[4]);
        // mlp_head
        gemm_weight(st, bf, tw.G2, LD_H, tw.G1, LD_H, tw.dWp[3], 256, 256, 256, n_rows, rows_max, tw.dbp[3], tw.part);
        gemm_data(st, bf, tw.G2, LD_H, tw.Wp[3], 256, tw.WT[3], tw.G1, LD_H, 256, 256, 256, n_rows, rows_max, sgG1);   // G1 <- dZ3
        gemm_weight(st, bf, tw.G1, LD_H, tw.H2, LD_H2, tw.dWp[2], 264, 256, 264, n_rows, rows_max, tw.dbp[2], tw.part);
        gemm_data(st, bf, tw.G1, LD_H, tw.Wp[2], 264, tw.WT[2], tw.H2, LD_H2, 264, 256, 256, n_rows, rows_max, sgH2);  // H2 <- [dZ2 | d extras]
        // mlp_base
        gemm_weight(st, bf, tw.H2, LD_H2, tw.H1, LD_H, tw.dWp[1], 256, 256, 256, n_rows, rows_max, tw.dbp[1], tw.part);
        gemm_data(st, bf, tw.H2, LD_H2, tw.Wp[1], 256, tw.WT[1], tw.H1, LD_H, 256, 256, 256, n_rows, rows_max, sgH1);  // H1 <- dZ1
        gemm_weight(st, bf, tw.H1, LD_H, tw.X0, LD_X0, tw.dWp[0], 288, 256, 288, n_rows, rows_max, tw.dbp[0], tw.part);
        gemm_data(st, bf, tw.H1, LD_H, tw.Wp[0], 288, tw.WT[0], tw.G2, LD_H, 224, 256, 0, n_rows, rows_max);     // G2 <- dX0[:, :224]
        // per-row point gradients into H1 (its dZ1 is consumed)
        if (want_points) {
            rowgrad = tw.H1;
            hipLaunchKernelGGL(k_train_rowgrad, eg, eb, 0, st, P, tw, ws.pt_rank, rowgrad);
        }
    }
    // point tensors: rows grouped by point, one ordered sum per point
    if (want_points) {
        int rcs = scan_exclusive_i32(tw.pt_cnt, tw.pt_start, rows_max, ws.n_sel + 3, nullptr, tw.pt_scan, st);
        if (rcs != PNR_OK) return rcs;
        hipLaunchKernelGGL(k_train_fill, dim3(1024), eb, 0, st, tw, ws.pt_rank);
        hipLaunchKernelGGL(k_train_point_sum, eg, eb, 0, st, tw, ws.n_sel, ws.pt_list, rowgrad, grads->d_embedding,
                           grads->d_color, grads->d_dir, grads->d_point_grads, grads->d_point_index,
                           (long long)grads->point_cap);
    }
    // weight gradients out of the padded buffers
    {
        NineMats mw{}, mb{};
        for (int i = 0; i < 9; ++i) {
            mw.src[i] = tw.dWp[i];
            mw.dst[i] = grads->d_w[i];
            mw.n_out[i] = W_OUT[i];
            mw.n_in[i] = W_IN[i];
            mw.ld[i] = W_LD[i];
            mb.src[i] = tw.dbp[i];
            mb.dst[i] = grads->d_b[i];
            mb.n_out[i] = 1;
            mb.n_in[i] = W_OUT[i];
            mb.ld[i] = W_OUT[i];
        }
        hipLaunchKernelGGL(k_unpad_add, dim3(32, 9), eb, 0, st, mw);
        hipLaunchKernelGGL(k_unpad_add, dim3(1, 9), eb, 0, st, mb);
    }
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

