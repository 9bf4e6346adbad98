// Shade stage, exact mode (PNR_PRECISION_FP32): v_mfma_f32_32x32x2_f32, weights streamed L2 -> VGPR.  See
// pnr_shade_common.h for the design overview.
#include "pnr_shade_common.h"

namespace pnr {

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

void launch_point_part_fp32(dim3 grid, hipStream_t stream, const ShadeParams &P)
{
    hipLaunchKernelGGL(k_point_part_f32, grid, dim3(TPB), 0, stream, P);
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

void launch_color_fp32(dim3 grid, hipStream_t stream, const ShadeParams &P)
{
    hipLaunchKernelGGL(k_shade_color, grid, dim3(TPB), 0, stream, P);
}

}  // namespace pnr
