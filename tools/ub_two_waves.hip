// Micro-benchmark (diagnostic): does a second wave per SIMD hide VALU-only phases behind the other wave's MFMAs?
// Each wave alternates a "layer phase" (NM k-steps of 3 dependent 16x16x32 bf16 MFMAs + 2 LDS fragment reads)
// with a "prologue phase" (NV dependent-ish VALU instructions), like k_shade_pairs_bf16 does per tile.
// Run with 1 workgroup per CU (one wave per SIMD) and 2 per CU (two waves per SIMD); report work per second.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MINW>
__global__ void __launch_bounds__(256, MINW) k(const u32x4 *w, float *out, int tiles, int nm, int nv)
{
    __shared__ u32x4 lds[1280];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 1280; i += 256) lds[i] = w[i];
    __syncthreads();
    bf16x8 xh[8], xl[8];
    for (int s = 0; s < 8; ++s) {
        xh[s] = __builtin_bit_cast(bf16x8, w[(s * 64 + lane) % 1280]);
        xl[s] = __builtin_bit_cast(bf16x8, w[(s * 64 + lane + 7) % 1280]);
    }
    f32x4 acc = {0, 0, 0, 0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = out[tid + 256 * i];
    u32x4 fa = lds[lane], fb = lds[64 + lane], ga = lds[128 + lane], gb = lds[192 + lane];
    for (int t = 0; t < tiles; ++t) {
        for (int it = 0; it < nm; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                u32x4 ca = lds[(2 * ((s + 2) % 8)) * 64 + lane], cb = lds[(2 * ((s + 2) % 8) + 1) * 64 + lane];
                const bf16x8 wh = __builtin_bit_cast(bf16x8, fa), wl = __builtin_bit_cast(bf16x8, fb);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[s], acc, 0, 0, 0);
                fa = ga; fb = gb; ga = ca; gb = cb;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int it = 0; it < nv; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);   // 8 independent chains
        }
    }
    float r = acc[0] + acc[1] + acc[2] + acc[3];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 256 + tid] = r;
}

int main()
{
    u32x4 *w; float *out;
    hipMalloc(&w, 1280 * 16); hipMemset(w, 0x3c, 1280 * 16);
    hipMalloc(&out, 256 * 2048 * 4); hipMemset(out, 0, 256 * 2048 * 4);
    // per tile of 16 rows: 1632 MFMAs = 68 x 8 k-steps ; VALU 4200 instr = 525 x 8
    const int nm = 68, nv = 525;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs = 1; wgs <= 2; ++wgs) {
        const int tiles = 40;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (wgs == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, w, out, tiles, nm, nv);
            else hipLaunchKernelGGL(k<2>, dim3(512), dim3(256), 0, 0, w, out, tiles, nm, nv);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        double wave_tiles = (double)wgs * 256 * 4 * tiles;
        printf("%d WG/CU: %.3f ms for %.0f wave-tiles -> %.2f us per wave-tile per SIMD-slot, %.1f k wave-tiles/ms\n", wgs, ms,
               wave_tiles, ms * 1e3 / tiles, wave_tiles / ms / 1e3);
    }
    // MFMA-only and VALU-only references at 1 WG/CU
    float ms;
    hipEventRecord(e0); hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, w, out, 40, nm, 0); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("MFMA only : %.2f us per wave-tile\n", ms * 1e3 / 40);
    hipEventRecord(e0); hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, w, out, 40, 0, nv); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("VALU only : %.2f us per wave-tile\n", ms * 1e3 / 40);
    return 0;
}
