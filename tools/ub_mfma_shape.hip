// Micro-benchmark (diagnostic, not part of the library): does the bf16 MFMA SHAPE change the speed of a power-limited
// bf16x3 inner loop?  Both kernels do the same arithmetic per step (a 32-row x 32-column x 32-deep block of the
// three hi/lo products), read the same A-fragment bytes from LDS (4 x ds_read_b128 per step) and carry the same VALU
// filler (the split work of the real kernel); one wave per SIMD, every CU busy, RANDOM operands (zeros raise the clock).
//   shape 0: v_mfma_f32_32x32x16_bf16, 2 k-steps x 3 MFMAs of 32 cycles
//   shape 1: v_mfma_f32_16x16x32_bf16, 2 row tiles x 2 column halves x 3 MFMAs of 16 cycles
// Reported: wall time per launch (what matters), shader cycles per step (s_memtime) and the clock they imply.
// build: hipcc --offload-arch=gfx950 -O3 tools/ub_mfma_shape.hip -o tools/bin/ub_mfma_shape ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int VALU>
__global__ void __launch_bounds__(256, 1) k(const u32x4 *w, float *out, unsigned long long *cyc, int iters)
{
    __shared__ u32x4 lds[4096];  // 64 KiB of fragments
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) lds[i] = w[i];
    __syncthreads();
    bf16x8 xh[16], xl[16];
    for (int s = 0; s < 16; ++s) {
        xh[s] = __builtin_bit_cast(bf16x8, w[(s * 64 + lane + 4096) % 8192]);
        xl[s] = __builtin_bit_cast(bf16x8, w[(s * 64 + lane + 4096 + 1031) % 8192]);
    }
    f32x16 acc = {0};
    f32x4 a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0};
    float v0 = out[tid], v1 = out[tid + 256];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {  // one step = 32 rows x 32 cols x 32 k, hi/lo x3
            const int f = ((it * 8 + s) * 4) & 63;
            const u32x4 f0 = lds[(f + 0) * 64 + lane], f1 = lds[(f + 1) * 64 + lane];
            const u32x4 f2 = lds[(f + 2) * 64 + lane], f3 = lds[(f + 3) * 64 + lane];
            const bf16x8 ah0 = __builtin_bit_cast(bf16x8, f0), al0 = __builtin_bit_cast(bf16x8, f1);
            const bf16x8 ah1 = __builtin_bit_cast(bf16x8, f2), al1 = __builtin_bit_cast(bf16x8, f3);
            if (SHAPE == 0) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, xh[2 * s], acc, 0, 0, 0);
                if (VALU) { v0 = fmaxf(v0, 0.1f * v0) + 1.0f; v1 = fmaxf(v1, 0.1f * v1) + 1.0f; v0 = v0 * 1.0001f; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, xl[2 * s], acc, 0, 0, 0);
                if (VALU) { v0 = v0 - v1 * 0.5f; v1 = v1 + v0 * 0.25f; v0 = v0 * 1.0001f; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, xh[2 * s], acc, 0, 0, 0);
                if (VALU) { v0 = v0 + 0.5f; v1 = v1 * 1.5f; v0 = v0 - v1; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, xh[2 * s + 1], acc, 0, 0, 0);
                if (VALU) { v0 = fmaxf(v0, 0.1f * v0) + 1.0f; v1 = fmaxf(v1, 0.1f * v1) + 1.0f; v0 = v0 * 1.0001f; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, xl[2 * s + 1], acc, 0, 0, 0);
                if (VALU) { v0 = v0 - v1 * 0.5f; v1 = v1 + v0 * 0.25f; v0 = v0 * 1.0001f; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, xh[2 * s + 1], acc, 0, 0, 0);
                if (VALU) { v0 = v0 + 0.5f; v1 = v1 * 1.5f; v0 = v0 - v1; }
            } else {
                // row tile 0 (ah0, al0) and 1 (ah1, al1) x column halves (xh/xl[2s], xh/xl[2s+1])
                a00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, xh[2 * s], a00, 0, 0, 0);
                if (VALU) { v0 = fmaxf(v0, 0.1f * v0) + 1.0f; v1 = fmaxf(v1, 0.1f * v1) + 1.0f; }
                a00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, xl[2 * s], a00, 0, 0, 0);
                if (VALU) { v0 = v0 * 1.0001f; v0 = v0 - v1 * 0.5f; }
                a00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al0, xh[2 * s], a00, 0, 0, 0);
                if (VALU) { v1 = v1 + v0 * 0.25f; v0 = v0 * 1.0001f; }
                a01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, xh[2 * s + 1], a01, 0, 0, 0);
                if (VALU) { v0 = v0 + 0.5f; v1 = v1 * 1.5f; }
                a01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, xl[2 * s + 1], a01, 0, 0, 0);
                if (VALU) { v0 = v0 - v1; v0 = fmaxf(v0, 0.1f * v0) + 1.0f; }
                a01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al0, xh[2 * s + 1], a01, 0, 0, 0);
                if (VALU) { v1 = fmaxf(v1, 0.1f * v1) + 1.0f; v0 = v0 * 1.0001f; }
                a10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, xh[2 * s], a10, 0, 0, 0);
                if (VALU) { v0 = v0 - v1 * 0.5f; v1 = v1 + v0 * 0.25f; }
                a10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, xl[2 * s], a10, 0, 0, 0);
                if (VALU) { v0 = v0 * 1.0001f; v0 = v0 + 0.5f; }
                a10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al1, xh[2 * s], a10, 0, 0, 0);
                if (VALU) { v1 = v1 * 1.5f; v0 = v0 - v1; }
                a11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, xh[2 * s + 1], a11, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, xl[2 * s + 1], a11, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al1, xh[2 * s + 1], a11, 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
    float r = v0 + v1;
    for (int i = 0; i < 16; ++i) r += acc[i];
    for (int i = 0; i < 4; ++i) r += a00[i] + a01[i] + a10[i] + a11[i];
    out[blockIdx.x * 256 + tid] = r;
}

int main()
{
    u32x4 *w; float *out; unsigned long long *cyc;
    std::vector<unsigned short> h(8192 * 8);
    srand(7);
    for (auto &v : h) {  // random bf16 in (-2, 2): random sign, exponent near 1, random mantissa
        const unsigned short mant = rand() & 0x7f, sign = (rand() & 1) << 15, ex = (125 + (rand() % 3)) << 7;
        v = sign | ex | mant;
    }
    hipMalloc(&w, 8192 * 16); hipMemcpy(w, h.data(), 8192 * 16, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 512 * 4); hipMemset(out, 0, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 4 * 8);
    const int iters = 40000;  // ~0.3 s per launch: long enough for the clock to settle
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int round = 0; round < 2; ++round)
        for (int v = 0; v < 4; ++v) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL((k<0, 0>), dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
            if (v == 1) hipLaunchKernelGGL((k<1, 0>), dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
            if (v == 2) hipLaunchKernelGGL((k<0, 1>), dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
            if (v == 3) hipLaunchKernelGGL((k<1, 1>), dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long hc[1024];
            hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 1024; ++i) s += hc[i];
            const double cyc_step = s / 1024 / iters / 8;
            printf("round %d shape %s valu %d: %.1f ms, %.1f cycles per step (MFMA floor 192), clock %.2f GHz, %.0f TFLOP/s executed\n",
                   round, (v & 1) ? "16x16x32" : "32x32x16", v >> 1, ms, cyc_step, s / 1024 / (ms * 1e6),
                   256.0 * 4 * iters * 8 * 6 * 32768 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
