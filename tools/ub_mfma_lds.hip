// Micro-benchmark (diagnostic, not part of the library): cycles per k-step of the bf16x3 inner loop
//   variant 0: 3 dependent MFMAs per k-step, operands in registers
//   variant 1: + two ds_read_b128 A-fragments per k-step, read two k-steps ahead (3-deep ring)
//   variant 2: variant 1 + 14 VALU ops per k-step placed between the MFMAs
// build: hipcc --offload-arch=gfx950 -O3 tools/ub_mfma_lds.hip -o gpurun_out/ub_mfma_lds ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int VARIANT>
__global__ void __launch_bounds__(256, 1) k(const u32x4 *w, float *out, unsigned long long *cyc, int iters)
{
    __shared__ u32x4 lds[2304];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2304; i += 256) lds[i] = w[i];
    __syncthreads();
    bf16x8 xh[16], xl[16];
    for (int s = 0; s < 16; ++s) {
        xh[s] = __builtin_bit_cast(bf16x8, w[(s * 64 + lane) % 2304]);
        xl[s] = __builtin_bit_cast(bf16x8, w[(s * 64 + lane + 7) % 2304]);
    }
    f32x16 acc = {0};
    float v0 = out[tid], v1 = out[tid + 256];
    u32x4 fa = lds[lane], fb = lds[64 + lane], ga = lds[128 + lane], gb = lds[192 + lane];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            u32x4 ca = fa, cb = fb;
            if (VARIANT >= 1) {
                ca = lds[(2 * ((s + 2) % 16)) * 64 + lane];
                cb = lds[(2 * ((s + 2) % 16) + 1) * 64 + lane];
            }
            const bf16x8 wh = __builtin_bit_cast(bf16x8, fa), wl = __builtin_bit_cast(bf16x8, fb);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[s], acc, 0, 0, 0);
            if (VARIANT >= 2) { v0 = fmaxf(v0, 0.1f * v0) + 1.0f; v1 = fmaxf(v1, 0.1f * v1) + 1.0f; v0 = v0 * 1.0001f; v1 = v1 * 0.9999f; }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[s], acc, 0, 0, 0);
            if (VARIANT >= 2) { v0 = v0 - v1 * 0.5f; v1 = v1 + v0 * 0.25f; v0 = v0 * 1.0001f; }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[s], acc, 0, 0, 0);
            if (VARIANT >= 2) { v0 = v0 + 0.5f; v1 = v1 * 1.5f; v0 = v0 - v1; }
            fa = ga; fb = gb; ga = ca; gb = cb;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
    float r = v0 + v1;
    for (int i = 0; i < 16; ++i) r += acc[i];
    out[blockIdx.x * 256 + tid] = r;
}

int main()
{
    u32x4 *w; float *out; unsigned long long *cyc;
    hipMalloc(&w, 2304 * 16); hipMemset(w, 0x3c, 2304 * 16);
    hipMalloc(&out, 256 * 512 * 4); hipMemset(out, 0, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 4 * 8);
    const int iters = 2000;
    for (int v = 0; v < 3; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
            if (v == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
            if (v == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, w, out, cyc, iters);
            hipDeviceSynchronize();
        }
        unsigned long long h[1024];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
        printf("variant %d: %.1f ticks per k-step (3 MFMAs; ideal 96)\n", v, s / 1024 / iters / 16);
    }
    return 0;
}
