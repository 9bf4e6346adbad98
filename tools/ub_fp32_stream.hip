// Micro-benchmark (diagnostic, not part of the library): the fp32 pair kernel's weight stream on its own -- what does
// one wave per SIMD sustain when EVERY CU streams the 840 weight groups of a tile (0.84 MB per wave and tile, L2 ->
// VGPR, one 1-KiB buffer_load per four v_mfma_f32_32x32x2_f32) and does nothing else?
//   variant 0: 4 MFMAs + 1 load per group, PF loads in flight
//   variant 1: + three VALU instructions behind every MFMA (the activation filler of the real kernel)
//   variant 2: variant 0 without loads (weights stay in registers): the MFMA issue floor
// prints cycles per MFMA (s_memtime, wave 0 of every workgroup, median) and the kernel time
// build: hipcc --offload-arch=gfx950 -O3 tools/ub_fp32_stream.hip -o tools/bin/ub_fp32_stream ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NG = 840;

__device__ __forceinline__ float4 load_w(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int VARIANT, int PF>
__global__ void __launch_bounds__(256, 1) k(const float *w, size_t wbytes, float *out, unsigned long long *cyc, int tiles)
{
    const int lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, (int)wbytes, 0x00020000);
    const int voff = lane * 16;
    float in[128], nx[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) { in[i] = w[(lane + 64 * i) & 4095]; nx[i] = 0.f; }
    f32x16 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = f32x16{0};
    float4 wq[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) wq[p] = load_w(rsrc, voff, p * 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        int base = 0;
        asm volatile("" : "+s"(base));
        // four layers of 8 output tiles x {8, 32, 33, 32} groups, as the pair kernel (nested loops: each unrolls fully)
#pragma unroll
        for (int L = 0; L < 4; ++L) {
            const int KG = L == 0 ? 8 : (L == 2 ? 33 : 32);
            const int G0 = L == 0 ? 0 : (L == 1 ? 64 : (L == 2 ? 320 : 584));
#pragma unroll
            for (int m = 0; m < 8; ++m) {
#pragma unroll
                for (int kg = 0; kg < 33; ++kg) {
                    if (kg >= KG) continue;
                    const int g = G0 + m * KG + kg;
                    float4 wv = wq[g % PF];
                    if (VARIANT != 2) wq[g % PF] = load_w(rsrc, voff, base + ((g + PF) % NG) * 1024);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float a = q == 0 ? wv.x : q == 1 ? wv.y : q == 2 ? wv.z : wv.w;
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, in[(4 * kg + q) & 127], acc[m], 0, 0, 0);
                        if (VARIANT == 1 && (4 * kg + q) < 16) {
                            const float v = acc[(m + 7) % 8][4 * kg + q];
                            float r;
                            const float y = 0.1f * v;
                            asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(y));
                            nx[16 * m + 4 * kg + q] = r;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (VARIANT == 1) {
#pragma unroll
            for (int i = 0; i < 128; ++i) in[i] = nx[i] * 1e-3f + in[i];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, int PF>
static void run(const char *name, const float *w, size_t wbytes, float *out, unsigned long long *cyc, int tiles)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V, PF>), dim3(256), dim3(256), 0, 0, w, wbytes, out, cyc, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, PF>), dim3(256), dim3(256), 0, 0, w, wbytes, out, cyc, tiles);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double n = (double)tiles * NG * 4;
    printf("%-34s %8.3f ms  cycles/MFMA median %.2f  max %.2f   (%.1f TFLOP/s)\n", name, ms, h[128] / n, h[255] / n,
           256.0 * 4 * n * 4096 / (ms * 1e-3) / 1e12);
}

int main()
{
    const size_t wfloats = (size_t)NG * 256 + 65536;
    float *w, *out;
    unsigned long long *cyc;
    hipMalloc(&w, wfloats * 4);
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    std::vector<float> h(wfloats);
    for (size_t i = 0; i < wfloats; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(w, h.data(), wfloats * 4, hipMemcpyHostToDevice);
    const int tiles = 80;
    for (int rep = 0; rep < 2; ++rep) {
        run<2, 6>("no loads (issue floor)", w, wfloats * 4, out, cyc, tiles);
        run<0, 6>("stream, PF 6", w, wfloats * 4, out, cyc, tiles);
        run<0, 12>("stream, PF 12", w, wfloats * 4, out, cyc, tiles);
        run<0, 24>("stream, PF 24", w, wfloats * 4, out, cyc, tiles);
        run<1, 12>("stream + 3 VALU per MFMA, PF 12", w, wfloats * 4, out, cyc, tiles);
    }
    return 0;
}
