// Micro-benchmark: what a stream of v_mfma_f32_32x32x2_f32 loses to the things around it.  One wave per SIMD
// (256 workgroups x 256 threads).
//   chains<N>  : N independent accumulator chains, no memory traffic
//   stream<L,B,F>: the pair kernel's inner structure -- 128 MFMAs per output tile on one accumulator,
//                L: one buffer_load_dwordx4 of weights per 4 MFMAs through a 6-deep rolling window (840 KB, L2-resident)
//                B: the 128 B operands in AGPRs (pinned) instead of one VGPR
//                F: behind the first 4 MFMAs of a tile, 16 accumulator values -> LeakyReLU -> next B operands
//   hipcc --offload-arch=gfx950 -O3 -mllvm -pragma-unroll-threshold=4000000 tools/ub_mfma_dep.hip -o tools/bin/ub_mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void __launch_bounds__(256, 1) k_chains(float *out, int iters, float a0, float b0)
{
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = (float)(threadIdx.x + c);
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / CHAINS; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__device__ __forceinline__ float to_a(float v)
{
    asm("" : "+a"(v));
    return v;
}
__device__ __forceinline__ float leaky(float x)
{
    float r;
    const float y = 0.1f * x;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

constexpr int PF = 6;
constexpr int NG = 840;   // weight groups of one pass (1 KiB each)

template <bool LOADS, bool BREGS, bool FILL, bool SKEW = false>
__global__ void __launch_bounds__(256, 1) k_stream(const float *w, float *out, int iters, float xs)
{
    const int lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, NG * 1024, 0x00020000);
    const int voff = lane * 16;
    float X[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) X[i] = BREGS ? to_a(xs * (float)((i * 7 + lane * 13) % 29 - 14)) : xs * (float)lane;
    u32x4 wq[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) wq[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, p * 1024, 0);
    f32x16 acc[2];
    for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        // SKEW: every wave of the device at its own place of the weight stream (no lockstep sharing of L1 lines)
        int sbase = SKEW ? (int)(((threadIdx.x >> 6) * 210 + blockIdx.x * 37) % NG) * 1024 : 0;
        sbase = __builtin_amdgcn_readfirstlane(sbase);
        asm volatile("" : "+s"(sbase));
        // 6 output tiles of 128 MFMAs (+ 72 to make 840 groups): 210 groups x 4 tiles
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int kg = 0; kg < 210; ++kg) {
                const int G = m * 210 + kg;
                const u32x4 wv = wq[G % PF];
                if (LOADS) wq[G % PF] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (sbase + ((G + PF) % NG) * 1024) % (NG * 1024), 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kk = (4 * kg + q) % 128;
                    acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wv[q]), X[BREGS ? kk : 0], acc[m & 1], 0, 0, 0);
                    if (FILL && kg % 32 == 0) {
                        // 4 values behind each of the first 4 MFMAs of every 128
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int idx = (16 * ((kg / 32) % 8) + 4 * q + r) % 128;
                            const float v = leaky(acc[(m & 1) ^ 1][4 * q + r]);
                            if (BREGS) X[idx] = to_a(v); else X[0] += v;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    for (int i = 0; i < 128; ++i) s += X[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static void timeit(const char *name, double mfma_per_wave, F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch(2);
    (void)hipDeviceSynchronize();
    const int iters = 200;
    (void)hipEventRecord(e0);
    launch(iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = mfma_per_wave * iters;
    printf("%-28s %.3f ms  %.2f ns per MFMA = %.1f cycles at 2.4 GHz\n", name, ms, ms * 1e6 / mfma, ms * 1e6 / mfma * 2.4);
}

int main(int argc, char **)
{
    float *d, *w;
    (void)hipMalloc(&d, 256 * 256 * 4);
    (void)hipMalloc(&w, NG * 1024);
    (void)hipMemset(w, 0, NG * 1024);
    const bool random_w = argc > 1;   // any argument: normally distributed weights instead of zeros (data-dependent power)
    if (random_w) {
        std::vector<float> hw(NG * 256);
        for (auto &v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.25f;
        (void)hipMemcpy(w, hw.data(), NG * 1024, hipMemcpyHostToDevice);
        printf("random weights\n");
    }
    const float xs = random_w ? 0.07f : 1e-30f;
    for (int rep = 0; rep < 2; ++rep) {
        timeit("chains 1", 32.0 * 100, [&](int it) { k_chains<1><<<256, 256>>>(d, it * 100, 1e-30f, 1e-30f); });
        timeit("chains 2", 32.0 * 100, [&](int it) { k_chains<2><<<256, 256>>>(d, it * 100, 1e-30f, 1e-30f); });
        timeit("stream", 3360.0, [&](int it) { k_stream<false, false, false><<<256, 256>>>(w, d, it, xs); });
        timeit("stream +loads", 3360.0, [&](int it) { k_stream<true, false, false><<<256, 256>>>(w, d, it, xs); });
        timeit("stream +bregs", 3360.0, [&](int it) { k_stream<false, true, false><<<256, 256>>>(w, d, it, xs); });
        timeit("stream +loads +bregs", 3360.0, [&](int it) { k_stream<true, true, false><<<256, 256>>>(w, d, it, xs); });
        timeit("stream +loads +bregs +fill", 3360.0, [&](int it) { k_stream<true, true, true><<<256, 256>>>(w, d, it, xs); });
        timeit("stream +loads +bregs +fill, waves skewed", 3360.0, [&](int it) { k_stream<true, true, true, true><<<256, 256>>>(w, d, it, xs); });
        timeit("stream +bregs +fill", 3360.0, [&](int it) { k_stream<false, true, true><<<256, 256>>>(w, d, it, xs); });
    }
    // sustained: the same stream for ~3 s, time of every 15th launch (does the device hold the rate?)
    {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        for (int l = 0; l < 165; ++l) {
            (void)hipEventRecord(e0);
            k_stream<true, true, true><<<256, 256>>>(w, d, 200, xs);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (l % 15 == 0) printf("sustained launch %3d: %.3f ms = %.1f cycles per MFMA at 2.4 GHz\n", l, ms, ms * 1e6 / (3360.0 * 200) * 2.4);
        }
    }
    return 0;
}
