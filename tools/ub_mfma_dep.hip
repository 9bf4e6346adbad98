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

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// layer 4 of the pair kernel: 8 output tiles of 128 MFMAs; every output value is LeakyReLU'd, multiplied into the
// density head, weighted and summed over the 8 lanes of its sample.
//   MODE 0: no sink (reference)          MODE 1: the sink in three pieces per value (butterfly sum, store per 4 values)
//   MODE 2: the three DPP steps pipelined over consecutive values, the segment's lane t keeps tile t, stores at the end
template <int MODE, int CH = 1>
__global__ void __launch_bounds__(256, 1) k_layer4(const float *w, float *out, float *agg, int iters, float xs)
{
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, NG * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(agg + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 1024, 0, 4096, 0x00020000);
    const int voff = lane * 16;
    const int slot = lane & 7;
    const int ooff = slot == 0 ? (lane & 24) * 128 + 16 * h : 0x40000000;
    const float wgt = 0.125f + 0.01f * slot;
    float X[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) X[i] = to_a(xs * (float)((i * 7 + lane * 13) % 29 - 14));
    u32x4 wq[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) wq[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, p * 1024, 0);
    f32x16 acc[2];
    for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
    f32x16 acc2[2];   // CH == 2: the second accumulator chain (output tile m + 1 of a pair)
    for (int i = 0; i < 16; ++i) acc2[0][i] = acc2[1][i] = 0.f;
    float part = 0.f, sv = 0.f, o4[4] = {0.f, 0.f, 0.f, 0.f};
    float p1 = 0.f, p2 = 0.f, p3 = 0.f, mine[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mine[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        int sbase = 0;
        asm volatile("" : "+s"(sbase));
        if (CH == 2) {
            // two output tiles at a time: MFMAs alternate between two independent accumulators, so every VALU
            // instruction sits between INDEPENDENT MFMAs
#pragma unroll
            for (int mp = 0; mp < 4; ++mp) {
                const float4 hwa = *reinterpret_cast<const float4 *>(w + 128 * mp + 4 * h);
                const float4 hwb = *reinterpret_cast<const float4 *>(w + 128 * mp + 64 + 4 * h);
#pragma unroll
                for (int kg = 0; kg < 32; ++kg) {
                    const int G = (mp * 32 + kg) * 2;
                    const u32x4 wa = wq[G % PF], wb2 = wq[(G + 1) % PF];
                    wq[G % PF] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, sbase + ((G + PF) % NG) * 1024, 0);
                    wq[(G + 1) % PF] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, sbase + ((G + 1 + PF) % NG) * 1024, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = 4 * kg + q;
                        acc[mp & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wa[q]), X[i], acc[mp & 1], 0, 0, 0);
                        acc2[mp & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wb2[q]), X[i], acc2[mp & 1], 0, 0, 0);
                        // one value of each of the two finished tiles per 8 k-steps
                        const int r = i >> 3;
                        if (MODE == 1 && (i & 7) == 2) {
                            const float va = leaky(acc[(mp & 1) ^ 1][r]), vb = leaky(acc2[(mp & 1) ^ 1][r]);
                            part += va * ((r & 3) == 0 ? hwa.x : (r & 3) == 1 ? hwa.y : (r & 3) == 2 ? hwa.z : hwa.w);
                            part += vb * ((r & 3) == 0 ? hwb.x : (r & 3) == 1 ? hwb.y : (r & 3) == 2 ? hwb.z : hwb.w);
                            sv = va * wgt;
                            p1 = vb * wgt;
                        }
                        if (MODE == 1 && (i & 7) == 4) {
                            o4[r & 3] = dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(sv)));
                            mine[r & 3] = dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(p1)));
                        }
                        if (MODE == 1 && (i & 7) == 6 && (r & 3) == 3) {
                            u32x4 v, v2;
                            v.x = __float_as_uint(o4[0]);
                            v.y = __float_as_uint(o4[1]);
                            v.z = __float_as_uint(o4[2]);
                            v.w = __float_as_uint(o4[3]);
                            v2.x = __float_as_uint(mine[0]);
                            v2.y = __float_as_uint(mine[1]);
                            v2.z = __float_as_uint(mine[2]);
                            v2.w = __float_as_uint(mine[3]);
                            __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ooff, 128 * ((2 * mp + 6) & 7) + 32 * (r >> 2), 0);
                            __builtin_amdgcn_raw_buffer_store_b128(v2, orsrc, ooff, 128 * ((2 * mp + 7) & 7) + 32 * (r >> 2), 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float4 hw = *reinterpret_cast<const float4 *>(w + 64 * m + 4 * h);
#pragma unroll
            for (int kg = 0; kg < 32; ++kg) {
                const int G = m * 32 + kg;
                const u32x4 wv = wq[G % PF];
                wq[G % PF] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, sbase + ((G + PF) % NG) * 1024, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 4 * kg + q;
                    acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wv[q]), X[i], acc[m & 1], 0, 0, 0);
                    const int r = i >> 3;
                    if (MODE == 1) {
                        if ((i & 7) == 2) {
                            const float v = leaky(acc[(m & 1) ^ 1][r]);
                            part += v * ((r & 3) == 0 ? hw.x : (r & 3) == 1 ? hw.y : (r & 3) == 2 ? hw.z : hw.w);
                            sv = v * wgt;
                        }
                        if ((i & 7) == 4) o4[r & 3] = dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(sv)));
                        if ((i & 7) == 6 && (r & 3) == 3) {
                            u32x4 v;
                            v.x = __float_as_uint(o4[0]);
                            v.y = __float_as_uint(o4[1]);
                            v.z = __float_as_uint(o4[2]);
                            v.w = __float_as_uint(o4[3]);
                            __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ooff, 128 * ((m + 7) & 7) + 32 * (r >> 2), 0);
                        }
                    }
                    if (MODE == 3 && (i & 7) == 2) {   // leaky + head product only
                        const float v = leaky(acc[(m & 1) ^ 1][r]);
                        part += v * ((r & 3) == 0 ? hw.x : (r & 3) == 1 ? hw.y : (r & 3) == 2 ? hw.z : hw.w);
                        sv += v * wgt;
                    }
                    if (MODE == 4) {   // MODE 1 without the stores
                        if ((i & 7) == 2) {
                            const float v = leaky(acc[(m & 1) ^ 1][r]);
                            part += v * ((r & 3) == 0 ? hw.x : (r & 3) == 1 ? hw.y : (r & 3) == 2 ? hw.z : hw.w);
                            sv = v * wgt;
                        }
                        if ((i & 7) == 4) o4[r & 3] += dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(sv)));
                    }
                    if (MODE == 5 && (i & 7) == 2) {   // only the accumulator read
                        sv += acc[(m & 1) ^ 1][r];
                    }
                    if ((MODE == 6 || MODE == 7) && i < (MODE == 6 ? 4 : 16)) {   // the whole sink clustered behind the first 4 / 16 MFMAs
                        constexpr int PER = MODE == 6 ? 4 : 1;
#pragma unroll
                        for (int e = 0; e < PER; ++e) {
                            const int rr = PER * i + e;
                            const float v = leaky(acc[(m & 1) ^ 1][rr]);
                            part += v * ((rr & 3) == 0 ? hw.x : (rr & 3) == 1 ? hw.y : (rr & 3) == 2 ? hw.z : hw.w);
                            o4[rr & 3] = dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(v * wgt)));
                            if ((rr & 3) == 3) {
                                u32x4 vv;
                                vv.x = __float_as_uint(o4[0]);
                                vv.y = __float_as_uint(o4[1]);
                                vv.z = __float_as_uint(o4[2]);
                                vv.w = __float_as_uint(o4[3]);
                                __builtin_amdgcn_raw_buffer_store_b128(vv, orsrc, ooff, 128 * ((m + 7) & 7) + 32 * (rr >> 2), 0);
                            }
                        }
                    }
                    if (MODE == 2 && (i & 7) == 4) {
                        const float v = leaky(acc[(m & 1) ^ 1][r]);
                        part += v * ((r & 3) == 0 ? hw.x : (r & 3) == 1 ? hw.y : (r & 3) == 2 ? hw.z : hw.w);
                        // value r - 3 of the stream leaves the pipeline; the lane whose slot is the tile keeps it
                        const float a = dpp_add<0x141>(p3);
                        const int L = 16 * m + r - 3;
                        if (L >= 0) mine[L & 15] = (slot == ((L >> 4) & 7)) ? a : mine[L & 15];
                        p3 = dpp_add<0x4E>(p2);
                        p2 = dpp_add<0xB1>(p1);
                        p1 = v * wgt;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x4 v;
                v.x = __float_as_uint(mine[4 * q]);
                v.y = __float_as_uint(mine[4 * q + 1]);
                v.z = __float_as_uint(mine[4 * q + 2]);
                v.w = __float_as_uint(mine[4 * q + 3]);
                __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, (lane & 24) * 128 + 16 * h + 128 * slot, 32 * q, 0);
            }
        }
    }
    float s = part + sv + p1 + p2 + p3 + o4[0] + o4[1] + o4[2] + o4[3];
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i] + mine[i] + acc2[0][i] + acc2[1][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Two waves per SIMD (512 threads per workgroup, one workgroup per CU): waves 0..3 run an MFMA chain, waves 4..7 a
// chain of independent v_fma.  ROLE 1: only the MFMA waves work, 2: only the VALU waves, 3: both -- does the SIMD overlap
// the VALU work of one wave with the MFMA work of another?
template <int ROLE>
__global__ void __launch_bounds__(512, 1) k_two_waves(float *out, int iters, float a0)
{
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (ROLE & 1) {
            f32x16 acc;
            for (int i = 0; i < 16; ++i) acc[i] = (float)threadIdx.x;
            const float a = a0 + threadIdx.x, b = a0;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            for (int i = 0; i < 16; ++i) s += acc[i];
        }
    } else if (ROLE & 2) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = a0 * (float)(threadIdx.x + i);
        // 32 MFMAs = 2048+ cycles; 512 independent v_fma = 2048 issue cycles at 4 per instruction
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 64; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], 1.0000001f, a0);
        }
        for (int i = 0; i < 8; ++i) s += x[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
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
    {
        float *agg;
        (void)hipMalloc(&agg, 1024 * 4096);
        for (int rep = 0; rep < 2; ++rep) {
            timeit("layer4, no sink", 1024.0, [&](int it) { k_layer4<0><<<256, 256>>>(w, d, agg, it, xs); });
            timeit("layer4, sink in 3 pieces", 1024.0, [&](int it) { k_layer4<1><<<256, 256>>>(w, d, agg, it, xs); });
            timeit("layer4, pipelined DPP sink", 1024.0, [&](int it) { k_layer4<2><<<256, 256>>>(w, d, agg, it, xs); });
            timeit("layer4, leaky + head only", 1024.0, [&](int it) { k_layer4<3><<<256, 256>>>(w, d, agg, it, xs); });
            timeit("layer4, 3 pieces w/o stores", 1024.0, [&](int it) { k_layer4<4><<<256, 256>>>(w, d, agg, it, xs); });
            timeit("layer4, acc read only", 1024.0, [&](int it) { k_layer4<5><<<256, 256>>>(w, d, agg, it, xs); });
            timeit("layer4, 2 chains, no sink", 1024.0, [&](int it) { k_layer4<0, 2><<<256, 256>>>(w, d, agg, it, xs); });
            timeit("layer4, 2 chains, sink 3 pieces", 1024.0, [&](int it) { k_layer4<1, 2><<<256, 256>>>(w, d, agg, it, xs); });
            timeit("layer4, sink behind MFMA 0..3", 1024.0, [&](int it) { k_layer4<6><<<256, 256>>>(w, d, agg, it, xs); });
            timeit("layer4, sink behind MFMA 0..15", 1024.0, [&](int it) { k_layer4<7><<<256, 256>>>(w, d, agg, it, xs); });
        }
    }
    {
        float *d2;
        (void)hipMalloc(&d2, 256 * 512 * 4);
        for (int rep = 0; rep < 2; ++rep) {
            timeit("two waves/SIMD: MFMA waves only", 32.0 * 10, [&](int it) { k_two_waves<1><<<256, 512>>>(d2, it * 10, 1e-30f); });
            timeit("two waves/SIMD: VALU waves only", 32.0 * 10, [&](int it) { k_two_waves<2><<<256, 512>>>(d2, it * 10, 1e-30f); });
            timeit("two waves/SIMD: both", 32.0 * 10, [&](int it) { k_two_waves<3><<<256, 512>>>(d2, it * 10, 1e-30f); });
        }
    }
    // sustained: the same stream for ~3 s, time of every 15th launch (does the device hold the rate?)
    {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        for (int l = 0; l < 46; ++l) {
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
