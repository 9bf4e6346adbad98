// Micro-benchmark (diagnostic, not part of the library): the alpha composite as ONE THREAD PER RAY with a serial loop over
// its compact sample list (what k_composite does, csrc/pnr_render.hip) against ONE WAVEFRONT PER HIT RAY with a
// multiplicative prefix scan of the transmittance over the lanes (the form BASELINE.json's north_star names), on a frame
// shaped like the metric's: 640 000 rays, 80 % of them without samples, the rest with 1..80 samples (mean ~11).
// Both produce sum_i w_i c_i, w_i = o_i prod_{j<i}(1 - o_j + 1e-10); the scan associates the product differently
// (last-bit differences), the serial loop is torch.cumprod's order.
// build: hipcc --offload-arch=gfx950 -O3 tools/ub_composite.hip -o tools/bin/ub_composite ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

__global__ void __launch_bounds__(256) k_thread(int R, const int *cnt, const int *off, const float4 *smp, float4 *out)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, acc = 0.f;
    const int o = off[r], n = cnt[r];
    for (int i = 0; i < n; ++i) {
        const float4 s = smp[o + i];
        const float op = 1.0f - expf(-s.x * 0.004f);
        const float w = op * T;
        T *= (1.0f - op + 1e-10f);
        cr += w * s.y; cg += w * s.z; cb += w * s.w; acc += w;
    }
    out[r] = make_float4(cr + 1.f - acc, cg + 1.f - acc, cb + 1.f - acc, acc);
}

// one wavefront per listed (hit) ray; lane i takes samples i and i + 64
__global__ void __launch_bounds__(256) k_wave(int n_hit, const int *hit, const int *cnt, const int *off, const float4 *smp,
                                              float4 *out)
{
    const int lane = threadIdx.x & 63;
    const int h = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= n_hit) return;
    const int r = hit[h];
    const int o = off[r], n = cnt[r];
    float T0 = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, acc = 0.f;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) s = smp[o + i];
        const float op = i < n ? 1.0f - expf(-s.x * 0.004f) : 0.f;
        float p = i < n ? (1.0f - op + 1e-10f) : 1.f;   // inclusive product scan over the lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float q = __shfl_up(p, d, 64);
            if (lane >= d) p *= q;
        }
        float excl = __shfl_up(p, 1, 64);
        if (lane == 0) excl = 1.f;
        const float w = op * T0 * excl;
        float a = w * s.y, b = w * s.z, c = w * s.w, d4 = w;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); c += __shfl_xor(c, d, 64); d4 += __shfl_xor(d4, d, 64);
        }
        cr += a; cg += b; cb += c; acc += d4;
        T0 *= __shfl(p, 63, 64);
    }
    if (lane == 0) out[r] = make_float4(cr + 1.f - acc, cg + 1.f - acc, cb + 1.f - acc, acc);
}

int main()
{
    const int R = 640000;
    std::vector<int> cnt(R), off(R), hit;
    srand(1);
    int total = 0;
    for (int r = 0; r < R; ++r) {
        // rays come in 16x16 tiles: whole tiles miss the object
        const bool tile_hit = ((r / 256) * 2654435761u >> 8) % 5 == 0;
        int n = 0;
        if (tile_hit) { n = 1 + (int)(-10.0 * log((rand() + 1.0) / (RAND_MAX + 2.0))); if (n > 80) n = 80; }
        cnt[r] = n; off[r] = total; total += n;
        if (n) hit.push_back(r);
    }
    std::vector<float4> smp(total);
    for (int i = 0; i < total; ++i) smp[i] = make_float4(300.f * rand() / RAND_MAX, (float)rand() / RAND_MAX, (float)rand() / RAND_MAX, (float)rand() / RAND_MAX);
    int *d_cnt, *d_off, *d_hit; float4 *d_smp, *d_a, *d_b;
    hipMalloc(&d_cnt, R * 4); hipMalloc(&d_off, R * 4); hipMalloc(&d_hit, hit.size() * 4 + 4);
    hipMalloc(&d_smp, (size_t)total * 16); hipMalloc(&d_a, (size_t)R * 16); hipMalloc(&d_b, (size_t)R * 16);
    hipMemcpy(d_cnt, cnt.data(), R * 4, hipMemcpyHostToDevice); hipMemcpy(d_off, off.data(), R * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_hit, hit.data(), hit.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_smp, smp.data(), (size_t)total * 16, hipMemcpyHostToDevice);
    hipMemset(d_b, 0, (size_t)R * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n_hit = (int)hit.size();
    for (int v = 0; v < 2; ++v) {
        float best = 1e9f;
        for (int it = 0; it < 20; ++it) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k_thread, dim3((R + 255) / 256), dim3(256), 0, 0, R, d_cnt, d_off, d_smp, d_a);
            else hipLaunchKernelGGL(k_wave, dim3((n_hit + 3) / 4), dim3(256), 0, 0, n_hit, d_hit, d_cnt, d_off, d_smp, d_b);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 3 && ms < best) best = ms;
        }
        printf("%-44s %.4f ms\n", v == 0 ? "one thread per ray, serial loop (all rays)" : "one wavefront per hit ray, shuffle scan", best);
    }
    std::vector<float4> a(R), b(R);
    hipMemcpy(a.data(), d_a, (size_t)R * 16, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d_b, (size_t)R * 16, hipMemcpyDeviceToHost);
    double err = 0;
    for (int r : hit) { err = fmax(err, fabs(a[r].x - b[r].x)); err = fmax(err, fabs(a[r].w - b[r].w)); }
    printf("rays %d, hit %d, samples %d (mean %.1f per hit ray); max |serial - scan| = %.2e\n", R, n_hit, total, (double)total / n_hit, err);
    return 0;
}
