// Micro-benchmark (diagnostic, not part of the library): cost of the per-pair pt_table gather of k_shade_pairs_bf16.
// Each wave fetches 32 rows x 1 KiB (32 x global_load_dwordx4 per lane), 4 waves per CU, every CU busy.
//   variant 0: the kernel's pattern -- lane (j, h) reads its own 16 B of row j, 32 distinct rows per instruction
//   variant 1: quad-coalesced -- the 4 lanes of a quad read the 4 x 16 B of ONE 64-byte segment
//   variant 2: fully contiguous -- an instruction reads 1 KiB of one row (what an LDS-DMA staging would issue)
// rows are random over `nrows` (argv[1], default 1.4M = 1.4 GB: mostly L2 misses; 4096 = L2 resident)
// build: hipcc --offload-arch=gfx950 -O3 tools/ub_gather.hip -o gpurun_out/ub_gather ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int VARIANT>
__global__ void __launch_bounds__(256, 1) k(const float4 *table, const int *rows, int nrows, float *out,
                                            unsigned long long *cyc, int iters)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    float acc = 0.f;
    unsigned long long t_issue = 0, t_wait = 0;
    for (int it = 0; it < iters; ++it) {
        const int base = ((blockIdx.x * iters + it) * 4 + wave) * 32;
        const int myrow = rows[(base + j) % (1 << 22)];
        float4 v[32];
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int B = 0; B < 8; ++B)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (VARIANT == 0) {
                    v[4 * B + q] = table[(long long)myrow * 64 + 8 * B + 4 * h + q];
                } else if (VARIANT == 1) {
                    // instruction q serves target lanes t = 16q + (lane >> 2); quad lane p reads chunk p
                    const int t = 16 * q + (lane >> 2);
                    const int row = __shfl(myrow, t & 31, 64);
                    v[4 * B + q] = table[(long long)row * 64 + 8 * B + 4 * (t >> 5) + (lane & 3)];
                } else {
                    const int row = __shfl(myrow, (4 * B + q) & 31, 64);
                    v[4 * B + q] = table[(long long)row * 64 + lane];
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 32; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
        t_issue += t1 - t0;
        t_wait += t2 - t1;
    }
    if (lane == 0) {
        cyc[(blockIdx.x * 4 + wave) * 2] = t_issue;
        cyc[(blockIdx.x * 4 + wave) * 2 + 1] = t_wait;
    }
    out[blockIdx.x * 256 + tid] = acc;
}

int main(int argc, char **argv)
{
    const int nrows = argc > 1 ? atoi(argv[1]) : 1400000;
    const int iters = 64, grid = 256;
    float4 *table;
    int *rows;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&table, (size_t)nrows * 1024);
    hipMemset(table, 0, (size_t)nrows * 1024);
    std::vector<int> h(1 << 22);
    srand(1);
    // neighbouring pairs share points in the renderer: draw rows from a sliding window to mimic that locality
    for (size_t i = 0; i < h.size(); ++i) h[i] = (int)(((long long)rand() * 65536 + rand()) % nrows);
    hipMalloc(&rows, h.size() * 4);
    hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, grid * 256 * 4);
    hipMalloc(&cyc, grid * 4 * 2 * 8);
    std::vector<unsigned long long> c(grid * 8);
    for (int variant = 0; variant < 3; ++variant) {
        for (int rep = 0; rep < 2; ++rep) {
            if (variant == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, table, rows, nrows, out, cyc, iters);
            if (variant == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, table, rows, nrows, out, cyc, iters);
            if (variant == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, table, rows, nrows, out, cyc, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int i = 0; i < grid * 4; ++i) {
            a += c[2 * i];
            b += c[2 * i + 1];
        }
        printf("rows %d variant %d: issue %.0f cycles, wait %.0f cycles per wave and 32-row gather (32 KiB)\n", nrows,
               variant, a / (grid * 4) / iters, b / (grid * 4) / iters);
    }
    return 0;
}
