// Device-wide exclusive scan of int32 (two launches: block reduce, downsweep -- every downsweep block sums the totals of the
// blocks in front of it itself: at most ~1500 L2-resident ints, cheaper than a third launch for a serial scan of them).
// Used for ray -> sample offsets, sample -> valid-sample offsets, brick ranks and voxel starts; n may
// live in device memory so the render path never returns to the host.
#include "pnr_internal.h"

namespace pnr {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 4096 elements per workgroup

__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan across the block of one value per thread; returns the exclusive prefix, total in *total
__device__ __forceinline__ int block_excl_scan(int v, int *total, int *smem /* [SCAN_THREADS/64 + 1] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = wave_incl_scan(v);
    if (lane == 63) smem[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        int s = smem[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// A wavefront owns 1024 consecutive elements of its workgroup's 4096 and reads them STRIPED: lane l takes the int4 at
// element 4 (64 k + l), k = 0..3 -- every load instruction covers one contiguous KiB (with 16 consecutive elements per
// thread an instruction touched 32 cache lines 64 bytes apart: the scan over the 6 M point flags of a render ran at
// 1.3 TB/s).  `vec`: both arrays are 16-byte aligned (every workspace array is); otherwise element loads.
__device__ __forceinline__ int4 scan_load4(const int *__restrict__ in, int64_t idx, int64_t n, bool vec)
{
    if (vec && idx + 3 < n) return *reinterpret_cast<const int4 *>(in + idx);
    int4 v = make_int4(0, 0, 0, 0);
    if (idx + 0 < n) v.x = in[idx + 0];
    if (idx + 1 < n) v.y = in[idx + 1];
    if (idx + 2 < n) v.z = in[idx + 2];
    if (idx + 3 < n) v.w = in[idx + 3];
    return v;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_reduce(const int *__restrict__ in, int64_t n_max,
                                                               const int *__restrict__ n_dev,
                                                               int *__restrict__ block_sums, int vec)
{
    __shared__ int smem[SCAN_THREADS / 64 + 1];
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_max) : n_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wbase = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)wave * (SCAN_TILE / (SCAN_THREADS / 64));
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS / 4; ++k) {
        const int4 v = scan_load4(in, wbase + 4 * (64 * k + lane), n, vec != 0);
        s += (v.x + v.y) + (v.z + v.w);
    }
    int tot;
    block_excl_scan(s, &tot, smem);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_scan_down(const int *__restrict__ in, int *__restrict__ out,
                                                             int64_t n_max, const int *__restrict__ n_dev,
                                                             const int *__restrict__ block_sums, int nblocks, int vec,
                                                             int64_t *__restrict__ total64)
{
    __shared__ int smem[SCAN_THREADS / 64 + 1];
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_max) : n_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // this block's offset = the totals of the blocks in front of it (block 0 also forms the grand total: every block's)
    int block_base, grand_total;
    {
        const int upto = blockIdx.x == 0 ? nblocks : (int)blockIdx.x;
        int part = 0;
        for (int b = threadIdx.x; b < upto; b += SCAN_THREADS) part += block_sums[b];
        int tot;
        block_excl_scan(part, &tot, smem);
        block_base = blockIdx.x == 0 ? 0 : tot;
        grand_total = tot;     // (meaningful in block 0 only)
    }
    const int64_t wbase = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)wave * (SCAN_TILE / (SCAN_THREADS / 64));
    // element order inside the wavefront's 1024: k-major, then lane, then component
    int4 v[SCAN_ITEMS / 4];
    int incl[SCAN_ITEMS / 4], tot[SCAN_ITEMS / 4];
    int wsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS / 4; ++k) {
        v[k] = scan_load4(in, wbase + 4 * (64 * k + lane), n, vec != 0);
        const int g = (v[k].x + v[k].y) + (v[k].z + v[k].w);
        incl[k] = wave_incl_scan(g);
        tot[k] = __shfl(incl[k], 63, 64);
        incl[k] -= g;             // exclusive inside the group row
        wsum += tot[k];
    }
    if (lane == 0) smem[wave] = wsum;
    __syncthreads();
    int base = block_base;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w)
        if (w < wave) base += smem[w];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS / 4; ++k) {
        const int64_t idx = wbase + 4 * (64 * k + lane);
        int e = base + incl[k];
        int4 o;
        o.x = e;
        e += v[k].x;
        o.y = e;
        e += v[k].y;
        o.z = e;
        e += v[k].z;
        o.w = e;
        if (vec && idx + 3 < n) {
            *reinterpret_cast<int4 *>(out + idx) = o;
        } else {
            if (idx + 0 < n) out[idx + 0] = o.x;
            if (idx + 1 < n) out[idx + 1] = o.y;
            if (idx + 2 < n) out[idx + 2] = o.z;
            if (idx + 3 < n) out[idx + 3] = o.w;
        }
        base += tot[k];
    }
    // out[n] = the total
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[n] = grand_total;
        if (total64) *total64 = grand_total;
    }
}

size_t scan_temp_bytes(int64_t n_max)
{
    int64_t nblocks = (n_max + SCAN_TILE - 1) / SCAN_TILE;
    if (nblocks < 1) nblocks = 1;
    return ((size_t)(nblocks + 1) * sizeof(int) + 255) & ~(size_t)255;
}

int scan_exclusive_i32(const int *in, int *out, int64_t n_max, const int *n_dev, int64_t *total64, void *temp,
                       hipStream_t stream)
{
    int64_t nb = (n_max + SCAN_TILE - 1) / SCAN_TILE;
    if (nb < 1) nb = 1;
    int *sums = (int *)temp;
    const int vec = ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0;
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, stream, in, n_max, n_dev, sums, vec);
    hipLaunchKernelGGL(k_scan_down, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, stream, in, out, n_max, n_dev,
                       sums, (int)nb, vec, total64);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

}  // namespace pnr
