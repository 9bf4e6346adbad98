// pnr_optim.hip -- the optimiser half of the training step for the point tensors (include/pnr.h, "row-sparse Adam").
//
// The reference trains `neural_points` with Adam at lr 2e-3 (studio_config.py:41-47): torch's dense Adam sweeps all
// N x 38 point values every step (6 M points: 4.2 ms of a 6 ms step) although a 4096-ray batch gives ~60 k rows a
// gradient.  A row whose gradient has been zero since the start has exp_avg = exp_avg_sq = 0 and dense Adam moves it by
// -step_size * 0 / (0 + eps) = 0: Adam over the rows that EVER had a gradient is dense Adam, bit for bit in which rows
// move.  pnr_rows_merge keeps that set (a flag per row + an append-only list with a device-side count, no host read);
// pnr_adam_rows applies torch.optim.Adam's update (no amsgrad, no weight decay; torch/optim/adam.py, the foreach form:
// lerp, mul + addcmul, sqrt / bias_correction2_sqrt + eps, addcdiv) to the listed rows of up to PNR_ADAM_MAX_TENSORS
// tensors in one launch.  Purely HBM-bound: 7 floats of traffic per value (p, g, m, v in; p, m, v out).
#include <algorithm>

#include "pnr_internal.h"

namespace pnr {

struct AdamTensors {
    float *p[PNR_ADAM_MAX_TENSORS];
    const float *g[PNR_ADAM_MAX_TENSORS];
    float *m[PNR_ADAM_MAX_TENSORS];
    float *v[PNR_ADAM_MAX_TENSORS];
    int width[PNR_ADAM_MAX_TENSORS];
    int first[PNR_ADAM_MAX_TENSORS + 1];   // column range of tensor t in a listed row's virtual [total] columns
    int n;
    int total;
};

// One thread per (listed row, column) of the tensors' concatenated columns; consecutive threads walk a row's columns, so
// the 128 bytes of an embedding row are one coalesced access and a row's small tensors follow it.
__global__ void __launch_bounds__(256) k_adam_rows(AdamTensors T, int64_t num_rows, const int *__restrict__ rows,
                                                   int64_t rows_cap, const long long *__restrict__ n_dev, float w1,
                                                   float beta2, float w2, float step_size, float bc2_sqrt, float eps)
{
    int64_t n = rows_cap;
    if (n_dev) n = min((int64_t)*n_dev, rows_cap);
    const int64_t total = n * T.total;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t li = e / T.total;
        const int c = (int)(e - li * T.total);
        const int64_t r = rows ? (int64_t)rows[li] : li;
        if (r < 0 || r >= num_rows) continue;
        int t = 0;
#pragma unroll
        for (int k = 1; k < PNR_ADAM_MAX_TENSORS; ++k)
            if (k < T.n && c >= T.first[k]) t = k;
        const int64_t at = r * T.width[t] + (c - T.first[t]);
        const float g = T.g[t][at];
        float m = T.m[t][at], v = T.v[t][at];
        // torch.optim.Adam's foreach form, kernel by kernel, with the roundings its device code has (this file is built with
        // -ffp-contract=off, torch's kernels with the compiler's default contraction: every `a + s * x` there is one fma):
        //   _foreach_lerp_(exp_avg, grad, 1 - beta1)              a + w (b - a)            (weight < 0.5)
        //   _foreach_mul_(exp_avg_sq, beta2); _foreach_addcmul_(exp_avg_sq, grad, grad, 1 - beta2)      a + s (b c)
        //   sqrt; _foreach_div_(., bias_correction2_sqrt); _foreach_add_(., eps); _foreach_addcdiv_(param, exp_avg, ., -step_size)
        m = fmaf(w1, g - m, m);
        v = v * beta2;
        v = fmaf(w2, g * g, v);
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        T.m[t][at] = m;
        T.v[t][at] = v;
        T.p[t][at] = fmaf(-step_size, m / denom, T.p[t][at]);
    }
}

// rows[0 .. n) -> the ever-touched set: a row enters the list the first time it is seen (atomicExch on its flag; the
// position in the list is the order of arrival, which nothing depends on -- the update is per element).
__global__ void __launch_bounds__(256) k_rows_merge(int *__restrict__ flags, int64_t num_rows, int *__restrict__ ever,
                                                    long long *__restrict__ ever_count, int64_t ever_cap,
                                                    const int *__restrict__ rows, int64_t rows_cap,
                                                    const long long *__restrict__ n_dev)
{
    int64_t n = rows_cap;
    if (n_dev) n = min((int64_t)*n_dev, rows_cap);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = rows[i];
        if (r < 0 || r >= num_rows) continue;
        if (atomicExch(&flags[r], 1) == 0) {
            const long long pos = (long long)atomicAdd(reinterpret_cast<unsigned long long *>(ever_count), 1ull);
            if (pos < ever_cap) ever[pos] = (int)r;
        }
    }
}

}  // namespace pnr

using namespace pnr;

extern "C" int pnr_rows_merge(int32_t *d_flags, int64_t num_rows, int32_t *d_ever, int64_t *d_ever_count, int64_t ever_cap,
                              const int32_t *d_rows, int64_t rows_cap, const int64_t *d_n_rows, void *stream_)
{
    PNR_REQUIRE(d_flags && d_ever && d_ever_count && d_rows, "pnr_rows_merge: null pointer");
    PNR_REQUIRE(num_rows >= 1 && num_rows < (int64_t)0x7FFFFFFF, "pnr_rows_merge: num_rows=%lld out of range",
                (long long)num_rows);
    // (every row can enter once: a list of num_rows entries can never overflow)
    PNR_REQUIRE(ever_cap >= num_rows, "pnr_rows_merge: ever_cap=%lld < num_rows=%lld", (long long)ever_cap,
                (long long)num_rows);
    PNR_REQUIRE(rows_cap >= 0 && rows_cap < (int64_t)0x7FFFFFFF, "pnr_rows_merge: rows_cap=%lld out of range",
                (long long)rows_cap);
    if (rows_cap == 0) return PNR_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((rows_cap + 255) / 256, 1024);
    hipLaunchKernelGGL(k_rows_merge, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, d_flags, num_rows, d_ever,
                       reinterpret_cast<long long *>(d_ever_count), ever_cap, d_rows, rows_cap,
                       reinterpret_cast<const long long *>(d_n_rows));
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

extern "C" int pnr_adam_rows(const pnr_adam_tensor_t *tensors, int32_t n_tensors, int64_t num_rows, const int32_t *d_rows,
                             int64_t rows_cap, const int64_t *d_n_rows, double beta1, double beta2, double eps,
                             double step_size, double bias_correction2_sqrt, void *stream_)
{
    PNR_REQUIRE(tensors != nullptr && n_tensors >= 1 && n_tensors <= PNR_ADAM_MAX_TENSORS,
                "pnr_adam_rows: n_tensors=%d not in [1,%d]", n_tensors, PNR_ADAM_MAX_TENSORS);
    PNR_REQUIRE(num_rows >= 1 && num_rows < (int64_t)0x7FFFFFFF, "pnr_adam_rows: num_rows=%lld out of range",
                (long long)num_rows);
    PNR_REQUIRE(d_rows != nullptr || (rows_cap == num_rows && d_n_rows == nullptr),
                "pnr_adam_rows: d_rows == NULL means every row: rows_cap must be num_rows and d_n_rows NULL");
    PNR_REQUIRE(rows_cap >= 0 && rows_cap <= num_rows, "pnr_adam_rows: rows_cap=%lld not in [0, num_rows]",
                (long long)rows_cap);
    PNR_REQUIRE(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0. && bias_correction2_sqrt > 0.,
                "pnr_adam_rows: betas (%g, %g) / eps %g / bias_correction2_sqrt %g out of range", beta1, beta2, eps,
                bias_correction2_sqrt);
    // torch's lerp takes the a + w (b - a) form for weights below one half only
    PNR_REQUIRE(beta1 > 0.5, "pnr_adam_rows: beta1=%g <= 0.5 is not supported", beta1);
    AdamTensors T{};
    T.n = n_tensors;
    int col = 0;
    for (int t = 0; t < n_tensors; ++t) {
        const pnr_adam_tensor_t &a = tensors[t];
        PNR_REQUIRE(a.d_param && a.d_grad && a.d_exp_avg && a.d_exp_avg_sq, "pnr_adam_rows: tensor %d has a null pointer", t);
        PNR_REQUIRE(a.width >= 1 && a.width <= 4096, "pnr_adam_rows: tensor %d width=%d out of range", t, a.width);
        T.p[t] = a.d_param;
        T.g[t] = a.d_grad;
        T.m[t] = a.d_exp_avg;
        T.v[t] = a.d_exp_avg_sq;
        T.width[t] = a.width;
        T.first[t] = col;
        col += a.width;
    }
    for (int t = n_tensors; t <= PNR_ADAM_MAX_TENSORS; ++t) T.first[t] = col;
    T.total = col;
    if (rows_cap == 0) return PNR_OK;
    // the grid covers the CAPACITY; the threads beyond the device-side count leave at their first loop test
    const unsigned blocks = (unsigned)std::min<int64_t>((rows_cap * T.total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_adam_rows, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, T, num_rows, d_rows, rows_cap,
                       reinterpret_cast<const long long *>(d_n_rows),
                       // the scalars as torch hands them to its kernels: evaluated in double on the host, then one cast
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)step_size,
                       (float)bias_correction2_sqrt, (float)eps);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}
