// Shade stage, host side: weight packing (PyTorch [out,in] -> MFMA operand order), the early-termination passes and
// launch_shade.  Kernels: pnr_shade_fp32.hip, pnr_shade_bf16.hip; overview: pnr_shade_common.h.
#include <mutex>
#include <stdlib.h>

#include "pnr_shade_common.h"

namespace pnr {

int ensure_dynamic_lds(const void *kernel, int bytes, int *cus_out)
{
    struct Entry { const void *kernel; int dev; };
    static Entry done[256];
    static int n_done = 0;
    static int cus_of[64] = {0};
    static std::mutex mu;
    int dev = 0;
    PNR_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (cus_of[dev & 63] == 0) {
        int n = 0;
        PNR_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        cus_of[dev & 63] = n > 0 ? n : 256;
    }
    if (cus_out) *cus_out = cus_of[dev & 63];
    for (int i = 0; i < n_done; ++i)
        if (done[i].kernel == kernel && done[i].dev == dev) return PNR_OK;
    PNR_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (n_done < 256) done[n_done++] = Entry{kernel, dev};   // (beyond the table: set again on every call, still correct)
    return PNR_OK;
}

// ------------------------------------------------------------------------------------------------
enum LayerKind { L_BASE0 = 0, L_HIDDEN = 1, L_HEAD0 = 2, L_COLOR0 = 3 };

__device__ __forceinline__ int hidden_feat(int t, int h)
{
    const int m = t >> 4, r = t & 15;
    return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// fp32 path: input feature of k-step t for lane half h
__device__ int feat_of(int kind, int t, int h, int n_in)
{
    switch (kind) {
    case L_BASE0:
        if (t < 16) return 16 * h + t;
        if (t < 112) {
            const int u = t - 16, sc = u & 1, df = u >> 1, d = df / 3 + 16 * h, f = df % 3;
            return 32 + 2 * (d * 3 + f) + sc;
        }
        if (t < 142) {
            const int u = t - 112, sc = u & 1, df = u >> 1, d = df / 5 + 3 * h, f = df % 5;
            return 224 + 2 * (d * 5 + f) + sc;
        }
        return -1;
    case L_HIDDEN:
        return t < n_in / 2 ? hidden_feat(t, h) : -1;
    case L_HEAD0:
        if (t < 128) return hidden_feat(t, h);
        if (t == 128) return h ? 257 : 256;
        if (t == 129) return h ? 259 : 258;
        if (t == 130) return h ? 261 : 260;
        if (t == 131) return h ? -1 : 262;
        return -1;
    case L_COLOR0:
        if (t < 128) return 8 * (t >> 2) + 4 * h + (t & 3);
        if (t < 140) return (h ? 268 : 256) + (t - 128);
        return -1;
    }
    return -1;
}

// bf16 path: input feature of element j of k-step s (16 features) for lane half h.  Hidden layers: the
// accumulator registers 8s'..8s'+7 of output tile m ARE k-step 2m+s' (row = 16s' + 8(j>>2) + 4h + (j&3)).
__device__ int feat16_of(int kind, int s, int h, int j, int n_in)
{
    const int hid = 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
    switch (kind) {
    case L_BASE0:
        return feat_of(L_BASE0, 8 * s + j, h, n_in);  // the lane's value i = 8s + j, same order as the fp32 path
    case L_HIDDEN:
        return s < n_in / 16 ? hid : -1;
    case L_HEAD0:
        if (s < 16) return hid;
        if (s == 16 && j < 4) {
            const int lo[4] = {256, 258, 260, 262}, hi[4] = {257, 259, 261, -1};
            return h ? hi[j] : lo[j];
        }
        return -1;
    case L_COLOR0: {
        if (s < 16) return 16 * s + 8 * h + j;
        const int f = 256 + (s - 16) * 16 + 8 * h + j;
        return f < 280 ? f : -1;
    }
    }
    return -1;
}

__device__ __forceinline__ void pack_layer_body(const float *__restrict__ W, int n_out, int n_in, int kind, int ksp,
                                                int t0, float *__restrict__ dst)
{
    // dst[((m*KG + g)*64 + lane)*4 + q] = W[32m + (lane&31)][feat(t0 + 4g+q, lane>>5)]
    const int kg = ksp / 4, mt = n_out / 32;
    const int64_t total = (int64_t)mt * kg * 64 * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i & 3), lane = (int)((i >> 2) & 63);
        const int64_t mg = i >> 8;
        const int g = (int)(mg % kg), m = (int)(mg / kg);
        const int f = feat_of(kind, t0 + 4 * g + q, lane >> 5, n_in);
        dst[i] = (f >= 0 && f < n_in) ? W[(int64_t)(32 * m + (lane & 31)) * n_in + f] : 0.f;
    }
}

__global__ void k_pack_layer(const float *__restrict__ W, int n_out, int n_in, int kind, int ksp, int t0,
                             float *__restrict__ dst)
{
    pack_layer_body(W, n_out, n_in, kind, ksp, t0, dst);
}

__device__ __forceinline__ void pack_layer_bf16_body(const float *__restrict__ W, int n_out, int n_in, int kind, int ks,
                                                     int kx, int s0, unsigned short *__restrict__ dst)
{
    // dst[(((m*KS + s)*2 + plane)*64 + lane)*8 + j] = {hi, lo}(W[32 rb + (lane&31)][feat16(s0 + s % kx, lane>>5, j)])
    // a ring tile m holds ks / kx row blocks rb = m * (ks / kx) + s / kx of kx k-steps each (kx == ks: one)
    const int mt = n_out / 32 / (ks / kx);
    const int64_t total = (int64_t)mt * ks * 64 * 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const int64_t ms = i >> 9;
        const int s = (int)(ms % ks), m = (int)(ms / ks);
        const int f = feat16_of(kind, s0 + s % kx, lane >> 5, j, n_in);
        const int rb = m * (ks / kx) + s / kx;
        const float w = (f >= 0 && f < n_in) ? W[(int64_t)(32 * rb + (lane & 31)) * n_in + f] : 0.f;
        const __bf16 hb = (__bf16)w;
        const __bf16 lb = (__bf16)(w - (float)hb);
        const int64_t base = ((ms * 2) * 64 + lane) * 8 + j;
        dst[base] = __builtin_bit_cast(unsigned short, hb);
        dst[base + 64 * 8] = __builtin_bit_cast(unsigned short, lb);
    }
}

__global__ void k_pack_layer_bf16(const float *__restrict__ W, int n_out, int n_in, int kind, int ks, int kx, int s0,
                                  unsigned short *__restrict__ dst)
{
    pack_layer_bf16_body(W, n_out, n_in, kind, ks, kx, s0, dst);
}

// density head weights in accumulator order: dst[(t * 2 + h) * 16 + r] = w4[32t + 8(r>>2) + 4h + (r&3)]
__global__ void k_pack_head_acc(const float *__restrict__ w4, float *__restrict__ dst)
{
    const int i = threadIdx.x, t = i >> 5, h = (i >> 4) & 1, r = i & 15;
    dst[i] = w4[32 * t + 8 * (r >> 2) + 4 * h + (r & 3)];
}

// colour head weights [3,128] in accumulator order: dst[((c * 4 + t) * 2 + h) * 16 + r] = w8[c][32t + 8(r>>2) + 4h + (r&3)]
__global__ void k_pack_color_head_acc(const float *__restrict__ w8, float *__restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 384) return;
    const int c = i >> 7, t = (i >> 5) & 3, h = (i >> 4) & 1, r = i & 15;
    dst[i] = w8[c * 128 + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3)];
}

__global__ void k_copy(const float *__restrict__ src, int n, float *__restrict__ dst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// Every packed form of pnr_weights_update in ONE launch: blockIdx.y selects the job (a training loop re-packs after
// every optimiser step; thirty small launches cost more than the packing).
enum { JOB_LAYER = 0, JOB_LAYER16 = 1, JOB_COPY = 2, JOB_HEAD_ACC = 3, JOB_COLOR_ACC = 4 };
struct PackJob {
    int type, n_out, n_in, kind, a, b, c;
    const float *src;
    void *dst;
};
constexpr int MAX_PACK_JOBS = 40;
struct PackJobs {
    PackJob j[MAX_PACK_JOBS];
};
__global__ void __launch_bounds__(256) k_pack_jobs(PackJobs jobs)
{
    const PackJob &J = jobs.j[blockIdx.y];
    switch (J.type) {
    case JOB_LAYER:
        pack_layer_body(J.src, J.n_out, J.n_in, J.kind, J.a, J.b, (float *)J.dst);
        break;
    case JOB_LAYER16:
        pack_layer_bf16_body(J.src, J.n_out, J.n_in, J.kind, J.a, J.b, J.c, (unsigned short *)J.dst);
        break;
    case JOB_COPY:
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < J.a; i += gridDim.x * blockDim.x)
            ((float *)J.dst)[i] = J.src[i];
        break;
    case JOB_HEAD_ACC:
        if (blockIdx.x == 0) {
            const int i = threadIdx.x, t = i >> 5, h = (i >> 4) & 1, r = i & 15;
            ((float *)J.dst)[i] = J.src[32 * t + 8 * (r >> 2) + 4 * h + (r & 3)];
        }
        break;
    default:
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 384; i += gridDim.x * blockDim.x) {
            const int c = i >> 7, t = (i >> 5) & 3, h = (i >> 4) & 1, r = i & 15;
            ((float *)J.dst)[i] = J.src[c * 128 + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3)];
        }
        break;
    }
}

// ------------------------------------------------------------------------------------------------
// early ray termination (opts.early_stop_eps > 0): samples are shaded front to back in chunks of a ray's sample
// index; between the chunks a ray whose transmittance fell below eps leaves.  Everything stays on the device.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) k_pass_init(int64_t R, float *__restrict__ ray_T, int *__restrict__ ray_alive,
                                                   int *__restrict__ n_sel)
{
    const int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (r == 0) {
        n_sel[4] = 0;
        n_sel[5] = 0;
        n_sel[6] = 0;
    }
    if (r >= R) return;
    ray_T[r] = 1.0f;
    ray_alive[r] = 1;
}

// flag[v] = valid sample v belongs to a live ray and its index inside the ray is in [lo, hi)
__global__ void __launch_bounds__(TPB) k_pass_flag(const int *__restrict__ n_sel, const int *__restrict__ vs_list,
                                                   const int *__restrict__ smp_ray, const int *__restrict__ ray_off,
                                                   const int *__restrict__ ray_alive, int lo, int hi,
                                                   int *__restrict__ flag)
{
    const int S_valid = n_sel[1];
    for (int64_t v = (int64_t)blockIdx.x * TPB + threadIdx.x; v < S_valid; v += (int64_t)gridDim.x * TPB) {
        const int s = vs_list[v];
        const int r = smp_ray[s];
        const int i = s - ray_off[r];
        flag[v] = (ray_alive[r] && i >= lo && i < hi) ? 1 : 0;
    }
}

__global__ void __launch_bounds__(TPB) k_pass_scatter(const int *__restrict__ n_sel, const int *__restrict__ vs_list,
                                                      const int *__restrict__ flag, const int *__restrict__ pos,
                                                      int *__restrict__ vs_all)
{
    const int S_valid = n_sel[1], base = n_sel[5];
    for (int64_t v = (int64_t)blockIdx.x * TPB + threadIdx.x; v < S_valid; v += (int64_t)gridDim.x * TPB)
        if (flag[v]) vs_all[base + pos[v]] = vs_list[v];
}

// the pass just listed occupies positions [n_sel[4], n_sel[5]) of vs_all
__global__ void k_pass_advance(int *__restrict__ n_sel, const int *__restrict__ pos)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int count = pos[n_sel[1]];
        n_sel[4] = n_sel[5];
        n_sel[5] = n_sel[5] + count;
    }
}

// transmittance of the live rays after samples [lo, hi): the composite's own arithmetic (k_composite, same
// ray_dist quirks), so that "T < eps" means what it means there
__global__ void __launch_bounds__(TPB) k_pass_update(CamRef cr, pnr_render_opts_t opts, int64_t R,
                                                     const int *__restrict__ ray_cnt, const int *__restrict__ ray_off,
                                                     const float4 *__restrict__ smp_loc,
                                                     const float *__restrict__ smp_sig_s, const int *__restrict__ n_sel,
                                                     int lo, int hi, float *__restrict__ ray_T,
                                                     float *__restrict__ ray_cm, int *__restrict__ ray_alive)
{
    const int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (r >= R || !ray_alive[r]) return;
    const int S = n_sel[0];
    const int off = ray_off[r];
    int cnt = ray_cnt[r];
    if ((int64_t)off + cnt > S) cnt = max(0, S - off);
    if (lo >= cnt) {
        ray_alive[r] = 0;
        return;
    }
    const Camera cam = load_cam_lanes(cr, cam_id(cr, r));
    const float vs = opts.vsize_z, two_vs = 2.0f * vs;
    auto zc = [&](float x, float y, float z) {
        const float sx = x - cam.o[0], sy = y - cam.o[1], sz = z - cam.o[2];
        return sx * cam.R[2] + sy * cam.R[5] + sz * cam.R[8];
    };
    const float z_unfilled = zc(0.f, 0.f, 0.f);
    float cm;
    if (lo == 0) {
        const float4 p = smp_loc[off];
        cm = zc(p.x, p.y, p.z);
    } else {
        cm = ray_cm[r];
    }
    float T = ray_T[r];
    const int end = min(hi, cnt);
    for (int i = lo; i < end; ++i) {
        float delta;
        if (i == opts.SR - 1) {
            delta = vs;
        } else {
            float z_next = z_unfilled;
            if (i + 1 < cnt) {
                const float4 p = smp_loc[off + i + 1];
                z_next = zc(p.x, p.y, p.z);
            }
            const float cm_next = fmaxf(cm, z_next);
            delta = cm_next - cm;
            cm = cm_next;
            if (delta < 1e-8f || delta > two_vs) delta = vs;
        }
        const float opacity = 1.0f - expf(-smp_sig_s[off + i] * delta);
        T = T * (1.0f - opacity + 1e-10f);
    }
    ray_T[r] = T;
    ray_cm[r] = cm;
    ray_alive[r] = (T > opts.early_stop_eps && hi < cnt) ? 1 : 0;
}

// Normalised inverse-distance weights of every neighbour slot of the samples [n_sel[i0], n_sel[i1]) of `list`
// (studio_model.py:285-286,467-475): w_k = mask_k / clamp(||p_k - s||, 1e-6), divided by clamp(sum_k w_k, 1e-8).  One
// thread per sample, the K slots in order.  For the pair kernel on dense units, whose rows of a sample may sit in
// two tiles: it reads its row's weight instead of summing over the sample's rows.
__global__ void __launch_bounds__(256) k_pair_weights(const int *__restrict__ n_sel, int i0, int i1,
                                                      const int *__restrict__ list, const int *__restrict__ smp_pidx,
                                                      const float4 *__restrict__ smp_loc,
                                                      const float4 *__restrict__ point_rows, int K,
                                                      float *__restrict__ smp_wgt)
{
    const int v0 = n_sel[i0], v1 = n_sel[i1];
    for (int v = v0 + blockIdx.x * 256 + threadIdx.x; v < v1; v += gridDim.x * 256) {
        const int s = list[v];
        const float4 loc = smp_loc[s];
        const int *pid = smp_pidx + (int64_t)s * K;
        float *out = smp_wgt + (int64_t)s * K;
        float wsum = 0.f;
        for (int k = 0; k < K; ++k) {
            const int p = pid[k];
            float w = 0.f;
            if (p >= 0) {
                const float4 a0 = point_rows[(int64_t)p * 12];
                const float dx = a0.x - loc.x, dy = a0.y - loc.y, dz = a0.z - loc.z;
                w = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-6f);
            }
            out[k] = w;
            wsum += w;
        }
        const float den = fmaxf(wsum, 1e-8f);
        for (int k = 0; k < K; ++k) out[k] = out[k] / den;
    }
}

__global__ void k_publish_shaded(const int *__restrict__ n_sel, int idx, int64_t *__restrict__ counters)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) counters[PNR_CNT_SAMPLES_SHADED] = n_sel[idx];
}

int launch_shade(const pnr_scene *scene, const pnr_weights *w, const CamRef &cr, const float *d_dirs,
                 const pnr_render_opts_t &opts, int64_t R, RenderWs &ws, int64_t cap, int64_t *d_counters,
                 hipStream_t stream, hipEvent_t ev_points, hipEvent_t ev_between)
{
    const int K = opts.K, precision = opts.precision;
    const bool early = opts.early_stop_eps > 0.f;
    ShadeParams P{};
    P.point_rows = reinterpret_cast<const float4 *>(scene->point_rows);
    P.wbuf = w->buf;
    P.wbytes = w->bytes;
    for (int i = 0; i < 9; ++i) {
        P.w_off[i] = w->w_off[i];
        P.w16_off[i] = w->w16_off[i];
        P.b_off[i] = w->b_off[i];
        P.Rw2c[i] = w->Rw2c[i];
    }
    P.cr = cr;
    P.dirs = d_dirs;
    P.smp_loc = ws.smp_loc;
    P.smp_ray = ws.smp_ray;
    P.smp_pidx = ws.smp_pidx;
    P.vs_list = early ? ws.vs_all : ws.vs_list;
    P.smp_wgt = ws.smp_wgt;
    P.n_sel = ws.n_sel;
    P.i_v0 = early ? 4 : 6;  // n_sel[6] == 0
    P.i_v1 = early ? 5 : 1;
    P.smp_sig_s = early ? ws.smp_sig_s : nullptr;
    P.smp_sigma = ws.smp_sigma;
    P.agg = ws.agg;
    P.smp_out = ws.smp_out;
    P.K = K;
    P.w16a_off = w->w16a_off;
    P.w16b_off = w->w16b_off;
    P.w4acc_off = w->w4acc_off;
    P.w8acc_off = w->w8acc_off;
    P.w32a_off = w->w32a_off;
    P.w32b_off = w->w32b_off;
    P.pt_rank = ws.pt_rank;
    P.pt_list = ws.pt_list;
    P.pt_table = reinterpret_cast<float4 *>(ws.pt_table);
    P.u_cap = (int)ws.u_cap;
    // a training render (opts.d_tape: the training workspace of the pnr_render_backward that follows) writes the
    // backward's activation tape as it shades; fp32, K <= 16 on the DPP kernels, no early termination
    bool taped = false;
    if (tape_supported(opts) && !early) {
        float *t4[4];
        size_t b4[4];
        if (train_tape_ptrs(opts.d_tape, opts.tape_bytes, cap, K, t4, b4, &P.tape_bits, &P.tape_bits_rows, &P.tape_rowz, P.ctape,
                            reinterpret_cast<void **>(&P.tape_sg))) {
            for (int l = 0; l < 4; ++l) {
                P.tape[l] = t4[l];
                P.tape_bytes[l] = b4[l];
            }
            taped = true;
        }
    }
    int dev = 0, cus = 256;
    PNR_HIP_CHECK(hipGetDevice(&dev));
    PNR_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    // decoded features of samples without neighbours (or not shaded) are zero (studio_model.py:361-362)
    if (!ws.out_cleared) {   // (a render's first launch did it: launch_select_expand)
        const int rcz = zero_async(ws.smp_out, (size_t)cap * sizeof(float4), stream);
        if (rcz != PNR_OK) return rcz;
    }
    const int seg = K <= 8 ? 8 : (K <= 16 ? 16 : 0);   // lanes per sample segment (0: exactly K)
    const bool bf = precision == PNR_PRECISION_BF16X3;
    // fp32, K that does not fill its segment: dense units -- su samples fill tu tiles of a wave (K = 12: 8 samples in 3
    // tiles, no padding); the best fill among 1..3 tiles, at most 16 samples per unit
    int su = 0, tu = 0;
#ifndef PNR_NO_DENSE_UNITS
    // (PNR_DENSE_UNITS=0 in the environment: the segment form, for A/B runs on one device)
    static const int dense_mode = [] {
        const char *e = getenv("PNR_DENSE_UNITS");
        return e ? atoi(e) : 1;     // 0: never, 1: where it fills more rows, 2: also for K = 8 and 16
    }();
    // (a render asked for the tape keeps the DPP kernels, the only ones that write it: pnr_render_backward derives "taped"
    // from the same opts -- tape_supported -- and must find what it expects)
    const bool dense_all = dense_mode == 2 && (K == 8 || K == 16) && !taped;
    if (!bf && ((K >= 11 && K != 16 && dense_mode != 0) || dense_all)) {   // (K >= 11, or K = 8 / 16 aligned to the
        // tiles: at most four samples touch a tile, see the kernel)
        double best = (double)K / (seg ? seg : K * (32 / K)) * 1.0001;   // what the segment form fills
        if (!seg) best = (double)(32 / K) * K / 32.0 * 1.0001;
        if (dense_all) best = 0.0;
        for (int t = 1; t <= (dense_all ? 1 : 3); ++t) {
            const int s_ = std::min(32 * t / K, 16);
            const double fill = (double)s_ * K / (32.0 * t);
            if (s_ >= 1 && fill > best) {
                best = fill;
                su = s_;
                tu = t;
            }
        }
    }
#endif
    const int spt = su ? su * WAVES : (32 / (seg ? seg : K)) * WAVES;
    const int64_t max_tiles = (cap + spt - 1) / spt;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cus, max_tiles));
    {
        if (!ws.pt_table) {
            set_error("launch_shade: the workspace lacks the point-part buffers (pnr_render_workspace_bytes_for)");
            return PNR_ERR_WORKSPACE;
        }
        const int64_t ptiles = (ws.u_cap + 32 * WAVES - 1) / (32 * WAVES);
        const dim3 pgrid((unsigned)std::max<int64_t>(1, std::min<int64_t>(cus, ptiles)));
        if (bf)
            launch_point_part_bf16(pgrid, stream, P);
        else
            launch_point_part_fp32(pgrid, stream, P);
    }
    if (ev_points) PNR_HIP_CHECK(hipEventRecord(ev_points, stream));
    auto launch_pairs = [&]() -> int {
        if (bf)
            launch_pairs_bf16(seg, dim3(grid), stream, P);
        else if (taped && !su && launch_pairs_fp32_tape(seg, dim3(grid), stream, P)) {
        } else if (su) {
            // the rows' weights: left by the neighbour search (k_knn3<16, true>, the full-frame form for K = 11..15), else
            // by a pass of their own
            if (!ws.wgt_from_knn)
                hipLaunchKernelGGL(k_pair_weights, dim3((unsigned)std::min<int64_t>((cap + 255) / 256, 256 * 16)), dim3(256),
                                   0, stream, ws.n_sel, P.i_v0, P.i_v1, P.vs_list, ws.smp_pidx, ws.smp_loc, P.point_rows, K,
                                   ws.smp_wgt);
            {
                const int rcd = launch_pairs_fp32_dense(su, tu, dim3(grid), stream, P);
                if (rcd != PNR_OK) return rcd;
            }
        }
        else
            launch_pairs_fp32(seg, dim3(grid), stream, P);
        return PNR_OK;
    };
    if (!early) {
        const int rcp = launch_pairs();
        if (rcp != PNR_OK) return rcp;
    } else {
        const unsigned rgrid = (unsigned)((R + TPB - 1) / TPB);
        const unsigned sgrid = (unsigned)std::min<int64_t>((cap + TPB - 1) / TPB, 256 * 32);
        {
            const int rcz = zero_async(ws.smp_sig_s, (size_t)cap * sizeof(float), stream);
            if (rcz != PNR_OK) return rcz;
        }
        hipLaunchKernelGGL(k_pass_init, dim3(rgrid), dim3(TPB), 0, stream, R, ws.ray_T, ws.ray_alive, ws.n_sel);
        int *flag = ws.smp_valid, *pos = ws.smp_voff;  // free again once the valid samples are listed
#ifndef PNR_ES_BOUNDS
#define PNR_ES_BOUNDS 0, 3, 6, 12, 24
#endif
        static const int bounds[] = {PNR_ES_BOUNDS};
        constexpr int NP = (int)(sizeof(bounds) / sizeof(bounds[0]));
        for (int p = 0; p < NP; ++p) {
            const int lo = bounds[p], hi = (p + 1 < NP) ? std::min(bounds[p + 1], opts.SR) : opts.SR;
            if (lo >= opts.SR) break;
            hipLaunchKernelGGL(k_pass_flag, dim3(sgrid), dim3(TPB), 0, stream, ws.n_sel, ws.vs_list, ws.smp_ray,
                               ws.ray_off, ws.ray_alive, lo, hi, flag);
            int rc = scan_exclusive_i32(flag, pos, cap, ws.n_sel + 1, nullptr, ws.scan_temp, stream);
            if (rc != PNR_OK) return rc;
            hipLaunchKernelGGL(k_pass_scatter, dim3(sgrid), dim3(TPB), 0, stream, ws.n_sel, ws.vs_list, flag, pos,
                               ws.vs_all);
            hipLaunchKernelGGL(k_pass_advance, dim3(1), dim3(64), 0, stream, ws.n_sel, pos);
            rc = launch_pairs();
            if (rc != PNR_OK) return rc;
            hipLaunchKernelGGL(k_pass_update, dim3(rgrid), dim3(TPB), 0, stream, cr, opts, R, ws.ray_cnt, ws.ray_off,
                               ws.smp_loc, ws.smp_sig_s, ws.n_sel, lo, hi, ws.ray_T, ws.ray_cm, ws.ray_alive);
        }
        P.i_v0 = 6;  // the colour MLP runs once over everything that was shaded
    }
    hipLaunchKernelGGL(k_publish_shaded, dim3(1), dim3(64), 0, stream, ws.n_sel, P.i_v1, d_counters);
    if (ev_between) PNR_HIP_CHECK(hipEventRecord(ev_between, stream));
    const int64_t ctiles = (cap + 32 * WAVES - 1) / (32 * WAVES);
    const unsigned cgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cus, ctiles));
    if (bf)
        launch_color_bf16(dim3(cgrid), stream, P);
    else
        launch_color_fp32(dim3(cgrid), stream, P);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}

}  // namespace pnr

using namespace pnr;

extern "C" int pnr_weights_create(pnr_weights_t **out)
{
    PNR_REQUIRE(out != nullptr, "pnr_weights_create: out is null");
    *out = new pnr_weights();
    return PNR_OK;
}

extern "C" int pnr_weights_destroy(pnr_weights_t *w)
{
    if (!w) return PNR_OK;
    if (w->buf) (void)hipFree(w->buf);
    delete w;
    return PNR_OK;
}

extern "C" int pnr_weights_pack(pnr_weights_t *w, const float *const d_w[9], const float *const d_b[9],
                                const float *d_Rw2c, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    PNR_REQUIRE(w && d_w && d_b && d_Rw2c, "pnr_weights_pack: null argument");
    for (int i = 0; i < 9; ++i) PNR_REQUIRE(d_w[i] && d_b[i], "pnr_weights_pack: tensor %d is null", i);
    static const int n_out[9] = {256, 256, 256, 256, 1, 128, 128, 128, 3};
    static const int n_in[9] = {284, 256, 263, 256, 256, 280, 128, 128, 128};
    static const int ksp[9] = {144, 128, 132, 128, 0, 140, 64, 64, 0};    // fp32 k-steps (2 features each)
    static const int ks16[9] = {18, 16, 17, 16, 0, 18, 8, 8, 0};          // bf16 k-steps (16 features each)
    static const int kind[9] = {L_BASE0, L_HIDDEN, L_HEAD0, L_HIDDEN, -1, L_COLOR0, L_HIDDEN, L_HIDDEN, -1};
    size_t off = 0;
    auto pad = [](size_t n) { return (n + 1023) / 1024 * 1024; };  // 4 KiB granules (LDS staging rounds)
    for (int i = 0; i < 9; ++i) {
        w->w_off[i] = off;
        off += pad(ksp[i] ? (size_t)(n_out[i] / 32) * (ksp[i] / 4) * 256 : (size_t)n_out[i] * n_in[i]);
    }
    for (int i = 0; i < 9; ++i) {
        w->w16_off[i] = off;
        // [mt][ks][2 planes][64 lanes][8 bf16] = mt * ks * 2048 bytes = mt * ks * 512 floats
        off += ks16[i] ? pad((size_t)(n_out[i] / 32) * ks16[i] * 512) : 0;
    }
    // mlp_base layer 0 split for the bf16x3 mode: point-only k-steps 0..13 and pair k-steps 14..17
    w->w16a_off = off;
    off += pad((size_t)8 * 14 * 512);
    w->w16b_off = off;
    off += pad((size_t)4 * 8 * 512);
    w->w32a_off = off;
    off += pad((size_t)8 * (112 / 4) * 256);
    w->w32b_off = off;
    off += pad((size_t)8 * (32 / 4) * 256);
    w->w4acc_off = off;
    off += pad(256);
    w->w8acc_off = off;
    off += pad(512);
    for (int i = 0; i < 9; ++i) {
        w->b_off[i] = off;
        off += pad((size_t)n_out[i]);
    }
    off += 2 * 1024 * 9;  // tail pad: a staged tile may read up to 36 KiB from its start
    if (!w->buf || w->bytes != off * sizeof(float)) {
        if (w->buf) (void)hipFree(w->buf);
        w->buf = nullptr;
        PNR_HIP_CHECK(hipMalloc((void **)&w->buf, off * sizeof(float)));
        w->bytes = off * sizeof(float);
    }
    PNR_HIP_CHECK(hipMemsetAsync(w->buf, 0, w->bytes, stream));
    for (int i = 0; i < 9; ++i) {
        if (ksp[i]) {
            hipLaunchKernelGGL(k_pack_layer, dim3(256), dim3(256), 0, stream, d_w[i], n_out[i], n_in[i], kind[i],
                               ksp[i], 0, w->buf + w->w_off[i]);
            hipLaunchKernelGGL(k_pack_layer_bf16, dim3(256), dim3(256), 0, stream, d_w[i], n_out[i], n_in[i], kind[i],
                               ks16[i], ks16[i], 0, reinterpret_cast<unsigned short *>(w->buf + w->w16_off[i]));
        } else {
            int n = n_out[i] * n_in[i];
            hipLaunchKernelGGL(k_copy, dim3((n + 255) / 256), dim3(256), 0, stream, d_w[i], n, w->buf + w->w_off[i]);
        }
        hipLaunchKernelGGL(k_copy, dim3((n_out[i] + 255) / 256), dim3(256), 0, stream, d_b[i], n_out[i],
                           w->buf + w->b_off[i]);
    }
    hipLaunchKernelGGL(k_pack_layer_bf16, dim3(256), dim3(256), 0, stream, d_w[0], 256, 284, (int)L_BASE0, 14, 14, 0,
                       reinterpret_cast<unsigned short *>(w->buf + w->w16a_off));
    hipLaunchKernelGGL(k_pack_layer, dim3(256), dim3(256), 0, stream, d_w[0], 256, 284, (int)L_BASE0, 112, 0,
                       w->buf + w->w32a_off);
    hipLaunchKernelGGL(k_pack_layer, dim3(256), dim3(256), 0, stream, d_w[0], 256, 284, (int)L_BASE0, 32, 112,
                       w->buf + w->w32b_off);
    hipLaunchKernelGGL(k_pack_head_acc, dim3(1), dim3(256), 0, stream, d_w[4], w->buf + w->w4acc_off);
    hipLaunchKernelGGL(k_pack_color_head_acc, dim3(2), dim3(256), 0, stream, d_w[8], w->buf + w->w8acc_off);
    hipLaunchKernelGGL(k_pack_layer_bf16, dim3(256), dim3(256), 0, stream, d_w[0], 256, 284, (int)L_BASE0, 8, 4, 14,
                       reinterpret_cast<unsigned short *>(w->buf + w->w16b_off));
    PNR_HIP_CHECK(hipGetLastError());
    PNR_HIP_CHECK(hipMemcpyAsync(w->Rw2c, d_Rw2c, 9 * sizeof(float), hipMemcpyDeviceToHost, stream));
    PNR_HIP_CHECK(hipStreamSynchronize(stream));
    w->packed = true;
    return PNR_OK;
}

extern "C" int pnr_weights_update(pnr_weights_t *w, const float *const d_w[9], const float *const d_b[9],
                                  int32_t precision, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    PNR_REQUIRE(w && d_w && d_b, "pnr_weights_update: null argument");
    for (int i = 0; i < 9; ++i) PNR_REQUIRE(d_w[i] && d_b[i], "pnr_weights_update: tensor %d is null", i);
    if (!w->packed) {
        set_error("pnr_weights_update: pack once with pnr_weights_pack (layout, padding and Rw2c come from there)");
        return PNR_ERR_STATE;
    }
    PNR_REQUIRE(precision == PNR_PRECISION_FP32 || precision == PNR_PRECISION_BF16X3 || precision < 0,
                "pnr_weights_update: unknown precision %d", precision);
    const bool f32 = precision != PNR_PRECISION_BF16X3, b16 = precision != PNR_PRECISION_FP32;
    static const int n_out[9] = {256, 256, 256, 256, 1, 128, 128, 128, 3};
    static const int n_in[9] = {284, 256, 263, 256, 256, 280, 128, 128, 128};
    static const int ksp[9] = {144, 128, 132, 128, 0, 140, 64, 64, 0};
    static const int ks16[9] = {18, 16, 17, 16, 0, 18, 8, 8, 0};
    static const int kind[9] = {L_BASE0, L_HIDDEN, L_HEAD0, L_HIDDEN, -1, L_COLOR0, L_HIDDEN, L_HIDDEN, -1};
    PackJobs jobs{};
    int n = 0;
    auto add = [&](int type, const float *src, void *dst, int no, int ni, int kd, int a, int b, int c) {
        jobs.j[n++] = PackJob{type, no, ni, kd, a, b, c, src, dst};
    };
    for (int i = 0; i < 9; ++i) {
        if (ksp[i]) {
            if (f32 && i != 0) add(JOB_LAYER, d_w[i], w->buf + w->w_off[i], n_out[i], n_in[i], kind[i], ksp[i], 0, 0);
            if (b16 && i != 0)
                add(JOB_LAYER16, d_w[i], w->buf + w->w16_off[i], n_out[i], n_in[i], kind[i], ks16[i], ks16[i], 0);
        } else {
            add(JOB_COPY, d_w[i], w->buf + w->w_off[i], 0, 0, 0, n_out[i] * n_in[i], 0, 0);
        }
        add(JOB_COPY, d_b[i], w->buf + w->b_off[i], 0, 0, 0, n_out[i], 0, 0);
    }
    // mlp_base layer 0 is only read in its two halves (point-only / pair inputs): both modes factorise it
    if (f32) {
        add(JOB_LAYER, d_w[0], w->buf + w->w32a_off, 256, 284, L_BASE0, 112, 0, 0);
        add(JOB_LAYER, d_w[0], w->buf + w->w32b_off, 256, 284, L_BASE0, 32, 112, 0);
    }
    if (b16) {
        add(JOB_LAYER16, d_w[0], w->buf + w->w16a_off, 256, 284, L_BASE0, 14, 14, 0);
        add(JOB_LAYER16, d_w[0], w->buf + w->w16b_off, 256, 284, L_BASE0, 8, 4, 14);
    }
    add(JOB_HEAD_ACC, d_w[4], w->buf + w->w4acc_off, 0, 0, 0, 0, 0, 0);
    add(JOB_COLOR_ACC, d_w[8], w->buf + w->w8acc_off, 0, 0, 0, 0, 0, 0);
    hipLaunchKernelGGL(k_pack_jobs, dim3(48, (unsigned)n), dim3(256), 0, stream, jobs);
    PNR_HIP_CHECK(hipGetLastError());
    return PNR_OK;
}
