// Clock probe: ONE wavefront that stays resident for `seconds` and, in windows of `window_ms`, reads the shader clock
// counter (s_memtime, clock64) next to the constant 100-MHz counter (s_memrealtime, wall_clock64): cycles per window /
// time per window = the EFFECTIVE shader clock of the XCD it runs on, whatever else the device is doing.  Run it in the
// background beside a workload of another process (the eval frame loop, the training-step loop) to see the clock those
// kernels actually get -- rocm-smi reports a level, the GRBM_GUI_ACTIVE counter a per-kernel average.
//   hipcc --offload-arch=gfx950 -O3 tools/ub_clock_probe.hip -o tools/bin/ub_clock_probe
//   tools/bin/ub_clock_probe [seconds=8] [window_ms=20]
// (The probe holds one wave slot of one SIMD: a 512-VGPR wave of a persistent kernel cannot share that SIMD while it runs,
// so the workload beside it is up to 1/256 slower; its clock is what is being measured, not its time.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <sys/time.h>

__global__ void k_probe(unsigned long long *out, int max_windows, unsigned long long window_ticks, unsigned long long total_ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long r_begin = wall_clock64();
    unsigned long long r0 = r_begin, c0 = clock64();
    int w = 0;
    while (w < max_windows) {
        unsigned long long r1;
        do {
            __builtin_amdgcn_s_sleep(64);
            r1 = wall_clock64();
        } while (r1 - r0 < window_ticks);
        const unsigned long long c1 = clock64();
        out[2 * w] = c1 - c0;
        out[2 * w + 1] = r1 - r0;
        ++w;
        r0 = r1;
        c0 = c1;
        if (r1 - r_begin >= total_ticks) break;
    }
    out[2 * max_windows] = (unsigned long long)w;
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 8.0;
    const double window_ms = argc > 2 ? atof(argv[2]) : 20.0;
    const double ref_hz = 100e6;   // s_memrealtime
    const int max_windows = (int)(seconds * 1e3 / window_ms) + 8;
    unsigned long long *d = nullptr;
    if (hipMalloc(&d, (2 * max_windows + 1) * sizeof(unsigned long long)) != hipSuccess) return 1;
    hipMemset(d, 0, (2 * max_windows + 1) * sizeof(unsigned long long));
    hipDeviceSynchronize();
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    const double t_launch = tv.tv_sec + tv.tv_usec * 1e-6;   // host time at the launch: window times are relative to it
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, max_windows, (unsigned long long)(window_ms * 1e-3 * ref_hz),
                       (unsigned long long)(seconds * ref_hz));
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::vector<unsigned long long> h(2 * max_windows + 1);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const int n = (int)h[2 * max_windows];
    printf("# t0 %.6f\n", t_launch);
    printf("# window  t_ms  effective_MHz\n");
    double t = 0;
    for (int i = 0; i < n; ++i) {
        const double dt = h[2 * i + 1] / ref_hz;
        t += dt;
        printf("%d %.1f %.1f\n", i, t * 1e3, h[2 * i] / dt / 1e6);
    }
    hipFree(d);
    return 0;
}
