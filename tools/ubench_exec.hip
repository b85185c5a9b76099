// ubench_exec.hip -- does the rate of FP64 VALU work depend on WHICH lanes of a wavefront are enabled?
// A loop of independent v_fma_f64 chains runs under an EXEC mask chosen per launch: all 64 lanes, every k-th lane, or
// the first 64/k lanes.  One wave per SIMD (1024 waves), like the integrator kernels at 65536 systems.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_exec.hip -o tools/ubench_exec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int CHAINS>
__global__ __launch_bounds__(256) void k(double *out, int iters, unsigned long long mask, double seed) {
    const int lane = threadIdx.x & 63;
    double s = 0;
    if ((mask >> lane) & 1ull) { // everything below runs with EXEC = mask
        double a[CHAINS];
        for (int i = 0; i < CHAINS; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
        const double b = 1.0000001, c = 1e-9;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 32 / CHAINS; r++)
#pragma unroll
                for (int i = 0; i < CHAINS; i++) a[i] = __builtin_fma(a[i], b, c);
        }
        for (int i = 0; i < CHAINS; i++) s += a[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
static float run(unsigned long long mask, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, threads = 256, iters = 20000;
    double *out;
    CK(hipMalloc(&out, sizeof(double) * blocks * threads));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, 100, mask, 1.5);
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, iters, mask, 1.5);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CK(hipFree(out));
    return best;
}

static unsigned long long every(int k, int off = 0) {
    unsigned long long m = 0;
    for (int l = off; l < 64; l += k) m |= 1ull << l;
    return m;
}
static unsigned long long first(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1); }

int main() {
    // cycles per wave-instruction per SIMD at a nominal 2.4 GHz: 20000 iterations x 32 FMAs per wave
    auto cyc = [](float ms, int w) { return ms * 1e-3 * 2.4e9 / (20000.0 * 32 * w); };
    for (int w : {1, 2}) {
        printf("waves/SIMD = %d, 8 independent chains per lane (cycles per wave-instruction per SIMD)\n", w);
        for (int kk : {1, 2, 3, 4, 5, 8, 10, 16, 32, 64})
            printf("  every %2d-th lane (%2d lanes): %6.2f     first %2d lanes: %6.2f\n", kk, (63 / kk) + 1,
                   cyc(run<8>(every(kk), w), w), (63 / kk) + 1, cyc(run<8>(first((63 / kk) + 1), w), w));
        printf("  lanes 0-15 only: %.2f   lanes 0-31: %.2f   lanes 0,16,32,48: %.2f   lanes 0-3: %.2f   one lane per 16 with 15 of the first 16: %.2f\n",
               cyc(run<8>(first(16), w), w), cyc(run<8>(first(32), w), w), cyc(run<8>(every(16), w), w), cyc(run<8>(first(4), w), w),
               cyc(run<8>(first(15) | every(16), w), w));
    }
    printf("first n lanes, 1 wave/SIMD:");
    for (int n = 7; n <= 14; n++) printf("  n=%d %.2f", n, cyc(run<8>(first(n), 1), 1));
    printf("\n");
    printf("1 wave/SIMD, 1 dependent chain per lane: all lanes %.2f, every 10th %.2f, first 7 %.2f\n", cyc(run<1>(~0ull, 1), 1),
           cyc(run<1>(every(10), 1), 1), cyc(run<1>(first(7), 1), 1));
    return 0;
}
