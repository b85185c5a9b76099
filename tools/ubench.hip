// ubench.hip -- FP64 VALU micro-benchmarks on gfx950: throughput of v_fma_f64 / v_mul_f64 /
// v_add_f64 / v_rsq_f64 / v_rcp_f64 / v_sqrt_f64 at 1, 2, 4 waves per SIMD, and the accuracy of the
// v_rsq_f64 seed and of rsqrt_full().  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = __builtin_fma(a[i], b, c);
                if (OP == 1) a[i] = a[i] * b;
                if (OP == 2) a[i] = a[i] + c;
                if (OP == 3) a[i] = __builtin_amdgcn_rsq(a[i]) + 2.0;      // rsq + add
                if (OP == 4) a[i] = __builtin_amdgcn_rcp(a[i]) + 2.0;      // rcp + add
                if (OP == 5) a[i] = __builtin_amdgcn_sqrt(a[i]) + 2.0;     // hw sqrt + add
                if (OP == 6) a[i] = 1.0 / a[i] + 2.0;                      // IEEE divide + add
                if (OP == 7) a[i] = sqrt(a[i]) + 2.0;                      // IEEE sqrt + add
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void acc_kernel(const double *x, double *y0, double *y1, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    y0[i] = y;
    double t = v * y;
    double e = __builtin_fma(-t, y, 1.0);
    double q = __builtin_fma(0.375, e, 0.5);
    y1[i] = __builtin_fma(y * e, q, y);
}

template <int OP>
static void run(const char *name, int waves_per_simd, double ops_per_iter_extra) {
    const int blocks = 256 * waves_per_simd, threads = 256, iters = 20000;
    double *out;
    CK(hipMalloc(&out, sizeof(double) * blocks * threads));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.5);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.5);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double wave_instr = (double)iters * 32 * (blocks * threads / 64);   // instructions of kind OP per wave
    const double per_simd = wave_instr / 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9 / per_simd;
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f cycles@2.4GHz per wave-instruction(+%g dependent add) per SIMD, "
           "%.2f Tinstr-lanes/s\n", name, waves_per_simd, ms, cyc, ops_per_iter_extra,
           wave_instr * 64 / (ms * 1e-3) / 1e12);
    CK(hipFree(out));
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", w, 0);
        run<1>("v_mul_f64", w, 0);
        run<2>("v_add_f64", w, 0);
        run<3>("v_rsq_f64+add", w, 1);
        run<4>("v_rcp_f64+add", w, 1);
        run<5>("v_sqrt_f64+add", w, 1);
        run<6>("ieee div+add", w, 1);
        run<7>("ieee sqrt+add", w, 1);
    }
    // accuracy
    const int n = 1 << 20;
    std::vector<double> x(n), y0(n), y1(n);
    srand(1);
    for (int i = 0; i < n; i++) x[i] = exp((rand() / (double)RAND_MAX - 0.5) * 40.0);
    double *dx, *d0, *d1;
    CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&d0, n * 8)); CK(hipMalloc(&d1, n * 8));
    CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(acc_kernel, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, n);
    CK(hipMemcpy(y0.data(), d0, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y1.data(), d1, n * 8, hipMemcpyDeviceToHost));
    double m0 = 0, m1 = 0;
    for (int i = 0; i < n; i++) {
        long double r = 1.0L / sqrtl((long double)x[i]);
        m0 = fmax(m0, (double)fabsl((y0[i] - r) / r));
        m1 = fmax(m1, (double)fabsl((y1[i] - r) / r));
    }
    printf("v_rsq_f64 seed max rel err %.3e (2^%.1f); rsqrt_full max rel err %.3e (%.2f ulp)\n", m0, log2(m0), m1,
           m1 / 1.11e-16);
    return 0;
}
