// capi.hip -- the extern "C" surface of libswarm_hip.so (include/swarm_hip.h) and the small
// utility kernels (active count, reactivate, energy, statistics, layout conversion).
#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "launch.hpp"

namespace swh {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));              \
            return SWARM_HIP_EHIP;                                                 \
        }                                                                          \
    } while (0)

static int check_ens(const swarm_hip_ensemble_t *ens) {
    if (!ens) {
        set_error("ensemble pointer is NULL");
        return SWARM_HIP_EINVAL;
    }
    if (ens->nsys < 0 || ens->nbod < SWARM_HIP_MIN_NBOD || ens->nbod > SWARM_HIP_MAX_NBOD ||
        ens->nbattr < 0 || ens->nsattr < 0) {
        set_error("bad ensemble shape nsys=%d nbod=%d nbattr=%d nsattr=%d (nbod must be %d..%d)", ens->nsys,
                  ens->nbod, ens->nbattr, ens->nsattr, SWARM_HIP_MIN_NBOD, SWARM_HIP_MAX_NBOD);
        return SWARM_HIP_EINVAL;
    }
    if (ens->nsys > (1 << 30)) { // lane indices are 32-bit ints in the kernels; grid * block must not wrap
        set_error("nsys=%d exceeds the supported maximum of 2^30 systems per ensemble", ens->nsys);
        return SWARM_HIP_EINVAL;
    }
    if (ens->nsys > 0 && (!ens->body || !ens->sysd || !ens->sysi)) {
        set_error("ensemble device pointers must not be NULL");
        return SWARM_HIP_EINVAL;
    }
    if (((uintptr_t)ens->body | (uintptr_t)ens->sysd) % sizeof(double) || (uintptr_t)ens->sysi % sizeof(int32_t)) {
        set_error("ensemble device pointers must be aligned to their element size");
        return SWARM_HIP_EINVAL;
    }
    return SWARM_HIP_OK;
}

static EnsView view_of(const swarm_hip_ensemble_t *ens) {
    EnsView v;
    v.nsys = ens->nsys;
    v.nbod = ens->nbod;
    v.nbf = 7 + ens->nbattr;
    v.nsf = 1 + ens->nsattr;
    v.body = ens->body;
    v.sysd = ens->sysd;
    v.sysi = ens->sysi;
    v.alist = nullptr;
    v.alist_n = nullptr;
    v.alist_cap = 0;
    return v;
}

// per-device scratch for the helpers (a counter), created by swarm_hip_init or lazily
constexpr int MAX_DEV = 64;
static int32_t *g_scratch[MAX_DEV] = {nullptr};

static std::mutex g_scratch_mu[MAX_DEV]; // a device's counter has one user at a time

// `hold` keeps the device's counter for the caller until it goes out of scope: swarm_hip_count_active runs
// memset, kernel and copy on it, and two host threads (or streams) counting on one GPU must not interleave
static int scratch_for_current(int32_t **out, std::unique_lock<std::mutex> &hold) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) {
        set_error("device ordinal %d out of range", dev);
        return SWARM_HIP_EINVAL;
    }
    hold = std::unique_lock<std::mutex>(g_scratch_mu[dev]);
    if (!g_scratch[dev]) HIP_TRY(hipMalloc((void **)&g_scratch[dev], 256));
    *out = g_scratch[dev];
    return SWARM_HIP_OK;
}

// ---- the allocation-free active count (swarm_hip_count_active_noalloc) ------------------------
// Its partial counts live in a file-scope device array instead of allocated memory: one int32 per
// workgroup of the first pass.  Each GPU holds its own copy (64 KiB) from the load of the code object
// to the end of the process.  `scratch` is a host-side stand-in here, not a device address: host
// code uses the per-device pointer hipGetSymbolAddress returns (noalloc_scratch), never the name.
constexpr int NOALLOC_SYS_PER_GROUP = 4096;                    // 256 threads x 16 systems
constexpr int NOALLOC_GROUPS = SWARM_HIP_NOALLOC_MAX_NSYS / NOALLOC_SYS_PER_GROUP; // 16384
static __device__ __align__(256) int32_t scratch[NOALLOC_GROUPS];
// Resolved once per device by swarm_hip_init (with that device current) and only read afterwards: the entry point
// itself takes no lock and makes no call that could load code or allocate.
static std::atomic<int32_t *> g_noalloc_ptr[MAX_DEV];
static std::mutex g_noalloc_mu;

static int noalloc_resolve_current() { // swarm_hip_init
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) {
        set_error("device ordinal %d out of range", dev);
        return SWARM_HIP_EINVAL;
    }
    std::lock_guard<std::mutex> lock(g_noalloc_mu);
    if (!g_noalloc_ptr[dev].load()) {
        int32_t *p = nullptr;
        HIP_TRY(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
        g_noalloc_ptr[dev].store(p);
    }
    return SWARM_HIP_OK;
}

static int noalloc_scratch(int32_t **out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    int32_t *p = (dev >= 0 && dev < MAX_DEV) ? g_noalloc_ptr[dev].load(std::memory_order_acquire) : nullptr;
    if (!p) {
        set_error("count_active_noalloc: device %d has not been set up -- call swarm_hip_init(%d) first (it resolves the "
                  "device's copy of the static scratch array; this entry point does no set-up of its own)", dev, dev);
        return SWARM_HIP_EINVAL;
    }
    *out = p;
    return SWARM_HIP_OK;
}

// ---- utility kernels ---------------------------------------------------------------------
// number_of_active_systems_kernel (swarm/gpu/utilities.cu:36-64) does one atomicAdd per ACTIVE
// system; here one per wavefront (ballot + popcount).
__global__ void count_active_kernel(EnsView e, int32_t *out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    bool act = false;
    if (gid < e.nsys) act = e.sysi[(size_t)(gid / W) * 2 * W + gid % W] == SWARM_HIP_ACTIVE;
    const unsigned long long m = __ballot(act);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, __popcll(m));
}

// The ACTIVE systems of the ensemble gathered into a list (lane-per-body launches of a thinned-out ensemble,
// prepare_body_launch): one ballot and one atomic per wavefront reserve the wave's run of slots; a wave's systems
// stay in index order, the order of the waves' runs is whatever the atomics give -- a system's result does not depend
// on which wavefront integrates it.  count[0] receives the number found; only the first `cap` are stored.  With
// `rest` the systems that are NOT ACTIVE are listed as well (count[1], room for all of them): hermite_adap and rkck
// call their monitor once before the step loop for every system, whatever its state (hermite_adap.hpp:130,
// rkck_integrator.hpp:140), and a second launch over that list makes those calls.
__global__ __launch_bounds__(256) void active_list_kernel(EnsView e, int32_t *list, int32_t *count, int cap, int32_t *rest) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = gid < e.nsys;
    bool act = false;
    if (in) act = e.sysi[(size_t)(gid / W) * 2 * W + gid % W] == SWARM_HIP_ACTIVE;
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long m = __ballot(act);
    if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(count, __popcll(m));
        base = __shfl(base, 0);
        const int at = base + __popcll(m & below);
        if (act && at < cap) list[at] = gid;
    }
    if (!rest) return;
    const unsigned long long r = __ballot(in && !act);
    if (r) {
        int base = 0;
        if (lane == 0) base = atomicAdd(count + 1, __popcll(r));
        base = __shfl(base, 0);
        if (in && !act) rest[base + __popcll(r & below)] = gid;
    }
}

// Two-pass count without atomics for swarm_hip_count_active_noalloc.  Pass 1: workgroup g counts the ACTIVE
// systems among [g*4096, (g+1)*4096) -- a lane reads 16 state rows, consecutive lanes consecutive systems -- and
// writes its sum to partial[g] (every slot the second pass reads is written: nothing has to start at zero).
__global__ __launch_bounds__(256) void count_active_partial_kernel(EnsView e, int32_t *partial) {
    __shared__ int wsum[4];
    const long long base = (long long)blockIdx.x * NOALLOC_SYS_PER_GROUP;
    int n = 0;
#pragma unroll 4
    for (int k = 0; k < NOALLOC_SYS_PER_GROUP / 256; k++) {
        const long long gid = base + k * 256 + threadIdx.x;
        bool act = false;
        if (gid < e.nsys) act = e.sysi[(size_t)(gid / W) * 2 * W + gid % W] == SWARM_HIP_ACTIVE;
        n += __popcll(__ballot(act)); // wave-uniform
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// Pass 2: one workgroup adds the ngroups partial counts and stores the total.
__global__ __launch_bounds__(256) void count_active_final_kernel(const int32_t *partial, int ngroups, int32_t *out) {
    __shared__ int wsum[4];
    int n = 0;
    for (int g = threadIdx.x; g < ngroups; g += 256) n += partial[g];
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) *out = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// reactivate_systems_kernel, swarm/gpu/utilities.cu:113-131
__global__ void reactivate_kernel(EnsView e) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= e.nsys) return;
    int32_t *st = e.sysi + (size_t)(gid / W) * 2 * W + gid % W;
    if (*st == SWARM_HIP_INACTIVE) *st = SWARM_HIP_ACTIVE;
}

// SystemRef::calc_total_energy, swarm/types/ensemble.hpp:402-413 (same operation order, no FMA)
__global__ void energy_kernel(EnsView e, double *out) {
#pragma clang fp contract(off)
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= e.nsys) return;
    const int chunk = gid / W, lane = gid % W, n = e.nbod;
    const double *b = e.body + (size_t)chunk * n * e.nbf * W + lane;
    double K = 0.0, U = 0.0;
    for (int i = 0; i < n; i++) {
        const double *r = b + (size_t)i * e.nbf * W;
        const double vx = r[3 * W], vy = r[4 * W], vz = r[5 * W];
        K += 0.5 * r[6 * W] * (vx * vx + vy * vy + vz * vz);
    }
    for (int i = 0; i < n; i++) {
        const double *ri = b + (size_t)i * e.nbf * W;
        for (int j = 0; j < i; j++) {
            const double *rj = b + (size_t)j * e.nbf * W;
            const double dx = ri[0] - rj[0], dy = ri[W] - rj[W], dz = ri[2 * W] - rj[2 * W];
            U += -ri[6 * W] * rj[6 * W] / sqrt(dx * dx + dy * dy + dz * dz);
        }
    }
    out[gid] = K + U;
}

// wave-level reductions (DPP/bpermute shuffles), then one atomic per wavefront
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ void atomic_max_pos(double *addr, double v) { // v >= 0
    atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ void atomic_min_pos(double *addr, double v) { // v >= 0
    atomicMin(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(v));
}

__global__ void stats_init_kernel(double *out, int nsys) {
    out[0] = 0.0;
    out[1] = 0.0;
    out[2] = __longlong_as_double(0x7ff0000000000000LL); // +inf
    out[3] = out[4] = out[5] = out[6] = 0.0;
    out[7] = (double)nsys;
}

__global__ void stats_kernel(EnsView e, const double *e0, const double *e1, double *out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = gid < e.nsys;
    double err = 0.0;
    int st = 12345;
    if (in) {
        err = fabs((e1[gid] - e0[gid]) / e0[gid]); // utils.cpp:117-118
        st = e.sysi[(size_t)(gid / W) * 2 * W + gid % W];
    }
    const double s = wave_sum(in ? err : 0.0);
    const double mx = wave_max(in ? err : 0.0);
    const double mn = wave_min(in ? err : __longlong_as_double(0x7ff0000000000000LL));
    const int na = __popcll(__ballot(in && st == SWARM_HIP_ACTIVE));
    const int ni = __popcll(__ballot(in && st == SWARM_HIP_INACTIVE));
    const int nd = __popcll(__ballot(in && st == SWARM_HIP_DISABLED));
    const int no = __popcll(__ballot(in && st != SWARM_HIP_ACTIVE && st != SWARM_HIP_INACTIVE &&
                                     st != SWARM_HIP_DISABLED));
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out + 0, s);
        if (mx == mx) atomic_max_pos(out + 1, mx);
        if (mn == mn) atomic_min_pos(out + 2, mn);
        if (na) atomicAdd(out + 3, (double)na);
        if (ni) atomicAdd(out + 4, (double)ni);
        if (nd) atomicAdd(out + 5, (double)nd);
        if (no) atomicAdd(out + 6, (double)no);
    }
}

// ---- reference layout <-> wave chunks ------------------------------------------------------
// Reference Body block (ensemble.hpp:131-150), RC = ref chunk: doubles
//   [x_pos RC][x_vel RC][y_pos RC][y_vel RC][z_pos RC][z_vel RC][mass RC][attr nbattr x RC]
// system i, body b lives in Body block (i/RC)*nbod + b at lane i%RC (ensemble.hpp:499-517).
// Reference Sys block (ensemble.hpp:227-238): [time RC doubles][{int state; int id} x RC][attr nsattr x RC].
// One workgroup of 64 threads converts one wave chunk.  The reference image of those 64 systems
// is 64/RC contiguous block groups: it is streamed through LDS with fully coalesced loads/stores
// and transposed there, so neither side sees strided 64-byte segments.
template <bool IMPORT>
__global__ __launch_bounds__(64) void ref_layout_kernel(EnsView e, double *ref_body, double *ref_sys, int RC) {
    extern __shared__ double lds[];
    const int chunk = blockIdx.x, lane = threadIdx.x;
    const int n = e.nbod, nbf = e.nbf, nsf = e.nsf;
    const int groups = W / RC; // reference block groups per wave chunk
    // ---- bodies: reference doubles for this wave chunk are contiguous
    const size_t body_doubles = (size_t)n * nbf * W;
    double *rb = ref_body + (size_t)chunk * body_doubles;
    double *ob = e.body + (size_t)chunk * body_doubles;
    // number of valid reference doubles (the last wave chunk may be short on the reference side)
    const long long nsys_here = min((long long)W, (long long)e.nsys - (long long)chunk * W);
    const int groups_here = (int)((nsys_here + RC - 1) / RC);
    const size_t valid = (size_t)groups_here * n * nbf * RC;
    // reference index r = ((g*n + b)*nbfr + fr)*RC + l  with fr order x_pos,x_vel,y_pos,y_vel,z_pos,z_vel,mass,attr
    // ours      index o = (b*nbf + f)*W + (g*RC + l)     with f  order x,y,z,vx,vy,vz,mass,attr
    if (IMPORT) {
        for (size_t k = lane; k < body_doubles; k += W) lds[k] = (k < valid) ? rb[k] : 0.0;
        __syncthreads();
        for (size_t o = lane; o < body_doubles; o += W) {
            const int sysl = (int)(o % W);
            const int f = (int)((o / W) % nbf), b = (int)(o / ((size_t)W * nbf));
            const int g = sysl / RC, l = sysl % RC;
            const int fr = f < 3 ? 2 * f : (f < 6 ? 2 * (f - 3) + 1 : f);
            ob[o] = lds[((size_t)(g * n + b) * nbf + fr) * RC + l];
        }
    } else {
        for (size_t o = lane; o < body_doubles; o += W) lds[o] = ob[o];
        __syncthreads();
        for (size_t k = lane; k < valid; k += W) {
            const int l = (int)(k % RC);
            const int fr = (int)((k / RC) % nbf);
            const int gb = (int)(k / ((size_t)RC * nbf));
            const int b = gb % n, g = gb / n;
            const int f = fr < 6 ? (fr % 2 == 0 ? fr / 2 : 3 + fr / 2) : fr;
            rb[k] = lds[((size_t)b * nbf + f) * W + g * RC + l];
        }
    }
    __syncthreads();
    // ---- systems: Sys block = (1 + 1 + nsattr) * RC doubles (the {state,id} pair is 8 bytes)
    const int sdr = (2 + (nsf - 1)) * RC;
    double *rs = ref_sys + (size_t)chunk * groups * sdr;
    const int gidx = chunk * W + lane;
    if (gidx < e.nsys) {
        const int g = lane / RC, l = lane % RC;
        double *blk = rs + (size_t)g * sdr;
        int32_t *bunch = reinterpret_cast<int32_t *>(blk + RC) + 2 * l;
        double *od = e.sysd + (size_t)chunk * nsf * W + lane;
        int32_t *oi = e.sysi + (size_t)chunk * 2 * W + lane;
        if (IMPORT) {
            od[0] = blk[l];
            oi[0] = bunch[0];
            oi[W] = bunch[1];
            for (int a = 1; a < nsf; a++) od[(size_t)a * W] = blk[(size_t)(1 + a) * RC + l];
        } else {
            blk[l] = od[0];
            bunch[0] = oi[0];
            bunch[1] = oi[W];
            for (int a = 1; a < nsf; a++) blk[(size_t)(1 + a) * RC + l] = od[(size_t)a * W];
        }
    } else if (IMPORT) {
        e.sysd[(size_t)chunk * nsf * W + lane] = 0.0;
        e.sysi[(size_t)chunk * 2 * W + lane] = SWARM_HIP_DISABLED;
        e.sysi[(size_t)chunk * 2 * W + W + lane] = -1;
    }
}

// Device staging for the AoS <-> wave-chunk conversion: the host arrays are copied as they are and
// the transposition runs on the GPU (one lane per system), instead of a scalar host loop over every
// double (which cost 2 x 22 ms for the 131072 x 8 shard of C4 against a 19 ms kernel).
struct AosStage {
    double *pos, *vel, *mass, *time, *battr, *sattr;
    int32_t *state, *id;
    void *base;
    bool owned; // false: the device's kept staging buffer
};
// The staging buffer of a device is kept between calls (grow-only, up to STAGE_KEEP_MAX; larger requests are
// private to their call): the reference pays a cudaMalloc/cudaFree pair per transfer (types/allocators.hpp:152-166),
// which for a 14..70 MB ensemble costs as much as the copy itself.  A call holds the device's staging mutex from
// aos_stage_alloc to its return, so transfers that overlap on one GPU run one after the other.
constexpr size_t STAGE_KEEP_MAX = (size_t)1 << 30;
struct StageCache {
    std::mutex mu;
    void *base = nullptr;
    size_t bytes = 0;
};
static StageCache g_stage[MAX_DEV];

static int aos_stage_alloc(const swarm_hip_ensemble_t *ens, AosStage &st, std::unique_lock<std::mutex> &hold) {
    const size_t nsys = ens->nsys, n = ens->nbod;
    const size_t nd = nsys * n * 3 * 2 + nsys * n + nsys + nsys * n * ens->nbattr + nsys * ens->nsattr;
    const size_t bytes = nd * sizeof(double) + 2 * nsys * sizeof(int32_t) + 64;
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    st.owned = true;
    if (dev >= 0 && dev < MAX_DEV && bytes <= STAGE_KEEP_MAX) {
        StageCache &c = g_stage[dev];
        hold = std::unique_lock<std::mutex>(c.mu);
        if (c.bytes < bytes) {
            if (c.base) (void)hipFree(c.base);
            c.base = nullptr, c.bytes = 0;
            HIP_TRY(hipMalloc(&c.base, bytes));
            c.bytes = bytes;
        }
        st.base = c.base;
        st.owned = false;
    } else {
        HIP_TRY(hipMalloc(&st.base, bytes));
    }
    double *p = (double *)st.base;
    st.pos = p, p += nsys * n * 3;
    st.vel = p, p += nsys * n * 3;
    st.mass = p, p += nsys * n;
    st.time = p, p += nsys;
    st.battr = p, p += nsys * n * ens->nbattr;
    st.sattr = p, p += nsys * ens->nsattr;
    st.state = (int32_t *)p;
    st.id = st.state + nsys;
    return SWARM_HIP_OK;
}
static void aos_stage_release(AosStage &st) {
    if (st.owned && st.base) (void)hipFree(st.base);
    st.base = nullptr;
}

// to_chunks: AoS staging -> wave chunks (padding lanes: DISABLED, id -1, zeros); else the reverse
template <bool TO_CHUNKS>
__global__ __launch_bounds__(256) void aos_convert_kernel(EnsView e, AosStage st, bool have_battr, bool have_sattr) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nlanes = ((size_t)e.nsys + W - 1) / W * W;
    if (gid >= nlanes) return;
    const size_t k = gid / W, l = gid % W;
    const int n = e.nbod, nba = e.nbf - 7, nsa = e.nsf - 1;
    const bool real = gid < (size_t)e.nsys;
    for (int b = 0; b < n; b++) {
        double *r = e.body + ((k * n + b) * e.nbf) * W + l;
        const size_t ib = gid * n + b;
        for (int c = 0; c < 3; c++) {
            if (TO_CHUNKS) {
                r[c * W] = real ? st.pos[ib * 3 + c] : 0.0;
                r[(3 + c) * W] = real ? st.vel[ib * 3 + c] : 0.0;
            } else if (real) {
                st.pos[ib * 3 + c] = r[c * W];
                st.vel[ib * 3 + c] = r[(3 + c) * W];
            }
        }
        if (TO_CHUNKS)
            r[6 * W] = real ? st.mass[ib] : 0.0;
        else if (real)
            st.mass[ib] = r[6 * W];
        for (int a = 0; a < nba; a++) {
            if (TO_CHUNKS)
                r[(7 + a) * W] = (real && have_battr) ? st.battr[ib * nba + a] : 0.0;
            else if (real)
                st.battr[ib * nba + a] = r[(7 + a) * W];
        }
    }
    double *d = e.sysd + k * e.nsf * W + l;
    int32_t *si = e.sysi + k * 2 * W + l;
    if (TO_CHUNKS) {
        d[0] = real ? st.time[gid] : 0.0;
        for (int a = 0; a < nsa; a++) d[(size_t)(1 + a) * W] = (real && have_sattr) ? st.sattr[gid * nsa + a] : 0.0;
        si[0] = real ? st.state[gid] : SWARM_HIP_DISABLED;
        si[W] = real ? st.id[gid] : -1;
    } else if (real) {
        st.time[gid] = d[0];
        for (int a = 0; a < nsa; a++) st.sattr[gid * nsa + a] = d[(size_t)(1 + a) * W];
        st.state[gid] = si[0];
        st.id[gid] = si[W];
    }
}


// The same conversion with the transposition staged through LDS: one workgroup per wave chunk.  A chunk's AoS data
// is a contiguous run of 64*F doubles per field group (F = 3n for pos and vel, n for mass, n*nbattr for the body
// attributes), read or written with consecutive lanes on consecutive doubles; the device side is whole 512-byte rows.
// The tile is [64][F|1]: an odd row stride keeps the column accesses free of bank conflicts.  The system data
// (time, attributes, state, id) are per-lane streams already and go through the first wavefront.
template <bool TO_CHUNKS>
__global__ __launch_bounds__(256) void aos_tile_kernel(EnsView e, AosStage st, bool have_battr, bool have_sattr) {
    extern __shared__ double tile[];
    const size_t k = blockIdx.x, s0 = k * W;
    const int t = threadIdx.x, n = e.nbod, nba = e.nbf - 7, nsa = e.nsf - 1;
    const int r = (size_t)e.nsys - s0 < (size_t)W ? (int)((size_t)e.nsys - s0) : W; // real systems of this chunk
    for (int g = 0; g < 4; g++) {
        const int F = g < 2 ? 3 * n : g == 2 ? n : n * nba;
        if (F == 0) continue;
        double *aos = (g == 0 ? st.pos : g == 1 ? st.vel : g == 2 ? st.mass : st.battr) + s0 * F;
        const bool zeros = TO_CHUNKS && g == 3 && !have_battr;
        const int S = F | 1, total = W * F;
        const int per_body = g < 2 ? 3 : g == 2 ? 1 : nba;            // fields of this group per body
        const int row0 = g == 0 ? 0 : g == 1 ? 3 : g == 2 ? 6 : 7;    // first chunk row of the group
        if (TO_CHUNKS) {
            for (int i = t; i < total; i += 256) {
                const int l = i / F, f = i - l * F;
                tile[l * S + f] = (!zeros && l < r) ? aos[i] : 0.0;
            }
            __syncthreads();
            for (int i = t; i < total; i += 256) {
                const int f = i / W, l = i % W, b = f / per_body, q = f - b * per_body;
                e.body[((k * n + b) * e.nbf + row0 + q) * W + l] = tile[l * S + f];
            }
        } else {
            for (int i = t; i < total; i += 256) {
                const int f = i / W, l = i % W, b = f / per_body, q = f - b * per_body;
                tile[l * S + f] = e.body[((k * n + b) * e.nbf + row0 + q) * W + l];
            }
            __syncthreads();
            for (int i = t; i < total; i += 256) {
                const int l = i / F, f = i - l * F;
                if (l < r) aos[i] = tile[l * S + f];
            }
        }
        __syncthreads();
    }
    if (t >= W) return;
    const size_t gid = s0 + t;
    const bool real = t < r;
    double *d = e.sysd + k * e.nsf * W + t;
    int32_t *si = e.sysi + k * 2 * W + t;
    if (TO_CHUNKS) {
        d[0] = real ? st.time[gid] : 0.0;
        for (int a = 0; a < nsa; a++) d[(size_t)(1 + a) * W] = (real && have_sattr) ? st.sattr[gid * nsa + a] : 0.0;
        si[0] = real ? st.state[gid] : SWARM_HIP_DISABLED;
        si[W] = real ? st.id[gid] : -1;
    } else if (real) {
        st.time[gid] = d[0];
        for (int a = 0; a < nsa; a++) st.sattr[gid * nsa + a] = d[(size_t)(1 + a) * W];
        st.state[gid] = si[0];
        st.id[gid] = si[W];
    }
}

// LDS-tiled kernel when its tile fits the default 64 KB of dynamic LDS, the lane-per-system one otherwise
template <bool TO_CHUNKS>
static void launch_aos_convert(const swarm_hip_ensemble_t *ens, const AosStage &st, bool have_battr, bool have_sattr) {
    const size_t nchunk = (size_t)swarm_hip_nchunk(ens->nsys);
    const size_t fmax = (size_t)ens->nbod * (ens->nbattr > 3 ? ens->nbattr : 3);
    const size_t lds = (size_t)W * (fmax | 1) * sizeof(double);
    if (lds <= 64 * 1024)
        hipLaunchKernelGGL((aos_tile_kernel<TO_CHUNKS>), dim3((unsigned)nchunk), dim3(256), lds, 0, view_of(ens), st,
                           have_battr, have_sattr);
    else
        hipLaunchKernelGGL((aos_convert_kernel<TO_CHUNKS>), dim3((unsigned)((nchunk * W + 255) / 256)), dim3(256), 0, 0,
                           view_of(ens), st, have_battr, have_sattr);
}

static int grid_for(int nsys, int block) { return (int)(((long long)nsys + block - 1) / block); }

// launch.hpp.  The lists live in stream-ordered memory (hipMallocAsync / hipFreeAsync on the launch stream): no
// synchronisation, and launches on different streams of one GPU do not share them.  Layout: int32 counts at [0] and
// [1], the ACTIVE list from [64] on, the list of the others behind it.
int prepare_body_launch(const LaunchCtx &c, int G, bool premonitor, BodyLaunch &bl) {
    bl.view = bl.rest = c.ens;
    bl.rest_grid = 0;
    bl.mem = nullptr;
    const int n = c.effective_nsys();
    bl.grid = (int)(((long long)n + G - 1) / G);
    if (n == c.ens.nsys) return SWARM_HIP_OK;
    const size_t na = ((size_t)n + 63) / 64 * 64, nr = premonitor ? (size_t)c.ens.nsys : 0;
    int32_t *mem = nullptr;
    HIP_TRY(hipMallocAsync((void **)&mem, (64 + na + nr) * sizeof(int32_t), c.stream));
    hipError_t err = hipMemsetAsync(mem, 0, 2 * sizeof(int32_t), c.stream);
    if (err == hipSuccess) {
        hipLaunchKernelGGL(active_list_kernel, dim3(grid_for(c.ens.nsys, 256)), dim3(256), 0, c.stream, c.ens, mem + 64, mem, n,
                           premonitor ? mem + 64 + na : nullptr);
        err = hipGetLastError();
    }
    if (err != hipSuccess) {
        (void)hipFreeAsync(mem, c.stream);
        HIP_TRY(err);
    }
    bl.mem = mem;
    bl.view.alist = mem + 64;
    bl.view.alist_n = mem;
    bl.view.alist_cap = n;
    if (premonitor) {
        bl.rest.alist = mem + 64 + na;
        bl.rest.alist_n = mem + 1;
        bl.rest.alist_cap = c.ens.nsys;
        // the hint may exceed the true count: room for every system (workgroups beyond the list return at once)
        bl.rest_grid = (int)(((long long)c.ens.nsys + G - 1) / G);
    }
    return SWARM_HIP_OK;
}
void finish_body_launch(const LaunchCtx &c, BodyLaunch &bl) {
    if (bl.mem) (void)hipFreeAsync(bl.mem, c.stream);
    bl.mem = nullptr;
}

} // namespace swh

using namespace swh;

// =============================================================================================
extern "C" {

const char *swarm_hip_last_error(void) { return g_err; }
const char *swarm_hip_version(void) { return "swarm_hip 0.1 (gfx950)"; }

int swarm_hip_init(int device) {
    HIP_TRY(hipSetDevice(device));
    int32_t *s = nullptr;
    std::unique_lock<std::mutex> hold;
    int rc = scratch_for_current(&s, hold);
    if (rc) return rc;
    return noalloc_resolve_current(); // this device's copy of the static scratch array
}

int64_t swarm_hip_nchunk(int32_t nsys) { return ((int64_t)nsys + W - 1) / W; }
size_t swarm_hip_body_bytes(int32_t nsys, int32_t nbod, int32_t nbattr) {
    return (size_t)swarm_hip_nchunk(nsys) * nbod * (7 + nbattr) * W * sizeof(double);
}
size_t swarm_hip_sysd_bytes(int32_t nsys, int32_t nsattr) {
    return (size_t)swarm_hip_nchunk(nsys) * (1 + nsattr) * W * sizeof(double);
}
size_t swarm_hip_sysi_bytes(int32_t nsys) { return (size_t)swarm_hip_nchunk(nsys) * 2 * W * sizeof(int32_t); }
int64_t swarm_hip_snapshot_record_bytes(int32_t nbod) { return (32 + 64 * (int64_t)nbod + 15) / 16 * 16; }

int swarm_hip_launch_integrator(const swarm_hip_ensemble_t *ens, const swarm_hip_launch_t *a) {
    int rc = check_ens(ens);
    if (rc) return rc;
    if (!a) {
        set_error("launch arguments pointer is NULL");
        return SWARM_HIP_EINVAL;
    }
    if (ens->nsys == 0) return SWARM_HIP_OK;
    LaunchCtx c;
    c.ens = view_of(ens);
    c.ip.max_iterations = a->max_iterations;
    c.ip.destination_time = a->destination_time;
    c.ip.time_step = a->time_step;
    c.ip.time_step_factor = a->time_step_factor;
    c.ip.min_time_step = a->min_time_step;
    c.ip.max_time_step = a->max_time_step;
    c.ip.error_tolerance = a->error_tolerance;
    c.ip.max_timestep = a->max_timestep;
    c.ip.timestep_scale = a->timestep_scale;
    c.ip.kepler_compensated = a->kepler_compensated != 0;
    if (a->monitor < SWARM_HIP_MON_EJECTION || a->monitor > SWARM_HIP_MON_LOG_TRANSIT) {
        set_error("unknown monitor id %d", a->monitor);
        return SWARM_HIP_EINVAL;
    }
    c.mp.kind = a->monitor;
    c.mp.ej_deactivate = a->deactivate_on_ejection != 0;
    c.mp.ej_log = a->log_on_ejection != 0;
    c.mp.ej_any = c.mp.ej_deactivate || c.mp.ej_log || (a->verbose_on_ejection != 0);
    c.mp.ce_deactivate = a->deactivate_on_close_encounter != 0;
    c.mp.ce_log = a->log_on_close_encounter != 0;
    c.mp.ce_any = c.mp.ce_deactivate || c.mp.ce_log || (a->verbose_on_close_encounter != 0);
    c.mp.rmax = a->rmax;
    // lower bound of { r2 : sqrt(r2) >= rmax }: a few ulps under rmax^2 (exact test follows on device)
    c.mp.r2_below = (a->rmax > 0 && a->rmax < 1e150) ? a->rmax * a->rmax * (1.0 - 8 * DBL_EPSILON) : 0.0;
    c.mp.dmin3 = a->close_approach * a->close_approach * a->close_approach;
    c.mp.log_interval = a->log_interval;
    c.mp.ej_verbose = a->verbose_on_ejection != 0;
    c.mp.ce_verbose = a->verbose_on_close_encounter != 0;
    c.mp.dmin = a->close_approach;
    c.mp.rceh = a->rceh;
    c.mp.mce_time_step = a->time_step; // mce_stat_params reads the same time_step= key (L45)
    c.mp.tr_tol = a->log_transit_tol;
    c.mp.tr_dt = a->time_step; // log_transit_params, log_transit.hpp:54
    c.mp.tr_log_transit = a->log_transits != 0;
    c.mp.tr_log_occultation = a->log_occultations != 0;
    c.lg.buf = a->log.d_buf;
    c.lg.cursor = a->log.d_cursor;
    c.lg.cap = a->log.capacity;
    if (c.lg.buf && !c.lg.cursor) {
        set_error("log.d_buf given without log.d_cursor");
        return SWARM_HIP_EINVAL;
    }
    // the cursor is an int32 that lanes bump before they check the capacity (as gpulog does,
    // bits/gpulog_write.h:14-22): keep room for the transient overshoot of every resident lane
    if (c.lg.buf && (c.lg.cap < 0 || c.lg.cap > 0x70000000)) {
        set_error("log.capacity=%d must be in [0, 0x70000000] bytes", c.lg.cap);
        return SWARM_HIP_EINVAL;
    }
    c.cnt.accepted = reinterpret_cast<unsigned long long *>(a->d_steps_accepted);
    c.cnt.attempted = reinterpret_cast<unsigned long long *>(a->d_steps_attempted);
    c.block = a->block_size > 0 ? a->block_size : 256;
    if (c.block % W != 0 || c.block > 256) {
        set_error("block_size=%d must be a multiple of 64 and at most 256", c.block);
        return SWARM_HIP_EINVAL;
    }
    c.grid = grid_for(ens->nsys, c.block);
    if (a->lane_mode < 0 || a->lane_mode > 2) {
        set_error("lane_mode=%d must be 0 (automatic), 1 (lane per system) or 2 (lane per body)", a->lane_mode);
        return SWARM_HIP_EINVAL;
    }
    c.lane_mode = a->lane_mode;
    if (a->active_hint < 0) {
        set_error("active_hint=%d must be >= 0 (0 = not known)", a->active_hint);
        return SWARM_HIP_EINVAL;
    }
    c.active_hint = a->active_hint;
    c.stream = (hipStream_t)a->stream;

    switch (a->integrator) {
    case SWARM_HIP_HERMITE:
    case SWARM_HIP_HERMITE_PROP:
    case SWARM_HIP_IRK2:
    case SWARM_HIP_IRK2_CPU:
    case SWARM_HIP_EULER:
    case SWARM_HIP_MIDPOINT:
    case SWARM_HIP_MVS:
        if (!(a->time_step > 0)) {
            set_error("time_step must be > 0 (got %g)", a->time_step);
            return SWARM_HIP_EINVAL;
        }
        break;
    case SWARM_HIP_RKCK_FIXED:
    case SWARM_HIP_RKCK_ADAPTIVE:
        if (!(a->max_time_step > 0) || !(a->error_tolerance > 0)) {
            set_error("rkck needs max_time_step > 0 and error_tolerance > 0");
            return SWARM_HIP_EINVAL;
        }
        break;
    case SWARM_HIP_VERLET:
        if (!(a->max_timestep > 0)) {
            set_error("verlet needs max_timestep > 0");
            return SWARM_HIP_EINVAL;
        }
        break;
    default:
        break;
    }
    if (a->monitor == SWARM_HIP_MON_MCE_STAT && a->integrator != SWARM_HIP_HERMITE && a->integrator != SWARM_HIP_IRK2) {
        set_error("monitor mce_stat is bound to the integrators hermite and irk2 only (plugins/hermite.cu:40, irk2.cu:39)");
        return SWARM_HIP_EINVAL;
    }
    if (a->monitor == SWARM_HIP_MON_LOG_TRANSIT && a->integrator != SWARM_HIP_HERMITE_ADAP) {
        set_error("monitor log_transit is bound to integrator hermite_adap only (plugins/hermite_adap.cu:61)");
        return SWARM_HIP_EINVAL;
    }
    switch (a->integrator) {
    case SWARM_HIP_HERMITE:
        if (a->monitor == SWARM_HIP_MON_MCE_STAT) return launch_hermite_mce(c);
        return launch_hermite(c, false);
    case SWARM_HIP_HERMITE_ADAP:
        return launch_hermite(c, true);
    case SWARM_HIP_RKCK_FIXED:
        return launch_rkck(c, false);
    case SWARM_HIP_RKCK_ADAPTIVE:
        return launch_rkck(c, true);
    case SWARM_HIP_MVS:
    case SWARM_HIP_EULER:
    case SWARM_HIP_VERLET:
    case SWARM_HIP_MIDPOINT:
    case SWARM_HIP_HERMITE_PROP:
        return launch_generic(c, a->integrator);
    case SWARM_HIP_IRK2:
        return launch_irk2(c, a->irk2_method ? a->irk2_method : 6, false);
    case SWARM_HIP_IRK2_CPU:
        return launch_irk2(c, 6, true);
    default:
        set_error("unknown integrator id %d", a->integrator);
        return SWARM_HIP_EINVAL;
    }
}

int swarm_hip_count_active(const swarm_hip_ensemble_t *ens, void *stream, int32_t *count) {
    int rc = check_ens(ens);
    if (rc) return rc;
    if (!count) {
        set_error("count pointer is NULL");
        return SWARM_HIP_EINVAL;
    }
    *count = 0;
    if (ens->nsys == 0) return SWARM_HIP_OK;
    int32_t *d = nullptr;
    std::unique_lock<std::mutex> hold; // released when the count has arrived on the host
    rc = scratch_for_current(&d, hold);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(d, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(count_active_kernel, dim3(grid_for(ens->nsys, 256)), dim3(256), 0, st, view_of(ens), d);
    rc = check_launch("count_active kernel launch");
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(count, d, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return SWARM_HIP_OK;
}

int swarm_hip_count_active_noalloc(const swarm_hip_ensemble_t *ens, int32_t *d_count, void *stream) {
    int rc = check_ens(ens);
    if (rc) return rc;
    if (!d_count) {
        set_error("count_active_noalloc: d_count is NULL");
        return SWARM_HIP_EINVAL;
    }
    if (ens->nsys > SWARM_HIP_NOALLOC_MAX_NSYS) {
        set_error("count_active_noalloc: nsys=%d exceeds SWARM_HIP_NOALLOC_MAX_NSYS=%d (the static partial-count "
                  "array has one slot per %d systems); use swarm_hip_count_active",
                  ens->nsys, SWARM_HIP_NOALLOC_MAX_NSYS, NOALLOC_SYS_PER_GROUP);
        return SWARM_HIP_EINVAL;
    }
    int32_t *partial = nullptr;
    rc = noalloc_scratch(&partial);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int ngroups = (int)(((long long)ens->nsys + NOALLOC_SYS_PER_GROUP - 1) / NOALLOC_SYS_PER_GROUP);
    if (ngroups > 0)
        hipLaunchKernelGGL(count_active_partial_kernel, dim3(ngroups), dim3(256), 0, st, view_of(ens), partial);
    hipLaunchKernelGGL(count_active_final_kernel, dim3(1), dim3(256), 0, st, partial, ngroups, d_count);
    return check_launch("count_active_noalloc kernel launch");
}

int swarm_hip_reactivate(const swarm_hip_ensemble_t *ens, void *stream) {
    int rc = check_ens(ens);
    if (rc) return rc;
    if (ens->nsys == 0) return SWARM_HIP_OK;
    hipLaunchKernelGGL(reactivate_kernel, dim3(grid_for(ens->nsys, 256)), dim3(256), 0, (hipStream_t)stream,
                       view_of(ens));
    return check_launch("reactivate kernel launch");
}

int swarm_hip_total_energy(const swarm_hip_ensemble_t *ens, double *d_energy, void *stream) {
    int rc = check_ens(ens);
    if (rc) return rc;
    if (!d_energy) {
        set_error("d_energy is NULL");
        return SWARM_HIP_EINVAL;
    }
    if (ens->nsys == 0) return SWARM_HIP_OK;
    hipLaunchKernelGGL(energy_kernel, dim3(grid_for(ens->nsys, 256)), dim3(256), 0, (hipStream_t)stream,
                       view_of(ens), d_energy);
    return check_launch("energy kernel launch");
}

int swarm_hip_energy_stats(const swarm_hip_ensemble_t *ens, const double *d_e0, const double *d_e1,
                           double *d_out, void *stream) {
    int rc = check_ens(ens);
    if (rc) return rc;
    if (!d_e0 || !d_e1 || !d_out) {
        set_error("energy_stats: NULL device pointer");
        return SWARM_HIP_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(stats_init_kernel, dim3(1), dim3(1), 0, st, d_out, ens->nsys);
    if (ens->nsys > 0)
        hipLaunchKernelGGL(stats_kernel, dim3(grid_for(ens->nsys, 256)), dim3(256), 0, st, view_of(ens), d_e0, d_e1,
                           d_out);
    return check_launch("stats kernel launch");
}

static int ref_layout(const swarm_hip_ensemble_t *ens, void *d_ref_body, void *d_ref_sys, int32_t ref_chunk,
                      void *stream, bool import) {
    int rc = check_ens(ens);
    if (rc) return rc;
    if (!d_ref_body || !d_ref_sys) {
        set_error("reference layout pointers must not be NULL");
        return SWARM_HIP_EINVAL;
    }
    if (ref_chunk < 1 || ref_chunk > W || (W % ref_chunk) != 0) {
        set_error("ref_chunk=%d must divide %d", ref_chunk, W);
        return SWARM_HIP_EINVAL;
    }
    if (ens->nsys == 0) return SWARM_HIP_OK;
    const int nchunk = (int)swarm_hip_nchunk(ens->nsys);
    const size_t lds = (size_t)ens->nbod * (7 + ens->nbattr) * W * sizeof(double);
    if (lds > 160 * 1024) {
        set_error("wave chunk of %zu bytes does not fit in LDS", lds);
        return SWARM_HIP_EUNSUPPORTED;
    }
    EnsView v = view_of(ens);
    if (import) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&ref_layout_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ref_layout_kernel<true>, dim3(nchunk), dim3(W), lds, (hipStream_t)stream, v,
                           (double *)d_ref_body, (double *)d_ref_sys, (int)ref_chunk);
    } else {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&ref_layout_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ref_layout_kernel<false>, dim3(nchunk), dim3(W), lds, (hipStream_t)stream, v,
                           (double *)d_ref_body, (double *)d_ref_sys, (int)ref_chunk);
    }
    return check_launch("ref layout kernel launch");
}

int swarm_hip_import_ref_layout(const swarm_hip_ensemble_t *ens, const void *d_ref_body, const void *d_ref_sys,
                                int32_t ref_chunk, void *stream) {
    return ref_layout(ens, const_cast<void *>(d_ref_body), const_cast<void *>(d_ref_sys), ref_chunk, stream, true);
}
int swarm_hip_export_ref_layout(const swarm_hip_ensemble_t *ens, void *d_ref_body, void *d_ref_sys,
                                int32_t ref_chunk, void *stream) {
    return ref_layout(ens, d_ref_body, d_ref_sys, ref_chunk, stream, false);
}

// ---- host AoS <-> device wave chunks ---------------------------------------------------------
int swarm_hip_upload_aos(const swarm_hip_ensemble_t *ens, const double *pos, const double *vel, const double *mass,
                         const double *time, const int32_t *state, const int32_t *id, const double *body_attr,
                         const double *sys_attr) {
    int rc = check_ens(ens);
    if (rc) return rc;
    if (!pos || !vel || !mass || !time || !state || !id) {
        set_error("upload_aos: NULL host array");
        return SWARM_HIP_EINVAL;
    }
    if (ens->nsys == 0) return SWARM_HIP_OK;
    const size_t nsys = ens->nsys, n = ens->nbod;
    AosStage st;
    std::unique_lock<std::mutex> hold;
    rc = aos_stage_alloc(ens, st, hold);
    if (rc) return rc;
    hipError_t err = hipSuccess;
    auto h2d = [&](void *d, const void *h, size_t bytes) {
        if (err == hipSuccess && h && bytes) err = hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
    };
    h2d(st.pos, pos, nsys * n * 3 * sizeof(double));
    h2d(st.vel, vel, nsys * n * 3 * sizeof(double));
    h2d(st.mass, mass, nsys * n * sizeof(double));
    h2d(st.time, time, nsys * sizeof(double));
    h2d(st.battr, body_attr, nsys * n * ens->nbattr * sizeof(double));
    h2d(st.sattr, sys_attr, nsys * ens->nsattr * sizeof(double));
    h2d(st.state, state, nsys * sizeof(int32_t));
    h2d(st.id, id, nsys * sizeof(int32_t));
    if (err == hipSuccess) {
        swh::launch_aos_convert<true>(ens, st, body_attr != nullptr, sys_attr != nullptr);
        err = hipGetLastError();
        if (err == hipSuccess) err = hipDeviceSynchronize();
    }
    aos_stage_release(st);
    HIP_TRY(err);
    return SWARM_HIP_OK;
}

int swarm_hip_download_aos(const swarm_hip_ensemble_t *ens, double *pos, double *vel, double *mass, double *time,
                           int32_t *state, int32_t *id, double *body_attr, double *sys_attr) {
    int rc = check_ens(ens);
    if (rc) return rc;
    if (ens->nsys == 0) return SWARM_HIP_OK;
    const size_t nsys = ens->nsys, n = ens->nbod;
    AosStage st;
    std::unique_lock<std::mutex> hold;
    rc = aos_stage_alloc(ens, st, hold);
    if (rc) return rc;
    swh::launch_aos_convert<false>(ens, st, true, true);
    hipError_t err = hipGetLastError();
    auto d2h = [&](void *h, const void *d, size_t bytes) {
        if (err == hipSuccess && h && bytes) err = hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
    };
    d2h(pos, st.pos, nsys * n * 3 * sizeof(double));
    d2h(vel, st.vel, nsys * n * 3 * sizeof(double));
    d2h(mass, st.mass, nsys * n * sizeof(double));
    d2h(time, st.time, nsys * sizeof(double));
    d2h(body_attr, st.battr, nsys * n * ens->nbattr * sizeof(double));
    d2h(sys_attr, st.sattr, nsys * ens->nsattr * sizeof(double));
    d2h(state, st.state, nsys * sizeof(int32_t));
    d2h(id, st.id, nsys * sizeof(int32_t));
    if (err == hipSuccess) err = hipDeviceSynchronize();
    aos_stage_release(st);
    HIP_TRY(err);
    return SWARM_HIP_OK;
}

// ---- device memory convenience ---------------------------------------------------------------
int swarm_hip_malloc(void **d_ptr, size_t bytes) {
    if (!d_ptr) {
        set_error("malloc: NULL out pointer");
        return SWARM_HIP_EINVAL;
    }
    *d_ptr = nullptr;
    if (bytes == 0) return SWARM_HIP_OK;
    HIP_TRY(hipMalloc(d_ptr, bytes));
    return SWARM_HIP_OK;
}
int swarm_hip_free(void *d_ptr) {
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return SWARM_HIP_OK;
}
static int check_copy(const char *what, const void *a, const void *b, size_t bytes) {
    if (bytes && (!a || !b)) {
        set_error("%s: NULL pointer with %zu bytes", what, bytes);
        return SWARM_HIP_EINVAL;
    }
    return SWARM_HIP_OK;
}
int swarm_hip_memset(void *d_ptr, int value, size_t bytes, void *stream) {
    if (int rc = check_copy("memset", d_ptr, d_ptr, bytes)) return rc;
    if (bytes) HIP_TRY(hipMemsetAsync(d_ptr, value, bytes, (hipStream_t)stream));
    return SWARM_HIP_OK;
}
int swarm_hip_memcpy_d2h(void *host, const void *d_ptr, size_t bytes) {
    if (int rc = check_copy("memcpy_d2h", host, d_ptr, bytes)) return rc;
    if (bytes) HIP_TRY(hipMemcpy(host, d_ptr, bytes, hipMemcpyDeviceToHost));
    return SWARM_HIP_OK;
}
int swarm_hip_memcpy_h2d(void *d_ptr, const void *host, size_t bytes) {
    if (int rc = check_copy("memcpy_h2d", d_ptr, host, bytes)) return rc;
    if (bytes) HIP_TRY(hipMemcpy(d_ptr, host, bytes, hipMemcpyHostToDevice));
    return SWARM_HIP_OK;
}
int swarm_hip_host_alloc(void **host_ptr, size_t bytes) {
    if (!host_ptr) {
        set_error("host_alloc: NULL out pointer");
        return SWARM_HIP_EINVAL;
    }
    *host_ptr = nullptr;
    if (bytes == 0) return SWARM_HIP_OK;
    HIP_TRY(hipHostMalloc(host_ptr, bytes, hipHostMallocDefault));
    return SWARM_HIP_OK;
}
int swarm_hip_host_free(void *host_ptr) {
    if (host_ptr) HIP_TRY(hipHostFree(host_ptr));
    return SWARM_HIP_OK;
}
int swarm_hip_synchronize(void *stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return SWARM_HIP_OK;
}
int swarm_hip_event_create(void **event) {
    if (!event) {
        set_error("event_create: NULL out pointer");
        return SWARM_HIP_EINVAL;
    }
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    *event = (void *)ev;
    return SWARM_HIP_OK;
}
int swarm_hip_event_record(void *event, void *stream) {
    HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return SWARM_HIP_OK;
}
int swarm_hip_event_elapsed_ms(void *start, void *stop, float *ms) {
    if (!start || !stop || !ms) {
        set_error("event_elapsed_ms: NULL argument");
        return SWARM_HIP_EINVAL;
    }
    HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SWARM_HIP_OK;
}
int swarm_hip_event_destroy(void *event) {
    HIP_TRY(hipEventDestroy((hipEvent_t)event));
    return SWARM_HIP_OK;
}
int swarm_hip_event_synchronize(void *event) {
    HIP_TRY(hipEventSynchronize((hipEvent_t)event));
    return SWARM_HIP_OK;
}
// ---- a second stream for the log flush ------------------------------------------
int swarm_hip_stream_create(void **stream) {
    if (!stream) {
        set_error("stream_create: NULL out pointer");
        return SWARM_HIP_EINVAL;
    }
    hipStream_t st;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); // no implicit ordering against the NULL stream
    *stream = (void *)st;
    return SWARM_HIP_OK;
}
int swarm_hip_stream_destroy(void *stream) {
    if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return SWARM_HIP_OK;
}
int swarm_hip_stream_wait_event(void *stream, void *event) {
    if (!event) {
        set_error("stream_wait_event: NULL event");
        return SWARM_HIP_EINVAL;
    }
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return SWARM_HIP_OK;
}
int swarm_hip_memcpy_d2h_async(void *host, const void *d_ptr, size_t bytes, void *stream) {
    if (int rc = check_copy("memcpy_d2h_async", host, d_ptr, bytes)) return rc;
    if (bytes) HIP_TRY(hipMemcpyAsync(host, d_ptr, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SWARM_HIP_OK;
}

} // extern "C"
