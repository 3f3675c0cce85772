// proto_xcd_barrier.hip -- what does a phase of a persistent multi-workgroup kernel cost when all of its workgroups sit on
// ONE XCD (one shared L2) instead of anywhere on the chip?  Stand-alone micro-benchmark behind the small-grid frozen year
// (DESIGN.md section 3d): G workgroups of 4 waves exchange one 512-byte column per wave and phase with their neighbours.
//
//   mode 0  workers anywhere (the first G workgroups): write-through (sc1) stores, L1-bypassing (sc1) loads, arrival counter
//           with agent-scope adds in 32 shards -- the barrier of k_year_persistent
//   mode 1  workers on XCD 0 only (HW_REG_XCC_ID; 8 G + 64 workgroups launched, those elsewhere exit, those on XCD 0 take
//           tickets): PLAIN stores (they stay in the XCD's L2), sc1 loads (L2-served), ONE arrival counter with
//           workgroup-scope adds (executed in that L2)
//
// every value is checked against the host's recurrence, so a stale read shows.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Args {
    double* buf[2];      // [ncol][64] ping-pong
    unsigned* arrive;    // mode 0: 32 shards on 128-byte lines; mode 1: one word
    unsigned* tickets;
    int* abort_flag;
    unsigned* census;    // [8] workgroups seen per XCD
    int G, phases, mode;
    long long spin_ticks;
};

__device__ __forceinline__ double ld_sc1(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256) k_phases(Args A) {
    __shared__ int s_id, s_ok;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        atomicAdd(A.census + xcc, 1u);
        int id = -1;
        if (A.mode == 0) id = ((int)blockIdx.x < A.G) ? (int)blockIdx.x : -1;
        else if (xcc == 0u) {
            const unsigned t = __hip_atomic_fetch_add(A.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            id = (t < (unsigned)A.G) ? (int)t : -1;
        }
        s_id = id;
    }
    __syncthreads();
    const int id = s_id;
    if (id < 0) return;
    const int ncol = A.G * 4, col = id * 4 + wave;
    const int left = (col + ncol - 1) % ncol, right = (col + 1) % ncol;
    unsigned epoch = 0;
    for (int p = 0; p < A.phases; ++p) {
        const double* src = A.buf[p & 1];
        double* dst = A.buf[(p + 1) & 1];
        const double a = ld_sc1(src + (size_t)left * 64 + lane), b = ld_sc1(src + (size_t)right * 64 + lane),
                     c = ld_sc1(src + (size_t)col * 64 + lane);
        const double v = 0.25 * a + 0.25 * b + 0.5 * c + 1.0;
        if (A.mode == 0) st_sc1(dst + (size_t)col * 64 + lane, v);
        else dst[(size_t)col * 64 + lane] = v;
        // ---- grid barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x < 64) {
            const unsigned target = (epoch + 1u) * (unsigned)A.G;
            if (lane == 0) {
                if (A.mode == 0) __hip_atomic_fetch_add(A.arrive + (size_t)(id % 32) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_fetch_add(A.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            int good = 1;
            long long spins = 0;
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            for (;;) {
                unsigned v2 = 0u;
                if (A.mode == 0) {
                    if (lane < 32) v2 = __hip_atomic_load(A.arrive + (size_t)lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int o = 32; o > 0; o >>= 1) v2 += __shfl_down(v2, o, 64);
                } else if (lane == 0) {
                    v2 = __hip_atomic_load(A.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const unsigned total = (unsigned)__builtin_amdgcn_readfirstlane((int)v2);
                if (total >= target) break;
                const int ab = __builtin_amdgcn_readfirstlane(
                    (lane == 0) ? __hip_atomic_load(A.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
                if (ab != 0 || ((++spins & 63) == 0 && (long long)__builtin_amdgcn_s_memrealtime() - t0 > A.spin_ticks)) {
                    if (lane == 0) __hip_atomic_store(A.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = 0;
                    break;
                }
            }
            if (lane == 0) s_ok = good;
        }
        __syncthreads();
        ++epoch;
        if (!s_ok) return;
    }
}

int main(int argc, char** argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 2000;
    for (int G : {13, 33, 52}) {
        for (int mode = 0; mode < 2; ++mode) {
            const int ncol = G * 4;
            Args A = {};
            A.G = G; A.phases = phases; A.mode = mode; A.spin_ticks = 100000000LL;   // 1 s
            CHECK(hipMalloc((void**)&A.buf[0], sizeof(double) * ncol * 64));
            CHECK(hipMalloc((void**)&A.buf[1], sizeof(double) * ncol * 64));
            CHECK(hipMalloc((void**)&A.arrive, 4096 * 4));
            CHECK(hipMalloc((void**)&A.tickets, 256));
            CHECK(hipMalloc((void**)&A.abort_flag, 256));
            CHECK(hipMalloc((void**)&A.census, 256));
            std::vector<double> x((size_t)ncol * 64), y(x.size());
            for (size_t i = 0; i < x.size(); ++i) x[i] = (double)(i % 97) * 0.01;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            double best = 1e30;
            bool ok = true;
            unsigned census[8] = {0};
            int aborted = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipMemcpy(A.buf[0], x.data(), sizeof(double) * x.size(), hipMemcpyHostToDevice));
                CHECK(hipMemset(A.arrive, 0, 4096 * 4));
                CHECK(hipMemset(A.tickets, 0, 256));
                CHECK(hipMemset(A.abort_flag, 0, 256));
                CHECK(hipMemset(A.census, 0, 256));
                const int grid = mode == 0 ? G : 8 * G + 64;
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_phases, dim3(grid), dim3(256), 0, 0, A);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CHECK(hipMemcpy(&aborted, A.abort_flag, 4, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(census, A.census, 32, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(y.data(), A.buf[phases & 1], sizeof(double) * y.size(), hipMemcpyDeviceToHost));
            }
            // host recurrence
            std::vector<double> a(x), b(x.size());
            for (int p = 0; p < phases; ++p) {
                for (int c = 0; c < ncol; ++c)
                    for (int l = 0; l < 64; ++l) {
                        const int le = (c + ncol - 1) % ncol, ri = (c + 1) % ncol;
                        b[(size_t)c * 64 + l] = 0.25 * a[(size_t)le * 64 + l] + 0.25 * a[(size_t)ri * 64 + l] + 0.5 * a[(size_t)c * 64 + l] + 1.0;
                    }
                a.swap(b);
            }
            size_t bad = 0;
            for (size_t i = 0; i < a.size(); ++i) bad += a[i] != y[i];
            ok = bad == 0 && !aborted;
            printf("G=%2d workgroups (%3d columns) mode %d (%s): %.3f us per phase over %d phases; results %s (%zu wrong), aborted %d; "
                   "workgroups per XCD of the last launch:", G, ncol, mode, mode == 0 ? "anywhere, agent scope" : "XCD 0 only, L2 scope",
                   1000.0 * best / phases, phases, ok ? "exact" : "WRONG", bad, aborted);
            for (int i = 0; i < 8; ++i) printf(" %u", census[i]);
            printf("\n");
            hipFree(A.buf[0]); hipFree(A.buf[1]); hipFree(A.arrive); hipFree(A.tickets); hipFree(A.abort_flag); hipFree(A.census);
        }
    }
    return 0;
}
