// Micro-benchmark 3: spline evaluation from an LDS-resident (value, slope) table, 16 bytes per row.
// A record = rows m and m+1 (32 contiguous bytes); the cubic's coefficients are the Hermite expressions of
// (Y0, S0, Y1, S1).  Rows uniformly random in the window (worst case for bank conflicts).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }

template <int THREADS, int UNROLL>
__global__ __launch_bounds__(THREADS) void lds_eval(const double2 *__restrict__ tab, int nrows, int iters, double *out)
{
    extern __shared__ double2 s_tab[];
    for (int i = threadIdx.x; i <= nrows; i += THREADS) s_tab[i] = tab[i];
    __syncthreads();
    unsigned s = (blockIdx.x * THREADS + threadIdx.x) * 2654435761u + 12345u;
    double acc = 0.0;
    for (int it = 0; it < iters; it += UNROLL) {
        double2 a[UNROLL], b[UNROLL];
        double pf[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            unsigned x = lcg(s);
            unsigned row = (x >> 8) % (unsigned)nrows;
            pf[u] = (double)(x & 255) * (1.0 / 256.0);
            a[u] = s_tab[row];
            b[u] = s_tab[row + 1];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const double d = b[u].x - a[u].x;
            const double c4 = 3.0 * d - 2.0 * a[u].y - b[u].y;
            const double c3 = a[u].y + b[u].y - 2.0 * d;
            acc += ((c3 * pf[u] + c4) * pf[u] + a[u].y) * pf[u] + a[u].x;
        }
    }
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

template <int THREADS, int UNROLL>
static void run(int nrows, double2 *tab, double *out, int wg_per_cu)
{
    const int blocks = 256 * wg_per_cu, iters = 4096;
    size_t lds = (size_t)(nrows + 1) * 16;
    CHECK(hipFuncSetAttribute((const void *)lds_eval<THREADS, UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(a));
        lds_eval<THREADS, UNROLL><<<blocks, THREADS, lds>>>(tab, nrows, iters, out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double recs = (double)blocks * THREADS * iters;
    printf("rows=%5d (%5.1f KB)  threads=%4d x %d WG/CU  unroll=%d  %8.3f ms  %6.3f rec/clk/CU\n", nrows, lds / 1024.0, THREADS, wg_per_cu, UNROLL, ms,
           recs / (ms * 1e-3) / 2.4e9 / 256.0);
}

int main()
{
    const int nmax = 6500;
    std::vector<double> h((size_t)(nmax + 1) * 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)(i % 977) * 1e-3;
    double2 *tab; CHECK(hipMalloc(&tab, h.size() * 8));
    CHECK(hipMemcpy(tab, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    double *out; CHECK(hipMalloc(&out, (size_t)256 * 8 * 1024 * 8));
    run<1024, 1>(6460, tab, out, 1);
    run<1024, 2>(6460, tab, out, 1);
    run<1024, 4>(6460, tab, out, 1);
    run<512, 1>(6460, tab, out, 1);
    run<512, 4>(6460, tab, out, 1);
    run<256, 4>(6460, tab, out, 1);
    run<512, 1>(3077, tab, out, 2);
    run<512, 2>(3077, tab, out, 2);
    run<512, 4>(3077, tab, out, 2);
    run<256, 2>(3077, tab, out, 3);
    run<1024, 2>(3077, tab, out, 1);
    run<1024, 2>(3077, tab, out, 2);
    return 0;
}
