// Micro-benchmark 2: vector-L1 behaviour of spline-row gathers when the lanes of a workgroup work on the same
// band of rows at the same time (neighbour rows sorted by distance at build time), for the record layouts that
// are candidates for the AEAM tile kernels.  Workgroups resident on one CU are out of phase with each other.
//   STRIDE = bytes between rows, PAY = 16-byte pieces fetched per record (contiguous from the row start).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }

template <int STRIDE, int PAY>
__global__ __launch_bounds__(256) void gather_band(const char *__restrict__ tab, int nrows, int band, int iters, double *out)
{
    extern __shared__ char pad[];
    unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    const int nband = nrows / band;
    unsigned phase = (blockIdx.x * 2654435761u) >> 8;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        unsigned b = (phase + it) % (unsigned)nband;
        unsigned row = b * band + lcg(s) % (unsigned)band;
        const double2 *p = (const double2 *)(tab + (size_t)row * STRIDE);
        double2 v[PAY];
#pragma unroll
        for (int q = 0; q < PAY; ++q) v[q] = p[q];
#pragma unroll
        for (int q = 0; q < PAY; ++q) acc = fma(v[q].x, 1.0000001, acc) + v[q].y;
    }
    if (acc == 1.2345) pad[threadIdx.x] = 1;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int STRIDE, int PAY>
static void run(const char *name, char *tab, int nrows, int band, double *out, int lds)
{
    const int blocks = 256 * 20, iters = 7 * 64;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(a));
        gather_band<STRIDE, PAY><<<blocks, 256, lds>>>(tab, nrows, band, iters, out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double recs = (double)blocks * 256 * iters;
    printf("%-30s band=%5d rows (%6.1f KB)  %8.3f ms  %6.3f rec/clk/CU\n", name, band, band * (double)STRIDE / 1024, ms, recs / (ms * 1e-3) / 2.4e9 / 256.0);
}

int main()
{
    const int nrows = 6400;
    double *out; CHECK(hipMalloc(&out, (size_t)256 * 20 * 256 * 8));
    size_t bytes = (size_t)(nrows + 8) * 64;
    std::vector<double> h(bytes / 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)(i % 977) * 1e-3;
    char *tab; CHECK(hipMalloc(&tab, bytes));
    CHECK(hipMemcpy(tab, h.data(), bytes, hipMemcpyHostToDevice));
    for (int lds : {30 * 1024, 16 * 1024}) {
        printf("--- %d KB LDS per workgroup (%d workgroups per CU)\n", lds / 1024, 160 * 1024 / lds > 8 ? 8 : 160 * 1024 / lds);
        for (int band : {6400, 3200, 1600, 800, 400, 200}) {
            run<32, 2>("32 B rec, stride 32 (density)", tab, nrows, band, out, lds);
            run<16, 2>("32 B rec, stride 16 (Y,S)", tab, nrows, band, out, lds);
            run<48, 3>("48 B rec, stride 48 (force)", tab, nrows, band, out, lds);
            run<32, 4>("64 B rec, stride 32 (Y,S x2)", tab, nrows, band, out, lds);
            run<32, 3>("48 B rec, stride 32 (dY,S x2)", tab, nrows, band, out, lds);
        }
    }
    return 0;
}
