// Micro-benchmark: how does the vector L1 (TCP) of gfx950 price a per-lane gather of small table rows when
// neighbouring lanes cooperate on one row?  Every "pair" needs one ROW bytes record at a random row of an
// L2-resident table.
//   mode 0: every lane fetches its own record with ROW/16 dwordx4 loads (what the AEAM tile kernels do).
//   mode 1: the G = ROW/16 lanes of a group fetch 16 B each of ONE record per instruction, G instructions serve
//           the G lanes' records, then a butterfly transpose (v_cndmask + DPP) hands every lane its record.
// Reported: ns per record per CU-lane and records/clk/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s; }

template <int ROW>  // bytes per record: 32 or 64
__global__ __launch_bounds__(256) void gather_own(const double2 *__restrict__ tab, int nrows, int iters, double *out)
{
    constexpr int Q = ROW / 16;
    unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        unsigned row = lcg(s) % (unsigned)nrows;
        const double2 *p = tab + (size_t)row * Q;
        double2 v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) v[q] = p[q];
#pragma unroll
        for (int q = 0; q < Q; ++q) acc = fma(v[q].x, 1.0000001, acc) + v[q].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__device__ __forceinline__ double2 quad_xor(double2 v, int m)
{
    // ds_swizzle-free lane exchange inside a quad
    double2 r;
    r.x = __shfl_xor(v.x, m, 64);
    r.y = __shfl_xor(v.y, m, 64);
    return r;
}

template <int ROW>
__global__ __launch_bounds__(256) void gather_coop(const double2 *__restrict__ tab, int nrows, int iters, double *out)
{
    constexpr int Q = ROW / 16;
    unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (Q - 1);
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        unsigned row = lcg(s) % (unsigned)nrows;
        double2 v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            // row of lane q of my group
            unsigned rq = __shfl(row, (lane & ~(Q - 1)) | q, 64);
            v[q] = tab[(size_t)rq * Q + sub];
        }
        // transpose v[q] (q = whose record) x sub (which piece) inside the group
        if (Q == 2) {
            double2 send = sub ? v[0] : v[1];
            double2 got = quad_xor(send, 1);
            if (sub) v[0] = got; else v[1] = got;
        } else {
#pragma unroll
            for (int m = 1; m < Q; m <<= 1) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    if (q & m) continue;
                    bool up = sub & m;
                    double2 send = up ? v[q] : v[q | m];
                    double2 got = quad_xor(send, m);
                    if (up) v[q] = got; else v[q | m] = got;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) acc = fma(v[q].x, 1.0000001, acc) + v[q].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}


template <int POLICY>
__device__ __forceinline__ double2 ld16(const double2 *p)
{
    double2 v;
    if (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int STRIDE, int PAY, int POLICY>
__global__ __launch_bounds__(256) void gather_pol(const double2 *__restrict__ tab, int nrows, int iters, double *out)
{
    unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        unsigned row = lcg(s) % (unsigned)nrows;
        const double2 *p = (const double2 *)((const char *)tab + (size_t)row * STRIDE);
        double2 v[PAY];
#pragma unroll
        for (int q = 0; q < PAY; ++q) v[q] = ld16<POLICY>(p + q);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < PAY; ++q) acc = fma(v[q].x, 1.0000001, acc) + v[q].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int STRIDE, int PAY, int POLICY>
static void runp(const char *name, double2 *tab, int nrows, double *out, int blocks, int iters)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(a));
        gather_pol<STRIDE, PAY, POLICY><<<blocks, 256>>>(tab, nrows, iters, out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double recs = (double)blocks * 256 * iters;
    printf("%-34s rows=%6d  %8.3f ms  %7.2f Grec/s  %6.3f rec/clk/CU\n", name, nrows, ms, recs / ms * 1e-6, recs / (ms * 1e-3) / 2.4e9 / 256.0);
}

// check: the cooperative version must deliver exactly the record a lane would have fetched itself
template <int ROW>
static void run(const char *name, int mode, double2 *tab, int nrows, double *out, int blocks, int iters, std::vector<double> &ref)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(a));
        if (mode == 0) gather_own<ROW><<<blocks, 256>>>(tab, nrows, iters, out);
        else gather_coop<ROW><<<blocks, 256>>>(tab, nrows, iters, out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
    }
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<double> h((size_t)blocks * 256);
    CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    const char *ok = "";
    if (mode == 0) ref = h; else { ok = " same-as-own"; for (size_t i = 0; i < h.size(); ++i) if (h[i] != ref[i]) { ok = " MISMATCH"; break; } }
    double recs = (double)blocks * 256 * iters;
    double clk = 2.4e9;  // nominal
    printf("%-28s rows=%6d  %8.3f ms  %7.2f Grec/s  %6.3f rec/clk/CU%s\n", name, nrows, ms, recs / ms * 1e-6, recs / (ms * 1e-3) / clk / 256.0, ok);
}

int main()
{
    const int blocks = 256 * 16, iters = 512;
    double *out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    for (int nrows : {64, 1000, 6500, 70000}) {
        size_t bytes = (size_t)nrows * 128;
        std::vector<double> h(bytes / 8);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (double)(i % 977) * 1e-3;
        double2 *tab; CHECK(hipMalloc(&tab, bytes));
        CHECK(hipMemcpy(tab, h.data(), bytes, hipMemcpyHostToDevice));
        std::vector<double> ref;
        run<32>("own  32 B (2 x dwordx4)", 0, tab, nrows, out, blocks, iters, ref);
        run<32>("pair 32 B (2 lanes x 16 B)", 1, tab, nrows, out, blocks, iters, ref);
        run<64>("own  64 B (4 x dwordx4)", 0, tab, nrows, out, blocks, iters, ref);
        run<64>("quad 64 B (4 lanes x 16 B)", 1, tab, nrows, out, blocks, iters, ref);
        runp<32, 2, 0>("asm 32/32 default", tab, nrows, out, blocks, iters);
        runp<32, 2, 1>("asm 32/32 nt", tab, nrows, out, blocks, iters);
        runp<32, 2, 2>("asm 32/32 sc0", tab, nrows, out, blocks, iters);
        runp<32, 2, 3>("asm 32/32 sc1", tab, nrows, out, blocks, iters);
        runp<32, 2, 4>("asm 32/32 sc0 sc1", tab, nrows, out, blocks, iters);
        runp<32, 2, 5>("asm 32/32 sc0 nt", tab, nrows, out, blocks, iters);
        runp<48, 3, 0>("asm 48 B stride 48", tab, nrows, out, blocks, iters);
        runp<64, 3, 0>("asm 48 B stride 64", tab, nrows, out, blocks, iters);
        runp<64, 3, 1>("asm 48 B stride 64 nt", tab, nrows, out, blocks, iters);
        runp<64, 3, 4>("asm 48 B stride 64 sc0 sc1", tab, nrows, out, blocks, iters);
        runp<16, 1, 0>("asm 16 B stride 16", tab, nrows, out, blocks, iters);
        runp<128, 1, 0>("asm 16 B stride 128", tab, nrows, out, blocks, iters);
        CHECK(hipFree(tab));
    }
    return 0;
}
