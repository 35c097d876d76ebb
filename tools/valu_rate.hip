// Issue-rate micro-benchmark for the vector / scalar ALUs of gfx950 (what the VALU roof in bench.py and tools/reduce_pmc.py rests on).
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Every wave runs ITER trips of a block of independent instructions of one kind (16 accumulators: no dependency stalls); the grid
// puts W waves on every SIMD of every CU (256 CUs x 4 SIMDs; one 64-lane wave per workgroup).  Reported: wave-instructions per
// SIMD per microsecond and, at the clock measured in the same launch (s_memtime over s_memrealtime at 100 MHz), SIMD cycles per
// wave64 instruction.  W = 1 shows what ONE wave's stream sustains, W >= 2 what the SIMD does with waves to interleave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 4096

template <int KIND>
__global__ __launch_bounds__(64) void k_rate(float* out, unsigned long long* clk, float seed, int iters, int si) {
    float a[16];
    double d[8];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = seed + i; p[i] = v2f{seed + i, seed - i}; }
    const float m = seed * 0.5f, c = seed * 0.25f;
    const double md = seed * 0.5, cd = seed * 0.25;
    int s0 = si, s1 = 3, s2 = 5, s3 = 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
        } else if (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
        } else if (KIND == 3) {
            // scalar ALU: 16 independent-ish s_add / s_xor
#pragma unroll
            for (int i = 0; i < 4; ++i)
                asm volatile("s_add_i32 %0, %0, %4\n s_xor_b32 %1, %1, %4\n s_add_i32 %2, %2, %4\n s_xor_b32 %3, %3, %4"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(it) : "scc");
        } else if (KIND == 4) {
            // the mix of a compare-heavy stage: v_sub, v_fma, v_cmp (to an SGPR pair), s_or of the masks
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned long long mk;
                asm volatile("v_sub_f32 %0, %0, %2\n v_fma_f32 %1, %0, %0, %1" : "+v"(a[2 * i]), "+v"(a[2 * i + 1]) : "v"(c));
                asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(mk) : "v"(a[2 * i + 1]), "v"(m));
                asm volatile("s_or_b32 %0, %0, %1" : "+s"(s0) : "s"((int)mk) : "scc");
            }
        } else if (KIND == 5) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
        } else if (KIND == 6) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += (float)d[i] + p[i].x + p[i].y;
    if (acc == 123.456f) out[0] = acc;
    if (s0 + s1 + s2 + s3 == 123456789) out[1] = 1.0f;
    if (threadIdx.x == 0) { clk[4 * blockIdx.x] = t1 - t0; clk[4 * blockIdx.x + 1] = r1 - r0; clk[4 * blockIdx.x + 2] = r0; clk[4 * blockIdx.x + 3] = r1; }
}

template <int KIND>
static void run(const char* name, int per_trip, int waves_per_simd) {
    const int grid = 256 * 4 * waves_per_simd;
    float* out; unsigned long long* clk;
    hipMalloc(&out, 8); hipMalloc(&clk, 32 * (size_t)grid);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_rate<KIND><<<grid, 64>>>(out, clk, 1.5f, ITER, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_rate<KIND><<<grid, 64>>>(out, clk, 1.5f, ITER, 1);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(4 * (size_t)grid);
    hipMemcpy(h.data(), clk, 32 * (size_t)grid, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    unsigned long long rmin = ~0ull, rmax = 0;
    for (int i = 0; i < grid; ++i) {
        ghz.push_back((double)h[4 * i] / (double)h[4 * i + 1] * 0.1); cyc.push_back((double)h[4 * i]);
        rmin = std::min(rmin, h[4 * i + 2]); rmax = std::max(rmax, h[4 * i + 3]);
    }
    // waves in flight at the middle of the launch (100 MHz real-time stamps): the residency the rates below really had
    const unsigned long long mid = rmin + (rmax - rmin) / 2;
    int inflight = 0;
    for (int i = 0; i < grid; ++i) if (h[4 * i + 2] <= mid && mid < h[4 * i + 3]) ++inflight;
    const double span_us = (double)(rmax - rmin) * 0.01;
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double clock = ghz[ghz.size() / 2];
    const double insts_per_wave = (double)ITER * per_trip;
    // in-kernel: a wave's own cycles per instruction; the SIMD's cycles per instruction = that / waves on the SIMD
    const double wave_cpi = cyc[cyc.size() / 2] / insts_per_wave;
    const double per_simd_per_us = insts_per_wave * waves_per_simd / (ms * 1e3);
    const double simd_cpi_span = span_us * 1e-6 * clock * 1e9 / (insts_per_wave * waves_per_simd);
    printf("%-16s launched %d waves/SIMD (%.2f in flight mid-launch): events %7.3f ms, in-kernel span %7.1f us, clock %.2f GHz, a wave sees %.2f cyc/inst, "
           "SIMD %.2f cyc per wave64 inst over the span (%.0f inst/SIMD/us by events)\n",
           name, waves_per_simd, inflight / 1024.0, ms, span_us, clock, wave_cpi, simd_cpi_span, per_simd_per_us);
    hipFree(out); hipFree(clk);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    for (int w : {1, 2, 4, 8}) run<0>("v_fma_f32", 16, w);
    for (int w : {1, 2, 4, 8}) run<1>("v_fma_f64", 16, w);
    for (int w : {1, 2, 4, 8}) run<2>("v_pk_fma_f32", 16, w);
    for (int w : {1, 2, 4, 8}) run<5>("v_mov_b32", 16, w);
    for (int w : {1, 2, 4, 8}) run<6>("v_add_u32", 16, w);
    for (int w : {1, 2, 4, 8}) run<3>("salu add/xor", 16, w);
    for (int w : {1, 2, 4, 8}) run<4>("sub/fma/cmp/s_or", 16, w);
    return 0;
}
