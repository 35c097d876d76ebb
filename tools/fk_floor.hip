// What does k_fk's memory pattern cost without its arithmetic?  56 B of q in and 128 B of pose out per configuration, same LDS
// staging and transposed stores, three launch shapes.   hipcc --offload-arch=gfx950 -O3 tools/fk_floor.hip -o build_tmp/fk_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int WAVES, bool PERSIST>
__global__ __launch_bounds__(64 * WAVES) void k_floor(const double* __restrict__ q, long B, double* __restrict__ T_out, int nq) {
    extern __shared__ double lds_all[];
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    double* lds = lds_all + wave * 64 * 17;
    const long nblk = (B + 63) / 64;
    for (long blk = (long)blockIdx.x * WAVES + wave; blk < nblk; blk += (long)gridDim.x * WAVES) {
        const long base = blk * 64;
        const double2* s2 = reinterpret_cast<const double2*>(q + base * nq);
        double2* d2 = reinterpret_cast<double2*>(lds);
        for (int i = lane; i < 64 * nq / 2; i += 64) d2[i] = s2[i];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double v[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = lds[lane * nq + j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double* row = lds + lane * 17;
#pragma unroll
        for (int e = 0; e < 16; ++e) row[e] = v[e % 7] + (double)e;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        double2* dst = reinterpret_cast<double2*>(T_out + base * 16);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int g = lane + 64 * kk;
            const int r = g >> 3, c2 = (g & 7) * 2;
            double2 o; o.x = lds[r * 17 + c2]; o.y = lds[r * 17 + c2 + 1];
            dst[g] = o;
        }
        if (!PERSIST) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int WAVES, bool PERSIST>
static void run(const char* name, const double* q, long B, double* out, int grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t lds = sizeof(double) * 64 * 17 * WAVES;
    for (int i = 0; i < 3; ++i) k_floor<WAVES, PERSIST><<<grid, 64 * WAVES, lds>>>(q, B, out, 7);
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int i = 0; i < 20; ++i) {
        hipEventRecord(a);
        k_floor<WAVES, PERSIST><<<grid, 64 * WAVES, lds>>>(q, B, out, 7);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("%-44s grid %6d: median %.4f ms  min %.4f ms  -> %.0f GB/s\n", name, grid, t[10], t[0], B * 184.0 / (t[10] * 1e-3) / 1e9);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const long B = 1000000;
    double *q, *out;
    hipMalloc(&q, 5 * B * 56); hipMalloc(&out, B * 128);
    hipMemset(q, 0, 5 * B * 56);
    const long nblk = (B + 63) / 64;
    run<1, false>("one wave per workgroup, one block each", q, B, out, (int)nblk);
    run<4, false>("four waves per workgroup, one block per wave", q, B, out, (int)((nblk + 3) / 4));
    run<1, true>("persistent, one wave per workgroup", q, B, out, 256 * 16);
    run<1, true>("persistent, one wave per workgroup", q, B, out, 256 * 32);
    run<4, true>("persistent, four waves per workgroup", q, B, out, 256 * 4);
    run<4, true>("persistent, four waves per workgroup", q, B, out, 256 * 8);
    return 0;
}
