// valu_rate -- what one SIMD of gfx950 sustains for the instruction kinds the warp stage is made of, by waves per
// SIMD: decides whether k_levels (61.8 M VALU wave-instructions per 159 us launch) sits at the VALU issue limit or
// far below it.  Each wave runs a long unrolled stream of INDEPENDENT instructions of one kind (8 accumulators), so
// dependencies do not limit it.  Prints wave-instructions per cycle per SIMD (clock from s_memtime / wall time).
// Build: hipcc --offload-arch=gfx950 -O2 tools/cpp/valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
constexpr int ITER = 400, REP = 16;      // x REP x 8 independent instructions per iteration (loop overhead < 3 %)

template <int KIND>
__global__ void k_rate(float* out, unsigned long long* cyc, float seed)
{
    float a[8]; double d[8]; int n[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; d[i] = a[i]; n[i] = (int)a[i]; p[i] = f2{ a[i], a[i] + 1 }; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int ii = 0; ii < 8 * REP; ii++) {
            const int i = ii & 7;
            // inline asm: exactly one instruction of the named kind per item, nothing for the compiler to pack or fold
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(1.0001f), "v"(0.5f));
            if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(1.0001), "v"(0.5));
            if (KIND == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(3));
            if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f2{ 1.0001f, 1.0002f }), "v"(f2{ 0.5f, 0.25f }));
            if (KIND == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(0.5));
            if (KIND == 5) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[i]) : "v"(n[i]));
            if (KIND == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(1.0001));
            if (KIND == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(3));
            if (KIND == 8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(n[(i + 1) & 7]), "v"(0x06050403));
            if (KIND == 9) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(d[i]));
            if (KIND == 10) asm volatile("v_rcp_f64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (KIND == 11) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(1.0001f));
            if (KIND == 12) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(n[i]) : "v"(3));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; i++) s += a[i] + (float)d[i] + n[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> int run(const char* name, int insts_per_item)
{
    float* out; unsigned long long* cyc;
    CK(hipMalloc((void**)&out, 256 * 8 * 1024 * 4)); CK(hipMalloc((void**)&cyc, 2048 * 8));
    std::printf("%-28s", name);
    for (int wps : { 1, 2, 4, 8 }) {                 // waves per SIMD: one block per CU of 4*wps waves
        const int threads = 64 * 4 * wps;            // up to 2048 -> two blocks of 1024
        const int bpc = threads > 1024 ? 2 : 1, th = threads / bpc;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_rate<KIND>, dim3(256 * bpc), dim3(th), 0, 0, out, cyc, 1.0f);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_rate<KIND>, dim3(256 * bpc), dim3(th), 0, 0, out, cyc, 1.0f);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> c(256 * bpc); CK(hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost));
        double mean = 0; for (auto v : c) mean += (double)v; mean /= c.size();
        // s_memtime ticks at a fixed 100 MHz on gfx9? report both: per-tick and per-wall-time at 2.4 GHz
        const double insts = (double)ITER * 8 * REP * insts_per_item * wps;          // per SIMD
        std::printf("  w/SIMD %d: %6.3f inst/clk@2.4GHz (%.1f us, %.0f ticks)", wps, insts / (ms * 1e-3 * 2.4e9), ms * 1e3, mean);
    }
    std::printf("\n");
    CK(hipFree(out)); CK(hipFree(cyc));
    return 0;
}

int main()
{
    run<0>("v_fma_f32", 1); run<11>("v_mul_f32", 1); run<7>("v_add_u32", 1); run<12>("v_mul_u32_u24", 1); run<2>("v_mul_lo_u32", 1);
    run<8>("v_perm_b32", 1); run<5>("v_cvt_f32_ubyte0", 1); run<3>("v_pk_fma_f32", 1);
    run<1>("v_fma_f64", 1); run<4>("v_add_f64", 1); run<6>("v_mul_f64", 1); run<9>("v_cvt_i32_f64", 1); run<10>("v_rcp_f64", 1);
    return 0;
}
