// wg_launch_cost -- what a workgroup costs just by existing on gfx950: a grid of N workgroups that each store one LDS
// word per thread, meet at one barrier and exit, for several (threads, LDS bytes) shapes.  k_levels runs ~9400
// workgroups of 512 threads and 51.6 KB LDS per keyframe (3 per CU); its skeleton (PF_ABLATE=3) takes 39 us.
// Build: hipcc --offload-arch=gfx950 -O2 tools/cpp/wg_launch_cost.hip -o wg_launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_wg(int* out, int work)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x + blockIdx.x;
    int acc = 0;
    for (int i = 0; i < work; i++) { acc += lds[(threadIdx.x + i) & (blockDim.x - 1)]; asm volatile("" : "+v"(acc)); }
    __syncthreads();
    if (threadIdx.x == 0 && (blockIdx.x & 1023) == 0) out[blockIdx.x >> 10] = lds[1] + acc;
}

int main()
{
    int* out; CK(hipMalloc((void**)&out, 4096));
    const int shapes[][2] = { { 512, 52 * 1024 }, { 512, 1024 }, { 512, 38 * 1024 }, { 256, 26 * 1024 }, { 256, 1024 }, { 1024, 100 * 1024 }, { 1024, 1024 }, { 64, 1024 } };
    for (auto& sh : shapes) {
        CK(hipFuncSetAttribute((const void*)k_wg, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int work : { 0, 2000 }) {
            const int n = 9408 * 512 / sh[0];
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(k_wg, dim3(n), dim3(sh[0]), sh[1], 0, out, work);
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_wg, dim3(n), dim3(sh[0]), sh[1], 0, out, work);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("threads %4d  LDS %6d B  work %4d  %6d workgroups: %7.2f us per launch\n", sh[0], sh[1], work, n, ms * 1e3 / 5);
        }
    }
    return 0;
}
