// pyrdown_taps -- the measurement behind "the 5-tap reductions read LDS instead of shuffling" (the
// north star names wavefront shuffles for the 5-tap filter reductions).  One workgroup stages a 39 x 72 tile of 16-byte
// pixels (c0, c1, c2, w) in LDS like stage A of the level kernel, then forms the horizontal pyrDown sums
//     h[q] = a[2q]*6 + (a[2q-1] + a[2q+1])*4 + a[2q-2] + a[2q+2]          (per component, cv::pyrDown's order)
// for 34 outputs per row, two ways, REP times:
//   lds     : five ds_read_b128 per output at the taps' addresses (what the kernel does; parity-split layout)
//   shuffle : every lane reads ITS pixel of the row once (one ds_read_b128), the five taps of output q come from lanes
//             2q-2 .. 2q+2 through ds_bpermute_b32 (the only cross-lane primitive that reaches a computed lane across all
//             64 lanes; DPP row shifts stay inside 16 lanes and cannot do the stride-2 gather) -- 4 dwords x 5 taps = 20
//             bpermutes per output, and decimation leaves half the lanes of the 64-wide row without an output.
// Both produce the same sums (checked).  Prints ns per row-sum-wave.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/cpp/pyrdown_taps.hip -o pyrdown_taps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 39, COLS = 72, HALF = 36, NQ = 34, REP = 200;

__device__ __forceinline__ f4 tapsum(f4 a0, f4 a1, f4 a2, f4 a3, f4 a4) { return a2 * 6.f + (a1 + a3) * 4.f + a0 + a4; }

template <bool SHUFFLE>
__global__ __launch_bounds__(512) void k_taps(const f4* __restrict__ in, f4* __restrict__ out)
{
    __shared__ f4 A[ROWS][2][HALF];            // [row][column parity][column / 2]
    __shared__ f4 L[ROWS][COLS];               // linear layout for the shuffle form (a lane reads its own column)
    const int tid = threadIdx.x;
    for (int i = tid; i < ROWS * COLS; i += 512) {
        const int r = i / COLS, c = i - r * COLS;
        const f4 v = in[(size_t)blockIdx.x * ROWS * COLS + i];
        A[r][c & 1][c >> 1] = v; L[r][c] = v;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    f4 acc = { 0, 0, 0, 0 };
    for (int rep = 0; rep < REP; rep++)
        for (int r = wave; r < ROWS; r += 8) {
            if (!SHUFFLE) {
                if (lane < NQ) {
                    const f4* ev = &A[r][0][lane]; const f4* od = &A[r][1][lane];
                    acc += tapsum(ev[0], od[0], ev[1], od[1], ev[2]);
                }
            } else {
                // lanes 0..63 hold columns 0..63 (outputs q = 0..29 use columns <= 62; the last four outputs need the
                // second load of columns 64..71 -- kept out of the timing: 30 outputs per wave here, 34 in the LDS form)
                const f4 mine = L[r][lane];
                const int m0 = __float_as_int(mine.x), m1 = __float_as_int(mine.y), m2 = __float_as_int(mine.z), m3 = __float_as_int(mine.w);
                f4 t[5];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const int srcl = ((2 * lane + k) & 63) << 2;
                    t[k].x = __int_as_float(__builtin_amdgcn_ds_bpermute(srcl, m0));
                    t[k].y = __int_as_float(__builtin_amdgcn_ds_bpermute(srcl, m1));
                    t[k].z = __int_as_float(__builtin_amdgcn_ds_bpermute(srcl, m2));
                    t[k].w = __int_as_float(__builtin_amdgcn_ds_bpermute(srcl, m3));
                }
                if (lane < 30) acc += tapsum(t[0], t[1], t[2], t[3], t[4]);
            }
            asm volatile("" : "+v"(acc));
        }
    out[(size_t)blockIdx.x * 512 + tid] = acc;
}

int main()
{
    const int blocks = 768;                    // three per CU, like the level kernel
    std::vector<f4> h((size_t)blocks * ROWS * COLS);
    unsigned v = 1; for (auto& p : h) for (int c = 0; c < 4; c++) { v = v * 1664525u + 1013904223u; p[c] = (float)(v >> 20) * (1.f / 4096); }
    f4 *din, *d0, *d1;
    CK(hipMalloc((void**)&din, h.size() * 16)); CK(hipMalloc((void**)&d0, (size_t)blocks * 512 * 16)); CK(hipMalloc((void**)&d1, (size_t)blocks * 512 * 16));
    CK(hipMemcpy(din, h.data(), h.size() * 16, hipMemcpyHostToDevice));
    float ms[2];
    for (int which = 0; which < 2; which++) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int it = 0; it < 2; it++) {
            if (it) CK(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL(k_taps<false>, dim3(blocks), dim3(512), 0, 0, din, d0);
            else            hipLaunchKernelGGL(k_taps<true>, dim3(blocks), dim3(512), 0, 0, din, d1);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[which], e0, e1));
    }
    std::vector<f4> r0((size_t)blocks * 512), r1((size_t)blocks * 512);
    CK(hipMemcpy(r0.data(), d0, r0.size() * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), d1, r1.size() * 16, hipMemcpyDeviceToHost));
    long bad = 0;
    for (size_t i = 0; i < r0.size(); i++) if ((i & 63) < 30) for (int c = 0; c < 4; c++) {
        if (r0[i][c] != r1[i][c] && bad < 4) std::printf("  mismatch at thread %zu comp %d: %.9g vs %.9g\n", i, c, r0[i][c], r1[i][c]);
        bad += r0[i][c] != r1[i][c];
    }
    const double rows = (double)blocks * ROWS * REP;
    std::printf("pyrDown horizontal 5-tap sums, %d workgroups x %d rows x %d repeats, same sums both ways: %s\n", blocks, ROWS, REP, bad ? "NO" : "yes (first 30 outputs of a row)");
    std::printf("  LDS taps (5 x ds_read_b128 per output, 34 outputs per row): %8.1f us  = %.2f ns per row and CU-third\n", ms[0] * 1e3, ms[0] * 1e6 / rows * 768);
    std::printf("  shuffle  (1 x ds_read_b128 + 20 x ds_bpermute_b32, 30 outputs): %8.1f us  = %.2f ns per row and CU-third  -> %.2fx the LDS form\n", ms[1] * 1e3, ms[1] * 1e6 / rows * 768, ms[1] / ms[0]);
    return 0;
}
