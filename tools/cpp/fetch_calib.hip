// fetch_calib.hip -- what FETCH_SIZE / WRITE_SIZE report for the access shapes of the level kernel.
// MI355X_MICROARCH.md documents the factor 2 for 16 B/lane coalesced streaming reads only and says "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below touches a KNOWN number of
// distinct bytes of a buffer that no cache holds (8 frames of 36 MB, walked round robin; L2 is 32 MB in total); run under
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- tools/cpp/fetch_calib.bin      (and again with WRITE_SIZE)
// and divide (tools/fetch_calib_summary.py).  Diagnostic only; not on the product path.
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/cpp/fetch_calib.bin tools/cpp/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

constexpr int COLS = 4000, ROWS = 3000, STEP = COLS * 3;
constexpr size_t FRAME = (size_t)STEP * ROWS;             // 36 MB

typedef uint32_t u2 __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u4a __attribute__((ext_vector_type(4)));
typedef uint32_t u2a __attribute__((ext_vector_type(2)));
typedef float f3 __attribute__((ext_vector_type(3)));

// coalesced streaming reads, W bytes per lane
template <int W> __global__ __launch_bounds__(256) void k_stream_rd(const uint8_t* __restrict__ src, size_t bytes, uint32_t* __restrict__ sink)
{
    uint32_t acc = 0;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * W; i + W <= bytes; i += (size_t)gridDim.x * 256 * W) {
        if constexpr (W == 16) { const u4a v = *(const u4a*)(src + i); acc += v.x ^ v.y ^ v.z ^ v.w; }
        else if constexpr (W == 8) { const u2a v = *(const u2a*)(src + i); acc += v.x ^ v.y; }
        else acc += *(const uint32_t*)(src + i);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// stage A's gather: a wave = 64 neighbouring pixels of a source row (3 bytes apart), each lane one UNALIGNED 8-byte load from the row and
// one from the row below; a workgroup walks its 64-column strip down the frame.  Every byte of the frame is covered (once per row as row 0
// of a pixel, once as row 1): distinct bytes = FRAME, requested bytes = 2 x 8 x pixels.
__global__ __launch_bounds__(256) void k_gather8(const uint8_t* __restrict__ src, uint32_t* __restrict__ sink)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    uint32_t acc = 0;
    if (x < COLS - 2)
        for (int y = blockIdx.y * 100 + wv; y < blockIdx.y * 100 + 100 && y < ROWS - 1; y += 4) {
            const u2 a = *(const u2*)(src + (size_t)y * STEP + 3 * x), b = *(const u2*)(src + (size_t)(y + 1) * STEP + 3 * x);
            acc += a.x ^ a.y ^ b.x ^ b.y;
        }
    if (acc == 0x12345678u) sink[0] = acc;
}

// the same gather in the block kernel's order: 71 x 39 pixel blocks for 64 x 32 kept (every source row is asked for by two block rows)
__global__ __launch_bounds__(512) void k_gather8_blocks(const uint8_t* __restrict__ src, uint32_t* __restrict__ sink)
{
    const int r0 = threadIdx.x / 71, c = threadIdx.x % 71;
    const int x = blockIdx.x * 64 - 4 + c;
    uint32_t acc = 0;
    if (r0 < 7 && x >= 0 && x < COLS - 2)
        for (int r = r0; r < 39; r += 7) {
            const int y = blockIdx.y * 32 - 4 + r;
            if (y < 0 || y >= ROWS - 1) continue;
            const u2 a = *(const u2*)(src + (size_t)y * STEP + 3 * x), b = *(const u2*)(src + (size_t)(y + 1) * STEP + 3 * x);
            acc += a.x ^ a.y ^ b.x ^ b.y;
        }
    if (acc == 0x12345678u) sink[0] = acc;
}

// stage D's stores: 12-byte pixels (3 floats) of a 256-px tile row per wave; KEEP of every 8 lanes store (the max-weight select lets a
// part of the pixels through), W_EACH: the 4-byte weight store that goes with each pixel to a second plane
template <int KEEP> __global__ __launch_bounds__(256) void k_store12(float* __restrict__ lap, float* __restrict__ w, size_t npx)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (size_t)gridDim.x * 256) {
        if ((int)(i & 7) >= KEEP) continue;
        f3 v = { (float)i, 1.f, 2.f };
        *(f3*)(lap + 3 * i) = v;
        w[i] = 0.5f;
    }
}
// streaming 16 B/lane stores (the documented exact case) for reference
__global__ __launch_bounds__(256) void k_stream_wr(u4a* __restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = u4a{ (uint32_t)i, 1, 2, 3 };
}

int main()
{
    constexpr int NF = 8, REP = 8;
    uint8_t* buf; uint32_t* sink; float *lap, *w;
    CK(hipMalloc((void**)&buf, FRAME * NF + 64)); CK(hipMalloc((void**)&sink, 64));
    const size_t npx = 12u << 20;                                   // 12 M pixels: 151 MB of 12-byte payload + 50 MB of weights per pass
    CK(hipMalloc((void**)&lap, npx * 12 * 2)); CK(hipMalloc((void**)&w, npx * 4 * 2));
    CK(hipMemset(buf, 7, FRAME * NF + 64));
    CK(hipDeviceSynchronize());
    for (int r = 0; r < REP; r++) {
        const uint8_t* f = buf + (size_t)(r % NF) * FRAME;
        hipLaunchKernelGGL(k_stream_rd<16>, dim3(2048), dim3(256), 0, 0, f, FRAME, sink);
        hipLaunchKernelGGL(k_stream_rd<8>, dim3(2048), dim3(256), 0, 0, f, FRAME, sink);
        hipLaunchKernelGGL(k_stream_rd<4>, dim3(2048), dim3(256), 0, 0, f, FRAME, sink);
        hipLaunchKernelGGL(k_gather8, dim3((COLS + 63) / 64, ROWS / 100), dim3(256), 0, 0, f, sink);
        hipLaunchKernelGGL(k_gather8_blocks, dim3((COLS + 63) / 64, (ROWS + 31) / 32), dim3(512), 0, 0, f, sink);
        float* lp = lap + (size_t)(r & 1) * npx * 3; float* wp = w + (size_t)(r & 1) * npx;
        hipLaunchKernelGGL(k_store12<8>, dim3(4096), dim3(256), 0, 0, lp, wp, npx);
        hipLaunchKernelGGL(k_store12<4>, dim3(4096), dim3(256), 0, 0, lp, wp, npx);
        hipLaunchKernelGGL(k_store12<2>, dim3(4096), dim3(256), 0, 0, lp, wp, npx);
        hipLaunchKernelGGL(k_stream_wr, dim3(4096), dim3(256), 0, 0, (u4a*)lp, npx * 12 / 16);
    }
    CK(hipDeviceSynchronize());
    std::printf("known bytes per dispatch: stream_rd<16|8|4> %zu read; gather8 / gather8_blocks %zu distinct (requested %zu / %zu); "
                "store12<8|4|2> %zu | %zu | %zu payload+weight bytes; stream_wr %zu\n",
                FRAME, FRAME, (size_t)16 * (COLS - 2) * (ROWS - 1), (size_t)16 * 71 * 39 * ((COLS + 63) / 64) * ((ROWS + 31) / 32),
                npx * 16, npx * 16 / 2, npx * 16 / 4, npx * 12 / 16 * 16);
    return 0;
}
