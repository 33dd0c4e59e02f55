// upload_probe -- one-shot diagnostic for the round-1 `test_row_padded_frames` mismatch ("The strided-upload
// mismatch"): does a 2-D pitched H2D copy from PAGEABLE memory on a side stream, followed by hipStreamSynchronize of that
// stream, leave every byte in HBM where a kernel on ANOTHER non-blocking stream then reads it?  Two upload forms are
// compared on the same frames, each frame checked once (no retry of anything):
//   old: hipMemcpy2DAsync(dev, row, host, step, row, rows, H2D, copy_stream) + hipStreamSynchronize(copy_stream)   [round 1, 4a63aee]
//   new: hipMemcpy(dev, host, (rows-1)*step + row)                                                                 [since a1dfea5]
// The host buffer is malloc'ed and freed per frame (a numpy temporary), the device buffer is reused and was read by a
// kernel before (lines resident in the XCD L2s), the check kernel runs on a third stream right after the copy returns.
// Build: hipcc --offload-arch=gfx950 -O2 tools/cpp/upload_probe.hip -o upload_probe ; run: ./upload_probe [frames=24]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_row_sums(const uint8_t* __restrict__ p, int rows, int row_bytes, long pitch, unsigned long long* __restrict__ sums)
{
    const int r = blockIdx.x;
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < row_bytes; i += blockDim.x) s += (unsigned long long)p[r * pitch + i] * (unsigned)(i % 251 + 1);
    __shared__ unsigned long long red[256];
    red[threadIdx.x] = s; __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) sums[r] = red[0];
}

int main(int argc, char** argv)
{
    const int frames = argc > 1 ? std::atoi(argv[1]) : 24;
    const int rows = 480, cols = 640, row = cols * 3, step = 2100, xoff = 90;
    hipStream_t copy_s, run_s;
    CK(hipStreamCreateWithFlags(&copy_s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&run_s, hipStreamNonBlocking));
    uint8_t *dev_old = nullptr, *dev_new = nullptr;
    unsigned long long* dsum = nullptr;
    CK(hipMalloc((void**)&dev_old, (size_t)rows * row + 64));
    CK(hipMalloc((void**)&dev_new, (size_t)rows * step + 64));
    CK(hipMalloc((void**)&dsum, rows * sizeof(unsigned long long)));
    std::vector<unsigned long long> got(rows), want(rows);
    int bad_old = 0, bad_new = 0, bad_rows_old = 0, bad_rows_new = 0;
    for (int k = 0; k < frames; k++) {
        uint8_t* host = (uint8_t*)std::malloc((size_t)rows * step);          // pageable, fresh per frame
        unsigned v = 12345u + 977u * k;
        for (size_t i = 0; i < (size_t)rows * step; i++) { v = v * 1664525u + 1013904223u; host[i] = (uint8_t)(v >> 24); }
        for (int r = 0; r < rows; r++) {
            unsigned long long s = 0;
            for (int i = 0; i < row; i++) s += (unsigned long long)host[(size_t)r * step + xoff + i] * (unsigned)(i % 251 + 1);
            want[r] = s;
        }
        // old form
        CK(hipMemcpy2DAsync(dev_old, row, host + xoff, step, row, rows, hipMemcpyHostToDevice, copy_s));
        CK(hipStreamSynchronize(copy_s));
        hipLaunchKernelGGL(k_row_sums, dim3(rows), dim3(256), 0, run_s, dev_old, rows, row, (long)row, dsum);
        CK(hipMemcpyAsync(got.data(), dsum, rows * 8, hipMemcpyDeviceToHost, run_s));
        CK(hipStreamSynchronize(run_s));
        int br = 0; for (int r = 0; r < rows; r++) br += got[r] != want[r];
        if (br) { bad_old++; bad_rows_old += br; std::printf("frame %d old form: %d rows differ (first row %d)\n", k, br, (int)(std::find_if(got.begin(), got.end(), [&](unsigned long long& g) { return g != want[&g - got.data()]; }) - got.begin())); }
        // new form
        CK(hipMemcpy(dev_new, host + xoff, (size_t)(rows - 1) * step + row, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_row_sums, dim3(rows), dim3(256), 0, run_s, dev_new, rows, row, (long)step, dsum);
        CK(hipMemcpyAsync(got.data(), dsum, rows * 8, hipMemcpyDeviceToHost, run_s));
        CK(hipStreamSynchronize(run_s));
        br = 0; for (int r = 0; r < rows; r++) br += got[r] != want[r];
        if (br) { bad_new++; bad_rows_new += br; std::printf("frame %d new form: %d rows differ\n", k, br); }
        std::free(host);
    }
    std::printf("upload_probe: %d frames; old form (2-D async pageable + stream sync): %d frames / %d rows wrong; new form (blocking linear): %d frames / %d rows wrong\n",
                frames, bad_old, bad_rows_old, bad_new, bad_rows_new);
    return 0;
}
