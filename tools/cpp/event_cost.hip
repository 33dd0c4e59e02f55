// event_cost.hip -- what do the cross-stream dependencies of a two-stream keyframe pipeline cost?  (r03 probe)
// Stream A runs a long kernel per "keyframe" (level 0), stream B a short one (upper levels).  Dependencies as the pipeline
// would need them: B(k) after A(k-1); A(k) after B(k-2).  Compared with everything on one stream.
//   hipcc --offload-arch=gfx950 -O2 -o event_cost.bin event_cost.hip && ./event_cost.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(long long cycles, int* sink)
{
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    const int N = 400;
    std::vector<hipEvent_t> ea(N), eb(N);
    for (int i = 0; i < N; i++) { hipEventCreateWithFlags(&ea[i], hipEventDisableTiming); hipEventCreateWithFlags(&eb[i], hipEventDisableTiming); }
    int* sink; hipMalloc(&sink, 4);
    // the grid: 768 long workgroups (one round of the chip) for A, 768 short ones for B
    const long long ca = 100 * 100 * 30, cb = 100 * 100 * 5;      // s_memtime runs at 100 MHz: ~300 us / ~50 us at 1 WG per slot? calibrate below
    auto time_one = [&](long long c) { hipLaunchKernelGGL(spin, dim3(768), dim3(512), 0, a, c, sink); hipStreamSynchronize(a);
        const double t0 = now(); for (int i = 0; i < 20; i++) hipLaunchKernelGGL(spin, dim3(768), dim3(512), 0, a, c, sink); hipStreamSynchronize(a); return (now() - t0) / 20 * 1e6; };
    const double ta = time_one(ca), tb = time_one(cb);
    printf("kernel A alone %.1f us, kernel B alone %.1f us per launch (back to back on one stream)\n", ta, tb);
    // 1. one stream: A B A B ...
    double t0 = now();
    for (int k = 0; k < N; k++) { hipLaunchKernelGGL(spin, dim3(768), dim3(512), 0, a, ca, sink); hipLaunchKernelGGL(spin, dim3(768), dim3(512), 0, a, cb, sink); }
    hipStreamSynchronize(a);
    const double one = (now() - t0) / N * 1e6;
    // 2. two streams without dependencies (upper bound of the overlap)
    t0 = now();
    for (int k = 0; k < N; k++) { hipLaunchKernelGGL(spin, dim3(768), dim3(512), 0, a, ca, sink); hipLaunchKernelGGL(spin, dim3(768), dim3(512), 0, b, cb, sink); }
    hipStreamSynchronize(a); hipStreamSynchronize(b);
    const double two_free = (now() - t0) / N * 1e6;
    // 3. two streams with the pipeline's dependencies
    t0 = now();
    for (int k = 0; k < N; k++) {
        if (k >= 2) hipStreamWaitEvent(a, eb[k - 2], 0);
        hipLaunchKernelGGL(spin, dim3(768), dim3(512), 0, a, ca, sink);
        hipEventRecord(ea[k], a);
        if (k >= 1) hipStreamWaitEvent(b, ea[k - 1], 0);
        hipLaunchKernelGGL(spin, dim3(768), dim3(512), 0, b, cb, sink);
        hipEventRecord(eb[k], b);
    }
    hipStreamSynchronize(a); hipStreamSynchronize(b);
    const double two_dep = (now() - t0) / N * 1e6;
    // 4. one stream with an event record after every launch (cost of the records alone)
    t0 = now();
    for (int k = 0; k < N; k++) { hipLaunchKernelGGL(spin, dim3(768), dim3(512), 0, a, ca, sink); hipEventRecord(ea[k], a); hipLaunchKernelGGL(spin, dim3(768), dim3(512), 0, a, cb, sink); hipEventRecord(eb[k], a); }
    hipStreamSynchronize(a);
    const double one_ev = (now() - t0) / N * 1e6;
    printf("per keyframe: one stream %.1f us | one stream + 2 event records %.1f us | two streams, no dependencies %.1f us | two streams, pipeline dependencies %.1f us\n",
           one, one_ev, two_free, two_dep);
    return 0;
}
