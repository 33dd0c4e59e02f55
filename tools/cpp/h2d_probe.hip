// h2d_probe -- what can a pinned staging ring inside pf_feed gain over the one blocking pageable hipMemcpy it does today (VERDICT r04
// item 7)?  One 4000x3000 BGR keyframe = 36 MB per feed; the caller's buffer is pageable (a cv::Mat of the tracker), so
// a ring has to COPY the rows into a pinned slot first.  Measured per form, frames per second and GB/s over `frames` keyframes, each form
// while a kernel of ~100 us per frame runs on another stream (the keyframe launch the upload has to overlap):
//   A  hipMemcpy(dev, pageable)                                   blocking, what pf_feed does (the runtime stages through its own pinned buffers)
//   B  hipMemcpyAsync(dev, pinned) from a ring of 4 slots         the link's ceiling: no host-side copy at all (a caller that already owns pinned memory)
//   C  memcpy(pinned slot, pageable) by 1 thread + hipMemcpyAsync the ring as VERDICT describes it
//   D  the same with the memcpy split over T threads (T = 2, 4, 8)
//   E  hipHostRegister(pageable) + hipMemcpyAsync + hipHostUnregister per frame
// Build: hipcc --offload-arch=gfx950 -O2 tools/cpp/h2d_probe.hip -o h2d_probe -lpthread ; run: ./h2d_probe [frames=40]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_busy(float* p, int iters)
{
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; i++) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(uint8_t* dst, const uint8_t* src, size_t n, int threads)
{
    if (threads <= 1) { std::memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t chunk = (n / threads + 4095) & ~(size_t)4095;
    for (int t = 0; t < threads; t++) {
        const size_t lo = std::min(n, chunk * t), hi = std::min(n, chunk * (t + 1));
        if (hi > lo) th.emplace_back([=] { std::memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto& t : th) t.join();
}

int main(int argc, char** argv)
{
    const int frames = argc > 1 ? std::atoi(argv[1]) : 40;
    const size_t bytes = (size_t)4000 * 3000 * 3;
    constexpr int kRing = 4;
    hipStream_t copy_s, run_s;
    CK(hipStreamCreateWithFlags(&copy_s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&run_s, hipStreamNonBlocking));
    uint8_t* dev[kRing]; uint8_t* pin[kRing]; hipEvent_t done[kRing];
    for (int i = 0; i < kRing; i++) { CK(hipMalloc((void**)&dev[i], bytes)); CK(hipHostMalloc((void**)&pin[i], bytes, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming)); std::memset(pin[i], i + 1, bytes); }
    float* busy; CK(hipMalloc((void**)&busy, 4096)); CK(hipMemset(busy, 0, 4096));
    // two pageable source frames, touched (a tracker's frames are resident)
    std::vector<uint8_t*> page(2);
    for (auto& p : page) { p = (uint8_t*)std::malloc(bytes); std::memset(p, 7, bytes); }
    auto kernel = [&]() { hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, run_s, busy, 12000); };
    auto report = [&](const char* name, double dt) { std::printf("%-58s %8.1f keyframes/s  %6.2f GB/s\n", name, frames / dt, bytes * frames / dt / 1e9); std::fflush(stdout); };

    for (int warm = 0; warm < 3; warm++) CK(hipMemcpy(dev[0], page[0], bytes, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    {   // A
        const double t0 = now();
        for (int f = 0; f < frames; f++) { CK(hipMemcpy(dev[f % kRing], page[f & 1], bytes, hipMemcpyHostToDevice)); kernel(); }
        CK(hipDeviceSynchronize());
        report("A blocking hipMemcpy from pageable memory (pf_feed today)", now() - t0);
    }
    {   // B
        const double t0 = now();
        for (int f = 0; f < frames; f++) {
            const int s = f % kRing;
            CK(hipMemcpyAsync(dev[s], pin[s], bytes, hipMemcpyHostToDevice, copy_s));
            CK(hipEventRecord(done[s], copy_s)); CK(hipStreamWaitEvent(run_s, done[s], 0)); kernel();
        }
        CK(hipDeviceSynchronize());
        report("B hipMemcpyAsync from pinned memory, no host copy (ceiling)", now() - t0);
    }
    for (int T : { 1, 2, 4, 8 }) {   // C / D
        for (int i = 0; i < kRing; i++) CK(hipEventRecord(done[i], copy_s));
        const double t0 = now();
        for (int f = 0; f < frames; f++) {
            const int s = f % kRing;
            CK(hipEventSynchronize(done[s]));                     // the slot's previous copy has left it
            par_copy(pin[s], page[f & 1], bytes, T);
            CK(hipMemcpyAsync(dev[s], pin[s], bytes, hipMemcpyHostToDevice, copy_s));
            CK(hipEventRecord(done[s], copy_s)); CK(hipStreamWaitEvent(run_s, done[s], 0)); kernel();
        }
        CK(hipDeviceSynchronize());
        char name[96]; std::snprintf(name, sizeof name, "%s memcpy into a pinned ring slot by %d thread%s + async H2D", T == 1 ? "C" : "D", T, T == 1 ? "" : "s");
        report(name, now() - t0);
    }
    {   // E
        const double t0 = now();
        for (int f = 0; f < frames; f++) {
            uint8_t* p = page[f & 1];
            CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
            CK(hipMemcpyAsync(dev[f % kRing], p, bytes, hipMemcpyHostToDevice, copy_s));
            CK(hipStreamSynchronize(copy_s));
            CK(hipHostUnregister(p));
            kernel();
        }
        CK(hipDeviceSynchronize());
        report("E hipHostRegister + async H2D + unregister per keyframe", now() - t0);
    }
    return 0;
}
