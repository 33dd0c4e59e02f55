// jpeg_device.hpp -- device back end of the file driver's JPEG decode (jpeg_device.hip)
#pragma once
#include <atomic>
#include "jpeg_decode.hpp"

namespace pf {

// One per consumer (a map, or the process-wide one behind pf_jpeg_decode_device).  decode_to() entropy-decodes on the calling thread
// into a pinned buffer (two, so the next frame's Huffman pass overlaps this frame's copy) and queues the upload and the two kernels on
// `stream` (a hipStream_t); the BGR8 frame, rows * cols * 3 bytes packed, is complete in `dev_bgr` in stream order.  Not thread-safe.
class JpegDevice {
public:
    enum { kSlots = 16 };
    struct Bytes { size_t coefficients, planes, frame; };
    JpegDevice() {}
    ~JpegDevice();
    JpegDevice(const JpegDevice&) = delete;
    JpegDevice& operator=(const JpegDevice&) = delete;
    bool decode_to(const uint8_t* data, size_t len, uint8_t* dev_bgr, int rows, int cols, void* stream);
    // A batch: the Huffman passes of n <= kSlots frames side by side on `threads` host threads (0: one each); ok[i] says whether frame i is
    // staged (its message otherwise comes out of submit).  Then submit(i, ...) per frame, in the order wanted.  rows = 0: any size.
    // one frame, alternating between two staging buffers: returns the buffer's index for submit()
    int  stage_one(const uint8_t* data, size_t len, unsigned char* ok);
    bool stage_batch(int n, const uint8_t* const* data, const size_t* len, int rows, int cols, int threads, unsigned char* ok);
    bool submit(int i, uint8_t* dev_bgr, void* stream);
    void staged_size(int i, int* rows, int* cols) const { *rows = slot_[i].f.rows; *cols = slot_[i].f.cols; }
    Bytes last_bytes() const { return last_; }          // of the most recent frame: what the two kernels read and wrote
    // frames whose Huffman pass ran on the GPU / fell back to the host after trying, and the rounds the most recent one took
    void  huffman_counts(long* on_device, long* fell_back, int* rounds) const { *on_device = par_frames_; *fell_back = fallback_frames_; *rounds = last_rounds_; }
private:
    struct Slot {
        void* host = nullptr; size_t cap = 0; void* done = nullptr; bool used = false, staged = false; JpegFrame f; std::string err;
        const uint8_t* data = nullptr; size_t len = 0;        // the stream (the caller's, valid until submit): a fallback decodes it again
        bool par = false; size_t par_bytes = 0, aux_words = 0;   // staged for the Huffman pass on the GPU: the scan's bytes (stuffing removed), restart tables
    };
    enum { kMaxRounds = 64 };     // launches of the round kernel (512 sweeps): the flags array and the launch limit; a stream that has not settled by then goes to the host's serial pass
    bool huffman_on_device(int i, void* stream, size_t coef_bytes);
    bool prepare(int i, const uint8_t* data, size_t len, int rows, int cols);
    bool entropy(int i, const uint8_t* data, size_t len);
    Slot   slot_[kSlots];
    int    next_ = 0;
    void*  dev_ = nullptr;    size_t dev_cap_ = 0;
    void*  planes_ = nullptr; size_t planes_cap_ = 0;
    Bytes  last_ = { 0, 0, 0 };
    void*  huff_ = nullptr;   size_t huff_cap_ = 0;          // plan, scan bytes, subsequence states and counts of the parallel Huffman pass
    void*  res_host_ = nullptr;
    int    last_rounds_ = 0, settle_hint_ = 4; long par_frames_ = 0, fallback_frames_ = 0;
    std::atomic<int> skip_par_{ 0 };      // entropy() runs on several host threads in a batch
};

}  // namespace pf
