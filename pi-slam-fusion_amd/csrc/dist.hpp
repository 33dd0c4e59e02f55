// dist.hpp -- seam exchange of the tile-sharded mosaic (dist.cpp)
#pragma once
#include "fusion_map.hpp"
#include "dist_plan.hpp"

namespace pf {

struct Transport {
    int rank = 0, nranks = 1;
    virtual ~Transport() {}
    virtual const char* name() const = 0;
    // all-to-all-v between ranks, buffers indexed by peer (own entry ignored); returns with the bytes landed
    virtual bool exchange_dev(const std::vector<const void*>& send, const std::vector<size_t>& send_bytes, const std::vector<void*>& recv,
                              const std::vector<size_t>& recv_bytes, hipStream_t st) = 0;      // device buffers
    virtual bool exchange_host(const std::vector<const void*>& send, const std::vector<size_t>& send_bytes, const std::vector<void*>& recv,
                               const std::vector<size_t>& recv_bytes, hipStream_t st) = 0;     // host buffers (control messages)
};

bool rccl_unique_id(void* out128);
Transport* make_rccl_transport(const void* id128, int rank, int nranks, int device);
Transport* make_host_transport(int rank, int nranks, pf_exchange_fn fn, void* user);

// BlendPlan, plan_blend: dist_plan.hpp

class DistMap {
public:
    DistMap(FusionMap* m, Transport* t);      // takes the transport
    ~DistMap();
    int  blend_changed(int* xy, uint8_t* bgr, int cap);        // tiles blended on this rank, -1 on failure
    bool save_to_memory(uint8_t* bgr, int* rows, int* cols, int* tx0, int* ty0);
    bool save(const char* filename);
    // 1 rendered / accepted, 0 rejected, -1 failure.  `produce` (root only, img->data == nullptr): the root's pixels are written into its
    // staged slot by the caller's own work on the map's stream (pf_dist_feed_jpeg: the decoder) instead of being uploaded
    int  feed(const pf_image* img, const double pose7[7], int root, const FusionMap::FrameProducer* produce = nullptr);
    int  rank() const;
    const pf_dist_stats& stats() const { return stats_; }
    const Transport* transport() const { return t_; }
    void set_verify(bool on) { verify_ = on; }
private:
    bool gather_lists(std::vector<std::vector<FusionMap::TileRec>>& all, std::vector<long long>& caps, long long my_cap);
    bool agree(bool ok_here);
    bool exchange_checked(const std::vector<const void*>& send, const std::vector<size_t>& sb, const std::vector<void*>& recv,
                          const std::vector<size_t>& rb, const char* what);
    FusionMap* m_;
    Transport* t_;
    DevBuf send_, recv_;
    pf_dist_stats stats_{};
    bool verify_ = false;                       // PF_DIST_VERIFY=1 or pf_dist_set_verify: hash every data exchange on both ends
    bool verify_now_ = false;                   // what the ranks agreed on for the next data exchange (any rank asking is enough)
};

}  // namespace pf
