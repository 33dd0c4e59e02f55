// dist.cpp -- the seam exchange of the tile-sharded mosaic inside the library: what a reference-side caller
// (the Map2DHIP subclass of INTEGRATION.md) needs to run draw() and save() across the GPUs of a node.
//
// feed() has no collective: every rank is fed every keyframe and renders the tiles it owns.  Two steps exchange data:
//   * draw()  -- Ele::blend's 3x3 neighbour gather (Map2DFusion/MultiBandMap2DCPU.cpp:724-741, :93-117): a changed tile
//               whose neighbours live on other ranks gets their edge strips (border 1<<(L-i) pixels at level i).  Per call:
//               ONE pack launch for all strips this rank provides, ONE grouped point-to-point exchange (ncclSend/ncclRecv
//               to the <= 8 neighbour owners: every xGMI link at once, no ring), ONE batched blend.
//   * save()  -- the per-level paste of every tile into the mosaic (.cpp:806-836): whole tiles travel once to rank 0, which
//               runs the single whole-mosaic collapse over its own and the gathered tiles.
// Both sides derive the exchange plan from the same all-gathered tile lists, so no header travels with the payload.
//
// Transport: RCCL (dlopen'ed: librccl is not a link-time dependency) or a caller-supplied host-buffer exchange hook
// (tests, gloo rehearsals on one GPU, MPI in a reference-side launcher).
#include "dist.hpp"
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

namespace pf {

#define HIP_OK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                  \
            return false;                                                                  \
        }                                                                                  \
    } while (0)

// ------------------------------------------------------------------ RCCL, resolved at run time
struct Id128 { char b[128]; };      // ncclUniqueId
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128 /* ncclUniqueId by value */, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
}  // namespace
static Rccl g_rccl;

static bool load_rccl()
{
    if (g_rccl.lib) return true;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { set_error("RCCL not found (librccl.so.1)"); return false; }
#define SYM(field, name) *(void**)&g_rccl.field = dlsym(h, name); if (!g_rccl.field) { set_error("RCCL symbol missing: " name); return false; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.lib = h;
    return true;
}

bool rccl_unique_id(void* out128)
{
    if (!load_rccl()) return false;
    const int rc = g_rccl.GetUniqueId(out128);
    if (rc) { set_error(std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(rc)); return false; }
    return true;
}

class RcclTransport : public Transport {
public:
    bool init(const void* id128, int r, int n, int device)
    {
        rank = r; nranks = n; device_ = device;
        if (!load_rccl()) return false;
        Id128 id; std::memcpy(id.b, id128, 128);
        HIP_OK(hipSetDevice(device));
        const int rc = g_rccl.CommInitRank(&comm_, n, id, r);
        if (rc) { set_error(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc)); return false; }
        return true;
    }
    ~RcclTransport() override { if (comm_) g_rccl.CommDestroy(comm_); if (stage_s_) (void)hipFree(stage_s_); if (stage_r_) (void)hipFree(stage_r_); }
    const char* name() const override { return "rccl"; }

    // grouped point-to-point: every pair with bytes to move gets one send and one recv, issued together
    bool exchange_dev(const std::vector<const void*>& send, const std::vector<size_t>& sb, const std::vector<void*>& recv,
                      const std::vector<size_t>& rb, hipStream_t st) override
    {
        int rc = g_rccl.GroupStart();
        for (int p = 0; p < nranks && !rc; p++) {
            if (p == rank) continue;
            if (sb[p]) rc = g_rccl.Send(send[p], sb[p], /* ncclUint8 */ 1, p, comm_, st);
            if (!rc && rb[p]) rc = g_rccl.Recv(recv[p], rb[p], 1, p, comm_, st);
        }
        const int rc2 = g_rccl.GroupEnd();
        if (rc || rc2) { set_error(std::string("ncclSend/Recv: ") + g_rccl.GetErrorString(rc ? rc : rc2)); return false; }
        HIP_OK(hipStreamSynchronize(st));
        return true;
    }
    // small control messages travel the same way through device staging
    bool exchange_host(const std::vector<const void*>& send, const std::vector<size_t>& sb, const std::vector<void*>& recv,
                       const std::vector<size_t>& rb, hipStream_t st) override
    {
        size_t ts = 0, tr = 0;
        for (int p = 0; p < nranks; p++) if (p != rank) { ts += sb[p]; tr += rb[p]; }
        if (!grow(stage_s_, cap_s_, ts) || !grow(stage_r_, cap_r_, tr)) return false;
        std::vector<const void*> ds(nranks, nullptr); std::vector<void*> dr(nranks, nullptr);
        size_t os = 0, orr = 0;
        for (int p = 0; p < nranks; p++) {
            if (p == rank) continue;
            ds[p] = stage_s_ + os; dr[p] = stage_r_ + orr;
            if (sb[p]) HIP_OK(hipMemcpyAsync(stage_s_ + os, send[p], sb[p], hipMemcpyHostToDevice, st));
            os += sb[p]; orr += rb[p];
        }
        HIP_OK(hipStreamSynchronize(st));
        if (!exchange_dev(ds, sb, dr, rb, st)) return false;
        orr = 0;
        for (int p = 0; p < nranks; p++) {
            if (p == rank) continue;
            if (rb[p]) HIP_OK(hipMemcpy(recv[p], stage_r_ + orr, rb[p], hipMemcpyDeviceToHost));
            orr += rb[p];
        }
        return true;
    }
private:
    static bool grow(char*& p, size_t& cap, size_t need)
    {
        if (need <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc((void**)&p, need + 4096) != hipSuccess) { set_error("dist: hipMalloc failed"); return false; }
        cap = need + 4096;
        return true;
    }
    void* comm_ = nullptr;
    int   device_ = 0;
    char *stage_s_ = nullptr, *stage_r_ = nullptr;
    size_t cap_s_ = 0, cap_r_ = 0;
};

// host-buffer hook: the caller moves bytes between ranks (gloo, MPI, an in-process rendezvous of threads ...)
class HostTransport : public Transport {
public:
    HostTransport(int r, int n, pf_exchange_fn fn, void* user) : fn_(fn), user_(user) { rank = r; nranks = n; }
    ~HostTransport() override { if (hs_) (void)hipHostFree(hs_); if (hr_) (void)hipHostFree(hr_); }
    const char* name() const override { return "host"; }
    bool exchange_host(const std::vector<const void*>& send, const std::vector<size_t>& sb, const std::vector<void*>& recv,
                       const std::vector<size_t>& rb, hipStream_t) override
    {
        if (!fn_(user_, send.data(), sb.data(), recv.data(), rb.data(), nranks)) { set_error("dist: the exchange hook failed"); return false; }
        return true;
    }
    bool exchange_dev(const std::vector<const void*>& send, const std::vector<size_t>& sb, const std::vector<void*>& recv,
                      const std::vector<size_t>& rb, hipStream_t st) override
    {
        size_t ts = 0, tr = 0;
        for (int p = 0; p < nranks; p++) if (p != rank) { ts += sb[p]; tr += rb[p]; }
        if (!grow(hs_, cs_, ts) || !grow(hr_, cr_, tr)) return false;
        std::vector<const void*> s(nranks, nullptr); std::vector<void*> r(nranks, nullptr);
        size_t os = 0, orr = 0;
        for (int p = 0; p < nranks; p++) {
            if (p == rank) continue;
            s[p] = hs_ + os; r[p] = hr_ + orr;
            if (sb[p]) HIP_OK(hipMemcpyAsync(hs_ + os, send[p], sb[p], hipMemcpyDeviceToHost, st));
            os += sb[p]; orr += rb[p];
        }
        HIP_OK(hipStreamSynchronize(st));
        if (!exchange_host(s, sb, r, rb, st)) return false;
        orr = 0;
        for (int p = 0; p < nranks; p++) {
            if (p == rank) continue;
            if (rb[p]) HIP_OK(hipMemcpyAsync(recv[p], hr_ + orr, rb[p], hipMemcpyHostToDevice, st));
            orr += rb[p];
        }
        HIP_OK(hipStreamSynchronize(st));
        return true;
    }
private:
    static bool grow(char*& p, size_t& cap, size_t need)
    {
        if (need <= cap) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        if (hipHostMalloc((void**)&p, need + 4096, hipHostMallocDefault) != hipSuccess) { set_error("dist: hipHostMalloc failed"); return false; }
        cap = need + 4096;
        return true;
    }
    pf_exchange_fn fn_; void* user_;
    char *hs_ = nullptr, *hr_ = nullptr; size_t cs_ = 0, cr_ = 0;
};

Transport* make_rccl_transport(const void* id128, int rank, int nranks, int device)
{
    RcclTransport* t = new RcclTransport();
    if (!t->init(id128, rank, nranks, device)) { delete t; return nullptr; }
    return t;
}
Transport* make_host_transport(int rank, int nranks, pf_exchange_fn fn, void* user) { return fn ? new HostTransport(rank, nranks, fn, user) : nullptr; }

// ------------------------------------------------------------------ DistMap
DistMap::DistMap(FusionMap* m, Transport* t) : m_(m), t_(t) { verify_ = std::getenv("PF_DIST_VERIFY") != nullptr; }
DistMap::~DistMap() { delete t_; send_.release(); recv_.release(); }

// every rank's tile list (coordinates + Ischanged) on every rank: sizes first, then the records
// (and every rank's cap on the tiles it will blend in this call: the providers must plan with the requester's cap)
bool DistMap::gather_lists(std::vector<std::vector<FusionMap::TileRec>>& all, std::vector<long long>& caps, long long my_cap)
{
    const int n = t_->nranks, me = t_->rank;
    all.assign(n, {});
    caps.assign(n, my_cap);
    m_->list_tiles(all[me]);
    if (n == 1) return true;
    hipStream_t st = m_->stream();
    struct Head { long long count, cap; } mine{ (long long)all[me].size(), my_cap };
    std::vector<Head> heads(n);
    std::vector<const void*> s(n, &mine); std::vector<void*> r(n, nullptr);
    std::vector<size_t> sb(n, sizeof(Head)), rb(n, sizeof(Head));
    for (int p = 0; p < n; p++) r[p] = &heads[p];
    sb[me] = rb[me] = 0;
    if (!t_->exchange_host(s, sb, r, rb, st)) return false;
    for (int p = 0; p < n; p++) {
        if (p == me) continue;
        caps[p] = heads[p].cap;
        all[p].resize((size_t)heads[p].count);
        s[p] = all[me].data(); sb[p] = all[me].size() * sizeof(FusionMap::TileRec);
        r[p] = all[p].data();  rb[p] = all[p].size() * sizeof(FusionMap::TileRec);
    }
    return t_->exchange_host(s, sb, r, rb, st);
}

// Every rank reports whether its side of the step so far went well; true iff all did.  A rank that failed locally (an
// allocation, a tile it does not hold) must not leave its peers waiting in the data exchange: everybody returns together.
bool DistMap::agree(bool ok_here)
{
    const int n = t_->nranks, me = t_->rank;
    verify_now_ = verify_;
    if (n == 1) return ok_here;
    // bit 0: this rank's side went well; bit 1: this rank wants the next data exchange hashed (PF_DIST_VERIFY / pf_dist_set_verify).
    // The verify wish travels with the status so that ranks started with different settings still run the SAME sequence of
    // collectives: if anyone asks, everyone hashes.
    int mine = (ok_here ? 1 : 0) | (verify_ ? 2 : 0);
    std::vector<int> theirs(n, 1);
    std::vector<const void*> s(n, &mine); std::vector<void*> r(n, nullptr);
    std::vector<size_t> sb(n, sizeof(int)), rb(n, sizeof(int));
    for (int p = 0; p < n; p++) r[p] = &theirs[p];
    sb[me] = rb[me] = 0;
    if (!t_->exchange_host(s, sb, r, rb, m_->stream())) return false;
    bool all = ok_here;
    for (int p = 0; p < n; p++) {
        if (p == me) continue;
        if (!(theirs[p] & 1)) { all = false; set_error("dist: rank " + std::to_string(p) + " reported a failure before the exchange"); }
        if (theirs[p] & 2) verify_now_ = true;
    }
    return all;
}

static uint64_t fnv1a(const unsigned char* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

// the data exchange, optionally verified (PF_DIST_VERIFY=1): every rank hashes what it sent to each peer and what it
// received from each peer; the hashes travel as a control message and must agree -- a first multi-GPU run that moves
// wrong bytes says which pair and how many bytes instead of producing a wrong mosaic
bool DistMap::exchange_checked(const std::vector<const void*>& send, const std::vector<size_t>& sb, const std::vector<void*>& recv,
                               const std::vector<size_t>& rb, const char* what)
{
    // verify_now_ was agreed on by the agree() that precedes every data exchange: all ranks take the same branch here
    if (!verify_now_) return t_->exchange_dev(send, sb, recv, rb, m_->stream());
    // Verified form: whatever goes wrong on THIS rank (the exchange itself, a copy, a hash that does not match) becomes a flag,
    // never an early return -- the hash exchange and the closing agree() below are collectives, and a rank that left early
    // would leave its peers waiting in them.  Every rank returns the same verdict.
    bool ok = t_->exchange_dev(send, sb, recv, rb, m_->stream());
    const int n = t_->nranks, me = t_->rank;
    struct Sum { uint64_t hash; uint64_t bytes; };
    std::vector<Sum> sent(n), got(n), claimed(n);
    std::vector<unsigned char> tmp;
    auto hash_of = [&](const void* dev, size_t bytes, uint64_t& h) {
        tmp.resize(bytes);
        if (hipMemcpy(tmp.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) { set_error(std::string("dist verify (") + what + "): device-to-host copy failed"); ok = false; return; }
        h = fnv1a(tmp.data(), bytes);
    };
    for (int p = 0; p < n; p++) {
        sent[p] = { 0, sb[p] }; got[p] = { 0, rb[p] }; claimed[p] = { 0, 0 };
        if (p == me) continue;
        if (sb[p]) hash_of(send[p], sb[p], sent[p].hash);
        if (rb[p]) hash_of(recv[p], rb[p], got[p].hash);
    }
    std::vector<const void*> s(n, nullptr); std::vector<void*> r(n, nullptr);
    std::vector<size_t> hs(n, sizeof(Sum)), hr(n, sizeof(Sum));
    for (int p = 0; p < n; p++) { s[p] = &sent[p]; r[p] = &claimed[p]; }
    hs[me] = hr[me] = 0;
    if (!t_->exchange_host(s, hs, r, hr, m_->stream())) ok = false;
    else
        for (int p = 0; p < n; p++) {
            if (p == me) continue;
            if (claimed[p].bytes != got[p].bytes || (got[p].bytes && claimed[p].hash != got[p].hash)) {
                char msg[256];
                std::snprintf(msg, sizeof msg, "dist verify (%s): rank %d sent %llu bytes (fnv %016llx), rank %d received %llu bytes (fnv %016llx)",
                              what, p, (unsigned long long)claimed[p].bytes, (unsigned long long)claimed[p].hash, me,
                              (unsigned long long)got[p].bytes, (unsigned long long)got[p].hash);
                set_error(msg);
                ok = false;
            }
        }
    const bool keep = verify_now_;
    ok = agree(ok);                                           // the sender of wrong bytes learns of it too
    verify_now_ = keep;
    stats_.verified += ok ? 1 : 0;
    return ok;
}

// plan_blend (the plan of one draw() across ranks, a pure function of the gathered tile lists): dist_plan.hpp

// draw() across ranks: blend this rank's changed tiles, with the strips of neighbours that live elsewhere
int DistMap::blend_changed(int* xy, uint8_t* bgr, int cap)
{
    const auto t_begin = std::chrono::steady_clock::now();
    stats_ = {};
    if (!m_->use_device()) return -1;
    const int n = t_->nranks, me = t_->rank;
    std::vector<std::vector<FusionMap::TileRec>> all;
    std::vector<long long> caps;
    if (!gather_lists(all, caps, cap)) return -1;
    // plan (pure function of the gathered lists, shared by every rank): who packs which strip for whom
    size_t hb9[9];
    for (int j = 0; j < 9; j++) hb9[j] = j == 4 ? 0 : m_->halo_bytes_for(j % 3 - 1, j / 3 - 1);
    BlendPlan plan;
    plan_blend(all, caps, me, m_->high_quality(), hb9, plan);
    std::vector<size_t>& send_bytes = plan.send_bytes;
    std::vector<size_t>& recv_bytes = plan.recv_bytes;
    std::vector<std::vector<FusionMap::StripReq>>& send_req = plan.send_req;
    std::vector<BlendPlan::Want>& wants = plan.wants;
    std::vector<std::pair<int, int>>& mine = plan.mine;
    // one send buffer and one receive buffer, peers back to back
    std::vector<size_t> soff(n, 0), roff(n, 0);
    size_t ts = 0, tr = 0;
    for (int p = 0; p < n; p++) { soff[p] = ts; ts += (send_bytes[p] + 255) / 256 * 256; roff[p] = tr; tr += (recv_bytes[p] + 255) / 256 * 256; }
    bool ok_here = send_.reserve(ts + 256) && recv_.reserve(tr + 256);
    std::vector<FusionMap::StripReq> reqs;
    for (int p = 0; p < n; p++) for (auto q : send_req[p]) { q.out_off += soff[p]; reqs.push_back(q); }
    const auto t_pack = std::chrono::steady_clock::now();
    ok_here = ok_here && m_->pack_strips(reqs, send_.p);      // ONE launch, ONE sync
    if (!agree(ok_here)) return -1;                           // every rank leaves here, or none does
    const auto t_xchg = std::chrono::steady_clock::now();
    if (n > 1) {
        std::vector<const void*> s(n, nullptr); std::vector<void*> r(n, nullptr);
        for (int p = 0; p < n; p++) { s[p] = (char*)send_.p + soff[p]; r[p] = (char*)recv_.p + roff[p]; }
        std::vector<size_t> sb = send_bytes, rb = recv_bytes;
        sb[me] = rb[me] = 0;
        if (!exchange_checked(s, sb, r, rb, "blend strips")) return -1;    // returns with the stream drained: the bytes have landed
    }
    const auto t_blend = std::chrono::steady_clock::now();
    std::vector<const void*> halo9(mine.size() * 9, nullptr);
    for (auto& w : wants) halo9[(size_t)w.tile * 9 + w.j] = (char*)recv_.p + roff[w.peer] + w.off;
    if (!mine.empty() && !m_->blend_tiles(mine, halo9.data(), bgr)) return -1;
    for (size_t i = 0; i < mine.size(); i++) { xy[2 * i] = mine[i].first; xy[2 * i + 1] = mine[i].second; }
    const auto t_end = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    for (int p = 0; p < n; p++) if (p != me) { stats_.bytes_sent += send_bytes[p]; stats_.bytes_received += recv_bytes[p]; stats_.peers += (send_bytes[p] || recv_bytes[p]) ? 1 : 0; }
    stats_.strips_sent = reqs.size(); stats_.strips_received = wants.size();
    stats_.plan_ms = ms(t_begin, t_pack); stats_.pack_ms = ms(t_pack, t_xchg); stats_.exchange_ms = ms(t_xchg, t_blend); stats_.compute_ms = ms(t_blend, t_end);
    stats_.tiles = mine.size();
    return (int)mine.size();
}

// save() across ranks: tiles travel once to rank 0, which collapses the whole mosaic (.cpp:779-847).  On the other ranks the
// call returns true with rows = cols = 0 (they hold no picture).
bool DistMap::save_to_memory(uint8_t* bgr, int* rows, int* cols, int* tx0, int* ty0)
{
    const auto t_begin = std::chrono::steady_clock::now();
    if (!m_->use_device()) return false;
    const int n = t_->nranks, me = t_->rank;
    if (n == 1) return m_->save_to_memory(bgr, rows, cols, tx0, ty0);
    std::vector<std::vector<FusionMap::TileRec>> all;
    std::vector<long long> caps;
    if (!gather_lists(all, caps, 0)) return false;
    const size_t nb = m_->tile_bytes();
    // query (bgr == nullptr): the extent is known from the lists alone
    if (!bgr) {
        int mnx = 1 << 30, mny = 1 << 30, mxx = -(1 << 30), mxy = -(1 << 30), cnt = 0;
        for (auto& l : all) for (auto& t : l) { cnt++; mnx = std::min(mnx, t.ix); mny = std::min(mny, t.iy); mxx = std::max(mxx, t.ix); mxy = std::max(mxy, t.iy); }
        if (!cnt) return false;
        if (me == 0) { *rows = (mxy + 1 - mny) * kElePixels; *cols = (mxx + 1 - mnx) * kElePixels; *tx0 = mnx; *ty0 = mny; }
        else { *rows = *cols = 0; *tx0 = *ty0 = 0; }
        return true;
    }
    std::vector<const void*> s(n, nullptr); std::vector<void*> r(n, nullptr);
    std::vector<size_t> sb(n, 0), rb(n, 0);
    stats_ = {};
    if (me != 0) {
        std::vector<std::pair<int, int>> mine;
        for (auto& t : all[me]) mine.push_back({ t.ix, t.iy });
        const bool ok_here = send_.reserve(nb * mine.size() + 256) && m_->export_tiles(mine, send_.p);
        if (!agree(ok_here)) return false;
        s[0] = send_.p; sb[0] = nb * mine.size();
        if (!exchange_checked(s, sb, r, rb, "save tiles")) return false;
        stats_.bytes_sent = sb[0]; stats_.tiles = mine.size();
        *rows = *cols = 0; *tx0 = *ty0 = 0;
        stats_.exchange_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        return true;
    }
    size_t total = 0;
    std::vector<size_t> off(n, 0);
    for (int p = 1; p < n; p++) { off[p] = total; total += nb * all[p].size(); }
    if (!agree(recv_.reserve(total + 256))) return false;
    for (int p = 1; p < n; p++) { r[p] = (char*)recv_.p + off[p]; rb[p] = nb * all[p].size(); }
    if (!exchange_checked(s, sb, r, rb, "save tiles")) return false;
    const auto t_x = std::chrono::steady_clock::now();
    std::vector<FusionMap::ForeignTile> foreign;
    for (int p = 1; p < n; p++)
        for (size_t k = 0; k < all[p].size(); k++) foreign.push_back({ all[p][k].ix, all[p][k].iy, (char*)recv_.p + off[p] + k * nb });
    const bool ok = m_->save_to_memory(bgr, rows, cols, tx0, ty0, &foreign);
    stats_.bytes_received = total; stats_.tiles = foreign.size();
    stats_.exchange_ms = std::chrono::duration<double, std::milli>(t_x - t_begin).count();
    stats_.compute_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_x).count();
    return ok;
}

// Map2D::feed across ranks (SURVEY 8e "one H2D + P2P over xGMI"): the tracker hands its keyframe to ONE rank (`root`,
// host pixels); every rank makes the call with the same pose and frame description.  All ranks derive, from the pose
// alone, which ranks own a tile of the frame's canvas; the root copies the pixels to its GPU once and sends them -- one
// grouped point-to-point exchange -- to exactly those ranks; every rank then renders its tiles (or, holding none of the
// canvas, only advances its grid).  No rank is fed through its own PCIe link except the root.
int DistMap::rank() const { return t_->rank; }

int DistMap::feed(const pf_image* img, const double pose7[7], int root, const FusionMap::FrameProducer* produce)
{
    stats_ = {};
    if (!img || !m_->use_device()) return -1;
    const int n = t_->nranks, me = t_->rank;
    if (root < 0 || root >= n) { set_error("pf_dist_feed: root out of range"); return -1; }
    // a frame that does not match the camera is rejected as Map2D::feed rejects it: 0 on every rank (each sees the same
    // description), before the grid is touched and before any collective
    if (m_->reject_mismatched_frame(img)) return 0;
    std::vector<unsigned char> needs;
    bool ok_here = m_->frame_needs(pose7, needs) && (int)needs.size() == n;
    if (ok_here && me == root && !img->data && !produce) { set_error("pf_dist_feed: the root has no pixels"); ok_here = false; }
    bool anyone = false;
    for (int p = 0; ok_here && p < n; p++) anyone = anyone || (needs[p] && p != root);
    int slot = -1; void* dev = nullptr; size_t bytes = 0;
    const bool i_take = ok_here && needs[me], i_send = ok_here && me == root && anyone;
    if (i_take || i_send) {
        slot = m_->stage_frame(img, me == root && !produce, &dev, &bytes);
        ok_here = slot >= 0;
        if (ok_here && me == root && produce) ok_here = (*produce)(dev, m_->stream());          // stream order puts it ahead of the exchange and the render
    }
    if (!agree(ok_here)) { m_->release_staged(slot); return -1; }
    if (n > 1 && anyone) {
        std::vector<const void*> s(n, nullptr); std::vector<void*> r(n, nullptr);
        std::vector<size_t> sb(n, 0), rb(n, 0);
        if (me == root) { for (int p = 0; p < n; p++) if (p != root && needs[p]) { s[p] = dev; sb[p] = bytes; stats_.bytes_sent += bytes; stats_.peers++; } }
        else if (needs[me]) { r[root] = dev; rb[root] = bytes; stats_.bytes_received = bytes; stats_.peers = 1; }
        const auto t0 = std::chrono::steady_clock::now();
        if (!exchange_checked(s, sb, r, rb, "keyframe")) { m_->release_staged(slot); return -1; }
        stats_.exchange_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    // a root that owns none of the canvas staged the frame only to send it: release the slot through a geometry-only feed
    const bool rendered = m_->feed_staged(needs[me] ? slot : -1, img, pose7);
    if (!needs[me] && slot >= 0) m_->release_staged(slot);
    stats_.tiles = needs[me];
    return rendered ? 1 : 0;
}

bool DistMap::save(const char* filename)
{
    int rows = 0, cols = 0, tx0 = 0, ty0 = 0;
    if (!save_to_memory(nullptr, &rows, &cols, &tx0, &ty0)) return false;
    std::vector<uint8_t> img((size_t)rows * cols * 3 + 1);
    if (!save_to_memory(img.data(), &rows, &cols, &tx0, &ty0)) return false;
    if (t_->rank != 0) return true;
    if (!write_image_file(filename, img.data(), rows, cols)) return false;
    std::printf("Resolution:[%d %d]\n", cols, rows);
    return true;
}

}  // namespace pf
