// dist_plan.hpp -- the plan of one draw() across ranks: a pure function of the all-gathered tile lists (no device, no transport, no
// HIP header), so that the CPU tests, the sanitizer targets (tests/cpp) and every rank run the very same code.
#pragma once
#include <cstddef>
#include <map>
#include <utility>
#include <vector>

namespace pf {

struct TileRec  { int ix, iy, changed; };
struct StripReq { int ix, iy, dx, dy; size_t out_off; };      // tile (ix,iy) of THIS rank hands its edge facing (-dx,-dy)... see pf_halo_pack

struct BlendPlan {
    std::vector<size_t> send_bytes, recv_bytes;                       // per peer
    std::vector<std::vector<StripReq>> send_req;                      // per requesting peer: this rank's tile to pack, offset inside the peer's region
    struct Want { int tile, j, peer; size_t off; };                   // this rank's tile `tile` (index in mine) gets neighbour j from peer
    std::vector<Want> wants;
    std::vector<std::pair<int, int>> mine;                            // this rank's changed tiles, in blend order, at most its cap
};

// The plan of one draw() across ranks, from the all-gathered tile lists alone (no device, no transport: also exported as
// pf_dist_plan_blend so that the CPU tests run THIS code between processes).  For rank r's changed tile whose 3x3
// neighbourhood exists somewhere (Ele::blend's condition, .cpp:93-117), every neighbour held by another rank p contributes
// one strip set p -> r.  Order: requester's tiles by (iy,ix), neighbours by j = 3*(dy+1)+(dx+1); every rank derives the
// same order, so no header travels with the payload.
inline void plan_blend(const std::vector<std::vector<TileRec>>& all, const std::vector<long long>& caps, int me, bool hq,
                const size_t halo_bytes9[9], BlendPlan& out)
{
    const int n = (int)all.size();
    std::map<std::pair<int, int>, int> owner;                 // (ix,iy) -> rank holding it
    for (int p = 0; p < n; p++) for (auto& t : all[p]) owner[{ t.ix, t.iy }] = p;
    out.send_bytes.assign(n, 0); out.recv_bytes.assign(n, 0);
    out.send_req.assign(n, {});
    out.wants.clear(); out.mine.clear();
    for (int r = 0; r < n; r++) {
        long long taken = 0;
        for (auto& t : all[r]) {
            if (!t.changed) continue;
            if (taken >= caps[r]) break;                      // rank r blends at most its own cap tiles in this call
            if (r == me) out.mine.push_back({ t.ix, t.iy });
            taken++;
            if (!hq) continue;
            bool full = true;
            for (int j = 0; j < 9 && full; j++) full = owner.count({ t.ix + j % 3 - 1, t.iy + j / 3 - 1 }) != 0;
            if (!full) continue;                              // blends alone (.cpp:134-145): no strips
            for (int j = 0; j < 9; j++) {
                if (j == 4) continue;
                const int dx = j % 3 - 1, dy = j / 3 - 1, p = owner[{ t.ix + dx, t.iy + dy }];
                if (p == r) continue;
                const size_t nb = halo_bytes9[j];
                if (p == me) { out.send_req[r].push_back({ t.ix + dx, t.iy + dy, dx, dy, out.send_bytes[r] }); out.send_bytes[r] += nb; }
                if (r == me) { out.wants.push_back({ (int)out.mine.size() - 1, j, p, out.recv_bytes[p] }); out.recv_bytes[p] += nb; }
            }
        }
    }
}

}  // namespace pf
