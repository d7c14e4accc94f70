// sched_common.hpp — per-lane constants of a scheduling instance (lane d <-> day d), built on the
// host once per ls_sched_instance_upload and read by every wavefront of the scheduling kernels.
#pragma once
#include <cstdint>

namespace lsamd {

struct SchedLaneConst {
    uint64_t scored_mask;  // bits of the scored days (same in every lane)
    uint64_t weekend_mask; // scored days that are Sat/Sun (same in every lane)
    uint64_t mask2;        // bit d+1 if day d+1 is scored                (H2, lib.rs:286-292)
    uint64_t h3mask;       // days compared with day d in a weekend-started 9-window (H3, lib.rs:295-315)
    uint64_t wdmask;       // scored days on day d's weekday; 0 on weekends / unscored lanes (S2, lib.rs:194-218)
    uint64_t adj;          // every scored day paired with day d by H2 or H3, in either direction (delta scorer)
    int32_t gsize;         // popcount(wdmask)
    int32_t flags;         // bit 0: lane is a scored day; bit 1: first lane of a Mon-Fri weekday group
};

inline void sched_build_lane_table(int D, int start_weekday, SchedLaneConst out[64])
{
    auto weekend = [&](int q) { return (start_weekday + q) % 7 >= 5; }; // is_weekend, lib.rs:220-222
    const uint64_t scored_mask = D >= 64 ? ~0ull : ((1ull << D) - 1);
    uint64_t wm = 0;
    for (int q = 0; q < D && q < 64; q++)
        if (weekend(q)) wm |= 1ull << q;
    const uint64_t every7 = 0x8102040810204081ull; // bits 0, 7, 14, ..., 63
    for (int d = 0; d < 64; d++) {
        SchedLaneConst L{};
        const bool scored = d < D;
        L.scored_mask = scored_mask;
        L.weekend_mask = wm;
        L.mask2 = (scored && d + 1 < D) ? (1ull << (d + 1)) : 0;
        uint64_t h3 = 0;
        if (scored) {
            // day d as window[0]: windows(9) starting at d with d, d+1 both weekend days: pairs (d,d+7), (d,d+8)
            if (d + 8 < D && weekend(d) && weekend(d + 1)) h3 |= (1ull << (d + 7)) | (1ull << (d + 8));
            // day d as window[1]: window starts at d-1: pairs (d,d+6), (d,d+7)
            if (d >= 1 && d - 1 + 8 < D && weekend(d - 1) && weekend(d)) h3 |= (1ull << (d + 6)) | (1ull << (d + 7));
        }
        L.h3mask = h3;
        const bool weekday = (start_weekday + d) % 7 < 5;
        L.wdmask = (scored && weekday) ? ((every7 << (d % 7)) & scored_mask) : 0;
        L.gsize = __builtin_popcountll(L.wdmask);
        L.flags = (scored ? 1 : 0) | ((scored && weekday && d < 7) ? 2 : 0);
        out[d] = L;
    }
    for (int d = 0; d < 64; d++) { // the pair relation of H2 / H3, symmetric
        uint64_t adj = out[d].mask2 | out[d].h3mask;
        for (int q = 0; q < 64; q++)
            if ((out[q].mask2 | out[q].h3mask) >> d & 1) adj |= 1ull << q;
        out[d].adj = adj;
    }
}

} // namespace lsamd
