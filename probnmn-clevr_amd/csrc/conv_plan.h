// Launch planner of the grouped convolution (host side): how a call's units are cut into segments of different splits.
// Used by conv_nhwc.hip (one launch per call).
#pragma once
#include <stdlib.h>

namespace pnmn {

// Launch plan (in units).  A launch of n workgroups on 256 CUs costs ceil(n / 256) rounds, and a last round that holds
// 8 workgroups costs as much as a full one (520 stem items = 3 rounds for 2.03 rounds of work).  So the units are cut
// into up to THREE segments of one launch, of non-decreasing split: as many as fill whole rounds go out with the first
// split, the remainder follows with larger splits -- whose rounds are shorter -- again in whole rounds first (667
// module-conv items: 512 at split 1, 128 at split 2 -- exactly one round of half the work -- and 27 at split 8).
// Relative costs only: one tap of one 128-channel chunk = 1 unit, every further segment ~ 0.3 unit.
struct LaunchPlan {
    int n_seg;
    int split[3], count[3];
};

// One split for every launch (0 = the planner decides): PNMN_CONV_KSPLIT=<1|2|4|8> or pnmn_conv_force_split() --
// scripts/conv_modes.py measures the splits with it, and tests/test_nmn_gpu.py pins each to show that the results do not
// depend on the split.  One instance per library (inline, C++17).
inline int& forced_split() {
    static int forced = [] {
        const char* e = getenv("PNMN_CONV_KSPLIT");
        const int v = e ? atoi(e) : 0;
        return (v == 1 || v == 2 || v == 4 || v == 6 || v == 8 || v == 14 || v == 26) ? v : 0;
    }();
    return forced;
}

// CUs a launch is planned for when the caller does not say (PNMN_CONV_CUS: tuning hook)
inline int default_conv_cus() {
    static const int v = [] {
        const char* e = getenv("PNMN_CONV_CUS");
        const long c = e ? atol(e) : 256;
        return (int)(c >= 8 && c <= 256 ? (c & ~7L) : 256);
    }();
    return v;
}

// Staging is hidden behind the contraction (no per-chunk cost); a workgroup computes 128 / split output channels;
// split 2 halves a wave's channels, split 4 / 8 also share an item's 13 m-tiles among 2 / 4 waves per channel tile
// (7 / 4 tiles on the longest wave).
inline LaunchPlan plan_launch(int n_items, int cout_blocks, int cin_chunks, int ntaps, int cu_budget = 0) {
    if (forced_split() && !((forced_split() == 6 || forced_split() > 8) && ntaps != 9)) return LaunchPlan{1, {forced_split(), 0, 0}, {n_items, 0, 0}};
    const double work = (double)ntaps * cin_chunks;
    constexpr int max_seg = 3;
    constexpr double seg_cost = 0.3;    // a further segment: its workgroups start behind a partly drained round
    constexpr double overhead = 0.25;   // start-up, epilogue
    // (a wave's (channel tile, m-tile) pairs: 26 / 13 / 7 / 5 / 4 / 2 / 1 at split 1 / 2 / 4 / 6 / 8 / 14 / 26)
    auto round_cost = [&](int s) {
        const double pairs = s == 1 ? 26.0 : s == 2 ? 13.0 : s == 4 ? 7.0 : s == 6 ? 5.0 : s == 8 ? 4.0 : s == 14 ? 2.0 : 1.0;
        return work * pairs / 26.0 + overhead;
    };
    // the splits a segment may take, ascending; 6 / 14 / 26 (3 / 7 / 13 waves per channel tile) exist for the 3x3 bodies only
    const int splits[7] = {1, 2, 4, ntaps == 9 ? 6 : 8, 8, 14, 26};
    const int n_splits = ntaps == 9 ? 7 : 4;
    LaunchPlan best{1, {1, 0, 0}, {n_items, 0, 0}};
    double best_t = 1e30;
    // CUs a round is planned for: all 256, unless the caller says the launch shares the chip (pnmn_conv_nhwc_cus: the
    // joint step's trunk runs on its own stream beside the seq2seq passes, whose multi-CU kernels hold 64-96 CUs for
    // hundreds of microseconds -- a launch cut for 256 workgroups then takes two rounds where one cut for the free
    // CUs takes one: 128-question step 7.33 -> 7.06 ms at 192, gpurun_out/r03w_ab.txt).  PNMN_CONV_CUS overrides the
    // default of launches that do not say (tuning hook).
    const long cus = (cu_budget >= 8 && cu_budget <= 256) ? cu_budget : default_conv_cus();
    auto rounds_of = [&](long n, int s) { return (n * cout_blocks * s + cus - 1) / cus; };
    auto full_of = [&](long n, int s) {  // items that fill whole rounds at split s
        const long per_item = (long)cout_blocks * s;
        const long m = ((long)n * per_item / cus) * cus / per_item;
        return m > n ? n : m;
    };
    auto consider = [&](const LaunchPlan& p, double t) {
        if (t < best_t * 0.97) {  // prefer fewer launches / smaller splits unless the gain is real
            best_t = t;
            best = p;
        }
    };
    for (int i0 = 0; i0 < n_splits; ++i0) {
        const int s0 = splits[i0];
        // one launch
        consider(LaunchPlan{1, {s0, 0, 0}, {n_items, 0, 0}}, (double)rounds_of(n_items, s0) * round_cost(s0));
        if (max_seg < 2) continue;
        const long m0 = full_of(n_items, s0);
        if (m0 <= 0 || m0 >= n_items) continue;
        const double t0 = (double)rounds_of(m0, s0) * round_cost(s0);
        const long r0 = n_items - m0;
        for (int i1 = i0; i1 < n_splits; ++i1) {
            const int s1 = splits[i1];
            consider(LaunchPlan{2, {s0, s1, 0}, {(int)m0, (int)r0, 0}}, t0 + (double)rounds_of(r0, s1) * round_cost(s1) + seg_cost);
            if (max_seg < 3 || s1 == s0) continue;
            const long m1 = full_of(r0, s1);
            if (m1 <= 0 || m1 >= r0) continue;
            const double t1 = t0 + (double)rounds_of(m1, s1) * round_cost(s1) + seg_cost;
            const long r1 = r0 - m1;
            for (int i2 = i1 + 1; i2 < n_splits; ++i2)
                consider(LaunchPlan{3, {s0, s1, splits[i2]}, {(int)m0, (int)m1, (int)r1}},
                         t1 + (double)rounds_of(r1, splits[i2]) * round_cost(splits[i2]) + seg_cost);
        }
    }
    return best;
}

}  // namespace pnmn
