// Host-side batch planner (no device work): the per-step half of probnmn/runtime/schedule.py.
//
// A joint-training step samples its programs on the device; between the sampling decode and the first module
// launch the host has to turn them into level-ordered work lists (the reference instead interprets one
// example at a time, probnmn/models/nmn.py:191-238).  The numpy formulation of this -- ~250 whole-array
// operations, 0.75 ms for 65 programs -- sat on the step's critical path at small batches; here it is one
// call.  Inputs: the template bank (a program STRUCTURE expanded into a table of primitives with levels and
// arena offsets, built once per structure in Python), the batch's template ids / call tokens / arena block
// bases, the per-token weight offset tables and the device base addresses.  Outputs: the record matrices of
// every grouped launch, bit-identical to the C structs of probnmn_hip.h, sorted by (level, weight), plus the
// (level, begin, end) cuts.  Checked row for row against the numpy planner (tests/test_schedule.py).
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <unordered_map>
#include <vector>

#include "../../include/probnmn_hip.h"

namespace {

// columns of a template row: keep in sync with schedule.py
enum { C_KIND, C_LEVEL, C_CALL, C_WIDX, C_DIL, C_AK, C_AO, C_BK, C_BO, C_OK, C_OO, C_ACH, C_BCH, C_ISMAX,
       C_MASKED, C_SCRATCH, C_PA, C_PB, NCOLS };
enum { L_SLOT, L_FEAT, L_ONES, L_FINAL };
enum { K_CONV, K_PROJ, K_DOT, K_SAME, K_MINMAX };
// record kinds of the output, in output order
enum { R_CONV, R_DGRAD, R_WG3, R_JOBS3, R_PROJ, R_PDGRAD, R_WGP, R_JOBSP, R_DOT, R_SAME, R_MINMAX, R_MASKBWD, R_COUNT };
// cut kinds
enum { CUT_CONV, CUT_PROJ, CUT_DOT, CUT_SAME, CUT_MINMAX, CUT_DGRAD, CUT_MASKBWD, CUT_PDGRAD, CUT_WGROUP };

struct Prim {
    const int64_t* r;
    int64_t ex, tok, blockbase;
    uint64_t a_f, a_g, b_f, b_g, o_f, o_g;
};

struct Mat {  // rows of `cols` uint64, zero initialised
    int cols;
    std::vector<uint64_t> v;
    explicit Mat(int c) : cols(c) {}
    uint64_t* add() {
        v.resize(v.size() + cols, 0);
        return v.data() + v.size() - cols;
    }
    void clear() { v.clear(); }
    size_t rows() const { return cols ? v.size() / cols : 0; }
    const uint64_t* row(size_t i) const { return v.data() + i * cols; }
};

// stable order by (level, key) -- numpy's lexsort((key, level)) -- or by level only.  The keys are copied next
// to the indices first: sorting indices through a comparator that chases two rows per comparison is several
// times slower (cache misses) than sorting 24-byte records.
struct SortKey {
    int64_t level;
    uint64_t key;
    int idx;
    bool operator<(const SortKey& o) const {
        if (level != o.level) return level < o.level;
        if (key != o.key) return key < o.key;
        return idx < o.idx;  // ties keep their original order
    }
};

std::vector<int> order_by(const std::vector<int64_t>& level, const Mat* m, int wcol) {
    std::vector<SortKey> keys(level.size());
    for (size_t i = 0; i < keys.size(); ++i) keys[i] = SortKey{level[i], m ? m->row(i)[wcol] : 0, (int)i};
    std::sort(keys.begin(), keys.end());
    std::vector<int> idx(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) idx[i] = keys[i].idx;
    return idx;
}

struct Out {
    uint64_t* words;
    int64_t cap, used;
    int64_t* meta;
    int32_t* cuts;
    int n_cuts, cap_cuts;
    bool overflow;

    void put(int kind, const Mat& m, const std::vector<int>* idx) {
        const int64_t n = (int64_t)m.rows();
        meta[2 + 3 * kind] = used;
        meta[3 + 3 * kind] = n;
        meta[4 + 3 * kind] = m.cols;
        if (used + n * m.cols > cap) {
            overflow = true;
            return;
        }
        for (int64_t i = 0; i < n; ++i)
            memcpy(words + used + i * m.cols, m.row(idx ? (*idx)[i] : i), sizeof(uint64_t) * m.cols);
        used += n * m.cols;
    }
    void cut(int kind, const std::vector<int64_t>& sorted_levels) {  // (level, begin, end) runs
        const size_t n = sorted_levels.size();
        size_t b = 0;
        while (b < n) {
            size_t e = b + 1;
            while (e < n && sorted_levels[e] == sorted_levels[b]) ++e;
            one(kind, sorted_levels[b], (int)b, (int)e);
            b = e;
        }
    }
    void one(int kind, int64_t level, int b, int e) {
        if (n_cuts >= cap_cuts) {
            overflow = true;
            return;
        }
        int32_t* c = cuts + 4 * n_cuts++;
        c[0] = kind, c[1] = (int32_t)level, c[2] = b, c[3] = e;
    }
};

std::vector<int64_t> permuted(const std::vector<int64_t>& v, const std::vector<int>& idx) {
    std::vector<int64_t> o(v.size());
    for (size_t i = 0; i < v.size(); ++i) o[i] = v[idx[i]];
    return o;
}

// schedule.py:_wgrad_jobs -- sort the items by key (stable), cut each run of one key into jobs of at most
// `chunk` items; returns the order, fills `jobs` (dw, dbias, begin | end << 32) and the group of each job
void wgrad_jobs(const std::vector<int64_t>& key, const std::vector<uint64_t>& dw, const std::vector<uint64_t>& db,
                const std::vector<int64_t>& group, int chunk, std::vector<int>& idx, Mat& jobs, std::vector<int64_t>& jgroup) {
    const size_t n = key.size();
    idx = order_by(key, nullptr, 0);
    size_t g0 = 0;
    while (g0 < n) {
        size_t g1 = g0 + 1;
        while (g1 < n && key[idx[g1]] == key[idx[g0]]) ++g1;
        for (size_t j = g0; j < g1; j += chunk) {
            const size_t je = std::min(j + (size_t)chunk, g1);
            uint64_t* r = jobs.add();
            r[0] = dw[idx[j]], r[1] = db[idx[j]];
            r[2] = (uint64_t)j | ((uint64_t)je << 32);
            jgroup.push_back(group[idx[j]]);
        }
        g0 = g1;
    }
}

}  // namespace

extern "C" int pnmn_plan_batch(const pnmn_plan_in* in, uint64_t* out_words, int64_t out_capacity, int64_t* meta,
                               int32_t* cuts, int32_t cuts_capacity) {
    if (!in || !out_words || !meta || !cuts) return PNMN_EINVAL;
    const int nv = in->nv;
    memset(meta, 0, sizeof(int64_t) * (2 + 3 * R_COUNT + 1));
    if (nv <= 0) return 0;
    if (!in->tables || !in->nprims || !in->tids || !in->examples || !in->base || !in->tokens || !in->w3 || !in->b3 ||
        !in->wt3 || !in->dotw || !in->dotb || in->cmax < 1 || in->wgrad_chunk < 1)
        return PNMN_EINVAL;
    const int64_t C = in->channels, map_bytes = (int64_t)in->hw * C * 4;
    const uint64_t kind_base[4] = {in->act, in->feat, 0, in->final_};
    const uint64_t kind_gbase[4] = {in->gact, in->gfeat, 0, in->gfinal};
    const int64_t kind_stride[4] = {0, map_bytes, 0, map_bytes};

    // Working storage is kept from call to call (per thread): a step's lists take a few megabytes, and handing
    // them back to the allocator each time means faulting the pages in again on the next step -- more time
    // than the arithmetic.
#define KEEP(type, name, ...)                  \
    static thread_local type name{__VA_ARGS__}; \
    name.clear()

    // every primitive of every valid example, example-major (the order of the gathered tables)
    KEEP(std::vector<Prim>, prims);
    for (int i = 0; i < nv; ++i) {
        const int64_t tid = in->tids[i];
        if (tid < 0 || tid >= in->n_templates) return PNMN_EINVAL;
        const int64_t np = in->nprims[tid];
        for (int64_t p = 0; p < np; ++p) {
            Prim q;
            q.r = in->tables + ((size_t)tid * in->pmax + p) * NCOLS;
            q.ex = in->examples[i];
            const int64_t call = q.r[C_CALL];
            if (call < 0 || call >= in->cmax) return PNMN_EINVAL;
            q.tok = in->tokens[(size_t)i * in->cmax + call];
            q.blockbase = in->base[i];
            auto addr = [&](int kcol, int ocol, bool grad) -> uint64_t {
                const int64_t k = q.r[kcol];
                uint64_t a = (grad ? kind_gbase : kind_base)[k] + (uint64_t)(kind_stride[k] * q.ex);
                if (k == L_SLOT) a += (uint64_t)((q.blockbase + q.r[ocol]) * 4);
                return a;
            };
            q.a_f = addr(C_AK, C_AO, false), q.a_g = addr(C_AK, C_AO, true);
            q.b_f = addr(C_BK, C_BO, false), q.b_g = addr(C_BK, C_BO, true);
            q.o_f = addr(C_OK, C_OO, false), q.o_g = addr(C_OK, C_OO, true);
            prims.push_back(q);
        }
    }
    const size_t N = prims.size();
    int64_t depth = 0;
    for (const Prim& q : prims) depth = std::max(depth, q.r[C_LEVEL]);
    meta[0] = (int64_t)N;
    meta[1] = depth;

    Out out{out_words, out_capacity, 0, meta, cuts, 0, cuts_capacity, false};
    auto P = [&](int64_t float_offset) { return in->params + (uint64_t)(float_offset * 4); };
    auto G = [&](int64_t float_offset) { return in->grads + (uint64_t)(float_offset * 4); };

    // ---- 3x3 convs ------------------------------------------------------------------------------
    {
        KEEP(Mat, fw, 12);
        KEEP(Mat, dg, 12);
        KEEP(Mat, wg, 6);
        KEEP(Mat, mb, 5);
        KEEP(Mat, jobs, 3);
        KEEP(std::vector<int64_t>, lv);
        KEEP(std::vector<int64_t>, wkey);
        KEEP(std::vector<int64_t>, grp);
        KEEP(std::vector<uint64_t>, dw);
        KEEP(std::vector<uint64_t>, db);
        KEEP(std::vector<uint64_t>, scratch_v);
        KEEP(std::vector<char>, masked_v);
        KEEP(std::vector<const Prim*>, src);
        int64_t depth3 = 0;
        for (const Prim& q : prims)
            if (q.r[C_KIND] == K_CONV) depth3 = std::max(depth3, q.r[C_LEVEL]);
        // "sole writer": no other masked conv of the same level adds into this example's dFEAT map
        static thread_local std::unordered_map<uint64_t, int> writers;
        writers.clear();
        for (const Prim& q : prims)
            if (q.r[C_KIND] == K_CONV && q.r[C_MASKED] == 1)
                ++writers[(uint64_t)q.r[C_LEVEL] * (1ull << 48) + (q.a_g >> 4)];
        for (const Prim& q : prims) {
            if (q.r[C_KIND] != K_CONV) continue;
            const int64_t t = q.tok, w = q.r[C_WIDX], level = q.r[C_LEVEL], dil = q.r[C_DIL];
            const bool masked = q.r[C_MASKED] == 1;
            const uint64_t mask_ptr = masked ? q.b_f : 0;  // 0 for the all-ones attention too
            const int64_t w_off = in->w3[t * 6 + w], b_off = in->b3[t * 6 + w], wt_off = in->wt3[t * 6 + w];
            const uint64_t scratch = in->gact + (uint64_t)((q.blockbase + q.r[C_SCRATCH]) * 4);
            uint64_t* f = fw.add();
            f[0] = q.a_f, f[2] = mask_ptr, f[4] = P(w_off), f[5] = P(b_off), f[6] = q.o_f, f[7] = (uint64_t)dil;
            uint64_t* d = dg.add();
            d[0] = q.o_g, d[3] = q.o_f, d[4] = in->wt + (uint64_t)(wt_off * 4);
            if (in->fuse_mask_bwd == 2) {
                // deferred d(feats): dx is stored to the conv's scratch map, only d(attention) stays in the epilogue;
                // pnmn_feat_grad_gather sums the scratch maps per example at the end of the backward pass
                d[6] = masked ? scratch : q.a_g;
                const bool att = masked && mask_ptr != 0;
                d[7] = (uint64_t)dil + (att ? (16ull << 32) : 0);
                d[8] = att ? q.a_f : 0;
                d[9] = att ? mask_ptr : 0;
                d[11] = att ? q.b_g : 0;
            } else if (in->fuse_mask_bwd) {
                d[6] = masked ? 0 : q.a_g;
                const bool sole = masked && in->sole_writer &&
                                  writers[(uint64_t)level * (1ull << 48) + (q.a_g >> 4)] == 1;
                d[7] = (uint64_t)dil + (masked ? (4ull << 32) : 0) + (sole ? (8ull << 32) : 0);
                d[8] = masked ? q.a_f : 0;
                d[9] = mask_ptr;
                d[10] = masked ? q.a_g : 0;
                d[11] = (masked && mask_ptr != 0) ? q.b_g : 0;
            } else {
                d[6] = masked ? scratch : q.a_g;
                d[7] = (uint64_t)dil;
            }
            uint64_t* g = wg.add();
            g[0] = q.a_f, g[2] = mask_ptr, g[3] = q.o_g, g[4] = q.o_f, g[5] = (uint64_t)dil;
            lv.push_back(level);
            const int64_t gid = (depth3 - level) * in->wgrad_groups / std::max<int64_t>(depth3, 1);
            grp.push_back(gid);
            wkey.push_back(gid * 4096 + t * 8 + w);
            dw.push_back(G(w_off));
            db.push_back(G(b_off));
            masked_v.push_back(masked);
            scratch_v.push_back(scratch);
            src.push_back(&q);
        }
        if (!lv.empty()) {
            const std::vector<int> idx = order_by(lv, in->sort_by_weight ? &fw : nullptr, 4);
            out.put(R_CONV, fw, &idx);
            out.cut(CUT_CONV, permuted(lv, idx));
            const std::vector<int> didx = order_by(lv, in->sort_by_weight ? &dg : nullptr, 4);
            out.put(R_DGRAD, dg, &didx);
            out.cut(CUT_DGRAD, permuted(lv, didx));
            if (in->fuse_mask_bwd == 2) {  // the gather's items: every masked conv, sorted by the d(feats) map it adds into
                std::vector<std::pair<uint64_t, int>> order;
                for (int i : idx)
                    if (masked_v[i]) order.emplace_back(src[i]->a_g, i);
                std::stable_sort(order.begin(), order.end(),
                                 [](const std::pair<uint64_t, int>& x, const std::pair<uint64_t, int>& y) { return x.first < y.first; });
                for (const auto& o : order) {
                    const Prim& q = *src[o.second];
                    uint64_t* m = mb.add();
                    m[0] = scratch_v[o.second], m[1] = q.a_f, m[2] = q.b_f, m[3] = q.a_g, m[4] = 0;
                }
                out.put(R_MASKBWD, mb, nullptr);
            } else if (!in->fuse_mask_bwd) {  // separate mask-backward records, in the forward order of the masked convs
                std::vector<int64_t> mlv;
                for (int i : idx) {
                    if (!masked_v[i]) continue;
                    const Prim& q = *src[i];
                    uint64_t* m = mb.add();
                    const uint64_t mask_ptr = q.b_f;
                    m[0] = scratch_v[i], m[1] = q.a_f, m[2] = mask_ptr, m[3] = q.a_g, m[4] = mask_ptr != 0 ? q.b_g : 0;
                    mlv.push_back(lv[i]);
                }
                out.put(R_MASKBWD, mb, nullptr);
                out.cut(CUT_MASKBWD, mlv);
            }
            std::vector<int> widx;
            std::vector<int64_t> jgrp;
            wgrad_jobs(wkey, dw, db, grp, in->wgrad_chunk, widx, jobs, jgrp);
            out.put(R_WG3, wg, &widx);
            out.put(R_JOBS3, jobs, nullptr);
            // job groups (deepest levels first): (lowest level of the group, first job, one past the last)
            size_t b = 0;
            while (b < jgrp.size()) {
                size_t e = b + 1;
                while (e < jgrp.size() && jgrp[e] == jgrp[b]) ++e;
                int64_t lo = INT64_MAX;
                for (size_t i = 0; i < lv.size(); ++i)
                    if (grp[i] == jgrp[b]) lo = std::min(lo, lv[i]);
                out.one(CUT_WGROUP, lo, (int)b, (int)e);
                b = e;
            }
        }
    }

    // ---- projections (ComparisonModule) -----------------------------------------------------------
    {
        KEEP(Mat, fw, 12);
        KEEP(Mat, pda, 12);
        KEEP(Mat, pdb, 12);
        KEEP(Mat, wg, 6);
        KEEP(Mat, jobs, 3);
        KEEP(std::vector<int64_t>, lv);
        KEEP(std::vector<int64_t>, key);
        KEEP(std::vector<int64_t>, grp);
        KEEP(std::vector<uint64_t>, dw);
        KEEP(std::vector<uint64_t>, db);
        for (const Prim& q : prims) {
            if (q.r[C_KIND] != K_PROJ) continue;
            const int64_t t = q.tok;
            const int64_t w_off = in->w3[t * 6], b_off = in->b3[t * 6], wt_off = in->wt3[t * 6];
            uint64_t* f = fw.add();
            f[0] = q.a_f, f[1] = q.b_f, f[4] = P(w_off), f[5] = P(b_off), f[6] = q.o_f, f[7] = 1;
            uint64_t* a = pda.add();
            a[0] = q.o_g, a[3] = q.o_f, a[4] = in->wt + (uint64_t)(wt_off * 4), a[6] = q.a_g, a[7] = 1ull | (1ull << 32);
            uint64_t* b = pdb.add();
            b[0] = q.o_g, b[3] = q.o_f, b[4] = in->wt + (uint64_t)((wt_off + C * C) * 4), b[6] = q.b_g, b[7] = 1ull | (1ull << 32);
            uint64_t* g = wg.add();
            g[0] = q.a_f, g[1] = q.b_f, g[3] = q.o_g, g[4] = q.o_f;
            lv.push_back(q.r[C_LEVEL]);
            key.push_back(t);
            grp.push_back(0);
            dw.push_back(G(w_off));
            db.push_back(G(b_off));
        }
        if (!lv.empty()) {
            const std::vector<int> idx = order_by(lv, in->sort_by_weight ? &fw : nullptr, 4);
            out.put(R_PROJ, fw, &idx);
            out.cut(CUT_PROJ, permuted(lv, idx));
            // two data gradients per projection (one per operand), both in ONE launch per level: they add (non-atomic
            // read-modify-write) into the gradients of two different values, and a launch of 2-7 items is all latency (15-20 us
            // each, 14 of them per 1024-question step).  The guard is structural: if ANY map is some item's first operand and
            // some item's second operand in this batch, the halves go out in launches of their own.  (Inside a half no two
            // items of one level share a map: the interpreter is a two-register machine, nmn.py:197-238 -- `saved_output` may
            // feed several binary modules of a program, but each also takes the running `output`, which depends on the
            // previous one, so the uses lie on different levels; different examples own disjoint arena blocks.)
            bool same_operand = false;
            {
                std::vector<uint64_t> as, bs;
                for (size_t i = 0; i < pda.rows(); ++i) as.push_back(pda.row(i)[6]), bs.push_back(pdb.row(i)[6]);
                std::sort(as.begin(), as.end());
                std::sort(bs.begin(), bs.end());
                size_t i = 0, j = 0;
                while (i < as.size() && j < bs.size() && !same_operand) {
                    if (as[i] == bs[j]) same_operand = true;
                    else if (as[i] < bs[j]) ++i;
                    else ++j;
                }
            }
            const int64_t odd = same_operand ? 1 : 0;
            KEEP(Mat, pd, 12);
            KEEP(std::vector<int64_t>, lv2);
            for (size_t i = 0; i < pda.rows(); ++i) memcpy(pd.add(), pda.row(i), 12 * sizeof(uint64_t)), lv2.push_back(lv[i] * 2);
            for (size_t i = 0; i < pdb.rows(); ++i) memcpy(pd.add(), pdb.row(i), 12 * sizeof(uint64_t)), lv2.push_back(lv[i] * 2 + odd);
            const std::vector<int> pidx = order_by(lv2, in->sort_by_weight ? &pd : nullptr, 4);
            out.put(R_PDGRAD, pd, &pidx);
            out.cut(CUT_PDGRAD, permuted(lv2, pidx));
            std::vector<int> widx;
            std::vector<int64_t> jgrp;
            wgrad_jobs(key, dw, db, grp, in->wgrad_chunk, widx, jobs, jgrp);
            out.put(R_WGP, wg, &widx);
            out.put(R_JOBSP, jobs, nullptr);
        }
    }

    // ---- one-channel heads, Same, And / Or ----------------------------------------------------------
    {
        KEEP(Mat, dot, 8);
        KEEP(Mat, same, 10);
        KEEP(Mat, mm, 8);
        KEEP(std::vector<int64_t>, ldot);
        KEEP(std::vector<int64_t>, lsame);
        KEEP(std::vector<int64_t>, lmm);
        for (const Prim& q : prims) {
            const int64_t kind = q.r[C_KIND], t = q.tok;
            if (kind == K_DOT) {
                uint64_t* r = dot.add();
                r[0] = q.a_f, r[1] = P(in->dotw[t]), r[2] = P(in->dotb[t]), r[3] = q.o_f, r[4] = q.o_g, r[5] = q.a_g;
                r[6] = G(in->dotw[t]), r[7] = G(in->dotb[t]);
                ldot.push_back(q.r[C_LEVEL]);
            } else if (kind == K_SAME) {
                uint64_t* r = same.add();
                r[0] = q.a_f, r[1] = q.r[C_BK] == L_ONES ? in->ones : q.b_f;
                r[2] = P(in->dotw[t]), r[3] = P(in->dotb[t]), r[4] = q.o_f, r[5] = q.o_g, r[6] = q.a_g, r[7] = q.b_g;
                r[8] = G(in->dotw[t]), r[9] = G(in->dotb[t]);
                lsame.push_back(q.r[C_LEVEL]);
            } else if (kind == K_MINMAX) {
                uint64_t* r = mm.add();
                r[0] = q.r[C_AK] == L_ONES ? in->ones : q.a_f;
                r[1] = q.r[C_BK] == L_ONES ? in->ones : q.b_f;
                r[2] = q.o_f, r[3] = q.o_g, r[4] = q.a_g, r[5] = q.b_g;
                r[6] = (uint64_t)q.r[C_ACH] | ((uint64_t)q.r[C_BCH] << 32);
                r[7] = (uint64_t)q.r[C_ISMAX];
                lmm.push_back(q.r[C_LEVEL]);
            }
        }
        if (!ldot.empty()) {
            const std::vector<int> idx = order_by(ldot, nullptr, 0);
            out.put(R_DOT, dot, &idx);
            out.cut(CUT_DOT, permuted(ldot, idx));
        }
        if (!lsame.empty()) {
            const std::vector<int> idx = order_by(lsame, nullptr, 0);
            out.put(R_SAME, same, &idx);
            out.cut(CUT_SAME, permuted(lsame, idx));
        }
        if (!lmm.empty()) {
            const std::vector<int> idx = order_by(lmm, nullptr, 0);
            out.put(R_MINMAX, mm, &idx);
            out.cut(CUT_MINMAX, permuted(lmm, idx));
        }
    }
#undef KEEP
    meta[2 + 3 * R_COUNT] = out.n_cuts;
    return out.overflow ? PNMN_EINVAL : 0;
}
