// Trunk planner (host side, no device code): sampled programs -> launched module programs in one library call.
//
// A joint-training step samples its programs on the device, and the NMN launch schedule depends on them: between
// the sampling decode and the first module launch the host has to compile the programs, look up (or build) the
// structure template of each, lay the batch out in the activation arena, write the work-item records of every
// grouped launch, get them to the device and issue ~60 launches.  The reference does the equivalent one example
// and one torch op at a time (probnmn/models/nmn.py:191-241).  Round 2 had the arithmetic of this in C
// (pnmn_compile_programs, pnmn_plan_batch, pnmn_run_launches) but the glue between those calls in Python -- cache
// look-ups per program, numpy marshalling, record packing, a launch list of Python tuples: 1.0-1.2 ms on the
// critical path of a 128-question step.  Here the whole chain is one call:
//
//   token rows --(cache: row bytes)--> compiled programs --(cache: call structure)--> templates
//     --> pnmn_plan_batch, writing straight into a page-locked staging slot --> ONE hipMemcpyAsync into a device
//     buffer the planner owns --> forward launch list issued; backward list handed back for later.
//
// build_template() restates probnmn/runtime/schedule.py:build_template (checked against it row for row by
// tests/test_trunk_planner.py, as pnmn_plan_batch is against the numpy planner).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/probnmn_hip.h"
#include "conv_plan.h"

namespace {

// ---- keep in sync with schedule.py / host_plan.hip / program_compiler.py -------------------------------------
enum { C_KIND, C_LEVEL, C_CALL, C_WIDX, C_DIL, C_AK, C_AO, C_BK, C_BO, C_OK, C_OO, C_ACH, C_BCH, C_ISMAX,
       C_MASKED, C_SCRATCH, C_PA, C_PB, NCOLS };
enum { L_SLOT, L_FEAT, L_ONES, L_FINAL };
enum { K_CONV, K_PROJ, K_DOT, K_SAME, K_MINMAX };
enum { SKIP = 0, SCENE, AND, OR, CMP, ATT, QUERY, REL, SAME };
enum { R_CONV, R_DGRAD, R_WG3, R_JOBS3, R_PROJ, R_PDGRAD, R_WGP, R_JOBSP, R_DOT, R_SAME, R_MINMAX, R_MASKBWD, R_COUNT };
enum { CUT_CONV, CUT_PROJ, CUT_DOT, CUT_SAME, CUT_MINMAX, CUT_DGRAD, CUT_MASKBWD, CUT_PDGRAD, CUT_WGROUP, CUT_KINDS };
constexpr int FEAT = 0, ONES = 1;
constexpr int64_t HW_ALIGN = 64;  // floats; every arena slot is 256-byte aligned
const int RELATE_DILATIONS[5] = {1, 2, 4, 8, 1};  // reference nmn_modules.py:146-150
// bytes per record of each kind (pnmn_conv_item 96, wgrad_item 48, wgrad_job 24, dot1 64, same 80, minmax 64, maskbwd 40)
const int RECORD_WORDS[R_COUNT] = {12, 12, 6, 3, 12, 12, 6, 3, 8, 10, 8, 5};

inline int64_t align_up(int64_t n) { return (n + HW_ALIGN - 1) / HW_ALIGN * HW_ALIGN; }

struct Template {
    std::vector<int64_t> table;  // [n_prims][NCOLS]
    int n_prims = 0, depth = 0;
    int64_t size = 0;            // arena floats per example (values + backward scratch)
    bool result_is_feat = false;
};

struct Program {
    bool valid = false;
    int tid = -1;
    std::vector<int64_t> tokens;  // token of every call
    uint64_t reach[4] = {0, 0, 0, 0};  // tokens of the calls the result depends on (the template's primitives)
};

struct Call {
    int kind, tok, a, b, a_ch, b_ch, out_ch;
};

// schedule.py:build_template
Template build_template(const std::vector<Call>& calls, int result, int hw, int channels) {
    const int64_t big = align_up((int64_t)hw * channels), small = align_up(hw);
    const int n = (int)calls.size();
    std::vector<char> needed(n, 0);
    {   // liveness: only calls that reach the result are executed
        std::vector<int> stack{result};
        while (!stack.empty()) {
            const int v = stack.back();
            stack.pop_back();
            if (v >= 2 && !needed[v - 2]) {
                needed[v - 2] = 1;
                stack.push_back(calls[v - 2].a);
                stack.push_back(calls[v - 2].b);
            }
        }
    }
    int64_t cursor = 0;
    auto alloc = [&](int64_t k) {
        const int64_t off = cursor;
        cursor += k;
        return off;
    };
    struct Loc {
        int64_t kind, off;
    };
    std::vector<Loc> loc(n + 2);
    std::vector<int> lvl(n + 2, 0), prod(n + 2, -1);
    loc[FEAT] = Loc{L_FEAT, 0};
    loc[ONES] = Loc{L_ONES, 0};
    Template t;
    auto prim = [&](int kind, int level, int call, int widx, int dil, Loc a, Loc b, Loc out, int a_ch, int b_ch, int is_max,
                    int masked, int pa, int pb) {
        const int64_t row[NCOLS] = {kind, level, call, widx, dil, a.kind, a.off, b.kind, b.off, out.kind, out.off,
                                    a_ch, b_ch, is_max, masked, -1, pa, pb};
        t.table.insert(t.table.end(), row, row + NCOLS);
        return t.n_prims++;
    };
    const Loc ones{L_ONES, 0}, feat{L_FEAT, 0};
    for (int ci = 0; ci < n; ++ci) {
        if (!needed[ci]) continue;
        const Call& c = calls[ci];
        const int vid = ci + 2;
        const bool is_result = vid == result;
        Loc out;
        if (c.out_ch == channels)
            out = is_result ? Loc{L_FINAL, 0} : Loc{L_SLOT, alloc(big)};
        else
            out = Loc{L_SLOT, alloc(small)};
        int level, last;
        if (c.kind == AND || c.kind == OR) {
            level = std::max(lvl[c.a], lvl[c.b]) + 1;
            last = prim(K_MINMAX, level, ci, 0, 1, loc[c.a], loc[c.b], out, c.a_ch, c.b_ch, c.kind == OR, 0, prod[c.a], prod[c.b]);
        } else if (c.kind == SAME) {
            level = lvl[c.a] + 1;
            last = prim(K_SAME, level, ci, 0, 1, feat, loc[c.a], out, 0, 0, 0, 0, -1, prod[c.a]);
        } else if (c.kind == CMP) {
            level = std::max(lvl[c.a], lvl[c.b]) + 1;
            const Loc t0{L_SLOT, alloc(big)}, t1{L_SLOT, alloc(big)};
            const int j0 = prim(K_PROJ, level, ci, 0, 1, loc[c.a], loc[c.b], t0, 0, 0, 0, 0, prod[c.a], prod[c.b]);
            const int j1 = prim(K_CONV, level + 1, ci, 1, 1, t0, ones, t1, 0, 0, 0, 0, j0, -1);
            last = prim(K_CONV, level + 2, ci, 2, 1, t1, ones, out, 0, 0, 0, 0, j1, -1);
            level += 2;
        } else {  // ATT / QUERY / REL
            const int nconv = c.kind == REL ? 5 : 2;
            level = lvl[c.a];
            Loc src = feat;
            last = -1;
            for (int k = 0; k < nconv; ++k) {
                ++level;
                const bool last_is_out = (k == nconv - 1) && c.kind == QUERY;
                const Loc dst = last_is_out ? out : Loc{L_SLOT, alloc(big)};
                const int dil = c.kind == REL ? RELATE_DILATIONS[k] : 1;
                if (k == 0)  // input is FEAT * attention (L_ONES -> no multiply)
                    last = prim(K_CONV, level, ci, 1, dil, src, loc[c.a], dst, 0, 0, 0, 1, -1, prod[c.a]);
                else
                    last = prim(K_CONV, level, ci, k + 1, dil, src, ones, dst, 0, 0, 0, 0, last, -1);
                src = dst;
            }
            if (c.kind != QUERY) {
                ++level;
                last = prim(K_DOT, level, ci, 0, 1, src, ones, out, 0, 0, 0, 0, last, -1);
            }
        }
        loc[vid] = out;
        lvl[vid] = level;
        prod[vid] = last;
    }
    for (int p = 0; p < t.n_prims; ++p) {  // backward scratch: gradient wrt (FEAT * attention) of each masked conv
        int64_t* r = t.table.data() + (size_t)p * NCOLS;
        if (r[C_MASKED]) r[C_SCRATCH] = alloc(big);
        t.depth = std::max<int>(t.depth, (int)r[C_LEVEL]);
    }
    t.size = cursor;
    t.result_is_feat = result < 2;
    return t;
}

struct Planner {
    // configuration
    std::vector<int32_t> kinds;
    std::vector<int64_t> w3, b3, wt3, dotw, dotb;
    int channels = 0, H = 0, W = 0, hw = 0;
    int wgrad_chunk = 8, wgrad_groups = 1, fuse_mask_bwd = 1, sole_writer = 1, sort_by_weight = 1;
    // caches
    std::unordered_map<std::string, int> program_ids, template_ids;
    std::vector<Program> programs;
    std::vector<Template> templates;
    std::vector<int64_t> bank, bank_nprims;  // dense [T][pmax][NCOLS] copy of the templates for pnmn_plan_batch
    int pmax = 0;
    bool bank_dirty = true;
    // staging: page-locked host slots (a slot is reused once the copy that read it has completed) and one device buffer
    struct Slot {
        void* host = nullptr;
        size_t capacity = 0;
        hipEvent_t copied = nullptr;
        bool in_flight = false;
    } slots[4];
    int next_slot = 0;
    void* dev = nullptr;
    size_t dev_capacity = 0;
    hipStream_t dev_stream = nullptr;
    bool dev_used = false;
    std::vector<uint64_t> host_only;  // launch == 0 on a box without a GPU: the records stay here
    // last lists
    std::vector<pnmn_launch> fwd;
    // per-call scratch kept between calls
    std::vector<int64_t> tids, examples, base, tokens;
    std::vector<int32_t> cuts;
    std::vector<int32_t> n_calls, result, calls;
    std::vector<uint8_t> cvalid;
    std::vector<int> miss;
    std::vector<int64_t> miss_rows;
};

void rebuild_bank(Planner& P) {
    P.pmax = 1;
    for (const Template& t : P.templates) P.pmax = std::max(P.pmax, t.n_prims);
    P.bank.assign(P.templates.size() * (size_t)P.pmax * NCOLS, 0);
    P.bank_nprims.resize(P.templates.size());
    for (size_t i = 0; i < P.templates.size(); ++i) {
        const Template& t = P.templates[i];
        std::copy(t.table.begin(), t.table.end(), P.bank.begin() + i * (size_t)P.pmax * NCOLS);
        P.bank_nprims[i] = t.n_prims;
    }
    P.bank_dirty = false;
}

// programs not seen before: compile them together, file them (and their templates) in the caches
int compile_new(Planner& P, const int64_t* rows, int length, const std::vector<int>& which, std::vector<int>& ids) {
    const int n = (int)which.size();
    P.miss_rows.resize((size_t)n * std::max(length, 1));
    for (int j = 0; j < n; ++j)
        memcpy(P.miss_rows.data() + (size_t)j * length, rows + (size_t)which[j] * length, sizeof(int64_t) * length);
    P.cvalid.resize(n);
    P.n_calls.resize(n);
    P.result.resize(n);
    P.calls.resize((size_t)n * std::max(length, 1) * 7);
    const int rc = pnmn_compile_programs(P.miss_rows.data(), n, length, P.kinds.data(), (int)P.kinds.size(), P.channels,
                                         P.cvalid.data(), P.n_calls.data(), P.calls.data(), P.result.data());
    if (rc != 0) return rc;
    for (int j = 0; j < n; ++j) {
        const std::string key(reinterpret_cast<const char*>(rows + (size_t)which[j] * length), sizeof(int64_t) * length);
        auto hit = P.program_ids.find(key);  // (the same new program may occur several times in the batch)
        if (hit != P.program_ids.end()) {
            ids[which[j]] = hit->second;
            continue;
        }
        Program prog;
        prog.valid = P.cvalid[j] != 0;
        if (prog.valid) {
            const int nc = P.n_calls[j];
            const int32_t* c = P.calls.data() + (size_t)j * length * 7;
            std::vector<Call> calls(nc);
            std::string skey;  // the call structure: everything but the tokens, plus the result
            skey.reserve((size_t)nc * 24 + 4);
            for (int k = 0; k < nc; ++k) {
                calls[k] = Call{c[k * 7 + 0], c[k * 7 + 1], c[k * 7 + 2], c[k * 7 + 3], c[k * 7 + 4], c[k * 7 + 5], c[k * 7 + 6]};
                prog.tokens.push_back(c[k * 7 + 1]);
                const int32_t s[6] = {c[k * 7 + 0], c[k * 7 + 2], c[k * 7 + 3], c[k * 7 + 4], c[k * 7 + 5], c[k * 7 + 6]};
                skey.append(reinterpret_cast<const char*>(s), sizeof(s));
            }
            const int32_t res = P.result[j];
            skey.append(reinterpret_cast<const char*>(&res), sizeof(res));
            auto th = P.template_ids.find(skey);
            if (th == P.template_ids.end()) {
                P.templates.push_back(build_template(calls, res, P.hw, P.channels));
                th = P.template_ids.emplace(skey, (int)P.templates.size() - 1).first;
                P.bank_dirty = true;
            }
            prog.tid = th->second;
            const Template& tpl = P.templates[prog.tid];
            for (int q = 0; q < tpl.n_prims; ++q) {
                const int64_t call = tpl.table[(size_t)q * NCOLS + 2];
                if (call >= 0 && call < (int64_t)prog.tokens.size()) {
                    const int64_t tok = prog.tokens[call];
                    if (tok >= 0 && tok < 256) prog.reach[tok >> 6] |= 1ull << (tok & 63);
                }
            }
        }
        if (P.program_ids.size() > 500000) {  // bounded: sampled programs keep arriving for a whole training run
            P.program_ids.clear();
            P.programs.clear();
        }
        P.programs.push_back(std::move(prog));
        const int id = (int)P.programs.size() - 1;
        P.program_ids.emplace(key, id);
        ids[which[j]] = id;
    }
    return 0;
}

inline pnmn_launch make_launch(int op, int n, const void* a, const void* b, const void* c, std::initializer_list<int> p) {
    pnmn_launch l;
    memset(&l, 0, sizeof(l));
    l.a = a, l.b = b, l.c = c, l.op = op, l.n = n;
    int i = 0;
    for (int v : p) l.p[i++] = v;
    return l;
}

hipError_t ensure_staging(Planner& P, size_t bytes, hipStream_t stream, Planner::Slot** out) {
    Planner::Slot& s = P.slots[P.next_slot];
    P.next_slot = (P.next_slot + 1) % 4;
    hipError_t e;
    if (s.in_flight) {
        if ((e = hipEventSynchronize(s.copied)) != hipSuccess) return e;
        s.in_flight = false;
    }
    if (s.capacity < bytes) {
        // a hipHostMalloc costs milliseconds: slots start at 4 MB (several times the list of a 1024-question step)
        // and double when a list outgrows them
        size_t cap = s.capacity ? s.capacity : (size_t)4 << 20;
        while (cap < bytes) cap *= 2;
        if (s.host && (e = hipHostFree(s.host)) != hipSuccess) return e;
        s.host = nullptr, s.capacity = 0;
        if ((e = hipHostMalloc(&s.host, cap, hipHostMallocDefault)) != hipSuccess) return e;
        s.capacity = cap;
    }
    if (!s.copied && (e = hipEventCreateWithFlags(&s.copied, hipEventDisableTiming)) != hipSuccess) return e;
    *out = &s;
    return hipSuccess;
}

hipError_t ensure_device(Planner& P, size_t bytes, hipStream_t stream) {
    hipError_t e;
    // The device buffer is rewritten in stream order behind its last readers (this step's forward AND backward
    // launches were queued on the stream the previous call was given).  Another stream this time: wait for the device.
    if (P.dev_used && P.dev_stream != stream && (e = hipDeviceSynchronize()) != hipSuccess) return e;
    if (P.dev_capacity < bytes) {
        size_t cap = P.dev_capacity ? P.dev_capacity : (size_t)4 << 20;
        while (cap < bytes) cap *= 2;
        if (P.dev) {
            if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
            if ((e = hipFree(P.dev)) != hipSuccess) return e;
            P.dev = nullptr, P.dev_capacity = 0;
        }
        if ((e = hipMalloc(&P.dev, cap)) != hipSuccess) return e;
        P.dev_capacity = cap;
    }
    P.dev_stream = stream;
    P.dev_used = true;
    return hipSuccess;
}

}  // namespace

extern "C" {

int pnmn_trunk_planner_create(const pnmn_trunk_config* c, void** planner) {
    if (!c || !planner || !c->kinds || !c->w3 || !c->b3 || !c->wt3 || !c->dotw || !c->dotb || c->n_kinds <= 0 ||
        c->channels <= 0 || c->H <= 0 || c->W <= 0 || c->wgrad_chunk < 1)
        return PNMN_EINVAL;
    Planner* P = new Planner();
    const size_t V = (size_t)c->n_kinds;
    P->kinds.assign(c->kinds, c->kinds + V);
    P->w3.assign(c->w3, c->w3 + V * 6);
    P->b3.assign(c->b3, c->b3 + V * 6);
    P->wt3.assign(c->wt3, c->wt3 + V * 6);
    P->dotw.assign(c->dotw, c->dotw + V);
    P->dotb.assign(c->dotb, c->dotb + V);
    P->channels = c->channels, P->H = c->H, P->W = c->W, P->hw = c->H * c->W;
    P->wgrad_chunk = c->wgrad_chunk, P->wgrad_groups = std::max(1, c->wgrad_groups);
    P->fuse_mask_bwd = c->fuse_mask_bwd, P->sole_writer = c->sole_writer, P->sort_by_weight = c->sort_by_weight;
    *planner = P;
    return 0;
}

int pnmn_trunk_planner_destroy(void* planner) {
    Planner* P = static_cast<Planner*>(planner);
    if (!P) return 0;
    for (auto& s : P->slots) {
        if (s.in_flight) (void)hipEventSynchronize(s.copied);
        if (s.host) (void)hipHostFree(s.host);
        if (s.copied) (void)hipEventDestroy(s.copied);
    }
    if (P->dev) (void)hipFree(P->dev);
    delete P;
    return 0;
}

int pnmn_trunk_last_forward(void* planner, pnmn_launch* out, int capacity) {
    Planner* P = static_cast<Planner*>(planner);
    if (!P || (!out && capacity > 0)) return PNMN_EINVAL;
    const int n = (int)P->fwd.size();
    for (int i = 0; i < n && i < capacity; ++i) out[i] = P->fwd[i];
    return n;
}

int64_t pnmn_trunk_last_records_bytes(void* planner, uint64_t* out, int64_t capacity_words) {
    Planner* P = static_cast<Planner*>(planner);
    if (!P || (!out && capacity_words > 0)) return PNMN_EINVAL;
    const int64_t n = (int64_t)P->host_only.size();
    for (int64_t i = 0; i < n && i < capacity_words; ++i) out[i] = P->host_only[i];
    return n * 8;
}

int pnmn_trunk_plan_and_launch(void* planner, pnmn_trunk_io* io, void* stream_) {
    Planner* Pp = static_cast<Planner*>(planner);
    if (!Pp || !io || io->n_programs < 0 || io->length < 0 || (io->n_programs > 0 && (!io->programs || !io->valid)))
        return PNMN_EINVAL;
    if ((io->n_fwd_tail > 0 && !io->fwd_tail) || (io->n_bwd_head > 0 && !io->bwd_head) || (io->n_bwd_tail > 0 && !io->bwd_tail))
        return PNMN_EINVAL;
    Planner& P = *Pp;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int B = io->n_programs, length = io->length;
    const int HW = P.hw, C = P.channels, H = P.H, W = P.W;
    const int64_t map_bytes = (int64_t)HW * C * 4;

    // ---- compile (cache by token row) ---------------------------------------------------------------------------
    std::vector<int> ids(B, -1);
    P.miss.clear();
    for (int i = 0; i < B; ++i) {
        const std::string key(reinterpret_cast<const char*>(io->programs + (size_t)i * length), sizeof(int64_t) * length);
        auto hit = P.program_ids.find(key);
        if (hit != P.program_ids.end())
            ids[i] = hit->second;
        else
            P.miss.push_back(i);
    }
    if (!P.miss.empty()) {
        const int rc = compile_new(P, io->programs, length, P.miss, ids);
        if (rc != 0) return rc;
    }
    if (P.bank_dirty) rebuild_bank(P);

    // ---- the batch: valid examples, templates, arena blocks, call tokens ----------------------------------------
    P.tids.clear(), P.examples.clear(), P.base.clear();
    int cmax = 1;
    int64_t arena = 0, n_total = 0;
    io->n_invalid = 0, io->n_feat_result = 0;
    io->touched_tokens[0] = io->touched_tokens[1] = io->touched_tokens[2] = io->touched_tokens[3] = 0;
    for (int i = 0; i < B; ++i) {
        const Program& pr = P.programs[ids[i]];
        io->valid[i] = pr.valid ? 1 : 0;
        if (!pr.valid) {
            ++io->n_invalid;
            continue;
        }
        const Template& t = P.templates[pr.tid];
        for (int w = 0; w < 4; ++w) io->touched_tokens[w] |= pr.reach[w];
        P.tids.push_back(pr.tid);
        P.examples.push_back(i);
        P.base.push_back(arena);
        arena += t.size;
        n_total += t.n_prims;
        cmax = std::max<int>(cmax, (int)pr.tokens.size());
        if (t.result_is_feat) ++io->n_feat_result;
    }
    const int nv = (int)P.tids.size();
    io->arena_floats = arena;
    io->n_prims = (int)n_total;
    io->n_bwd = 0, io->bwd_piece_cut = -1, io->n_fwd = 0, io->depth = 0;
    if (arena > io->act_capacity) return PNMN_EAGAIN;
    P.tokens.assign((size_t)nv * cmax, 0);
    for (int v = 0; v < nv; ++v) {
        const Program& pr = P.programs[ids[P.examples[v]]];
        std::copy(pr.tokens.begin(), pr.tokens.end(), P.tokens.begin() + (size_t)v * cmax);
    }

    // ---- records: planned straight into the staging slot --------------------------------------------------------
    const size_t n_rows = (size_t)io->n_invalid + 2 * (size_t)io->n_feat_result;  // SET_ROWS / ACCUMULATE items
    const size_t row_words = n_rows * 3;
    const size_t plan_words = (size_t)n_total * 48 + 64;
    const size_t total_bytes = (row_words + plan_words) * 8;
    uint64_t* words = nullptr;
    Planner::Slot* slot = nullptr;
    const bool on_device = io->launch != 0;
    if (on_device) {
        hipError_t e = ensure_staging(P, total_bytes, stream, &slot);
        if (e != hipSuccess) return (int)e;
        if ((e = ensure_device(P, total_bytes, stream)) != hipSuccess) return (int)e;
        words = static_cast<uint64_t*>(slot->host);
    } else {
        P.host_only.resize((total_bytes + 7) / 8);
        words = P.host_only.data();
    }
    const uint64_t dev_base = on_device ? reinterpret_cast<uint64_t>(P.dev) : 0x10000;

    // rows first: [invalid -> zero final row] [feat result -> final row = feat row] | [feat result -> gfeat row += gfinal row]
    {
        uint64_t* r = words;
        for (int i = 0; i < B; ++i)
            if (!io->valid[i]) r[0] = 0, r[1] = io->final_ + (uint64_t)(map_bytes * i), r[2] = (uint64_t)HW * C, r += 3;
        for (int v = 0; v < nv; ++v)
            if (P.templates[P.tids[v]].result_is_feat) {
                const uint64_t off = (uint64_t)(map_bytes * P.examples[v]);
                r[0] = io->feat + off, r[1] = io->final_ + off, r[2] = (uint64_t)HW * C, r += 3;
            }
        for (int v = 0; v < nv; ++v)
            if (P.templates[P.tids[v]].result_is_feat) {
                const uint64_t off = (uint64_t)(map_bytes * P.examples[v]);
                r[0] = io->gfinal + off, r[1] = io->gfeat + off, r[2] = (uint64_t)HW * C, r += 3;
            }
    }
    int64_t meta[2 + 3 * R_COUNT + 1];
    memset(meta, 0, sizeof(meta));
    P.cuts.resize(4 * 4096);
    int n_cuts = 0;
    if (nv > 0) {
        pnmn_plan_in in;
        memset(&in, 0, sizeof(in));
        in.tables = P.bank.data(), in.nprims = P.bank_nprims.data(), in.tids = P.tids.data(), in.examples = P.examples.data();
        in.base = P.base.data(), in.tokens = P.tokens.data();
        in.w3 = P.w3.data(), in.b3 = P.b3.data(), in.wt3 = P.wt3.data(), in.dotw = P.dotw.data(), in.dotb = P.dotb.data();
        in.params = io->params, in.grads = io->grads, in.wt = io->wt, in.act = io->act, in.gact = io->gact;
        in.feat = io->feat, in.gfeat = io->gfeat, in.final_ = io->final_, in.gfinal = io->gfinal, in.ones = io->ones;
        in.n_templates = (int)P.templates.size(), in.pmax = P.pmax, in.nv = nv, in.cmax = cmax, in.hw = HW, in.channels = C;
        // items per weight-gradient job: a job walks its items one after the other, so a small batch (few items per
        // weight) is cut finer to put more workgroups on the chip -- 128 questions: 8 -> 3 items per job
        in.wgrad_chunk = std::min<int>(P.wgrad_chunk, std::max<int>(2, (int)(n_total / 256)));
        in.wgrad_groups = P.wgrad_groups, in.fuse_mask_bwd = P.fuse_mask_bwd;
        in.sole_writer = P.sole_writer, in.sort_by_weight = P.sort_by_weight;
        const int rc = pnmn_plan_batch(&in, words + row_words, (int64_t)plan_words, meta, P.cuts.data(), 4096);
        if (rc != 0) return rc;
        n_cuts = (int)meta[2 + 3 * R_COUNT];
    }
    io->depth = (int)meta[1];
    io->n_conv = (int)meta[3 + 3 * R_CONV], io->n_proj = (int)meta[3 + 3 * R_PROJ];
    size_t used_words = row_words;
    for (int k = 0; k < R_COUNT; ++k) used_words = std::max(used_words, row_words + (size_t)(meta[2 + 3 * k] + meta[3 + 3 * k] * meta[4 + 3 * k]));
    // device address of record `i` of kind `k`
    auto rec = [&](int k, int64_t i) -> const void* {
        return reinterpret_cast<const void*>(dev_base + (row_words + (uint64_t)meta[2 + 3 * k] + (uint64_t)i * RECORD_WORDS[k]) * 8);
    };
    auto rows_at = [&](size_t i) -> const void* { return reinterpret_cast<const void*>(dev_base + i * 24); };

    // ---- launch lists (schedule.py:_order, engine.py:_run_forward_launches / _queue_backward) ------------------
    struct Cut {
        int b, e;
    };
    std::vector<std::vector<Cut>> at[CUT_KINDS];  // at[kind][level] -> cuts (pdgrad: up to two per level)
    const int depth = io->depth;
    for (auto& a : at) a.assign(depth + 2, {});
    int n_jobs3 = (int)meta[3 + 3 * R_JOBS3], n_jobsp = (int)meta[3 + 3 * R_JOBSP];
    for (int i = 0; i < n_cuts; ++i) {
        const int32_t* c = P.cuts.data() + 4 * i;
        if (c[0] == CUT_WGROUP) continue;  // (the deferred weight gradients go out as one launch)
        const int lv = c[0] == CUT_PDGRAD ? c[1] / 2 : c[1];
        if (c[0] < 0 || c[0] >= CUT_KINDS || lv < 0 || lv > depth) return PNMN_EINVAL;
        at[c[0]][lv].push_back(Cut{c[2], c[3]});
    }
    const void* conv_cus = reinterpret_cast<const void*>((uintptr_t)(io->conv_cus > 0 ? io->conv_cus : 0));

    if (on_device && used_words) {
        hipError_t e = hipMemcpyAsync(P.dev, words, used_words * 8, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return (int)e;
        if ((e = hipEventRecord(slot->copied, stream)) != hipSuccess) return (int)e;
        slot->in_flight = true;
    }
    P.fwd.clear();
    const size_t n_zero = (size_t)io->n_invalid, n_copy = (size_t)io->n_feat_result;
    if (n_zero + n_copy) P.fwd.push_back(make_launch(PNMN_OP_SET_ROWS, (int)(n_zero + n_copy), rows_at(0), nullptr, nullptr, {}));
    for (int lv = 1; lv <= depth; ++lv) {
        for (const Cut& c : at[CUT_MINMAX][lv])
            P.fwd.push_back(make_launch(PNMN_OP_MINMAX_FWD, c.e - c.b, rec(R_MINMAX, c.b), nullptr, nullptr, {HW, C}));
        for (const Cut& c : at[CUT_SAME][lv])
            P.fwd.push_back(make_launch(PNMN_OP_SAME_FWD, c.e - c.b, rec(R_SAME, c.b), nullptr, nullptr, {HW}));
        for (const Cut& c : at[CUT_DOT][lv])
            P.fwd.push_back(make_launch(PNMN_OP_DOT_FWD, c.e - c.b, rec(R_DOT, c.b), nullptr, nullptr, {HW}));
        for (const Cut& c : at[CUT_PROJ][lv])
            P.fwd.push_back(make_launch(PNMN_OP_CONV, c.e - c.b, rec(R_PROJ, c.b), nullptr, conv_cus, {H, W, 2, 1, C, C, 1, 1}));
        for (const Cut& c : at[CUT_CONV][lv])
            P.fwd.push_back(make_launch(PNMN_OP_CONV, c.e - c.b, rec(R_CONV, c.b), nullptr, conv_cus, {H, W, 1, 9, C, C, 1, 1}));
    }
    for (int i = 0; i < io->n_fwd_tail; ++i) P.fwd.push_back(io->fwd_tail[i]);
    io->n_fwd = (int)P.fwd.size();

    if (io->need_backward) {
        std::vector<pnmn_launch> bwd;
        for (int i = 0; i < io->n_bwd_head; ++i) bwd.push_back(io->bwd_head[i]);
        if (arena > 0)
            bwd.push_back(make_launch(PNMN_OP_ZERO, 0, reinterpret_cast<const void*>(io->gact),
                                      reinterpret_cast<const void*>((uintptr_t)(arena * 4)), nullptr, {}));
        if (n_copy) bwd.push_back(make_launch(PNMN_OP_ACCUMULATE, (int)n_copy, rows_at(n_zero + n_copy), nullptr, nullptr, {}));
        for (int lv = depth; lv >= 1; --lv) {
            for (const Cut& c : at[CUT_MINMAX][lv])
                bwd.push_back(make_launch(PNMN_OP_MINMAX_BWD, c.e - c.b, rec(R_MINMAX, c.b), nullptr, nullptr, {HW, C}));
            for (const Cut& c : at[CUT_SAME][lv])
                bwd.push_back(make_launch(PNMN_OP_SAME_BWD, c.e - c.b, rec(R_SAME, c.b), nullptr, nullptr, {HW}));
            for (const Cut& c : at[CUT_DOT][lv])
                bwd.push_back(make_launch(PNMN_OP_DOT_BWD, c.e - c.b, rec(R_DOT, c.b), nullptr, nullptr, {HW}));
            for (const Cut& c : at[CUT_PDGRAD][lv])
                bwd.push_back(make_launch(PNMN_OP_CONV, c.e - c.b, rec(R_PDGRAD, c.b), nullptr, conv_cus, {H, W, 1, 1, C, C, 1, 0}));
            for (const Cut& c : at[CUT_DGRAD][lv])
                bwd.push_back(make_launch(PNMN_OP_CONV, c.e - c.b, rec(R_DGRAD, c.b), nullptr, conv_cus, {H, W, 1, 9, C, C, 1, 0}));
            for (const Cut& c : at[CUT_MASKBWD][lv])
                bwd.push_back(make_launch(PNMN_OP_MASK_BWD, c.e - c.b, rec(R_MASKBWD, c.b), nullptr, nullptr, {HW}));
        }
        // deferred d(feats) of the masked convs (fuse_mask_bwd == 2): one gather over all of them
        if (P.fuse_mask_bwd == 2 && meta[3 + 3 * R_MASKBWD] > 0)
            bwd.push_back(make_launch(PNMN_OP_FEAT_GATHER, (int)meta[3 + 3 * R_MASKBWD], rec(R_MASKBWD, 0),
                                      reinterpret_cast<const void*>(io->gfeat), nullptr, {B, HW}));
        // every weight gradient of the module convs in ONE grouped launch: all (example, conv) pairs that share a
        // weight are contracted by the same workgroups
        if (n_jobs3)
            bwd.push_back(make_launch(PNMN_OP_WGRAD, n_jobs3, rec(R_WG3, 0), rec(R_JOBS3, 0), nullptr, {H, W, 9, 1, 1, C, C, io->wgrad_cus}));
        if (n_jobsp)
            bwd.push_back(make_launch(PNMN_OP_WGRAD, n_jobsp, rec(R_WGP, 0), rec(R_JOBSP, 0), nullptr, {H, W, 1, 2, 1, C, C, io->wgrad_cus}));
        io->bwd_piece_cut = (int)bwd.size();
        for (int i = 0; i < io->n_bwd_tail; ++i) bwd.push_back(io->bwd_tail[i]);
        io->n_bwd = (int)bwd.size();
        if (io->n_bwd > io->bwd_capacity || (io->n_bwd > 0 && !io->bwd)) return PNMN_EINVAL;
        std::copy(bwd.begin(), bwd.end(), io->bwd);
    }
    if (on_device && !P.fwd.empty()) return pnmn_run_launches(P.fwd.data(), (int)P.fwd.size(), stream_);
    return 0;
}

}  // extern "C"
