// Host-side batch compiler for prefix CLEVR programs (no device work).
//
// The per-program state machine of probnmn/runtime/program_compiler.py (`ProgramCompiler._compile`,
// which restates the validity rules of the reference's interpreter, nmn.py:191-238 / SURVEY App. C)
// for a whole batch in one call: in joint training every step samples a few hundred programs the
// compiler has not seen before, and the Python loop over them (~10 us each) sat on the critical path
// between the sampling decode and the first NMN launch.
#include <stddef.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"

namespace {
// token kinds and value ids: keep in sync with program_compiler.py
enum { SKIP = 0, SCENE, AND, OR, CMP, ATT, QUERY, REL, SAME };
constexpr int FEAT = 0, ONES = 1;
}  // namespace

extern "C" int pnmn_compile_programs(const int64_t* tokens, int n_programs, int length, const int32_t* kinds,
                                     int n_kinds, int channels, uint8_t* valid, int32_t* n_calls, int32_t* calls,
                                     int32_t* result) {
    if (n_programs <= 0) return 0;
    if (!tokens || !kinds || !valid || !n_calls || !calls || !result || length < 0 || n_kinds <= 0) return PNMN_EINVAL;
    const int D = channels;
    for (int p = 0; p < n_programs; ++p) {
        const int64_t* row = tokens + (size_t)p * length;
        int32_t* out_calls = calls + (size_t)p * length * 7;
        int out = FEAT, out_c = D, saved = -1, saved_c = 0, n = 0;
        bool ok = true;
        for (int t = length - 1; t >= 0 && ok; --t) {  // the reference walks the sequence right to left
            const int64_t tok = row[t];
            if (tok < 0 || tok >= n_kinds) {  // the reference's vocabulary lookup would raise KeyError
                ok = false;
                break;
            }
            const int kind = kinds[tok];
            if (kind == SKIP) continue;
            if (kind == SCENE) {
                saved = out;
                saved_c = out_c;
                out = ONES;
                out_c = 1;
                continue;
            }
            int a, b, ca, cb, oc;
            if (kind == AND || kind == OR) {
                if (saved < 0) { ok = false; break; }
                a = out, b = saved, ca = out_c, cb = saved_c;
                oc = out_c > saved_c ? out_c : saved_c;  // min/max broadcast 1 <-> D channels
            } else if (kind == CMP) {
                if (saved < 0 || out_c != D || saved_c != D) { ok = false; break; }
                a = out, b = saved, ca = D, cb = D, oc = D;
            } else {  // ATT / QUERY / REL / SAME take (FEAT, attention)
                if (out_c != 1) { ok = false; break; }
                a = out, b = FEAT, ca = 1, cb = D, oc = (kind == QUERY) ? D : 1;
            }
            int32_t* c = out_calls + (size_t)n * 7;
            c[0] = kind, c[1] = (int32_t)tok, c[2] = a, c[3] = b, c[4] = ca, c[5] = cb, c[6] = oc;
            ++n;
            out = n + 1;
            out_c = oc;
        }
        if (ok && out_c != D) ok = false;
        valid[p] = ok ? 1 : 0;
        n_calls[p] = ok ? n : 0;
        result[p] = ok ? out : FEAT;
    }
    return 0;
}
