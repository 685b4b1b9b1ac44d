"""Fixed cost of a recurrent launch: launch time at two sequence lengths -> per-step slope and intercept
(LSTM layer from the token table, teacher-forced decoder; forward, no autograd).   usage: python scripts/r05_recurrent_fixed.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "probnmn-clevr_amd")]
import torch
from probnmn.modules.seq2seq_base import _LSTMLayerSeq, _AttnLSTMDecoder, pack_fragments
dev = torch.device("cuda:0")
H = 256


def clock(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


r = lambda *s, scale=1.0: torch.randn(*s, device=dev) * scale  # noqa: E731
with torch.no_grad():
    for B in (128, 512):
        res = {}
        for T in (6, 46):
            V = 93
            table, w = r(V, 4 * H, scale=0.5), r(4 * H, H, scale=0.05)
            wp = pack_fragments(w)
            tok = torch.randint(0, V, (B, T), device=dev)
            res[("lstm", T)] = clock(lambda: _LSTMLayerSeq.apply(table, w, wp, None, tok))
            S, V = 27, 96
            enc, h0, mask = r(B, S, H), r(B, H), torch.ones(B, S, device=dev)
            w_c, w_hh, w_p, b_p = r(4 * H, H, scale=0.05), r(4 * H, H, scale=0.05), r(V, H, scale=0.3), r(V)
            packs = (pack_fragments(w_c), pack_fragments(w_hh), None, None)
            etable = r(V, 4 * H, scale=0.5)
            teacher = torch.randint(0, V, (B, T), device=dev)
            res[("decoder", T)] = clock(lambda: _AttnLSTMDecoder.apply(None, etable, enc, mask, h0, w_c, w_hh, w_p, b_p, 0, T, 5, 0, 0, 1, 2, packs, teacher))
        for k in ("lstm", "decoder"):
            slope = (res[(k, 46)] - res[(k, 6)]) / 40
            print("B=%4d %-8s T=6: %7.1f us  T=46: %7.1f us  -> %.2f us per step + %.1f us per launch"
                  % (B, k, res[(k, 6)], res[(k, 46)], slope, res[(k, 6)] - 6 * slope))
