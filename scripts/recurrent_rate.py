"""Forward launch time per step of the persistent recurrent kernels by batch size: an LSTM layer (input projection
given per (row, step) / looked up in the per-token table) and the attention decoder (teacher forced from the table,
sampling).  usage: python scripts/recurrent_rate.py"""
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
    for B in (128, 512, 1024):
        T, V = 46, 93
        xp, table, w = r(B, T, 4 * H, scale=0.5), r(V, 4 * H, scale=0.5), r(4 * H, H, scale=0.05)
        wp = pack_fragments(w)
        tok = torch.randint(0, V, (B, T), device=dev)
        a = clock(lambda: _LSTMLayerSeq.apply(xp, w, wp, None))
        b = clock(lambda: _LSTMLayerSeq.apply(table, w, wp, None, tok))
        out = "B=%4d  lstm xp %.2f us/step, table %.2f" % (B, a / T, b / T)
        for name, T, S, mode in (("tf", 46, 27, 0), ("sample", 26, 46, 1)):
            V = 96 if mode == 0 else 44
            enc, h0, mask = r(B, S, H), r(B, H), torch.ones(B, S, device=dev)
            w_c, w_hh, w_p, b_p = r(4 * H, H, scale=0.05), r(4 * H, H, scale=0.05), r(V, H, scale=0.3), r(V)
            packs = (pack_fragments(w_c), pack_fragments(w_hh), None, None)
            etable = r(V, 4 * H, scale=0.5)
            teacher = torch.randint(0, V, (B, T), device=dev) if mode == 0 else None
            t = clock(lambda: _AttnLSTMDecoder.apply(None, etable, enc, mask, h0, w_c, w_hh, w_p, b_p, mode, T, 5, 0, 0, 1, 2, packs, teacher))
            out += "   decoder %s %.2f us/step" % (name, t / T / max(1, -(-B // 512)))
        print(out)
