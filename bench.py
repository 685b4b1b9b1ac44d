#!/usr/bin/env python3
"""Headline benchmark: CLEVR questions/sec of one joint_training step on MI355X (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W

(N > 1: one process per GPU over RCCL -- under torch.distributed.run as the driver starts it, or on its own: without
WORLD_SIZE in the environment the script launches its N ranks itself.)

Workload (BASELINE.json configs[3]): configs/joint_training_ours.yml (alpha 100, beta 0.1, gamma 1, delta
0.99, lr 1e-6), GLOBAL batch 1024 questions split evenly over the N ranks ("scaling": "strong" -- configs[3]
as written is 1024 questions over 8 GPUs; N = 1 runs all 1024 on one GPU).  `--scaling weak` gives every rank
1024 questions instead; for N > 1 that run rides along as the side object `weak_scaling`.  14x14x1024
features, synthetic CLEVR-shaped batch (probnmn.data.synthetic: programs from the eight template
shapes, half of the examples with program supervision -- what the reference's
SupervisionWeightedRandomSampler yields, data/samplers.py:5-27).  A step is the reference's
iteration (joint_training_trainer.py:128-198, _trainer.py:135-151): ProgramGenerator samples a
program per unsupervised question -> QuestionReconstructor / ProgramPrior / NMN on the samples ->
REINFORCE-ELBO + gamma * answer loss; teacher-forced cross entropies on the supervised half;
backward -> [gradient all-reduce] -> clamp(-5,5) -> Adam over all three trainable models.

Weights are random-init (seed 0) except that the ProgramGenerator is first fitted for a few hundred
supervised iterations on the synthetic batch (outside the timed region), so that its SAMPLES are
valid programs and the NMN part of the step does the work it does in the reference's joint phase,
which starts from a question_coding checkpoint (an untrained generator emits invalid programs, which
the NMN skips -- a step without its heaviest part).  `config.valid_program_fraction` reports it.
Inputs are resident in HBM before the timed region; the sampled programs travel device -> host every
step because they decide the NMN launch schedule (the one sync the algorithm itself requires).

Besides the contract fields the JSON line carries
  roofline       for the kernel family with the largest share of step time (conv_nhwc: forward +
                 data gradients of every 3x3 / 1x1 conv of the NMN): algorithmic FLOPs / launch time
                 measured with events on the launch stream in separate instrumented steps, against
                 the 157.3 TFLOP/s fp32 matrix peak of gfx950
  cpu_baseline   the CPU oracle's identical step (same weights), timed on this host's cores on a
                 bounded sample of the same workload (N=1 only)
  module_training / question_coding / joint_training_b128 / joint_training_28x28   (N = 1)
                 side measurements of BASELINE.json configs[1], configs[2], of configs[3] read as
                 1024 questions over 8 GPUs (128 per GPU) and of configs[4] (28x28 maps, programs of up
                 to 40 tokens, 128 per GPU); never `value`
  joint_training_ingest
                 the headline step with its features gathered every step from a pinned host store over PCIe
                 (PrefetchingLoader beside the step): questions/s, PCIe GB/s, slowdown vs resident features
  collectives / rccl_ranks / allreduce_ms_per_step / allreduce_hidden_frac   (N > 1)
                 the step's gradient all-reduces alone on an idle chip, what the step's stream actually waits
                 for them, and the fraction hidden behind backward
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# (PNMN_PKG_DIR: A/B aid -- time another checkout of the package against the same library, scripts/ab.sh)
for p in (ROOT, os.environ.get("PNMN_PKG_DIR") or os.path.join(ROOT, "probnmn-clevr_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from probnmn import launch_guard  # noqa: E402

PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
PEAK_HBM_GBS = 8000.0

JOINT = dict(objective="ours", alpha=100.0, beta=0.1, gamma=1.0, delta=0.99, lr=1e-6)


def log(*a):
    launch_guard.beat()
    if os.environ.get("RANK", "0") == "0":
        print("[bench %7.1fs]" % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def cpu_state(module):
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def cpu_baseline(vocab, sds, sample, steps, seed):
    """The oracle's joint-training step on this host's cores.  torch's default (one thread per
    physical core) is far from the best setting for the reference's batch-1 convolutions -- on a
    128-core host 128 threads run ~3x slower than 16 -- so a few thread counts are tried and the
    FASTEST is reported (the baseline gets the benefit of the doubt); `cores` is the thread count of
    that run."""
    from oracle.train_oracle import OracleJointTrainer
    from probnmn.data.synthetic import synthetic_batch

    batch = synthetic_batch(vocab, sample, seed=seed)
    best = None
    for threads in (8, 16, 32):
        if threads > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        trainer = OracleJointTrainer(sds["pg"], sds["qr"], sds["prior"], sds["nmn"],
                                     vocab.get_index_to_token_vocabulary("programs"), **JOINT)
        trainer.step(batch)  # warm-up (allocations, oneDNN primitive caches)
        times = []
        for _ in range(steps):
            t0 = time.perf_counter()
            trainer.step(batch)
            times.append(time.perf_counter() - t0)
        times.sort()
        med = times[len(times) // 2]
        log("cpu baseline: %d threads -> %.1f questions/s" % (threads, sample / med))
        if best is None or sample / med > best[0]:
            best = (sample / med, threads)
        del trainer
    return {
        "value": round(best[0], 2),
        "unit": "questions/s",
        "cores": best[1],
        "host_logical_cpus": os.cpu_count(),
        "kind": "port",
        "sample": "%d-question batch of the same synthetic workload and weights, %d timed joint_training "
                  "steps (median) per thread count in {8,16,32}, best reported; PyTorch-CPU fp32 restatement "
                  "of the reference step" % (sample, steps),
    }


def cpu_config1(vocab, nmn_sd, threads, steps=3, batch=32):
    """BASELINE configs[0]: module_training.yml, batch 32, 14x14x1024 random features, the CPU path -- the
    oracle's module-training iteration (ground-truth programs) on this host's cores, at the thread count the
    joint-step baseline found best.  A plumbing number: it shows the CPU restatement runs the configuration."""
    from oracle.train_oracle import OracleModuleTrainer
    from probnmn.data.synthetic import synthetic_batch

    torch.set_num_threads(threads)
    b = synthetic_batch(vocab, batch, seed=2000)
    trainer = OracleModuleTrainer(nmn_sd, vocab.get_index_to_token_vocabulary("programs"), lr=1e-4)
    trainer.step(b)
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        trainer.step(b)
        times.append(time.perf_counter() - t0)
    times.sort()
    return {"metric": "CLEVR questions/sec (module_training step)", "value": round(batch / times[len(times) // 2], 2),
            "unit": "questions/s", "cores": threads, "global_batch": batch, "steps": steps, "kind": "port",
            "workload": "module_training.yml, batch 32, CPU oracle (configs[0])"}


def source_sha():
    """Hash of everything a kernel's HBM traffic depends on (kernel sources, the C ABI, the launch planner): a PMC
    summary is only quoted beside a bench line taken on the same sources.  (The GPU box has no .git; profiles
    written in the build container also carry the commit, see profiles/summarize.py.)"""
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    try:
        from summarize import source_sha as f
    finally:
        sys.path.pop(0)
    return f(ROOT)


def kernel_rooflines(engine, step_fn, passes, trainer=None):
    """Instrumented steps: the library brackets every conv / wgrad launch with events on the launch stream
    (pnmn_launch_trace_begin / _end: the shipped host path -- trunk planner, launch lists -- runs unchanged).
    ONE stream, so that each kernel's duration is its own and not that of two kernels sharing the chip: a trainer
    that runs the NMN beside the seq2seq passes is switched to its single-stream schedule for these passes.

    Every pass is aggregated on its own and the MEDIAN pass is reported (per kernel family: the pass with the
    median FLOP/s; per call site: the median over passes) -- a stalled event interval or an allocator refill in
    one pass then moves nothing.  Each pass's whole step is timed on the same stream; a kernel family whose summed
    launch time exceeds the single-stream step it ran in cannot be right, and the result is marked suspect."""
    side_stream = getattr(trainer, "nmn_stream", None)
    if side_stream is not None:
        trainer.nmn_stream = False
    per_pass, step_ms = [], []
    table = os.environ.get("PNMN_LAUNCH_TABLE")  # debugging aid: one line per conv / wgrad call of the instrumented steps
    try:
        step_fn()  # (the single-stream schedule's first step allocates)
        for _ in range(passes):
            engine.begin_trace()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            step_fn()
            s1.record()
            torch.cuda.synchronize()
            events = engine.end_trace()
            step_ms.append(s0.elapsed_time(s1))
            agg = {}
            for kern, what, flops, ms, nbytes, launches in events:
                a = agg.setdefault(kern, {"flops": 0.0, "ms": 0.0, "launches": 0, "bytes": 0.0, "by": {}})
                a["flops"] += flops
                a["bytes"] += nbytes
                a["ms"] += ms
                a["launches"] += launches  # (a call with a remainder is two kernel launches)
                b = a["by"].setdefault(what, [0.0, 0.0, 0])
                b[0] += flops
                b[1] += ms
                b[2] += 1
                if table:
                    with open(table, "a") as f:
                        f.write("%-11s %-20s %9.3f GFLOP %8.4f ms %7.1f TF  %d launches\n" % (kern, what, flops / 1e9, ms, flops / ms / 1e9, launches))
            per_pass.append(agg)
    finally:
        try:
            engine.end_trace()  # (a step that raised leaves the trace on)
        except Exception:
            pass
        if side_stream is not None:
            trainer.nmn_stream = side_stream

    def median(xs):
        xs = sorted(xs)
        return xs[(len(xs) - 1) // 2]

    out = {}
    for kern in sorted({k for p in per_pass for k in p}):
        runs = [p[kern] for p in per_pass if kern in p and p[kern]["ms"] > 0]
        mid = median([(r["flops"] / r["ms"], i) for i, r in enumerate(runs)])[1]
        chosen = dict(runs[mid])
        by = {}
        for what in sorted({w for r in runs for w in r["by"]}):
            site = [r["by"][what] for r in runs if what in r["by"] and r["by"][what][1] > 0]
            tf = median([x[0] / x[1] for x in site])
            ms = median([x[1] for x in site])
            by[what] = [tf * ms, ms, median([x[2] for x in site])]
        chosen["by"] = by
        chosen["tflops_per_pass"] = [round(r["flops"] / r["ms"] / 1e9, 2) for r in runs]
        # per LAUNCH figures over all passes: the launch count of a step varies with its sampled programs (the deep ones
        # add tiny launches), and the PMC summaries these are compared with average over their own run's steps
        chosen["bytes_per_launch_all"] = sum(r["bytes"] for r in runs) / max(1, sum(r["launches"] for r in runs))
        chosen["launches_per_pass"] = [r["launches"] for r in runs]
        out[kern] = chosen
    out["_step_ms"] = median(step_ms)
    out["_passes"] = len(per_pass)
    return out


KERNEL_FAMILY = {"conv_nhwc": ("conv_stream_kernel",)}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summaries (profiles/
    rNN_pmc_{FETCH,WRITE}_SIZE.txt: separate `rocprofv3 --pmc` passes of this same command, see
    scripts/profile_round.sh).  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950
    counts 128-byte read requests at 64 bytes, MI355X_MICROARCH.md, HBM).  Counters cannot be collected inside
    this process, so the figure is quoted ONLY when the summaries were taken on these very sources (their
    `# source_sha` header equals source_sha()); None otherwise."""
    import glob
    import re

    tags = sorted({os.path.basename(f).split("_pmc_")[0] for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_FETCH_SIZE.txt"))})
    if not tags:
        return None
    want = source_sha()
    tag, total, calls = tags[-1], 0.0, 0
    win_total, win_alg, win_seen = 0.0, [], 0
    for counter, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        path = os.path.join(ROOT, "profiles", "%s_pmc_%s.txt" % (tag, counter))
        if not os.path.exists(path):
            return None
        n, sha = 0, None
        for line in open(path):
            m = re.match(r"#\s*source_sha:\s*(\w+)", line)
            if m:
                sha = m.group(1)
            # the counters of the PMC run's own instrumented passes, beside that run's algorithmic bytes (one population)
            m = re.match(r"# window %s: last (\d+) launches .*: %s ([0-9.]+) per launch; algorithmic bytes per launch (\d+)" % (kernel, counter), line)
            if m:
                win_total += factor * 1024.0 * float(m.group(2))
                win_alg.append(float(m.group(3)))
                win_seen += 1
            m = re.match(r"(\S+)\s+%s\s+(\d+)\s+([0-9.]+)" % counter, line)
            # (the conv_nhwc family -- pnmn_conv_nhwc calls -- is conv_stream_kernel since round 4; a call is one launch)
            if m and any(name in m.group(1) for name in KERNEL_FAMILY.get(kernel, (kernel,))):
                n += int(m.group(2))
                total += factor * 1024.0 * float(m.group(3))
        if sha != want:
            return None
        calls = max(calls, n)
    if not calls:
        return None
    out = {"bytes_per_launch": round(total / calls), "source": "profiles/%s_pmc_*_SIZE.txt (source_sha %s)" % (tag, want)}
    if win_seen == 2 and min(win_alg) > 0:
        # traffic / algorithmic on ONE population: the PMC runs' instrumented passes (each run's own algorithmic bytes)
        out.update(window_bytes_per_launch=round(win_total), window_algorithmic_bytes_per_launch=round(sum(win_alg) / 2),
                   ratio=round(win_total / (sum(win_alg) / 2), 3))
    return out


def pmc_table(suffix):
    """{kernel name: HBM bytes per launch} from the newest committed pair profiles/*_pmc_<suffix>{FETCH,WRITE}_SIZE.txt whose
    `# source_sha` equals these sources' (None otherwise): 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes (gfx950 correction as in
    pmc_traffic)."""
    import glob
    import re

    tags = sorted({os.path.basename(f).split("_pmc_")[0] for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_%sFETCH_SIZE.txt" % suffix))})
    if not tags:
        return None
    want, table = source_sha(), {}
    for counter, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        path = os.path.join(ROOT, "profiles", "%s_pmc_%s%s.txt" % (tags[-1], suffix, counter))
        if not os.path.exists(path):
            return None
        sha = None
        for line in open(path):
            m = re.match(r"#\s*source_sha:\s*(\w+)", line)
            if m:
                sha = m.group(1)
            m = re.match(r"(\S+)\s+%s\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)" % counter, line)
            if m:
                table[m.group(1)] = table.get(m.group(1), 0.0) + factor * 1024.0 * float(m.group(4))
        if sha != want:
            return None
    table["_source"] = "profiles/%s_pmc_%s*_SIZE.txt" % (tags[-1], suffix)
    return table


def roofline_object(agg, kernel=None):
    kerns = {k: v for k, v in agg.items() if not k.startswith("_")}
    dom = kernel or max(kerns, key=lambda k: kerns[k]["ms"])
    a = kerns[dom]
    achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
    pmc = pmc_traffic(dom)
    total_ms = sum(v["ms"] for v in kerns.values())
    out = {
        "kernel": dom,
        "bound": "mfma",
        "achieved": round(achieved, 2),
        "peak": PEAK_FP32_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
        # PMC figures: per launch of the PMC runs' instrumented passes where the summaries carry them (`# window` lines:
        # traffic and algorithmic bytes of ONE population of launches -> traffic_ratio), else of the whole PMC process
        "traffic": (pmc.get("window_bytes_per_launch") or pmc["bytes_per_launch"]) if pmc else None,
        "traffic_unit": "HBM bytes per launch (PMC, %s)" % pmc["source"] if pmc else None,
        "traffic_ratio": pmc.get("ratio") if pmc else None,
        "traffic_algorithmic_bytes_per_launch": pmc.get("window_algorithmic_bytes_per_launch") if pmc else None,
        "algorithmic_bytes_per_launch": round(a["bytes_per_launch_all"]),
        "launches_per_pass": a["launches_per_pass"],
        "avg_launch_ms": round(a["ms"] / a["launches"], 4),
        "launches_per_step": a["launches"],
        "passes": agg["_passes"],
        "tflops_per_pass": a["tflops_per_pass"],
        "single_stream_step_ms": round(agg["_step_ms"], 3),
        # the timed steps run the NMN trunk on its own stream beside the seq2seq passes; a kernel's duration is its own
        # only when nothing shares the chip with it, so the instrumented passes (and profiles/*_kernel_stats.txt, taken
        # with PNMN_NMN_STREAM=0) use the single-stream schedule of the same launches
        "schedule": "instrumented passes: one stream (PNMN_NMN_STREAM=0); timed steps: trunk on its own stream",
        "kernels": {
            k: {"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "ms_per_step": round(v["ms"], 3),
                "by_call_site": {w: {"tflops": round(b[0] / (b[1] * 1e-3) / 1e12, 2), "ms_per_step": round(b[1], 3)}
                                 for w, b in v["by"].items()}}
            for k, v in kerns.items()
        },
    }
    if total_ms > 1.02 * agg["_step_ms"]:
        # the instrumented kernels cannot take longer than the single-stream step they ran in: the timing is broken
        # (a stalled event interval in most passes) -- say so instead of reporting a fraction
        out.update({"suspect": True, "achieved": None, "frac": None,
                    "why": "sum of instrumented kernel time %.2f ms > single-stream step %.2f ms" % (total_ms, agg["_step_ms"])})
    return out


def recurrent_kernel_report(dev):
    """Launch time, latency per time step and achieved FLOP/s of the persistent recurrent kernels at
    the shapes of the headline step (these kernels are bound by the per-step hand-off
    latency, not by a roofline; both figures are reported)."""
    from probnmn.modules.seq2seq_base import _AttnLSTMDecoder, _LSTMLayerSeq, pack_fragments

    g = torch.Generator().manual_seed(0)
    r = lambda *shape, scale=1.0: (torch.randn(*shape, generator=g) * scale).to(dev)  # noqa: E731

    def clock(fn, reps=5):
        # median of per-call timings: one call in a few hundred lands on an allocator refill (tens of ms)
        for _ in range(2):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2]

    out = {}
    Hd = 256
    # LSTM layer over the questions (rows, steps)
    B, T = 1024, 46
    xp, w = r(B, T, 4 * Hd, scale=0.5).requires_grad_(True), r(4 * Hd, Hd, scale=0.05)
    dhs = r(B, T, Hd)
    # (the fragment-order copies of the recurrent weights are made once per optimiser step and shared by every pass --
    # DerivedParams -- so they are operands here, not part of the launch: rounds 1-4 timed two permute kernels with it)
    wp, w_t = pack_fragments(w), pack_fragments(w.t())
    fwd = clock(lambda: _LSTMLayerSeq.apply(xp.detach(), w, wp, w_t))
    both = clock(lambda: _LSTMLayerSeq.apply(xp, w, wp, w_t).backward(dhs))
    flops = 2.0 * B * T * Hd * 4 * Hd
    # PMC traffic of exactly these launches: scripts/profile_round.sh runs THIS report alone under rocprofv3 --pmc
    # (scripts/r06_recurrent_pmc.py) -> profiles/*_pmc_recurrent_*; quoted only on matching sources
    pmc = pmc_table("recurrent_") or {}

    def traffic(*names):
        hit = [v for k, v in pmc.items() if k != "_source" and any(n in k for n in names)]  # (mangled names begin with "_Z")
        return round(sum(hit)) if hit else None

    out["lstm_layer"] = {"rows": B, "steps": T, "fwd_ms": round(fwd, 3), "fwd_us_per_step": round(fwd / T * 1e3, 2),
                         "fwd_tflops": round(flops / fwd / 1e9, 2), "fwd_bwd_ms": round(both, 3),
                         # step inputs + hidden / cell states + activated gates once; backward: dh, saved gates and cells, dgates
                         "fwd_algorithmic_bytes": B * T * (4 * Hd + 2 * Hd + 4 * Hd) * 4, "fwd_traffic": traffic("lstm_seq_fwd_cluster_kernel"),
                         "bwd_algorithmic_bytes": B * T * (Hd + 4 * Hd + Hd + 4 * Hd) * 4, "bwd_traffic": traffic("lstm_seq_bwd_cluster_kernel")}
    # decoders: the reconstructor's teacher-forced decode and the generator's sampling decode
    for name, B, T, S, mode in (("decoder_teacher_forced", 1024, 46, 27, 0), ("decoder_sampling", 512, 26, 46, 1)):
        V = 96 if mode == 0 else 44
        enc, h0 = r(B, S, Hd).requires_grad_(True), r(B, Hd)
        mask = torch.ones(B, S, device=dev)
        w_c, w_hh = r(4 * Hd, Hd, scale=0.05), r(4 * Hd, Hd, scale=0.05)
        w_p, b_p = r(V, Hd, scale=0.3), r(V)
        etable = r(V, 4 * Hd, scale=0.5)
        teacher = torch.randint(0, V, (B, T), device=dev) if mode == 0 else None  # (inputs = rows of etable, as the models run it)
        dh = r(B, T, Hd)
        packs = (pack_fragments(w_c), pack_fragments(w_hh), pack_fragments(w_c.t()), pack_fragments(w_hh.t()))
        run = lambda e: _AttnLSTMDecoder.apply(None, etable, e, mask, h0, w_c, w_hh, w_p, b_p, mode, T, 5, 0, 0, 1, 2, packs, teacher)[0]  # noqa: E731
        fwd = clock(lambda: run(enc.detach()))
        both = clock(lambda: run(enc).backward(dh))
        flops = B * T * (2.0 * 2 * Hd * 4 * Hd + 4.0 * S * Hd + (2.0 * V * Hd if mode else 0.0))
        launches = max(1, -(-B // 512))
        fwd_kernel = "attn_lstm_fwd_multi_kernelILb1ELb0" if mode == 0 else "attn_lstm_fwd_multi_kernelILb1ELb1"
        t_f, t_b = traffic(fwd_kernel), (traffic("attn_lstm_bwd_multi_kernel") if mode == 0 else None)
        out[name] = {"rows": B, "steps": T, "source_positions": S, "fwd_ms": round(fwd, 3),
                     "fwd_us_per_step": round(fwd / T * 1e3 / launches, 2),
                     "fwd_tflops": round(flops / fwd / 1e9, 2), "fwd_bwd_ms": round(both, 3),
                     # forward: encoder outputs and the token table once, h / c / context / gates / attention written;
                     # backward: those read again + dh, dgates / dctx / dscore / weights written
                     "fwd_algorithmic_bytes": (B * T * (7 * Hd + S) + B * S * Hd + V * 4 * Hd) * 4,
                     "fwd_traffic": t_f * launches if t_f else None,
                     "bwd_algorithmic_bytes": (B * T * (12 * Hd + 3 * S) + B * S * Hd) * 4,
                     "bwd_traffic": t_b * launches if t_b else None}
    out["note"] = ("fwd_us_per_step = launch time / steps (per 512-row launch for the decoders); peak fp32 %.1f TFLOP/s; *_traffic = HBM bytes "
                   "(PMC, 2 x FETCH_SIZE + WRITE_SIZE) of the same launches (%s; null: no summary on these sources; the teacher-forced "
                   "backward kernel's figure also averages the sampling decoder's backward launches)" % (PEAK_FP32_TFLOPS, pmc.get("_source")))
    return out


def hbm_kernel_report(dev, optimizer=None):
    """The HBM-bound kernels of the step at the headline shapes: algorithmic bytes / launch time as a
    fraction of the 8.0 TB/s HBM peak -- the layout change of the input features, the ReLU-gated max-pool (forward
    and backward), the one-channel attention head, And / Or, and the fused clamp + Adam over every trainable
    parameter of the joint step (28 bytes per parameter: gradient, parameter and both moments read, parameter
    and both moments written)."""
    import numpy as np

    from probnmn import _hip

    lib, st = _hip.lib(), _hip.stream_ptr(dev)
    out = {}

    def clock(fn, reps=5):
        # median of per-call timings: one call in a few hundred lands on an allocator refill (tens of ms)
        for _ in range(2):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2]

    def entry(name, ms, nbytes, what):
        out[name] = {"ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1), "frac_of_hbm_peak": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 3),
                     "algorithmic_bytes": int(nbytes), "what": what}

    B, HW, Cin, Cp, Cm = 512, 196, 1024, 1024, 128
    x = torch.randn(B, Cin, HW, device=dev)
    y = torch.empty(B, HW, Cin, device=dev)
    entry("nchw_to_nhwc", clock(lambda: _hip.check(lib.pnmn_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), B, Cin, HW, st), "layout")),
          2.0 * x.numel() * 4, "input features, %d examples of 1024 x 14 x 14" % B)
    cls = torch.randn(B, HW, Cp, device=dev)
    pooled = torch.empty(B, Cp * 49, device=dev)
    dcls = torch.empty_like(cls)
    entry("maxpool_fwd", clock(lambda: _hip.check(lib.pnmn_maxpool2_flatten_fwd(cls.data_ptr(), pooled.data_ptr(), B, 14, 14, Cp, st), "pool")),
          (cls.numel() + pooled.numel()) * 4.0, "ReLU-gated 2x2 max-pool + flatten of the classifier conv, %d examples" % B)
    entry("maxpool_bwd", clock(lambda: _hip.check(lib.pnmn_maxpool2_flatten_bwd(cls.data_ptr(), pooled.data_ptr(), dcls.data_ptr(), B, 14, 14, Cp, st), "pool bwd")),
          (2 * cls.numel() + pooled.numel()) * 4.0, "its backward (input re-read for the arg-max, gradient map written)")
    n = 4096
    a, b2, o = torch.rand(n, HW, Cm, device=dev), torch.rand(n, HW, device=dev), torch.empty(n, HW, Cm, device=dev)
    rec = np.zeros(n, _hip.MINMAX_ITEM)
    e = np.arange(n, dtype=np.int64)
    rec["a"], rec["b"], rec["out"] = a.data_ptr() + e * HW * Cm * 4, b2.data_ptr() + e * HW * 4, o.data_ptr() + e * HW * Cm * 4
    rec["a_channels"], rec["b_channels"], rec["is_max"] = Cm, 1, 1
    items = _hip.to_device(rec, dev)
    entry("minmax_fwd", clock(lambda: _hip.check(lib.pnmn_minmax_fwd(items.data_ptr(), n, HW, Cm, st), "minmax")),
          n * HW * (2 * Cm + 1) * 4.0, "And / Or of a 128-channel map with a 1-channel map, %d items" % n)
    if optimizer is not None:
        params = sum(a.total for a in optimizer.arenas) + sum(p.numel() for p in optimizer.loose if p.grad is not None)
        entry("clamp_adam", clock(lambda: optimizer.step(), reps=3), 28.0 * params,
              "element-wise clamp + Adam over the %d trainable parameters of the step" % params)
    out["note"] = "HBM peak %.1f TB/s (MI355X_MICROARCH.md; measured copy peak 6.29 TB/s)" % (PEAK_HBM_GBS / 1e3)
    return out


def timed(step_fn, steps, warmup, dev, world, trainer=None):
    """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both
    sides; returns (max-over-ranks seconds, host seconds to enqueue, host seconds blocked on the GPU:
    waiting for a staging-ring slot or -- joint training -- for the sampled programs to arrive)."""
    from probnmn import _hip

    def blocked_now():
        return _hip.ring_wait_seconds() + (getattr(trainer, "blocked_seconds", 0.0) if trainer is not None else 0.0)

    # (the warm-up runs as the timed loop does -- the host ahead of the GPU: that is when the pinned staging ring and the
    # allocator's pools reach their steady size; a synchronise behind every warm-up step left that growth to the first
    # timed steps: 45 instead of 34 ms on the 28x28 side object, whose records are the largest)
    for i in range(warmup):
        launch_guard.beat()
        step_fn()
        if i == 0:
            torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    w0 = blocked_now()
    t0 = time.perf_counter()
    for _ in range(steps):
        launch_guard.beat()  # (one utime() per step under the N > 1 launch guard, nothing otherwise)
        step_fn()
    host = time.perf_counter() - t0
    blocked = blocked_now() - w0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, host, blocked


def device_batch(vocab, n, seed, dev, **kw):
    """A synthetic batch resident in HBM.  `supervision` stays on the host (it decides the split sizes,
    as the data loader's CPU tensor does in the reference) and so does `program` for module training
    (it drives the host-side launch schedule)."""
    from probnmn.data.synthetic import synthetic_batch

    batch = synthetic_batch(vocab, n, seed=seed, **kw)
    sup = batch["supervision"]
    batch = {k: v.to(dev) for k, v in batch.items()}
    batch["supervision"] = sup
    return batch


def fit_program_generator(pg, vocab, batch, dev, max_iters, target, min_iters=0):
    """Supervised teacher-forced iterations on the synthetic batch until the generator's samples are
    mostly valid programs (outside every timed region).  Returns the valid fraction of a sampling pass."""
    from probnmn import parallel
    from probnmn.optim import ClampAdam
    from probnmn.runtime.program_compiler import ProgramCompiler

    compiler = ProgramCompiler(vocab.get_index_to_token_vocabulary("programs"))
    opt = ClampAdam(list(pg.parameters()), lr=2e-3, clamp=5.0)
    parallel.broadcast_parameters([], opt.loose)

    def valid_fraction():
        pg.eval()
        with torch.no_grad():
            z = pg(batch["question"], decoding_strategy="sampling")["predictions"].cpu()
        pg.train()
        t = torch.tensor([sum(1 for p in compiler.compile_batch(z) if p.valid), z.shape[0]], dtype=torch.float64, device=dev)
        t = parallel.all_reduce_scalars(t)
        return float(t[0] / t[1])

    # (`min_iters`: the valid fraction is itself a sampled quantity -- a fit that stops the first time one sampling pass
    # reaches the target leaves the 40-token generator anywhere between 150 and 600 iterations from run to run, and the
    # step's module primitives between 2 100 and 3 300: the 28x28 side fits a fixed number of iterations first)
    frac, it = valid_fraction(), 0
    while (frac < target or it < min_iters) and it < max_iters:
        for _ in range(50):
            launch_guard.beat()
            opt.zero_grad()
            pg(batch["question"], batch["program"], decoding_strategy="sampling")["loss"].mean().backward()
            parallel.all_reduce_gradients([], opt.loose)
            opt.step()
        it += 50
        frac = valid_fraction()
        log("program generator fit: %d iterations, %.3f of sampled programs valid" % (it, frac))
    return frac, it


def config5_side(vocab, prior, dev, rank, world, args):
    """BASELINE configs[4]: joint_training_ours.yml with 28x28 feature maps and programs of up to 40
    tokens (deeper module chains), 128 questions per GPU (1024 over 8 GPUs, as BASELINE.md tabulates it).
    Own models (the NMN's classifier is 200704 -> 1024 at this size, the generator decodes 40 steps), own
    pre-fit of the generator on the deep synthetic programs, own conv roofline."""
    from probnmn import parallel
    from probnmn.models import NeuralModuleNetwork, ProgramGenerator, QuestionReconstructor
    from probnmn.trainers.joint_training import JointTrainingStep

    n = args.batch28
    torch.manual_seed(5)
    nmn = NeuralModuleNetwork(vocab, image_feature_size=(1024, 28, 28)).to(dev)
    pg, qr = ProgramGenerator(vocab, max_decoding_steps=40).to(dev), QuestionReconstructor(vocab).to(dev)
    for m in (pg, qr):
        m.sample_row_offset = rank * n
    batch = device_batch(vocab, n, 5000 + rank, dev, image_feature_size=(1024, 28, 28), deep=True)
    valid_fraction, fit_iters = fit_program_generator(pg, vocab, batch, dev, args.fit_iters, args.fit_target, min_iters=min(600, args.fit_iters))
    trainer = JointTrainingStep(pg, qr, prior, nmn, **JOINT)
    parallel.broadcast_parameters(trainer.optimizer.arenas, trainer.optimizer.loose)
    for _ in range(4):
        trainer.step(batch)
    torch.cuda.synchronize()
    elapsed, host, blocked = timed(lambda: trainer.step(batch), 10, 4, dev, world, trainer)
    agg = kernel_rooflines(nmn.engine, lambda: trainer.step(batch), passes=4, trainer=trainer)
    out = None
    if rank == 0:
        roof = roofline_object(agg, "conv_nhwc")
        for k in ("traffic", "traffic_unit"):  # (the committed PMC summaries are of the 14x14 headline)
            roof[k] = None
        out = {
            "metric": "CLEVR questions/sec (joint_training step)",
            "value": round(n * world * 10 / elapsed, 1), "unit": "questions/s", "ms_per_step": round(elapsed / 10 * 1e3, 3),
            "global_batch": n * world, "steps": 10, "warmup": 4,
            "workload": "joint_training_ours.yml with NMN.IMAGE_FEATURE_SIZE [1024,28,28] and programs of up to 40 tokens "
                        "(ProgramGenerator max_decoding_steps 40; synthetic deep programs, mean %.1f tokens), %d questions "
                        "per GPU (configs[4])" % (float((batch["program"] != 0).sum(1).float().mean()), n),
            "valid_program_fraction": round(valid_fraction, 4),
            "program_generator_fit_iterations": fit_iters,
            "module_primitives_per_step": nmn.engine.last_plan.n_prims if nmn.engine.last_plan else None,
            "host_busy_ms_per_step": round((host - blocked) / 10 * 1e3, 3),
            "host_blocked_ms_per_step": round(blocked / 10 * 1e3, 3),
            "roofline": roof,
        }
    trainer.close()
    del trainer, nmn, pg, qr, batch
    torch.cuda.empty_cache()
    return out


def dropin_side(vocab, dev, n, joint_ms, steps=10, warmup=4):
    """What a maintainer of the reference gets from ``probnmn_graft.install()`` with NOTHING else changed: the
    reference's own ``_Trainer.step`` + ``JointTrainingTrainer._do_iteration``
    (/root/reference/probnmn/trainers/_trainer.py:135-151, joint_training_trainer.py:128-198) restated against the class
    surface only -- index the batch by supervision, ``JointTrainingElbo(...)`` on the unsupervised rows (inside it: the
    generator's sampling pass, the reconstructor, the prior, the NMN), the generator and the reconstructor AGAIN on the
    supervised rows, one ``backward()``, the Python clamp loop over every parameter, ``torch.optim.Adam`` over all
    trainable parameters (``_trainer.py:103-108``).  Fresh models (torch's Adam must not step the headline's), same
    batch shape; reported next to this build's ``JointTrainingStep``, which batches the four seq2seq calls into two,
    fuses the objective and runs clamp + Adam as one kernel over the parameter arenas."""
    import itertools

    from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.modules.elbo import JointTrainingElbo

    torch.manual_seed(1)
    nmn = NeuralModuleNetwork(vocab).to(dev)
    pg, qr = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev)
    prior = ProgramPrior(vocab, hidden_size=256).to(dev)
    for p in prior.parameters():
        p.requires_grad_(False)  # (checkpointed and frozen in this phase: joint_training_trainer.py:91-94)
    batch = device_batch(vocab, n, 4000, dev)
    fit_program_generator(pg, vocab, batch, dev, 400, 0.9)  # (an untrained generator samples invalid programs: no NMN work)
    elbo = JointTrainingElbo(pg, qr, prior, nmn, beta=JOINT["beta"], gamma=JOINT["gamma"], baseline_decay=JOINT["delta"],
                             objective="ours")
    params = [p for m in (pg, qr, nmn) for p in m.parameters() if p.requires_grad]
    optimizer = torch.optim.Adam(params, lr=JOINT["lr"], weight_decay=0.0)

    def step():
        optimizer.zero_grad()
        sup = batch["supervision"].nonzero().squeeze()
        nosup = (1 - batch["supervision"]).nonzero().squeeze()
        out = elbo(batch["question"][nosup], batch["image"][nosup], batch["answer"][nosup])
        loss = JOINT["gamma"] * out.pop("nmn_loss") - out["elbo"]
        pg_sup = pg(batch["question"][sup], batch["program"][sup], decoding_strategy="sampling")
        qr_sup = qr(batch["program"][sup], batch["question"][sup], decoding_strategy="sampling")
        loss = loss + JOINT["alpha"] * (pg_sup["loss"].mean() + qr_sup["loss"].mean())
        loss.backward()
        for p in itertools.chain(pg.parameters(), qr.parameters(), nmn.parameters()):
            if p.grad is not None:
                p.grad.clamp_(min=-5, max=5)
        optimizer.step()

    e, h, bl = timed(step, steps, warmup, dev, 1)
    ms = e / steps * 1e3
    return {"metric": "CLEVR questions/sec (joint_training step, the reference's own iteration over the grafted classes)",
            "value": round(n * steps / e, 1), "unit": "questions/s", "ms_per_step": round(ms, 3), "global_batch": n,
            "steps": steps, "warmup": warmup, "host_busy_ms_per_step": round((h - bl) / steps * 1e3, 3),
            "slowdown_vs_joint_training_step": round(ms / joint_ms, 3),
            "workload": "the reference's _Trainer.step + JointTrainingTrainer._do_iteration call for call (four separate "
                        "seq2seq passes, JointTrainingElbo, Python clamp loop, torch.optim.Adam), %d questions" % n}


def evaluation_side(vocab, pg, nmn, dev, n=256, num_batches=8):
    """Validation answer accuracy as the reference's training loop runs it every CHECKPOINT_EVERY iterations
    (/root/reference/probnmn/evaluators/_evaluator.py:67-115, joint_training_evaluator.py:74-103, scripts/train.py:135-140):
    eval mode, no gradients, the generator teacher-forced on the ground-truth programs with "greedy" decoding, the NMN on
    its predictions; ``num_batches + 2`` batches of ``n`` questions (the reference's loop condition)."""
    from probnmn.evaluators import evaluate_answer_accuracy

    batches = [device_batch(vocab, n, 5000 + i, dev) for i in range(num_batches + 2)]
    evaluate_answer_accuracy(pg, nmn, batches[:3], 1)  # (program structures, allocator pools)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    metrics = evaluate_answer_accuracy(pg, nmn, batches, num_batches)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    seen = (num_batches + 2) * n
    return {"metric": "CLEVR questions/sec (validation answer accuracy: forward only)", "value": round(seen / dt, 1),
            "unit": "questions/s", "ms_per_batch": round(dt / (num_batches + 2) * 1e3, 3), "batch": n,
            "batches": num_batches + 2, "answer_accuracy": round(float(metrics["nmn"]["answer_accuracy"]), 4),
            "workload": "evaluate_answer_accuracy: ProgramGenerator teacher-forced greedy + NMN forward, eval mode, no "
                        "autograd, %d batches of %d questions (num_batches = %d)" % (num_batches + 2, n, num_batches)}


def extraction_side(dev, n=128, k=5):
    """The offline feature extractor the reference runs before any training (/root/reference/scripts/preprocess/
    extract_features.py:98-131: torchvision ResNet-101 up to stage 3 on 224x224 images) on csrc/resnet.hip: images / s and
    the fraction of the fp32 matrix roof of its 94 convolutions, random weights, one batch of synthetic images."""
    from probnmn.data.feature_extractor import ResNet101Stage3

    model = ResNet101Stage3().to(dev)
    images = torch.randn(n, 3, 224, 224, device=dev)
    for _ in range(2):
        feats = model(images)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        feats = model(images)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    tf = model.flops_per_image() * n / dt / 1e12
    shape = list(feats.shape[1:])
    del model, images, feats
    torch.cuda.empty_cache()
    return {"metric": "CLEVR images/sec (ResNet-101 stage-3 feature extraction)", "value": round(n / dt, 1), "unit": "images/s",
            "ms_per_batch": round(dt * 1e3, 3), "batch": n, "tflops": round(tf, 2), "frac_of_fp32_mfma_peak": round(tf / 157.3, 4),
            "features": shape, "dtype": "f32",
            "workload": "ResNet101Stage3.forward (94 x pnmn_conv2d_nhwc + max pool), %d synthetic 224x224 images, random weights; "
                        "70 000 training images = %.0f s" % (n, 70000 * dt / n)}


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves (one process per GPU,
    rendezvous on 127.0.0.1) with the same arguments, pass their output through and exit with their status."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (the host driver only supports dmabuf IPC)
    env.setdefault("OMP_NUM_THREADS", "8")
    log("launching %d ranks: %s" % (args.gpus, " ".join(cmd[2:])))
    raise SystemExit(subprocess.call(cmd, env=env))


def collective_report(trainer, step_fn, dev, world, passes=4):
    """What the gradient all-reduces cost the step (N > 1): `exposed` = time the step's stream is held at the
    waits in all_reduce_gradients (events on that stream around them, instrumented steps); `standalone` = the same
    sequence of collectives (same sizes: the early-reduced tensors, the arena pieces, the small-tensor bucket)
    issued back to back on an otherwise idle GPU; hidden = 1 - exposed / standalone."""
    from probnmn import parallel

    parallel.TIMING = []
    for _ in range(passes):
        launch_guard.beat()
        step_fn()
    torch.cuda.synchronize()
    timing, parallel.TIMING = parallel.TIMING, None
    exposed = sorted(e0.elapsed_time(e1) for e0, e1 in timing)
    exposed = exposed[(len(exposed) - 1) // 2] if exposed else 0.0
    opt = trainer.optimizer
    early = getattr(trainer, "_early", None)
    sizes = []
    covered = {}
    if early is not None:
        sizes += [p.numel() for p in early.params] + [hi - lo for _, lo, hi in early.pieces]
        for a in opt.arenas:
            covered[id(a)] = sum(hi - lo for lo, hi in early.covers(a))
    sizes += [a.total - covered.get(id(a), 0) for a in opt.arenas if a.total > covered.get(id(a), 0)]
    first = {id(p) for p in early.params} if early is not None else set()
    rest = [p.numel() for p in opt.loose if id(p) not in first]
    small = sum(n for n in rest if n * 4 < parallel.SMALL_BUCKET_BYTES)
    sizes += [n for n in rest if n * 4 >= parallel.SMALL_BUCKET_BYTES] + ([small] if small else [])
    bufs = [torch.zeros(n, dtype=torch.float32, device=dev) for n in sizes]
    times = []
    for _ in range(5):
        dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for h in [dist.all_reduce(b, async_op=True) for b in bufs]:
            h.wait()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    standalone = sorted(times)[2]
    t = torch.tensor([exposed, standalone], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    exposed, standalone = float(t[0]), float(t[1])
    nbytes = 4 * sum(sizes)
    return {
        "rccl_ranks": dist.get_world_size(),  # (after real all-reduces on this process group)
        "backend": dist.get_backend(),
        "allreduce_bytes_per_step": nbytes,
        "collectives_per_step": len(sizes),
        "allreduce_ms_per_step": round(standalone, 3),
        "allreduce_exposed_ms_per_step": round(exposed, 3),
        "allreduce_hidden_frac": round(max(0.0, 1.0 - exposed / standalone), 3) if standalone > 0 else None,
        "allreduce_algbw_GBs": round(nbytes / standalone / 1e6, 1) if standalone > 0 else None,
        "cluster_cus": int(__import__("probnmn._hip", fromlist=["lib"]).lib().pnmn_cluster_reserve_cus(-1)),
        "reserve": int(os.environ.get("PNMN_DP_RESERVE_CUS", "32")),
        "serial_collectives": parallel.serial_collectives(),
        "note": "allreduce_ms_per_step: the step's collectives alone on an idle chip; exposed: what the step's stream "
                "waits for them behind backward (median of %d instrumented steps, max over ranks)" % passes,
    }


def ingest_side(vocab, trainer, dev, rank, world, args, resident_ms):
    """The same joint step fed by its ingest (reference data/readers.py:63-108, datasets.py:137-142,
    _trainer.py:272-287), fresh random rows of a store of `--ingest-rows` feature rows every step through
    PrefetchingLoader.  Two stores: ``resident`` (DeviceFeatureStore: all rows in HBM in the stem's layout, the step's
    stem conv1 / weight-gradient records point at the selected rows -- no feature byte moves; the reported object) and
    ``pinned_host`` (PinnedFeatureStore: the gather kernel reads the rows over PCIe on the loader's stream while the
    previous step runs -- for sets that do not fit HBM; PNMN_INGEST=dma: the copy engines instead)."""
    from probnmn.data.feature_store import DeviceFeatureStore, PinnedFeatureStore, PrefetchingLoader
    from probnmn.data.synthetic import synthetic_batch

    n, rows, k, w = args.batch, args.ingest_rows, args.steps, 4
    g = torch.Generator().manual_seed(77 + rank)
    store = PinnedFeatureStore.__new__(PinnedFeatureStore)  # (filled in place: no second host copy of 6.6 GB)
    store.shape = (rows, 1024, 14, 14)
    store.store = torch.empty(store.shape, dtype=torch.float32, pin_memory=True)
    blk = torch.randn(256, 1024, 14, 14, generator=g).relu_()
    for lo in range(0, rows, 256):
        store.store[lo:lo + 256] = blk[: min(256, rows - lo)]
    host = synthetic_batch(vocab, n, seed=1000 + rank)
    del host["image"]

    def batches(count):
        for _ in range(count):
            b = dict(host)
            b["image_index"] = torch.randint(0, rows, (n,), generator=g)
            yield b

    def run(the_store, method):
        it = iter(PrefetchingLoader(batches(w + k + 1), the_store, dev, method=method))
        for _ in range(w):
            launch_guard.beat()
            trainer.step(next(it))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            launch_guard.beat()
            trainer.step(next(it))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        it.close()  # (the loader holds one prefetched batch and its stream)
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed / k * 1e3

    feat_bytes = n * 1024 * 196 * 4
    method = os.environ.get("PNMN_INGEST", "kernel")  # (A/B hook: "dma" = one copy-engine transfer per row)
    host_ms = run(store, method)
    pinned = {"ms_per_step": round(host_ms, 3), "value": round(n * world / (host_ms * 1e-3), 1), "method": method,
              "pcie_GBs_per_gpu": round(feat_bytes / (host_ms * 1e-3) / 1e9, 2), "slowdown_vs_resident": round(host_ms / resident_ms, 4)}
    t0 = time.perf_counter()
    resident = DeviceFeatureStore(store.store.numpy(), dev)  # (the same rows, once, into HBM)
    fill_s = time.perf_counter() - t0
    ms = run(resident, "resident")
    del resident, store
    import gc
    gc.collect()
    torch.cuda.empty_cache()  # (the 6.6 GB store goes back to the driver, not into the allocator's pool of the next side object)
    return {"metric": "CLEVR questions/sec (joint_training step, features of fresh rows of an HBM-resident store every step)",
            "value": round(n * world / (ms * 1e-3), 1), "unit": "questions/s", "ms_per_step": round(ms, 3),
            "steps": k, "warmup": w, "store_rows": rows, "store_GB": round(rows * 1024 * 196 * 4 / 1e9, 2),
            "store_fill_seconds": round(fill_s, 2), "slowdown_vs_resident": round(ms / resident_ms, 4), "method": "resident",
            "pinned_host": pinned,
            "workload": "the headline step with batch['image'] = rows of a %d-row fp32 NHWC store in HBM (%.1f GB; 70 000 CLEVR "
                        "images would take 56 GB of the 288), other rows every step: stem conv1 and its weight gradient read "
                        "them through per-example pointers, no gather, no layout pass, no PCIe; `pinned_host`: the same from a "
                        "page-locked host store (gather kernel over PCIe on the loader's stream)" % (rows, rows * 1024 * 196 * 4 / 1e9)}


def main():
    if os.environ.get("PNMN_DUMP_AFTER"):  # debugging aid: where is the host if the run has not finished by then?
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["PNMN_DUMP_AFTER"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed steps first: allocator pools, GEMM heuristics and the program / template caches settle")
    ap.add_argument("--batch", type=int, default=1024,
                    help="questions of the joint_training step: in total (--scaling strong, the default) or per GPU (weak)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="strong (default): --batch questions in total, split evenly over the N ranks -- BASELINE "
                         "configs[3] as written, 1024 over 8 GPUs; weak: --batch questions per GPU (global batch "
                         "grows with N).  N = 1 is the same run either way")
    ap.add_argument("--batch28", type=int, default=128, help="questions per GPU of the 28x28 / 40-token side measurement")
    ap.add_argument("--cpu-sample", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--settle", type=int, default=8, help="set-up iterations before the warm-up (see main)")
    ap.add_argument("--fit-iters", type=int, default=1500, help="cap on the generator's pre-fit iterations")
    ap.add_argument("--fit-target", type=float, default=0.95)
    ap.add_argument("--ingest-rows", type=int, default=8192, help="rows of the pinned host feature store of the ingest side measurement")
    ap.add_argument("--roofline-passes", type=int, default=12,
                    help="instrumented steps of the roofline object (every step samples fresh programs: 73-89 conv launches per step, "
                         "the deep programs adding tiny ones -- the median of a dozen steps does not hang on which four were taken)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements")
    ap.add_argument("--no-extras-but-ingest", action="store_true", help="of the side measurements only joint_training_ingest")
    ap.add_argument("--sides", default=None, help="comma-separated side measurements to run (default: all); A/B aid")
    ap.add_argument("--extras", action="store_true", help="N > 1: run the single-GPU side measurements too (default: N = 1 only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if world > 1 and not launch_guard.is_worker() and os.environ.get("PNMN_BENCH_GUARD", "1") != "0":
        # N > 1: this process only SUPERVISES the rank (probnmn/launch_guard.py, DESIGN 6 "Hang guard"): the rank's work
        # runs in a worker process whose heartbeat it watches; a hang or a crash on any rank restarts all workers once with
        # PNMN_DP_SERIAL_COLLECTIVES=1, and if that fails too rank 0 still prints a JSON line ("hung": true) and every
        # rank exits non-zero instead of waiting forever.
        raise SystemExit(launch_guard.supervise(
            [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
            watchdog_s=float(os.environ.get("PNMN_BENCH_WATCHDOG_S", "150")),
            first_beat_s=float(os.environ.get("PNMN_BENCH_FIRST_BEAT_S", "420")),
            last_resort=lambda info: {
                "metric": "CLEVR questions/sec (joint_training step)", "value": None, "unit": "questions/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic", "hung": True,
                "config": {"workload": "joint_training_ours.yml, global batch %d over %d GPUs -- NOT MEASURED: the ranks hung or "
                                       "failed twice (overlapped and serial collectives)" % (args.batch, world),
                           "parallelism": "dp%d" % world},
                "launch_guard": info}))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback for the measured path)")
    total = args.batch  # as given on the command line
    if args.scaling == "strong":
        if args.batch % world:
            raise SystemExit("--scaling strong needs --batch divisible by the number of ranks")
        global_batch, per_rank = args.batch, args.batch // world
    else:
        global_batch, per_rank = args.batch * world, args.batch
    args.batch = per_rank  # from here on: questions per rank
    # test hooks for boxes with fewer GPUs than ranks (the multi-rank logic of this file can then be
    # exercised with several processes on ONE device over gloo): never set by the driver
    backend = os.environ.get("PNMN_BENCH_BACKEND", "nccl")
    if "PNMN_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["PNMN_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from probnmn import parallel
    from probnmn.models import NeuralModuleNetwork, ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.trainers.joint_training import JointTrainingStep, QuestionCodingStep
    from probnmn.trainers.module_training import ModuleTrainingStep
    from probnmn.vocabulary import Vocabulary

    log("building models")
    vocab = Vocabulary.clevr()
    torch.manual_seed(0)
    nmn = NeuralModuleNetwork(vocab).to(dev)  # module_training.yml / joint_training_ours.yml dims = defaults
    pg, qr = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev)
    prior = ProgramPrior(vocab, hidden_size=256).to(dev)
    for m in (pg, qr):
        m.sample_row_offset = rank * args.batch  # distinct sampler streams per rank

    # every rank its own shard of the global batch
    batch = device_batch(vocab, args.batch, 1000 + rank, dev)
    valid_fraction, fit_iters = fit_program_generator(pg, vocab, batch, dev, args.fit_iters, args.fit_target)

    trainer = JointTrainingStep(pg, qr, prior, nmn, **JOINT)
    parallel.broadcast_parameters(trainer.optimizer.arenas, trainer.optimizer.loose)
    sds = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sds = {"pg": cpu_state(pg), "qr": cpu_state(qr), "prior": cpu_state(prior), "nmn": cpu_state(nmn)}

    # set-up, like the generator fit above: a few iterations so that what only happens at the start of a
    # run (allocator pools growing, GEMM heuristics, templates of program structures seen for the first
    # time) is over before the W warm-up + K timed steps, whatever W the caller passes
    if (os.environ.get("PNMN_BENCH_FORCE_HANG") == "1" and world > 1 and rank == world - 1
            and launch_guard.is_worker() and launch_guard.attempt() == 0):
        # test hook (tests/test_bench_launch_gpu.py): the last rank never joins the first step's collectives -- what a
        # deadlock between a collective and a recurrent kernel looks like from the outside
        log_all = "[bench rank %d] PNMN_BENCH_FORCE_HANG: sleeping instead of stepping" % rank
        print(log_all, file=sys.stderr, flush=True)
        time.sleep(1e6)
    for _ in range(args.settle):
        launch_guard.beat()
        trainer.step(batch)
    torch.cuda.synchronize()
    log("joint_training: warmup + %d timed steps" % args.steps)
    elapsed, host, blocked = timed(lambda: trainer.step(batch), args.steps, args.warmup, dev, world, trainer)
    log("timed region: %.3f s for %d steps" % (elapsed, args.steps))
    prims = nmn.engine.last_plan.n_prims if nmn.engine.last_plan else None

    collectives = None
    if world > 1:
        try:  # every rank takes part
            collectives = collective_report(trainer, lambda: trainer.step(batch), dev, world)
        except Exception as exc:
            collectives = {"error": "%s: %s" % (type(exc).__name__, exc)}

    roof = None
    if not args.no_roofline:
        # every rank runs the instrumented steps (they contain the step's collectives); rank 0 reports
        agg = kernel_rooflines(nmn.engine, lambda: (launch_guard.beat(), trainer.step(batch)), passes=args.roofline_passes, trainer=trainer)
        if rank == 0:
            roof = roofline_object(agg)
        log("roofline pass done")

    cpu = cpu1 = None
    if sds is not None:
        cpu = cpu_baseline(vocab, sds, args.cpu_sample, args.cpu_steps, seed=1000)
        try:
            cpu1 = cpu_config1(vocab, sds["nmn"], cpu["cores"])
            log("cpu module_training (configs[0]): %.1f questions/s" % cpu1["value"])
        except Exception as exc:
            cpu1 = {"error": "%s: %s" % (type(exc).__name__, exc)}

    recurrent = None
    if rank == 0 and not args.no_roofline:
        try:
            recurrent = recurrent_kernel_report(dev)
        except Exception as exc:  # a side report must not take the headline line down
            recurrent = {"error": "%s: %s" % (type(exc).__name__, exc)}
    hbm_kernels = None
    if rank == 0 and not args.no_roofline:
        try:  # (after every timed run of the headline: the optimiser steps below move its parameters)
            # (the optimiser is timed at N = 1 only: stepping rank 0 alone would let its replica drift from the others)
            hbm_kernels = hbm_kernel_report(dev, trainer.optimizer if world == 1 else None)
        except Exception as exc:
            hbm_kernels = {"error": "%s: %s" % (type(exc).__name__, exc)}

    extras = {}
    chosen = set(args.sides.split(",")) if args.sides else None

    def on(name):
        return chosen is None or name in chosen

    want_extras = not args.no_extras and not args.no_extras_but_ingest and (world == 1 or args.extras)

    def side(name, metric, workload, n, make, k=10):
        if not on(name):
            return
        try:
            b = device_batch(vocab, n, 3000 + rank, dev)
            step = make()
            if name == "module_training":
                b["program"] = b["program"].cpu()
            e, h, bl = timed(lambda: step.step(b), k, 6, dev, world, step)
            extras[name] = {"metric": metric, "value": round(n * world * k / e, 1), "unit": "questions/s",
                            "ms_per_step": round(e / k * 1e3, 3), "global_batch": n * world, "steps": k,
                            "warmup": 6, "host_busy_ms_per_step": round((h - bl) / k * 1e3, 3),
                            "host_blocked_ms_per_step": round(bl / k * 1e3, 3), "workload": workload}
            log("%s: %.1f questions/s" % (name, extras[name]["value"]))
        except Exception as exc:  # the headline line must survive a failure of a side measurement
            extras[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    if not args.no_extras and world > 1 and args.scaling == "strong":
        # weak scaling beside the strong-scaling headline: the one-GPU workload (all of --batch) on EVERY rank
        side("weak_scaling", "CLEVR questions/sec (joint_training step)",
             "joint_training_ours.yml, %d questions per GPU (global batch %d): per-GPU work fixed as N grows" % (total, total * world),
             total, lambda: trainer, k=args.steps)
        if "value" in extras.get("weak_scaling", {}):
            extras["weak_scaling"]["scaling"] = "weak"
    if want_extras:
        # (Rounds 3-4 ran this side in a process of its own: whichever of it and the ingest side came second in one process
        # ran 5-13 ms per step slower on the GPU.  Not the allocator: each built a side / loader stream of its own, and HIP
        # multiplexes a process's streams onto four hardware queues -- the trainers and loaders now share one stream per
        # role, probnmn.trainers.joint_training.shared_stream.)
        try:
            if on("joint_training_28x28"):
                extras["joint_training_28x28"] = config5_side(vocab, prior, dev, rank, world, args)
            if rank == 0 and on("joint_training_28x28"):
                log("joint_training_28x28: %.1f questions/s" % extras["joint_training_28x28"]["value"])
        except Exception as exc:
            extras["joint_training_28x28"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if not args.no_extras and args.ingest_rows > 0 and on("joint_training_ingest"):
        try:
            extras["joint_training_ingest"] = ingest_side(vocab, trainer, dev, rank, world, args, elapsed / args.steps * 1e3)
            log("joint_training_ingest: %.1f questions/s" % extras["joint_training_ingest"]["value"])
        except Exception as exc:
            extras["joint_training_ingest"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if want_extras:
        try:  # (before the trainers below re-hook the models)
            if on("evaluate_answer_accuracy"):
                extras["evaluate_answer_accuracy"] = evaluation_side(vocab, pg, nmn, dev)
                log("evaluate_answer_accuracy: %.1f questions/s" % extras["evaluate_answer_accuracy"]["value"])
        except Exception as exc:
            extras["evaluate_answer_accuracy"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        try:
            # train -> validate -> train, as the reference's loop does every CHECKPOINT_EVERY iterations
            # (/root/reference/scripts/train.py:135-140): the headline step timed again right behind the validation pass
            # above (eval mode, other batch sizes, other workspaces) -- a process that has been through an evaluation
            # must train at the speed it trained at before
            if on("train_validate_train") and "value" in extras.get("evaluate_answer_accuracy", {}):
                e2, h2, bl2 = timed(lambda: trainer.step(batch), args.steps, 2, dev, world, trainer)
                extras["train_validate_train"] = {
                    "metric": "CLEVR questions/sec (joint_training step, timed again behind a validation pass)",
                    "value": round(total * args.steps / e2, 1), "unit": "questions/s",
                    "ms_per_step": round(e2 / args.steps * 1e3, 3), "ms_per_step_before": round(elapsed / args.steps * 1e3, 3),
                    "slowdown_vs_before": round(e2 / elapsed, 4), "steps": args.steps, "warmup": 2,
                    "workload": "the headline step, %d steps, right behind evaluate_answer_accuracy (%d batches of 256 "
                                "questions) in the same process" % (args.steps, extras["evaluate_answer_accuracy"]["batches"])}
                log("train_validate_train: %.3f ms per step (%.3f before)" % (e2 / args.steps * 1e3, elapsed / args.steps * 1e3))
        except Exception as exc:
            extras["train_validate_train"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        side("joint_training_b128", "CLEVR questions/sec (joint_training step)",
             "joint_training_ours.yml, 128 questions per GPU (configs[3] read as 1024 over 8 GPUs)", 128,
             lambda: trainer, k=40)  # (8 ms steps whose sampled programs differ: 10 of them scatter by +-4 %)
        side("question_coding", "CLEVR questions/sec (question_coding step)",
             "question_coding_ours.yml (ProgramGenerator + QuestionReconstructor + frozen ProgramPrior, REINFORCE-ELBO), "
             "512 questions per GPU (configs[2])", 512,
             lambda: QuestionCodingStep(pg, qr, prior, objective="ours", alpha=100.0, beta=0.1, delta=0.99, lr=1e-3))
        trainer.close()  # (the next trainer hooks the same fully connected layer for its own early all-reduce)
        side("module_training", "CLEVR questions/sec (module_training step)",
             "module_training.yml, 256 questions per GPU, ground-truth programs, NMN fwd+bwd+clamp+Adam (configs[1])", 256,
             lambda: ModuleTrainingStep(nmn, lr=1e-4, weight_decay=0.0, report_metrics=False))
        try:
            if on("joint_training_dropin"):
                extras["joint_training_dropin"] = dropin_side(vocab, dev, total, elapsed / args.steps * 1e3)
                log("joint_training_dropin: %.1f questions/s" % extras["joint_training_dropin"]["value"])
        except Exception as exc:
            extras["joint_training_dropin"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        try:  # (last: its 128-image activations are the largest blocks any side object leaves in the allocator)
            if on("feature_extraction"):
                extras["feature_extraction"] = extraction_side(dev)
                log("feature_extraction: %.1f images/s" % extras["feature_extraction"]["value"])
        except Exception as exc:
            extras["feature_extraction"] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * world / (elapsed / args.steps)
        line = {
            "metric": "CLEVR questions/sec (joint_training step)",
            "value": round(value, 1),
            "unit": "questions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "host_enqueue_ms_per_step": round(host / args.steps * 1e3, 3),
            "host_busy_ms_per_step": round((host - blocked) / args.steps * 1e3, 3),
            "host_blocked_ms_per_step": round(blocked / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (random-init weights; ProgramGenerator pre-fitted on the synthetic batch so that "
                    "sampled programs are valid)",
            "config": {
                "workload": "joint_training_ours.yml (NMN + seq2seq + REINFORCE), global batch %d = %d questions per GPU "
                            "x %d (BASELINE configs[3]%s), 14x14x1024 features, half of the batch with program "
                            "supervision, fwd+bwd+[all-reduce]+clamp+Adam"
                            % (global_batch, args.batch, world, "" if (global_batch == 1024 and args.scaling == "strong") else "; non-default batch/scaling"),
                "global_batch": global_batch,
                "per_gpu_batch": args.batch,
                "parallelism": "dp%d" % world,
                "valid_program_fraction": round(valid_fraction, 4),
                "program_generator_fit_iterations": fit_iters,
                "module_primitives_per_step": prims,
            },
            "roofline": roof,
            "cpu_baseline": cpu,
            "module_training_cpu_b32": cpu1,
            "recurrent_kernels": recurrent,
            "hbm_bound_kernels": hbm_kernels,
        }
        if collectives is not None:
            line["collectives"] = collectives
            for k in ("rccl_ranks", "allreduce_ms_per_step", "allreduce_hidden_frac"):
                line[k] = collectives.get(k)
        line.update(extras)
        if cpu:
            line["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
