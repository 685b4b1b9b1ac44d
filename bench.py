#!/usr/bin/env python3
"""Headline benchmark: CLEVR questions/sec of one module_training step on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): configs/module_training.yml dims (1024x14x14 features,
128 module channels, 1024 projection channels, 1024 classifier units, lr 1e-4), batch 256 per GPU,
synthetic CLEVR-shaped batch (probnmn.data.synthetic, seed 0, programs from the eight template
shapes), random-init weights (seed 0).  A step = zero_grad -> NMN forward -> mean loss -> backward
-> [gradient all-reduce] -> clamp(-5,5) -> Adam, exactly the reference's iteration
(_trainer.py:135-151, module_training_trainer.py:88-98).  Inputs are resident in HBM before the
timed region; programs are re-scheduled on the host every step (nothing is cached across steps
except the per-structure templates).

Besides the contract fields the JSON line carries
  roofline      for the kernel with the largest share of step time (conv_nhwc: forward + data
                gradients of every 3x3 / 1x1 conv): algorithmic FLOPs / launch time measured with
                events on the launch stream in a separate instrumented pass, against the 157.3
                TFLOP/s fp32 matrix peak of gfx950
  cpu_baseline  the CPU oracle's identical step, timed on this host's cores on a 16-question
                sample of the same workload (N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "probnmn-clevr_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
PEAK_HBM_GBS = 8000.0


def log(*a):
    if os.environ.get("RANK", "0") == "0":
        print("[bench %7.1fs]" % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def cpu_baseline(vocab, state_dict, sample, steps, seed):
    """The oracle's step on this host's cores.  torch's default (one thread per physical core) is
    far from the best setting for the reference's batch-1 convolutions -- on a 128-core host 128
    threads run ~3x slower than 16 -- so a few thread counts are tried and the FASTEST is reported
    (the baseline gets the benefit of the doubt); `cores` is the thread count of that run."""
    from oracle.train_oracle import OracleModuleTrainer
    from probnmn.data.synthetic import synthetic_batch

    batch = synthetic_batch(vocab, sample, seed=seed)
    best = None
    for threads in (8, 16, 32):
        if threads > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(threads)
        trainer = OracleModuleTrainer(state_dict, vocab.get_index_to_token_vocabulary("programs"), lr=1e-4)
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
        "sample": "%d-question batch of the same synthetic workload, %d timed steps (median) per thread "
                  "count in {8,16,32}, best reported; PyTorch-CPU fp32 restatement of the reference step"
                  % (sample, steps),
    }


def kernel_rooflines(engine, trainer, batch, passes):
    """Instrumented pass: events around every conv / wgrad launch on the launch stream.  The pass
    runs with the weight-gradient overlap switched off (one stream), so that each kernel's duration
    is its own and not that of two kernels sharing the chip."""
    overlap, engine.overlap_wgrad = engine.overlap_wgrad, False
    engine.event_log = []
    for _ in range(passes):
        trainer.step(batch)
    torch.cuda.synchronize()
    log, engine.event_log = engine.event_log, None
    engine.overlap_wgrad = overlap
    agg = {}
    for kern, what, flops, e0, e1 in log:
        a = agg.setdefault(kern, {"flops": 0.0, "ms": 0.0, "launches": 0, "by": {}})
        ms = e0.elapsed_time(e1)
        a["flops"] += flops
        a["ms"] += ms
        a["launches"] += 1
        b = a["by"].setdefault(what, [0.0, 0.0, 0])
        b[0] += flops
        b[1] += ms
        b[2] += 1
    return agg


def joint_training_extra(vocab, nmn, dev, rank, world, batch_size, steps=5, warmup=2):
    """Side measurement (BASELINE.json configs[3] shape: joint_training_ours.yml, 128 questions per GPU):
    ProgramGenerator sampling + QuestionReconstructor + ProgramPrior + NMN + REINFORCE/ELBO + supervised
    cross entropies, backward, gradient all-reduce, clamp, Adam.  Random-init weights (so most sampled
    programs are invalid and the NMN part is lighter than with a trained generator); reported next to
    the headline, never as `value`."""
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models import ProgramGenerator, ProgramPrior, QuestionReconstructor
    from probnmn.trainers.joint_training import JointTrainingStep

    torch.manual_seed(1)
    pg, qr = ProgramGenerator(vocab).to(dev), QuestionReconstructor(vocab).to(dev)
    prior = ProgramPrior(vocab, hidden_size=256).to(dev)
    for m in (pg, qr):
        m.sample_row_offset = rank * batch_size
    step = JointTrainingStep(pg, qr, prior, nmn, objective="ours", alpha=100.0, beta=0.1, gamma=1.0, delta=0.99, lr=1e-6)
    batch = synthetic_batch(vocab, batch_size, seed=2000 + rank)
    sup = batch["supervision"]
    batch = {k: v.to(dev) for k, v in batch.items()}
    batch["supervision"] = sup
    for _ in range(warmup):
        step.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"metric": "CLEVR questions/sec (joint_training step)", "value": round(batch_size * world * steps / dt, 1),
            "ms_per_step": round(dt / steps * 1e3, 2), "global_batch": batch_size * world, "steps": steps,
            "config": "joint_training_ours.yml (alpha 100, beta 0.1, gamma 1, delta 0.99), %d questions per GPU, "
                      "random-init weights, synthetic batch" % batch_size}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="questions per GPU")
    ap.add_argument("--cpu-sample", type=int, default=16)
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-joint", action="store_true", help="skip the joint_training side measurement")
    ap.add_argument("--joint-batch", type=int, default=128, help="questions per GPU of the joint_training side measurement")
    ap.add_argument("--overlap-wgrad", action="store_true",
                    help="run weight gradients on a second stream concurrently with the data-gradient chain")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from probnmn import parallel
    from probnmn.data.synthetic import synthetic_batch
    from probnmn.models.nmn import NeuralModuleNetwork
    from probnmn.trainers.module_training import ModuleTrainingStep
    from probnmn.vocabulary import Vocabulary

    log("building network")
    vocab = Vocabulary.clevr()
    torch.manual_seed(0)
    net = NeuralModuleNetwork(vocab)  # module_training.yml dims are the constructor defaults
    cpu_sd = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net.to(dev)
    trainer = ModuleTrainingStep(net, lr=1e-4, weight_decay=0.0, report_metrics=False)
    net.engine.overlap_wgrad = args.overlap_wgrad
    parallel.broadcast_parameters(trainer.optimizer.arenas, trainer.optimizer.loose)
    # weak scaling: every rank gets its own batch of --batch questions
    batch = synthetic_batch(vocab, args.batch, seed=1000 + rank, device=dev)
    # programs drive the host-side launch schedule: they stay where the data loader produces them
    # (host memory); everything the kernels read is resident in HBM
    batch["program"] = batch["program"].cpu()

    log("batch ready; warmup")
    for i in range(args.warmup):
        trainer.step(batch)
        torch.cuda.synchronize()
        log("warmup step", i, "done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    from probnmn import _hip
    w0 = _hip.ring_wait_seconds()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(batch)
    host_elapsed = time.perf_counter() - t0
    host_blocked = _hip.ring_wait_seconds() - w0  # part of it spent waiting for the GPU to free a staging slot  # time to ENQUEUE the steps (host scheduling + launches)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log("timed region: %.3f s for %d steps" % (elapsed, args.steps))
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    roof = None
    if rank == 0 and not args.no_roofline:
        agg = kernel_rooflines(net.engine, trainer, batch, passes=2)
        dom = max(agg, key=lambda k: agg[k]["ms"])
        a = agg[dom]
        achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
        roof = {
            "kernel": dom,
            "bound": "mfma",
            "achieved": round(achieved, 2),
            "peak": PEAK_FP32_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
            "traffic": None,
            "avg_launch_ms": round(a["ms"] / a["launches"], 4),
            "launches_per_step": a["launches"] // 2,
            "kernels": {
                k: {"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "ms_per_step": round(v["ms"] / 2, 3),
                    "by_call_site": {w: {"tflops": round(b[0] / (b[1] * 1e-3) / 1e12, 2), "ms_per_step": round(b[1] / 2, 3)}
                                     for w, b in v["by"].items()}}
                for k, v in agg.items()
            },
        }

    log("roofline pass done")
    cpu = None
    if cpu_sd is not None:
        cpu = cpu_baseline(vocab, cpu_sd, args.cpu_sample, args.cpu_steps, seed=1000)

    joint = None
    if not args.no_joint:
        try:
            joint = joint_training_extra(vocab, net, dev, rank, world, args.joint_batch)
        except Exception as exc:  # the headline line must survive a failure of the side measurement
            joint = {"error": "%s: %s" % (type(exc).__name__, exc)}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * world / (elapsed / args.steps)
        plan = net.engine.last_plan
        line = {
            "metric": "CLEVR questions/sec (module_training step)",
            "value": round(value, 1),
            "unit": "questions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "host_enqueue_ms_per_step": round(host_elapsed / args.steps * 1e3, 3),
            "host_busy_ms_per_step": round((host_elapsed - host_blocked) / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "module_training.yml, batch %d per GPU, 14x14x1024 features, programs from 8 CLEVR "
                            "template shapes, fwd+bwd+clamp+Adam" % args.batch,
                "global_batch": args.batch * world,
                "parallelism": "dp%d" % world,
                "streams": 2 if args.overlap_wgrad else 1,
                "module_primitives_per_step": plan.n_prims if plan else None,
            },
            "roofline": roof,
            "cpu_baseline": cpu,
            "joint_training": joint,
        }
        if cpu:
            line["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
