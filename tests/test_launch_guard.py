"""probnmn.launch_guard: N supervisors around N workers -- a hang or a crash on one rank restarts every worker once with
PNMN_DP_SERIAL_COLLECTIVES=1 on a fresh rendezvous; a second failure ends every rank non-zero with a last-resort JSON
line from rank 0.  CPU only (gloo), no GPU: the workers are tests/helpers/guard_worker.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUPERVISOR = os.path.join(ROOT, "tests", "helpers", "guard_supervisor.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_plain(mode, world=2):
    """The supervisors started directly (no launcher: rank 0's supervisor hosts the guard store)."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PNMN_GUARD_PORT=str(_free_port()) if r == 0 else "")
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        procs.append((r, env))
    gport = procs[0][1]["PNMN_GUARD_PORT"]
    running = []
    for r, env in procs:
        env["PNMN_GUARD_PORT"] = gport
        running.append(subprocess.Popen([sys.executable, SUPERVISOR, mode], env=env, stdout=subprocess.PIPE,
                                        stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=240) for p in running]
    return [p.returncode for p in running], [o[0].decode() for o in outs], [o[1].decode() for o in outs]


def _json_line(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text
    return json.loads(lines[0])


def test_healthy_run_passes_output_through():
    rcs, outs, errs = _run_plain("ok")
    assert rcs == [0, 0], errs
    line = _json_line(outs[0])
    assert line["value"] == 9.0 and line["serial"] is False and line["attempt"] == 0
    assert line["launch_guard"]["attempts"] == 1 and line["launch_guard"]["hung"] is False
    assert "not the json line" in outs[0] and "{" not in outs[1]  # (gloo itself prints a line per rank)


@pytest.mark.parametrize("mode", ["hang0", "crash0"])
def test_hang_or_crash_restarts_with_serial_collectives(mode):
    rcs, outs, errs = _run_plain(mode)
    assert rcs == [0, 0], errs
    line = _json_line(outs[0])
    assert line["value"] == 9.0 and line["serial"] is True and line["attempt"] == 1
    g = line["launch_guard"]
    assert g["attempts"] == 2 and g["hung"] is True and g["fallback"] == {"PNMN_DP_SERIAL_COLLECTIVES": "1"}
    assert line["hung_first_attempt"] is True and len(g["reasons"]) == 1
    assert ("no heartbeat" in g["reasons"][0]) if mode == "hang0" else ("exit 7" in g["reasons"][0])


def test_second_failure_ends_every_rank_with_a_line():
    rcs, outs, errs = _run_plain("hang")
    assert rcs == [3, 3], errs
    line = _json_line(outs[0])
    assert line["hung"] is True and line["value"] is None and line["launch_guard"]["attempts"] == 2
    assert "{" not in outs[1]


def test_under_torch_distributed_run():
    """The driver's launcher: the supervisors share the launcher's store; the fallback workers rendezvous on a fresh port."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), SUPERVISOR, "hang0"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = _json_line(p.stdout.decode())
    assert line["serial"] is True and line["launch_guard"]["attempts"] == 2 and line["value"] == 9.0
