"""Hang guard for the one-process-per-GPU data-parallel launch (``torch.distributed.run`` starts N ranks; reference:
probnmn/trainers/_trainer.py:94-100 replicates inside ONE process, so it has nothing to guard).

Why: this build's recurrent kernels share a row tile among workgroups that spin for each other, and RCCL's collectives
keep workgroups resident that wait for their peers on other GPUs; the early gradient all-reduces put the two on one
chip at the same time.  ``parallel.dp_safe`` sizes the recurrent grids so that both fit, but that rule has not been
through every RCCL version -- and a rank that never returns from a step leaves a benchmark or a training job with no
result at all.  So every rank runs as a SUPERVISOR (this module, no GPU context of its own) around the process that
does the work:

* the worker touches a heartbeat file (``beat()``) at every step; a worker whose heartbeat is older than the limit, or
  that exits non-zero, makes its supervisor raise an abort flag in a small key-value store all supervisors share
  (the launcher's own rendezvous store when there is one);
* on abort every supervisor kills ITS worker (the process group it started, by id) and the N supervisors start the
  workers again with the fallback environment -- ``PNMN_DP_SERIAL_COLLECTIVES=1``: every gradient collective behind
  backward, nothing overlapped (probnmn.parallel) -- on a fresh rendezvous port;
* if that attempt fails too, rank 0's supervisor prints the caller's last-resort line (``"hung": true``) and all exit
  non-zero -- the launcher never waits for a rank that will not come back.

Rank 0's worker's stdout is held back and released when an attempt has succeeded on EVERY rank, with a ``launch_guard``
object merged into its last JSON line (attempts, whether the fallback ran, the limits)."""
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import threading
import time
from typing import Callable, Dict, List, Optional

CHILD_ENV = "PNMN_GUARD_CHILD"
BEAT_ENV = "PNMN_GUARD_BEAT"
ATTEMPT_ENV = "PNMN_GUARD_ATTEMPT"
SERIAL_ENV = "PNMN_DP_SERIAL_COLLECTIVES"

_beat_path = os.environ.get(BEAT_ENV)


def is_worker() -> bool:
    return os.environ.get(CHILD_ENV) == "1"


def attempt() -> int:
    return int(os.environ.get(ATTEMPT_ENV, "0"))


def beat() -> None:
    """Worker side: "this rank is alive and making progress" (one utime call; nothing outside a guarded launch)."""
    if _beat_path:
        try:
            os.utime(_beat_path, None)
        except OSError:
            pass


def _log(rank: int, *a) -> None:
    print("[launch_guard rank %d]" % rank, *a, file=sys.stderr, flush=True)


class _Store:
    """The supervisors' shared flags: the launcher's c10d store (torch.distributed.run hosts one on MASTER_PORT) or, without
    a launcher, one that rank 0's supervisor hosts on ``PNMN_GUARD_PORT`` (default MASTER_PORT + 1)."""

    def __init__(self, rank: int, world: int):
        from torch.distributed import TCPStore
        from datetime import timedelta

        host = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ["MASTER_PORT"])
        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
        if agent:
            self.store = TCPStore(host, port, is_master=False, timeout=timedelta(seconds=120))
        else:
            gport = int(os.environ.get("PNMN_GUARD_PORT", port + 1))
            self.store = TCPStore(host, gport, world_size=None, is_master=(rank == 0), wait_for_workers=False,
                                  timeout=timedelta(seconds=120))
        self.prefix = "pnmn_guard/%s/" % os.environ.get("TORCHELASTIC_RUN_ID", "none")

    def set(self, key: str, value: str) -> None:
        self.store.set(self.prefix + key, value)

    def get(self, key: str) -> str:
        return self.store.get(self.prefix + key).decode()

    def has(self, key: str) -> bool:
        return bool(self.store.check([self.prefix + key]))

    def add(self, key: str, n: int = 1) -> int:
        return int(self.store.add(self.prefix + key, n))


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _kill(child: subprocess.Popen) -> None:
    """End the worker this supervisor started: its own process group (``start_new_session``), by id."""
    if child.poll() is not None:
        return
    try:
        os.killpg(child.pid, signal.SIGTERM)
    except ProcessLookupError:
        return
    try:
        child.wait(timeout=5)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        try:
            child.wait(timeout=20)
        except subprocess.TimeoutExpired:
            pass


def _wait_count(store: _Store, key: str, world: int, seconds: float, abort_key: Optional[str] = None) -> str:
    """Until ``key`` has been added to by every rank ("all"), ``abort_key`` appears ("abort") or time runs out ("timeout")."""
    end = time.monotonic() + seconds
    while time.monotonic() < end:
        if store.add(key, 0) >= world:
            return "all"
        if abort_key is not None and store.has(abort_key):
            return "abort"
        time.sleep(0.2)
    return "timeout"


def _leave(store: _Store, rank: int, world: int, tag: str) -> None:
    """Last store access of a supervisor: rank 0 may be HOSTING the store (no launcher), so it stays until every rank has
    signed off -- a supervisor still polling a counter when the host exits would die on the closed connection."""
    try:
        store.add(tag, 1)
        if rank == 0:
            _wait_count(store, tag, world, 30.0)
    except Exception as exc:  # (nothing left to coordinate: the outcome is decided)
        _log(rank, "sign-off: %r" % (exc,))


def supervise(worker_argv: List[str], watchdog_s: float = 150.0, first_beat_s: float = 420.0,
              fallback_env: Optional[Dict[str, str]] = None,
              last_resort: Optional[Callable[[Dict], Dict]] = None) -> int:
    """Run ``worker_argv`` as this rank's worker under the guard (see the module docstring); returns the exit status the
    rank should leave with.  ``watchdog_s``: oldest heartbeat tolerated once the worker has beaten once; ``first_beat_s``:
    time allowed up to the first heartbeat (interpreter + torch import on a cold box, process-group set-up).
    ``last_resort(info)``: the JSON object rank 0 prints when both attempts failed."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    fallback_env = {SERIAL_ENV: "1"} if fallback_env is None else fallback_env
    store = _Store(rank, world)
    info = {"attempts": 0, "hung": False, "fallback": None, "watchdog_s": watchdog_s, "first_beat_s": first_beat_s,
            "reasons": []}
    beat_file = tempfile.NamedTemporaryFile(prefix="pnmn_beat_r%d_" % rank, delete=False)
    beat_file.close()
    try:
        for a in range(2):
            info["attempts"] = a + 1
            env = dict(os.environ)
            env[CHILD_ENV], env[BEAT_ENV], env[ATTEMPT_ENV] = "1", beat_file.name, str(a)
            if a == 1:
                env.update(fallback_env)
                info["fallback"] = dict(fallback_env)
                # a fresh rendezvous for the second group of workers: keys of the first group's process group are still in
                # the launcher's store, so rank 0's worker hosts a new one on a port rank 0's supervisor picked
                if rank == 0:
                    store.set("a1/port", str(_free_port()))
                env["MASTER_PORT"] = store.get("a1/port")
                env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
            os.utime(beat_file.name, None)
            started = time.monotonic()
            first_mtime = os.stat(beat_file.name).st_mtime
            child = subprocess.Popen(worker_argv, env=env, stdout=subprocess.PIPE, start_new_session=True)
            captured: List[bytes] = []
            reader = threading.Thread(target=lambda: captured.extend(child.stdout), daemon=True)
            reader.start()
            abort_key, done_key = "a%d/abort" % a, "a%d/done" % a
            status = None
            while status is None:
                rc = child.poll()
                mtime = os.stat(beat_file.name).st_mtime
                beaten = mtime != first_mtime
                age = time.time() - mtime if beaten else time.monotonic() - started
                if rc is not None:
                    status = "ok" if rc == 0 else "exit %d" % rc
                elif age > (watchdog_s if beaten else first_beat_s):
                    status = "no heartbeat for %.0f s" % age
                elif store.has(abort_key):
                    status = "aborted"
                else:
                    time.sleep(0.25)
            if status == "ok":
                store.add(done_key, 1)
                end = _wait_count(store, done_key, world, max(watchdog_s, 60.0), abort_key)
                if end != "all":
                    status = "aborted" if end == "abort" else "other ranks did not finish"
            if status != "ok":
                if status != "aborted":
                    store.set(abort_key, "rank %d: %s" % (rank, status))
                    _log(rank, "attempt %d failed here: %s" % (a, status))
                _kill(child)
            reader.join(timeout=5)
            if status == "ok":
                _leave(store, rank, world, "a%d/left" % a)
                _release(rank, captured, info)
                return 0
            reason = store.get(abort_key) if store.has(abort_key) else status
            info["reasons"].append(reason)
            info["hung"] = True
            # nobody starts the next attempt before every worker of this one is gone
            store.add("a%d/cleared" % a, 1)
            _wait_count(store, "a%d/cleared" % a, world, 90.0)
            if rank == 0:
                _log(rank, "attempt %d aborted (%s)%s" % (a, reason, "; restarting the workers with %s" % fallback_env if a == 0 else ""))
        _leave(store, rank, world, "left")
        if rank == 0 and last_resort is not None:
            print(json.dumps(last_resort(info)), flush=True)
        return 3
    finally:
        try:
            os.unlink(beat_file.name)
        except OSError:
            pass


def _release(rank: int, captured: List[bytes], info: Dict) -> None:
    """Pass the worker's stdout on; rank 0's last JSON line gains the ``launch_guard`` object."""
    lines = [ln.decode(errors="replace").rstrip("\n") for ln in captured]
    if rank == 0:
        for i in range(len(lines) - 1, -1, -1):
            if lines[i].startswith("{"):
                try:
                    obj = json.loads(lines[i])
                except ValueError:
                    break
                obj["launch_guard"] = info
                if info["hung"]:
                    obj["hung_first_attempt"] = True
                lines[i] = json.dumps(obj)
                break
    for ln in lines:
        print(ln, flush=True)
