"""Single-node rank launcher for the data-parallel entry points (no counterpart in the reference, which is single-device:
ml/scripts/train.py:113-130).

`python bench.py --gpus N` must run by itself.  The process that parses the command line therefore never touches the GPU: when
N > 1 and no launcher has set WORLD_SIZE, it starts N fresh children of the same script (one rank per GPU, the torch.distributed
environment contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relays rank 0's single JSON line and exits with
the first failing child's code.  Nothing is re-executed in place: a process that has initialised HIP must never exec (the parent
here has not, and it only spawns).  Children are killed by their exact PIDs (each is its own process group), never by pattern.
Pure standard library on purpose: importing this module does not import torch.
"""
from __future__ import annotations

import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence, Tuple


def needs_spawn(n_ranks: int, environ: Optional[Dict[str, str]] = None) -> bool:
    """True when this process is the un-launched parent of an N-rank job (N > 1 and no launcher has set WORLD_SIZE)."""
    env = os.environ if environ is None else environ
    return n_ranks > 1 and "WORLD_SIZE" not in env


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this driver
    return env


def _kill(procs: Sequence[subprocess.Popen], grace_s: float = 5.0) -> None:
    """Terminate the exact children we started (each leads its own process group), then kill what is left."""
    for sig, wait in ((signal.SIGTERM, grace_s), (signal.SIGKILL, 2.0)):
        live = [p for p in procs if p.poll() is None]
        if not live:
            return
        for p in live:
            try:
                os.killpg(p.pid, sig)
            except (ProcessLookupError, PermissionError):
                pass
        t_end = time.monotonic() + wait
        while time.monotonic() < t_end and any(p.poll() is None for p in live):
            time.sleep(0.05)


def last_json_line(text: str) -> Optional[str]:
    """The last line of `text` that parses as a JSON object (rank 0 prints exactly one; warnings of libraries may surround it)."""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                json.loads(line)
                return line
            except ValueError:
                continue
    return None


def spawn_ranks(cmd: Sequence[str], world: int, timeout_s: float = 1500.0, env: Optional[Dict[str, str]] = None,
                poll_s: float = 0.1) -> Tuple[int, Optional[str], str]:
    """Run `cmd` as `world` rank processes and wait for all of them.

    Returns (exit code, rank 0's JSON line or None, diagnostics).  Exit code 0 only if every rank exited 0 and rank 0 printed a JSON
    line; the first non-zero child code otherwise (the other ranks are killed at once: a collective with a dead peer would hang until
    its own timeout); 124 when the time limit ran out.  Rank 0's stdout is captured (and relayed by the caller); the other ranks'
    stdout and every rank's stderr go to this process's stderr."""
    port = free_port()
    procs: List[subprocess.Popen] = []
    out0: List[str] = []
    try:
        for r in range(world):
            procs.append(subprocess.Popen(list(cmd), env=rank_env(r, world, port, env), stdin=subprocess.DEVNULL,
                                          stdout=subprocess.PIPE if r == 0 else 2, stderr=None,      # fd 2 = this process's stderr
                                          start_new_session=True, text=True))
    except Exception:
        _kill(procs)
        raise

    def drain():
        assert procs[0].stdout is not None
        for line in procs[0].stdout:
            out0.append(line)
    reader = threading.Thread(target=drain, daemon=True)
    reader.start()

    t_end = time.monotonic() + timeout_s
    code, why = 0, ""
    try:
        while True:
            states = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(states) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                code = c if c > 0 else 128 - c              # a signal's negative code reads like a shell's 128 + signo
                why = f"rank {r} exited with code {c}; the other ranks were stopped"
                break
            if all(c == 0 for c in states):
                break
            if time.monotonic() > t_end:
                code, why = 124, f"time limit of {timeout_s:.0f} s reached; all ranks were stopped"
                break
            time.sleep(poll_s)
    finally:
        _kill(procs)
    reader.join(timeout=5.0)
    line = last_json_line("".join(out0))
    if code == 0 and line is None:
        code, why = 1, "every rank exited 0 but rank 0 printed no JSON line"
    return code, line, why


def main_spawn(script: str, argv: Sequence[str], world: int, timeout_s: float = 1500.0) -> int:
    """What an entry point calls when needs_spawn(): run `python script argv` as `world` ranks, print rank 0's line, return the code."""
    code, line, why = spawn_ranks([sys.executable, script, *argv], world, timeout_s)
    if line is not None and code == 0:
        print(line, flush=True)
    if why:
        print(f"[launch] {why}", file=sys.stderr, flush=True)
    return code
