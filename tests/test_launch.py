"""The rank launcher behind `python bench.py --gpus N` (fluidnet_amd/training/launch.py): spawn / relay / failure logic driven with
dummy children -- no GPU, no torch in the children."""
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

from fluidnet_amd.training import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text("import json, os, sys, time\nrank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n" + textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_needs_spawn_only_for_an_unlaunched_multi_rank_run():
    assert launch.needs_spawn(2, {})
    assert not launch.needs_spawn(1, {})
    assert not launch.needs_spawn(8, {"WORLD_SIZE": "8"})          # torch.distributed.run (or this launcher) already started the ranks


def test_rank_environment_follows_the_torch_distributed_contract():
    e = launch.rank_env(3, 4, 29123, base={"X": "1"})
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"], e["MASTER_PORT"]) == ("3", "3", "4", "127.0.0.1", "29123")
    assert e["X"] == "1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_rank0_line_is_relayed_and_every_rank_ran(tmp_path):
    cmd = child(tmp_path, f"""
        open(os.path.join({str(tmp_path)!r}, f"ran{{rank}}"), "w").write(os.environ["MASTER_PORT"])
        print("some library chatter")
        if rank == 0:
            print(json.dumps({{"n_gpus": world, "value": 1.5}}))
        else:
            print("{{\\"not\\": \\"rank 0\\"}}")                     # other ranks' stdout must not become the line
    """)
    code, line, why = launch.spawn_ranks(cmd, 3, timeout_s=30)
    assert code == 0 and why == ""
    assert json.loads(line) == {"n_gpus": 3, "value": 1.5}
    ports = {open(tmp_path / f"ran{r}").read() for r in range(3)}
    assert len(ports) == 1                                          # one rendezvous port shared by all ranks


def test_a_failing_rank_stops_the_others_and_sets_the_exit_code(tmp_path):
    cmd = child(tmp_path, """
        if rank == 1:
            sys.exit(7)
        time.sleep(60)                                              # a peer waiting in a collective
    """)
    t0 = time.monotonic()
    code, line, why = launch.spawn_ranks(cmd, 2, timeout_s=30)
    assert code == 7 and line is None and "rank 1" in why
    assert time.monotonic() - t0 < 15                               # did not wait for the sleeping rank


def test_time_limit_kills_all_ranks(tmp_path):
    cmd = child(tmp_path, f"""
        open(os.path.join({str(tmp_path)!r}, f"pid{{rank}}"), "w").write(str(os.getpid()))
        time.sleep(60)
    """)
    code, line, why = launch.spawn_ranks(cmd, 2, timeout_s=1.5)
    assert code == 124 and line is None and "time limit" in why
    for r in range(2):
        pid = int(open(tmp_path / f"pid{r}").read())
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)


def test_no_json_line_is_a_failure(tmp_path):
    code, line, why = launch.spawn_ranks(child(tmp_path, "print('nothing useful')\n"), 2, timeout_s=30)
    assert code == 1 and line is None and "no JSON line" in why


def test_bench_parent_spawns_before_touching_the_gpu(tmp_path):
    """`python bench.py --gpus 2` without a launcher: the parent must start two ranks and relay the outcome.  There is no GPU here,
    so each rank stops at bench.py's own "needs an MI355X" check (exit code 1) -- which proves the ranks were started with the
    distributed environment set, and that the parent itself got past argument parsing without a CUDA call of its own."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--spawn-timeout", "120"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the GPU-side test covers the full run")
    assert r.returncode == 1, (r.returncode, r.stderr[-2000:])
    assert "needs an MI355X" in r.stderr and "[launch] rank" in r.stderr
    assert r.stdout.strip() == ""
