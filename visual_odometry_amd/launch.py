"""Start one process per GPU from a plain `python script.py --gpus N` command.

The sharded configurations (BASELINE configs[4]: one sequence per GPU; bench.py --gpus N) run one process per GPU
with torch.distributed for the single statistics collective.  When the command is not already running under a
launcher (no WORLD_SIZE in the environment), the parent -- BEFORE it imports torch or makes any GPU call, so the
process that forks never holds a HIP context and nothing is ever re-exec'd after GPU initialisation -- starts N
children of the same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (a free port,
127.0.0.1), lets them share its stdout / stderr (rank 0 prints the result line), waits for all of them, and
returns non-zero if any child did.  A child that fails takes the others down with it (they would otherwise wait
at the rendezvous for ever).  No torch import here.

Reference loop this shards: module_tests/VisualOdometry_test.m:88-103 (one independent frame loop per sequence).
"""
import os
import signal
import socket
import subprocess
import sys
import time


def under_launcher():
    """True when RANK / WORLD_SIZE were set by a launcher (torch.distributed.run or spawn_ranks)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(script, argv, world, extra_env=None, poll_s=0.1, grace_s=10.0):
    """Run `python script argv...` as `world` rank processes; returns the exit code for the parent (0 = all ok)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this pool
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, "-u", script] + list(argv), env=env))
    code = 0
    live = list(procs)
    try:
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 1
                    print(f"launch: rank {procs.index(p)} exited with {rc}; stopping the other ranks", file=sys.stderr)
                    _stop(live, grace_s)
                    live = []
            if live:
                time.sleep(poll_s)
    except KeyboardInterrupt:
        _stop(live, grace_s)
        code = 130
    return code


def _stop(procs, grace_s):
    """SIGTERM exactly the processes this launcher started, then SIGKILL what is left after the grace period."""
    for p in procs:
        if p.poll() is None:
            p.send_signal(signal.SIGTERM)
    t0 = time.time()
    for p in procs:
        while p.poll() is None and time.time() - t0 < grace_s:
            time.sleep(0.05)
        if p.poll() is None:
            p.kill()
            p.wait()
