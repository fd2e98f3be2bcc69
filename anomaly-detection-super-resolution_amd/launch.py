"""One process per GPU.  ``--gpus N`` without a launcher (no WORLD_SIZE in the environment) starts N fresh ranks from a
parent that has NOT touched the GPU: a process that has initialised HIP must neither fork nor exec workers, so the
children are ``spawn``-started interpreters and the parent only waits for them and passes their exit code on.
Rendezvous is always 127.0.0.1 (the container hostname may not resolve).  ``launch_plan`` is the pure part (tested on
the CPU); ``spawn`` runs it."""
from __future__ import annotations

import os
import socket
from typing import Callable, Dict, List, Optional


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_plan(n_gpus: int, port: Optional[int] = None, env: Optional[Dict[str, str]] = None) -> List[Dict[str, str]]:
    """Environment of each rank to start, or [] when this process must run as the single rank / is already a rank of a
    launcher (``torch.distributed.run`` sets WORLD_SIZE)."""
    env = os.environ if env is None else env
    if n_gpus <= 1 or "WORLD_SIZE" in env:
        return []
    port = free_port() if port is None else port
    return [{"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_gpus), "MASTER_ADDR": "127.0.0.1",
             "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
            for r in range(n_gpus)]


def _entry(rank: int, plan: List[Dict[str, str]], target: Callable, args: tuple) -> None:
    os.environ.update(plan[rank])
    target(*args)


def spawn(target: Callable, n_gpus: int, args: tuple = ()) -> bool:
    """Start the planned ranks running ``target(*args)`` (a module-level function; it reads RANK / WORLD_SIZE from the
    environment like a rank started by ``torch.distributed.run``).  Returns False when nothing had to be started (the
    caller then runs ``target`` itself); raises if a rank fails."""
    plan = launch_plan(n_gpus)
    if not plan:
        return False
    import torch.multiprocessing as mp
    mp.start_processes(_entry, args=(plan, target, args), nprocs=len(plan), join=True, start_method="spawn")
    return True
