"""One process per GPU: make `script --gpus N` really run N ranks.

The multi-GPU layout of this package is one process per GPU under torch.distributed (backend "nccl" = RCCL over xGMI).
A benchmark started as `python bench.py --gpus N` — no launcher around it — therefore re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, and a benchmark that IS under a
launcher checks that the launcher's world size is the N it was asked for.  Either way a mismatch ends the run with a message and
a non-zero exit status instead of a one-GPU number labelled N.
"""
import os
import socket
import subprocess
import sys


class LaunchError(SystemExit):
    def __init__(self, msg):
        sys.stderr.write(f"error: {msg}\n")
        super().__init__(2)


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_launcher():
    """torch.distributed.run / torchrun export LOCAL_RANK (and TORCHELASTIC_RUN_ID) beside WORLD_SIZE / RANK; a scheduler wrapper that only
    exports WORLD_SIZE and RANK is not a launcher of ours."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ and ("LOCAL_RANK" in os.environ or "TORCHELASTIC_RUN_ID" in os.environ)


def ensure_ranks(gpus, script, argv, need_gpus=True):
    """Returns (world, rank, local_rank) of THIS process, or never returns:

    * under a launcher (WORLD_SIZE / RANK set): the launcher's world must equal `gpus`, else LaunchError;
    * gpus == 1 and no launcher: (1, 0, 0);
    * gpus > 1 and no launcher: runs `script argv` as `gpus` ranks under torch.distributed.run on 127.0.0.1 (stdout / stderr
      inherited: rank 0's result line is this process's result line) and exits with the launcher's status.
    need_gpus: refuse unless at least `gpus` devices are visible (False for the CPU dry run of the launch path).
    """
    gpus = int(gpus)
    if gpus < 1:
        raise LaunchError(f"--gpus {gpus}: need at least one")
    if under_launcher():
        world = int(os.environ["WORLD_SIZE"])
        if world != gpus:
            raise LaunchError(f"--gpus {gpus} but the launcher started {world} rank(s) (WORLD_SIZE={world}): "
                              f"use --nproc-per-node {gpus}")
        return world, int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    if need_gpus:
        import torch
        have = torch.cuda.device_count()
        if have < gpus:
            raise LaunchError(f"--gpus {gpus} but only {have} GPU(s) are visible on this node (one rank per GPU; no oversubscription)")
    if gpus == 1:
        return 1, 0, 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    # the port is free when it is probed, not necessarily when the rendezvous binds it: a clash (the launcher's own exit status for a
    # rendezvous failure) is retried on another port
    rc = 1
    for attempt in range(3):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(script)] + list(argv)
        sys.stderr.write("launching: " + " ".join(cmd) + "\n")
        sys.stderr.flush()
        p = subprocess.run(cmd, env=env, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(p.stderr)
        rc = p.returncode
        if rc == 0 or not any(m in p.stderr for m in ("Address already in use", "EADDRINUSE", "address already in use")):
            break
    sys.exit(rc)
