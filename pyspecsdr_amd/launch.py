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
    """WORLD_SIZE and RANK set = somebody started this process as one rank of a job: torch.distributed.run / torchrun (which also export
    LOCAL_RANK / TORCHELASTIC_RUN_ID), or a scheduler wrapper (srun, mpirun) that exports only those two — a rank of such a multi-rank job
    must never start ranks of its own (N ranks x N children on N GPUs).  The one exception: WORLD_SIZE = 1 without a launcher's own
    variables is what a batch environment leaves behind for a plain `python script`: that process may still launch its ranks itself."""
    if "WORLD_SIZE" not in os.environ or "RANK" not in os.environ:
        return False
    if "LOCAL_RANK" in os.environ or "TORCHELASTIC_RUN_ID" in os.environ:
        return True
    return int(os.environ["WORLD_SIZE"]) > 1


def _local_rank():
    """LOCAL_RANK as the launcher gives it; under a scheduler wrapper that does not: its own per-node index, else RANK modulo the devices."""
    for k in ("LOCAL_RANK", "SLURM_LOCALID", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK"):
        if k in os.environ:
            return int(os.environ[k])
    rank = int(os.environ["RANK"])
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        n = 0
    return rank % n if n > 0 else rank


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
        return world, int(os.environ["RANK"]), _local_rank()
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
    import threading
    import time
    rc = 1
    for attempt in range(3):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(script)] + list(argv)
        sys.stderr.write("launching: " + " ".join(cmd) + "\n")
        sys.stderr.flush()
        # the ranks' stderr is passed on line by line AS IT ARRIVES (progress, warnings, the traceback of a job that hangs) while it is
        # scanned for the rendezvous's bind failure; only a job that dies of it within its first seconds is started again
        t0 = time.monotonic()
        clash = [False]
        p = subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE, text=True, bufsize=1)

        def pump():
            for line in p.stderr:
                if time.monotonic() - t0 < 30.0 and any(m in line for m in ("Address already in use", "EADDRINUSE", "address already in use")):
                    clash[0] = True
                sys.stderr.write(line)
                sys.stderr.flush()
        th = threading.Thread(target=pump, daemon=True)
        th.start()
        rc = p.wait()
        th.join(timeout=5.0)
        if rc == 0 or not clash[0] or time.monotonic() - t0 > 60.0:
            break
    sys.exit(rc)
