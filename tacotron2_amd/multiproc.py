"""One training process per MI355X of this node (reference multiproc.py:1-23).

    python -m tacotron2_amd.multiproc -m tacotron2_amd.train -o out -l logs --hparams=distributed_run=True
    python -m tacotron2_amd.multiproc path/to/script.py <args>

Every child gets ``--n_gpus=<N> --group_name=group_<stamp> --rank=<i>`` appended, as the
reference launcher does; rank 0 keeps the terminal, the others write to
``logs/<stamp>_GPU_<i>.log``.  Differences: the children inherit an environment prepared for
RCCL over xGMI on this driver (dmabuf IPC), a failing rank takes the others down instead of
leaving them blocked in a collective, and the exit status is the first non-zero child status.
``torch.distributed.run`` works as well (``train.py`` honours RANK / WORLD_SIZE / LOCAL_RANK).
"""
import os
import subprocess
import sys
import time


def child_commands(argv, num_gpus, stamp, python=None):
    """The N command lines the launcher starts (pure function: covered by a CPU test)."""
    python = python or sys.executable
    shared = list(argv) + ['--n_gpus={}'.format(num_gpus), '--group_name=group_{}'.format(stamp)]
    return [[python] + shared + ['--rank={}'.format(i)] for i in range(num_gpus)]


def _supervise(children, sinks, poll_s):
    """Wait for the children; a failing rank takes the others down (exact PIDs we started, never a pattern) instead of
    leaving them blocked in a collective.  Returns the first non-zero exit status."""
    status = 0
    try:
        live = list(children)
        while live:
            time.sleep(poll_s)
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and status == 0:
                    status = rc
                    for q in live:
                        q.terminate()
    finally:
        for p in children:
            if p.poll() is None:
                p.kill()
        for s in sinks:
            if s is not None:
                s.close()
    return status


def _rccl_env():
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return env


def launch(argv, num_gpus=None, log_dir='logs', poll_s=0.5):
    if num_gpus is None:
        import torch
        num_gpus = torch.cuda.device_count()
    if num_gpus < 1:
        raise SystemExit("tacotron2_amd.multiproc: no MI355X visible")
    stamp = time.strftime("%Y_%m_%d-%H%M%S")
    env = _rccl_env()
    os.makedirs(log_dir, exist_ok=True)
    children, sinks = [], []
    for i, cmd in enumerate(child_commands(argv, num_gpus, stamp)):
        sink = None if i == 0 else open(os.path.join(log_dir, "{}_GPU_{}.log".format(stamp, i)), "w")
        sinks.append(sink)
        print(cmd)
        children.append(subprocess.Popen(cmd, stdout=sink, stderr=subprocess.STDOUT if sink else None, env=env))
    return _supervise(children, sinks, poll_s)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_environments(num_ranks, port, base=None):
    """The N environments of a torch.distributed.run-style launch on this node (pure function: covered by a CPU test):
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR (loopback: the container host name may not resolve) / MASTER_PORT."""
    envs = []
    for i in range(num_ranks):
        env = dict(base if base is not None else _rccl_env())
        env.update(RANK=str(i), LOCAL_RANK=str(i), WORLD_SIZE=str(num_ranks), LOCAL_WORLD_SIZE=str(num_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        envs.append(env)
    return envs


def launch_env(script_argv, num_ranks, log_dir='logs', poll_s=0.2, python=None):
    """Start ``python <script_argv>`` once per rank with the rank in the ENVIRONMENT (what ``bench.py`` and ``train.py``
    read, as under torch.distributed.run) instead of the reference launcher's ``--rank`` arguments.  Rank 0 keeps the
    terminal, the others log to ``<log_dir>/rank<i>.log``; returns the first non-zero exit status."""
    python = python or sys.executable
    os.makedirs(log_dir, exist_ok=True)
    children, sinks = [], []
    for i, env in enumerate(rank_environments(num_ranks, free_port())):
        sink = None if i == 0 else open(os.path.join(log_dir, "rank{}.log".format(i)), "w")
        sinks.append(sink)
        children.append(subprocess.Popen([python] + list(script_argv), stdout=sink,
                                         stderr=subprocess.STDOUT if sink else None, env=env))
    return _supervise(children, sinks, poll_s)


if __name__ == '__main__':
    sys.exit(launch(sys.argv[1:]))
