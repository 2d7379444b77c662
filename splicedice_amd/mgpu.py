"""Running a sub-command on several GPUs of one node: one process per GPU, started by
`python -m torch.distributed.run --nproc-per-node N ... -m splicedice_amd <subcommand> ...`.

The reference is single-process; this is new.  The launcher's environment (RANK, WORLD_SIZE, LOCAL_RANK,
MASTER_*) is read ONCE, before anything touches the GPU; torch.distributed (gloo) is the control plane
only (barrier, the 128-byte RCCL id); the data plane is the library's own RCCL collectives
(distributed.RcclComm).  Every rank runs the same sub-command on its junction rows and writes its rows
of each table into a part file; rank 0 stitches the parts together, so ONE set of output files
appears, byte-identical to the single-process run.
"""
import os
import shutil


class Launcher:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.pg = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo")
            self.pg = dist

    @property
    def root(self):
        return self.rank == 0

    def barrier(self):
        if self.pg:
            self.pg.barrier()

    def bcast_bytes(self, b, n):
        if not self.pg:
            return b
        import torch
        t = torch.frombuffer(bytearray(b if self.rank == 0 else bytes(n)), dtype=torch.uint8).clone()
        self.pg.broadcast(t, src=0)
        return bytes(t.numpy().tobytes())

    def comm(self, engine):
        """RCCL communicator for the HIP Context; gloo for a host engine (CPU tests); none for one rank"""
        from . import distributed
        if self.world == 1:
            return distributed.SingleComm()
        if hasattr(engine, "comm_init"):
            return distributed.RcclComm(engine, self.rank, self.world, self.bcast_bytes)
        return distributed.GlooComm()

    def row_block(self, n, rank=None):
        r = self.rank if rank is None else rank
        return r * n // self.world, (r + 1) * n // self.world

    def part(self, path):
        return f"{path}.part{self.rank}"

    def stitch(self, path):
        """all ranks have written `part(path)`: rank 0 concatenates them into `path` in rank order"""
        self.barrier()
        if self.root:
            with open(path, "wb") as out:
                for r in range(self.world):
                    with open(f"{path}.part{r}", "rb") as src:
                        shutil.copyfileobj(src, out, 1 << 24)
                    os.remove(f"{path}.part{r}")
        self.barrier()


_launcher = None


def launcher():
    global _launcher
    if _launcher is None:
        _launcher = Launcher()
    return _launcher
