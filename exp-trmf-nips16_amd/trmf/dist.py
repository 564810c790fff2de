"""Multi-GPU bootstrap: one process per GPU, the library's own RCCL communicator over xGMI.

``torch.distributed`` is plumbing only: it carries the 128-byte RCCL unique id from rank 0 to the
other ranks (and, in tests, implements the host-staged all-gather callback over gloo).  The data
path -- all-gathers of freshly solved factor rows / Gram blocks between half-iterations -- lives in
the C++ library (csrc/comm.hpp).  No reference counterpart: the reference is single-node OpenMP.

    import torch.distributed as dist                     # torch FIRST (see note below)
    from trmf import dist as tdist
    dist.init_process_group('nccl', ...)
    tdist.init_rccl()                                    # every rank; both element-type libraries, device = LOCAL_RANK
    ... Session(...) / trmf.train(...)                   # identical inputs on every rank
    tdist.finalize()

Note: import torch before this package touches the GPU so that the process binds to ONE HIP/RCCL
runtime (torch bundles its own copies under the same sonames).
"""
import ctypes

import numpy as np

from . import session


def _dtypes(dtype):
    """``dtype=None`` means both element-type libraries: each trmf_float{32,64}.so keeps its own communicator."""
    return (np.float32, np.float64) if dtype is None else (dtype,)


def init_rccl(dtype=None, device=None):
    """Create the library's RCCL communicator(s) on every rank of the default torch process group.

    ``dtype=None`` (default) initialises both element-type libraries, so that fp32 and fp64 training are both sharded;
    ``device`` defaults to ``LOCAL_RANK`` (one process per GPU, as torchrun launches them)."""
    import os
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    if device is None:
        device = int(os.environ.get('LOCAL_RANK', 0))
    for dt in _dtypes(dtype):
        lib = session.lib_for(dt)
        if lib.trmf_set_device(int(device)) != 0:
            raise RuntimeError(lib.trmf_last_error().decode())
        ident = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            if lib.trmf_dist_get_unique_id(buf) != 0:
                raise RuntimeError(lib.trmf_last_error().decode())
            ident = [buf.raw]
        dist.broadcast_object_list(ident, src=0)
        if lib.trmf_dist_init(rank, world, ident[0]) != 0:
            raise RuntimeError(lib.trmf_last_error().decode())
    return rank, world


_keepalive = {}


def init_host_staged(dtype=np.float32, group=None):
    """Host-staged communicator: all-gathers go device -> host -> torch.distributed (any backend,
    e.g. gloo) -> device.  For tests and RCCL-less setups; same sharding logic as the RCCL path."""
    import torch
    import torch.distributed as dist
    lib = session.lib_for(dtype)
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def allgatherv(buf, offsets, nworld, ctx):
        try:
            off = [int(offsets[i]) for i in range(nworld + 1)]
            total = off[-1]
            arr = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_uint8)), shape=(total,))
            for r in range(nworld):
                n = off[r + 1] - off[r]
                if n == 0:
                    continue
                piece = torch.from_numpy(arr[off[r]:off[r + 1]].copy() if r == rank else np.empty(n, dtype=np.uint8))
                dist.broadcast(piece, src=dist.get_global_rank(group, r) if group is not None else r, group=group)
                if r != rank:
                    arr[off[r]:off[r + 1]] = piece.numpy()
            return 0
        except Exception as exc:    # noqa: BLE001 - must not propagate through the C frame
            import sys
            sys.stderr.write('allgatherv callback failed: {}\n'.format(exc))
            return -1

    cb = session.ALLGATHERV_FN(allgatherv)
    _keepalive[(id(lib), 'cb')] = cb
    if lib.trmf_dist_init_callback(rank, world, cb, None) != 0:
        raise RuntimeError(lib.trmf_last_error().decode())
    return rank, world


def init_solo(rank, world, dtype=np.float32):
    """Measurement aid: rank ``rank`` of ``world`` with no peers (gathers skipped) -- a rank's compute share alone on
    the GPU; the factors a session produces under it are not a solution."""
    lib = session.lib_for(dtype)
    if lib.trmf_dist_init_solo(int(rank), int(world)) != 0:
        raise RuntimeError(lib.trmf_last_error().decode())
    return rank, world


def finalize(dtype=None):
    """Drop the library communicator(s).  Sessions created under one keep it alive until they are closed."""
    for dt in _dtypes(dtype):
        session.lib_for(dt).trmf_dist_finalize()
        _keepalive.pop((id(session.lib_for(dt)), 'cb'), None)
