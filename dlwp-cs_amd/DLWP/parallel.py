"""
Data-parallel plumbing of the engine: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests).  The path shards over the batch axis only: every rank holds a full
replica of the 2.7 MB parameter buffer and exchanges exactly one flat fp32 gradient buffer per step.
"""
import torch
import torch.distributed as dist


def world():
    """(rank, world_size) of this process; (0, 1) outside a process group."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_parameters(flat_params, src=0):
    """Make every replica start from rank `src`'s parameters (called once by Model.compile)."""
    if world()[1] > 1:
        dist.broadcast(flat_params, src=src)
    return flat_params


# ---------------------------------------------------------------------------------------------------------------------- #
# The library's own RCCL communicator (include/dlwpcs.h: dlwpcs_comm_* / dlwpcs_allreduce_f32; round 5): the step's one exchange
# is enqueued on the COMPUTE stream through the C ABI, so that in a captured training step it is a plain node of the step's graph
# (torch's ProcessGroupNCCL runs its collectives on a stream of its own: a fork / join around the collective, 17 us per step in
# the captured form).  Created on first use -- a collective over the torch process group, which carries the 128-byte unique id
# and the ranks' agreement that every one of them has a communicator (else nobody uses it: torch's all-reduce serves).
# DLWPCS_NATIVE_RCCL=0 turns it off.
# ---------------------------------------------------------------------------------------------------------------------- #
_native = {'tried': False, 'comm': None}


def native_comm():
    """ctypes handle of the library-owned communicator of this process group, or None.  Collective on first call."""
    import os
    if _native['tried']:
        return _native['comm']
    if not (group_alive() and dist.get_backend() == 'nccl' and torch.cuda.is_available()):
        return None
    _native['tried'] = True
    if os.environ.get('DLWPCS_NATIVE_RCCL', '1') == '0':
        return None
    import ctypes
    from . import _native as nat
    lib = nat.lib()
    rank, n = dist.get_rank(), dist.get_world_size()
    dev = torch.device('cuda', torch.cuda.current_device())
    path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
    ok = os.path.exists(path) and lib.dlwpcs_comm_load(path.encode()) == 0
    idbuf = (ctypes.c_char * 128)()
    if rank == 0 and ok:
        ok = lib.dlwpcs_comm_unique_id(idbuf) == 0
    # the id travels as a device tensor over the existing group (every rank takes part, whatever its own `ok` says)
    idt = torch.frombuffer(bytearray(bytes(idbuf)), dtype=torch.uint8).to(dev)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    if n > 1:
        dist.broadcast(idt, src=0)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    comm = ctypes.c_void_p()
    if int(flag.item()):
        raw = bytes(idt.cpu().numpy().tobytes())
        ok = lib.dlwpcs_comm_init(ctypes.byref(comm), raw, rank, n) == 0
    else:
        ok = False
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    if n > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()):
        _native['comm'] = comm
    elif ok:
        lib.dlwpcs_comm_destroy(comm)
    return _native['comm']


def native_comm_release():
    """Destroy the library-owned communicator (before the process group it was created over goes away)."""
    if _native['comm'] is not None:
        from . import _native as nat
        torch.cuda.synchronize()
        nat.lib().dlwpcs_comm_destroy(_native['comm'])
    _native['comm'], _native['tried'] = None, False


def allreduce_gradients(flat_grads):
    """
    Sum the flat gradient buffer over all ranks (in place) and return the scale (1/world) the optimizer applies, so that
    the update equals the gradient of the mean loss over the global batch (equal per-rank batch sizes).  RCCL: through the
    library's communicator on the CURRENT stream (capturable as a plain graph node); else torch's all_reduce.
    """
    w = world()[1]
    if _exchange_wanted():
        comm = native_comm() if flat_grads.is_cuda and flat_grads.dtype == torch.float32 and flat_grads.is_contiguous() else None
        if not exchange_enabled():
            pass                                        # bench.py only: the step keeps its form, the collective itself is left out
        elif comm is not None:
            from . import _native as nat
            nat.check(nat.lib().dlwpcs_allreduce_f32(comm, nat.ptr(flat_grads), flat_grads.numel(), nat.stream_ptr()),
                      'dlwpcs_allreduce_f32')
        else:
            dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return 1.0 / w


# bench.py ONLY: time a step WITHOUT its exchange, to report how much of the all-reduce is exposed.  The replicas diverge while
# this is set; it is a module attribute (not an environment variable) so that nothing can leak it into a real training run.
SKIP_EXCHANGE_FOR_TIMING = False
_skip_warned = False


def exchange_enabled():
    global _skip_warned
    if SKIP_EXCHANGE_FOR_TIMING:
        if not _skip_warned and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            import warnings
            warnings.warn('DLWP.parallel.SKIP_EXCHANGE_FOR_TIMING is set: gradients are NOT summed over the ranks (timing only)')
            _skip_warned = True
        return False
    return True


def _exchange_wanted():
    """world size > 1 -- or DLWPCS_EXCHANGE_FORCE=1 with a process group of ONE rank: the collectives are issued although they
    change nothing, so that RCCL's stream / the async handles / the graph replays around them run on a single GPU (tests).
    (SKIP_EXCHANGE_FOR_TIMING does not change the answer: the step keeps its data-parallel FORM -- reduction | exchange | one
    launch that applies the update -- and only the collective call itself is left out.)"""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('DLWPCS_EXCHANGE_FORCE', '0') == '1'


def exchange_wanted():
    """Does a training step of this process exchange its gradients (world size > 1, or the forced one-rank form of the tests)?"""
    return _exchange_wanted()


def exchange_capturable():
    """Can the all-reduce be captured inside a hipGraph?  RCCL (backend 'nccl') collectives are stream operations; gloo's are
    host-side and never capturable."""
    return group_alive() and dist.get_backend() == 'nccl'


def device_is_shared():
    """Do two ranks of the process group run on the same GPU (the single-GPU multi-rank tests)?  Collective: every rank calls it."""
    if not group_alive() or dist.get_world_size() < 2 or not torch.cuda.is_available():
        return False
    import socket
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    me = (socket.gethostname(), str(getattr(props, 'uuid', '')) or str(torch.cuda.current_device()), torch.cuda.current_device())
    if dist.get_backend() == 'nccl':
        # (RCCL refuses two ranks on one device anyway)
        return False
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, me)
    return len(set(everyone)) < len(everyone)


def all_ranks_agree(flag):
    """True iff `flag` is true on every rank (collective over the process group; True outside one)."""
    if not group_alive() or dist.get_world_size() < 2:
        return bool(flag)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def group_alive():
    """A process group exists: its watchdog thread issues HIP calls of its own (graph captures then use thread-local error mode)."""
    return dist.is_available() and dist.is_initialized()


def allreduce_start(flat_slice):
    """Start the sum of one bucket of the flat gradient buffer over all ranks and return a handle for allreduce_wait (None at
    world size 1).  The collective runs on the process group's own stream behind everything enqueued on the current stream
    so far (RCCL; with gloo on a helper thread), so the launches that follow on the current stream overlap it."""
    if _exchange_wanted() and exchange_enabled() and flat_slice.numel():
        return dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, async_op=True)
    return None


def allreduce_wait(handle):
    """The current stream waits for the bucket (RCCL: a stream dependency, no host block)."""
    if handle is not None:
        handle.wait()


def shard_bounds(n, rank=None, world_size=None):
    """[start, stop) of this rank's contiguous shard of `n` samples (weak scaling: global batch = per-GPU batch x world)."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)
