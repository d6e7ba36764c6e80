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
# The library's own RCCL communicator (include/dlwpcs.h: dlwpcs_comm_* / dlwpcs_allreduce_f32): the step's one exchange enqueued
# on the COMPUTE stream through the C ABI, so that in a captured training step it is a plain node of the step's graph (torch's
# ProcessGroupNCCL runs its collectives on a stream of its own: a fork / join around the collective, 17 us per captured step).
#
# OPT-IN (enable_native_comm(True) or DLWPCS_NATIVE_RCCL=1): it has never run on more than one GPU -- torch's all_reduce is the
# default exchange until a run with >= 2 ranks has been seen.  Whoever opts in gets a communicator that
#   * is agreed on by EVERY rank (the request itself is part of the agreement: a rank that did not opt in, could not load RCCL or
#     could not create its communicator moves ALL ranks to torch's all_reduce -- no subset of ranks can wait for the others),
#   * has summed a test vector over the ranks and found n(n+1)/2 everywhere before it is handed out,
#   * belongs to ONE process group: a new init_process_group() releases it and the next call creates a fresh one,
#   * is never created inside a graph capture (the agreement reads flags back): a capture that comes first takes torch's path.
# ---------------------------------------------------------------------------------------------------------------------- #
NATIVE_COMM = None      # None: DLWPCS_NATIVE_RCCL decides (default off); True / False: enable_native_comm()
_native = {'tried': False, 'comm': None, 'group': None, 'why': 'not requested'}


def enable_native_comm(on=True):
    """Ask for (or refuse) the library-owned RCCL communicator for the gradient exchange; every rank must make the same call before
    its first training step (ranks that disagree all end up on torch's all_reduce).  Overrides DLWPCS_NATIVE_RCCL."""
    global NATIVE_COMM
    NATIVE_COMM = None if on is None else bool(on)
    if _native['tried']:
        native_comm_release()           # the next native_comm() call decides anew (a collective: every rank makes this call)


def _native_wanted():
    import os
    if NATIVE_COMM is not None:
        return NATIVE_COMM
    return os.environ.get('DLWPCS_NATIVE_RCCL', '0') == '1'


def _group_token():
    """The default process group OBJECT (identity = this init_process_group() call), None outside one."""
    try:
        return dist.distributed_c10d._get_default_group() if group_alive() else None
    except Exception:
        return None


def _min_over_ranks(ok, dev, n):
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    if n > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


def native_comm():
    """ctypes handle of the library-owned communicator of the CURRENT process group, or None (not asked for, not possible, not
    agreed on; native_comm_status() says which).  Collective on the first call per process group: every rank calls it at the same
    point (Model.compile does)."""
    import os
    token = _group_token()
    if _native['tried'] and _native['group'] is not token:
        native_comm_release(stale=True)                 # the group it was created over is gone (destroy + init again)
    if _native['tried']:
        return _native['comm']
    if token is None or dist.get_backend() != 'nccl' or not torch.cuda.is_available():
        return None
    if torch.cuda.is_current_stream_capturing():
        return None                                     # (not remembered: the first call outside a capture decides)
    _native['tried'], _native['group'] = True, token
    import ctypes
    from . import _native as nat
    lib = nat.lib()
    rank, n = dist.get_rank(), dist.get_world_size()
    dev = torch.device('cuda', torch.cuda.current_device())
    path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
    want = _native_wanted()
    ok = want and os.path.exists(path) and lib.dlwpcs_comm_load(path.encode()) == 0
    idbuf = (ctypes.c_char * 128)()
    if rank == 0 and ok:
        ok = lib.dlwpcs_comm_unique_id(idbuf) == 0
    # the id travels as a device tensor over the existing group; EVERY rank takes part in both collectives whatever its own
    # `want` / `ok` says -- the request is agreed on like everything else
    idt = torch.frombuffer(bytearray(bytes(idbuf)), dtype=torch.uint8).to(dev)
    if n > 1:
        dist.broadcast(idt, src=0)
    if not _min_over_ranks(ok, dev, n):
        _native['why'] = ('not requested (enable_native_comm / DLWPCS_NATIVE_RCCL=1)' if not want else
                          'RCCL could not be loaded or a rank did not ask for it')
        return None
    comm = ctypes.c_void_p()
    raw = bytes(idt.cpu().numpy().tobytes())
    ok = lib.dlwpcs_comm_init(ctypes.byref(comm), raw, rank, n) == 0
    if not _min_over_ranks(ok, dev, n):
        if ok:
            lib.dlwpcs_comm_destroy(comm)
        _native['why'] = 'a rank could not create its communicator: %s' % (lib.dlwpcs_last_error() or b'').decode()
        return None
    # real sums before anybody trusts it: rank r contributes r + 1 in every slot
    probe = torch.full((4099,), float(rank + 1), dtype=torch.float32, device=dev)
    rc = lib.dlwpcs_allreduce_f32(comm, probe.data_ptr(), probe.numel(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ok = rc == 0 and bool((probe == float(n * (n + 1) // 2)).all().item())
    if not _min_over_ranks(ok, dev, n):
        lib.dlwpcs_comm_destroy(comm)
        _native['why'] = 'the test sum over %d ranks came back wrong on a rank' % n
        import warnings
        warnings.warn('DLWP.parallel: the library-owned RCCL communicator failed its test sum; torch.distributed.all_reduce serves')
        return None
    _native['comm'], _native['why'] = comm, 'ok (%d ranks, test sum verified)' % n
    return comm


def native_comm_status():
    """Why native_comm() answers what it answers (bench.py writes it into its JSON line)."""
    return _native['why']


def native_comm_release(stale=False):
    """Destroy the library-owned communicator (before the process group it was created over goes away; native_comm() does it
    itself when it finds a new group)."""
    if _native['comm'] is not None:
        from . import _native as nat
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        nat.lib().dlwpcs_comm_destroy(_native['comm'])
    _native.update(comm=None, tried=False, group=None, why='released' if not stale else 'not requested')


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
