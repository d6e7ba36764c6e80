"""
Host -> HBM feed for Model.fit on arrays that live in host memory (the reference's own feeding mode: numpy batches from
DLWP/model/generators.py or plain arrays handed to keras' fit, Azure/train_cs.py:452-456).

A batch of the headline configuration is 49.5 MB of fp32 (x + y) against a 0.93 ms training step.  Measured on the box
(tools/stage_probe.py): one host pass over a 24.8 MB array 0.35 ms (0.56 ms for a fancy-indexed gather), the PCIe copy
0.44 ms per array — so uploading on the COMPUTE stream puts 0.9 ms of copy in front of every 0.93 ms step (and 0.7-1.8 ms
of host passes before that).  What is done about it:

  * every array of a batch is gathered / converted in ONE pass straight into a pinned fp32 staging buffer (no `a[sel]`
    temporary, no `ascontiguousarray`), split over a few worker threads (numpy releases the GIL while it copies);
  * the copy to the device runs on a COPY stream; the training loop only makes the compute stream wait for the upload's
    event.  A step is merely enqueued by the host, so the staging pass and the upload of batch k+1 overlap the training of
    batch k; the pinned buffers and device tensors are reused through events.

Nothing here touches the numbers: the device receives the same fp32 values as with a plain `tensor.to(device)`.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


class LazyTake(object):
    """rows `sel` (index array or slice) of a host array, not materialised yet"""
    __slots__ = ('array', 'sel', 'shape')

    def __init__(self, array, sel):
        self.array = array
        self.sel = sel
        n = len(range(*sel.indices(array.shape[0]))) if isinstance(sel, slice) else len(sel)
        self.shape = (n,) + tuple(array.shape[1:])

    def materialise(self):
        return self.array[self.sel]


def _rows(sel, i0, i1):
    if isinstance(sel, slice):
        start, stop, step = sel.indices(1 << 62)
        return slice(start + i0 * step, start + i1 * step, step)
    return sel[i0:i1]


class Stager(object):
    MAX_SLOTS = 32
    # pinned buffers of one shape with an upload still in flight before the host waits for the oldest: a step is only
    # ENQUEUED by fit(), so without a bound the feed would stage up to MAX_SLOTS buffers ahead (1.6 GB of pinned memory at the
    # headline configuration).  (Model._feed bounds the run-ahead against the COMPUTE stream the same way.)
    MAX_AHEAD = 6

    def __init__(self, device, workers=4):
        self.device = device
        self.workers = int(workers)
        self.copy_stream = torch.cuda.Stream(device=device)
        self._rings = {}            # shape -> [[pinned tensor, event recorded behind its last upload or None], ...]
        self._pool = None

    # ---- pinned buffers ------------------------------------------------------------------------------------------
    def _slot(self, shape, in_use):
        """a pinned buffer of `shape` that no upload still reads; `in_use`: slots handed out earlier in the same upload()"""
        ring = self._rings.setdefault(shape, [])
        busy = [s for s in ring if s[1] is not None and not s[1].query() and not any(s is u for u in in_use)]
        if len(busy) >= self.MAX_AHEAD:
            busy[0][1].synchronize()            # back-pressure: wait for the oldest upload of this shape
        for slot in ring:
            if any(slot is u for u in in_use):
                continue
            if slot[1] is None or slot[1].query():
                slot[1] = None
                return slot
        if len(ring) < self.MAX_SLOTS:
            ring.append([torch.empty(shape, dtype=torch.float32, pin_memory=True), None])
            return ring[-1]
        for slot in ring:                       # every slot busy: wait for the first that is not part of this batch
            if not any(slot is u for u in in_use) and slot[1] is not None:
                slot[1].synchronize()
                slot[1] = None
                return slot
        raise RuntimeError('Stager: a batch needs more than %d staging buffers of shape %s' % (self.MAX_SLOTS, shape))

    def _fill(self, dst, src):
        """one pass: gather / convert `src` (array, host tensor or LazyTake) into the pinned buffer `dst` (numpy view)"""
        if isinstance(src, LazyTake):
            arr, sel, n = src.array, src.sel, src.shape[0]
        else:
            if isinstance(src, torch.Tensor):       # host tensor; numpy has no bfloat16 / float16-on-some-builds view
                src = (src if src.dtype in (torch.float32, torch.float64) else src.float()).detach().numpy()
            arr, sel, n = np.asarray(src), slice(None), dst.shape[0]

        def part(i0, i1):
            np.copyto(dst[i0:i1], arr[_rows(sel, i0, i1)], casting='unsafe')

        w = max(1, min(self.workers, n))
        if w == 1 or dst.nbytes < (4 << 20):
            part(0, n)
            return
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=self.workers)
        step = -(-n // w)
        list(self._pool.map(lambda i0: part(i0, min(i0 + step, n)), range(0, n, step)))

    # ---- one batch -----------------------------------------------------------------------------------------------
    def upload(self, items, dtypes):
        """items: host arrays / LazyTake / tensors; dtypes: the torch dtype each one is wanted in.  Returns the device tensors
        and the event (on the copy stream) the consumer's stream has to wait for; the tensors are allocated on the copy
        stream, so the consumer also has to `record_stream` them."""
        staged, in_use = [], []
        for item in items:
            if isinstance(item, torch.Tensor) and item.is_cuda:
                staged.append((item, None))
                continue
            shape = tuple(item.shape)
            if int(np.prod(shape)) == 0:
                staged.append((torch.zeros(shape, dtype=torch.float32), None))
                continue
            slot = self._slot(shape, in_use)
            in_use.append(slot)
            self._fill(slot[0].numpy(), item)
            staged.append((slot[0], slot))
        out = [None] * len(staged)
        # Device-resident items pass through UNTOUCHED: whoever produced them did so on the compute stream, and the copy
        # stream does not wait for that stream -- converting them here would race with their producer.  Model._forward casts
        # on the compute stream.
        for i, (t, slot) in enumerate(staged):
            if slot is None and t.is_cuda:
                out[i] = t
        with torch.cuda.stream(self.copy_stream):
            for i, ((t, slot), dt) in enumerate(zip(staged, dtypes)):
                if out[i] is not None:
                    continue
                d = t.to(self.device, non_blocking=True)
                out[i] = d if d.dtype == dt else d.to(dt)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            for t, slot in staged:
                if slot is not None:
                    slot[1] = ev
        return out, ev


class Downloader(object):
    """Device -> host for predict(): results leave on a copy stream into pinned buffers; the host copies batch k-1 into the
    caller's array while batch k computes."""

    def __init__(self, device, depth=3):
        self.device = device
        self.depth = int(depth)
        self.stream = torch.cuda.Stream(device=device)
        self._rings = {}
        self._pending = []          # [(destination numpy view, pinned tensor, event)]

    def _pinned(self, shape):
        ring = self._rings.setdefault(shape, {'bufs': [], 'next': 0})
        if len(ring['bufs']) < self.depth:
            ring['bufs'].append(torch.empty(shape, dtype=torch.float32).pin_memory())
            return ring['bufs'][-1]
        buf = ring['bufs'][ring['next'] % self.depth]
        ring['next'] += 1
        return buf

    def push(self, dst, t):
        """queue `dst[...] = t` (t: device tensor, any float dtype)"""
        while len(self._pending) >= self.depth - 1:
            self._retire()
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        buf = self._pinned(tuple(t.shape))
        with torch.cuda.stream(self.stream):
            buf.copy_(t if t.dtype == torch.float32 else t.float(), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        t.record_stream(self.stream)
        self._pending.append((dst, buf, ev))

    def _retire(self):
        dst, buf, ev = self._pending.pop(0)
        ev.synchronize()
        np.copyto(dst, buf.numpy())

    def flush(self):
        while self._pending:
            self._retire()
