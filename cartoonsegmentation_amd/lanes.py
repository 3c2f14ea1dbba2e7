"""Frame lanes -- several frames in flight on one MI355X (new functionality; the reference loops image by image,
run_kenburns_batch.py:36-62).

One 1024 x 1024 frame is ~500 kernel launches, many of them too small to fill 256 CUs (RTMDet's 20 x 20 / 40 x 40 maps), with two
host reads in the chain (the detector's kept-instance count, the six depth-range scalars the reference also reads on the host).  A
single in-order frame loop therefore leaves CUs idle at batch 1.  FrameLanes runs L worker threads, each with its OWN pipeline
object (programs, workspaces), its own HIP streams and its own share of the frames; while one lane waits for a host read the others
keep enqueueing, and the small kernels of one frame run under the large ones of another.  Results are those of the serial loop:
every frame is computed by the same programs on its own stream, and frames are independent.
"""
import queue
import threading

import torch


class FrameLanes:
    def __init__(self, make_worker, lanes=2, device=None):
        """make_worker(lane_index) -> callable(item) -> result; called once per lane, ON the lane's thread and stream, so whatever it
        builds (KenBurnsPipeline, WarpFrame, ...) is private to the lane.  Build / tune the first pipeline before creating lanes: the
        conv autotuner times kernels and should not share the GPU with other lanes."""
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.n = max(1, int(lanes))
        self._in = [queue.Queue() for _ in range(self.n)]
        self._out = queue.Queue()
        self._streams = [torch.cuda.Stream(self.device) for _ in range(self.n)]
        self._threads = []
        self._ready = threading.Barrier(self.n + 1)
        self._make = make_worker
        for i in range(self.n):
            t = threading.Thread(target=self._run, args=(i,), daemon=True)
            t.start()
            self._threads.append(t)
        self._ready.wait()

    @staticmethod
    def _record(res, stream, seen=None):
        """results are allocated on the lane's stream and consumed on the caller's: tell the caching allocator, so that a block the
        caller drops is not handed back to the lane while the caller's stream still has kernels reading it.  `seen` is the cycle guard
        of ONE top-level call (an argument, not shared state: several lane threads hand results over at the same time)"""
        if seen is None:
            seen = set()
        if isinstance(res, torch.Tensor):
            if res.is_cuda:
                res.record_stream(stream)
        elif isinstance(res, dict):
            for v in res.values():
                FrameLanes._record(v, stream, seen)
        elif isinstance(res, (list, tuple, set, frozenset)):
            for v in res:
                FrameLanes._record(v, stream, seen)
        elif hasattr(res, '__dict__') or hasattr(res, '__slots__'):
            # an object that HOLDS tensors (a dataclass, an AnimeInstances, a KenBurnsConfig): walk its attributes once -- a worker that
            # returns such an object would otherwise keep the cross-stream allocator hazard silently
            if id(res) in seen:
                return
            seen.add(id(res))
            names = list(getattr(res, '__dict__', {}).keys())
            for klass in type(res).__mro__:                # inherited __slots__ as well
                sl = klass.__dict__.get('__slots__', ())
                names += [n for n in ((sl,) if isinstance(sl, str) else sl) if isinstance(n, str)]
            for n in names:
                try:
                    v = getattr(res, n)
                except AttributeError:
                    continue
                if isinstance(v, (torch.Tensor, dict, list, tuple, set, frozenset)) or hasattr(v, '__dict__') or hasattr(v, '__slots__'):
                    FrameLanes._record(v, stream, seen)

    def _run(self, i):
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(self._streams[i]):
            try:
                fn = self._make(i)
            except BaseException as e:                     # surfaced by every later map()
                fn, self._err = None, e
            self._ready.wait()
            while True:
                job = self._in[i].get()
                if job is None:
                    return
                gen, idx, item, consumer = job
                try:
                    if fn is None:
                        raise RuntimeError("lane %d failed to build its worker: %r" % (i, self._err))
                    res = fn(item)
                    self._streams[i].synchronize()         # the result is complete when it is handed over
                    self._record(res, consumer)
                    self._out.put((gen, idx, res, None))
                except BaseException as e:
                    self._out.put((gen, idx, None, e))

    def map(self, items):
        """process items (frame k on lane k mod L); returns the results in input order.  If an item raises, ALL results of the call
        are collected first (nothing of this call is left in the queue for the next one) and the first error is re-raised."""
        if getattr(self, '_err', None) is not None:
            raise self._err
        items = list(items)
        cur = torch.cuda.current_stream(self.device)
        for s in self._streams:
            s.wait_stream(cur)                             # inputs produced on the caller's stream are visible to the lanes
        self._gen = getattr(self, '_gen', 0) + 1
        for k, it in enumerate(items):
            self._in[k % self.n].put((self._gen, k, it, cur))
        out, first_err, got = [None] * len(items), None, 0
        while got < len(items):
            gen, idx, res, err = self._out.get()
            if gen != self._gen:                           # a straggler of an earlier call (cannot happen after a complete drain; belt and braces)
                continue
            got += 1
            if err is not None and first_err is None:
                first_err = err
            out[idx] = res
        if first_err is not None:
            raise first_err
        return out

    def close(self):
        for q in self._in:
            q.put(None)
        for t in self._threads:
            t.join(timeout=5)
