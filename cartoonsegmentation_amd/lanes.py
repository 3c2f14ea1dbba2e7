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

    def _run(self, i):
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(self._streams[i]):
            try:
                fn = self._make(i)
            except BaseException as e:                     # surface construction errors on the first map()
                fn, self._err = None, e
            self._ready.wait()
            while True:
                job = self._in[i].get()
                if job is None:
                    return
                idx, item = job
                try:
                    res = fn(item)
                    self._streams[i].synchronize()         # the result is complete when it is handed over
                    self._out.put((idx, res, None))
                except BaseException as e:
                    self._out.put((idx, None, e))

    def map(self, items):
        """process items (frame k on lane k mod L); returns the results in input order"""
        if getattr(self, '_err', None) is not None:
            raise self._err
        items = list(items)
        cur = torch.cuda.current_stream(self.device)
        for s in self._streams:
            s.wait_stream(cur)                             # inputs produced on the caller's stream are visible to the lanes
        for k, it in enumerate(items):
            self._in[k % self.n].put((k, it))
        out = [None] * len(items)
        for _ in items:
            idx, res, err = self._out.get()
            if err is not None:
                raise err
            out[idx] = res
        return out

    def close(self):
        for q in self._in:
            q.put(None)
        for t in self._threads:
            t.join(timeout=5)
