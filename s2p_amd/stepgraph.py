"""Segmented hipGraph capture of a train step.

A train step is ~430 short kernel launches; replaying it from a hipGraph removes the host launch cost.  With data
parallelism the step also contains collectives (RCCL), event records and cross-stream waits that must stay OUTSIDE the
captured graphs -- and they sit in the middle of the step (the gradient all-reduce of a bucket is issued while the rest
of the backward still runs).  `StepGraph` lets the step function mark those points itself:

    sg = StepGraph()
    def step():
        ...kernels...                      # captured
        sg.cut(lambda: launch_allreduce()) # graph ends here; the action runs eagerly; a new graph begins
        ...kernels...                      # captured
    sg.capture(step)                       # runs `step` once: captures the segments, runs the actions for real
    sg.replay()                            # graph 0, action 0, graph 1, action 1, ...

Outside `capture` (eager mode, or `enabled=False`) `cut(action)` simply calls `action()`, so the same step function
serves eager training, warm-up and capture.  All segments share one memory pool: tensors produced in one segment
and consumed in a later one (activations saved for backward, gradient buffers) keep their addresses across replays.
"""
import ctypes
import gc

import torch

_ACTIVE = None      # the StepGraph that is capturing right now (one per process)
CAPTURE_PRIORITY = 0   # priority of the stream the step is captured on (-1: above the side streams); measured: see DESIGN.md 6c


def capturing_now():
    """True while a StepGraph of this process is recording a segment (every side stream must re-join before a cut then)."""
    sg = _ACTIVE
    return sg is not None and sg.capturing


def fork(side, main=None):
    """`side.wait_stream(main)` for a side stream that is about to receive work, with the one rule hipGraph capture imposes
    on this code base checked: while a step is being captured every fork must start from the CAPTURING stream.  A side stream
    forked from another side stream made hipStreamEndCapture abort the process (ROCm 7.2) -- here it is a
    Python error at the fork instead."""
    main = main if main is not None else torch.cuda.current_stream()
    sg = _ACTIVE
    if sg is not None and sg.capturing and side != main and main != sg._stream:
        raise RuntimeError("stream fork from a side stream while a hipGraph segment is being captured: fork from the capturing "
                           "stream only (run the forking code on the main stream, or join the outer side stream first)")
    side.wait_stream(main)


class StepGraph:
    def __init__(self, enabled=True):
        self.enabled = enabled
        self.seq = []              # ("graph", CUDAGraph) | ("call", fn)
        self._capturing = False
        self._cur = None
        self._pool = None
        self._stream = None        # the stream the segments are captured on
        self.captured = False

    # ---- used by the step function ---------------------------------------------------------------------
    def cut(self, action):
        """End the current graph segment, run `action` eagerly (and on every replay), start the next segment."""
        if not self._capturing:
            action()
            return
        self._end_segment()
        action()
        self.seq.append(("call", action))
        self._begin_segment()

    # ---- capture / replay ----------------------------------------------------------------------------------
    @staticmethod
    def _node_count(g):
        """Nodes of a captured (not yet instantiated) graph, asked from the HIP runtime; None when it cannot be asked."""
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            n = ctypes.c_size_t(0)
            rc = hip.hipGraphGetNodes(ctypes.c_void_p(g.raw_cuda_graph()), None, ctypes.byref(n))
            return int(n.value) if rc == 0 else None
        except Exception:
            return None

    def _begin_segment(self):
        # keep_graph: capture_end leaves the hipGraph un-instantiated, so that an empty segment can be recognised (and dropped)
        # by COUNTING ITS NODES -- no warning text to parse, no process-global warning state touched from autograd's worker thread
        g = torch.cuda.CUDAGraph(keep_graph=True)
        # "relaxed": a cut may come from autograd's worker thread (a hook inside a backward), i.e. the capture is ended /
        # begun by a different thread than the one that started it, and RCCL's watchdog thread polls events meanwhile
        g.capture_begin(pool=self._pool, capture_error_mode="relaxed")
        self._cur = g

    def _end_segment(self):
        # two cuts back to back (or a cut as the last thing of the step) leave a segment without a single node: it is dropped,
        # so a replay does not pay a graph launch + host round trip for nothing.  A segment whose node count cannot be read is kept.
        self._cur.capture_end()
        n = self._node_count(self._cur)
        if n is None or n > 0:
            self._cur.instantiate()
            self.seq.append(("graph", self._cur))
        self._cur = None

    def capture(self, step_fn):
        if not self.enabled:
            step_fn()
            return self
        self.seq = []
        torch.cuda.synchronize()
        gc.collect()
        self._pool = torch.cuda.graph_pool_handle()
        global _ACTIVE
        stream = torch.cuda.Stream(priority=CAPTURE_PRIORITY)
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            self._capturing = True
            self._stream = stream
            _ACTIVE = self
            try:
                self._begin_segment()
                step_fn()
                self._end_segment()
            finally:
                self._capturing = False
                _ACTIVE = None
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        self.captured = True
        return self

    def replay(self):
        for kind, obj in self.seq:
            if kind == "graph":
                obj.replay()
            else:
                obj()

    @property
    def capturing(self):
        """True while `capture` is recording: kernels issued now are recorded, not run, so an eager action that mutates
        state from their results (an optimizer step on freshly reduced gradients) must skip that part."""
        return self._capturing

    @property
    def n_graphs(self):
        return sum(1 for k, _ in self.seq if k == "graph")
