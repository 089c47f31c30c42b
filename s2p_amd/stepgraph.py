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
import gc

import torch


class StepGraph:
    def __init__(self, enabled=True):
        self.enabled = enabled
        self.seq = []              # ("graph", CUDAGraph) | ("call", fn)
        self._capturing = False
        self._cur = None
        self._pool = None
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
    def _begin_segment(self):
        g = torch.cuda.CUDAGraph()
        # "relaxed": a cut may come from autograd's worker thread (a hook inside a backward), i.e. the capture is ended /
        # begun by a different thread than the one that started it, and RCCL's watchdog thread polls events meanwhile
        g.capture_begin(pool=self._pool, capture_error_mode="relaxed")
        self._cur = g

    def _end_segment(self):
        self._cur.capture_end()
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
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            self._capturing = True
            try:
                self._begin_segment()
                step_fn()
                self._end_segment()
            finally:
                self._capturing = False
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
