"""Stream capture of a whole step into a hipGraph: the one place that knows how a capture has to be fenced on this stack.

  * error mode "thread_local": torch's default ("global") turns any OTHER thread's event query into a capture error;
    ProcessGroupNCCL's watchdog thread polls its work events continuously, so with a process group alive a capture would abort at
    random (seen with bench.py, round 4).
  * the cyclic garbage collector runs BEFORE the capture and is switched off DURING it.  torch 2.10's `torch.cuda.graph.__enter__`
    no longer collects (torch.compiler.config.force_cudagraph_gc), so a generation-0 collection can fire on any allocation inside the
    captured Python forward; if the cycle it frees owns device resources - an earlier trainer's hipGraphs, pinned staging buffers,
    events - their destructors run HIP calls that are not allowed while the thread's stream is capturing, and a failed HIP call in
    a destructor is an abort() (round 6: `Fatal Python error: Aborted ... Garbage-collecting` inside ModelTrainer's capture, one run
    in three of the trainer tests).
"""
import contextlib
import gc

import torch

CAPTURE_MODE = "thread_local"


@contextlib.contextmanager
def capture(graph, stream, pool=None):
    """`with capture(graph, stream): step()` - torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local") between a
    full collection and a pause of the collector."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        kw = {} if pool is None else {"pool": pool}
        with torch.cuda.graph(graph, stream=stream, capture_error_mode=CAPTURE_MODE, **kw):
            yield graph
    finally:
        if was:
            gc.enable()
