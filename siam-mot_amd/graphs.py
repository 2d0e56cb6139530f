"""hipGraph capture of launch-bound loops of the EMM path.

A frame pair is four kernel launches of 7-25 us each, enqueued from Python through ctypes (about 40 us of host time
per frame pair against 63 us of GPU time at 30 tracks, bench.py ``per_rank.host_enqueue_us_per_step``).  With the
inputs in static buffers the launches of one or several frame pairs can be captured ONCE into a hipGraph and replayed
with a single ``hipGraphLaunch`` — the host then spends a few microseconds per replay, which is what eight ranks on
one host want.

``capture(fn)`` is the generic form (any callable that only enqueues work of this library / torch on the current
stream).  ``FramePairRing`` is the frame-pair loop of ``bench.py``: ``EMM.forward`` on frame k with the memory of frame
k-1, then ``EMM.extract_cache`` on frame k, over a ring of resident feature sets, captured as ONE graph per ring
revolution; the memory of the last step is copied back into the static buffers the first step reads, so replays chain
exactly like the eager loop.

Capture rules of this library (ops.py): the kernel timers must be off (their launches carry events), the first call of
every kernel must have happened before capture (LDS opt-in / weight packing run eagerly once) — ``capture`` runs the
callable ``warmup`` times first — and results live in the graph's private memory pool: read them after ``replay()`` +
synchronisation, they are overwritten by the next replay.
"""
import torch

from .emm import OrderHint


def capture(fn, warmup=2, device=None):
    """Run ``fn()`` ``warmup`` times eagerly on a side stream, then capture one call into a graph.
    Returns ``(graph, result_of_the_captured_call)``; ``graph.replay()`` re-runs the captured launches."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(max(1, warmup)):
            fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        out = fn()
    return graph, out


class FramePairRing(object):
    """``len(feature_sets)`` frame pairs as one graph: step k = ``emm(features[k], memory)`` + ``emm.extract_cache(
    features[k], detections)``; ``results[k]`` is the BoxList step k produced (static storage, valid after a replay)."""

    def __init__(self, emm, feature_sets, detections, memory):
        self.emm, self.feats, self.det = emm, list(feature_sets), detections
        z, sr, d = memory
        # static memory buffers the first step of every revolution reads
        self.z0, self.d0 = z.clone(), d
        self.sr0_bbox = sr[0].bbox.clone()
        self.sr0 = sr[0].__class__(self.sr0_bbox, sr[0].size, mode=sr[0].mode)
        for field in sr[0].fields():
            self.sr0.add_field(field, sr[0].get_field(field))
        # the extraction's order hint (emm.OrderHint) chains through the ring like the memory it describes — only when
        # the memory handed in was made from the very detections every step extracts from (the hint names rows of them)
        h = sr[0].__dict__.get("order_hint")
        self.hint0 = None
        if h is not None and h.boxes is d[0].bbox and d[0].bbox is detections.bbox and h.sr is sr[0].bbox:
            self.hint0 = h.data.clone()
            self.sr0.order_hint = OrderHint(self.hint0, d[0].bbox, self.sr0_bbox, h.scales)
        self.results = None
        self.graph, _ = capture(self._revolution)

    def _revolution(self):
        z, sr, d = self.z0, [self.sr0], self.d0
        results = []
        for f in self.feats:
            _, res, _ = self.emm(f, d, sr, template_features=z)
            results.append(res[0])
            z, sr, d = self.emm.extract_cache(f, self.det)
        self.z0.copy_(z)                        # close the ring: the next revolution starts from this memory
        self.sr0_bbox.copy_(sr[0].bbox)
        if self.hint0 is not None:
            self.hint0.copy_(sr[0].order_hint.data)
            self.sr0.order_hint.sr_version = self.sr0_bbox._version      # (this copy is the hint's own update)
        self.results = results
        return results

    def replay(self):
        self.graph.replay()
        return self.results
