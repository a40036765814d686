"""Pipelined inference: consecutive forwards of ``VoxelNetwork_depth`` issued round-robin on several HIP streams.

One forward is a chain of ~170 launches whose big members (the persistent 3-D convolutions: one 512-thread workgroup per CU, all of
its registers and most of its LDS) own the chip while they run, but 2.1 ms of it is the MIOpen 2-D backbone and another ~1 ms are
the small pyramid levels and the tail - launches that leave most CUs idle.  Batches are independent (the reference evaluates them
one after the other, ``network/voxel_net_depth.py:244-275`` has no state across calls), so the backbone of batch i+1 can run in
those gaps of batch i: two streams give 717 -> 770 frames/s at B = 8 on one MI355X (``tools/diag/two_stream.py``; round 3, with the
shorter 7^3 front layer: 767 on one stream, 819 on two, 831 on three - six alternating runs on one box, 815 on four - so ``bench.py``
issues on three; putting each forward's backbone on a low-priority stream and its 3-D part on a high-priority one measured
slower, 753 vs 780; replaying each replica as a hipGraph adds ~1 %, ``tools/diag/two_stream_graphs.py``; capping the persistent kernels to half the
CUs so that the two forwards' big kernels run side by side changes nothing, 772.7 vs 772.4).

Every stream gets its own module replica: the replicas alias the parameters and constant tables of the first module (no second copy
of the weights as nn.Parameters), but pack their own kernels' weights and own their scratch, input caches and output buffers, so two
forwards in flight never share a mutable buffer.  Results equal those of the plain forward to its own run-to-run reproducibility (the
MIOpen backbone splits K with atomic adds in some layers: ~6e-6 m in the joints between any two runs, tools/diag/stream_determinism.py).

Stream ordering (what PyTorch's single-stream semantics would have given for free is restated here explicitly):
  * inputs: by default the pipeline stream WAITS for everything the caller's current stream has queued at call time (an event is
    recorded there), so a non-blocking upload or a pre-processing kernel issued just before the call is complete before the forward
    reads it.  ``inputs_ready=<event>`` waits for that event instead; ``inputs_ready=False`` skips the wait (the caller guarantees
    the inputs are complete - what a throughput loop over resident inputs wants: a consumer that makes its stream wait for batch i
    would otherwise also hold back batch i+1).
  * every tensor argument is ``record_stream``-ed on the pipeline stream, so the caching allocator does not hand its block to someone
    else while the forward still reads it, even if the caller drops the tensor right after the call.
  * outputs are allocated on the pipeline stream.  A consumer on another stream must (1) ``wait_event(done)`` and (2) keep ``out``
    alive until its own reads are queued AND call ``t.record_stream(consumer_stream)`` on what it reads (``PipelinedForward.hand_over``
    does both for the current stream); after ``done.synchronize()`` on the host neither is needed.
"""
from __future__ import annotations

import copy

import torch


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            yield from _tensors(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors(o)


class PipelinedForward:
    """``pf = PipelinedForward(net, n_streams=2); out, done = pf(img, ..., depth_map_batch=depth)``.

    ``out`` is what ``net(...)`` returns, produced on one of the pipeline's streams; ``done`` is a ``torch.cuda.Event`` recorded
    behind it.  See the module docstring for the ordering rules on both sides."""

    def __init__(self, net, n_streams: int = 2):
        if n_streams < 1:
            raise ValueError("n_streams must be >= 1")
        p = next(net.parameters())
        if not p.is_cuda:
            raise RuntimeError("PipelinedForward needs the module on a HIP device")
        self.nets = [net]
        for _ in range(n_streams - 1):
            # the compiled state (packed V2V program with its split-K workspace, folded backbone, graphs, input caches) is per replica
            # and rebuilt below: detach it for the copy instead of duplicating several hundred MB that would be dropped right after
            held = (net.volume_net._program, net._folded, net._graphs, net._xbuf)
            net.volume_net._program, net._folded, net._graphs, net._xbuf = None, None, {}, {}
            try:
                rep = copy.deepcopy(net)
            finally:
                net.volume_net._program, net._folded, net._graphs, net._xbuf = held
            for a, b in zip(rep.parameters(), net.parameters()):
                a.data = b.data                                   # alias, do not duplicate
            for a, b in zip(rep.buffers(), net.buffers()):
                a.data = b.data
            # a replica of an already compiled module must pack its own kernels: the copied program would share nothing mutable, but
            # repacking from the aliased parameters keeps every per-program table (fused skip weights, workspaces) consistent
            rep._invalidate()
            rep.volume_net._program = None
            self.nets.append(rep.eval())
        self.streams = [torch.cuda.Stream(device=p.device) for _ in range(n_streams)]
        self._next = 0

    def __len__(self):
        return len(self.nets)

    def enable_graphs(self, flag: bool = True):
        """Replay every replica's forward as a captured hipGraph (``VoxelNetwork_depth.enable_graphs``).  At batch 1 the eager pipeline is
        bound by the host's launch rate (~175 launches per frame from one Python thread): three streams give 357 frames/s eager and 567
        with graph replay on one MI355X (tools/diag/streams_graphs_sweep.py; batch 8: 910 / 915).  A replayed forward returns the replica's
        STATIC output tensors: they are overwritten by that replica's next call (``len(self)`` calls later) - consume or copy them before."""
        for n in self.nets:
            n.enable_graphs(flag)
        return self

    @torch.no_grad()
    def __call__(self, *args, inputs_ready=None, **kwargs):
        k = self._next
        self._next = (k + 1) % len(self.nets)
        s = self.streams[k]
        if inputs_ready is None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(s.device))
            s.wait_event(ev)
        elif inputs_ready is not False:
            s.wait_event(inputs_ready)
        for t in _tensors((args, kwargs)):
            if t.is_cuda:
                t.record_stream(s)
        with torch.cuda.stream(s):
            out = self.nets[k](*args, **kwargs)
            done = torch.cuda.Event()
            done.record(s)
        return out, done

    @staticmethod
    def hand_over(out, done, stream=None):
        """Make ``stream`` (default: the current one) the consumer of ``out``: it waits for ``done`` and every tensor of ``out`` is
        recorded on it, so the blocks are not recycled under its reads."""
        stream = stream or torch.cuda.current_stream()
        stream.wait_event(done)
        for t in _tensors(out):
            if t.is_cuda:
                t.record_stream(stream)
        return out

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
