"""Several independent depth maps in flight on one GPU.

The reference's ``inference.py`` runs its loader one reference view at a time; every forward here is ~200 dependent kernel
launches, and each launch ends with a partially filled last wave of tiles (the z|r gate convolution: 925 tiles on 512 block
slots - 99 of the 256 CUs idle for the last fifth of the launch).  Two forwards on two HIP streams fill each other's tails:
measured at DTU 1600x1184 x 10 views x 32 iterations 18.3 ms per depth map against 20.8 ms one at a time (tools/archive/exp_streams.py;
three streams: 18.0), outputs bit-identical.  Default three: with the fp8-correction convs 16.4-16.5 ms against 16.7 with two (four:
17.1, six: 16.7 - an even number of forwards tends to run in phase; tools/archive/exp_streams2.py).  ``DepthMapPipeline`` keeps ``streams`` forwards in flight: one replica of the model
per stream (a deep copy - the packed weights, feature buffers and workspaces of a forward are per replica, the library itself is
stateless; ``refresh_weights`` re-synchronises the replicas after the original's parameters changed), round-robin submission from
one host thread."""
import collections
import copy

import torch


class DepthMapPipeline:
    """``groups`` (round 5, sharded models - ``RAFT(view_group=...)``): one torch.distributed process group PER replica, all spanning the same
    ranks (``[dist.new_group(ranks) for _ in range(streams)]``, created in the same order on every rank).  Depth map i then runs its
    collectives - the feature all-gather / halo exchange of shard="slab", the volume all-reduce of shard="views" - on communicator
    i mod streams: collectives of different depth maps never interleave on one communicator (on RCCL each group also has its own
    internal stream, so they overlap), and every rank submits the same sequence, so per communicator the order is the same everywhere.
    At G = 8 a slab conv covers 175-325 tiles on 512 block slots: that is where forwards in flight pay most."""

    def __init__(self, model, streams=3, device=None, groups=None):
        if streams < 1:
            raise ValueError("DepthMapPipeline: streams must be >= 1")
        dev = device if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("DepthMapPipeline: the model must live on a GPU (there is no CPU path)")
        self.device = dev
        sharded = getattr(model, "view_group", None) is not None
        if sharded and streams > 1:
            if groups is None or len(groups) != streams:
                raise ValueError("DepthMapPipeline: a sharded model (view_group) with several depth maps in flight needs `groups`: one process "
                                 "group per replica (collectives of different depth maps must not share a communicator)")
        elif groups is not None and not sharded:
            raise ValueError("DepthMapPipeline: `groups` is for models built with a view_group")
        self.models = [model]
        for k in range(1, streams):
            vg = getattr(model, "view_group", None)
            if sharded:
                model.view_group = None                    # (a ProcessGroup is not copyable: the replica gets its own below)
                model.__dict__.pop("_slab_ex", None)       # (nor is the exchange object that holds one; it is rebuilt on the next forward)
            try:
                rep = copy.deepcopy(model)
            finally:
                if sharded:
                    model.view_group = vg
            if sharded:
                rep.view_group = groups[k]
            self.models.append(rep)
        if sharded and groups is not None:
            model.view_group = groups[0]
        for m in self.models:
            m.eval()
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(streams)]
        self._next = 0
        self._closed = False

    def __len__(self):
        return len(self.streams)

    def refresh_weights(self):
        """Copy the first model's parameters into the replicas (after fine-tuning / load_state_dict on the original); the replicas'
        packed weights follow through RAFT's own parameter-version check on their next forward."""
        sd = self.models[0].state_dict()
        for m in self.models[1:]:
            m.load_state_dict(sd)

    def submit(self, images, poses, intrinsics, scale, **kw):
        """Enqueue one test-mode forward on the next stream; returns a handle for ``result``.  The inputs must stay alive and
        unmodified until the result has been taken."""
        if self._closed:
            raise RuntimeError("DepthMapPipeline: submit() after close()")
        k = self._next % len(self.streams)
        self._next += 1
        st = self.streams[k]
        m = self.models[k]
        if k and getattr(m, "gru_precision", None) == "auto" and m._auto_pending():
            # one arithmetic form for all replicas: the first model's calibration decides.  If that model has not calibrated these weights
            # yet (first submission, or right after refresh_weights), this submission goes to it
            if not m.adopt_precision(self.models[0]) and self.models[0]._auto_pending():
                k = 0
                st = self.streams[0]
        st.wait_stream(torch.cuda.current_stream(self.device))       # inputs produced on the caller's stream
        with torch.cuda.stream(st), torch.no_grad():
            out = self.models[k](images, poses, intrinsics, scale=scale, **kw)
            done = torch.cuda.Event()
            done.record(st)
        for t in (images, poses, intrinsics):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(st)
        return out, done, st

    @staticmethod
    def result(handle, wait_on_host=True):
        """The disparity of a submitted forward.  ``wait_on_host``: block until it is ready (read-back follows); otherwise only the
        caller's current stream is made to wait for it."""
        out, done, st = handle
        if wait_on_host:
            done.synchronize()
        else:
            torch.cuda.current_stream(out.device).wait_event(done)
        out.record_stream(torch.cuda.current_stream(out.device))
        return out

    def map(self, items, **kw):
        """items: iterable of (images, poses, intrinsics, scale) -> yields the disparities in order, ``len(self)`` forwards in flight."""
        pending = collections.deque()
        for it in items:
            pending.append(self.submit(*it, **kw))
            if len(pending) >= len(self.streams):
                yield self.result(pending.popleft())
        while pending:
            yield self.result(pending.popleft())

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def check_overflow(self, raise_error=True):
        return self.models[0].check_overflow(self.device, raise_error=raise_error)

    def poll_overflow(self):
        """Bits of the last completed asynchronous flag snapshot (RAFT.overflow_policy "lazy"), 0 if clean or none is ready: never blocks."""
        from . import ops
        return ops.overflow_poll(self.device) or 0

    def close(self):
        """Drop the replicas and the per-stream workspaces of the cost volume (ops._LINES_WS holds ~300 MB per stream at DTU size)."""
        from . import ops
        for st in self.streams:
            ops.release_lines_workspace(self.device, st)
        self.models = self.models[:1]
        self.streams = self.streams[:1]                    # (len(pipe) and the stream list follow the models)
        self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
