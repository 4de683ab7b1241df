"""Data parallelism for the training hot path: one process per GPU, gradients all-reduced over RCCL/xGMI.

The reference uses single-process ``torch.nn.DataParallel`` (scripts/train.py:93-94): parameters re-broadcast every
step, the 17.8 MB/image logits gathered to GPU 0, gradients reduce-added to GPU 0.  Here every rank owns whole
multi-view groups, computes its loss locally, and the ONLY exchange is a bucketed gradient all-reduce that starts
while backward is still running.  xGMI is point-to-point (7 links x ~153 GB/s), so buckets are large
(default 32 MiB per bucket, one chain of buckets per gradient dtype: ResNet-50 with bf16 convolution-weight gradients leaves in
4 collectives) to stay bandwidth- not latency-bound.
"""
import os

import torch
import torch.distributed as dist

# How a completed bucket's gradients are packed into its flat buffer: "foreach" (torch._foreach_copy_ into the parameter-strided views; on the
# GPU the same loop in C++, csrc/torch_glue.cpp pack_bucket) or "cat" (torch.cat of memory-order flat aliases, written straight into the flat
# buffer).  Same bytes either way (tests/test_host_logic.py sets this attribute to compare them); "foreach" is the training path.
BUCKET_PACK = "foreach"


def _memory_order_flat(t):
    """A 1-D alias of a dense tensor in the order its elements lie in memory, or None if it is not dense."""
    if t.is_contiguous():
        return t.reshape(-1)
    order = sorted(range(t.dim()), key=lambda d: (-t.stride(d), -t.shape[d]))
    q = t.permute(order)
    return q.reshape(-1) if q.is_contiguous() else None


def init_from_env(backend=None, set_device=True, timeout_minutes=120):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local).
    ``timeout_minutes``: collective timeout -- generous by default, because rank 0 runs the whole validation and evaluation between
    epochs while the other ranks wait in a barrier (scripts/train.py), which the default watchdog would kill on a full H36M validation."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"      # "nccl" IS RCCL on ROCm
        if backend == "nccl" and set_device:
            torch.cuda.set_device(local)
        import datetime
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=timeout_minutes))
    return rank, world, local


def shard_groups(n_group_global, rank, world):
    """Whole multi-view groups per rank (never split a group's views): -> (first group, number of groups)."""
    base, rem = divmod(n_group_global, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def broadcast_module(module, src=0, optimizer=None):
    """One-time broadcast of parameters and buffers from rank ``src`` (replaces DataParallel's per-step replicate).  The
    tensors are written in place under ``no_grad`` (which bumps their version counters, so bf16 training copies notice); pass the
    optimizer when it already exists and its copies are refreshed right away."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src)
    if optimizer is not None and hasattr(optimizer, "refresh_training_copies"):
        optimizer.refresh_training_copies()


class _DirectHandle:
    """One bucket of the "direct" collective in flight (BucketedGradSync._start_collective): wait() finishes the scatter phase and reduces this rank's
    slice (fp32 sum in rank order -- the same result on every backend -- scaled by 1 / world BEFORE the one rounding back to the bucket's dtype),
    start_gather() / wait_gather() send it to every peer and collect theirs."""

    def __init__(self, flat, slices, recv, acc, works, me, n):
        self.flat, self.slices, self.recv, self.acc, self.works, self.me, self.n = flat, slices, recv, acc, works, me, n
        self.gather = []

    def wait(self):
        for w in self.works:
            w.wait()
        # slice by slice into ONE fp32 accumulator of slice size (1 / world of the bucket; round 5 materialised recv.float(): world x the slice in fp32 per
        # bucket and step) -- add_ promotes the bf16 operand inside the kernel, no copy is made
        acc = self.acc
        acc.zero_()
        for r in range(self.n):
            acc.add_(self.slices[self.me] if r == self.me else self.recv[r])
        acc.mul_(1.0 / self.n)
        self.slices[self.me].copy_(acc)

    def start_gather(self):
        ops = []
        for peer in range(self.n):
            if peer != self.me:
                ops.append(dist.P2POp(dist.isend, self.slices[self.me], peer))
                ops.append(dist.P2POp(dist.irecv, self.slices[peer], peer))
        self.gather = dist.batch_isend_irecv(ops)

    def wait_gather(self):
        for w in self.gather:
            w.wait()
        self.gather = []


class BucketedGradSync:
    """Average gradients across ranks with a few large all-reduces overlapped with backward.

    Parameters are packed (in reverse registration order ~ the order backward produces them) into flat buckets.  Before
    backward every ``.grad`` is ``None``, so autograd hands each gradient over without an accumulate kernel; when the last
    gradient of a bucket has arrived, ONE multi-tensor copy moves the bucket's gradients into the flat buffer, each
    parameter's ``.grad`` is re-pointed at its slice, and the bucket's all-reduce is launched asynchronously from that same
    autograd hook (overlap with the rest of backward).  ``finish()`` waits and makes the result the mean over ranks.
    Per step this costs a handful of launches per bucket instead of one accumulate kernel per parameter (161 for
    ResNet-50) plus the bucket memsets.  One backward per step (no gradient accumulation across backward calls).
    """

    def __init__(self, module, bucket_bytes=32 << 20, grad_dtype=None, optimizer=None, collective="allreduce"):
        # collective (xGMI is point-to-point, a ring is bound by ONE link):
        #   "allreduce"  one dist.all_reduce per bucket -- the library (RCCL) picks algorithm and channels (NCCL_ALGO / NCCL_PROTO pin them);
        #   "direct"     reduce-scatter + all-gather written as grouped point-to-point transfers: rank r sends slice j of the bucket straight to rank j
        #                (N - 1 concurrent sends, one per xGMI link), sums the N slices it owns in fp32, and sends the reduced slice straight to every peer.
        #                Each link carries S / N per phase instead of the ring's 2 (N - 1) S / N over one link.
        if collective not in ("allreduce", "direct"):
            raise ValueError("collective must be 'allreduce' or 'direct', got %r" % (collective,))
        if collective == "direct" and dist.is_initialized() and dist.get_backend() == "gloo" and any(p.is_cuda for p in module.parameters()):
            # gloo has no point-to-point transfers of device tensors (its all_reduce takes them): the direct form is for RCCL, or for CPU tensors on gloo (the tests)
            raise ValueError("collective='direct' needs the nccl (RCCL) backend for GPU tensors; gloo only moves CPU tensors point to point")
        self.collective = collective
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        copies = optimizer.training_copies() if hasattr(optimizer, "training_copies") else {}
        if copies:      # convolution weights trained through bf16 copies: their gradients live on the copies (bf16:
            # half the xGMI traffic); same parameter order as the optimizer's
            self.params = [copies.get(i, p) for i, p in enumerate(self.params)]
        self.buckets = []          # (flat tensor, [params], [views])
        self._direct_bufs = {}     # id(flat) -> (receive buffer, fp32 slice accumulator) of the "direct" collective
        self._pending = {}
        self._handles = []
        # sum in the collective, scale afterwards (one small kernel per bucket): ReduceOp.AVG would save those kernels on
        # RCCL but is not available on every backend / dtype combination
        self._avg = False
        # One OPEN bucket per gradient dtype, filled in reverse registration order (~ the order backward produces the
        # gradients) and closed when it reaches ``bucket_bytes``.  A dtype change does NOT close a bucket: with FusedAdam's bf16
        # training copies the network alternates conv (bf16) / BatchNorm (fp32) parameters layer by layer, and cutting at
        # every change produced 106 latency-bound collectives on ResNet-50 instead of a handful of bandwidth-bound ones.
        open_buckets = {}          # dtype -> ([params], bytes)
        order = []                 # closed buckets, in closing order
        for p in reversed(self.params):
            dt = grad_dtype or p.dtype
            nbytes = p.numel() * dt.itemsize
            cur, cur_bytes = open_buckets.get(dt, ([], 0))
            if cur and cur_bytes + nbytes > bucket_bytes:
                order.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            open_buckets[dt] = (cur, cur_bytes + nbytes)
        for dt, (cur, _) in open_buckets.items():
            if cur:
                order.append(cur)
        for plist in order:
            self._make_bucket(plist, grad_dtype)
        # Step 1 hooks every parameter and learns, per bucket, which gradient arrives last; from step 2 on only those
        # parameters keep a hook (a Python hook costs the autograd thread ~8 us: 161 of them made backward host-bound).
        # A launch that finds a gradient missing (the graph changed) is deferred to finish(), which is always correct.
        self._hooks = []
        self._launched = set()
        self._deferred = []        # complete buckets whose launch waits for the next hook (GPU: see _launch_from_hook)
        self._pipeline = True      # GPU: a complete bucket is launched one hook later, behind an event (see _launch_from_hook)
        self._last = {}
        self._learning = True
        self._declare_hook_free([])
        for bi, (_, plist, _) in enumerate(self.buckets):
            for p in plist:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self.zero_grad()

    def _declare_hook_free(self, params):
        """Tell the C++ glue which parameters carry no live hook (csrc/torch_glue.cpp declare_hook_free): a removed post-accumulate hook
        leaves its empty holder on the tensor, which the glue would otherwise read as "somebody consumes this gradient inside the pass"
        and keep the weight gradient on the main stream, one launch per layer -- measured 8.0 instead of 7.05 ms/step on the forced
        bucket path before this declaration existed."""
        self._declared_free = list(params)
        if any(p.is_cuda for _, plist, _ in self.buckets for p in plist):
            from . import hip
            hip.glue().declare_hook_free(self._declared_free)

    def _make_bucket(self, plist, grad_dtype):
        total = sum(p.numel() for p in plist)
        if self.collective == "direct" and self.world > 1:
            total = (total + self.world - 1) // self.world * self.world       # equal slices (the padding stays zero)
        flat = torch.zeros(total, dtype=grad_dtype or plist[0].dtype, device=plist[0].device)
        off, views = 0, []
        for p in plist:
            # a dense view with exactly the parameter's strides (channels-last weights included): the optimizer reads the
            # gradient in place, and the multi-tensor copy stays on its fast path
            view = torch.as_strided(flat, p.shape, p.stride(), storage_offset=off)
            views.append(view)
            off += p.numel()
        self.buckets.append((flat, plist, views))

    def _launch_from_hook(self, bi):
        """Inside backward: bucket ``bi`` is complete.  On the GPU the gradients of this bucket may still be in flight on the second
        stream (grouped weight gradients, deferred slab sums: csrc/torch_glue.cpp), so the bucket is not packed now: everything
        pending is enqueued there behind an event (no stream waits for anything) and the bucket is launched at the NEXT hook -- one
        bucket of backward later, when its event has long fired -- or by finish().  (Joining the streams at every hook instead cost
        5.4 % of the step, profiles/r03_bucket_path_overhead_a_join_per_bucket.txt.)"""
        flat = self.buckets[bi][0]
        if not flat.is_cuda or not self._pipeline:
            self._launch(bi)
            return
        from . import hip
        ticket = hip.glue().flush_pending_async()
        self._deferred.append((bi, ticket))
        while len(self._deferred) > 1:
            b, t = self._deferred.pop(0)
            hip.glue().wait_flush_ticket(t, flat.device.index)
            self._launch(b, flushed=True)

    def _launch(self, bi, flushed=False):
        flat, plist, views = self.buckets[bi]
        if flat.is_cuda and not flushed:
            # split weight gradients are summed by ONE launch at the end of backward (csrc/torch_glue.cpp); a bucket that leaves
            # earlier needs the sums of what has been computed so far now
            from . import hip
            hip.glue().flush_pending_reduces()
        if BUCKET_PACK == "cat" and self._pack_cat(flat, plist, views):
            self._finish_launch(bi, flat, plist, views)
            return
        if flat.is_cuda and BUCKET_PACK == "foreach":
            from . import hip
            hip.glue().pack_bucket(plist, views)         # the loop below, in C++ (0.3 ms of Python per step otherwise)
            self._launched.add(bi)
            self._start_collective(flat)
            return
        dst, src = [], []
        for p, v in zip(plist, views):
            g = p.grad
            if g is None:
                v.zero_()                    # parameter without a gradient this step
            elif g.data_ptr() != v.data_ptr():
                if g.stride() != v.stride():
                    if g.shape == v.shape and all(a == b for n, a, b in zip(g.shape, g.stride(), v.stride()) if n > 1):
                        # same memory order, only the strides of extent-1 dimensions differ (a 1x1 convolution weight's
                        # gradient comes back "contiguous", the parameter is "channels_last"): re-stride the alias, so
                        # that the multi-tensor copy keeps its fast path (one mismatch sends the WHOLE list down the
                        # one-kernel-per-tensor route)
                        g = g.as_strided(g.shape, v.stride(), g.storage_offset())
                    else:
                        v.copy_(g)           # genuinely different layout: its own strided copy
                        continue
                dst.append(v)
                src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)
        self._finish_launch(bi, flat, plist, views)

    def _pack_cat(self, flat, plist, views):
        """torch.cat of the gradients' memory-order aliases into the flat buffer.  False (nothing written) when a gradient is
        missing, already lives in the bucket, or does not have its parameter's memory order."""
        parts = []
        for p, v in zip(plist, views):
            g = p.grad
            if g is None or g.data_ptr() == v.data_ptr() or g.dtype != flat.dtype or g.shape != v.shape:
                return False
            if not all(a == b for n, a, b in zip(g.shape, g.stride(), v.stride()) if n > 1):
                return False
            f = _memory_order_flat(g)
            if f is None:
                return False
            parts.append(f)
        torch.cat(parts, out=flat)
        return True

    def _finish_launch(self, bi, flat, plist, views):
        for p, v in zip(plist, views):
            p.grad = v
        self._launched.add(bi)
        self._start_collective(flat)

    def _start_collective(self, flat):
        if self.world <= 1:
            return
        if self.collective == "allreduce":
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            self._handles.append(dist.all_reduce(flat, op=op, async_op=True))
            return
        # direct, phase 1 (scatter): slice j of my bucket -> rank j; the slices of MY index arrive from everybody
        n, me = self.world, dist.get_rank()
        slices = flat.view(n, -1)
        # receive buffer (bucket size, bucket dtype) and the fp32 accumulator of this rank's slice: allocated once per bucket, reused every step
        bufs = self._direct_bufs.get(id(flat))
        if bufs is None:
            bufs = self._direct_bufs[id(flat)] = (torch.empty_like(slices), torch.empty(slices.shape[1], dtype=torch.float32, device=flat.device))
        recv, acc = bufs
        ops = []
        for peer in range(n):
            if peer != me:
                ops.append(dist.P2POp(dist.isend, slices[peer], peer))
                ops.append(dist.P2POp(dist.irecv, recv[peer], peer))
        self._handles.append(_DirectHandle(flat, slices, recv, acc, dist.batch_isend_irecv(ops), me, n))

    def _make_hook(self, bi):
        def hook(param):
            if self._learning:
                self._pending[bi] -= 1
                if self._pending[bi] == 0:
                    self._last[bi] = param
                    self._launch_from_hook(bi)
            elif all(p.grad is not None for p in self.buckets[bi][1]):
                self._launch_from_hook(bi)   # the learned last arrival, and indeed every gradient is here
        return hook

    def zero_grad(self):
        """Call before backward(): drops every ``.grad`` (autograd then hands gradients over instead of accumulating)."""
        if self.buckets and self.buckets[0][0].is_cuda:
            from . import hip
            hip.glue().clear_grads([p for _, plist, _ in self.buckets for p in plist])
        else:
            for _, plist, _ in self.buckets:
                for p in plist:
                    p.grad = None
        self._pending = {bi: len(plist) for bi, (_, plist, _) in enumerate(self.buckets)}
        self._handles = []
        self._launched = set()
        self._deferred = []
        self._recheck_hook_free()

    def _recheck_hook_free(self):
        """A parameter declared hook-free must STAY hook-free: if somebody registered a post-accumulate hook on one of them since (an
        optimizer-in-backward, a user hook), the glue would still leave its gradient on the second stream / unreduced and that hook would read
        an unfinished tensor.  Checked before every backward pass (a dict-length test per parameter); a parameter that grew a live hook is
        withdrawn from the declaration and takes the safe path from this pass on."""
        declared = getattr(self, "_declared_free", None)
        if not declared:
            return
        still = [p for p in declared if not getattr(p, "_post_accumulate_grad_hooks", None)]
        if len(still) != len(declared):
            self._declare_hook_free(still)

    def finish(self):
        """Wait for the in-flight all-reduces; gradients become the mean over ranks.  Call between backward() and
        optimizer.step()."""
        # (backward() has returned: the end-of-pass callback of the glue has joined the streams, every gradient is final)
        for b, _ in self._deferred:                   # complete buckets still waiting for their turn
            self._launch(b, flushed=True)
        self._deferred = []
        for bi in range(len(self.buckets)):           # buckets whose hook did not fire / found a gradient missing
            if bi not in self._launched:
                self._launch(bi)
        for h in self._handles:
            h.wait()
        for h in self._handles:               # direct form: the all-gather phases of all buckets, started together, then awaited
            if isinstance(h, _DirectHandle):
                h.start_gather()
        for h in self._handles:
            if isinstance(h, _DirectHandle):
                h.wait_gather()
        self._handles = []
        if self.world > 1 and not self._avg and self.collective != "direct":      # (the direct form averaged in fp32 before its one rounding)
            inv = 1.0 / self.world
            for flat, _, _ in self.buckets:
                flat.mul_(inv)
        if self._learning and len(self._last) == len(self.buckets):
            self._learning = False
            for h in self._hooks:
                h.remove()
            self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(bi)) for bi, p in self._last.items()]
            last = {id(p) for p in self._last.values()}
            self._declare_hook_free([p for _, plist, _ in self.buckets for p in plist if id(p) not in last])

    def total_bytes(self):
        return sum(f.numel() * f.element_size() for f, _, _ in self.buckets)
