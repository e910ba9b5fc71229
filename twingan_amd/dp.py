"""Data-parallel clone reduction: the reference's only "collective".

deployment/model_deploy.py builds ``num_clones`` towers on one host, divides every clone's loss by
``num_clones`` (:265-268) and sums the per-variable gradients with ``tf.add_n`` on the optimiser
device (:473-503).  Here a clone is one process on one MI355X; because every variable of an
optimiser group lives in ONE flat fp32 gradient buffer (params.ParamStore), the whole ``add_n``
fan-in is a handful of large sum all-reduces (RCCL over xGMI when the backend is "nccl"; gloo on
CPU in the tests).  Buckets are issued asynchronously -- the trainer calls start() once per backward
segment, on the range of the flat buffer whose gradients that segment completed (params.grad_phase), and keeps
enqueuing the next segment's backward kernels while the sum is in flight -- and waited for just before Adam.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of the ~35 MB group buffer is
per-link bound (~0.4 ms), so few large buckets beat many small ones -- the default is 2 buckets.
"""
import torch
import torch.distributed as dist


def loss_scale_for_clones(loss_scale, world):
  """model_deploy.py:265-268 (clone loss / num_clones) folded with the static mixed-precision loss
  scale of model_inheritor.py:568-570: the scalar every rank multiplies its loss by before backward."""
  return float(loss_scale) / float(world)


def bucket_bounds(numel, n_buckets, align=64):
  """Splits [0, numel) into <= n_buckets contiguous ranges whose starts are multiples of ``align``."""
  n_buckets = max(1, int(n_buckets))
  per = (numel + n_buckets - 1) // n_buckets
  per = (per + align - 1) // align * align
  out, lo = [], 0
  while lo < numel:
    hi = min(numel, lo + per)
    out.append((lo, hi))
    lo = hi
  return out


class GradReducer:
  """Sum all-reduce of one flat gradient buffer across clones, in ``n_buckets`` async pieces.

  Diagnostics for the first run on real xGMI (``stats()``): bytes handed to the collective and, per finish(), the time
  the compute stream sat waiting for it -- HIP events around the wait on a GPU (resolved lazily, no host sync inside a
  step), wall clock on the CPU backends.  That wait is the EXPOSED part of the all-reduce: with the segmented backward
  only the last, small range should show up in it."""

  def __init__(self, world_size=1, process_group=None, n_buckets=2, always=False):
    """``always``: issue the collectives for a single clone too (tools/rccl_smoke.py: exercises RCCL and the
    segment / all-reduce interleaving on a one-GPU box; a one-rank sum is the identity)."""
    self.world = int(world_size)
    self.pg = process_group
    self.n_buckets = n_buckets
    self.always = always
    self._pending = []
    self.reset_stats()

  def reset_stats(self):
    self._bytes = 0
    self._collectives = 0
    self._finishes = 0
    self._waits = []        # (start event, end event) on a GPU, seconds (float) on the CPU

  @property
  def active(self):
    return self.world > 1 or self.always

  def start(self, flat_grad, n_buckets=None):
    """Enqueues the bucketed all-reduce of ``flat_grad`` (a 1-D view of a flat gradient buffer; no-op for a single
    clone) behind everything already enqueued on the current stream.  May be called several times before finish().
    Returns the number of buckets issued."""
    if not self.active:
      return 0
    if not dist.is_initialized():
      raise RuntimeError('world_size %d but torch.distributed is not initialised' % self.world)
    bounds = bucket_bounds(flat_grad.numel(), n_buckets or self.n_buckets)
    for lo, hi in bounds:
      self._pending.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
    self._bytes += flat_grad.numel() * flat_grad.element_size()
    self._collectives += len(bounds)
    self._device = flat_grad.device
    return len(bounds)

  def finish(self):
    """Makes the reduced gradients visible to the stream Adam is enqueued on."""
    if not self._pending:
      return
    on_gpu = self._device.type == 'cuda'
    if on_gpu:
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
    else:
      import time
      t0 = time.perf_counter()
    for w in self._pending:
      w.wait()
    if on_gpu:
      e1.record()
      self._waits.append((e0, e1))
      if len(self._waits) > 4096:      # a long training run: keep the tail
        del self._waits[:2048]
    else:
      self._waits.append(time.perf_counter() - t0)
    self._finishes += 1
    self._pending = []

  def stats(self):
    """dict(allreduce_bytes, collectives, finishes, exposed_allreduce_ms = mean per finish(), exposed_max_ms) since the
    last reset_stats().  Synchronises the device (call it outside the timed region)."""
    ms = []
    for w in self._waits:
      if isinstance(w, tuple):
        w[1].synchronize()
        ms.append(w[0].elapsed_time(w[1]))
      else:
        ms.append(1e3 * w)
    return dict(allreduce_bytes=int(self._bytes), collectives=int(self._collectives), finishes=int(self._finishes),
                exposed_allreduce_ms=(sum(ms) / len(ms)) if ms else 0.0, exposed_max_ms=max(ms) if ms else 0.0)

  def allreduce(self, flat_grad):
    self.start(flat_grad)
    self.finish()
    return flat_grad
