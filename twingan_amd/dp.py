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
  """Sum all-reduce of one flat gradient buffer across clones, in ``n_buckets`` async pieces."""

  def __init__(self, world_size=1, process_group=None, n_buckets=2, always=False):
    """``always``: issue the collectives for a single clone too (tools/rccl_smoke.py: exercises RCCL and the
    segment / all-reduce interleaving on a one-GPU box; a one-rank sum is the identity)."""
    self.world = int(world_size)
    self.pg = process_group
    self.n_buckets = n_buckets
    self.always = always
    self._pending = []

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
    return len(bounds)

  def finish(self):
    """Makes the reduced gradients visible to the stream Adam is enqueued on."""
    for w in self._pending:
      w.wait()
    self._pending = []

  def allreduce(self, flat_grad):
    self.start(flat_grad)
    self.finish()
    return flat_grad
