"""Data-parallel fine-tuning across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code; the few-shot fine-tune shards naturally:
clips are independent, every rank holds a replica of the frozen embedding and of the 18 507-parameter
head, and the only exchange per optimizer step is ONE all-reduce (sum) of the flat head gradient
(the 18 507 gradients plus the step's two loss/accuracy sums = 74 036 B), followed by the identical Keras-Adam update on every rank with grad_scale = 1/world.
At that size the collective is latency-bound (tens of microseconds over xGMI), so there is nothing
to bucket or overlap.
"""


def is_distributed():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:
        return False


def world_size():
    import torch.distributed as dist
    return dist.get_world_size() if is_distributed() else 1


def rank():
    import torch.distributed as dist
    return dist.get_rank() if is_distributed() else 0


def allreduce_sum_(t, force=False):
    """In-place sum over ranks (no-op single-process unless `force` and a process group exists).
    `t` may alias a handle's device buffer."""
    if is_distributed() or (force and _initialized()):
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def _initialized():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:
        return False


def broadcast_(t, src=0):
    if is_distributed():
        import torch.distributed as dist
        dist.broadcast(t, src=src)
    return t


def dp_step(head, emb, labels, lr, beta1=0.9, beta2=0.999, eps=1e-7, force_collective=None):
    """One data-parallel optimizer step on a head-like object (loss_grad / grad_view / adam_step):
    local gradient of the local-mean loss -> all-reduce(sum) -> Adam with grad_scale 1/world.
    Returns the stats tensor [sum of row losses, #correct] summed over ranks (a view of the reduced buffer,
    valid until the next loss_grad).  force_collective=True ALSO issues the all-reduce in a world of one
    (the single-GPU RCCL smoke test); it can never switch the collective off when there are several ranks."""
    stats = head.loss_grad(emb, labels)
    w = world_size()
    if w > 1 or force_collective:
        # ONE collective per step: the two statistics ride behind the gradient in the same flat buffer
        buf = allreduce_sum_(head.grad_view(with_stats=True), force=bool(force_collective))
        stats = buf[-2:]
    head.adam_step(lr=lr, beta1=beta1, beta2=beta2, eps=eps, grad_scale=1.0 / w)
    return stats
