"""Data-parallel training of the background NeRF over a process group -- the reference's one multi-GPU feature: train.py:26-28 wraps both
background nets in nn.DataParallel (the ray dimension of every network call is scattered over the visible GPUs, the outputs gathered, the
loss formed on GPU 0 and the gradients summed there), i.e. one optimiser step on the FULL batch with the network work split by rays.

Here: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the tests).  Every rank draws the same batch
(same seed) and keeps the rays rank, rank + world, ... of it (ray_batches.BackgroundRayBatcher(rank=, world=), or shard_batch below), runs
the existing iteration on them with the loss terms normalised by the GLOBAL element counts (so that the SUM over the ranks of the local
losses is the full-batch loss of trainers/vanilla_nerf_trainer.py:45-96, and the sum of the local gradients its gradient), and the
gradients of both nets travel as ONE flat float32 buffer through ONE all_reduce per iteration (2 x 595,844 floats = 4.8 MB, with the four
loss values riding in its tail) before Adam.  Weights start equal (broadcast_parameters) and stay equal: every rank applies the same step.
A tiny all_gather (six floats per rank: element counts and the largest densities) precedes the backward pass: the normalisers and the
dead-network test of :88-94 need the whole batch.
"""
import torch
import torch.distributed as dist


def rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_batch(batch, rank, world):
    """rays rank, rank + world, ... of a batch dict (tensors whose first dimension is the ray count; everything else passes through)"""
    if world == 1:
        return batch
    n = batch['origin'].shape[0]
    return {k: (v[rank::world].contiguous() if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n else v) for k, v in batch.items()}


def _collective_device(t):
    """gloo moves host tensors: ranks sharing one GPU in the tests reduce through the host"""
    return t.cpu() if (t.is_cuda and dist.get_backend() == "gloo") else t


def all_gather_floats(values, group=None, on_device=False):
    """[world, len(values)] float64: each rank's small vector of python / 0-d tensor numbers through ONE collective -- on the host (one read-back),
    or with on_device left where the values live (no host synchronisation: what the trainer's iteration uses between its forward and backward pass)"""
    rank, world = rank_world(group)
    dev = values_device(values)
    mine = torch.stack([torch.as_tensor(v, dtype=torch.float64).reshape(()).to(dev) for v in values])
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return mine[None] if on_device else mine[None].cpu()
    send = _collective_device(mine)
    out = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(out, send, group=group)
    got = torch.stack(out)
    return got.to(dev) if on_device else got.cpu()


def values_device(values):
    for v in values:
        if torch.is_tensor(v):
            return v.device
    return torch.device('cpu')


def broadcast_parameters(modules, src=0, group=None):
    """every parameter and buffer of `modules` <- rank src's (after construction and after a dead-network re-initialisation)"""
    if not (dist.is_available() and dist.is_initialized()):
        return
    tensors = [t for m in modules if m is not None for t in list(m.parameters()) + list(m.buffers())]
    if not tensors:
        return
    with torch.no_grad():
        flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
        buf = _collective_device(flat)
        dist.broadcast(buf, src=src, group=group)
        flat = buf.to(flat.device)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
            off += n


def broadcast_optimizer_state(optim, src=0, group=None):
    """rank src's optimiser state (Adam's moments, step counts, the groups' hyper-parameters) on every rank: after a resume() the ranks may have
    read different or missing checkpoints, and parameters that agree step apart again if the moments do not.  One object broadcast (the state
    is the size of the parameters; this runs once per resume, not per iteration)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    rank, _ = rank_world(group)
    box = [optim.state_dict() if rank == src else None]
    if rank == src:
        box[0] = {'state': {k: {n: (v.cpu() if torch.is_tensor(v) else v) for n, v in st.items()} for k, st in box[0]['state'].items()},
                  'param_groups': box[0]['param_groups']}
    dist.broadcast_object_list(box, src=src, group=group)
    if rank != src:
        optim.load_state_dict(box[0])                           # (load_state_dict moves the tensors to the parameters' device)


def decorrelate_device_rng(rank):
    """The ranks of a data-parallel trainer are seeded alike so that they draw the same batches; the per-SAMPLE draws of an iteration (the
    stratified jitter of sample_z, raw2outputs' density noise) come from the device's global generator and would then repeat rank 0's pattern on
    every rank's shard.  Re-seed that generator with seed + rank: independent jitter per ray across the whole batch, as in a single process."""
    if torch.cuda.is_available() and rank > 0:
        torch.cuda.manual_seed((torch.cuda.initial_seed() + 7919 * rank) % (1 << 63))


class GradSync:
    """The gradients of `params` as views of one flat float32 buffer (+ `n_extra` trailing floats), summed over the ranks by ONE all_reduce.

    After attach() every p.grad IS its slice of the buffer, so autograd accumulates straight into it (zero with zero(), or
    optimizer.zero_grad(set_to_none=False)); a p.grad that was replaced meanwhile (zero_grad's default set_to_none) is copied in by reduce()."""

    def __init__(self, params, n_extra=0, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.n_grad = sum(self.sizes)
        self.flat = torch.zeros(self.n_grad + n_extra, device=dev, dtype=torch.float32)
        self.views, off = [], 0
        for p, n in zip(self.params, self.sizes):
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n
        self.extra = self.flat[self.n_grad:]
        self.attach()

    def attach(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero(self):
        self.flat.zero_()
        self.attach()

    def reduce(self, extra=None):
        """all_reduce(SUM) of the gradients (and `extra`, a list of 0-d tensors / floats placed in the tail) -> the summed extra values as
        python floats (one read-back), gradients left in place as every p.grad"""
        with torch.no_grad():
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                p.grad = v
            if extra is not None:
                self.extra.zero_()
                self.extra[:len(extra)].copy_(torch.stack([torch.as_tensor(e, dtype=torch.float32, device=self.flat.device).reshape(()) for e in extra]))
            if dist.is_available() and dist.is_initialized():
                buf = _collective_device(self.flat)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                if buf.data_ptr() != self.flat.data_ptr():
                    self.flat.copy_(buf)
        return self.extra[:len(extra)].tolist() if extra is not None else []
