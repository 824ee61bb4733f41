"""The reference trainer's loop (train_detector_3D_angle.py:337-417) for one process per GPU.

What the reference does per iteration (:362-408): ``zero_grad`` -> forward of ``[im, label]`` -> ``.mean()`` of each of the
three losses (DataParallel has gathered one value per replica) -> sum -> skip the iteration if the sum is exactly 0 ->
``backward`` -> ``clip_grad_norm_(params, 0.1)`` -> ``optimizer.step()``; any exception inside the iteration is printed and the
iteration skipped (:406-408).  Per epoch (:410-417): ``scheduler.step(mean of the epoch's losses)`` (ReduceLROnPlateau,
patience 4, mode "min", :338) and ``torch.save(retinanet.state_dict(), "..._e{epoch}.pt")``.

With image-sharded data parallelism the same loop needs three things the single-process one gets for free:
  * every rank must take the SAME skip / continue decision, or the ranks that go on wait in the gradient all-reduce for one
    that skipped.  The replicas' mean losses and a "my forward raised" flag travel in ONE 4-float all-reduce right after the
    forward (the reference synchronises there as well: ``bool(loss == 0)``, :380); if any rank failed, or the mean loss is 0,
    every rank skips -- which is what DataParallel does when one replica raises.  The same all-reduce carries a "I have a batch"
    flag: when the ranks' shards are of unequal length the ones that run out keep taking part in it (and in nothing else) until
    every rank has run out, so all ranks leave the epoch together and no collective is ever left unmatched;
  * the scheduler must be stepped with the mean over ranks (ddp.mean_losses' point), so the learning rates cannot diverge;
  * one rank writes the checkpoint; the weights are identical on all of them (same initial weights, same averaged gradients).
The gradient average itself happens inside ``backward`` (ddp.GradReducer attached to the model).

What is NOT caught: an exception in ``loss.backward()`` or ``optimizer.step()`` (the reference's try / except spans them too,
:369-408, but there a failed replica cannot leave the others inside a collective).  Here the other ranks may already be inside the
gradient all-reduce, so the rank re-raises, exits non-zero, and the launcher (torch.distributed.run) tears the others down; never
retry in-process after a GPU error.
"""
import os

import torch
import torch.distributed as dist


def _world(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def agree(losses, failed, device, group=None, has_batch=True):
    """-> (mean losses over the ranks that computed them [3] as a host list, number of ranks whose forward failed, number of ranks
    that had a batch).  ONE all-reduce of 5 floats."""
    vec = torch.zeros(5, dtype=torch.float32, device=device)
    if has_batch and not failed:
        vec[:3] = torch.stack([l.detach().reshape(-1).float().mean() for l in losses]).to(device)
    elif has_batch:
        vec[3] = 1.0
    vec[4] = 1.0 if has_batch else 0.0
    if _world(group) > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    host = vec.tolist()                                        # the iteration's one host read (the reference: bool(loss == 0))
    n_failed, n_have = int(round(host[3])), int(round(host[4]))
    ok_ranks = max(n_have - n_failed, 1)
    return [v / ok_ranks for v in host[:3]], n_failed, n_have


def dataparallel_state_dict(sd):
    """The same state_dict with nn.DataParallel's ``module.`` prefix on every key: what the reference's MULTI-GPU trainer saves and
    resumes from (train_detector_3D_angle.py:415-417 saves ``retinanet.state_dict()`` of the DataParallel wrapper; :301-310 loads into
    it).  ``ResNet.load_state_dict`` of this package accepts both spellings."""
    return {"module." + k: v for k, v in sd.items()}


def train(net, optimizer, scheduler, batches, epochs, *, start_epoch=0, clip_norm=None, checkpoint=None, rank=0, group=None,
          log=print, freeze_bn=True, log_every=2, skip_zero_loss=True, dataparallel_keys=False):
    """net([im, label]) -> (cls_loss, reg_loss, vp_loss); batches(epoch) -> iterable of (im, label) already on the device (this
    rank's shard); optimizer: ``optim.ClipAdam`` (clip fused: leave clip_norm None) or any torch optimizer (clip_norm = 0.1
    reproduces :385); scheduler: ``ReduceLROnPlateau`` or None; checkpoint: a path pattern with ``{}`` for the epoch, written
    by rank 0 after every epoch: ``net.state_dict()`` with bare keys -- the format of the reference's SINGLE-GPU runs -- or, with
    ``dataparallel_keys=True``, with the ``module.`` prefix its multi-GPU trainer writes and resumes from (dataparallel_state_dict).
    The ranks' shards may differ in length (see the module docstring).  Returns the per-epoch history
    [{"epoch", "mean_loss", "iterations", "skipped", "lr"}]."""
    params = [p for p in net.parameters() if p.requires_grad]
    device = params[0].device
    history = []
    for epoch in range(start_epoch, epochs):
        net.train()
        if freeze_bn and hasattr(net, "freeze_bn"):
            net.freeze_bn()                                    # :357, 368
        epoch_loss, skipped = [], 0
        stream, it = iter(batches(epoch)), -1
        while True:
            it += 1
            batch = next(stream, None)
            losses, failed = None, False
            if batch is not None:
                im, label = batch
                optimizer.zero_grad()
                try:
                    losses = net([im, label])
                except Exception as e:                         # :406-408 -- but the other ranks must learn of it
                    log("rank %d, epoch %d, iteration %d: %s" % (rank, epoch, it, e))
                    failed = True
            mean, n_failed, n_have = agree(losses, failed, device, group, has_batch=batch is not None)
            if n_have == 0:
                break                                          # every rank's shard is exhausted: all leave the epoch here
            total = sum(mean)
            # :380-381, decided identically on every rank; a rank without a batch makes everybody skip (it has no gradient to give)
            if n_failed or n_have < _world(group) or (skip_zero_loss and total == 0):
                skipped += 1
                continue
            loss = sum(l.mean() for l in losses)
            loss.backward()                                    # gradient all-reduce inside (ddp.GradReducer)
            if clip_norm is not None:
                torch.nn.utils.clip_grad_norm_(params, clip_norm)          # :385
            optimizer.step()                                   # :387
            epoch_loss.append(total)
            if rank == 0 and log_every and it % log_every == 0:
                log("Epoch: %d | Iteration: %d | Classification loss: %1.5f | Regression loss: %1.4f | VP loss: %1.4f | Running loss: %1.4f"
                    % (epoch, it, mean[0], mean[1], mean[2], sum(epoch_loss) / len(epoch_loss)))
        mean_loss = sum(epoch_loss) / len(epoch_loss) if epoch_loss else float("nan")
        if scheduler is not None and epoch_loss:
            scheduler.step(mean_loss)                          # :412 -- the same number on every rank
        lr = optimizer.param_groups[0]["lr"]
        if checkpoint is not None and rank == 0:               # :415-417
            path = checkpoint.format(epoch)
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            torch.save(dataparallel_state_dict(net.state_dict()) if dataparallel_keys else net.state_dict(), path)
        if _world(group) > 1:
            dist.barrier(group=group)                          # nobody starts the next epoch before the checkpoint is on disk
        history.append({"epoch": epoch, "mean_loss": mean_loss, "iterations": len(epoch_loss), "skipped": skipped, "lr": lr})
        if rank == 0:
            log("Epoch %d training complete: mean loss %.5f over %d iterations (%d skipped), lr %.3g"
                % (epoch, mean_loss, len(epoch_loss), skipped, lr))
    return history
