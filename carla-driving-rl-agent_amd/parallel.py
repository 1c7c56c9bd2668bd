"""Data-parallel PPO learner: one process per GPU, rollout buffer sharded by env shard, the contiguous gradient arena
all-reduced over RCCL once per pass -- as one coalesced early group under the tower's backward plus the tower slice
behind it: 2 + 2 collectives per update-step, the BatchNorm moving statistics riding in the value pass's early group (7 single-slice
calls before round 6).

The gradient arena is laid out [policy | trunk | value] so that the policy pass reduces the
contiguous slice [policy | trunk] and the value pass the slice [trunk | value]; the loss kernels
already scale gradients by 1/world_size, so a SUM all-reduce yields the average and every rank
applies the identical clip + Adam update (parameters and Adam state stay replicated).
BatchNorm statistics are local to a rank's minibatch (N independent reference agents sharing
weights); the moving statistics are averaged with a second, tiny all-reduce.
"""
import os

import torch
import torch.distributed as dist


class _EventWork:
    """Work-like handle around an event recorded on the communication stream: wait() = the current stream waits for it."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class DataParallelLearner:
    def __init__(self, engine, group=None, sync_bn_stats=True, force_collectives=False, overlap=True):
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.sync_bn_stats = sync_bn_stats
        self.force = force_collectives and dist.is_initialized()     # run the collectives even at world 1 (smoke tests)
        p_off, p_n = engine.region('policy', True)
        t_off, t_n = engine.region('trunk', True)
        v_off, v_n = engine.region('value', True)
        assert p_off == 0 and t_off == p_n and v_off == p_n + t_n, 'gradient arena must be [policy|trunk|value]'
        self._policy_slice = (0, p_n + t_n)
        self._value_slice = (t_off, t_off + t_n + v_n)
        # Overlap: the trunk's tail tensors (everything behind the image tower: feature nets, GRUs, concat
        # BN + Dense) sit at the END of the trunk region and are final ~1 ms into the backward, like the head's; they go out
        # as early buckets on a communication stream that the engine releases at that point, and only the tower's slice
        # waits for the end of the backward.  9.6 MB in total, latency-bound on 7 x 153 GB/s xGMI either way.
        # The boundary comes from the ENGINE (cdrl_learner_tail_offset: the same op-list position that releases the
        # communication stream), so the early bucket can never contain a tensor whose gradient is still being written; the
        # parameter table is only used to cross-check it.
        tower_n = engine.tail_offset() if hasattr(engine, 'tail_offset') else t_n
        assert 0 <= tower_n <= t_n, (tower_n, t_n)
        table = getattr(engine, 'tables', None)
        if table is not None and tower_n < t_n:
            ents = [e for e in table['trunk'].entries if e['trainable']]
            tower_end = max((e['offset'] + e['numel'] for e in ents if e['name'].startswith('img.')), default=0)
            tail_begin = min((e['offset'] for e in ents if not e['name'].startswith('img.')), default=t_n)
            assert tower_end <= tower_n <= tail_begin, \
                f'image-tower tensors must precede the tail tensors in the trunk arena ({tower_end} <= {tower_n} <= {tail_begin})'
        self._tower = (t_off, t_off + tower_n)
        self._policy_early = [(0, p_n), (t_off + tower_n, t_off + t_n)]
        self._value_early = [(t_off + tower_n, t_off + t_n + v_n)]
        self._comm = None
        if os.environ.get('CDRL_DP_OVERLAP', '1') == '0':       # one fused all-reduce per pass after the backward (round 1 form)
            overlap = False
        if self.use_comm_stream(overlap, self.world, self.force, tower_n, t_n,
                                bool(getattr(engine, 'device', None)) and hasattr(engine, 'set_comm_stream') and engine.grads.is_cuda):
            self._comm = torch.cuda.Stream(device=engine.grads.device)
            engine.set_comm_stream(self._comm)
        # BatchNorm moving statistics = everything the replicated Adam update does not touch:
        # [policy_state | trunk_state | value_state] (contiguous) and the old policy's copy of the policy-head statistics.
        # The old-policy WEIGHTS are identical on every rank by construction (policy_apply copies the replicated policy) and
        # are NOT reduced: SUM x 1/world would only reproduce them exactly for power-of-two world sizes.
        s0, _ = engine.region('policy', False)
        v0, vn = engine.region('value', False)
        o0, on = engine.region('old_policy', False)
        self._state_slices = [(s0, v0 + vn), (o0, o0 + on)]

    @staticmethod
    def use_comm_stream(overlap, world, force, tower_n, t_n, device_engine) -> bool:
        """Early buckets on a communication stream only when the engine RELEASES that stream in the middle of the backward.
        Under hipGraph replay it does not (the external stream must stay out of the capture) and reports tail_offset() == the
        whole trunk: nothing would order an early bucket -- not even the heads' -- behind the replayed pass, so the single
        post-pass all-reduce on the launch stream is taken instead (a premature bucket would average stale gradients)."""
        return bool(overlap and (world > 1 or force) and device_engine and tower_n < t_n)

    def reduce_policy_gradients(self):
        """All-reduce of the policy pass's gradients [policy | trunk]; call after policy_forward_backward has been enqueued."""
        self._reduce_gradients(self._policy_early, self._policy_slice)

    def reduce_value_gradients(self):
        self._reduce_gradients(self._value_early, self._value_slice)

    def _reduce_gradients(self, early, full, with_stats=False):
        """Gradient all-reduce of one pass, issued after the pass has been ENQUEUED: early buckets on the communication
        stream (released by the engine in the middle of the backward), the tower slice on the current stream.
        with_stats: the BatchNorm moving statistics ride in the same early group (they are final once the pass's FORWARD has run,
        long before the release point): SUM with the gradients, scaled by 1 / world on the communication stream afterwards -- no
        collective of their own.  Returns True when the statistics were taken along."""
        if self.world == 1 and not self.force:
            return False
        g = self.engine.grads
        if self._comm is None:
            dist.all_reduce(g[full[0]:full[1]], op=dist.ReduceOp.SUM, group=self.group)
            return False
        works = []
        stats = with_stats and self.sync_bn_stats and self._nccl()
        with torch.cuda.stream(self._comm):
            tensors = [g[lo:hi] for lo, hi in early if hi > lo]
            if stats:
                tensors += [self.engine.params[lo:hi] for lo, hi in self._state_slices if hi > lo]
            works += self._all_reduce_tensors(tensors, dist.ReduceOp.SUM)      # ONE coalesced launch for the pass's early buckets
            if stats and self.world > 1:
                for w in works:
                    w.wait()                # (stream-level, on the communication stream)
                for lo, hi in self._state_slices:
                    self.engine.params[lo:hi].mul_(1.0 / self.world)
                ev = torch.cuda.Event()
                ev.record(self._comm)
                works = [_EventWork(ev)]
        works.append(dist.all_reduce(g[self._tower[0]:self._tower[1]], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()                    # stream-level: the current stream (where *_apply is enqueued next) waits
        return stats

    def _nccl(self) -> bool:
        return dist.is_initialized() and dist.get_backend(self.group) == 'nccl'

    def _all_reduce_many(self, flat, slices, op):
        return self._all_reduce_tensors([flat[lo:hi] for lo, hi in slices if hi > lo], op)

    def _all_reduce_tensors(self, tensors, op):
        """Asynchronous all-reduce of several tensors (slices of the flat arenas).  On RCCL they go out as ONE coalesced group (one
        launch instead of one per slice: at world 1 every collective of the update-step costs ~45 us of stream time whatever its
        size); other backends (gloo in the CPU tests) take them one by one.  Returns the work handles."""
        if len(tensors) > 1 and self._nccl() and hasattr(dist, '_coalescing_manager'):
            with dist._coalescing_manager(group=self.group, device=tensors[0].device, async_ops=True) as cm:
                for t in tensors:
                    dist.all_reduce(t, op=op, group=self.group)
            return [cm]
        return [dist.all_reduce(t, op=op, group=self.group, async_op=True) for t in tensors]

    def _allreduce(self, flat, lo, hi, scale=None):
        if self.world == 1 and not self.force:
            return
        view = flat[lo:hi]
        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        if scale is not None:
            view.mul_(scale)

    def broadcast_parameters(self, src=0):
        if self.world > 1 or self.force:
            dist.broadcast(self.engine.params, src=src, group=self.group)
            dist.broadcast(self.engine.adam_m, src=src, group=self.group)
            dist.broadcast(self.engine.adam_v, src=src, group=self.group)

    def policy_step(self, batch, resample=None):
        """resample=(seed, offset): the reference-faithful loss on a fresh Beta sample of the new policy drawn on the device
        (each rank passes its own Philox offset); None: stored-action loss (batch['u'])."""
        e = self.engine
        if resample is not None:
            e.policy_forward_backward_resample(batch, seed=resample[0], offset=resample[1], grad_scale=1.0 / self.world)
        else:
            e.policy_forward_backward(batch, grad_scale=1.0 / self.world)
        self._reduce_gradients(self._policy_early, self._policy_slice)
        e.policy_apply()

    def value_step(self, batch, with_stats=False):
        """with_stats: average the BatchNorm moving statistics in this pass's early group (update_step: the value pass is the last
        forward of an update-step).  Returns True when that happened (no separate sync_moving_statistics() needed)."""
        e = self.engine
        e.value_forward_backward(batch, grad_scale=1.0 / self.world)
        done = self._reduce_gradients(self._value_early, self._value_slice, with_stats=with_stats)
        e.value_apply()
        return done

    def sync_moving_statistics(self):
        """Average of the BatchNorm moving statistics over the ranks: both state slices in ONE collective.  RCCL: a coalesced AVG
        all-reduce (no scaling kernels behind it); other backends: SUM + scale per slice."""
        if not (self.sync_bn_stats and (self.world > 1 or self.force)):
            return
        if self._nccl():
            for w in self._all_reduce_many(self.engine.params, self._state_slices, dist.ReduceOp.AVG):
                w.wait()
            return
        for lo, hi in self._state_slices:
            self._allreduce(self.engine.params, lo, hi, scale=1.0 / self.world)

    def update_step(self, policy_batch, value_batch, resample=None):
        """One PPO update-step = one policy minibatch step + one value minibatch step
        (reference rl/agents/ppo.py:199-224)."""
        if self.world == 1 and not self.force and os.environ.get('CDRL_SEQUENCE', '1') != '0':
            # no collective between the four calls: one hand-over between the caller's stream and the engine's around all of them
            with self.engine.sequence():
                self.policy_step(policy_batch, resample)
                self.value_step(value_batch, with_stats=True)
            return
        self.policy_step(policy_batch, resample)
        if not self.value_step(value_batch, with_stats=True):
            self.sync_moving_statistics()
