"""Data parallelism for the batched step: one process per GPU, the global batch sharded by
question (every recurrence and the per-question BatchNorm statistics stay inside one sample, so
ranks never exchange activations), ONE exchange per iteration: an all-reduce (sum) of the flat
fp32 gradient buffer over RCCL/xGMI, issued in buckets (decoder | frame encoder | rest) as their
gradients become final, so the first two overlap the rest of backward.  The 1/world_size scale is folded into the fused Adam kernel (``grad_scale``).

The reference has no distributed code at all; this module is new.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def exchange_needed(group=None) -> bool:
    """True when gradients must be all-reduced: more than one rank — or MMQG_FORCE_DP=1 with an
    initialised process group, which sends a single rank through the same RCCL calls (the way to
    rehearse the N>1 launch sequence on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("MMQG_FORCE_DP", "0") == "1"


# A persistent time loop needs every workgroup of its grid resident at once, and an RCCL channel kernel that waits for a
# slow peer keeps its CUs for as long as it waits.  So in data-parallel runs the one persistent launch that runs BESIDE
# collectives (the text encoder's backward loop: the decoder and frame-encoder buckets travel meanwhile) leaves
# RESERVED_CUS compute units out of its grid (mmqg_persist_set_reserved_cus), and RCCL is capped to as many channels
# (one workgroup each) as fit there with two collectives in flight.  Every other persistent launch (forward loops, the
# decoder's backward loop) runs when no bucket is final yet and no collective of the previous step is left (Adam has
# waited for them): they keep the whole chip.
RESERVED_CUS = int(os.environ.get("MMQG_DP_RESERVE_CUS", "32"))


def configure_rccl_env(env=os.environ) -> None:
    """Call BEFORE init_process_group: cap RCCL's channels so that two concurrent all-reduces fit the reserved CUs.
    A value the user has set is left alone."""
    if RESERVED_CUS > 0:
        env.setdefault("NCCL_MAX_NCHANNELS", str(max(2, RESERVED_CUS // 2)))
        env.setdefault("NCCL_MIN_NCHANNELS", str(min(4, max(2, RESERVED_CUS // 2))))


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Rank's contiguous slice of a global batch (dim 0 of every tensor is the question index)."""
    B = next(iter(batch.values())).shape[0]
    if B % world:
        raise ValueError(f"global batch {B} is not divisible by world size {world}")
    per = B // world
    return {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}


class GradReducer:
    """All-reduce of a flat gradient buffer in named buckets.

    ``segments`` maps a name to a [start, end) element range of ``flat_g``; ``order`` lists the
    buckets in the order their gradients become final during backward.  ``reduce(name)`` starts
    the bucket's all-reduce asynchronously (RCCL runs it on its own stream once the producer
    stream reaches this point); ``finish()`` makes the current stream wait for all of them.
    Works on any device / backend, which is how the world_size-2 gloo tests exercise it.
    """

    def __init__(self, flat_g: torch.Tensor, buckets: Sequence[Tuple[str, int, int]], group=None):
        self.flat_g = flat_g
        self.buckets = {name: (a, b) for name, a, b in buckets}
        self._done: set = set()
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = exchange_needed(group)
        self._pending: Dict[str, object] = {}

    def reduce(self, name: str) -> None:
        if not self.active or name not in self.buckets or name in self._done:
            return
        self._done.add(name)
        a, b = self.buckets[name]
        if b > a:
            self._pending[name] = dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self, *names: str) -> None:
        """Make the current stream wait for these buckets' all-reduces (the others stay in flight)."""
        for name in names:
            w = self._pending.pop(name, None)
            if w is not None:
                w.wait()

    def reduce_remaining(self) -> None:
        for name in self.buckets:
            self.reduce(name)

    def discard(self) -> None:
        """Wait for whatever is in flight and forget which buckets were reduced (after a pass whose
        gradients are thrown away)."""
        self.finish()

    def finish(self) -> None:
        for w in self._pending.values():
            w.wait()
        self._pending.clear()
        self._done.clear()

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def trainer_buckets(segments: Dict[str, Tuple[int, int]], n_params: int) -> List[Tuple[str, int, int]]:
    """Buckets from the trainer's flat layout, in the order their gradients become final during
    backward: the decoder's after the decoder backward, the frame encoder's after its (short)
    backward on the side stream, everything else (text encoder, shared embedding) only at the end.
    Layout dec | vid | text | emb gives three buckets; any other order falls back to dec | rest."""
    d0, d1 = segments["dec"]
    v0, v1 = segments.get("vid", (d1, d1))
    if 0 <= v0 - d1 < 4 and v1 > v0 and v1 < n_params:
        return [("dec", d0, v0), ("vid", v0, v1), ("rest", v1, n_params)]
    return [("dec", d0, d1), ("rest", d1, n_params)]


def broadcast_parameters(flat_p: torch.Tensor, group=None, src: int = 0) -> None:
    """Initial replica sync (rank ``src``'s parameters everywhere)."""
    if exchange_needed(group):
        dist.broadcast(flat_p, src=src, group=group)
