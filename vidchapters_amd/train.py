"""Training step of the reference's dvc.py loop (dvc.py:71-133), MI355X-native.

``Trainer.step(batch)`` reproduces the recipe exactly -- generative pass, optional denoising pass on the cached
ViT output, ``loss = generative*l1 + denoising*l2``, clip_grad_norm_, Adam, time-token renorm (twice on the tied
tensor), LR schedule (util/misc.py:15-42) -- but drives the engine directly (no autograd graph) and runs the
optimizer as ONE fused kernel over the flat parameter arena.

Data parallelism (new capability: the reference never wraps the model in DDP): one process per
GPU, every rank holds a replica, the flat fp32 gradient arena is all-reduced (SUM, scaled by 1/world inside the
Adam kernel) in large contiguous slices on a side stream as soon as the backward pass has finished the
corresponding parameters (decoder -> encoder+embedding -> ViT), overlapping RCCL over xGMI with the rest of
backward.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import lib as L


def lr_at(step: int, total: int, base_lr: float, schedule: str = "", warmup_frac: float = 0.1) -> float:
    """util/misc.py:15-42 (adjust_learning_rate)."""
    warm = round(warmup_frac * total)
    if schedule == "":
        return base_lr
    if schedule not in ("linear_with_warmup", "cosine_with_warmup"):
        raise NotImplementedError(schedule)
    if step < warm:
        return base_lr * (float(step) / float(max(1, warm)))        # gamma first, like the reference: bit-identical LR
    if schedule == "linear_with_warmup":
        return base_lr * max(0.0, float(total - step) / float(max(1, total - warm)))
    if schedule == "cosine_with_warmup":
        return base_lr * (1 + math.cos(math.pi * float(step - warm) / float(max(1, total - warm)))) / 2
    raise NotImplementedError(schedule)


def bucket_plan(start: int, end: int, chunk: int, world: int, shard: bool) -> List[Tuple[str, int, int]]:
    """Collectives that reduce arena[start:end]: ("rs", offset, n) = reduce-scatter of n elements (n a multiple of 64 * world, rank r
    receives [offset + r*n/world, offset + (r+1)*n/world)), ("ar", offset, n) = all-reduce.  Pure host logic (tested on the CPU)."""
    plan: List[Tuple[str, int, int]] = []
    o = start
    while o < end:
        e = min(end, o + chunk)
        n_sh = ((e - o) // (64 * world)) * (64 * world) if shard else 0
        if n_sh:
            plan.append(("rs", o, n_sh))
        if o + n_sh < e:
            plan.append(("ar", o + n_sh, e - o - n_sh))
        o = e
    return plan


class GradSync:
    """Bucketed asynchronous reduction of the gradient arena on a dedicated stream (RCCL over xGMI: backend "nccl" on ROCm).

    ``ready(start, end)`` hands a finished, contiguous slice of the fp32 gradient arena to the reduction: the side stream waits for
    the producing streams through events, then issues one collective per ``bucket_bytes`` of WIRE data (default 48 MiB: xGMI is
    point to point, a ring collective is bound per link, and a few tens of MB per collective keep the links streaming while the
    first buckets of a slice overlap the rest of backward).  ``comm_dtype="bf16"`` halves the wire bytes: the slice is rounded into a
    bf16 staging arena, reduced there and widened back into the fp32 arena (the optimizer keeps reading fp32).  ``force=True`` issues
    the collectives even for a one-rank group (the GPU test that loads RCCL on a single-GPU box).

    ``shard=True`` (sharded optimizer, the SURVEY 5 design: direct reduce-scatter + all-gather instead of an all-reduce): every
    bucket is REDUCE-SCATTERED -- rank r ends up with the summed r-th 1/world of each bucket (``owned`` ranges) -- the optimizer
    then updates only those stripes (Adam reads and writes 8.7 GB / world per rank instead of 8.7 GB), and ``gather_shadow`` all-gathers
    the updated bf16 SHADOW weights (0.58 GB instead of a second 1.16 GB half of the all-reduce).  Ranges handed over with
    ``replicate=True`` (the small fp32-consumed parameters and the time-token rows, which every rank needs as fp32 masters) are
    all-reduced and updated everywhere.  The fp32 masters of the stripes a rank does not own go stale: ``gather_master`` brings
    them up to date for checkpoints / evaluation through the nn.Module."""

    def __init__(self, arena, group=None, bucket_bytes: int = 48 << 20, comm_dtype: str = "fp32", force: bool = False, shard: bool = False):
        self.arena = arena
        self.group = group
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if live else 1
        self.rank = dist.get_rank(group) if live else 0
        self.active = self.world > 1 or (force and live)
        if comm_dtype not in ("fp32", "bf16"):
            raise ValueError(f"grad_comm_dtype must be 'fp32' or 'bf16' (got {comm_dtype!r})")
        self.comm_dtype = comm_dtype
        self.shard = bool(shard) and self.active
        esize = 4 if comm_dtype == "fp32" else 2
        unit = 64 * (self.world if self.shard else 1)                      # stripes stay 64-element (256-byte) aligned
        self.chunk = max(unit, max(1, bucket_bytes // esize) // unit * unit)     # elements per collective
        dev = arena.grad.device
        self.stream = torch.cuda.Stream() if self.active else None
        self.stage = torch.empty(arena.numel, dtype=torch.bfloat16, device=dev) if self.active and comm_dtype == "bf16" else None
        # gloo (the CPU-backend tests with two ranks on one GPU) has no reduce-scatter / all-gather for device tensors: staged through the host
        self._host_staged = self.shard and live and dist.get_backend(group) == "gloo"
        self._rs_out = torch.empty(self.chunk // self.world, dtype=torch.float32 if comm_dtype == "fp32" else torch.bfloat16, device=dev) if self.shard else None
        self._ag_in = torch.empty(self.chunk // self.world, dtype=torch.bfloat16, device=dev) if self.shard else None
        self.works: List = []
        self.owned: List[Tuple[int, int]] = []        # sharded mode, per step: arena ranges whose summed gradient this rank holds
        self.replicated: List[Tuple[int, int]] = []   # ... ranges every rank holds (all-reduced)
        self.buckets: List[Tuple[int, int]] = []      # ... (start, elements) of every reduce-scattered bucket (all-gathered after the update)
        self.collectives = 0          # statistics of the current step (reset by Trainer.step)
        self.bytes_reduced = 0
        self.exposed_ms_events = None # (start, end) events around the final wait of the last step
        # sharded mode: after an optimizer step the fp32 masters / Adam moments of the stripes OTHER ranks own are out of date on this
        # rank until gather_master() / gather_stripes() (collectives) bring them in; consumers check the flags instead of reading stale data
        self.masters_stale = False
        self.moments_stale = False

    def begin_step(self) -> None:
        self.collectives = self.bytes_reduced = 0
        self.owned, self.replicated, self.buckets = [], [], []

    def range_of(self, first: str, last: str) -> Tuple[int, int]:
        a = self.arena
        n = 1
        for s in a.shapes[last]:
            n *= s
        return a.offsets[first], (a.offsets[last] + n + 63) // 64 * 64

    # ---- collectives (device tensors; host-staged for gloo) ----------------------------------------------------------------------
    def _reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        if self._host_staged:
            hi = inp.cpu()
            ho = torch.empty(out.numel(), dtype=hi.dtype)
            if hi.dtype == torch.bfloat16:          # gloo reduces fp32
                hi = hi.float(); ho = ho.float()
            dist.reduce_scatter_tensor(ho, hi, op=dist.ReduceOp.SUM, group=self.group)
            out.copy_(ho.to(out.dtype))
        else:
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()   # stream-side dependency only

    def _all_gather(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        if self._host_staged:
            hi = inp.cpu().view(torch.uint8)                 # a gather moves bits: bytes are a type gloo knows
            ho = torch.empty(2 * out.numel(), dtype=torch.uint8)
            dist.all_gather_into_tensor(ho, hi, group=self.group)
            out.copy_(ho.view(torch.bfloat16))
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True).wait()

    def ready(self, start: int, end: int, also=(), replicate: bool = False) -> None:
        """Gradients in arena[start:end] are final on the current stream (and on the streams in ``also``, e.g. the
        weight-gradient stream): reduce them in the background.  ``replicate``: keep the range whole on every rank even in sharded
        mode (all-reduce)."""
        if not self.active or end <= start:
            return
        for st in (torch.cuda.current_stream(),) + tuple(also):
            ev = torch.cuda.Event()
            ev.record(st)
            self.stream.wait_event(ev)
        g = self.arena.grad
        W, r = self.world, self.rank
        with torch.cuda.stream(self.stream):
            for kind, o, n in bucket_plan(start, end, self.chunk, W, self.shard and not replicate):
                if kind == "rs":                           # reduce-scatter: this rank receives the sum of its 1/W stripe of the bucket
                    part = n // W
                    mine = (o + r * part, o + (r + 1) * part)
                    if self.stage is None:
                        self._reduce_scatter(self._rs_out[:part], g[o:o + n])
                        self.bytes_reduced += n * 4
                    else:
                        L.cast_bf16(g[o:o + n], self.stage[o:o + n], n)
                        self._reduce_scatter(self._rs_out[:part], self.stage[o:o + n])
                        self.bytes_reduced += n * 2
                    g[mine[0]:mine[1]].copy_(self._rs_out[:part])
                    self.owned.append(mine)
                    self.buckets.append((o, n))
                else:                                      # all-reduce (unsharded mode, replicated ranges, the ragged end of a range)
                    e = o + n
                    if self.stage is None:
                        self.works.append(dist.all_reduce(g[o:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                        self.bytes_reduced += n * 4
                    else:
                        L.cast_bf16(g[o:e], self.stage[o:e], n)                          # fp32 -> bf16 (RNE) on the side stream
                        w = dist.all_reduce(self.stage[o:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                        w.wait()                                                          # stream-side dependency only (no host block with NCCL/RCCL)
                        g[o:e].copy_(self.stage[o:e])                                     # widen back for the fp32 optimizer
                        self.bytes_reduced += n * 2
                    if self.shard:
                        self.replicated.append((o, e))
                self.collectives += 1

    def finish(self) -> None:
        if not self.active:
            return
        for w in self.works:
            w.wait()
        self.works.clear()
        main = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        main.wait_stream(self.stream)
        e1.record(main)
        self.exposed_ms_events = (e0, e1)

    def gather_shadow(self) -> None:
        """Sharded mode, after the optimizer updated the owned stripes: all-gather the bf16 shadow weights of every reduce-scattered
        bucket (on the current stream: the next forward reads them)."""
        sh = self.arena.shadow
        W, r = self.world, self.rank
        for o, n in self.buckets:
            part = n // W
            self._ag_in[:part].copy_(sh[o + r * part:o + (r + 1) * part])
            self._all_gather(sh[o:o + n], self._ag_in[:part])
            self.collectives += 1
            self.bytes_reduced += n * 2

    def gather_shadow_async(self, groups) -> Dict[str, torch.cuda.Event]:
        """gather_shadow on the communication stream, in the order the next forward needs the weights.  ``groups`` = [(name, start, end)]
        arena ranges in need order (encoder + embedding, ViT, decoder); returns {name: event recorded behind the group's last bucket}.
        The stream waits for the current stream (the Adam launches) first; nothing on the main stream waits here."""
        sh = self.arena.shadow
        W, r = self.world, self.rank
        ev0 = torch.cuda.Event()
        ev0.record(torch.cuda.current_stream())
        self.stream.wait_event(ev0)
        events: Dict[str, torch.cuda.Event] = {}
        with torch.cuda.stream(self.stream):
            for name, g0, g1 in groups:
                for o, n in self.buckets:
                    if not (g0 <= o < g1):
                        continue
                    part = n // W
                    self._ag_in[:part].copy_(sh[o + r * part:o + (r + 1) * part])
                    self._all_gather(sh[o:o + n], self._ag_in[:part])
                    self.collectives += 1
                    self.bytes_reduced += n * 2
                ev = torch.cuda.Event()
                ev.record(self.stream)
                events[name] = ev
        return events

    def gather_stripes(self, buf: torch.Tensor) -> None:
        """Sharded mode: all-gather an fp32 arena-shaped buffer whose owned stripes are current on every rank (master weights, Adam
        moments) so that it is whole everywhere.  Collective: every rank calls it.  Uses the bucket layout of the LAST step."""
        if not self.shard:
            return
        W, r = self.world, self.rank
        for o, n in self.buckets:
            part = n // W
            tmp = buf[o + r * part:o + (r + 1) * part].clone()
            if self._host_staged:
                ho = torch.empty(n, dtype=torch.float32)
                dist.all_gather_into_tensor(ho, tmp.cpu(), group=self.group)
                buf[o:o + n].copy_(ho)
            else:
                dist.all_gather_into_tensor(buf[o:o + n], tmp, group=self.group)

    def gather_master(self) -> None:
        """Sharded mode: bring the fp32 master weights of the stripes other ranks own up to date (checkpoints, evaluation through the
        nn.Module, switching to an unsharded optimizer)."""
        self.gather_stripes(self.arena.master)
        self.masters_stale = False

    def exposed_ms(self) -> float:
        """Time the main stream spent waiting for the last step's reductions after backward had finished (synchronises)."""
        if self.exposed_ms_events is None:
            return 0.0
        e0, e1 = self.exposed_ms_events
        e1.synchronize()
        return float(e0.elapsed_time(e1))


class Trainer:
    def __init__(self, model, lr: float = 3e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 clip_max_norm: float = 1.0, generative: float = 1.0, denoising: float = 1.0, schedule: str = "",
                 fraction_warmup_steps: float = 0.1, num_training_steps: int = 1, group=None, bucket_bytes: int = 48 << 20,
                 grad_comm_dtype: str = "fp32", force_collectives: bool = False, shard_optimizer: Optional[bool] = None):
        """``shard_optimizer``: False / None (default) = replicated optimizer (all-reduce + full Adam on every rank: a reference-style
        `if is_main_process(): save(model.state_dict())` loop, dvc.py:310-330, works unchanged).  True = sharded (reduce-scatter + Adam on
        the owned 1/N stripes + overlapped bf16 all-gather: less wire traffic and 1/N of the optimizer's HBM traffic per rank) -- OPT-IN
        until its RCCL path has run on a real multi-GPU node: it then needs :meth:`prepare_checkpoint` (a collective, every rank) before
        ``model.state_dict()`` / ``Trainer.state_dict()``.  :meth:`close` detaches the guards it installs on the model."""
        self.model = model
        self.eng = model.engine()
        a = self.eng.arena
        dev = self.eng.device
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.clip, self.gen, self.den = clip_max_norm, generative, denoising
        self.schedule, self.warm, self.total = schedule, fraction_warmup_steps, num_training_steps
        self.m = torch.zeros(a.numel, dtype=torch.float32, device=dev)
        self.v = torch.zeros(a.numel, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.high_priority = False       # experiment (see step): faster step by step, slower in a pipelined loop -> off
        self.overlap_gather = True       # sharded optimizer: all-gather the bf16 shadow on the communication stream under the next forward
        self._hi_stream = None
        shard_optimizer = bool(shard_optimizer)
        self.sync = GradSync(a, group, bucket_bytes=bucket_bytes, comm_dtype=grad_comm_dtype, force=force_collectives, shard=shard_optimizer)
        self.world = self.sync.world
        self._sq_ws = torch.empty(1024, dtype=torch.float32, device=dev)
        self._gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._renorm_ws = torch.empty(self.eng.V + 2, dtype=torch.float32, device=dev)
        self._text_sumsq = torch.zeros(self.eng.V, dtype=torch.float32, device=dev) if self.sync.shard else None
        if self.sync.shard:
            # stale-master guards (sharded optimizer): a re-cast of the bf16 shadow from the masters (an in-place torch update of a
            # Parameter, mark_dirty, load_state_dict) or a state_dict() of the module would silently use the (world-1)/world stale
            # matrices -- fail loudly instead; gather_master() (a collective: every rank) makes them current
            import weakref
            me = weakref.ref(self)             # the guards must not keep a discarded Trainer alive (or raise for it forever): weak reference + close()

            def guard(what, me=me):
                t = me()
                if t is not None:
                    t._stale_guard(what)
            a.stale_guard = self._guard = guard
            self._sd_hook = model.register_state_dict_pre_hook(lambda module, prefix, keep_vars: guard("model.state_dict()"))
        # arena ranges (engine._arena_order): decoder matrices | encoder matrices | ViT matrices + pos_embed | the small fp32-consumed
        # parameters | the tied embedding, whose last rows (time tokens + the zero tail) every rank keeps whole (renorm reads them)
        names = a.names
        small = [n for n in names if type(self.eng).is_small_param(n, a.params[n])]
        first_enc = next(i for i, n in enumerate(names) if n.startswith("t5_model.encoder."))
        first_vis = next(i for i, n in enumerate(names) if not n.startswith("t5_model."))
        self._r_dec = self.sync.range_of(names[0], names[first_enc - 1])
        self._r_enc = self.sync.range_of(names[first_enc], names[first_vis - 1])
        self._enc_first = names[first_enc]
        self._r_vis = self.sync.range_of(names[first_vis], "visual_encoder.pos_embed")
        self._r_small = self.sync.range_of(small[0], small[-1])
        sh0, sh1 = a.offsets["t5_model.shared.weight"], a.numel
        tt0 = sh0 + ((self.eng.V - model.num_bins) * self.eng.d) // 64 * 64 if model.num_bins else sh1
        self._r_shared, self._r_timetok = (sh0, tt0), (tt0, sh1)
        assert self._r_dec[1] == self._r_enc[0] and self._r_enc[1] == self._r_vis[0] and self._r_vis[1] == self._r_small[0] and self._r_small[1] == sh0

    def close(self) -> None:
        """Detach this Trainer from the model: removes the stale-master guards (sharded optimizer) so that a replaced / discarded Trainer
        neither stays alive nor makes ``model.state_dict()`` raise.  Call :meth:`gather_master` first if its masters are still needed."""
        h = getattr(self, "_sd_hook", None)
        if h is not None:
            h.remove()
            self._sd_hook = None
        # only OUR guard: a Trainer that replaced this one on the same model has installed its own since (this object's finaliser may run later)
        mine = getattr(self, "_guard", None)
        if mine is not None and getattr(self.eng.arena, "stale_guard", None) is mine:
            self.eng.arena.stale_guard = None
        self._guard = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stale_guard(self, what: str) -> None:
        if self.sync.masters_stale:
            raise RuntimeError(f"{what}: the fp32 master weights of the stripes other ranks own are stale (sharded optimizer). Call "
                               "Trainer.gather_master() on EVERY rank first (it is a collective), then save / evaluate / update.")

    # ------------------------------------------------------------------------------------------------
    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """One optimizer step on ``batch`` = {video, input_ids, output_ids[, den_input_ids, den_output_ids]}; optional
        ``input_lens`` / ``den_input_lens`` / ``output_lens`` / ``den_output_lens`` (host lists of valid lengths, e.g. from the data loader) let the padding-free
        encoder plan its rows without reading the mask back.  Returns device scalars."""
        if not self.high_priority or not self.eng.overlap:
            return self._step_impl(batch)
        # The chain forward -> dgrad / attention backward -> optimizer is the critical path; the weight-gradient, ViT and K|V streams
        # only fill what it leaves idle.  Issued from a HIGH-priority HIP stream its kernels win the dispatch arbitration against the
        # side streams' (default priority): 56.3 -> 55.6 ms when every step is synchronised (tools/prio_probe.py, interleaved), but
        # 54.1 -> 64.3 ms per step in bench.py's pipelined loop (the host enqueues steps back to back): off by default.
        caller = torch.cuda.current_stream()
        if self._hi_stream is None:
            self._hi_stream = torch.cuda.Stream(device=self.eng.device, priority=-1)
        self._hi_stream.wait_stream(caller)
        with torch.cuda.stream(self._hi_stream):
            losses = self._step_impl(batch)
        caller.wait_stream(self._hi_stream)
        for v in losses.values():
            v.record_stream(caller)
        for v in batch.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(self._hi_stream)
        return losses

    def _step_impl(self, batch: Dict[str, torch.Tensor], hyper_dev: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        m, eng = self.model, self.eng
        assert m.training, "call model.train() first"
        main = torch.cuda.current_stream()
        overlap = eng.overlap
        eng.prepare(wait_shadow=False)      # the forward waits for a pending shadow all-gather group by group (Engine.wait_shadow)
        eng.begin_grad_step()
        try:
            return self._step_body(batch, hyper_dev)
        finally:
            eng.abort_grad_step()        # no-op after a completed step; after an exception: later backward calls accumulate again

    def _step_body(self, batch: Dict[str, torch.Tensor], hyper_dev: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        m, eng = self.model, self.eng
        main = torch.cuda.current_stream()
        overlap = eng.overlap
        self.sync.begin_step()
        losses: Dict[str, torch.Tensor] = {}
        vtape: Dict = {}
        vis, vis_ready = None, None
        if m.use_video:
            video = batch["video"]
            if overlap:          # temporal ViT on its own stream, beside the T5 encoder of the first pass
                eng.vstream.wait_stream(main)
                with torch.cuda.stream(eng.vstream):
                    vis = eng.vit_forward(video, vtape).view(video.shape[0], video.shape[1], eng.d)
                    vis_ready = torch.cuda.Event()
                    vis_ready.record(eng.vstream)
                video.record_stream(eng.vstream)
            else:
                vis = eng.vit_forward(video, vtape).view(video.shape[0], video.shape[1], eng.d)
        tapes = []
        if self.gen:
            t1: Dict = {}
            ids = batch["input_ids"]
            losses["loss"] = eng.t5_loss_forward(vis, ids, ids != 0, batch["output_ids"], batch["output_ids"] != 0, t1,
                                                 vis_ready=vis_ready, input_lens=batch.get("input_lens"), head_grad_scale=float(self.gen),
                                                 output_lens=batch.get("output_lens"))
            tapes.append((t1, self.gen))
        if self.den:
            t2: Dict = {}
            ids = batch["den_input_ids"]
            losses["denoising_loss"] = eng.t5_loss_forward(vis, ids, ids != 0, batch["den_output_ids"],
                                                           batch["den_output_ids"] != 0, t2, vis_ready=vis_ready,
                                                           input_lens=batch.get("den_input_lens"), head_grad_scale=float(self.den),
                                                           output_lens=batch.get("den_output_lens"))
            tapes.append((t2, self.den))

        # backward: later passes first; parameter gradients accumulate in the arena.  The ViT backward starts as soon as
        # the decoder stack of the LAST pass has produced d(loss)/d(vis) and runs beside that pass's encoder backward.
        state = {"dvis": None}

        def add_dvis(dv):
            if dv is None:
                return
            if state["dvis"] is None:
                state["dvis"] = dv
            else:
                L.add(state["dvis"], dv, state["dvis"], dv.numel())

        def launch_vit_backward():
            if not m.use_video:
                return
            dvis = state["dvis"]
            if overlap:
                eng.vstream.wait_stream(main)
                with torch.cuda.stream(eng.vstream):
                    eng.vit_backward(vtape, dvis)
                dvis.record_stream(eng.vstream)
            else:
                eng.vit_backward(vtape, dvis)

        n = len(tapes)
        for k, (tape, coef) in enumerate(reversed(tapes)):
            last = (k == n - 1)
            g = torch.full((1,), float(coef), dtype=torch.float32, device=eng.device)

            def after_decoder(dv, last=last):
                add_dvis(dv)
                if last:
                    if self.sync.active:
                        eng.join_wgrads()
                        self.sync.ready(*self._r_dec)
                    launch_vit_backward()
                    if self.sync.active and m.use_video:     # the (short) ViT backward is fully enqueued: reduce its slice beside the
                        self.sync.ready(*self._r_vis, also=(eng.vstream, eng.wstream) if overlap else ())   # encoder backward
                        state["vis_sent"] = True

            enc_sent = [self._r_enc[0]]          # encoder gradients up to this arena offset are already being reduced

            def encoder_layer_done(i, last=last):
                # every 4 encoder layers (backward runs 11 -> 0): hand their slice to the reduction while the next layers compute,
                # instead of one 0.45 GB all-reduce after the whole stack
                if last and self.sync.active and i > 0 and i % 4 == 0:
                    end = self.sync.range_of(self._enc_first, eng._sa("encoder", i) + "o.weight")[1]      # last matrix of block i in arena order
                    self.sync.ready(enc_sent[0], end, also=(eng.wstream,) if eng.overlap else ())
                    enc_sent[0] = end

            def after_encoder(last=last):
                if last and self.sync.active:
                    eng.join_wgrads()
                    self.sync.ready(enc_sent[0], self._r_enc[1])
                    self.sync.ready(*self._r_shared)
                    self.sync.ready(*self._r_timetok, replicate=True)

            eng.t5_loss_backward(tape, g, after_decoder=after_decoder, after_encoder=after_encoder,
                                 encoder_layer_done=encoder_layer_done)
        if n == 0:
            launch_vit_backward()
        if m.use_video and overlap:
            main.wait_stream(eng.vstream)
        eng.join_wgrads()
        if m.use_video and not state.get("vis_sent"):
            self.sync.ready(*self._r_vis)
        self.sync.ready(*self._r_small, replicate=True)       # norm weights, biases, bias tables: final only now (ViT + both stacks)
        self.sync.finish()
        eng.end_grad_step()
        eng.shadow_events = None            # every group was waited for during this step's forward
        self._optimizer_step(hyper_dev)
        return losses

    def _lr_of_step(self, k: int) -> float:
        """dvc.py:128-133 adjusts the LR *after* optimizer.step(): step 0 runs at args.lr, step k at schedule(k-1)"""
        return self.lr if k == 0 else lr_at(k - 1, self.total, self.lr, self.schedule, self.warm)

    def _optimizer_step(self, hyper_dev: Optional[torch.Tensor] = None) -> None:
        """``hyper_dev`` (graph capture): the kernel reads lr / bias corrections from that device pair instead of the by-value arguments,
        and the step counter is advanced by the caller."""
        eng, a = self.eng, self.eng.arena
        if hyper_dev is None:
            self.step_count += 1
        lr = self._lr_of_step(self.step_count - 1) if hyper_dev is None else self.lr
        self._gnorm_sq.zero_()
        step_no = max(1, self.step_count)
        if self.sync.shard:
            # sharded optimizer: |g|^2 = all-reduce(sum over the stripes this rank owns) + the replicated ranges (counted once), then
            # Adam on the owned stripes and on the replicated ranges only, then the bf16 shadow of every bucket is all-gathered
            rng = self.sync.owned + self.sync.replicated
            if self.clip > 0:
                for s0, e0 in self.sync.owned:
                    L.sqnorm(a.grad[s0:e0], e0 - s0, self._sq_ws, self._gnorm_sq)
                dist.all_reduce(self._gnorm_sq, op=dist.ReduceOp.SUM, group=self.sync.group)
                for s0, e0 in self.sync.replicated:
                    L.sqnorm(a.grad[s0:e0], e0 - s0, self._sq_ws, self._gnorm_sq)
            for s0, e0 in rng:
                L.adam_step(a.master[s0:e0], self.m[s0:e0], self.v[s0:e0], a.grad[s0:e0], a.shadow[s0:e0], e0 - s0, lr, self.betas[0],
                            self.betas[1], self.eps, self.wd, step_no, gnorm_sq=self._gnorm_sq if self.clip > 0 else None,
                            max_norm=self.clip, grad_scale=1.0 / self.world, hyper_dev=hyper_dev)
        else:
            if self.clip > 0:
                L.sqnorm(a.grad, a.numel, self._sq_ws, self._gnorm_sq)
            L.adam_step(a.master, self.m, self.v, a.grad, a.shadow, a.numel, lr, self.betas[0], self.betas[1], self.eps, self.wd,
                        step_no, gnorm_sq=self._gnorm_sq if self.clip > 0 else None, max_norm=self.clip,
                        grad_scale=1.0 / self.world, hyper_dev=hyper_dev)
        if self.sync.shard:
            self.sync.masters_stale = self.sync.moments_stale = self.sync.world > 1
        if self.model.num_bins:
            emb = a.f("t5_model.shared.weight")
            embb = a.w("t5_model.shared.weight")
            if self.sync.shard:
                # the frozen (text) rows are reduce-scattered: this rank's fp32 masters are current only inside its own stripes, so
                # their norms come from per-rank partial row sums of squares (owned stripes: all-reduced; replicated ranges: added once,
                # locally).  The time-token rows are whole on every rank (replicated range).  One all-reduce of V floats per step.
                V, d, nb = eng.V, eng.d, self.model.num_bins
                sh0 = a.offsets["t5_model.shared.weight"]
                t1 = sh0 + (V - nb) * d                                    # end of the frozen rows (arena offset)
                sq = self._text_sumsq
                sq.zero_()
                for s0, e0 in self.sync.owned:
                    lo, hi = max(s0, sh0), min(e0, t1)
                    if lo < hi:
                        L.rowsumsq_range(emb, V, d, lo - sh0, hi - sh0, sq)
                if self.sync.world > 1:
                    if self.sync._host_staged:
                        h = sq.cpu(); dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.sync.group); sq.copy_(h)
                    else:
                        dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=self.sync.group)
                for s0, e0 in self.sync.replicated:
                    lo, hi = max(s0, sh0), min(e0, t1)
                    if lo < hi:
                        L.rowsumsq_range(emb, V, d, lo - sh0, hi - sh0, sq)
                for _ in range(2):  # dvc.py:120-126 twice on the tied tensor: the frozen rows do not change in between
                    L.timetoken_renorm_sq(emb, embb, V, d, nb, sq, self._renorm_ws)
            else:
                for _ in range(2):      # dvc.py:120-126 renormalises `shared` and then `lm_head` -- the same tied tensor
                    L.timetoken_renorm(emb, embb, eng.V, eng.d, self.model.num_bins, self._renorm_ws)
        if self.sync.shard:
            # all-gather of the updated bf16 shadow stripes: LAST collective of the step (the process group runs collectives in issue
            # order: the renorm's small all-reduce above must not queue behind 0.5 GB of gathers), on the communication stream, in
            # the order the next forward needs the weights -- encoder + embedding, ViT, decoder -- with an event per group: the next
            # step's encoder forward (~8 ms) runs while the decoder's 0.2 GB are still on the wire (Engine.wait_shadow)
            if self.overlap_gather:
                groups = [("enc", self._r_enc[0], self._r_enc[1]), ("enc", self._r_shared[0], self._r_shared[1]),
                          ("vit", self._r_vis[0], self._r_vis[1]), ("dec", self._r_dec[0], self._r_dec[1])]
                # one event per name: "enc" is recorded twice (after the encoder range and after the embedding): keep the later one
                eng.shadow_events = self.sync.gather_shadow_async(groups)
            else:
                self.sync.gather_shadow()

    # ------------------------------------------------------------------------------------------------ captured step
    def step_graph(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """:meth:`step` replayed from a hipGraph: the ~2500 launches of a step (all three streams) are captured once and the host then
        enqueues a step with ONE graph launch plus the input copies (host time per step: ~20 ms of Python / ctypes -> well under 1 ms).
        Everything that changes from step to step lives in device memory the captured kernels read:
          * the batch: static buffers, refilled by copy before each replay (a new shape runs one eager step and re-captures on the
            call after it; callers that pass PINNED host tensors must not rewrite them before the step has consumed them);
          * dropout: the by-value seeds of the captured launches are XOR-ed with a device salt word that is rewritten every step
            (v2s_set_seed_salt), so every replay draws new masks;
          * Adam: lr and the two bias corrections come from a device pair (v2s_adam_args.hyper_dev), computed on the host per step
            (LR schedule, dvc.py:128-133).
        The first call runs an ordinary eager step (it creates every lazily allocated workspace), the second call captures.  The
        padding-free encoder changes shapes with the batch and is switched off; data-parallel runs keep the eager path (collectives
        are not captured).  Returns the same device scalars on every call (their values are those of the last replay)."""
        eng = self.eng
        if self.sync.active:
            return self.step(batch)
        tensors = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        key = tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in tensors.items()))
        st = getattr(self, "_g", None)
        if st is None or st["key"] != key:
            # first call, or new shapes: an ordinary eager step first (it creates the LUTs, workspaces and kernel attributes of these
            # shapes -- host-to-device copies are illegal inside a capture); the next call captures
            self._g = {"key": key, "graph": None}
            pack = eng.pack
            eng.pack = False
            try:
                return self.step(tensors)
            finally:
                eng.pack = pack                        # eager steps of the caller keep their padding-free setting
        if st["graph"] is None:
            dev = eng.device
            st["batch"] = {k: torch.empty_like(v) for k, v in tensors.items()}
            for k, v in tensors.items():
                st["batch"][k].copy_(v)
            st["salt"] = torch.zeros(1, dtype=torch.int32, device=dev)
            st["hyper"] = torch.zeros(2, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            L.set_seed_salt(st["salt"])
            pack = eng.pack
            eng.pack = False
            try:
                with torch.cuda.graph(g):
                    st["losses"] = self._step_impl(st["batch"], hyper_dev=st["hyper"])
            finally:
                L.set_seed_salt(None)
                eng.pack = pack
            st["graph"] = g
        for k, v in tensors.items():
            st["batch"][k].copy_(v, non_blocking=True)
        self.step_count += 1
        k = self.step_count
        lr = self._lr_of_step(k - 1)
        # by-value fills (the scalars travel as kernel arguments): a pinned staging buffer reused every step could be rewritten by the
        # host for step k+1 before the queued copy of step k has run -- the host is many steps ahead of the device here
        st["hyper"][0:1].fill_(lr / (1.0 - self.betas[0] ** k))
        st["hyper"][1:2].fill_(1.0 / math.sqrt(1.0 - self.betas[1] ** k))
        seed0, _ = eng.rng_state()
        salt = ((seed0 ^ (k * 0x9E3779B1)) * 0x85EBCA6B) & 0x7FFFFFFF
        st["salt"].fill_(salt)
        st["graph"].replay()
        return st["losses"]

    def gather_master(self) -> None:
        """Sharded optimizer only: make every rank's fp32 master weights (= the nn.Module's parameters) current.  Call before
        ``model.state_dict()`` / evaluation through the module; a no-op otherwise."""
        ev = self.eng.shadow_events            # an asynchronous shadow all-gather may still be writing the bf16 copies mark_dirty re-casts
        if ev:
            for e in ev.values():
                torch.cuda.current_stream().wait_event(e)
            self.eng.shadow_events = None
        self.sync.gather_master()
        self.eng.mark_dirty()

    def prepare_checkpoint(self) -> None:
        """Sharded optimizer only (a no-op otherwise): COLLECTIVE -- every rank calls it -- that makes the fp32 masters and both Adam
        moments whole on every rank, so that ``model.state_dict()`` and ``Trainer.state_dict()`` can then be taken by any single rank
        (dvc.py:310-330 saves from the main process only)."""
        if not self.sync.shard:
            return
        self.gather_master()
        self.sync.gather_stripes(self.m)
        self.sync.gather_stripes(self.v)
        self.sync.moments_stale = False

    def state_dict(self) -> Dict:
        """Optimizer state for checkpoint / resume (dvc.py:310-330 saves optimizer.state_dict() next to the model): Adam moments (flat,
        arena order), step count, and the dropout stream position so that a resumed run draws the masks the uninterrupted one would.
        With a sharded optimizer call :meth:`prepare_checkpoint` on every rank first (this method itself is NOT a collective and raises on
        stale moments instead of deadlocking a main-process-only save)."""
        if self.sync.shard and self.sync.moments_stale:
            raise RuntimeError("Trainer.state_dict(): the Adam moments of the stripes other ranks own are stale (sharded optimizer). Call "
                               "Trainer.prepare_checkpoint() on EVERY rank first (a collective); the usual `if is_main_process(): "
                               "save(trainer.state_dict())` of dvc.py:310-330 then works unchanged.")
        return {"step_count": self.step_count, "exp_avg": self.m, "exp_avg_sq": self.v, "dropout_rng": self.eng.rng_state(),
                "arena_names": list(self.eng.arena.names)}

    def load_state_dict(self, sd: Dict) -> None:
        if list(sd["arena_names"]) != list(self.eng.arena.names):
            raise ValueError("optimizer state was saved for a different parameter layout")
        self.step_count = int(sd["step_count"])
        self.m.copy_(sd["exp_avg"]); self.v.copy_(sd["exp_avg_sq"])
        self.sync.moments_stale = False        # whole moments were just loaded on this rank
        self.eng.set_rng_state(sd["dropout_rng"])

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the (world-averaged) gradient of the last step, as a device scalar."""
        return self._gnorm_sq.sqrt() / self.world
