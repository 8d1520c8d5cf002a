"""Hand-orchestrated forward/backward of the Vid2Seq hot path as a sequence of C-ABI HIP kernel launches.

Maps to the reference as follows:
  V1-V3  model/vit.py:117-133,73-76,38-55,16-22      -> Engine.vit_forward / vit_backward
  T1-T9  model/modeling_t5.py:263-277,397-460,462-588,598-656,304-354,670-769,930-1138
                                                      -> Engine._self_attn / _cross_attn / _ffn (+ *_bwd), _stack_*
  T10/11 model/modeling_t5.py:845-868,1587-1738       -> Engine.t5_loss_forward / t5_loss_backward
  A1     model/vid2seq.py:58-98                       -> Engine.forward
  D1-D3  transformers 4.28 greedy_search (vid2seq.py:150-162) -> Engine.greedy

PyTorch supplies device buffers (torch.empty), the stream, int64 index prep (shift/mask, a few elements) and
the autograd hook; all floating-point work runs in libvid2seq_hip.so.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import lib as L
from .arena import ParamArena


def _bucket_lut(nq: int, nk: int, bidirectional: bool, num_buckets: int, max_distance: int) -> torch.Tensor:
    """int32 LUT over relative positions d = k - q, d in [-(nq-1), nk-1], computed on the host exactly as the
    reference does (modeling_t5.py:397-443: torch float32 log) so that bucket edges agree bit for bit."""
    rel = torch.arange(-(nq - 1), nk, dtype=torch.long)
    out = torch.zeros_like(rel)
    nb = num_buckets
    if bidirectional:
        nb //= 2
        out = out + (rel > 0).long() * nb
        n = rel.abs()
    else:
        n = (-rel).clamp(min=0)
    exact = nb // 2
    big = exact + (torch.log(n.float() / exact) / math.log(max_distance / exact) * (nb - exact)).long()
    big = big.clamp(max=nb - 1)
    return (out + torch.where(n < exact, n, big)).to(torch.int32)


class _Rec(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class Engine:
    def __init__(self, model, device: torch.device):
        self.model = model
        self.device = device
        self.cfg = model.cfg
        c = self.cfg
        self.d, self.inner, self.H, self.ff, self.V = c.d_model, c.inner, c.heads, c.d_ff, c.vocab
        self.ldv = (self.V + 63) // 64 * 64       # row pitch of logits / d(logits); also the padded row count of the tied embedding (arena tail)
        self.vd, self.vH, self.vmlp = model.vit_dim, model.vit_heads, model.vit_mlp
        L.lib()                                                   # fail loudly if the HIP library is missing
        order = self._arena_order()
        assert order[-1][0] == "t5_model.shared.weight"
        self.arena = ParamArena(order, device, tail_pad=(self.ldv - self.V) * self.d)
        assert self.arena.adjacent(*[self._sa("encoder", 0) + w for w in ("q.weight", "k.weight", "v.weight")])
        self._luts: Dict[Tuple[int, int, bool], torch.Tensor] = {}
        self._far: Dict[Tuple[int, int, bool], Tuple[int, int]] = {}
        # dropout stream: like the reference (torch's seeded per-process generator) the masks follow torch.manual_seed() and differ
        # between data-parallel ranks; (seed, site counter) are part of the training state (rng_state / set_rng_state)
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        self._seed = (int(torch.initial_seed()) ^ (rank * 0x9E3779B1) ^ 0x1234) & 0xFFFFFFFF
        self._site = 0
        self._ws: Dict[str, torch.Tensor] = {}
        # autograd anchor: the coarse Functions take it as an input so that their outputs get a grad_fn even though
        # parameter gradients are written straight into the arena (backward returns None for it)
        self.anchor = torch.zeros(1, device=device, requires_grad=True)
        # HIP streams: weight-gradient GEMMs are off the critical path of backward (nothing downstream reads them before
        # the optimizer), so they run on `wstream` next to the dgrad / attention kernels of the main stream and fill the
        # CUs those leave idle (tails, epilogues, small decoder/ViT launches).  `vstream` lets the temporal ViT (small
        # launches, independent of the T5 encoder) run beside the encoder in both directions (train.Trainer).
        self.overlap = True
        self.skip_grad_memset = True
        self._fresh_grads = None  # Trainer.step only: names of the weight matrices whose gradient has been written in this step (begin_grad_step)
        self.fused_head = True    # Trainer path: LM head + CE + their backward chunk by chunk inside the forward (no [B*Lo, vocab] tensor)
        self.head_rows = 2048     # decoder rows per chunk: 264 MB of fp32 logits + 132 MB of bf16 d(logits) scratch at vocab 32200
        self.head_ce_fused = True # (round 6) ... and the logits are never WRITTEN: v2s_lmhead_ce_fwd reduces each 128 x 128 logits tile to row statistics in
                                  # the GEMM epilogue (one launch over all rows), v2s_lmhead_ce_bwd recomputes the tiles of a chunk into bf16 d(logits); False =
                                  # the round-2 flow (fp32 logits chunk -> v2s_ce_fwd -> v2s_ce_bwd)
        self.pack = True          # run the text encoder on the valid (non-pad) tokens only: exact, see _pack_plan
        self.pack_dec = True      # likewise the decoder rows of pad targets (labels -100, masked as keys): see _pack_plan_dec
        self.pack_mem = True      # and the [video ; text] memory the decoder attends to: see _mem_plan
        self.wstream = torch.cuda.Stream(device=device)
        self.vstream = torch.cuda.Stream(device=device)
        self.kstream = torch.cuda.Stream(device=device)   # cross-attention K|V projections of all decoder layers (forward) / the d(memory) chain (backward)
        self.overlap_kv = True    # see decoder_forward / _cross_attn_bwd
        self.group_wgrads = True  # decoder / ViT weight gradients of one projection across the layers as ONE grouped launch (_wgrad, flush_wgrads)
        self.decode_mem_attn = 1      # generate(): cross-attention of a decode step on the encoder memory itself instead of per-layer K / V caches
                                      # (_cross_on_memory): 0 = never; 1 = greedy / sampling when the step is bandwidth-bound (>= 40 000 valid memory
                                      # keys in the batch: measured +10 % at B 128, +5 % at 58 000 = B 64, +2 % at 43 000 = B 48, -9 % at 28 000 = B 32; default); 2 = greedy / sampling
                                      # always; 3 = beam search with <= 4 beams too (slower than the grouped K/V kernel at 16 entries x 4 beams)
        self.decode_mem_attn_min_keys = 40000     # mode 1: memory positions of the call (B x S: shape, not content) from which the memory path is taken
        self.beam_on_device = True    # beam search (no sampling, <= 16 beams): hypothesis bookkeeping on the device (v2s_beam_advance): no host round trip per step
        self.group_flush_layers = 4  # ... every this many decoder layers (the launches then run beside the NEXT layers' under-filled 8192-row kernels)
        self._wgrad_groups: Dict = {}
        self.decode_fuse_tail = True  # greedy(): argmax + next embedding + step counter as one launch (v2s_argmax_step_tail)
        self.shadow_events = None # sharded optimizer: {"vit" | "enc" | "dec": event after which that group's bf16 shadow weights are whole}
        # parity taps (tests only; None = off, the default): ``dbg_logits`` = list that receives the fp32 logits [rows, vocab] of every head
        # launch (whole tensor on the autograd route, one entry per chunk of the fused head); ``dbg_tap`` = list that receives, per sublayer of
        # a stack's backward, (stack, kind, block, d(output) as it arrived, d(input) as it left)
        self.dbg_logits: Optional[List] = None
        self.dbg_tap: Optional[List] = None
        self.arena.refresh_shadow(force=True)

    # ------------------------------------------------------------------------------------------ names / arena order
    @staticmethod
    def _sa(stack: str, i: int) -> str:
        return f"t5_model.{stack}.block.{i}.layer.0.SelfAttention."

    @staticmethod
    def _ca(i: int) -> str:
        return f"t5_model.decoder.block.{i}.layer.1.EncDecAttention."

    @staticmethod
    def _ln(stack: str, i: int, j: int) -> str:
        return f"t5_model.{stack}.block.{i}.layer.{j}.layer_norm.weight"

    @staticmethod
    def _ffp(stack: str, i: int) -> str:
        return f"t5_model.{stack}.block.{i}.layer.{2 if stack == 'decoder' else 1}.DenseReluDense."

    def _arena_order(self):
        """The weight matrices in (approximately) the order their gradients complete during backward, q|k|v adjacent; then the small
        fp32-consumed parameters; the tied embedding last."""
        named = dict(self.model.named_parameters())
        order: List[str] = []
        c = self.cfg
        for stack, n in (("decoder", c.n_dec), ("encoder", c.n_enc)):
            order.append(f"t5_model.{stack}.final_layer_norm.weight")
            ffj = 2 if stack == "decoder" else 1
            for i in reversed(range(n)):
                order += [self._ffp(stack, i) + "wi.weight", self._ffp(stack, i) + "wo.weight", self._ln(stack, i, ffj)]
                if stack == "decoder":
                    order += [self._ca(i) + w for w in ("q.weight", "k.weight", "v.weight", "o.weight")]
                    order.append(self._ln(stack, i, 1))
                order += [self._sa(stack, i) + w for w in ("q.weight", "k.weight", "v.weight", "o.weight")]
                order.append(self._ln(stack, i, 0))
                if i == 0:
                    order.append(self._sa(stack, i) + "relative_attention_bias.weight")
        if self.model.proj_v2t is not None:
            order += ["proj_v2t.weight", "proj_v2t.bias"]
        order += ["visual_encoder.norm.weight", "visual_encoder.norm.bias"]
        for i in reversed(range(self.model.vit_depth)):
            p = f"visual_encoder.blocks.{i}."
            order += [p + s for s in ("mlp.fc2.weight", "mlp.fc2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "norm2.weight",
                                      "norm2.bias", "attn.proj.weight", "attn.proj.bias", "attn.qkv.weight", "attn.qkv.bias",
                                      "norm1.weight", "norm1.bias")]
        order += ["visual_encoder.pos_embed", "t5_model.shared.weight"]
        missing = set(named) - set(order)
        assert not missing, f"parameters without an arena slot: {sorted(missing)[:5]}"
        # The parameters the kernels read in FP32 (norm weights, biases, the relative-position tables: all 1-D or tiny) sit together
        # in front of the tied embedding: a data-parallel run with a sharded optimizer (train.GradSync shard=True) all-gathers only
        # the bf16 shadow of the matrices, so these few fp32-consumed tensors form ONE contiguous range that every rank reduces
        # and updates in full
        small = [n for n in order if Engine.is_small_param(n, named[n])]
        big = [n for n in order if n not in set(small) and n != "t5_model.shared.weight"]
        order = big + small + ["t5_model.shared.weight"]
        return [(n, named[n]) for n in order]

    @staticmethod
    def is_small_param(name: str, p) -> bool:
        """Consumed by the kernels as fp32 master values (Engine.arena.f): replicated, never sharded."""
        return p.dim() <= 1 or name.endswith("relative_attention_bias.weight")

    # ------------------------------------------------------------------------------------------ small helpers
    def _bf(self, *shape) -> torch.Tensor:
        return torch.empty(*shape, dtype=torch.bfloat16, device=self.device)

    def _f32(self, *shape) -> torch.Tensor:
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def _splitk_ws(self) -> torch.Tensor:
        t = self._ws.get("splitk")
        if t is None:     # 8 slices of the largest few-tile weight gradient (d_ff x d_model)
            t = self._ws["splitk"] = self._f32(8 * max(self.ff * self.d, 3 * self.inner * self.d, self.vmlp * self.vd))
        return t

    def rng_state(self) -> Tuple[int, int]:
        """(seed, site counter) of the dropout stream: save with a checkpoint, restore with :meth:`set_rng_state` to resume the
        mask sequence where it stopped."""
        return (self._seed, self._site)

    def set_rng_state(self, state) -> None:
        self._seed, self._site = int(state[0]) & 0xFFFFFFFF, int(state[1])

    def mark_dirty(self) -> None:
        """Force the bf16 shadow weights to be re-cast from the fp32 masters at the next forward.  Needed only after updates that
        bypass the Parameters' version counters (``p.data.add_``, legacy optimizers, EMA swaps): updates made through the
        Parameters themselves (optimizer.step, load_state_dict, in-place ops) are detected automatically."""
        self.arena._seen_version = -1

    def _head_ws(self) -> torch.Tensor:
        t = self._ws.get("head_splitk")
        if t is None:     # split-K slices of the [head_rows, d_model] fp32 d(hidden) chunk (main stream; the wgrad stream has its own)
            t = self._ws["head_splitk"] = self._f32(16 * self.head_rows * self.d)
        return t

    def _next_seed(self) -> int:
        self._site += 1
        return (self._seed * 0x9E3779B1 + self._site * 0x85EBCA6B) & 0xFFFFFFFF

    def _lut(self, nq: int, nk: int, bidirectional: bool) -> torch.Tensor:
        key = (nq, nk, bidirectional)
        if key not in self._luts:
            lut = _bucket_lut(nq, nk, bidirectional, self.cfg.buckets, self.cfg.max_distance)
            # far regions: largest d <= 0 such that every d' <= d shares lut[d]'s bucket, smallest d >= 1 likewise
            v = lut.tolist()
            lo_i = 0
            while lo_i + 1 < len(v) and v[lo_i + 1] == v[0] and (lo_i + 1) - (nq - 1) <= 0:
                lo_i += 1
            hi_i = len(v) - 1
            while hi_i - 1 >= 0 and v[hi_i - 1] == v[-1] and (hi_i - 1) - (nq - 1) >= 1:
                hi_i -= 1
            self._far[key] = (lo_i - (nq - 1), hi_i - (nq - 1))
            self._luts[key] = lut.to(self.device)
        return self._luts[key]

    def _drop(self, x: torch.Tensor, p: float, seed: int) -> torch.Tensor:
        if p <= 0.0:
            return x
        y = torch.empty_like(x)
        L.dropout(x, y, x.numel(), p, seed)
        return y

    # linear helpers ------------------------------------------------------------------------------------------
    def _wgrad(self, dy: torch.Tensor, x: torch.Tensor, wname: str, n_out: int, n_in: int, rows: int, ld_dy=None, ld_x=None,
               alpha: float = 1.0, shape=None, side_ok: bool = False, group: Optional[str] = None) -> None:
        """dW[n_out, n_in] += alpha * dy[rows, n_out]^T @ x[rows, n_in]  (fp32 accumulate into the gradient arena).
        Runs on the weight-gradient stream unless the target is the tied embedding (whose gradient is also written by
        the embedding scatter-add on the main stream)."""
        # Trainer.step (begin_grad_step): the first weight-gradient GEMM of a matrix in a step OVERWRITES its gradient, so the 1.1 GB of
        # matrix gradients need no memset; later contributions (second pass of the two-pass recipe, the tied embedding) accumulate
        fresh = self._fresh_grads
        acc = fresh is None or wname in fresh or wname == "t5_model.shared.weight"
        if fresh is not None:
            fresh.add(wname)

        if getattr(self, "dbg_skip_wgrad", 0):     # timing probe (tools/step_ab.py "eng:dbg_skip_wgrad=1"): what the weight-gradient stream costs the step
            return
        # ``group``: the same projection of every decoder / ViT layer (short contraction: 8192 / 3200 rows, 36-144 output tiles each) is
        # collected and launched as ONE grouped GEMM when the stack's backward is done (flush_wgrads): whole-K tiles of twelve problems
        # fill the chip without split-K and its reduce launches (v2s_gemm_grouped)
        if group is not None and self.group_wgrads and alpha == 1.0 and rows % 64 == 0 and n_out % 8 == 0 and n_in % 8 == 0:
            key = (group, n_out, n_in, rows, ld_dy if ld_dy is not None else n_out, ld_x if ld_x is not None else n_in, acc)
            self._wgrad_groups.setdefault(key, []).append((dy, x, self.arena.g(wname, shape)))
            return

        def launch():
            L.gemm(dy, x, self.arena.g(wname, shape), n_out, n_in, rows, transA=True, transB=True,
                   lda=ld_dy if ld_dy is not None else n_out, ldb=ld_x if ld_x is not None else n_in, ldc=n_in,
                   accumulate=acc, alpha=alpha, workspace=self._splitk_ws())
        if not self.overlap or (wname == "t5_model.shared.weight" and not side_ok):
            if self.overlap:      # the shared split-K workspace is owned by wstream: wait for its users first
                torch.cuda.current_stream().wait_stream(self.wstream)
            launch()
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.wstream.wait_event(ev)
        with torch.cuda.stream(self.wstream):
            launch()
        dy.record_stream(self.wstream)
        x.record_stream(self.wstream)

    def flush_wgrads(self) -> None:
        """Launch the collected weight-gradient groups (see _wgrad) on the weight-gradient stream, behind everything the current stream
        has enqueued so far (every member's operands were produced on it)."""
        groups, self._wgrad_groups = self._wgrad_groups, {}
        if not groups:
            return
        cur = torch.cuda.current_stream()
        if self.overlap:
            ev = torch.cuda.Event()
            ev.record(cur)
            self.wstream.wait_event(ev)
        with torch.cuda.stream(self.wstream if self.overlap else cur):
            for (_, n_out, n_in, rows, ld_dy, ld_x, acc), items in groups.items():
                for i in range(0, len(items), 16):
                    part = items[i:i + 16]
                    L.gemm_grouped([t[0] for t in part], [t[1] for t in part], [t[2] for t in part], n_out, n_in, rows,
                                   lda=ld_dy, ldb=ld_x, ldc=n_in, accumulate=acc)
        if self.overlap:
            for items in groups.values():
                for dy, x, _ in items:
                    dy.record_stream(self.wstream); x.record_stream(self.wstream)

    def join_wgrads(self) -> None:
        """Make the current stream wait for every weight-gradient GEMM issued so far."""
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self.wstream)

    def _dgrad(self, dy: torch.Tensor, w: torch.Tensor, rows: int, n_in: int, n_out: int, out=None, ld_dy=None, **epi):
        """dx[rows, n_in] = dy[rows, n_out] @ W[n_out, n_in]."""
        dx = out if out is not None else self._bf(rows, n_in)
        L.gemm(dy, w, dx, rows, n_in, n_out, transB=True, lda=ld_dy if ld_dy is not None else n_out, ldb=n_in, **epi)
        return dx

    # ========================================================================================== T5 sublayers (forward)
    def _self_attn(self, stack: str, i: int, h, B: int, N: int, bias_diag, key_mask, causal: bool, p: float, tape, pack=None):
        """``pack`` = (seq_off int32 [B+1] on the device, total rows): the rows of ``h`` are the valid tokens only
        (sequence b = rows seq_off[b]..seq_off[b+1]), N stays the nominal length."""
        a = self.arena
        M, d, inner = (pack[1] if pack is not None else B * N), self.d, self.inner
        n = self._bf(M, d); rstd = self._f32(M)
        lnw = a.f(self._ln(stack, i, 0))
        L.rmsnorm_fwd(h, lnw, n, rstd, M, d, self.cfg.eps)
        qkv = self._bf(M, 3 * inner)
        L.gemm(n, a.w(self._sa(stack, i) + "q.weight", (3 * inner, d)), qkv, M, 3 * inner, d)
        ctx = self._bf(M, inner)
        ml = self._f32(B, self.H, N, 2) if tape is not None else None
        seed_a = self._next_seed()
        st = (N * 3 * inner, 3 * inner)
        args = L.attn_args(B, self.H, N, N, qkv, qkv[:, inner:], qkv[:, 2 * inner:], ctx, st, st, st, (N * inner, inner),
                           ml=ml, scale=1.0, bias_diag=bias_diag, key_mask=None if pack is not None else key_mask, causal=causal,
                           dropout_p=p, dropout_seed=seed_a, seq_off=pack[0] if pack is not None else None)
        L.attn_fwd(args)
        out = self._bf(M, d)
        seed_o = self._next_seed()
        L.gemm(ctx, a.w(self._sa(stack, i) + "o.weight"), out, M, d, inner, residual=h, dropout_p=p, dropout_seed=seed_o)
        if tape is not None:
            tape.append(_Rec(kind="self", stack=stack, i=i, h=h, n=n, rstd=rstd, qkv=qkv, ctx=ctx, ml=ml, args=args,
                             B=B, N=N, M=M, p=p, seed_o=seed_o))
        return out

    def _cross_attn(self, i: int, h, B: int, Nq: int, mem, S: int, mem_mask, p: float, tape, kv=None, pack=None, kpack=None):
        """``pack`` = (seq_off, rows): the query rows are packed (padding-free decoder).  ``kpack`` = (kv_off, rows, real rows): the
        memory is packed too (_mem_plan), else it is dense [B, S] with ``mem_mask``."""
        a = self.arena
        Mq, Mk, d, inner = (pack[1] if pack is not None else B * Nq), (kpack[1] if kpack is not None else B * S), self.d, self.inner
        n = self._bf(Mq, d); rstd = self._f32(Mq)
        L.rmsnorm_fwd(h, a.f(self._ln("decoder", i, 1)), n, rstd, Mq, d, self.cfg.eps)
        q = self._bf(Mq, inner)
        L.gemm(n, a.w(self._ca(i) + "q.weight"), q, Mq, inner, d)
        if kv is None:
            kv = self._bf(Mk, 2 * inner)
            L.gemm(mem, a.w(self._ca(i) + "k.weight", (2 * inner, d)), kv, Mk, 2 * inner, d)
        elif isinstance(kv, tuple):            # projected ahead on the K|V stream (_cross_kv_ahead)
            kv, ready = kv
            torch.cuda.current_stream().wait_event(ready)
        ctx = self._bf(Mq, inner)
        ml = self._f32(B, self.H, Nq, 2) if tape is not None else None
        seed_a = self._next_seed()
        kst = (S * 2 * inner, 2 * inner)
        args = L.attn_args(B, self.H, Nq, S, q, kv, kv[:, inner:], ctx, (Nq * inner, inner), kst, kst, (Nq * inner, inner),
                           ml=ml, scale=1.0, key_mask=None if kpack is not None else mem_mask, dropout_p=p, dropout_seed=seed_a,
                           seq_off=pack[0] if pack is not None else None, seq_q_only=pack is not None,
                           kv_seq_off=kpack[0] if kpack is not None else None)
        L.attn_fwd(args)
        out = self._bf(Mq, d)
        seed_o = self._next_seed()
        L.gemm(ctx, a.w(self._ca(i) + "o.weight"), out, Mq, d, inner, residual=h, dropout_p=p, dropout_seed=seed_o)
        if tape is not None:
            tape.append(_Rec(kind="cross", i=i, h=h, n=n, rstd=rstd, q=q, kv=kv, ctx=ctx, ml=ml, args=args, mem=mem,
                             B=B, Nq=Nq, S=S, p=p, seed_o=seed_o, Mq=Mq, Mk=Mk, k_real=kpack[2] if kpack is not None else Mk))
        return out

    def _ffn(self, stack: str, i: int, h, M: int, p: float, tape):
        a = self.arena
        d, ff = self.d, self.ff
        j = 2 if stack == "decoder" else 1
        n = self._bf(M, d); rstd = self._f32(M)
        L.rmsnorm_fwd(h, a.f(self._ln(stack, i, j)), n, rstd, M, d, self.cfg.eps)
        u = self._bf(M, ff)
        seed_u, seed_o = self._next_seed(), self._next_seed()
        L.gemm(n, a.w(self._ffp(stack, i) + "wi.weight"), u, M, ff, d, act=L.ACT_RELU, dropout_p=p, dropout_seed=seed_u)
        out = self._bf(M, d)
        L.gemm(u, a.w(self._ffp(stack, i) + "wo.weight"), out, M, d, ff, residual=h, dropout_p=p, dropout_seed=seed_o)
        if tape is not None:
            tape.append(_Rec(kind="ffn", stack=stack, i=i, h=h, n=n, rstd=rstd, u=u, M=M, p=p, seed_u=seed_u, seed_o=seed_o))
        return out

    def _final_norm(self, stack: str, h, M: int, p: float, tape):
        n = self._bf(M, self.d); rstd = self._f32(M)
        L.rmsnorm_fwd(h, self.arena.f(f"t5_model.{stack}.final_layer_norm.weight"), n, rstd, M, self.d, self.cfg.eps)
        seed = self._next_seed()
        out = self._drop(n, p, seed)
        if tape is not None:
            tape.append(_Rec(kind="final", stack=stack, h=h, rstd=rstd, M=M, p=p, seed=seed))
        return out

    def _embed(self, ids: torch.Tensor, p: float, tape):
        n = ids.numel()
        out = self._bf(n, self.d)
        seed = self._next_seed()
        flat = ids.reshape(-1).contiguous()
        L.embed_fwd(flat, self.arena.w("t5_model.shared.weight"), out, n, self.d, self.V, p, seed)
        if tape is not None:
            tape.append(_Rec(kind="embed", ids=flat, n=n, p=p, seed=seed))
        return out

    # ========================================================================================== T5 sublayers (backward)
    def _norm_bwd_next(self, x, wname, rstd, dn, dh, rows, nxt):
        """RMSNorm backward of a sublayer + the residual-gradient add.  ``nxt`` = (p, seed) of the sublayer the backward pass visits
        next: its input gradient dropout(dx) is produced by the same launch (second output) instead of a separate elementwise pass over
        the residual-stream gradient.  Returns (dx, dropout(dx) or None)."""
        a = self.arena
        dx = self._bf(rows, self.d)
        dxd = self._bf(rows, self.d) if (nxt is not None and nxt[0] > 0.0) else None
        if dxd is not None:
            L.rmsnorm_bwd(x, a.f(wname), rstd, dn, dx, dh, a.g(wname), rows, self.d, dx_drop=dxd, dropout_p=nxt[0], dropout_seed=nxt[1])
        else:
            L.rmsnorm_bwd(x, a.f(wname), rstd, dn, dx, dh, a.g(wname), rows, self.d)
        return dx, dxd

    def _self_attn_bwd(self, r, dh, dbias_diag, df=None, nxt=None):
        a = self.arena
        B, N, M, d, inner = r.B, r.N, r.M, self.d, self.inner
        sa = self._sa(r.stack, r.i)
        if df is None:
            df = self._drop(dh, r.p, r.seed_o)
        grp = "dec.sa." if r.stack == "decoder" else None           # the encoder's (32000-row contraction) stay single launches
        self._wgrad(df, r.ctx, sa + "o.weight", d, inner, M, group=grp and grp + "o")
        dctx = self._dgrad(df, a.w(sa + "o.weight"), M, inner, d)
        dqkv = self._bf(M, 3 * inner)
        delta = self._f32(B, self.H, N, 4)       # row statistics handed from the dQ to the dK/dV kernel (v2s_attn_bwd workspace)
        st = (N * 3 * inner, 3 * inner)
        self._lut(N, N, r.stack == "encoder")
        L.attn_bwd(r.args, dctx, (N * inner, inner), delta, dqkv, dqkv[:, inner:], dqkv[:, 2 * inner:], st, st, st,
                   dbias_diag=dbias_diag, far=self._far[(N, N, r.stack == "encoder")])
        self._wgrad(dqkv, r.n, sa + "q.weight", 3 * inner, d, M, shape=(3 * inner, d), group=grp and grp + "qkv")
        dn = self._dgrad(dqkv, a.w(sa + "q.weight", (3 * inner, d)), M, d, 3 * inner)
        return self._norm_bwd_next(r.h, self._ln(r.stack, r.i, 0), r.rstd, dn, dh, M, nxt)

    def _cross_attn_bwd(self, r, dh, dmem, first: bool, df=None, nxt=None):
        a = self.arena
        B, Nq, S, d, inner = r.B, r.Nq, r.S, self.d, self.inner
        Mq, Mk = r.Mq, r.Mk
        ca = self._ca(r.i)
        if df is None:
            df = self._drop(dh, r.p, r.seed_o)
        self._wgrad(df, r.ctx, ca + "o.weight", d, inner, Mq, group="dec.ca.o")
        dctx = self._dgrad(df, a.w(ca + "o.weight"), Mq, inner, d)
        dq = self._bf(Mq, inner); dkv = self._bf(Mk, 2 * inner)
        if r.k_real < Mk:
            dkv[r.k_real:].zero_()        # filler rows of a packed memory belong to no sequence: the kernels never write them
        delta = self._f32(B, self.H, Nq, 4)
        kst = (S * 2 * inner, 2 * inner)
        L.attn_bwd(r.args, dctx, (Nq * inner, inner), delta, dq, dkv, dkv[:, inner:], (Nq * inner, inner), kst, kst)
        self._wgrad(dq, r.n, ca + "q.weight", inner, d, Mq, group="dec.ca.q")
        dn = self._dgrad(dq, a.w(ca + "q.weight"), Mq, d, inner)
        self._wgrad(dkv, r.mem, ca + "k.weight", 2 * inner, d, Mk, shape=(2 * inner, d))
        # d(memory) of this layer: a chain of residual epilogues into the accumulator ``dmem``, in place.  Nothing in the decoder's backward reads it,
        # so it runs on the K|V stream beside the decoder's small launches; t5_loss_backward joins before using it.  (Round 5 tried twelve plain GEMMs
        # into separate slices + one fp32 sum: each GEMM 99 -> 71 us alone, +-0 in the step, +650 MB -- removed in round 6.)
        out = dmem
        epi = {} if first else dict(residual=dmem)
        if self.overlap and self.overlap_kv:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.kstream.wait_event(ev)
            with torch.cuda.stream(self.kstream):
                self._dgrad(dkv, a.w(ca + "k.weight", (2 * inner, d)), Mk, d, 2 * inner, out=out, **epi)
            dkv.record_stream(self.kstream)
            out.record_stream(self.kstream)
        else:
            self._dgrad(dkv, a.w(ca + "k.weight", (2 * inner, d)), Mk, d, 2 * inner, out=out, **epi)
        return self._norm_bwd_next(r.h, self._ln("decoder", r.i, 1), r.rstd, dn, dh, Mq, nxt)

    def _ffn_bwd(self, r, dh, df=None, nxt=None):
        a = self.arena
        M, d, ff = r.M, self.d, self.ff
        fp = self._ffp(r.stack, r.i)
        if df is None:
            df = self._drop(dh, r.p, r.seed_o)
        grp = "dec.ff." if r.stack == "decoder" else None
        self._wgrad(df, r.u, fp + "wo.weight", d, ff, M, group=grp and grp + "wo")
        # (timing probe, round 3: without the ReLU-mask operand z this launch would take the deferred-epilogue kernel and the step
        # 55.48 -> 54.33 ms; a byte mask written by the wi forward would have to be packed inside that kernel's slack-free write-out phases)
        du = self._dgrad(df, a.w(fp + "wo.weight"), M, ff, d, dact=L.ACT_RELU, z=r.u, dropout_p=r.p, dropout_seed=r.seed_u)
        self._wgrad(du, r.n, fp + "wi.weight", ff, d, M, group=grp and grp + "wi")
        dn = self._dgrad(du, a.w(fp + "wi.weight"), M, d, ff)
        return self._norm_bwd_next(r.h, self._ln(r.stack, r.i, 2 if r.stack == "decoder" else 1), r.rstd, dn, dh, M, nxt)

    def _final_norm_bwd(self, r, dout, nxt=None):
        dn = self._drop(dout, r.p, r.seed)
        return self._norm_bwd_next(r.h, f"t5_model.{r.stack}.final_layer_norm.weight", r.rstd, dn, None, r.M, nxt)

    def _embed_bwd(self, r, dh):
        # the LM head's weight-gradient chunks accumulate into the same tensor on the weight-gradient stream (fused head, side_ok): the
        # non-atomic read-modify-write epilogue of those GEMMs must be complete before the scatter-add touches the rows
        ev = getattr(self, "_head_wgrad_ev", None)
        if ev is not None and self.overlap:
            torch.cuda.current_stream().wait_event(ev)
        L.embed_bwd(r.ids, dh, self.arena.g("t5_model.shared.weight"), r.n, self.d, self.V, r.p, r.seed)

    # ========================================================================================== T5 stacks
    def _bias_diag(self, stack: str, nq: int, nk: int) -> Tuple[torch.Tensor, torch.Tensor]:
        lut = self._lut(nq, nk, stack == "encoder")
        diag = self._f32(self.H, nq + nk - 1)
        L.bias_diag_fwd(self.arena.f(self._sa(stack, 0) + "relative_attention_bias.weight"), lut, diag, self.H, nq + nk - 1,
                        self.cfg.buckets)
        return diag, lut

    def encoder_forward(self, ids: torch.Tensor, mask_u8: torch.Tensor, p: float, tape, pack=None):
        """``pack`` = (seq_off, rows, tok_rows, n_seq, Lx, n_valid) from _pack_plan: run the stack on the valid tokens only (``ids``
        is then the 1-D packed id vector); returns [rows, d] (the first n_valid rows are the tokens) instead of [B*Lx, d]."""
        if pack is not None:
            B, Lx, M = pack[3], pack[4], pack[1]
        else:
            B, Lx = ids.shape
            M = B * Lx
        h = self._embed(ids, p, tape)
        diag, _ = self._bias_diag("encoder", Lx, Lx)
        for i in range(self.cfg.n_enc):
            h = self._self_attn("encoder", i, h, B, Lx, diag, mask_u8, False, p, tape, pack=pack[:2] if pack is not None else None)
            h = self._ffn("encoder", i, h, M, p, tape)
        return self._final_norm("encoder", h, M, p, tape)

    # ---- padding-free encoder ------------------------------------------------------------------------------------------
    def _pack_plan(self, input_ids: torch.Tensor, input_mask: torch.Tensor, lens=None):
        """Row bookkeeping for running the text encoder on the valid tokens only.  The reference computes the rows of pad tokens and
        then masks them as keys everywhere (modeling_t5.py:996,1005), so nothing downstream depends on them: dropping them is exact
        for the loss and every gradient.  Needs the usual right-padded batch (mask = a prefix per row, dvc.py:44-53 + the collate
        of dataset/dvc_dataset.py:179-226); anything else -> None (dense path).  ``lens``: per-row valid lengths if the caller has
        them on the host (DeviceBatcher / bench), else they are read back from the mask (one small device->host sync)."""
        B, Lx = input_ids.shape
        if lens is None:
            m = input_mask.to(torch.bool)
            prefix = bool((m[:, 1:] <= m[:, :-1]).all().item()) if Lx > 1 else True
            if not prefix:
                return None
            lens = m.sum(1).tolist()
        lens = [int(x) for x in lens]
        if min(lens) < 1 or sum(lens) == B * Lx:
            return None                                   # nothing to drop (or an empty row: keep the reference's dense semantics)
        key = (B, Lx, tuple(lens))
        plans = self._ws.setdefault("pack_plans", {})
        plan = plans.get(key)
        if plan is None:
            if len(plans) >= 24:                          # (tapes keep the plans they use alive: t5_loss_forward keep_alive)
                plans.clear()
            total = int(sum(lens))
            padded = (total + 63) // 64 * 64          # row count a multiple of 64: the weight-gradient GEMMs contract over it (K % 64)
            # the < 64 filler rows form extra dummy sequences (each no longer than the nominal length Lx that sizes the attention
            # grid and its bias window): they attend among themselves, stay finite, receive a zero gradient and are never
            # scattered into the memory
            seqs, fill = list(lens), padded - total
            while fill > 0:
                seqs.append(min(fill, Lx)); fill -= seqs[-1]
            off = np.zeros(len(seqs) + 1, dtype=np.int32)
            off[1:] = np.cumsum(seqs)
            rows = np.concatenate([np.arange(n, dtype=np.int64) + b * Lx for b, n in enumerate(lens)] +
                                  [np.zeros(padded - total, dtype=np.int64)])
            plan = plans[key] = (torch.from_numpy(off).to(self.device), padded, torch.from_numpy(rows).to(self.device), len(seqs), total)
        return (plan[0], plan[1], plan[2], plan[3], Lx, plan[4], tuple(lens))

    def _mem_plan(self, lens, T: int):
        """Padding-free [video ; text] memory: sample b occupies rows [off[b], off[b+1]) = its T visual rows followed by its valid text
        rows; the row count is rounded up to a multiple of 64 with zero rows that belong to no sample (the cross-attention key ranges
        never reach them).  Returns (kv_off int32 [B+1], rows, vis_pos [B*T], txt_pos [sum(lens)], real_rows), device tensors cached."""
        key = ("mem", T, tuple(lens))
        plans = self._ws.setdefault("pack_plans", {})
        plan = plans.get(key)
        if plan is None:
            if len(plans) >= 24:                          # bounded like _pack_plan's (tapes keep the plans they use alive)
                plans.clear()
            B = len(lens)
            off = np.zeros(B + 1, dtype=np.int32)
            off[1:] = np.cumsum([T + n for n in lens])
            total = int(off[-1])
            vis_pos = (off[:B, None].astype(np.int64) + np.arange(T, dtype=np.int64)[None, :]).reshape(-1)
            txt_pos = np.concatenate([off[b] + T + np.arange(n, dtype=np.int64) for b, n in enumerate(lens)])
            dev = self.device
            plan = plans[key] = (torch.from_numpy(off).to(dev), (total + 63) // 64 * 64, torch.from_numpy(vis_pos).to(dev),
                                 torch.from_numpy(txt_pos).to(dev), total)
        return plan

    def _pack_plan_dec(self, output_mask: torch.Tensor, lens=None):
        """Row bookkeeping for running the decoder on the rows of real targets only.  Decoder position j of a sample consumes target
        j-1 and predicts target j; positions at and beyond the sample's target length are pad inputs with label -100 that the reference
        masks as keys (vid2seq.py:92, modeling_t5.py:996): nothing that reaches the loss reads them.  Unlike the encoder the stack is
        causal, so the row count is brought to a multiple of 64 (the weight-gradient GEMMs contract over it) by simply KEEPING a few of
        the pad positions at the end of sequences that have them -- they come after every valid row of their sequence, so no valid
        query ever sees them -- instead of adding dummy sequences.  Returns (seq_off, rows, tok_rows, B, Lo) or None (dense path)."""
        B, Lo = output_mask.shape
        if lens is None:
            m = output_mask.to(torch.bool)
            if Lo > 1 and not bool((m[:, 1:] <= m[:, :-1]).all().item()):
                return None
            lens = m.sum(1).tolist()
        lens = [max(1, int(x)) for x in lens]           # an empty target row still feeds the start token (row 0)
        total = sum(lens)
        fill = (-total) % 64
        if total + fill >= B * Lo:
            return None
        key = ("dec", B, Lo, tuple(lens))
        plans = self._ws.setdefault("pack_plans", {})
        plan = plans.get(key)
        if plan is None:
            if len(plans) >= 24:                            # bounded: without a text encoder (use_speech=False) _pack_plan never empties it
                plans.clear()
            ext = list(lens)
            for b in range(B - 1, -1, -1):              # pad positions kept as rows, from the last sequence backwards
                take = min(fill, Lo - ext[b])
                ext[b] += take; fill -= take
                if fill == 0:
                    break
            off = np.zeros(B + 1, dtype=np.int32)
            off[1:] = np.cumsum(ext)
            rows = np.concatenate([np.arange(n, dtype=np.int64) + b * Lo for b, n in enumerate(ext)])
            plan = plans[key] = (torch.from_numpy(off).to(self.device), int(off[-1]), torch.from_numpy(rows).to(self.device))
        return (plan[0], plan[1], plan[2], B, Lo)

    def _cross_kv_ahead(self, mem, Mk: int):
        """The K|V projections of the memory for ALL decoder layers depend on nothing but the encoder output: issued on their own
        stream before the decoder starts, the big [Mk, 2*inner] GEMMs run beside the decoder's small (8192-row: 384..1152 tiles on 512
        slots) launches instead of in between them.  Returns [(kv_i, ready event_i)] or None (single-stream execution)."""
        if not (self.overlap and self.overlap_kv):
            return None
        a, d, inner = self.arena, self.d, self.inner
        main = torch.cuda.current_stream()
        self.kstream.wait_stream(main)
        out = []
        with torch.cuda.stream(self.kstream):
            self.wait_shadow("dec")
            for i in range(self.cfg.n_dec):
                kv = self._bf(Mk, 2 * inner)
                L.gemm(mem, a.w(self._ca(i) + "k.weight", (2 * inner, d)), kv, Mk, 2 * inner, d)
                ev = torch.cuda.Event()
                ev.record(self.kstream)
                kv.record_stream(main)
                out.append((kv, ev))
        mem.record_stream(self.kstream)
        return out

    def decoder_forward(self, dec_ids, dec_mask_u8, mem, S: int, mem_mask_u8, p: float, tape, pack=None, kpack=None):
        """``pack`` = (seq_off, rows, tok_rows, B, Lo) from _pack_plan_dec: ``dec_ids`` is then the 1-D packed id vector and the result
        has ``rows`` rows."""
        self.wait_shadow("dec")
        if pack is not None:
            B, Lo, M = pack[3], pack[4], pack[1]
            ahead = self._cross_kv_ahead(mem, kpack[1] if kpack is not None else B * S)
            h = self._embed(dec_ids, p, tape)
            diag, _ = self._bias_diag("decoder", Lo, Lo)
            for i in range(self.cfg.n_dec):
                h = self._self_attn("decoder", i, h, B, Lo, diag, None, True, p, tape, pack=pack[:2])
                h = self._cross_attn(i, h, B, Lo, mem, S, mem_mask_u8, p, tape, pack=pack[:2], kpack=kpack,
                                     kv=ahead[i] if ahead is not None else None)
                h = self._ffn("decoder", i, h, M, p, tape)
            return self._final_norm("decoder", h, M, p, tape)
        B, Lo = dec_ids.shape
        ahead = self._cross_kv_ahead(mem, kpack[1] if kpack is not None else B * S)
        h = self._embed(dec_ids, p, tape)
        diag, _ = self._bias_diag("decoder", Lo, Lo)
        for i in range(self.cfg.n_dec):
            h = self._self_attn("decoder", i, h, B, Lo, diag, dec_mask_u8, True, p, tape)
            h = self._cross_attn(i, h, B, Lo, mem, S, mem_mask_u8, p, tape, kpack=kpack, kv=ahead[i] if ahead is not None else None)
            h = self._ffn("decoder", i, h, B * Lo, p, tape)
        return self._final_norm("decoder", h, B * Lo, p, tape)

    def _stack_backward(self, tape: List[_Rec], dout, stack: str, nq: int, dmem=None, layer_done=None):
        """Replays ``tape`` (records of one stack) in reverse.  Returns nothing: parameter gradients go to the arena,
        the cross-attention memory gradient to ``dmem``."""
        a = self.arena
        lut = self._lut(nq, nq, stack == "encoder")
        ddiag = torch.zeros(self.H, 2 * nq - 1, dtype=torch.float32, device=self.device)
        dh, df = dout, None
        first_cross = True
        recs = list(reversed(tape))
        for idx, r in enumerate(recs):
            # (p, seed) of the output dropout of the sublayer visited next: its gradient operand dropout(dh) comes out of this
            # sublayer's norm backward (second output)
            nr = recs[idx + 1] if idx + 1 < len(recs) else None
            nxt = (nr.p, nr.seed_o) if (nr is not None and nr.kind in ("ffn", "cross", "self")) else None
            dh_arrived = dh
            if r.kind == "final":
                dh, df = self._final_norm_bwd(r, dh, nxt)
            elif r.kind == "ffn":
                dh, df = self._ffn_bwd(r, dh, df, nxt)
            elif r.kind == "cross":
                dh, df = self._cross_attn_bwd(r, dh, dmem, first_cross, df, nxt)
                first_cross = False
            elif r.kind == "self":
                dh, df = self._self_attn_bwd(r, dh, ddiag, df, nxt)
                if stack == "decoder" and r.i % self.group_flush_layers == 0:
                    self.flush_wgrads()               # the collected decoder layers' weight gradients: one grouped launch per projection
                if layer_done is not None:        # all parameter gradients of block r.i are enqueued (except block 0's bias table)
                    layer_done(r.i)
            elif r.kind == "embed":
                self._embed_bwd(r, dh)
            if self.dbg_tap is not None and r.kind != "embed":
                self.dbg_tap.append((stack, r.kind, r.get("i", -1), dh_arrived, dh))
        self.flush_wgrads()
        L.bias_bucket_bwd(ddiag, lut, a.g(self._sa(stack, 0) + "relative_attention_bias.weight"), self.H, 2 * nq - 1,
                          self.cfg.buckets)

    # ========================================================================================== temporal ViT
    def vit_forward(self, video: torch.Tensor, tape):
        """model/vit.py:117-133.  video: [B, T, C] (fp32 or bf16) -> bf16 [B*T, d_model]."""
        a, m = self.arena, self.model
        B, T, C = video.shape
        assert C == self.vd, f"feature dim {C} != embed_dim {self.vd}"
        self.wait_shadow("vit")
        M, p = B * T, (m.vis_drop if m.training else 0.0)
        if video.dtype != torch.bfloat16:          # fp32 (the reference's features), fp16, fp64, ...: through fp32 to bf16
            xb = self._bf(M, C)
            L.cast_bf16(video.contiguous().float().view(-1), xb, M * C)
        else:
            xb = video.contiguous().view(M, C)
        pos = a.w("visual_encoder.pos_embed", (m.num_features, C))
        idx = None
        if T != m.num_features:       # nearest-neighbour resize (vit.py:119-123): cold path, plain index ops
            idx = torch.floor(torch.arange(T, device=self.device, dtype=torch.float32) * (m.num_features / T)).long()
            idx = idx.clamp_(max=m.num_features - 1)
            pos = pos.index_select(0, idx).contiguous()
        x = self._bf(M, C)
        L.add_bcast(xb, pos, x, M * C, T * C)
        seed0 = self._next_seed()
        x = self._drop(x, p, seed0)
        recs = []
        Hh, mlp = self.vH, self.vmlp
        for i in range(m.vit_depth):
            pre = f"visual_encoder.blocks.{i}."
            n1 = self._bf(M, C); mean1 = self._f32(M); rstd1 = self._f32(M)
            L.layernorm_fwd(x, a.f(pre + "norm1.weight"), a.f(pre + "norm1.bias"), n1, mean1, rstd1, M, C, 1e-5)
            qkv = self._bf(M, 3 * C)
            L.gemm(n1, a.w(pre + "attn.qkv.weight"), qkv, M, 3 * C, C, bias=a.f(pre + "attn.qkv.bias"))
            ctx = self._bf(M, C)
            ml = self._f32(B, Hh, T, 2) if tape is not None else None
            seed_a, seed_p, seed_u, seed_2 = (self._next_seed() for _ in range(4))
            st = (T * 3 * C, 3 * C)
            args = L.attn_args(B, Hh, T, T, qkv, qkv[:, C:], qkv[:, 2 * C:], ctx, st, st, st, (T * C, C), ml=ml,
                               scale=(C // Hh) ** -0.5, dropout_p=p, dropout_seed=seed_a)
            L.attn_fwd(args)
            x1 = self._bf(M, C)
            L.gemm(ctx, a.w(pre + "attn.proj.weight"), x1, M, C, C, bias=a.f(pre + "attn.proj.bias"), residual=x,
                   dropout_p=p, dropout_seed=seed_p)
            n2 = self._bf(M, C); mean2 = self._f32(M); rstd2 = self._f32(M)
            L.layernorm_fwd(x1, a.f(pre + "norm2.weight"), a.f(pre + "norm2.bias"), n2, mean2, rstd2, M, C, 1e-5)
            u = self._bf(M, mlp); upre = self._bf(M, mlp) if tape is not None else None
            L.gemm(n2, a.w(pre + "mlp.fc1.weight"), u, M, mlp, C, bias=a.f(pre + "mlp.fc1.bias"), act=L.ACT_GELU, pre=upre,
                   dropout_p=p, dropout_seed=seed_u)
            x2 = self._bf(M, C)
            L.gemm(u, a.w(pre + "mlp.fc2.weight"), x2, M, C, mlp, bias=a.f(pre + "mlp.fc2.bias"), residual=x1,
                   dropout_p=p, dropout_seed=seed_2)
            if tape is not None:
                recs.append(_Rec(pre=pre, x=x, n1=n1, mean1=mean1, rstd1=rstd1, qkv=qkv, ctx=ctx, ml=ml, args=args, x1=x1, n2=n2,
                                 mean2=mean2, rstd2=rstd2, u=u, upre=upre, seed_p=seed_p, seed_u=seed_u, seed_2=seed_2))
            x = x2
        out = self._bf(M, C); meanf = self._f32(M); rstdf = self._f32(M)
        L.layernorm_fwd(x, a.f("visual_encoder.norm.weight"), a.f("visual_encoder.norm.bias"), out, meanf, rstdf, M, C, 1e-5)
        vis = out
        if m.proj_v2t is not None:
            vis = self._bf(M, self.d)
            L.gemm(out, a.w("proj_v2t.weight"), vis, M, self.d, C, bias=a.f("proj_v2t.bias"))
        if tape is not None:
            tape.update(B=B, T=T, M=M, p=p, seed0=seed0, idx=idx, recs=recs, xf=x, meanf=meanf, rstdf=rstdf, normed=out)
        return vis

    def vit_backward(self, tape, dvis: torch.Tensor) -> None:
        a, m = self.arena, self.model
        B, T, M, p, C, mlp, Hh = tape["B"], tape["T"], tape["M"], tape["p"], self.vd, self.vmlp, self.vH
        dvis = dvis.contiguous().view(M, -1)
        if m.proj_v2t is not None:
            L.colsum(dvis, M, self.d, a.g("proj_v2t.bias"))
            self._wgrad(dvis, tape["normed"], "proj_v2t.weight", self.d, C, M)
            dvis = self._dgrad(dvis, a.w("proj_v2t.weight"), M, C, self.d)
        recs = list(reversed(tape["recs"]))

        def ln_bwd(x, pre_norm, mean, rstd, dn, dres, nxt_seed):
            """LayerNorm backward (+ residual-gradient add) with dropout(dx) for the next consumer as a second output"""
            dx_ = self._bf(M, C)
            dxd = self._bf(M, C) if p > 0.0 else None
            kw = dict(dx_drop=dxd, dropout_p=p, dropout_seed=nxt_seed) if dxd is not None else {}
            L.layernorm_bwd(x, a.f(pre_norm + "weight"), mean, rstd, dn, dx_, dres, a.g(pre_norm + "weight"), a.g(pre_norm + "bias"), M, C, **kw)
            return dx_, (dxd if dxd is not None else dx_)

        first_seed = recs[0].seed_2 if recs else tape["seed0"]
        dx, df2 = ln_bwd(tape["xf"], "visual_encoder.norm.", tape["meanf"], tape["rstdf"], dvis, None, first_seed)
        for idx, r in enumerate(recs):
            pre = r.pre
            d_arrived = dx
            L.colsum(df2, M, C, a.g(pre + "mlp.fc2.bias"))
            self._wgrad(df2, r.u, pre + "mlp.fc2.weight", C, mlp, M, group="vit.fc2")
            du = self._dgrad(df2, a.w(pre + "mlp.fc2.weight"), M, mlp, C, dact=L.ACT_GELU, z=r.upre, dropout_p=p,
                             dropout_seed=r.seed_u)
            L.colsum(du, M, mlp, a.g(pre + "mlp.fc1.bias"))
            self._wgrad(du, r.n2, pre + "mlp.fc1.weight", mlp, C, M, group="vit.fc1")
            dn2 = self._dgrad(du, a.w(pre + "mlp.fc1.weight"), M, C, mlp)
            dx1, df1 = ln_bwd(r.x1, pre + "norm2.", r.mean2, r.rstd2, dn2, dx, r.seed_p)
            L.colsum(df1, M, C, a.g(pre + "attn.proj.bias"))
            self._wgrad(df1, r.ctx, pre + "attn.proj.weight", C, C, M, group="vit.proj")
            dctx = self._dgrad(df1, a.w(pre + "attn.proj.weight"), M, C, C)
            dqkv = self._bf(M, 3 * C); delta = self._f32(B, Hh, T, 4)
            st = (T * 3 * C, 3 * C)
            L.attn_bwd(r.args, dctx, (T * C, C), delta, dqkv, dqkv[:, C:], dqkv[:, 2 * C:], st, st, st)
            L.colsum(dqkv, M, 3 * C, a.g(pre + "attn.qkv.bias"))
            self._wgrad(dqkv, r.n1, pre + "attn.qkv.weight", 3 * C, C, M, group="vit.qkv")
            dn1 = self._dgrad(dqkv, a.w(pre + "attn.qkv.weight"), M, C, 3 * C)
            nseed = recs[idx + 1].seed_2 if idx + 1 < len(recs) else tape["seed0"]      # next block's fc2 dropout, or the input dropout
            dx, df2 = ln_bwd(r.x, pre + "norm1.", r.mean1, r.rstd1, dn1, dx1, nseed)
            if self.dbg_tap is not None:
                self.dbg_tap.append(("vit", "block", len(recs) - 1 - idx, d_arrived, dx))
        self.flush_wgrads()
        dx = df2                                                   # = dropout(dx; seed0), vit.py:126
        gpos = a.g("visual_encoder.pos_embed", (m.num_features, C))
        if tape["idx"] is None:
            L.bcast_grad(dx, gpos, M * C, T * C)
        else:
            tmp = torch.zeros(T, C, dtype=torch.float32, device=self.device)
            L.bcast_grad(dx, tmp, M * C, T * C)
            gpos.index_add_(0, tape["idx"], tmp)

    # ========================================================================================== loss head
    def t5_loss_forward(self, vis, input_ids, input_mask, output_ids, output_mask, tape, vis_ready=None, input_lens=None,
                        head_grad_scale: Optional[float] = None, output_lens=None):
        """Encoder on the ASR tokens, [video ; text] memory, decoder on the shifted targets, tied LM head and
        label-smoothed CE (vid2seq.py:63-98 -> modeling_t5.py:1587-1738).  ``vis``: bf16 [B, T, d] or None.

        ``head_grad_scale`` (Trainer: the loss coefficient d(total)/d(this loss), known before the forward): run the LM head FUSED with
        its backward, ``head_rows`` decoder rows at a time -- logits chunk -> CE statistics -> d(logits) chunk -> embedding weight
        gradient (accumulated into the arena) and d(hidden) -- so that no [B*Lo, vocab] tensor ever exists: the 1.05 GB fp32 logits
        and the 0.53 GB bf16 d(logits) of the unfused path become two scratch chunks that live in the Infinity Cache between their
        producer and consumer kernels.  Needs a tape (training) and a zeroed / accumulating gradient arena, which Trainer.step provides;
        the autograd route (loss.backward() of drop-in callers learns the upstream gradient only later) keeps the unfused head."""
        m, c = self.model, self.cfg
        train = m.training
        pe, pd = (m.enc_drop if train else 0.0), (m.dec_drop if train else 0.0)
        B = output_ids.shape[0]
        enc_tape, dec_tape = ([], []) if tape is not None else (None, None)
        self.wait_shadow("enc")             # encoder matrices + the tied embedding (the decoder's arrive under the encoder forward)
        parts, masks = [], []
        T = 0
        if m.use_video:
            T = vis.shape[1]
            parts.append(vis.view(B, T, self.d))
            masks.append(torch.ones(B, T, dtype=torch.uint8, device=self.device))
        Lx = 0
        plan = None
        if m.use_speech:
            Lx = input_ids.shape[1]
            in_mask = input_mask.to(torch.uint8).contiguous()
            plan = self._pack_plan(input_ids, input_mask, input_lens) if self.pack else None
            if plan is not None:
                enc = self.encoder_forward(input_ids.reshape(-1).index_select(0, plan[2]), None, pe, enc_tape, pack=plan)
            else:
                enc = self.encoder_forward(input_ids, in_mask, pe, enc_tape)
                parts.append(enc.view(B, Lx, self.d))
            masks.append(in_mask)
        S = T + Lx
        if vis_ready is not None:           # the ViT ran on its own stream beside the encoder
            torch.cuda.current_stream().wait_event(vis_ready)
            vis.record_stream(torch.cuda.current_stream())
        mem_rows = vis_rows = kpack = None
        if plan is not None and self.pack_mem:      # padding-free memory: [video ; valid text] rows per sample, no pad rows at all
            kv_off, Mk_p, vis_pos, txt_pos, real = self._mem_plan(plan[6], T)
            mem = torch.zeros(Mk_p, self.d, dtype=torch.bfloat16, device=self.device) if Mk_p > real else self._bf(Mk_p, self.d)
            if T:
                mem.index_copy_(0, vis_pos, parts[0].reshape(B * T, self.d))
                vis_rows = vis_pos
            mem.index_copy_(0, txt_pos, enc[:plan[5]])
            mem_rows, mem_mask, kpack = txt_pos, None, (kv_off, Mk_p, real)
        elif plan is not None:              # [video ; text] memory with the packed encoder rows scattered back; pad rows = 0
            mem3 = torch.zeros(B, S, self.d, dtype=torch.bfloat16, device=self.device)
            if T:
                mem3[:, :T] = parts[0]
            tr = plan[2][:plan[5]]
            mem_rows = tr + (torch.div(tr, Lx, rounding_mode="floor") * T + T) if T else tr
            mem = mem3.view(B * S, self.d)
            mem.index_copy_(0, mem_rows, enc[:plan[5]])
            mem_mask = torch.cat(masks, 1).contiguous() if len(masks) == 2 else masks[0]
        elif len(parts) == 2:               # torch.cat of vid2seq.py:78-79 (pure data movement)
            mem = torch.cat(parts, 1).view(B * S, self.d)
            mem_mask = torch.cat(masks, 1).contiguous()
        else:
            mem, mem_mask = parts[0].contiguous().view(B * S, self.d), masks[0]
        # integer prep (vid2seq.py:86-88, modeling_t5.py:845-868): a few KB of int64
        targets = output_ids.masked_fill(output_ids == c.pad_id, -100)
        dec_in = torch.full_like(targets, c.pad_id)
        dec_in[:, 1:] = targets[:, :-1]
        dec_in[:, 0] = c.dec_start_id
        dec_in = dec_in.masked_fill(dec_in == -100, c.pad_id)
        Lo = dec_in.shape[1]
        dplan = self._pack_plan_dec(output_mask, output_lens) if (self.pack and self.pack_dec) else None
        if dplan is not None:               # decoder rows of real targets only (plus a few kept pad rows: row count % 64 == 0)
            hs = self.decoder_forward(dec_in.reshape(-1).index_select(0, dplan[2]), None, mem, S, mem_mask, pd, dec_tape, pack=dplan, kpack=kpack)
            Md = dplan[1]
            labels = targets.reshape(-1).index_select(0, dplan[2]).contiguous()
        else:
            hs = self.decoder_forward(dec_in, output_mask.to(torch.uint8).contiguous(), mem, S, mem_mask, pd, dec_tape, kpack=kpack)
            Md = B * Lo
            labels = targets.reshape(-1).contiguous()
        alpha = self.d ** -0.5                                   # tie_word_embeddings rescale (modeling_t5.py:1709-1712)
        E = self.arena.w("t5_model.shared.weight")
        row = self._f32(Md, 2)
        acc = torch.zeros(2, dtype=torch.float32, device=self.device)      # (loss_sum, count)
        logits = dhs = None
        if head_grad_scale is not None and tape is not None and self.fused_head:
            # d(loss)/d(logits) needs 1 / (number of non-ignored targets) of the WHOLE batch before the first chunk: count it here
            gscale = (float(head_grad_scale) / (labels != -100).sum().clamp(min=1).float()).reshape(1).contiguous()
            dhs = self._bf(Md, self.d)
            R_ = min(self.head_rows, Md)
            Epad = self.arena.shadow[self.arena.offsets["t5_model.shared.weight"]:][:self.ldv * self.d].view(self.ldv, self.d)
            ce_fused = self.head_ce_fused and self.d % 64 == 0 and Md * self.d < (1 << 30) and self.ldv * self.d < (1 << 30)
            lg = self._f32(R_, self.ldv) if (not ce_fused or self.dbg_logits is not None) else None      # logits scratch, reused by every chunk (stream-ordered)
            dh32 = self._f32(R_, self.d) if not ce_fused else None
            if ce_fused:                                          # statistics of ALL rows in one launch: no logits in memory
                part = self._f32(L.lmhead_ce_workspace_floats(Md, self.ldv))
                L.lmhead_ce_fwd(hs, self.d, Epad, Md, self.V, self.ldv, self.d, alpha, labels, m.label_smoothing, part, row, acc[0:1], acc[1:2])
            for r0 in range(0, Md, R_):
                n = min(R_, Md - r0)
                dlog = self._bf(n, self.ldv)                      # fresh per chunk: the weight-gradient stream may still read the previous one
                if ce_fused:
                    L.lmhead_ce_bwd(hs[r0:r0 + n], self.d, Epad, n, self.V, self.ldv, self.d, alpha, labels[r0:r0 + n], row[r0:r0 + n], m.label_smoothing,
                                    gscale, dlog, self.ldv)
                    if self.dbg_logits is not None:               # (tests only: the logits this head never writes, from a plain GEMM)
                        L.gemm(hs[r0:r0 + n], E, lg, n, self.V, self.d, ldc=self.ldv, alpha=alpha)
                        self.dbg_logits.append(lg[:n, :self.V].clone())
                else:
                    L.gemm(hs[r0:r0 + n], E, lg, n, self.V, self.d, ldc=self.ldv, alpha=alpha)
                    L.ce_fwd(lg, self.ldv, labels[r0:r0 + n], n, self.V, m.label_smoothing, row[r0:r0 + n], acc[0:1], acc[1:2])
                    if self.dbg_logits is not None:
                        self.dbg_logits.append(lg[:n, :self.V].clone())
                    L.ce_bwd(lg, self.ldv, labels[r0:r0 + n], row[r0:r0 + n], n, self.V, m.label_smoothing, gscale, dlog, self.ldv)
                # embedding weight gradient on the weight-gradient stream (the scatter-adds into the same tensor wait for the event
                # recorded after the last chunk, see _embed_bwd); d(hidden): contraction over the padded vocabulary (zero pad columns x zero pad rows), few
                # output tiles -> fp32 split-K, then one rounding to bf16
                self._wgrad(dlog, hs[r0:r0 + n], "t5_model.shared.weight", self.V, self.d, n, ld_dy=self.ldv, alpha=alpha, side_ok=True)
                if ce_fused:      # the split-K reduction rounds to bf16 itself (no fp32 d(hidden) chunk, no cast launch)
                    L.gemm(dlog, Epad, dhs[r0:r0 + n], n, self.d, self.ldv, transB=True, lda=self.ldv, ldb=self.d, alpha=alpha, workspace=self._head_ws())
                else:
                    L.gemm(dlog, Epad, dh32[:n], n, self.d, self.ldv, transB=True, lda=self.ldv, ldb=self.d, alpha=alpha, workspace=self._head_ws())
                    L.cast_bf16(dh32[:n].view(-1), dhs[r0:r0 + n].view(-1), n * self.d)
            if self.overlap:                                      # _embed_bwd waits for this before its scatter-add into the same tensor
                self._head_wgrad_ev = torch.cuda.Event()
                self._head_wgrad_ev.record(self.wstream)
        else:
            logits = self._f32(Md, self.ldv)
            L.gemm(hs, E, logits, Md, self.V, self.d, ldc=self.ldv, alpha=alpha)
            L.ce_fwd(logits, self.ldv, labels, Md, self.V, m.label_smoothing, row, acc[0:1], acc[1:2])
            if self.dbg_logits is not None:
                self.dbg_logits.append(logits[:, :self.V].clone())
        loss = acc[0] / acc[1]
        if tape is not None:
            tape.update(enc=enc_tape, dec=dec_tape, B=B, T=T, Lx=Lx, Lo=Lo, S=S, Md=Md, hs=hs, logits=logits, labels=labels, row=row,
                        acc=acc, alpha=alpha, mem_rows=mem_rows, vis_rows=vis_rows, Mk=kpack[1] if kpack is not None else B * S,
                        mem_packed=kpack is not None, keep_alive=(plan, dplan, kpack),   # the saved attention args point into these
                        enc_rows=plan[1] if plan is not None else 0, dhs=dhs,
                        head_grad_scale=head_grad_scale)
        return loss

    def t5_loss_backward(self, tape, gloss: torch.Tensor, after_decoder=None, after_encoder=None,
                         encoder_layer_done=None) -> Optional[torch.Tensor]:
        """Returns d(loss)/d(vis) (bf16 [B, T, d]) or None.  ``after_decoder`` / ``after_encoder`` are called once the
        gradients of that stack are complete (data-parallel all-reduce hooks)."""
        m = self.model
        B, T, Lx, Lo, S, d = tape["B"], tape["T"], tape["Lx"], tape["Lo"], tape["S"], self.d
        Md = tape["Md"]
        if tape.get("dhs") is not None:                      # fused head: its backward ran inside the forward (t5_loss_forward)
            dhs = tape["dhs"]
            tape["dhs"] = None
        else:
            gscale = (gloss.reshape(1).float() / tape["acc"][1:2]).contiguous()
            dlog = self._bf(Md, self.ldv)
            L.ce_bwd(tape["logits"], self.ldv, tape["labels"], tape["row"], Md, self.V, m.label_smoothing, gscale, dlog, self.ldv)
            tape["logits"] = None
            E = self.arena.w("t5_model.shared.weight")
            self._wgrad(dlog, tape["hs"], "t5_model.shared.weight", self.V, d, Md, ld_dy=self.ldv, alpha=tape["alpha"])
            dhs = self._bf(Md, d)
            # contraction over the vocabulary padded to a multiple of 64 (zero pad columns of d(logits) x zero pad rows of the arena
            # tail): K = 32256 instead of 32200 keeps this GEMM on the LDS-DMA kernels
            Epad = self.arena.shadow[self.arena.offsets["t5_model.shared.weight"]:][:self.ldv * d].view(self.ldv, d)
            L.gemm(dlog, Epad, dhs, Md, d, self.ldv, transB=True, lda=self.ldv, ldb=d, alpha=tape["alpha"])
            del dlog
        dmem = self._bf(tape["Mk"], d)
        self._stack_backward(tape["dec"], dhs, "decoder", Lo, dmem=dmem)
        if self.overlap and self.overlap_kv:
            torch.cuda.current_stream().wait_stream(self.kstream)          # the d(memory) work of _cross_attn_bwd
        packed_mem = bool(tape.get("mem_packed"))
        dmem3 = None if packed_mem else dmem.view(B, S, d)
        dvis = None
        if m.use_video:
            if packed_mem:
                dvis = dmem.index_select(0, tape["vis_rows"]).view(B, T, d)
            else:
                dvis = dmem3[:, :T].contiguous() if Lx else dmem3
        if after_decoder is not None:
            after_decoder(dvis)               # DP hook / ViT backward may start now, beside the encoder backward
        if m.use_speech:
            if tape.get("mem_rows") is not None:          # packed encoder: gather the gradient rows of the valid tokens
                mr = tape["mem_rows"]
                denc = torch.zeros(tape["enc_rows"], d, dtype=torch.bfloat16, device=self.device)     # filler rows: zero gradient
                torch.index_select(dmem, 0, mr, out=denc[:mr.numel()])
            else:
                denc = dmem3[:, T:].contiguous().view(B * Lx, d) if T else dmem
            self._stack_backward(tape["enc"], denc, "encoder", Lx, layer_done=encoder_layer_done)
        if after_encoder is not None:
            after_encoder()
        return dvis

    # ========================================================================================== public forward
    def prepare(self, wait_shadow: bool = True) -> None:
        """Bring the bf16 shadow weights up to date and make sure param.grad views exist.  ``wait_shadow``: also wait (on the current
        stream) for a sharded optimizer's in-flight all-gather of the shadow weights; Trainer.step passes False and lets the forward
        wait group by group instead (wait_shadow("vit" | "enc" | "dec")), so that the decoder's gather overlaps the encoder forward."""
        self.arena.refresh_shadow()
        if wait_shadow and self.shadow_events:
            for ev in self.shadow_events.values():
                torch.cuda.current_stream().wait_event(ev)
            self.shadow_events = None

    def wait_shadow(self, group: str) -> None:
        """Current stream waits until the shadow weights of ``group`` ("vit", "enc" = encoder + tied embedding, "dec") are whole
        (sharded data-parallel optimizer: their all-gather runs on the communication stream behind the previous step's Adam)."""
        ev = self.shadow_events.get(group) if self.shadow_events else None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def forward(self, video, input_tokenized, output_tokenized):
        m = self.model
        self.prepare()
        need_grad = torch.is_grad_enabled()
        vis = None
        video_dict = None
        if m.use_video:
            if isinstance(video, dict):
                vis, atts = video["video"], video["atts_vis"]
                video_dict = video
                if vis.dtype != torch.bfloat16:      # the cached visual tokens come back in the dtype they were handed out in (fp32, like the
                    vis = vis.to(torch.bfloat16)     # reference's vid2seq.py:61-66): exact, they were bf16 values
            else:
                vis = _VitFn.apply(self.anchor, self, video, need_grad)
                atts = torch.ones(vis.shape[:-1], dtype=torch.long, device=self.device)
                # the reference returns its visual tokens as fp32 (vid2seq.py:60-67); callers only hand them back for the second pass (dvc.py:95-101)
                video_dict = {"video": vis.float(), "atts_vis": atts}
        in_ids = input_tokenized["input_ids"] if m.use_speech else None
        in_mask = input_tokenized["attention_mask"] if m.use_speech else None
        loss = _T5LossFn.apply(self.anchor, self, vis, in_ids, in_mask, output_tokenized["input_ids"], output_tokenized["attention_mask"],
                               need_grad)
        return {"loss": loss}, video_dict

    def zero_grad(self) -> None:
        self.arena.grad.zero_()
        self.arena.attach_grads()

    def begin_grad_step(self) -> None:
        """Start of a native training step (Trainer): gradients of the weight MATRICES are overwritten by their first weight-gradient
        GEMM of the step (_wgrad), so only what is accumulated with atomics or from several producers is zeroed here -- pos_embed, the
        small fp32-consumed parameters and the tied embedding: the tail of the arena (0.1 GB instead of 1.16 GB).  Needs every matrix
        to receive a gradient in the step: models that skip a branch (use_video / use_speech off) zero everything."""
        a, m = self.arena, self.model
        if not (m.use_video and m.use_speech and self.skip_grad_memset):
            self._fresh_grads = None
            a.grad.zero_()
            return
        self._fresh_grads = set()
        a.grad[a.offsets["visual_encoder.pos_embed"]:].zero_()

    def end_grad_step(self) -> None:
        """End of a completed native step: every weight matrix must have received its (overwriting) first weight-gradient GEMM --
        a matrix that did not would keep the PREVIOUS step's gradient (its memset was skipped)."""
        fresh = self._fresh_grads
        if fresh is not None:
            a = self.arena
            end = a.offsets["visual_encoder.pos_embed"]
            def covered(n):          # fused projections: one weight-gradient GEMM writes q|k|v (cross-attention: k|v)
                if n in fresh:
                    return True
                if n.endswith("SelfAttention.k.weight") or n.endswith("SelfAttention.v.weight"):
                    return n[:-len("k.weight")] + "q.weight" in fresh
                return n.endswith("EncDecAttention.v.weight") and n[:-len("v.weight")] + "k.weight" in fresh
            missing = [n for n in a.names if a.offsets[n] < end and a.params[n].dim() >= 2 and not covered(n)]
            assert not missing, f"skip_grad_memset: no weight gradient was written for {missing[:4]} in this step"
        self._fresh_grads = None

    def abort_grad_step(self) -> None:
        """Leave overwrite mode whatever happened (Trainer wraps the step body in try/finally): after an exception inside a step the
        autograd-path backward must accumulate again instead of overwriting."""
        self._fresh_grads = None
        self._wgrad_groups = {}

    def _begin_backward(self) -> None:
        """If the caller dropped the gradients (optimizer.zero_grad(set_to_none=True)) start from a zeroed arena."""
        self._wgrad_groups = {}           # nothing collected by a backward that raised may leak into this one
        if self.arena.attach_grads():
            self.arena.grad.zero_()

    # ========================================================================================== decoding
    @torch.no_grad()
    def encode(self, video, input_tokenized):
        m = self.model
        self.prepare()
        B = video.shape[0] if m.use_video else input_tokenized["input_ids"].shape[0]
        parts, masks = [], []
        if m.use_video:
            T = video.shape[1]
            parts.append(self.vit_forward(video, None).view(B, T, self.d))
            masks.append(torch.ones(B, T, dtype=torch.uint8, device=self.device))
        if m.use_speech:
            ids = input_tokenized["input_ids"]
            im = input_tokenized["attention_mask"].to(torch.uint8).contiguous()
            Lx = ids.shape[1]
            plan = self._pack_plan(ids, input_tokenized["attention_mask"]) if self.pack else None
            masks.append(im)
            if plan is not None:                   # padding-free encoder, rows scattered back into the zero-filled memory
                T = parts[0].shape[1] if parts else 0
                enc = self.encoder_forward(ids.reshape(-1).index_select(0, plan[2]), None, 0.0, None, pack=plan)
                mem = torch.zeros(B, T + Lx, self.d, dtype=torch.bfloat16, device=self.device)
                if T:
                    mem[:, :T] = parts[0]
                tr = plan[2][:plan[5]]
                rows = tr + (torch.div(tr, Lx, rounding_mode="floor") * T + T) if T else tr
                mem.view(B * (T + Lx), self.d).index_copy_(0, rows, enc[:plan[5]])
                mask = torch.cat(masks, 1).contiguous() if len(masks) == 2 else masks[0]
                return mem, mask
            parts.append(self.encoder_forward(ids, im, 0.0, None).view(B, Lx, self.d))
        mem = torch.cat(parts, 1).contiguous() if len(parts) == 2 else parts[0].contiguous()
        mask = torch.cat(masks, 1).contiguous() if len(masks) == 2 else masks[0]
        return mem, mask

    def _cross_on_memory(self, mem: torch.Tensor, mem_mask: torch.Tensor, G: int):
        """Decode-step cross-attention WITHOUT per-layer K / V caches (csrc/v2s_memattn.hip): scores and weighted sums are taken on the
        encoder memory itself with the K projection folded into the query and the V projection applied to the result -- half the bytes
        of a step's largest stream, one tensor for all layers, and no cross K / V projection of the memory at all.  Needs d_model 768
        with 64-wide heads, at most 48 query rows per entry (G beams x H heads), and the usual memory mask (a prefix of valid rows
        per entry, none empty) and, in the default mode, a batch large enough to be bandwidth-bound; returns None otherwise and the
        K / V-cache path is used.  Either path gives a sequence the same result in any batch that takes that path; the two paths differ by
        bf16 rounding.  Returns ``attend(i, x, rows, ctx, folded, eps)``."""
        a, c = self.arena, self.cfg
        B, S, d = mem.shape
        H, inner = self.H, self.inner
        if not self.decode_mem_attn or (G > 1 and self.decode_mem_attn < 3) or d != 768 or inner != H * 64 or G * H > 48:
            return None
        # small batches: the K/V-cache kernels' shorter launch chain wins.  The choice is made from the call's SHAPE (B x S memory positions,
        # pad positions included: the threshold was calibrated on batches of ~90 % valid keys), never from the batch's content, so that which
        # path a sequence takes -- the two differ by bf16 rounding -- does not depend on its neighbours' lengths; and it is made BEFORE the mask
        # is read back, so that the early-out costs no host synchronisation
        if self.decode_mem_attn == 1 and B * S < self.decode_mem_attn_min_keys:
            return None
        m = mem_mask.to(torch.bool)
        ok = bool((m[:, 1:] <= m[:, :-1]).all().item()) if S > 1 else True
        klen = m.sum(1).to(torch.int32)
        klen_h = klen.tolist()
        if not ok or min(klen_h) < 1:
            return None
        assert mem.is_contiguous() and mem.dtype == torch.bfloat16
        plan = L.MemAttnPlan(klen_h, G * H, self.device)
        wkT, wv = [], []
        for i in range(c.n_dec):
            kvw = a.w(self._ca(i) + "k.weight", (2 * inner, d))
            wkT.append(kvw[:inner].t().contiguous())
            wv.append(kvw[inner:])
        qp = self._bf(B * G * H, d)
        nrm = [None]

        def attend(i, x, rows, ctx, wq_folded, eps):
            """ctx[rows, inner] = cross-attention of layer i for the residual rows x (RMSNorm inside: folded weights + eps, or applied here)"""
            if wq_folded is not None:
                L.decode_qfold(x, rows, wq_folded, wkT[i], eps, qp, H, d)
            else:
                if nrm[0] is None:
                    nrm[0] = (self._bf(rows, d), self._f32(rows))
                L.rmsnorm_fwd(x, a.f(self._ln("decoder", i, 1)), nrm[0][0], nrm[0][1], rows, d, eps)
                L.decode_qfold(nrm[0][0], rows, a.w(self._ca(i) + "q.weight"), wkT[i], 0.0, qp, H, d)
            L.decode_memattn(qp, mem, S * d, plan, d)
            L.decode_ctxfold(plan, rows, G, H, wv[i], ctx, d)
        return attend

    def _decode_weights(self):
        """Per generate() call: decoder projection weights with the preceding RMSNorm weight folded into their columns
        (W' = W * diag(w_ln)), so that the decode step runs norm + projection as ONE skinny GEMM with the row scale
        rsqrt(mean(x^2)+eps) computed in its prologue (v2s_gemm rms_eps).  Temporary bf16 copies (t5-base: 162 MB)."""
        a, c = self.arena, self.cfg
        d, inner = self.d, self.inner
        out = {"qkv": [], "cq": [], "wi": []}
        for i in range(c.n_dec):
            for key, wname, shape, ln in (("qkv", self._sa("decoder", i) + "q.weight", (3 * inner, d), 0),
                                          ("cq", self._ca(i) + "q.weight", (inner, d), 1),
                                          ("wi", self._ffp("decoder", i) + "wi.weight", (self.ff, d), 2)):
                t = self._bf(*shape)
                L.scale_cols(a.w(wname, shape), a.f(self._ln("decoder", i, ln)), t, shape[0], shape[1])
                out[key].append(t)
        E = a.w("t5_model.shared.weight")
        Ef = self._bf(E.shape[0], d)
        L.scale_cols(E, a.f("t5_model.decoder.final_layer_norm.weight"), Ef, E.shape[0], d)
        out["head"] = Ef
        return out

    @torch.no_grad()
    def greedy(self, video, input_tokenized, max_new_tokens: int = 256, stop_at_eos: bool = True, use_graph: bool = True,
               repetition_penalty: float = 1.0, sample=None, min_length: int = 1) -> torch.Tensor:
        """HF-4.28 greedy_search semantics on a static KV cache: the cross K/V of every layer are
        projected once; the self K/V grow in place (no torch.cat, no cache reorder).  One decode step is ~150 small
        launches, so it is captured ONCE into a hipGraph (through torch.cuda.CUDAGraph) and replayed: every
        step-dependent quantity (cache position, number of keys, bias row, output column) is read by the kernels from a
        device-resident step counter, so the same graph serves all steps.  The all-rows-finished test of HF is evaluated
        every 8 replays; the returned tensor is trimmed to exactly the length HF would have produced.
        ``sample=(top_p, temperature, seed[, top_k])`` switches the token choice from argmax to sampling (temperature, top-k, top-p
        warpers; same loop and stopping rule as HF's sample()); ``min_length`` bans EOS while the decoder sequence is shorter (HF MinLengthLogitsProcessor)."""
        a, c = self.arena, self.cfg
        mem, mem_mask = self.encode(video, input_tokenized)
        B, S, d = mem.shape
        inner, H, nl = self.inner, self.H, c.n_dec
        mem2 = mem.view(B * S, d)
        on_mem = self._cross_on_memory(mem, mem_mask, 1)
        self.last_cross_path = "memory" if on_mem is not None else "kv-cache"
        cross = []
        for i in range(nl if on_mem is None else 0):
            kv = self._bf(B * S, 2 * inner)
            L.gemm(mem2, a.w(self._ca(i) + "k.weight", (2 * inner, d)), kv, B * S, 2 * inner, d)
            cross.append(kv)
        maxlen = max_new_tokens
        cache = [self._bf(B, maxlen, 2 * inner) for _ in range(nl)]
        diag, _ = self._bias_diag("decoder", maxlen, maxlen)           # [H, 2*maxlen-1]; row t starts at column maxlen-1-t
        seq = torch.full((B, maxlen + 1), c.pad_id, dtype=torch.long, device=self.device)
        seq[:, 0] = c.dec_start_id
        unfinished = torch.ones(B, dtype=torch.int32, device=self.device)
        nxt = torch.full((B,), c.dec_start_id, dtype=torch.long, device=self.device)   # token fed to the next step
        pos = torch.zeros(1, dtype=torch.int32, device=self.device)                   # device-resident step counter
        logits = self._f32(B, self.ldv)
        E = a.w("t5_model.shared.weight")
        n = self._bf(B, d); rstd = self._f32(B)
        qkv = self._bf(B, 3 * inner); q = self._bf(B, inner); ctx = self._bf(B, inner); u = self._bf(B, self.ff)
        ha, hb = self._bf(B, d), self._bf(B, d)
        eos = c.eos_id if stop_at_eos else -1

        fw = self._decode_weights() if (d % 128 == 0 and B <= 512) else None     # fused norm + projection (v2s_gemm rms_eps: decode rows <= 512, K % 128 == 0)
        fused_head = fw is not None and B <= 64                                  # the wide LM-head kernel stages 64 rows
        eps = c.eps

        def proj(x, rows, key, i, wname, shape, ln_idx, out, **kw):
            """RMSNorm + Linear of a decode step: one fused skinny GEMM (folded weights) or norm kernel + GEMM."""
            if fw is not None:
                L.gemm(x, fw[key][i], out, rows, shape[0], d, rms_eps=eps, **kw, decode=True)
            else:
                L.rmsnorm_fwd(x, a.f(self._ln("decoder", i, ln_idx)), n, rstd, rows, d, eps)
                L.gemm(n, a.w(wname, shape), out, rows, shape[0], d, **kw, decode=True)

        # greedy tail as one launch (argmax + the next step's embedding row + the step counter): the step then neither opens with an embedding
        # launch nor closes with a counter launch; the first step's embedding is looked up once, before the loop
        fuse_tail = bool(self.decode_fuse_tail) and sample is None
        ticket = torch.zeros(1, dtype=torch.int32, device=self.device)
        if fuse_tail:
            L.embed_fwd(nxt, E, ha, B, d, self.V)

        def step():
            h, h2 = ha, hb
            if not fuse_tail:
                L.embed_fwd(nxt, E, h, B, d, self.V)
            for i in range(nl):
                sa, ca, fp = self._sa("decoder", i), self._ca(i), self._ffp("decoder", i)
                proj(h, B, "qkv", i, sa + "q.weight", (3 * inner, d), 0, qkv)
                L.decode_attn(B, H, maxlen, qkv, 3 * inner, cache[i], cache[i][:, :, inner:], maxlen * 2 * inner, 2 * inner,
                              ctx, inner, bias_row=diag, bias_ld=2 * maxlen - 1, pos_dev=pos, bias_maxlen=maxlen,
                              new_k=qkv[:, inner:], new_v=qkv[:, 2 * inner:], new_bs=3 * inner)       # cache append fused
                L.gemm(ctx, a.w(sa + "o.weight"), h2, B, d, inner, residual=h, decode=True)
                if on_mem is not None:
                    on_mem(i, h2, B, ctx, fw["cq"][i] if fw is not None else None, eps)
                else:
                    proj(h2, B, "cq", i, ca + "q.weight", (inner, d), 1, q)
                    L.decode_attn(B, H, S, q, inner, cross[i], cross[i][:, inner:], S * 2 * inner, 2 * inner, ctx, inner,
                                  key_mask=mem_mask, mask_ld=S)
                L.gemm(ctx, a.w(ca + "o.weight"), h, B, d, inner, residual=h2, decode=True)
                proj(h, B, "wi", i, fp + "wi.weight", (self.ff, d), 2, u, act=L.ACT_RELU)
                L.gemm(u, a.w(fp + "wo.weight"), h2, B, d, self.ff, residual=h, decode=True)
                h, h2 = h2, h
            if fused_head:
                L.gemm(h, fw["head"], logits, B, self.V, d, ldc=self.ldv, alpha=d ** -0.5, rms_eps=eps, decode=True)
            else:
                L.rmsnorm_fwd(h, a.f("t5_model.decoder.final_layer_norm.weight"), n, rstd, B, d, eps)
                L.gemm(n, E, logits, B, self.V, d, ldc=self.ldv, alpha=d ** -0.5, decode=True)
            if repetition_penalty != 1.0:           # HF RepetitionPenaltyLogitsProcessor on the raw logits (greedy_search)
                L.repetition_penalty(logits, self.ldv, B, self.V, seq, repetition_penalty, pos_dev=pos)
            if sample is None and min_length > 1 and eos >= 0:      # MinLengthLogitsProcessor of greedy_search (sampling: inside the sampling kernel)
                L.ban_token(logits, self.ldv, B, self.V, c.eos_id, pos, min_length)
            if sample is not None:                  # HF sample(): processors, then temperature / top-k / top-p warpers, multinomial
                L.topp_sample_step(logits, self.ldv, B, self.V, sample[0], sample[1], sample[2], nxt, unfinished, eos, c.pad_id,
                                   seq_out=seq, seq_ld=maxlen + 1, pos_dev=pos, min_length=min_length,
                                   top_k=sample[3] if len(sample) > 3 else 0)
            elif fuse_tail:
                L.argmax_step_tail(logits, self.ldv, B, self.V, nxt, unfinished, eos, c.pad_id, seq, maxlen + 1, pos, E, ha, d, self.V, ticket)
            else:
                L.argmax_step_seq(logits, self.ldv, B, self.V, nxt, unfinished, eos, c.pad_id, seq, maxlen + 1, pos)
            if not fuse_tail:
                L.counter_add(pos, 1)

        c0 = L.launch_count
        step()                                   # step 0 eagerly (also warms every code path before capture)
        self.last_decode_launches = L.launch_count - c0          # library launches of one decode step (bench.py reports it)
        done_steps = 1
        graph = None
        if use_graph and maxlen > 1:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
        while done_steps < maxlen:
            if stop_at_eos and (done_steps % 8 == 1) and int(unfinished.max().item()) == 0:
                break
            if graph is not None:
                graph.replay()
            else:
                step()
            done_steps += 1
        out = seq[:, :done_steps + 1]
        if stop_at_eos:
            # HF stops right after the step in which the last unfinished row emitted EOS: trim the pads decoded after that
            is_eos = out[:, 1:] == c.eos_id
            first = torch.where(is_eos.any(1), is_eos.float().argmax(1) + 1, torch.full((B,), done_steps, device=self.device))
            out = out[:, :int(first.max().item()) + 1]
        return out

    @torch.no_grad()
    def beam_search(self, video, input_tokenized, num_beams: int, max_new_tokens: int, length_penalty: float = 1.0,
                    use_graph: bool = True, min_length: int = 1, repetition_penalty: float = 1.0, num_return: int = 1,
                    sample=None, teacher=None) -> torch.Tensor:
        """HF-4.28 beam_search + BeamSearchScorer semantics (call site vid2seq.py:150-162) on static caches.
        ``sample=(top_p, temperature, seed, top_k)`` turns the step into HF's beam_sample (do_sample with num_beams > 1): the
        candidates of a row come from ``v2s_beam_sample_cand`` (warped scores + Gumbel keys) instead of ``v2s_topk_logprob``.
        The encoder memory is NOT replicated per beam: cross K/V are projected once per batch entry and the nb beams of an
        entry read the same rows (``kv_group``).  A step = decoder forward for B*nb rows -> ``v2s_topk_logprob`` (log-softmax +
        running beam score + per-beam top 2*nb) -> ``v2s_beam_advance`` (BeamSearchScorer.process on the device: next tokens, scores,
        source rows, finished hypotheses, the beam permutation), captured as a hipGraph and replayed back to back; with sampling or
        a teacher the host merges the candidates instead (beam.BeamScorer) and sends back next tokens, scores and source rows.  The self-attention cache is never moved: a [rows][maxlen] ``row_map`` says
        which cache row holds each (beam, position) key, and a beam reorder permutes the rows of that table (v2s_decode_attn
        row_map) -- HF copies every cached K/V row (modeling_t5.py:1771-1793), 12 layers x rows x len x 3 KB per step.
        ``teacher`` (parity tests at real shapes, tests/test_configs_gpu.py): a list of per-step decisions (tokens int64 [rows],
        scores float32 [rows], source rows int32 [rows], optionally the decoder ids so far [rows, maxlen + 1] for the repetition
        penalty) taken from a reference trajectory: the engine then FOLLOWS them instead of its own scorer -- same kernels, same graph,
        same row-map reorders -- and returns the per-step candidates [(values [rows, K], tokens [rows, K])] it would have scored."""
        from .beam import BeamScorer
        a, c = self.arena, self.cfg
        mem, mem_mask = self.encode(video, input_tokenized)
        B, S, d = mem.shape
        nb = num_beams
        R = B * nb
        if not 1 <= nb <= 128:
            raise ValueError(f"num_beams must be in [1, 128] on the HIP path (got {nb})")
        if sample is not None and nb > 16:
            raise ValueError(f"beam-sample keeps at most 64 warped candidates per beam row: num_beams <= 16 with sampling (got {nb})")
        # per-beam candidate lists are sorted: a longer list only adds entries the merge never reaches, so any num_beams <= 16 uses the next
        # instantiated size; wider beams (the reference forwards any num_beams, vid2seq.py:150-162) take the generic K = 2*nb rounds kernel
        K = next((k for k in (2, 4, 8, 16, 32) if k >= 2 * nb), 2 * nb)
        inner, H, nl = self.inner, self.H, c.n_dec
        mem2 = mem.view(B * S, d)
        on_mem = self._cross_on_memory(mem, mem_mask, nb)
        cross = []
        for i in range(nl if on_mem is None else 0):
            kv = self._bf(B * S, 2 * inner)
            L.gemm(mem2, a.w(self._ca(i) + "k.weight", (2 * inner, d)), kv, B * S, 2 * inner, d)
            cross.append(kv)
        maxlen = max_new_tokens
        if R > 65535 or maxlen > 4096:
            raise ValueError(f"beam search: at most 65535 beam rows and 4096 new tokens (got {R}, {maxlen})")
        cache = [self._bf(R, maxlen, 2 * inner) for _ in range(nl)]
        row_map = torch.zeros(R, maxlen, dtype=torch.int32, device=self.device)
        diag, _ = self._bias_diag("decoder", maxlen, maxlen)
        # host <-> device traffic of a step is one copy each way through pinned buffers: [tokens int64 | scores f32 | source rows i32]
        # down, [candidate values f32 | candidate tokens i32] up
        h2d = torch.zeros(4 * R, dtype=torch.int32, device=self.device)
        pins = self._ws.setdefault("beam_pinned", {})          # pinned staging is allocated once per (rows, K): hipHostMalloc costs ms
        NP = 3 if sample is not None else 2                    # candidate planes: values, tokens (, sampling keys)
        if (R, K, NP) not in pins:
            pins[(R, K, NP)] = (torch.zeros(4 * R, dtype=torch.int32).pin_memory(), torch.zeros(NP, R, K, dtype=torch.int32).pin_memory())
        h2d_host, cand_host = pins[(R, K, NP)]
        nxt = h2d[:2 * R].view(torch.long)
        nxt.fill_(c.dec_start_id)
        pos = torch.zeros(1, dtype=torch.int32, device=self.device)
        bscore = h2d[2 * R:3 * R].view(torch.float32)
        src_dev = h2d[3 * R:]
        cand = torch.zeros(NP, R, K, dtype=torch.int32, device=self.device)
        cand_val, cand_tok = cand[0].view(torch.float32), cand[1]
        cand_key = cand[2].view(torch.float32) if sample is not None else None
        h_tok, h_score, h_src = (h2d_host[:2 * R].view(torch.long).numpy(), h2d_host[2 * R:3 * R].view(torch.float32).numpy(),
                                 h2d_host[3 * R:].numpy())
        logits = self._f32(R, self.ldv)
        E = a.w("t5_model.shared.weight")
        n = self._bf(R, d); rstd = self._f32(R)
        qkv = self._bf(R, 3 * inner); q = self._bf(R, inner); ctx = self._bf(R, inner); u = self._bf(R, self.ff)
        ha, hb = self._bf(R, d), self._bf(R, d)
        cbs = maxlen * 2 * inner
        rp = repetition_penalty != 1.0
        dev_scorer = self.beam_on_device and sample is None and teacher is None and nb <= 16 and K <= 32
        hist = torch.zeros(R, maxlen + 1, dtype=torch.long, device=self.device) if (rp or dev_scorer) else None     # decoder ids so far, per beam
        row_lse = self._f32(R) if rp else None
        bstate = L.BeamState(B, nb, maxlen + 1, self.device, length_penalty) if dev_scorer else None

        fw = self._decode_weights() if (d % 128 == 0 and R <= 512) else None     # fused norm + projection (v2s_gemm rms_eps)
        fused_head = fw is not None and R <= 64
        eps = c.eps

        def proj(x, key, i, wname, shape, ln_idx, out, **kw):
            if fw is not None:
                L.gemm(x, fw[key][i], out, R, shape[0], d, rms_eps=eps, **kw, decode=True)
            else:
                L.rmsnorm_fwd(x, a.f(self._ln("decoder", i, ln_idx)), n, rstd, R, d, eps)
                L.gemm(n, a.w(wname, shape), out, R, shape[0], d, **kw, decode=True)

        def step():
            h, h2 = ha, hb
            L.embed_fwd(nxt, E, h, R, d, self.V)
            for i in range(nl):
                sa, ca, fp = self._sa("decoder", i), self._ca(i), self._ffp("decoder", i)
                proj(h, "qkv", i, sa + "q.weight", (3 * inner, d), 0, qkv)
                L.decode_attn(R, H, maxlen, qkv, 3 * inner, cache[i], cache[i][:, :, inner:], cbs, 2 * inner,
                              ctx, inner, bias_row=diag, bias_ld=2 * maxlen - 1, pos_dev=pos, bias_maxlen=maxlen,
                              new_k=qkv[:, inner:], new_v=qkv[:, 2 * inner:], new_bs=3 * inner,         # cache append fused
                              row_map=row_map, row_map_ld=maxlen)
                L.gemm(ctx, a.w(sa + "o.weight"), h2, R, d, inner, residual=h, decode=True)
                if on_mem is not None:
                    on_mem(i, h2, R, ctx, fw["cq"][i] if fw is not None else None, eps)
                else:
                    proj(h2, "cq", i, ca + "q.weight", (inner, d), 1, q)
                    L.decode_attn(R, H, S, q, inner, cross[i], cross[i][:, inner:], S * 2 * inner, 2 * inner, ctx, inner,
                                  key_mask=mem_mask, mask_ld=S, kv_group=nb)
                L.gemm(ctx, a.w(ca + "o.weight"), h, R, d, inner, residual=h2, decode=True)
                proj(h, "wi", i, fp + "wi.weight", (self.ff, d), 2, u, act=L.ACT_RELU)
                L.gemm(u, a.w(fp + "wo.weight"), h2, R, d, self.ff, residual=h, decode=True)
                h, h2 = h2, h
            if fused_head:
                L.gemm(h, fw["head"], logits, R, self.V, d, ldc=self.ldv, alpha=d ** -0.5, rms_eps=eps, decode=True)
            else:
                L.rmsnorm_fwd(h, a.f("t5_model.decoder.final_layer_norm.weight"), n, rstd, R, d, eps)
                L.gemm(n, E, logits, R, self.V, d, ldc=self.ldv, alpha=d ** -0.5, decode=True)
            if rp:          # processor on the log-probabilities (beam_search): rewrites the logits against the stored row lse
                L.repetition_penalty(logits, self.ldv, R, self.V, hist, repetition_penalty, pos_dev=pos, row_lse=row_lse)
            if sample is not None:
                L.beam_sample_cand(logits, self.ldv, R, self.V, K, bscore, sample[0], sample[1], sample[3], sample[2], cand_val, cand_tok,
                                   cand_key, ban_token=c.eos_id, pos_dev=pos, min_length=min_length, row_lse=row_lse if rp else None)
            else:
                L.topk_logprob(logits, self.ldv, R, self.V, K, bscore, cand_val, cand_tok, ban_token=c.eos_id, pos_dev=pos,
                               min_length=min_length, row_lse=row_lse if rp else None)
            if dev_scorer:      # BeamSearchScorer.process on the device: next tokens / scores / source rows, finished hypotheses, the beam
                L.beam_advance(cand_val, cand_tok, K, bstate, c.eos_id, c.pad_id, pos, hist, row_map, nxt, bscore, src_dev)      # permutation of hist and row_map
            L.counter_add(pos, 1)

        scorer = BeamScorer(B, nb, length_penalty, c.eos_id, c.pad_id, c.dec_start_id, max_new_tokens + 1, sample=sample is not None)
        if hist is not None:
            hist.copy_(torch.from_numpy(scorer.seqs))
        bscore.copy_(torch.from_numpy(scorer.scores.reshape(-1)))
        identity = np.arange(R, dtype=np.int32)
        graph = None
        recorded = []
        for t in range(maxlen if teacher is None else min(maxlen, len(teacher) + 1)):
            if t == 0 or not use_graph:
                c0 = L.launch_count
                step()
                self.last_decode_launches = L.launch_count - c0
            else:
                if graph is None:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    pos_keep = pos.clone()
                    with torch.cuda.graph(graph):
                        step()
                    pos.copy_(pos_keep)            # capture does not execute, but keep the counter explicit
                graph.replay()
            if dev_scorer:          # the graph replays back to back; the all-entries-done test of HF is evaluated every 8 steps
                if t % 8 == 7 and int(bstate.ndone.item()) == B:
                    break
                continue
            cand_host.copy_(cand, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            if teacher is not None:
                recorded.append((cand_host[0].view(torch.float32).numpy().copy(), cand_host[1].numpy().copy()))
                if t >= len(teacher):
                    break
                tt = teacher[t]
                h_tok[:] = tt[0]; h_score[:] = tt[1]; h_src[:] = tt[2]
                h2d.copy_(h2d_host, non_blocking=True)
                if rp:
                    hist.copy_(torch.as_tensor(tt[3]))
                if not np.array_equal(np.asarray(tt[2], dtype=np.int32), identity):
                    row_map.copy_(row_map.index_select(0, src_dev))
                continue
            tok, src, finished = scorer.advance(cand_host[0].view(torch.float32).numpy(), cand_host[1].numpy(),
                                                cand_host[2].view(torch.float32).numpy() if sample is not None else None)
            if finished:
                break
            h_tok[:] = tok; h_score[:] = scorer.scores.reshape(-1); h_src[:] = src
            h2d.copy_(h2d_host, non_blocking=True)
            if rp:
                hist.copy_(torch.from_numpy(scorer.seqs))
            if not np.array_equal(src, identity):
                row_map.copy_(row_map.index_select(0, src_dev))
        if teacher is not None:
            return recorded
        if not 1 <= num_return <= nb:
            raise ValueError(f"num_captions must be in [1, num_beams] (got {num_return})")
        if dev_scorer:              # hand the device state to the host scorer's finalize (BeamSearchScorer.finalize)
            scorer.seqs = hist.cpu().numpy()
            scorer.scores = bscore.cpu().numpy().reshape(B, nb).copy()
            scorer.cur_len = int(pos.item()) + 1
            scorer.done = bstate.done.cpu().numpy().astype(bool)
            hn, ht, hl = bstate.heap_n.cpu().numpy(), bstate.hyp_tok.cpu().numpy(), bstate.hyp_len.cpu().numpy()
            hs, ho, hw = bstate.hyp_score.cpu().numpy(), bstate.hyp_order.cpu().numpy(), bstate.heap_worst.cpu().numpy()
            for b in range(B):      # in insertion order: ties between equal scores resolve like on the host
                slots = sorted(range(int(hn[b])), key=lambda i: int(ho[b, i]))
                scorer.heaps[b].items = [(float(hs[b, i]), ht[b, i, :int(hl[b, i])].astype(np.int64)) for i in slots]
                scorer.heaps[b].worst = float(hw[b])
        return torch.from_numpy(scorer.finalize(num_return)).to(self.device)


# ==============================================================================================================
# autograd hooks (coarse: one node for the ViT, one for T5 encoder+decoder+loss)
# ==============================================================================================================
class _VitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, eng: Engine, video: torch.Tensor, need_grad: bool):
        tape = {} if need_grad else None
        vis = eng.vit_forward(video, tape)
        ctx.eng, ctx.tape = eng, tape
        return vis.view(video.shape[0], video.shape[1], eng.d)

    @staticmethod
    def backward(ctx, dvis):
        eng = ctx.eng
        eng._begin_backward()
        eng.vit_backward(ctx.tape, dvis.to(torch.bfloat16))
        eng.join_wgrads()
        ctx.tape = None
        return None, None, None, None


class _T5LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, eng: Engine, vis, in_ids, in_mask, out_ids, out_mask, need_grad: bool):
        tape = {} if need_grad else None
        loss = eng.t5_loss_forward(vis, in_ids, in_mask, out_ids, out_mask, tape)
        ctx.eng, ctx.tape = eng, tape
        ctx.has_vis = vis is not None
        return loss

    @staticmethod
    def backward(ctx, gloss):
        eng = ctx.eng
        eng._begin_backward()
        dvis = eng.t5_loss_backward(ctx.tape, gloss)
        eng.join_wgrads()
        ctx.tape = None
        return None, None, (dvis if ctx.has_vis else None), None, None, None, None, None
