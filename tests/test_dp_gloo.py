"""world_size-2 gloo test (CPU) of the data-parallel bookkeeping: slice-wise all-reduce of a flat gradient arena,
and the DP parity definition of SURVEY 8e (N ranks x batch b == one process on the concatenated batch when every rank
has the same number of target tokens), checked with the CPU oracle as the per-rank "engine"."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vid2seq_ref as R
from vidchapters_amd import synth


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    cfg = R.RefConfig.small()
    P = synth.init_params(R.param_shapes(cfg), 13, cfg.d_model, cfg.inner, cfg.d_ff)
    for v in P.values():
        v.requires_grad_(True)
    full = synth.make_batch(4, 10, 20, 12, cfg.vocab, 21, cfg.vit_dim)
    full["output_ids"] = torch.where(full["output_ids"] == 0, torch.full_like(full["output_ids"], 5), full["output_ids"])  # equal #targets per rank
    sl = slice(rank * 2, rank * 2 + 2)
    out, _ = R.vid2seq_forward(P, cfg, full["video"][sl], full["input_ids"][sl], full["input_ids"][sl] != 0,
                               full["output_ids"][sl], full["output_ids"][sl] != 0)
    names = list(P)
    grads = torch.autograd.grad(out["loss"], [P[k] for k in names])
    # flat arena + slice-wise SUM all-reduce, scaled by 1/world (what GradSync + the Adam kernel's grad_scale do)
    flat = torch.cat([g.reshape(-1) for g in grads])
    chunk = 50000
    for o in range(0, flat.numel(), chunk):
        dist.all_reduce(flat[o:o + chunk], op=dist.ReduceOp.SUM)
    flat /= world
    if rank == 0:
        out1, _ = R.vid2seq_forward(P, cfg, full["video"], full["input_ids"], full["input_ids"] != 0, full["output_ids"], full["output_ids"] != 0)
        g1 = torch.cat([g.reshape(-1) for g in torch.autograd.grad(out1["loss"], [P[k] for k in names])])
        ret["err"] = float((flat - g1).abs().max() / g1.abs().max())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_large_batch():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret["err"] < 1e-5, ret["err"]


def test_gradsync_ranges_cover_arena_once():
    """The arena order behind GradSync's ready-ranges (host logic only): decoder matrices | encoder matrices | ViT matrices + pos_embed |
    the small fp32-consumed parameters (replicated by a sharded optimizer) | the tied embedding."""
    from vidchapters_amd import SyntheticTokenizer, Vid2Seq
    from vidchapters_amd.engine import Engine
    cfg = R.RefConfig.small()
    m = Vid2Seq(dict(d_model=cfg.d_model, d_kv=cfg.d_kv, heads=cfg.heads, d_ff=cfg.d_ff, n_enc=cfg.n_enc, n_dec=cfg.n_dec),
                num_features=cfg.num_features, embed_dim=cfg.vit_dim, depth=cfg.vit_depth, heads=cfg.vit_heads, mlp_dim=cfg.vit_mlp,
                tokenizer=SyntheticTokenizer(512, 100), init_seed=1)
    named = dict(m.named_parameters())
    order = [n for n, _ in Engine._arena_order(type("E", (), {"model": m, "cfg": m.cfg, "_sa": staticmethod(Engine._sa), "_ca": staticmethod(Engine._ca),
                                                               "_ln": staticmethod(Engine._ln), "_ffp": staticmethod(Engine._ffp)})())]
    small = [n for n in order if Engine.is_small_param(n, named[n])]
    first_small = order.index(small[0])
    assert order[first_small:first_small + len(small)] == small and order[-1] == "t5_model.shared.weight" and first_small + len(small) == len(order) - 1
    big = order[:first_small]
    assert all(named[n].dim() >= 2 for n in big) and big[-1] == "visual_encoder.pos_embed"
    first_enc = next(i for i, n in enumerate(big) if n.startswith("t5_model.encoder."))
    first_vis = next(i for i, n in enumerate(big) if not n.startswith("t5_model."))
    assert all(n.startswith("t5_model.decoder.") for n in big[:first_enc])
    assert all(n.startswith("t5_model.encoder.") for n in big[first_enc:first_vis])
    assert all(n.endswith(("layer_norm.weight", ".bias", "norm.weight", "norm1.weight", "norm2.weight", "relative_attention_bias.weight")) for n in small)
    assert len(order) == len(set(order)) == len(named)
    # the per-4-layers hand-off of Trainer ends a slice at the last matrix of an encoder block
    i_o = big.index(Engine._sa("encoder", 0) + "o.weight")
    assert big[i_o + 1] == "proj_v2t.weight" or big[i_o + 1].startswith("visual_encoder.") or i_o + 1 == len(big)


def _shard_worker(rank, world, port, ret):
    """Sharded optimizer arithmetic on the CPU: bucket_plan + reduce-scatter + Adam on the owned stripes + all-gather must equal
    all-reduce + Adam on everything (the GPU path runs the same plan through GradSync)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vidchapters_amd.train import bucket_plan
    torch.manual_seed(3)
    n = 64 * 41 + 64 * 7                        # two ranges, neither a multiple of the bucket size
    master = torch.randn(n)
    g = torch.randn(n, generator=torch.Generator().manual_seed(100 + rank))
    ranges = [(0, 64 * 41, False), (64 * 41, n, True)]          # the second one is replicated (time-token rows / small parameters)
    chunk = 64 * world * 5

    def adam(p, m, v, grad, coef):
        gg = grad * coef
        m.mul_(0.9).add_(gg, alpha=0.1); v.mul_(0.999).addcmul_(gg, gg, value=0.001)
        p.addcdiv_(m / (1 - 0.9), (v / (1 - 0.999)).sqrt() + 1e-8, value=-1e-2)

    # reference: all-reduce everything, clip coefficient from the global norm, Adam everywhere
    gr = g.clone(); dist.all_reduce(gr)
    coef_ref = min(1.0, 1.0 / (float(gr.norm()) / world + 1e-6)) / world
    p_ref, m_ref, v_ref = master.clone(), torch.zeros(n), torch.zeros(n)
    adam(p_ref, m_ref, v_ref, gr, coef_ref)
    # sharded
    gs = g.clone(); p, m, v = master.clone(), torch.zeros(n), torch.zeros(n)
    shadow = master.clone().bfloat16()
    owned, repl, buckets = [], [], []
    for s0, e0, rep in ranges:
        for kind, o, k in bucket_plan(s0, e0, chunk, world, not rep):
            if kind == "rs":
                assert k % (64 * world) == 0
                part = k // world
                out = torch.empty(part)
                dist.reduce_scatter_tensor(out, gs[o:o + k].clone())
                gs[o + rank * part:o + (rank + 1) * part] = out
                owned.append((o + rank * part, o + (rank + 1) * part)); buckets.append((o, k))
            else:
                dist.all_reduce(gs[o:o + k]); repl.append((o, o + k))
    sq = torch.zeros(1)
    for a0, b0 in owned:
        sq += (gs[a0:b0] ** 2).sum()
    dist.all_reduce(sq)
    for a0, b0 in repl:
        sq += (gs[a0:b0] ** 2).sum()
    coef = min(1.0, 1.0 / (float(sq.sqrt()) / world + 1e-6)) / world
    for a0, b0 in owned + repl:
        adam(p[a0:b0], m[a0:b0], v[a0:b0], gs[a0:b0], coef)
        shadow[a0:b0] = p[a0:b0].bfloat16()
    for o, k in buckets:
        part = k // world
        full = torch.empty(2 * k, dtype=torch.uint8)                 # a gather moves bits: bytes are a type gloo knows
        dist.all_gather_into_tensor(full, shadow[o + rank * part:o + (rank + 1) * part].clone().view(torch.uint8))
        shadow[o:o + k] = full.view(torch.bfloat16)
    covered = torch.zeros(n, dtype=torch.int32)
    for a0, b0 in owned + repl:
        covered[a0:b0] += 1
    allcov = covered.clone(); dist.all_reduce(allcov)
    ret[f"cover{rank}"] = bool(((allcov == 1) | (allcov == world)).all()) and bool((covered <= 1).all())
    ret[f"coef{rank}"] = abs(coef - coef_ref) / coef_ref
    ret[f"shadow{rank}"] = bool(torch.equal(shadow, p_ref.bfloat16()))      # every rank ends with the full updated bf16 weights
    mine = torch.zeros(n, dtype=torch.bool)
    for a0, b0 in owned + repl:
        mine[a0:b0] = True
    ret[f"master{rank}"] = float((p[mine] - p_ref[mine]).abs().max())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_optimizer_equals_allreduce_two_ranks():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_shard_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[f"cover{r}"] and ret[f"coef{r}"] < 1e-5 and ret[f"shadow{r}"] and ret[f"master{r}"] < 1e-6, dict(ret)


def test_padding_free_plans_host_logic():
    """Row bookkeeping of the padding-free decoder and memory (host logic only, device tensors on the CPU): offsets, kept pad rows,
    row counts a multiple of 64, disjoint cover of the packed memory."""
    import numpy as np
    from vidchapters_amd.engine import Engine
    stub = type("E", (), {"_ws": {}, "device": torch.device("cpu")})()
    B, Lo = 5, 40
    lens = [40, 9, 15, 31, 1]
    mask = (torch.arange(Lo)[None, :] < torch.tensor(lens)[:, None])
    off, rows, tok_rows, b_, lo_ = Engine._pack_plan_dec(stub, mask)
    off = off.numpy()
    assert (b_, lo_) == (B, Lo) and rows % 64 == 0 and rows == off[-1] == tok_rows.numel() and rows < B * Lo
    ext = np.diff(off)
    assert all(lens[i] <= ext[i] <= Lo for i in range(B)) and ext.sum() - sum(lens) == (-sum(lens)) % 64
    want = np.concatenate([np.arange(ext[i]) + i * Lo for i in range(B)])
    assert np.array_equal(tok_rows.numpy(), want)                       # every sequence is a prefix of its padded row range
    assert Engine._pack_plan_dec(stub, mask, lens=lens)[1] == rows      # host lengths: same plan, no read-back
    holes = mask.clone(); holes[1, 3] = False
    assert Engine._pack_plan_dec(stub, holes) is None                   # not a prefix mask -> dense path
    full = torch.ones(B, Lo, dtype=torch.bool)
    assert Engine._pack_plan_dec(stub, full) is None                    # nothing to drop
    tight = (torch.arange(Lo)[None, :] < torch.tensor([40, 40, 40, 40, 33])[:, None])
    assert Engine._pack_plan_dec(stub, tight) is None                   # 193 rows + 63 filler rows is more than the 200 dense rows
    T, tl = 10, (7, 30, 1)
    kv_off, mrows, vis_pos, txt_pos, real = Engine._mem_plan(stub, tl, T)
    kv_off = kv_off.numpy()
    assert real == sum(T + n for n in tl) == kv_off[-1] and mrows % 64 == 0 and 0 <= mrows - real < 64
    cover = np.concatenate([vis_pos.numpy(), txt_pos.numpy()])
    assert np.array_equal(np.sort(cover), np.arange(real))              # each real row is written exactly once
    for b in range(len(tl)):
        assert np.array_equal(vis_pos.numpy()[b * T:(b + 1) * T], kv_off[b] + np.arange(T))      # [video ; text] per sample
