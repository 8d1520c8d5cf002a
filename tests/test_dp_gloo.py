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
    """GradSync's four ready-ranges (decoder, encoder, shared, ViT) tile the arena exactly (host logic only)."""
    from vidchapters_amd import SyntheticTokenizer, Vid2Seq
    from vidchapters_amd.engine import Engine
    cfg = R.RefConfig.small()
    m = Vid2Seq(dict(d_model=cfg.d_model, d_kv=cfg.d_kv, heads=cfg.heads, d_ff=cfg.d_ff, n_enc=cfg.n_enc, n_dec=cfg.n_dec),
                num_features=cfg.num_features, embed_dim=cfg.vit_dim, depth=cfg.vit_depth, heads=cfg.vit_heads, mlp_dim=cfg.vit_mlp,
                tokenizer=SyntheticTokenizer(512, 100), init_seed=1)
    order = [n for n, _ in Engine._arena_order(type("E", (), {"model": m, "cfg": m.cfg, "_sa": staticmethod(Engine._sa), "_ca": staticmethod(Engine._ca),
                                                               "_ln": staticmethod(Engine._ln), "_ffp": staticmethod(Engine._ffp)})())]
    first_enc = next(i for i, n in enumerate(order) if n.startswith("t5_model.encoder."))
    first_vis = next(i for i, n in enumerate(order) if not n.startswith("t5_model."))
    assert all(n.startswith("t5_model.decoder.") for n in order[:first_enc])
    assert all(n.startswith("t5_model.encoder.") for n in order[first_enc:first_vis])
    assert order[-1] == "t5_model.shared.weight" and order[-2] == "visual_encoder.pos_embed"
    assert len(order) == len(set(order)) == len(dict(m.named_parameters()))
