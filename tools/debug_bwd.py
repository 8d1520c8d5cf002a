"""Debug helper: run the small config and report the first backward sublayer producing non-finite values."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vid2seq_ref as R
from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
from vidchapters_amd import engine as E

cfg = R.RefConfig.small()
model = Vid2Seq(dict(d_model=cfg.d_model, d_kv=cfg.d_kv, heads=cfg.heads, d_ff=cfg.d_ff, n_enc=cfg.n_enc, n_dec=cfg.n_dec),
                num_features=cfg.num_features, embed_dim=cfg.vit_dim, depth=cfg.vit_depth, heads=cfg.vit_heads, mlp_dim=cfg.vit_mlp,
                tokenizer=SyntheticTokenizer(512, 100), vis_drop=0., enc_drop=0., dec_drop=0., init_seed=7).to("cuda").eval()
b = synth.make_batch(3, 10, 24, 12, cfg.vocab, 7, cfg.vit_dim)
tok = lambda ids: {"input_ids": ids.cuda(), "attention_mask": ids.cuda() != 0}

def fin(t):
    return t is None or bool(torch.isfinite(t.float()).all())

def wrap(name):
    orig = getattr(E.Engine, name)
    def f(self, *a, **k):
        out = orig(self, *a, **k)
        torch.cuda.synchronize()
        r = a[0]
        tag = f"{name} kind={r.get('kind','?') if isinstance(r, dict) else ''} i={r.get('i','') if isinstance(r, dict) else ''}"
        ins = [x for x in a[1:] if torch.is_tensor(x)]
        print(f"{tag}: in finite={[fin(x) for x in ins]} out finite={fin(out) if torch.is_tensor(out) else out}", end="")
        if isinstance(r, dict) and "ml" in r and r["ml"] is not None:
            ml = r["ml"]
            print(f" ml finite={fin(ml)} lmin={ml[..., 1].min().item():.3g} mmax={ml[...,0].abs().max().item():.3g}", end="")
        if name == "_cross_attn_bwd":
            print(f" dmem finite={fin(a[2])}", end="")
        print()
        return out
    setattr(E.Engine, name, f)

for n in ("_final_norm_bwd", "_ffn_bwd", "_cross_attn_bwd", "_self_attn_bwd"):
    wrap(n)
out, vd = model(b["video"].cuda(), tok(b["input_ids"]), tok(b["output_ids"]))
print("loss", out["loss"].item())
out["loss"].backward()
torch.cuda.synchronize()
bad = [k for k, p in model.named_parameters() if not torch.isfinite(p.grad).all()]
print("non-finite grads:", len(bad), bad[:6])
