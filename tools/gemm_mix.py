"""Launch the step's GEMM shapes of ONE kernel variant in a fixed order (3 launches each) so that a rocprofv3 --pmc pass can
attribute HBM traffic per shape by dispatch order.  usage: gemm_mix.py  (shape list below = the dgrad/NT shapes that dispatch to
gemm_dma_kernel<false, true> in the cfg-2 step, with their per-step launch counts)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidchapters_amd import lib as L

# (M, N, K, launches per step) of the two dominant kernel variants in the cfg-2 step
# gemm_dma_kernel<false, true>: dx = dy @ W (transB) -- round 5: what is LEFT on it after the persistent gemm_a4p kernel took the plain dgrads with
# >= 256 tiles (encoder o / qkv / wi dgrads, the first cross K|V dgrad, decoder wo): the decoder's and ViT's short launches, the cross K|V
# accumulate chain (residual epilogue) and the two masked wo dgrads
SHAPES_DGRAD = [(35200, 768, 1536, 11, "res"), (8192, 768, 768, 36), (8192, 768, 3072, 12), (8192, 768, 2304, 12), (3200, 768, 2304, 12),
                (3200, 768, 2048, 12), (3200, 2048, 768, 12), (3200, 768, 768, 12),
                (32000, 3072, 768, 12, "dact"), (8192, 3072, 768, 12, "dact")]
# gemm_a4p_kernel<true, 0>: the plain dgrads the persistent kernel takes
SHAPES_A4P_DGRAD = [(32000, 768, 3072, 12), (32000, 768, 2304, 12), (32000, 768, 768, 12), (35200, 768, 1536, 1)]
# gemm_dma_kernel<false, false>: y = x @ W^T (forward): encoder qkv/o/wo, every decoder and ViT projection, cross K/V
SHAPES_NT = [(32000, 2304, 768, 12), (32000, 768, 768, 12), (32000, 768, 3072, 12), (35200, 1536, 768, 12), (8192, 2304, 768, 12),
             (8192, 768, 768, 36), (8192, 3072, 768, 12), (8192, 768, 3072, 12), (3200, 2304, 768, 12), (3200, 768, 768, 12),
             (3200, 2048, 768, 12), (3200, 768, 2048, 12)]
VARIANT = sys.argv[1] if len(sys.argv) > 1 else "nt"
SHAPES = SHAPES_NT if VARIANT == "nt" else (SHAPES_A4P_DGRAD if VARIANT == "a4p" else SHAPES_DGRAD)
if __name__ == "__main__":
    dev = "cuda"
    names = []
    for M, N, K, _, *ep in SHAPES:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        B = torch.randn((N, K) if VARIANT == "nt" else (K, N), device=dev).to(torch.bfloat16)
        C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        kw = {}
        if ep == ["dact"]:                       # d(hidden) masked by the saved dropout(relu(.)) activations, 1/(1-p) scale
            kw = dict(dact=L.ACT_RELU, z=torch.relu(torch.randn(M, N, device=dev)).to(torch.bfloat16), dropout_p=0.1, dropout_seed=3)
        if ep == ["res"]:                        # d(memory) accumulated in place over the decoder layers
            kw = dict(residual=C)
        for _ in range(3):
            L.gemm(A, B, C, M, N, K, transB=(VARIANT != "nt"), **kw)
        names.append(L.lib().v2s_last_gemm_kernel().decode())
    torch.cuda.synchronize()
    print(json.dumps(names))
