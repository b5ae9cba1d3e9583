"""Kernels must return the same bits whatever else is resident on the chip.  Under data parallelism RCCL's kernels run beside the backward
by design, and the weight-gradient side stream (FK_WGRAD_STREAM=1) puts a GEMM beside the attention backward.  Round 2 found the dQ / dK
of fk_attn_bwd differing run to run in that situation; round 3 traced it to compiler-packed fp32 (v_pk_mul_f32 -> v_pk_fma_f32 with
op_sel half-swaps) in every RoPE rotation — wrong low-half results in lanes 48-63 while waves of the small bf16 weight-gradient GEMM
share the SIMD — and removed the packing (-fno-slp-vectorize, frankenstein_amd/build.py).  One pass of each former victim
beside that occupant, compared bit for bit with the quiet run (tools/coresidency_sweep.py is the wide version)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rope_kernels_keep_their_bits_beside_a_concurrent_gemm():
    from frankenstein_amd import kernels as K
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
    B, H, N, D = 3, 5, 4864, 64
    d = H * D
    M = B * N
    x, w_qkv = rnd(M, d), rnd(3 * d, d)
    table = torch.randn(N, D // 2, 2, device=dev, generator=g)
    qkv = rnd(M, 3 * d)
    q3 = qkv.view(B, N, 3 * d)
    q, k, v = (q3[..., i * d:(i + 1) * d].unflatten(-1, (H, D)) for i in range(3))
    mask = K.Mask(K.MASK_BLOCK_CAUSAL, 256)
    o, lse = K.attn_fwd(q, k, v, mask)
    do = rnd(B, N, H, D)
    ga, gb = rnd(M, 320), rnd(M, 840)

    def bwd(prescaled):
        dqkv = torch.empty_like(q3)
        dq, dk, dv = (dqkv[..., i * d:(i + 1) * d].unflatten(-1, (H, D)) for i in range(3))
        K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, mask, rope_table=table, rope_off=0, q_prescaled=prescaled)
        return dqkv

    def rope_inplace():
        t = q3.clone()
        K.rope_(t, 2 * H, D, table, 0)
        return t

    victims = {
        "fk_attn_bwd + inverse rope": lambda: bwd(False),
        "fk_attn_bwd + inverse rope, pre-scaled queries": lambda: bwd(True),
        "fk_gemm_nt_rope": lambda: K.gemm_nt_rope(x, w_qkv, None, table, N, 0, D, 2 * d),
        "fk_rope": rope_inplace,
    }
    side = torch.cuda.Stream()
    for name, fn in victims.items():
        ref = fn().clone()
        torch.cuda.synchronize()
        for rep in range(3):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(10):
                    K.gemm_tn(ga, gb)                       # the small (128 x 128, register-staged) bf16 weight-gradient kernel
            out = fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            assert torch.equal(out, ref), f"{name}: {int((out != ref).sum())} elements differ beside a concurrent GEMM (rep {rep})"


def test_few_query_attention_keeps_its_bits_beside_a_concurrent_gemm():
    """the kernels whose eight waves split the keys and merge through LDS (attn_fwd_fewq_kernel / attn_bwd_dq_fewq_kernel): their merge
    order is fixed, so the result must not depend on what else runs — three passes beside the same occupant, bit for bit."""
    from frankenstein_amd import kernels as K
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
    B, H, T, N, D = 8, 6, 32, 6144, 64
    q, kv, do = rnd(B, T, H, D), rnd(B, N, 2 * H * D), rnd(B, T, H, D)
    k, v = kv[..., : H * D].unflatten(-1, (H, D)), kv[..., H * D:].unflatten(-1, (H, D))
    ga, gb = rnd(B * N, 320), rnd(B * N, 840)

    def run():
        o, lse = K.attn_fwd(q, k, v)
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        K.attn_bwd(q, k, v, o, do, lse, dq, dkv[..., : H * D].unflatten(-1, (H, D)), dkv[..., H * D:].unflatten(-1, (H, D)))
        return o, lse, dq, dkv

    ref = [t.clone() for t in run()]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for rep in range(3):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(10):
                K.gemm_tn(ga, gb)
        out = run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for name, a, b in zip(("o", "lse", "dq", "dkv"), out, ref):
            assert torch.equal(a, b), f"{name}: {int((a != b).sum())} elements differ beside a concurrent GEMM (rep {rep})"


@pytest.mark.parametrize("M", [40_000 + 77, 40_000 + 64])
def test_fused_mlp_backward_keeps_its_bits_beside_a_concurrent_gemm(M):
    """fk_mlp_bwd_fused (one wave per SIMD, the whole LDS of a CU, hand-counted waits on its LDS-DMA streams): three passes beside the same
    occupant on a second stream, compared bit for bit with the quiet run.  40 077 rows take the hipcc kernel, 40 064 (whole 128-token tiles)
    the generated-stream kernel, whose fp32 arithmetic is hand-written scalar VALU between the MFMAs of a wave that is alone on its SIMD."""
    from frankenstein_amd import kernels as K
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(2)
    rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
    d, H = 384, 1536
    dy, h13, w2t, w13t = rnd(M, d), rnd(M, 2 * H), rnd(H, d), rnd(d, 2 * H)
    ga, gb = rnd(40_000, 320), rnd(40_000, 840)
    ref = [t.clone() for t in K.mlp_bwd_fused(dy, w2t, h13, w13t)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for rep in range(3):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(10):
                K.gemm_tn(ga, gb)
        out = K.mlp_bwd_fused(dy, w2t, h13, w13t)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for name, a, b in zip(("dh13", "dx"), out, ref):
            assert torch.equal(a, b), f"{name}: {int((a != b).sum())} elements differ beside a concurrent GEMM (rep {rep})"
