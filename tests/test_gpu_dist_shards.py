"""The multi-GPU decomposition with the REAL operator, on one GPU (run with `-m gpu`): `dist.shard_inputs` cuts q, k, v
for world = 2 and 8 ranks exactly as `sharded_attention` does on a node; here all ranks' shards are run one after the other
in one process through the HIP operator and their concatenation must equal the unsharded run BIT FOR BIT - batch split and
kv-head-group split, HND and NHD (the head split hands the operator non-contiguous views), fp16- and fp8-PV, with LSE.
(The process-group side - partitioning over real ranks and the gathers - is covered on CPU by tests/test_dist_gloo.py;
an RCCL run needs a multi-GPU node, which the builder does not have.)"""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from lowbit_quant_fa2_paddle_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("layout", ["HND", "NHD"])
@pytest.mark.parametrize("B,H,Hkv,expect", [(8, 4, 4, "batch"), (9, 2, 2, "batch"), (1, 16, 8, "head"), (3, 8, 8, "head")])
@pytest.mark.parametrize("pv", ["fp16", "fp8"])
def test_shards_concatenate_to_the_unsharded_result(dev, world, layout, B, H, Hkv, expect, pv):
    import lowbit_quant_fa2_paddle_amd as lb
    from lowbit_quant_fa2_paddle_amd import dist as lbd
    if expect == "batch" and B < world:
        pytest.skip("fewer batch elements than ranks: covered by the head split")
    S, D = 320, 64
    g = torch.Generator(device=dev)
    g.manual_seed(17)
    shq = (B, H, S, D) if layout == "HND" else (B, S, H, D)
    shk = (B, Hkv, S, D) if layout == "HND" else (B, S, Hkv, D)
    q = torch.randn(shq, generator=g, device=dev).half()
    k = (torch.randn(shk, generator=g, device=dev) + 0.3).half()
    v = torch.randn(shk, generator=g, device=dev).half()
    fn = lb.lowbit_fa_qk_int8_pv_fp16_triton if pv == "fp16" else lb.lowbit_fa_qk_int8_pv_fp8_cuda
    causal = (B + world) % 2 == 1
    o_full, lse_full = fn(q, k, v, tensor_layout=layout, is_causal=causal, return_lse=True)
    hdim = 1 if layout == "HND" else 2
    outs, lses = [], []
    for rank in range(world):
        qs, ks, vs, spec = lbd.shard_inputs(q, k, v, layout, world, rank)
        assert spec == (expect if B < world or expect == "batch" else spec)
        if 0 in qs.shape:
            continue  # uneven split: a rank may get nothing
        o, lse = fn(qs, ks, vs, tensor_layout=layout, is_causal=causal, return_lse=True)
        outs.append(o)
        lses.append(lse)
    spec = lbd.shard_spec(B, H, Hkv, world)
    o_cat = torch.cat(outs, dim=0 if spec == "batch" else hdim)
    lse_cat = torch.cat(lses, dim=0 if spec == "batch" else 1)
    assert torch.equal(o_cat, o_full), "concatenated shards differ from the unsharded run"
    assert torch.equal(lse_cat, lse_full)
