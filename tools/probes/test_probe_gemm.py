"""Tests of the rejected GEMM schedules (gemm256w4 / f8 / p.hip in this directory).  NOT part of the product suite: they need a probe
build of libafk.so (`make -C audio-flamingo_amd/csrc PROBES=1`) and are run explicitly:

    python -m pytest tools/probes/test_probe_gemm.py -q        (on a GPU box)
"""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_ops_gpu import BF, _cmp, _ops, _rand  # noqa: E402  (helpers of the product suite)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _need_probes(variant):
    """variants 3..13 are the rejected 256x256 schedules: they exist only in a `make PROBES=1` build of libafk.so"""
    from audio_flamingo_amd import _lib
    if variant > 2 and not _lib.has_probes():
        pytest.skip("rejected GEMM schedule: needs a -DAFK_PROBES build (make -C audio-flamingo_amd/csrc PROBES=1)")


@pytest.mark.parametrize("variant", [3, 6, 10, 13])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 768, 192), (300, 260, 320), (1000, 1280, 1280), (8, 512, 4096),
                                   (777, 1028, 64), (2048, 256, 2048)])
def test_gemm_probe_variants(dev, variant, M, N, K):
    """all NT kernels (1: 128x128; 2: 256x256 8-wave ping-pong; probe builds only - 3: 256x256 4-wave x 128x128; 6 / 10: 256x256 8-wave free-running;
    13: persistent tile loop) on every edge shape: K-tiles 1/2/3/many (prologue + tail waits), M/N tails, tiny M"""
    _need_probes(variant)
    ops = _ops()
    ops.gemm_set_variant(variant)
    try:
        a = _rand((M, K), dev, seed=11).to(BF)
        b = _rand((N, K), dev, seed=12).to(BF)
        bias = _rand((N,), dev, seed=13).to(BF)
        c = ops.gemm_nt(a, b, bias=bias)
        ref = a.float() @ b.float().t() + bias.float()
        _cmp(f"gemm v{variant} {M}x{N}x{K}", c, ref, atol=0.02 * math.sqrt(K), rtol=1e-2)
        # repeat to shake out races between the LDS-DMA ring and the fragment reads: results must be bit-identical
        for _ in range(3):
            assert torch.equal(ops.gemm_nt(a, b, bias=bias), c), "non-deterministic GEMM result (LDS race?)"
    finally:
        ops.gemm_set_variant(0)


@pytest.mark.parametrize("M,N,K", [(4096, 8192, 512), (5120, 7680, 64), (5000, 7700, 192), (8192, 9472, 1280)])
def test_gemm_persistent_tile_loop_matches_one_tile_per_workgroup(dev, M, N, K):
    """variant 13 (gemm256p.hip: one workgroup per CU walks 2-5 tiles, next prologue issued behind the previous tile's stores) must give
    the ping-pong kernel's result bit for bit - same per-tile arithmetic - with every epilogue the step uses, and stay so when repeated"""
    _need_probes(13)
    ops = _ops()
    a = _rand((M, K), dev, seed=31).to(BF)
    b = _rand((N, K), dev, seed=32).to(BF)
    bias = _rand((N,), dev, seed=33).to(BF)
    res = _rand((M, N), dev, seed=34).to(BF)
    outs = {}
    for variant in (2, 13):
        ops.gemm_set_variant(variant)
        try:
            pre = torch.empty((M, N), device=dev, dtype=BF)
            outs[variant] = (ops.gemm_nt(a, b), ops.gemm_nt(a, b, bias=bias, gelu=True, preact_out=pre), pre,
                             ops.gemm_nt(a, b, residual=res), ops.gemm_nt(a, b, out=res.clone(), accumulate=True))
            if variant == 13:
                for _ in range(3):
                    assert torch.equal(ops.gemm_nt(a, b), outs[13][0]), "persistent GEMM not deterministic (LDS hand-over between tiles?)"
        finally:
            ops.gemm_set_variant(0)
    for x, y in zip(outs[2], outs[13]):
        assert torch.equal(x, y)
    _cmp("persistent vs fp32", outs[13][0], a.float() @ b.float().t(), atol=0.02 * math.sqrt(K), rtol=1e-2)


