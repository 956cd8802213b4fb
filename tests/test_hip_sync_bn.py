"""--sync_bn (/root/reference/tools/train.py:32,130-131, torch.nn.SyncBatchNorm.convert_sync_batchnorm) on the hot path's BatchNorm1d
layers: two ranks sharing one GPU (gloo), btcdet_amd/spconv/fused_bn.py SyncBatchNormReLUFunction.
  * one layer: output rows, input gradient, parameter gradients and running statistics equal torch's BatchNorm1d + ReLU over the
    CONCATENATED batch (float64 on the CPU) -- with unequal row counts per rank;
  * the whole step: HotPathTrainer(sync_bn=True) marks every BatchNorm1d, runs in order, and the ranks' running statistics and
    parameters stay identical (they see the same global statistics and the same reduced gradients)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sync_bn_worker  # noqa: E402


def test_combine_stats_is_the_concatenated_batch_cpu():
    from btcdet_amd.spconv.fused_bn import combine_stats
    g = torch.Generator().manual_seed(1)
    parts = [torch.randn(n, 7, generator=g, dtype=torch.float64) * (i + 1) + i for i, n in enumerate((5, 0, 123, 40))]
    rows = []
    for p in parts:
        if p.shape[0]:
            v, m = torch.var_mean(p, dim=0, unbiased=False)
        else:
            v = m = torch.zeros(7, dtype=torch.float64)
        rows.append(torch.cat([m, v, torch.tensor([float(p.shape[0])], dtype=torch.float64)]))
    mean, var, n = combine_stats(torch.stack(rows), 7)
    v_all, m_all = torch.var_mean(torch.cat(parts), dim=0, unbiased=False)
    assert float(n) == 168 and torch.allclose(mean, m_all, rtol=0, atol=1e-12) and torch.allclose(var, v_all, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_sync_bn_layer_equals_the_concatenated_batch_two_ranks_one_gpu():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(sync_bn_worker.layer, args=(2, 29801, out), nprocs=2, join=True)
    for r in range(2):
        o = out[r]
        assert o["y"] < 5e-6 and o["dx"] < 5e-6, o            # fp32 kernels against float64 over the whole batch
        assert o["dw"] < 1e-5 and o["db"] < 1e-5, o
        assert o["rm"] < 1e-6 and o["rv"] < 1e-6 and o["nbt"] == 1, o


@pytest.mark.gpu
def test_trainer_sync_bn_two_ranks_one_gpu():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(sync_bn_worker.model, args=(2, 29803, out), nprocs=2, join=True)
    a, b = out[0], out[1]
    for o in (a, b):
        assert o["n_sync"] >= 25 and o["schedule"] == "split" and o["finite"], o
    # same global statistics + same reduced gradients on both ranks: buffers and parameters identical
    assert abs(a["buf_digest"] - b["buf_digest"]) <= 1e-9 * a["buf_abs"], (a, b)
    assert abs(a["params"] - b["params"]) <= 1e-9 * abs(a["params"]), (a, b)


@pytest.mark.gpu
def test_checkpoint_state_carries_rank0_buffers_two_ranks_one_gpu():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(sync_bn_worker.buffers, args=(2, 29805, out), nprocs=2, join=True)
    a, b = out[0], out[1]
    assert abs(a["before"] - b["before"]) > 1e-9 * a["before"]          # different scenes: rank-local running statistics differ ...
    assert a["after"] == a["before"] and b["after"] == a["after"]       # ... and the checkpoint of EITHER rank holds rank 0's
    assert a["it"] == 1 and "model_state" in a["keys"] and "optimizer_state_lst" in a["keys"]


@pytest.mark.gpu
def test_rank_without_rows_still_enters_a_torch_syncbatchnorm():
    """SparseSequential applies dense modules only to tensors with rows -- except the ones that hold a collective: a torch SyncBatchNorm
    (what convert_sync_batchnorm leaves for layers it cannot mark) in training mode is called with the 0-row batch too, or the ranks
    that do have rows would wait for this one forever.  One process: the module records that it was entered."""
    from btcdet_amd import spconv

    class Recording(torch.nn.SyncBatchNorm):
        calls = 0

        def forward(self, x):
            Recording.calls += 1
            return x

    dev = torch.device("cuda:0")
    seq = spconv.SparseSequential(Recording(8, affine=False)).to(dev).train()
    x = spconv.SparseConvTensor(torch.zeros((0, 8), device=dev), torch.zeros((0, 4), dtype=torch.int32, device=dev), [4, 8, 8], 1)
    seq(x)
    assert Recording.calls == 1
    seq.eval()(x)                       # eval mode holds no collective: the empty tensor skips the module as before
    assert Recording.calls == 1
