"""btcdet_amd/streams.py: which streams' chains of launches do not overlap (one hardware queue, or two queues of one pipe), and drawing
streams that do"""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_same_queue_is_an_equivalence_and_distinct_streams_overlap():
    from btcdet_amd._lib import check, lib
    from btcdet_amd.streams import distinct_stream, same_queue
    torch.cuda.set_device(0)
    main = torch.cuda.current_stream()
    assert same_queue(main, main)
    pool = [main] + [torch.cuda.Stream() for _ in range(9)]
    rel = [[same_queue(a, b) for b in pool] for a in pool]
    n = len(pool)
    for i in range(n):
        for j in range(n):
            assert rel[i][j] == rel[j][i]
            for k in range(n):
                assert not (rel[i][j] and rel[j][k]) or rel[i][k], (i, j, k)        # transitive: classes = hardware queues
    classes = {tuple(r) for r in rel}
    assert 2 <= len(classes) <= 4        # (the command processor's four pipes)
    a, ok_a = distinct_stream([main])
    b, ok_b = distinct_stream([main, a], priority=-1)
    c, ok_c = distinct_stream([main, a, b])
    assert ok_a and ok_b and ok_c
    four = [main, a, b, c]
    for i in range(4):
        for j in range(i + 1, 4):
            assert not same_queue(four[i], four[j])
    # all four side by side: four chains of ten 40 us idle waves take about one chain's time, not four
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        for s in four:
            check(lib().btc_spin(40, s.cuda_stream), "btc_spin")
    torch.cuda.synchronize()
    assert (time.perf_counter() - t0) * 1e6 < 800
