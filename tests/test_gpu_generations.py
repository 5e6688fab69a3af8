"""Provers created after others have folded on the same contexts (a service proving image after image): the later generations must fold
about as fast as the first.  They used not to — 20–250 steps/s instead of 800–930 — because the fold's high-priority streams waited for
the producer's low-priority streams in GPU-side barriers (tools/prover_generations.py is the long form of this test)."""
import time

import numpy as np
import pytest


@pytest.mark.gpu
def test_later_generations_of_provers_on_the_same_contexts_fold_as_fast_as_the_first():
    import bench
    from vimz_amd import folding, hip
    from vimz_amd.distributed import fold_segments_merged
    rows, z0 = bench.build_inputs("contrast", "HD")
    rows = np.stack(rows)[:96]
    ctxs = [hip.Context(0) for _ in range(3)]

    def generation(cyclefold):
        circuit, params = folding.prepare_folding(ctxs[0], "contrast", "HD", backend="sonobe" if cyclefold else "nova-snark")
        ck2 = params.secondary_key()
        cls = hip.CycleFoldIVC if cyclefold else hip.IVC
        provers = [cls(c, circuit, params.ck, ck2, max_batch=32) for c in ctxs]
        rates = []
        try:
            for _ in range(4):
                t0 = time.time()
                m = fold_segments_merged(provers, rows, z0, {}, **({"merged_cls": hip.CycleFoldMerged} if cyclefold else {}))
                rates.append(len(rows) / (time.time() - t0))
                assert m.verify(len(rows), z0) == 0
                m.close()
        finally:
            for v in provers:
                v.close()
            params.free()
        return sorted(rates[1:])[1]          # median of the folds after the first

    try:
        first = generation(False)
        second = generation(False)
        third = generation(True)
        fourth = generation(False)
    finally:
        for c in ctxs:
            c.close()
    # measured: 920 / 810–840 / 680–720 / 920 steps/s; before the cure the second and third generations ran at 20–250
    # (thresholds well below what is measured and well above the 0.03–0.25 of the old behaviour: a timing test must not be a flaky one)
    assert second > 0.45 * first and fourth > 0.45 * first, (first, second, third, fourth)
    assert third > 0.35 * first, (first, second, third, fourth)
