"""Body of tests/test_gpu_cyclefold.py::test_cyclefold_verifiers_reject_tampering, run as a process of its own with VIMZ_HIP_LIBRARY=testing
(the hook that overwrites a prover's vectors, vimz_cf_poke, exists only in libvimz_hip_testing.so).  Test infrastructure."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tests import _cyclefold as cfo
    from tests import _oracle
    from tests._oracle import from_limbs
    from tests.test_circuits import step_inputs
    from vimz_amd import _lib, hip
    from vimz_amd.circuit import Circuit
    assert _lib.SO_PATH == _lib.TESTING_SO_PATH, "start this script with VIMZ_HIP_LIBRARY=testing"
    oracle = _oracle.load()
    ctx = hip.Context(0)
    ck1 = ctx.bases_generate(_lib.CURVE_BN254_G1, 1 << 19)
    ck2 = ctx.bases_generate(_lib.CURVE_GRUMPKIN, 1 << 13, b"ck-cyclefold")
    c = Circuit.for_resolution("contrast", "HD")
    z0, inputs = step_inputs("contrast")
    steps = np.stack(inputs)
    cf = hip.CycleFoldIVC(ctx, c, ck1, ck2, max_batch=4)
    try:
        cf.reset(z0)
        cf.fold(steps[:5])
        assert cf.verify(5, z0) == 0
        info = cf.info()
        rs = np.random.default_rng(3)
        # one wrong element in any of the five vectors: both verifiers reject
        for which, n, side, what in ((0, info["main_wires"], 0, hip.IX_RUNNING_Z), (1, info["main_wires"], 0, hip.IX_FRESH_Z), (2, info["cyclefold_wires"], 1, hip.IX_RUNNING_Z),
                                     (3, info["main_constraints"], 0, hip.IX_RUNNING_E), (4, info["cyclefold_constraints"], 1, hip.IX_RUNNING_E)):
            vec = from_limbs(cf.export(side, what))
            for idx in sorted(set([1, n - 1] + [int(x) for x in rs.integers(1, n, size=3)])):
                old = vec[idx]
                cf.poke(which, idx, (old + 1) % _lib.MODULUS[side])
                assert cf.verify(5, z0) != 0, (which, idx)
                if idx in (1, n - 1):
                    failed, _ = cfo.verify(oracle, cf, ck1, ck2, 5, z0, check_commitments=False)
                    assert failed, (which, idx)
                cf.poke(which, idx, old)
        assert cf.verify(5, z0) == 0
        # an unsatisfiable row is refused and leaves the proof where it was
        bad = steps[5:7].copy(); bad[1, 200, 0] ^= np.uint64(0xFF)      # a transformed pixel that is no longer the contrast of the original
        try:
            cf.fold(bad)
            raise AssertionError("an unsatisfiable row was folded")
        except _lib.VimzError as e:
            assert e.code == _lib.ERR_UNSAT
        assert cf.state()[1] == 5 and cf.verify(5, z0) == 0
        cf.fold(steps[5:])
        assert cf.verify(10, z0) == 0
    finally:
        cf.close()
        ck1.free(); ck2.free()
        ctx.close()
    print("tamper ok")


if __name__ == "__main__":
    main()
