import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from vimz_amd import hip, _lib as L
from vimz_amd.circuit import Circuit
from tests.test_circuits import step_inputs
ctx = hip.Context(0)
t = sys.argv[1] if len(sys.argv) > 1 else "contrast"
z0, inputs = step_inputs(t)
steps = np.stack(inputs)
n = len(steps)
c = Circuit.for_resolution(t, "HD")
ck = ctx.bases_generate(L.CURVE_BN254_G1, 1 << 19)
ck2 = ctx.bases_generate(L.CURVE_GRUMPKIN, 8192, b"ck2")
t0 = time.time(); ivc = hip.IVC(ctx, c, ck, ck2, max_batch=16); print("create %.2fs" % (time.time() - t0), ivc.info(), flush=True)
ivc.reset(z0)
t0 = time.time(); ivc.fold(steps[:3]); print("first 3 ok", flush=True); ivc.fold(steps[3:]); dt = time.time() - t0
print("fold %d steps %.3fs -> %.1f steps/s" % (n, dt, n / dt))
print("verify", ivc.verify(), "state", ivc.state())
for k, v in ivc.profile().items(): print("  %-34s %8.3f ms/step (%d)" % (k, 1e3 * v[0] / max(v[1], 1), v[1]))
p = hip.Prover(ctx, c, ck, max_batch=16); p.reset(z0); p.fold(steps)
zi = p.instance()["z"]; print("accumulator state equal:", [sum(int(zi[i, k]) << (64 * k) for k in range(4)) for i in range(len(z0))] == ivc.state()[0])
# steady state: the same ten rows six times over (every row still satisfies its step relation; the chain just continues)
many = np.concatenate([steps] * 12)
ivc.close(); ivc = hip.IVC(ctx, c, ck, ck2, max_batch=64)
ivc.reset(z0); ivc.fold(many[:16]); ivc.reset(z0); t0 = time.time(); ivc.fold(many); dt = time.time() - t0
print("steady: %d steps %.1f steps/s verify=%d" % (len(many), len(many) / dt, ivc.verify()))
for k, v in ivc.profile().items(): print("  %-34s %8.3f ms/step (%d)" % (k, 1e3 * v[0] / max(v[1], 1), v[1]))
