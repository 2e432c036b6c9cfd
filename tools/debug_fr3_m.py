"""Scratch diagnostic (GPU box, -DJH_V3_DEBUG build): inertia / smooth force / accelerations of the cooperative fr3 kernel vs the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_fr3 import _controls
from judo_amd.rollout_backend import GpuRolloutBackend
np.set_printoptions(precision=5, suppress=True, linewidth=200)
N, H = 4, 40
om, task, knots, U = _controls(128, 40, seed=1)
U = np.repeat(U[33:34, :H], N, axis=0)  # rollout 33's first control for every rollout
x0 = task.default_state()
be = GpuRolloutBackend("fr3_pick", N)
gs, gsens, _ = be.rollout(x0, U)
flat = gs.reshape(-1)
rest = flat[H * 31:H * 31 + 192].reshape(12, 16)
f = om.forward(x0[:16], x0[16:], U[0, 0])
print('fs   gpu', rest[0, :15]); print('fs   ref', f.get('qfrc_smooth', None))
print('a0   gpu', rest[1, :15]); print('a0   ref', f.get('qacc_smooth', None))
print('a    gpu', rest[2, :15]); print('qacc ref', f['qacc'])
print('qacc gpu (implicit)', rest[3, :15])
rs, _ = om.rollout(x0, U[:1]); gs[0, 0] = gs[0, 0]
print('vel gpu', gs[0, 0, 16:]); print('vel ref', rs[0, 0, 16:])
for nm, row in zip(['eD', 'earef', 'ejar', 'jf', 'fD', 'fl', 'lims', 'iters'], rest[4:]): print(nm, row)
print('x0 fingers', x0[14:16], 'ctrl', U[0, 0])

its = flat[H * 31 + 192:H * 31 + 192 + 8 * 8 * 16].reshape(8, 8, 16)
for it in range(8):
    print(f'it {it}: gn {its[it, 0, 0]:.4e} snorm {its[it, 7, 0]:.4e} gp {its[it, 1, 0]:.4e} alpha {its[it, 2, 0]:.5f} act {its[it, 6, 0]:.0f} | g13 {its[it, 3, 13]:+.4e} g14 {its[it, 3, 14]:+.4e} p13 {its[it, 4, 13]:+.4f} p14 {its[it, 4, 14]:+.4f} a13 {its[it, 5, 13]:+.4f} a14 {its[it, 5, 14]:+.4f} | g6 {its[it, 3, 6]:+.3e} p6 {its[it, 4, 6]:+.4f}')
for it in (1, 2):
    print('it', it, 'g', its[it, 3]); print('     p', its[it, 4]); print('     a', its[it, 5])
for it in range(4): print('alpha by lane it', it, its[it, 2]); print('   gp', its[it, 1]); print('   gn', its[it, 0])

ls = flat[H * 31 + 192 + 1024:H * 31 + 192 + 1024 + 6 * 4 * 16].reshape(6, 4, 16)
np.set_printoptions(precision=8, suppress=False, linewidth=250)
for e in range(6): print('ls eval', e, 'alpha', ls[e, 2]); print('    d1', ls[e, 0]); print('    d2', ls[e, 1]); print('    d1raw', ls[e, 3])
