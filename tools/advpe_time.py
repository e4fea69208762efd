import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from gfdl_atmos_cubed_sphere_amd.cubed_sphere import CubedSphere
from gfdl_atmos_cubed_sphere_amd.lib import Context
cs = CubedSphere(385); g = cs.gridstruct(0); npz = 127; bd = g.bd
ctx = Context(g, npz)
rng = np.random.default_rng(0)
ua = ctx.from_host(np.asfortranarray(rng.uniform(-20,20,bd.shape("A",npz)))); va = ctx.from_host(np.asfortranarray(rng.uniform(-20,20,bd.shape("A",npz))))
dp = ctx.from_host(np.asfortranarray(rng.uniform(500,1500,bd.shape("A",npz)))); om = ctx.zeros("A", npz)
for _ in range(3): ctx.adv_pe(300.0, ua, va, dp, om)
ctx.sync(); ctx.profile(True)
for _ in range(10): ctx.adv_pe(300.0, ua, va, dp, om)
ctx.sync(); print({k:(v[0], round(v[1]/10,4)) for k,v in ctx.profile_report().items()})
