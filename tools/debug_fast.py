import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import pedoni_amd as hip
from pedoni_amd import abi
from oracle import pyoracle as oracle
import helpers
sc = helpers.random_obstacle_scenario(200.0, 300)
field = helpers.oracle_field(oracle, sc)
pos, dest, v0, vel = helpers.inject_crowd(field, sc.field.size, 50000, 4, seed=21)
cpu = oracle.OracleModel(sc.field.size)
gpu = hip.HipModel(hip.Options(math_mode=abi.MATH_FAST), sc.field.size, field.distance_map, field.potential_maps, field.unit, sc.obstacle_array())
cpu.spawn_pedestrians(field, pos, dest, v0, vel); gpu.append(pos, dest, v0, vel); gpu.sort_despawn()
sp, sd, sv, s0 = cpu.download()
acc_c = cpu.calc_accelerations(field); acc_g = gpu.calc_accelerations(len(sp))
cpu.update_states(field); gpu.update_states()
gp, gd, gv, g0 = gpu.download(); wp, wd, wv, w0 = cpu.download()
err = np.linalg.norm(gv.astype(np.float64) - wv, axis=1); ref = np.maximum(np.linalg.norm(wv.astype(np.float64), axis=1), 1e-3)
ref = np.maximum(ref, np.linalg.norm(acc_c.astype(np.float64), axis=1) * 0.1)
print("worst err/scale", np.nanmax(err / ref), "p99.9", np.nanquantile(err / ref, 0.999))
bad = np.where(~(err <= 1e-5 * ref))[0]
if len(bad) == 0: sys.exit(0)
for i in bad[:8]:
    print(i, "pos", sp[i], "vel_ref", wv[i], "vel_gpu", gv[i], "err/ref", err[i]/ref[i], "acc_ref", acc_c[i], "acc_gpu", acc_g[i], "dist", field.get_obstacle_distance(sp[i]), "v0", s0[i])

print("---- pairs of agent", bad[0])
i = int(bad[0]); F = np.float32
px, py = sp[i]
g = field.get_potential_grad(int(sd[i]), sp[i]); e = g * (F(1) / np.sqrt(g[0]*g[0] + g[1]*g[1]))
d = sp - sp[i]; near = np.where((d[:,0]**2 + d[:,1]**2 <= 4.0))[0]
for j in near:
    if j == i: continue
    dx, dy = F(px - sp[j,0]), F(py - sp[j,1]); d2 = dx*dx + dy*dy
    dist = np.sqrt(d2); nx, ny = dx/dist, dy/dist
    t1x, t1y = dx - sv[j,0]*F(0.1), dy - sv[j,1]*F(0.1); l = np.sqrt(t1x*t1x + t1y*t1y); t2 = dist + l
    vl = np.sqrt(sv[j,0]**2 + sv[j,1]**2) * F(0.1); arg = t2*t2 - vl*vl; b = np.sqrt(arg) * F(0.5)
    k = F(2.1)/F(0.3) * np.exp(-b/F(0.3)); fx, fy = k*t2*(nx + t1x/l)/(4*b), k*t2*(ny + t1y/l)/(4*b)
    flen = np.hypot(fx, fy); lhs = -(e[0]*fx + e[1]*fy); rhs = flen * F(-0.17364817766693036)
    print(f"  j={j} d={dist:.5f} l={l:.5f} vl={vl:.5f} t2^2-vl^2={arg:.6g} b={b:.5f} |f|={flen:.5g} (lhs-rhs)/|f|={(lhs-rhs)/flen:.3e}")
