import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nis-sar-amtigmti-video_amd"))
import sarx
from sarx.targets import generate_destroyer
k = sarx.batch_constants(); tg = generate_destroyer(center_pos=(0,0,0))
t_vec = np.linspace(-2.5, 2.5, 25000)[:2500]; pos, vel = sarx.orbit_arc(t_vec, k); l_ant = k["Lambda"]*k["R0"]/500.0
ctx = sarx.default_context(); d = None
def frame():
    global d
    d, t_start, n_s, v = sarx.run_physics_spotlight(tg, t_vec, pos, vel, 45.0, 15.0, l_ant, consts=k, device=True, out=d)
    n = 2500*n_s
    sp,_ = sarx.power_stats(d, n); sarx.add_noise_dev(d, n, sp, 20.0, 10.0, 1.0, seed=1)
    return sarx.tdbp_gpu(d, pos, vel, t_start, n_s, v, t_vec, 500.0, 512, 512, consts=k)
frame(); frame()
t0=time.perf_counter()
for _ in range(10): frame()
print("ms/frame", (time.perf_counter()-t0)*100)
pr=cProfile.Profile(); pr.enable()
for _ in range(10): frame()
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
