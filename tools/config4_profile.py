"""Where the host time of a configs[4] retrieval goes: cProfile over whole retrievals (the bench's own problem)."""
import copy, cProfile, gc, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as BC
from spectrobot_amd import engine, retrieval

engine.set_device(0)
scene = BC.two_gas_scene(40000, 8000, 60000, 60)
bs, pixels, x_true = BC.retrieval_problem(scene)
bs0 = copy.deepcopy(bs)
retrieval.inversion_fast_limb(scene, copy.deepcopy(bs0), pixels, max_it=20)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
starts = [copy.deepcopy(bs0) for _ in range(2 * n)]
gc.collect(); gc.freeze()
torch.cuda.synchronize()
t0 = time.perf_counter(); n_it = 0
for i in range(n):
    r = retrieval.inversion_fast_limb(scene, starts[i], pixels, max_it=20)
    n_it += len(r[3].history)
dt = time.perf_counter() - t0
print("%d retrievals, %d iterations: %.3f ms per retrieval, %.4f ms per iteration" % (n, n_it, dt / n * 1e3, dt / n_it * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    retrieval.inversion_fast_limb(scene, starts[n + i], pixels, max_it=20)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
