import sys, os, time, contextlib
sys.path.insert(0, '/root/repo')
import torch
import bench
class A: pass
args = A(); args.batch = 256; args.algo_hparams = ''; args.prefetch = 1
cfg = bench.CONFIGS['mslr10k']
dev = torch.device('cuda', 0)
with contextlib.redirect_stdout(sys.stderr):
    data, exp, model, feed = bench.build_world(cfg, args, dev)
def step():
    f, _ = feed.get_batch(data, check_validation=True)
    return model.step(None, f, False)
for _ in range(20): step()
torch.cuda.synchronize()
for n in (10, 20, 40):
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('n=%d enqueue %.1f us/step, total %.1f us/step' % (n, (t1-t0)/n*1e6, (t2-t0)/n*1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
