import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ['bench_sampler.py', '--noise', 'philox', '--batch', '1', '--reps', '1', '--steps', '60']
import runpy
ns = runpy.run_path(ROOT + '/scripts/bench_sampler.py')
out = ns['out']
print('nonzero fraction of the result:', float((out != 0).float().mean()), 'absmax', float(out.abs().max()))
mc = ns['mc']; mc.num_steps = 3
o3 = mc.sample(ns['xt'], ns['model'], ns['x0'], ns['target'], ns['y'], 0.5, 0.0)
print('after 3 steps: nonzero', float((o3 != 0).float().mean()), 'absmax', float(o3.abs().max()))
