import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "board-game-simulator-python_amd")]
from simulator.game.connect import Config
import numpy as np
from simulator.game.bounce import Config as BConfig
c = Config(6,7,4)
s = c.sample_initial_state()
t0=time.perf_counter(); n=0
for g in range(20):
    s = c.sample_initial_state()
    while not s.has_ended:
        s = random.choice(s.actions).sample_next_state(); n+=1
dt=time.perf_counter()-t0
print(f"connect object API: {dt/n*1e6:.0f} us per transition ({n} transitions)")
g=np.zeros((9,6),dtype=np.int64); g[1]=g[7]=[1,2,3,3,2,1]
bc=BConfig(g); s=bc.sample_initial_state()
t0=time.perf_counter(); n=0
for k in range(5):
    s=bc.sample_initial_state(); p=0
    while not s.has_ended and p<200:
        s=random.choice(s.actions).sample_next_state(); n+=1; p+=1
dt=time.perf_counter()-t0
print(f"bounce object API: {dt/n*1e6:.0f} us per transition ({n} transitions)")
