"""development: time one linearisation of a constrained Talos-size problem (tests/problems.py names) in a given derivative mode"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ddp_pinocchio_amd import capi
from problems import make
name, S, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
T = 200
model, spec, o = make(name, T, batch=S, fd_mode=mode, first_order_fd=0 if mode == 1 else 1)
ctx = capi.Context(spec)
us = 0.1 * np.random.default_rng(0).normal(size=(S, T * o.m))
ctx.upload("X", np.zeros((S, (T + 1) * o.nx))); ctx.upload("U", us); ctx.rollout()
ctx.linearize()
t0 = time.perf_counter(); ctx.linearize(); print(name, "mode", mode, "seeds", S, "linearize ms", (time.perf_counter() - t0) * 1e3)
