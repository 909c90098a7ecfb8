"""per-kernel timing of the attention backward (delta, dQ, dK/dV) via rocprof-free events around separate generations; prints total only"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["lm"])
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_attn2.py")).read())
