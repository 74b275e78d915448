import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "scripts")); sys.path.insert(0, os.getcwd())
import importlib
ch = importlib.import_module("gpu_fuzz_chain"); bd = importlib.import_module("gpu_fuzz_band")
for s in (78, 228): ch.run(1, s, 1)
for s in (449, 1425): bd.run(1, s)
