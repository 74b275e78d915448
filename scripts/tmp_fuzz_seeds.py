import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "scripts")); sys.path.insert(0, os.getcwd())
import importlib
ch = importlib.import_module("gpu_fuzz_chain")
for s in (78, 110, 123, 124, 196, 228): ch.run(1, s, 1)
