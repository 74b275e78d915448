import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "scripts")); sys.path.insert(0, os.getcwd())
import importlib
bd = importlib.import_module("gpu_fuzz_band")
for s in (449, 513, 1004, 1055, 1330, 1425): bd.run(1, s)
