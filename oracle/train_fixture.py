"""ORACLE support - build-container only.  Short-trains the REAL reference (main.py, unmodified)
to obtain BER-meaningful weights, because the upstream pretrained models are absent from the
mount (.MISSING_LARGE_BLOBS).  Run from a scratch directory (main.py writes ./logs and ./tmp):

    cd /tmp/trainrun && OMP_NUM_THREADS=4 python /root/repo/oracle/train_fixture.py <extra main.py flags>

The resulting ./tmp/torch_model_<id>.pt is converted to a fixture by oracle/make_golden.py.
"""
import math, fractions, os, runpy, sys
import numpy as np
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
if not hasattr(np, "complex"):
    np.complex = complex
if not hasattr(fractions, "gcd"):
    fractions.gcd = math.gcd
os.makedirs("logs", exist_ok=True)
os.makedirs("tmp", exist_ok=True)
sys.argv = ["main.py"] + sys.argv[1:]
runpy.run_path("/root/reference/main.py", run_name="__main__")
