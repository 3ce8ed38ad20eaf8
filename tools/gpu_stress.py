import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.chdir("/root/repo/tests")
import test_gpu_pipeline as P, test_gpu_regions as R
kind, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
P._check(kind, n, seed)
R._check(kind, max(200, n // 4), seed + 1)
print("ok", kind, n, seed, {k: v for k, v in os.environ.items() if k.startswith("EMA_")})
