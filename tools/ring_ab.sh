python - <<'P' > gpurun_out/r04_ring_selftest.txt 2>&1
import os, sys, subprocess, json
code = """import sys, time; sys.path.insert(0,'.'); from c6amd_loader import load_package; p=load_package()
for geo in (dict(ids=16,waves=4,tries=8,launches=30000,seed=3), dict(ids=16,waves=8,tries=6,launches=30000,seed=4), dict(ids=24,waves=16,tries=4,launches=30000,seed=6), dict(ids=1000,waves=4096,tries=4,launches=200,seed=5), dict(ids=37,waves=2048,tries=8,launches=3000,seed=9), dict(ids=16,waves=256,tries=4,launches=20000,seed=11), dict(ids=64,waves=1024,tries=2,launches=10000,seed=13)):
    t=time.time(); r=p.pool_ring_selftest(**geo); print(geo, '->', r, round(time.time()-t,2), 's')
"""
for name, lib in (("credit protocol (this round's ring_pop)", None), ("add-then-undo (round 3's ring_pop, tools/ab_patches/tree_ring_pop_add_undo.sed)", os.getcwd() + "/tools/ab/libold_ring.so")):
    env = dict(os.environ)
    if lib: env["C6_LIB_PATH"] = lib
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    print(name); print(r.stdout.strip(), r.stderr.strip()[-300:])
P
cat gpurun_out/r04_ring_selftest.txt
