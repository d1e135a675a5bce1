import importlib, json, os, subprocess, sys
ROOT = "/root/repo"
code = """
import importlib, sys, json
sys.path.insert(0, {root!r}); sys.argv = ['measure_misc.py', 'batch', {m!r}, '4096', '0', '0']
capi = importlib.import_module('sfm-gms_amd.capi'); capi.library_path = lambda: {lib!r}
__file__ = {root!r} + '/tools/measure_misc.py'
exec(open(__file__).read())
"""
for rnd in range(2):
    for lib in sys.argv[1:]:
        for m in ("16384", "12000"):
            r = subprocess.run([sys.executable, "-c", code.format(root=ROOT, m=m, lib=ROOT + "/sfm-gms_amd/csrc/" + lib)], capture_output=True, text=True)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1]); print(lib, m, round(d["pairs_per_s"]))
            except Exception:
                print(lib, m, "failed", r.stderr[-300:])
