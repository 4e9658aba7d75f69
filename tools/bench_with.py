"""bench.py with module-level switches of 3dssd_amd.utils.layers_util set first:  python tools/bench_with.py MLP_GRANULE4=True -- <bench args>"""
import ast, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
i = sys.argv.index("--")
lu = importlib.import_module("3dssd_amd.utils.layers_util")
for kv in sys.argv[1:i]:
    k, v = kv.split("=")
    setattr(lu, k, ast.literal_eval(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[i + 1:]
import bench
bench.main()
