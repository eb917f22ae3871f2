"""A/B builds of the library: python scratch/ab_build.py NAME -DFLAG=V ...  ->  scratch/ab/libkv_NAME.so (load it with KV_LIB_PATH)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
name, flags = sys.argv[1], sys.argv[2:]
srcs = [os.path.join(g.CSRC, s) for s in g.HIP_SOURCES]
hdrs = [os.path.join(g.CSRC, h) for h in g.HIP_HEADERS]
out = os.path.join(g.ROOT, 'scratch', 'ab', 'libkv_{}.so'.format(name))
os.makedirs(os.path.dirname(out), exist_ok=True)
g._compile(srcs, hdrs, out, False, flags)
print(out)
