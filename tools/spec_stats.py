"""Speculative search of the pivot loop, counted (make -C blu_amd/csrc spst):
   BLU_HIP_LIB=$PWD/blu_amd/libblu_hip_spst.so python tools/spec_stats.py [C2|C3|C4]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blu_amd
from blu_amd.matrices import CONFIGS
c = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
cp, ri, v = blu_amd.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"], c["offscale"])
h = blu_amd.BLU(c["m"], len(ri))
b = [h.stat(100 + i) for i in range(5)]
st = h.factorize(cp[:-1], cp[1:], ri, v)
a = [h.stat(100 + i) for i in range(5)]
d = [int(x - y) for x, y in zip(a, b)]
print("status", st, "kind-1 pivots", int(h.stat(54)))
print("speculative search: started %d | walk complete (K unmoved candidates) %d | still possible after the line updates %d | became the next search %d | of which with updated columns merged in %d" % (d[0], d[1], d[2], d[3], d[4]))
