"""Phase breakdown of the pivot loop from the diagnostic build (make -C blu_amd/csrc prof).
   BLU_HIP_LIB=blu_amd/libblu_hip_prof.so python tools/prof_phases.py [C2|C3] [block]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blu_amd
from blu_amd import keys as K
from blu_amd.matrices import CONFIGS
c = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
cp, ri, v = blu_amd.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"], c["offscale"])
h = blu_amd.BLU(c["m"], len(ri))
if len(sys.argv) > 2:
    h.dbg_set_block(int(sys.argv[2]))
for rep in range(2):
    st = h.factorize(cp[:-1], cp[1:], ri, v)
p = [h.stat(60 + k) for k in range(48)]
tp = h.stat(K.STAT_DEV_TIME_PIVOT_LOOP)
tot = sum(p[:4])
if tot == 0:
    print("status", st, "t_pivot %.1f ms (not a diagnostic build: no phase ticks)" % (1e3 * tp))
    sys.exit(0)
print("status", st, "t_pivot %.1f ms" % (1e3 * tp), "ticks total %.3g -> %.2f GHz-equivalent" % (tot, tot / tp / 1e9))
n1, n2, n3 = p[4], p[5], p[6]
f = tp / tot * 1e6  # us per tick
print("search+setup: %.2f us/pivot (%.0f%%)" % (p[0] * f / max(1, n1 + n2 + n3), 100 * p[0] / tot))
print("fast small  : n=%d %.2f us each (%.0f%%), of which line updates %.2f us" % (n1, p[1] * f / max(1, n1), 100 * p[1] / tot, p[7] * f / max(1, n1)))
print("fast scol   : n=%d %.2f us each (%.0f%%)" % (n2, p[2] * f / max(1, n2), 100 * p[2] / tot))
print("general     : n=%d %.2f us each (%.0f%%)" % (n3, p[3] * f / max(1, n3), 100 * p[3] / tot))
names = ["enter", "walk lists / express", "entries+rowmeta+cost", "argmin", "pivot col->LDS, row load", "col metadata+hash", "row hash+sums", "barrier"]
print("search stages (us per fast pivot): " + " | ".join("%s %.2f" % (nm, p[8 + k] * f / max(1, n1 + n2)) for k, nm in enumerate(names)))
nf = max(1, p[21])
print("inside (cycles per fast pivot): entries addr+load+drain %.0f | row metadata load+drain %.0f | LDS stores+costs %.0f || list heads %.0f | first candidate link+meta %.0f"
      % (p[16] / nf, p[17] / nf, p[18] / nf, p[19] / nf, p[20] / nf))
n1 = max(1, n1)
cyc = lambda k: p[k] / n1
print("   list wave, inside 'stores issued': unlink %.0f | tails + key groups %.0f" % (cyc(30), cyc(31)))
print("fast small, finalize step (cycles after the barrier): U row written @%.0f | L column written @%.0f | list wave: enters @%.0f, loads %.0f, runs %.0f, stores issued %.0f, drained %.0f, then %.0f until all waves are past the barrier"
      % (cyc(22), cyc(23), cyc(24), cyc(25), cyc(26), cyc(27), cyc(28), cyc(29)))
print("fast small, line updates as wave 1 sees them: %.1f column + %.1f row tasks per pivot; cycles: loads issued %.0f | arrived %.0f | task 1 %.0f | task 2 %.0f | task 3 %.0f | rest + drain %.0f | waiting for the others %.0f"
      % (cyc(32), cyc(33), cyc(34), cyc(35), cyc(36), cyc(37), cyc(38), cyc(39), cyc(40)))
ne = max(1, p[45])
print("speculative search on the unlink wave, beside the line updates (%d of %d ran to the end): unlink done @%.0f | walk %.0f | staging %.0f | reduction %.0f cycles"
      % (p[45], n1, p[41] / ne, p[42] / ne, p[43] / ne, p[44] / ne))
