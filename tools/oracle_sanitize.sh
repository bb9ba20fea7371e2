#!/bin/bash
# The CPU oracle (the checker of every parity test) under AddressSanitizer + UBSan: the oracle-only tests and the
# first N cases of the fuzz sweep's seeds (matrix generation, factorize with growth, solves, updates), CPU only.
#   bash tools/oracle_sanitize.sh [N=300]
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-300}
make -s -C $R/oracle liborc_san.so || exit 1
export ORC_LIB=$R/oracle/liborc_san.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
cd $R
python -m pytest tests/test_oracle.py tests/test_update_oracle.py tests/test_model_vs_oracle.py -x -q -p no:cacheprovider || exit 1
python - "$N" <<'PY' || exit 1
import sys
import numpy as np
sys.path.insert(0, ".")
from blu_amd import keys as K
import tools.fuzz_gpu as fz
n = int(sys.argv[1])
for seed in (777, 12345):
    rng = np.random.default_rng(seed)
    st = {}
    for case in range(n):
        c, mat = fz.draw(rng)
        o, so = fz.oracle_of(c, mat)
        st[so] = st.get(so, 0) + 1
        if so in (K.OK, K.WARNING_SINGULAR_MATRIX):
            o.get_factors()
            for trans, b, ir, xr in fz.draw_solves(rng, c["m"]):
                o.solve_dense(b, trans)
                o.solve_sparse(ir, xr, trans)
    print("seed", seed, ":", n, "cases through the sanitized oracle, statuses", st)
PY
echo "oracle clean under ASan + UBSan"
