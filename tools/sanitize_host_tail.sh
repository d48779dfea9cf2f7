#!/bin/bash
# Sanitizers over the library's host side (mcfost_amd/csrc/host_tail.cpp, through tests/emu/emu_host_tail.cpp), CPU only:
#   tools/sanitize_host_tail.sh [thread|address|undefined ...]      (default: all three)
# Whole packets of frozen and live runs (2D, 3D, with the random walk) on 4 host threads.  ThreadSanitizer is the one that
# matters here: the packets of a job share E_abs, the SED bins and the counters.
R=$(cd "$(dirname "$0")/.." && pwd)
KINDS=${@:-thread address undefined}
cat > /tmp/san_host_tail.py <<'PY'
import sys, ctypes as C
R = sys.argv[2]
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
from mcfost_amd.host import model as M
from oracle import Oracle
from oracle.binding import _Opts, _p, N_COUNTERS
import test_host_tail as H
lib = C.CDLL(sys.argv[1])
quick = len(sys.argv) > 3   # (under ThreadSanitizer: the small 2D grid alone -- what the threads share does not depend on the grid)
cases = [(M.build_model(M.small()), 3000)]
if not quick:
    from test_mrw import thick_disk, thick_disk_3d
    cases += [(M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True)), 2000), (thick_disk(), 1500), (thick_disk_3d(), 1500)]
for m, n in cases:
    orc = Oracle(m, n)
    prior = orc.run_thermal(2000, seed=1, n_threads=1)["E_abs"]
    print("frozen", H.run(lib, orc, n, 7, prior, 4)["counters"])
    E = np.zeros(m.n_cells); sed = np.zeros((9, m.cfg.N_phi, m.cfg.N_thet, m.n_lambda)); ns = np.zeros(m.n_lambda); cnt = np.zeros(N_COUNTERS, np.uint64)
    o = _Opts(7, 0, n, 1, 0, 0, 1.0)   # live: Temp_LTE reads E_abs while other threads fold their deposits into it
    ms = C.c_double(0)
    rc = lib.emu_host_tail_thermal(C.byref(orc.cm), C.byref(o), None, _p(E, C.c_double), _p(sed, C.c_double), _p(ns, C.c_double), _p(cnt, C.c_uint64), 4, C.byref(ms))
    print("live  ", rc, [int(c) for c in cnt])
PY
RC=0
for K in $KINDS; do
  g++ -O1 -g -std=c++17 -fPIC -ffp-contract=fast -mfma -Wno-unknown-pragmas -pthread -fsanitize=$K -fno-omit-frame-pointer -shared \
      -o /tmp/libemu_host_tail_$K.so "$R/tests/emu/emu_host_tail.cpp" || exit 1
  LIB=$(gcc -print-file-name=lib$([ $K = thread ] && echo tsan || ([ $K = address ] && echo asan || echo ubsan)).so)
  OUT=$(LD_PRELOAD=$LIB TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
        timeout 600 python -u /tmp/san_host_tail.py /tmp/libemu_host_tail_$K.so "$R" $([ $K = thread ] && echo quick) 2>&1)
  N=$(echo "$OUT" | grep -c -E "WARNING: ThreadSanitizer|ERROR: AddressSanitizer|runtime error")
  echo "sanitizer ($K): $N report(s); $(echo "$OUT" | grep -c -E '^frozen|^live') runs"
  echo "$OUT" | grep -E "SUMMARY|runtime error" | sort | uniq -c | head -5
  [ "$N" = 0 ] || RC=1
done
exit $RC
