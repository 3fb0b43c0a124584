import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
g.build()
from oracle.harness import console_of, CONSOLE_CASES, lib_path
for problem, fd in CONSOLE_CASES:
    hip = os.path.join(os.path.dirname(lib_path("oracle")), "libdrv_%s_fd%d_hip.so" % (problem, fd))
    open("gpurun_out/console_%s_fd%d.txt" % (problem, fd), "w").write(console_of(hip, problem, fd))
