"""Profiling target: 20 whole-database ticks on one stream (run under rocprofv3 --pmc SQ_INSTS_VALU ... to get the VALU
instruction count of every kernel of a tick)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nclt_slam_project_amd.engine import Engine
import bench

e = Engine(0, 640, 480, 2048)
frames, db, base_poses = bench.build_workload(e, 10000, "fixed64", 8)
e.db_upload(*db)
fd = [e.to_device(f) for f in frames]
e.set_exclusive(False)
for i in range(20):
    e.tick_dev(fd[i % 8], 640, 480, base_poses[i % 8], False, True, i)
    e.sync()
print("done")
