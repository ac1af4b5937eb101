import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bench_configs as bc, chiron_amd as ca
for ns in (3, 4, 5, 6, 8):
    os.environ["BENCH_SLOTS"] = str(ns)
    bc.run("RNA_default seg500 jump490 b400 beam50 slots%d" % ns, ca.rna_default_spec(), 500, 490, 400, 50, steps=200)
for ns in (3, 4, 6):
    os.environ["BENCH_SLOTS"] = str(ns)
    bc.run("RNA_default seg500 jump490 b400 greedy slots%d" % ns, ca.rna_default_spec(), 500, 490, 400, 0, steps=200)
