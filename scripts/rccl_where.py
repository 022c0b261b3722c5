import sys
sys.path.insert(0, ".")
import torch
import meshfem_amd as M
from meshfem_amd import distributed as D
c = M.Context(0)
comm = D.Comm.rccl(c, 0, 1)
print("describe:", comm.describe())
comm.selftest()
print("selftest ok")
import subprocess, os
print(subprocess.run("grep -i rccl /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid(), shell=True, capture_output=True, text=True).stdout)
