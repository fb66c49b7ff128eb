"""Which RCCL (and how many) a process ends up with when it uses the library's communicator: `none` = no torch in the process,
`torch_first` = torch imported before the communicator is created (bench.py's order), `torch_after` = the order that maps two
copies (the system's and PyTorch's) and aborts at exit with a double free: what csrc/tz_comm.cpp's choice of library avoids.
Prints the librccl mappings of the process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
if order == "torch_first":
    import torch
from takzero_amd import comm as CM
c = CM.Comm.rccl(CM.unique_id(), 0, 1, 0)
print(c.all_gather(b"abc"), c.info())
c.close()
if order == "torch_after":
    import torch
maps = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l]
print(order, sorted(set(maps)))
