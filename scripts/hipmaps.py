"""Debug: which HIP runtime libraries a process maps when torch is imported before / after libkomb_accel.so (the load-order
note in INTEGRATION.md section C).  usage: hipmaps.py torch_first|komb_first"""
import sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
if order == "torch_first":
    import torch; torch.cuda.is_available()
    import komb_amd; komb_amd._lib.load()
else:
    import komb_amd; komb_amd._lib.load()
    import torch; torch.cuda.is_available()
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "libhsa-runtime" in l})
print(order, libs)
