import sys, ctypes
mode = sys.argv[1]
if mode == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(4, device="cuda"); print("alloc ok", x.sum().item())
    lib = ctypes.CDLL("/root/repo/wave_tracer_amd/libwtgpu.so")
    hip = ctypes.CDLL("libamdhip64.so.7")
    n = ctypes.c_int(0); print("hipGetDeviceCount rc", hip.hipGetDeviceCount(ctypes.byref(n)), n.value)
elif mode == "lib_first":
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so.7")
    n = ctypes.c_int(0); print("hipGetDeviceCount rc", hip.hipGetDeviceCount(ctypes.byref(n)), n.value)
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
elif mode == "lib_only":
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so.7")
    n = ctypes.c_int(0); print("hipGetDeviceCount rc", hip.hipGetDeviceCount(ctypes.byref(n)), n.value)
with open("/proc/self/maps") as f:
    print(sorted(set(l.split()[-1] for l in f if "amdhip" in l or "hsa-runtime" in l)))
